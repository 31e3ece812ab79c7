// Gram form of the recurrent cell: ONE launch per layer-step.
//
// The reference's layer is relu(p U_k + h S_k + x Wk_k + b_k) with the materialised
// S_k = (I - (Dn_k/alpha_k)^T Dn_k)^T (custom_layers.py:361-369, enhance.py:172-181).  With
// G_k = Dn_k^T Dn_k (symmetric, built by drnmf_prepare_params) and everything that does not depend
// on the chain hoisted out of it --
//     c_k[t] = (x_t Dn_k) / alpha_k + b_k          one frame-parallel GEMM per layer (gemm_nt.h)
// -- a layer-step is
//     layer 0 : h_0 = relu(q + u0o sum(p)),   q = (u0d - u0o) p + c_0[t]        (elementwise)
//     layer k : h_k = relu(h_{k-1} - (h_{k-1} G_k) / alpha_k + c_k[t] + uko sum(p))
// i.e. one B x N x N contraction, against two dependent B x F x N contractions (two launches) of
// the factored form.  It executes 2 N^2 instead of 4 F N flops per row and streams an N x N matrix
// per layer, so it pays where a launch is latency rather than work: small batches (the shipped
// training configuration, B = 32) -- gram_wanted() below, measured.  Layer 0 has no contraction
// (U_0 is diagonal + constant): it is folded into the first launch of the frame as a transform of
// the A operand, and its t-dependent part q is left at a fixed address by the previous frame's last
// launch, so that no kernel has a frame-index-dependent load ahead of its contraction.
//
// The backward chain has the same shape: dh_{k-1} = dz_k - (dz_k / alpha_k) G_k.
//
// Kernel = cell_b_kernel's contraction (16 rows x 16 output atoms per workgroup, 8 waves split the
// N input atoms, operands straight from the tile-packed buffers into MFMA registers) + an update
// epilogue in cell_a_kernel's role.
#pragma once
#include "cell_shared.h"

namespace {

constexpr int NW_G = 8;

struct GramFwdArgs {
    const float* G;          // packed G_k [Np/16][Np/16][256]
    const float* a_in;       // packed [Bp][Np]: q (first launch of a frame) or h_{k-1}
    const float* ia;         // [Np] 1/alpha_k
    const float* Cp;         // packed c, ring over frames: [t mod 2 GRAM_TB][K][Bp/16][Np/16][256]
    float* h_out;            // packed h_k (last layer: unused)
    float* state;            // packed state
    float* qnext;            // packed q of the NEXT frame (written by the last layer)
    float* rs_part;          // [2][numO][Bp] row sums of the state per output tile, by frame parity
    float* psum;             // [Bp]
    float* psum_all;         // [T][Bp]
    const unsigned char* valid;
    float* out;              // [B][T][out_width]
    const int* t_rd;
    int* t_wr;
    int t_wr_add;
    float u0d, u0o, uko;
    int B, T, N, K, k, Bp, Np, numO;
    int out_width, all_hidden;
    int par;                 // parity of this node's frame index (static: graphs hold an even number
                             // of frames), selects the rs_part / q buffers without waiting for t
    int cp_mask;             // frame t's c_k live in slot t & cp_mask of Cp (ring of 2 GRAM_TB frames:
                             // 2 GRAM_TB - 1; all T frames resident: 0x7fffffff)
};

// Contraction shared by the forward and backward kernels: s(row, o) = sum_i A[row][i] G[o][i] for
// the workgroup's 16 rows x 16 outputs; returns the element (erow = tid/16, ecol = tid%16) to the
// first 256 threads.  FIRST: the A operand is relu(a_in + add[row]) (layer 0 on the fly).
// `mid()` runs after the first operand loads have been issued (the first launch of a frame reduces
// sum(p) there: its own loads were issued BEFORE, so the in-order vmcnt wait covers only them).
template <int GS, bool FIRST, class Mid>
__device__ __forceinline__ float gram_contract(const float* __restrict__ G,
                                               const float* __restrict__ a_in, const float* add16,
                                               int m, int ot, int NAC, float* red, Mid mid) {
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63, j = l & 15, q = l >> 4;
    const float* arow = a_in + (size_t)m * NAC * 256 + l * 4;
    const float* brow = G + (size_t)ot * NAC * 256 + l * 4;
    const int clast = NAC - 1;
    const int per_wave = (NAC - w + NW_G - 1) / NW_G;
    f32x4 av[GS], bv[GS];
    auto load_slot = [&](int i, int g) {
        int c = w + NW_G * i;
        c = c > clast ? clast : c;
        av[g] = *(const f32x4*)(arow + 256 * c);
        bv[g] = *(const f32x4*)(brow + 256 * c);
    };
#pragma unroll
    for (int g = 0; g < GS - 1; ++g) load_slot(g, g);
    __builtin_amdgcn_sched_barrier(0);
    mid();
    float addv = 0.f;
    if (FIRST) addv = add16[j];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base < per_wave; base += GS) {
#pragma unroll
        for (int g = 0; g < GS; ++g) {
            load_slot(base + g + GS - 1, (g + GS - 1) % GS);
            __builtin_amdgcn_sched_barrier(0);
            const bool ok = base + g < per_wave;
            f32x4 a4 = av[g];
            if (FIRST) {
#pragma unroll
                for (int e = 0; e < 4; ++e) a4[e] = fmaxf(a4[e] + addv, 0.f);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float a1 = ok ? a4[s] : 0.f;
                if (s & 1) acc1 = mfma16(a1, bv[g][s], acc1);
                else acc0 = mfma16(a1, bv[g][s], acc0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) red[(w * 16 + 4 * q + v) * 17 + j] = acc0[v] + acc1[v];
    __syncthreads();
    float s = 0.f;
    if (tid < 256) {
        const int erow = tid >> 4, ecol = tid & 15;
#pragma unroll
        for (int ww = 0; ww < NW_G; ++ww) s += red[(ww * 16 + erow) * 17 + ecol];
    }
    return s;
}

// BPTT, bottom of a frame: gradient w.r.t. the state p that entered it, from dz_0 (this atom), the
// row sum s0 of dz_0 and sp = uko * sum_{k>=1} rowsum(dz_k).  One fixed sequence of fused operations,
// shared by bwd_edge_kernel and the persistent chain (cell_gram_persist.h): the compiler's own
// choice of contractions differs between the two kernels and the results must not.
__device__ __forceinline__ float bptt_state_grad(float u0d, float u0o, float dz0, float s0, float sp) {
    return fmaf(u0d, dz0, fmaf(u0o, s0 - dz0, sp));
}

// One layer-step k >= 1 of one frame (FIRST: k == 1, carries layer 0; LAST: k == K-1).
template <int GS, bool FIRST, bool LAST>
__global__ void __launch_bounds__(64 * NW_G) gram_fwd_kernel(const GramFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float red[NW_G * 16 * 17];
    __shared__ float part[32][17];
    __shared__ float ps16[16], psv[16];
    const int m = blockIdx.x >> 3;                          // grid layout: see cell_a_kernel
    const int ot_raw = blockIdx.y * 8 + (blockIdx.x & 7);
    const bool live = ot_raw < a.numO;
    const int ot = live ? ot_raw : a.numO - 1;
    const int tid = threadIdx.x;
    const int NAC = a.Np / 16;
    const int t = *a.t_rd;
    if (a.t_wr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.t_wr = t + a.t_wr_add;

    // Epilogue operands: nothing the update needs may wait for the contraction to finish.  What
    // does not depend on the frame index is loaded first (ahead of the MFMA operands: the in-order
    // vmcnt wait of the first MFMA then covers a handful of small loads more); what does (c_k[t],
    // the validity flag) is loaded in `mid`, behind the first operand loads, when the scalar load
    // of t has long returned.
    const bool ethr = tid < 256;
    const int erow = (tid & 255) >> 4, ecol = tid & 15;
    const int rg = m * 16 + erow, n = ot * 16 + ecol;
    const size_t hoff = ((size_t)m * NAC + ot) * 256 + hp_pos(erow, ecol);
    const size_t cstride = (size_t)a.Bp * a.Np;                       // one (t, k) slice of Cp
    float hraw = 0.f, iav = 0.f, psl = 0.f, st_old = 0.f, ck = 0.f, cnext = 0.f;
    bool vld = true;
    // sum(p) of the 16 rows (first launch of a frame): the row sums left per output tile by the
    // previous frame's last layer, added in a fixed order (deterministic); reduced in `mid`.
    float rsum = 0.f;
    if (FIRST) {
        const int row = tid & 15, pt = tid >> 4;            // 32 parts
        const float* rp = a.rs_part + (size_t)a.par * a.numO * a.Bp + m * 16 + row;
        for (int b2 = pt; b2 < a.numO; b2 += 32) rsum += rp[(size_t)b2 * a.Bp];
    }
    if (ethr) {
        hraw = a.a_in[hoff];
        iav = a.ia[n];
        if (!FIRST) psl = a.psum[rg];
        if (LAST) st_old = a.state[hoff];
    }
    auto mid = [&]() {
        if (ethr) {
            ck = a.Cp[((size_t)(t & a.cp_mask) * a.K + a.k) * cstride + hoff];
            vld = a.valid[(size_t)t * a.Bp + rg] != 0;
            if (LAST) {
                const int tn = t + 1 < a.T ? t + 1 : t;
                cnext = a.Cp[((size_t)(tn & a.cp_mask) * a.K) * cstride + hoff];
            }
        }
        if (!FIRST) return;
        part[tid >> 4][tid & 15] = rsum;
        __syncthreads();
        if (tid < 16) {
            float tot = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) tot += part[i][tid];
            ps16[tid] = a.u0o * tot;
            psv[tid] = tot;
        }
        __syncthreads();
    };
    const float s = gram_contract<GS, FIRST>(a.G, a.a_in, ps16, m, ot, NAC, red, mid);
    if (!ethr) return;

    const float ps = FIRST ? psv[erow] : psl;
    if (FIRST && ot_raw == 0 && ecol == 0) {
        a.psum[rg] = ps;
        a.psum_all[(size_t)t * a.Bp + rg] = ps;
    }
    float hprev = hraw;
    if (FIRST) hprev = fmaxf(hprev + a.u0o * ps, 0.f);                // h_0
    const float pre = hprev - s * iav + ck + a.uko * ps;
    const float hn = fmaxf(pre, 0.f);
    if (live && rg < a.B && n < a.N) {
        // K.rnn masking: a masked step repeats the previous output (zeros before the first valid)
        float* orow = a.out + ((size_t)rg * a.T + t) * a.out_width;
        if (a.all_hidden) {
            if (FIRST) {
                float o0 = hprev;
                if (!vld) o0 = (t > 0) ? orow[n - (ptrdiff_t)a.out_width] : 0.f;
                orow[n] = o0;
            }
            float o = hn;
            const int off = a.k * a.N + n;
            if (!vld) o = (t > 0) ? orow[off - (ptrdiff_t)a.out_width] : 0.f;
            orow[off] = o;
        } else if (LAST) {
            float o = hn;
            if (!vld) o = (t > 0) ? orow[n - (ptrdiff_t)a.out_width] : 0.f;
            orow[n] = o;
        }
    }
    if (!live) return;
    if (LAST) {
        const float st = vld ? hn : st_old;                           // a masked step keeps the state
        a.state[hoff] = st;
        const float rs = row16_sum(st);
        if (ecol == 0) a.rs_part[((size_t)(a.par ^ 1) * a.numO + ot) * a.Bp + rg] = rs;
        // q of the next frame: everything of its layer 0 except u0o * sum(p)
        a.qnext[hoff] = (a.u0d - a.u0o) * st + cnext;
    } else {
        a.h_out[hoff] = hn;
    }
}

// q of frame 0 from the initial state (prologue), one thread per packed element
__global__ void __launch_bounds__(256)
gram_init_q_kernel(const float* __restrict__ state, const float* __restrict__ Cp,
                   float* __restrict__ q, float u0d, float u0o, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) q[i] = (u0d - u0o) * state[i] + Cp[i];
}

// c_k[t] = (x_t Dn_k) / alpha_k + b_k for one block of frames, stored packed: gemm_nt epilogue,
// row = b * TBc + (t - t0), col = atom; Cp points at ring slot (t0 mod ring) of layer k
struct EpiCPack {
    float* Cp;
    const float* ia;
    const float* bias;
    int TBc, K, NAC;         // frames in this block
    size_t cstride;          // Bp * Np
    __device__ f32x2 pre(int64_t, int col) const { return f32x2{ia[col], bias[col]}; }
    __device__ void operator()(int64_t row, int col, float acc, f32x2 pv) const {
        const int b = (int)(row / TBc), tl = (int)(row - (int64_t)b * TBc);
        Cp[(size_t)tl * K * cstride + ((size_t)(b >> 4) * NAC + (col >> 4)) * 256 +
           hp_pos(b & 15, col & 15)] = acc * pv[0] + pv[1];
    }
};

// The same for ALL K layers in one product (untied dictionaries: the K matrices Dn_k^T lie one
// behind the other in the prepared block, so [x] . [Dn_0^T; ...; Dn_{K-1}^T]^T is one GEMM with
// K * Np columns): col = k * Np + atom, ia / bias are the [K][Np] arrays.  K launches of a few
// workgroups each per block of frames become one.
struct EpiCPackAll {
    float* Cp;
    const float* ia;
    const float* bias;
    int TBc, K, NAC, Np;
    size_t cstride;
    __device__ f32x2 pre(int64_t, int col) const { return f32x2{ia[col], bias[col]}; }
    __device__ void operator()(int64_t row, int col, float acc, f32x2 pv) const {
        const int b = (int)(row / TBc), tl = (int)(row - (int64_t)b * TBc);
        const int k = col / Np, n = col - k * Np;
        Cp[((size_t)tl * K + k) * cstride + ((size_t)(b >> 4) * NAC + (n >> 4)) * 256 +
           hp_pos(b & 15, n & 15)] = acc * pv[0] + pv[1];
    }
};

// A tied dictionary: x Dn is the same product for every layer, only 1/alpha_k and b_k differ -- one
// GEMM of Np columns whose epilogue writes all K layers.
struct EpiCPackTied {
    float* Cp;
    const float* ia;         // [K][Np]
    const float* bias;       // [K][Np]
    int TBc, K, NAC, Np;
    size_t cstride;
    static constexpr bool EARLY = false;
    __device__ f32x2 pre(int64_t, int) const { return f32x2{0.f, 0.f}; }
    __device__ void operator()(int64_t row, int col, float acc, f32x2) const {
        const int b = (int)(row / TBc), tl = (int)(row - (int64_t)b * TBc);
        float* dst = Cp + (size_t)tl * K * cstride + ((size_t)(b >> 4) * NAC + (col >> 4)) * 256 +
                     hp_pos(b & 15, col & 15);
        for (int k = 0; k < K; ++k)
            dst[(size_t)k * cstride] = acc * ia[(size_t)k * Np + col] + bias[(size_t)k * Np + col];
    }
};

// x[b][t0 .. t0+TBc)[F] -> xblk[b][tl][Fp] (padding bins zero)
__global__ void __launch_bounds__(256)
gather_block_kernel(const float* __restrict__ x, float* __restrict__ xblk, int B, int T, int F,
                    int Fp, int t0, int TBc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * TBc * Fp) return;
    const int f = (int)(i % Fp);
    const size_t r = i / Fp;
    const int tl = (int)(r % TBc), b = (int)(r / TBc);
    xblk[i] = f < F ? x[((size_t)b * T + t0 + tl) * F + f] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// backward chain
struct GramBwdArgs {
    const float* G;          // packed G_k
    const float* dGp_in;     // packed dG_k = dz_k / alpha_k
    const float* dzp_in;     // packed dz_k
    const float* ia_prev;    // [Np] of layer k-1
    float* dzp_out;          // packed dz_{k-1}
    float* dGp_out;          // packed dG_{k-1}
    const float* hall;
    float* dz_all;
    float* dz0s_part;        // [2][numO][Bp]
    float* dps_part;
    const int* c_rd;
    int* c_wr;
    float uko;
    int k, B, T, N, K, Bp, Np, numO;
};

template <int GS>
__global__ void __launch_bounds__(64 * NW_G) gram_bwd_kernel(const GramBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float red[NW_G * 16 * 17];
    const int m = blockIdx.x >> 3;
    const int ot_raw = blockIdx.y * 8 + (blockIdx.x & 7);
    const bool live = ot_raw < a.numO;
    const int ot = live ? ot_raw : a.numO - 1;
    const int tid = threadIdx.x;
    const int NAC = a.Np / 16;
    const int cnt = *a.c_rd;
    if (a.c_wr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.c_wr = cnt + 1;
    // (epilogue operands early, as in gram_fwd_kernel)
    const bool ethr = tid < 256;
    const int erow = (tid & 255) >> 4, ecol = tid & 15;
    const int rg = m * 16 + erow, n = ot * 16 + ecol;
    const size_t hoff = ((size_t)m * NAC + ot) * 256 + hp_pos(erow, ecol);
    const int KN = a.K * a.N;
    const bool in = rg < a.B && n < a.N;
    float dzk = 0.f, iap = 0.f, hprev = 0.f, dps_old = 0.f;
    const size_t po = ((size_t)(cnt & 1) * a.numO + ot) * a.Bp + rg;
    if (ethr) {
        dzk = a.dzp_in[hoff];
        iap = a.ia_prev[n];
        if (ecol == 0) dps_old = a.dps_part[po];   // (read-modify-write: the read goes out with the rest)
    }
    auto mid = [&]() {
        const int t_ = a.T - 1 - cnt;
        if (ethr && in) hprev = a.hall[((size_t)rg * a.T + t_) * KN + (size_t)(a.k - 1) * a.N + n];
    };
    const float s = gram_contract<GS, false>(a.G, a.dGp_in, nullptr, m, ot, NAC, red, mid);
    if (!ethr || !live) return;
    const int t = a.T - 1 - cnt;
    const float dzn = hprev > 0.f ? dzk - s : 0.f;
    if (in) a.dz_all[((size_t)rg * a.T + t) * KN + (size_t)(a.k - 1) * a.N + n] = dzn;
    a.dzp_out[hoff] = dzn;
    a.dGp_out[hoff] = dzn * iap;
    const float sk = row16_sum(dzk), s0 = row16_sum(dzn);
    if (ecol == 0) {
        a.dps_part[po] = fmaf(a.uko, sk, dps_old);
        if (a.k == 1) a.dz0s_part[po] = s0;
    }
}

// Dn (cell_b packing) * ia[n] -> row-major [Fp][Np]: operand of d R_k = (dz_k / alpha_k) Dn_k^T
__global__ void __launch_bounds__(256)
unpack_scaled_kernel(const float* __restrict__ Dp, const float* __restrict__ ia,
                     float* __restrict__ out, int Fp, int Np) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)Fp * Np) return;
    const int f = (int)(i / Np), n = (int)(i % Np);
    out[i] = ia[n] * Dp[((size_t)(f >> 4) * (Np / 16) + (n >> 4)) * 256 +
                        (((n & 15) >> 2) * 16 + (f & 15)) * 4 + (n & 3)];
}

void* pick_gram_fwd(int NAC, bool first, bool last) {
    const int per_wave = (NAC + NW_G - 1) / NW_G;
    if (per_wave <= 4) {
        if (first) return last ? (void*)&gram_fwd_kernel<4, true, true> : (void*)&gram_fwd_kernel<4, true, false>;
        return last ? (void*)&gram_fwd_kernel<4, false, true> : (void*)&gram_fwd_kernel<4, false, false>;
    }
    if (first) return last ? (void*)&gram_fwd_kernel<8, true, true> : (void*)&gram_fwd_kernel<8, true, false>;
    return last ? (void*)&gram_fwd_kernel<8, false, true> : (void*)&gram_fwd_kernel<8, false, false>;
}
void* pick_gram_bwd(int NAC) {
    return (NAC + NW_G - 1) / NW_G <= 4 ? (void*)&gram_bwd_kernel<4> : (void*)&gram_bwd_kernel<8>;
}

}  // namespace
