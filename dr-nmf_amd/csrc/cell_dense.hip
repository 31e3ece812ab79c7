// General (dense-matrix) SimpleDeepRNN cell forward on gfx950.
//
// Reference semantics: SimpleDeepRNN.step (custom_layers.py:343-375) as written, for ANY per-layer
// matrices -- the ones a caller's own maps_from_alt produce, directly trainable W/U/b/S weights
// (custom_layers.py:234-287), or build_alt's maps once log_U1 / log_Uk have been trained away from
// their rank-structured initialisation:
//     h_k = act( p U_k + [k > 0] h_{k-1} S_{k-1} + [connect] x_t Wk_k + b_k ),   p = h_{K-1} of frame t-1
// (the factored kernels of cell_forward.hip cover the build_alt form with untrained U only).
// Masking follows K.rnn as in cell_forward.hip; `flag_connect_input_to_layers = False` drops the
// input from EVERY layer, layer 0 included (custom_layers.py:366-368).
//
// One layer-step is ONE launch: the three products are a single contraction of the row block
// [p | h_{k-1} | x_t] (16 rows x L, L = Np + Np + Fp) with the stacked matrix [U_k; S_{k-1}; Wk_k]
// (L x N), which `drnmf_dense_prepare_params` stores once in the cell_a operand packing (common.h:
// 512-float blocks of 16 contraction rows x 32 atoms, two 16-byte loads per lane).  Activations
// (state, hidden ping-pong, packed input) are kept in the A-operand order (rp_pos) so that every
// operand load is one contiguous 1 KB per wave.  Workgroup = 16 rows x 32 atoms, the contraction
// split over 4 waves and reduced through LDS, as cell_a_kernel; 2*B*(2N+F)*N flops per launch.
// K launches per frame, several frames per cached hipGraph.
#include "cell_shared.h"

namespace {

struct DenseArgs {
    const float* M;          // stacked matrix of this layer, cell_a packing: block (c, ab) of 512 floats
                             // (HALF: fp16, blocks (c32, ab) of 2 x 512 halves, pack_dense16_kernel)
    const float* bias;       // [Np]
    const float* xp;         // [T][Bp][Fp] packed input (pack_input_kernel)
    const float* h_in;       // [Bp][Np] packed h of layer k-1 (unused for layer 0)
    float* h_out;            // [Bp][Np] packed h of this layer (unused for the last layer)
    float* state;            // [2][Bp][Np] packed recurrent state, by frame parity
    const unsigned char* valid;   // [T][Bp]
    float* out;              // [B][T][out_width]
    const int* t_rd;
    int* t_wr;
    int t_wr_add;
    int B, T, N, Bp, Fp, Np, numA;
    int nP, nH, nX;          // 16-wide chunks of the contraction per segment (nH = 0: layer 0)
    int act;                 // DRNMF_ACT_*
    int out_width, out_off;
    const float* drop;       // [B][N] recurrent dropout mask B_U (custom_layers.py:377-384: 0 or 1/(1-p),
                             // one per sequence and atom, constant over frames and layers); ones without
};

__device__ __forceinline__ float activate(float v, int act) {
    switch (act) {
        case DRNMF_ACT_RELU: return fmaxf(v, 0.f);
        case DRNMF_ACT_TANH: return tanhf(v);
        case DRNMF_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case DRNMF_ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));
        case DRNMF_ACT_HARD_SIGMOID: return fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f);
        default: return v;
    }
}

// G operand slots per wave, prefetch distance G-1 (same rotation as cell_a_kernel); NW waves split
// the contraction (the stacked matrices do not fit the Infinity Cache at the large shapes -- 36 MB
// per layer at F=513, N=2000 -- so the operand stream comes from HBM and needs many loads in flight)
// HALF (drnmf_dense_desc_t.operand_f16): the matrices are stored as fp16 in the B-operand order of
// v_mfma_f32_16x16x32_f16 for 32-row chunks of the contraction; the activations stay in their fp32 blocks
// (they are a few % of the bytes) and are rounded to fp16 in registers: a 32-row chunk is two 16-row blocks,
// lane (q, j) holds k = 16 b + 4 s + q (b = 0, 1; s = 0..3) of row j -- the matrix packing puts the same
// eight rows, in that order, into lane (q, atom) (nP / nH / nX then count 32-row chunks).
template <int G, int NW, bool IS_LAST, bool WRITE_OUT, bool HALF = false>
__global__ void __launch_bounds__(64 * NW) dense_step_kernel(const DenseArgs a) {
    __shared__ __attribute__((aligned(16))) float red[NW * ROWS * ATOMS];
    const int mb = blockIdx.x >> 3;                           // grid layout: see cell_a_kernel
    const int ab_raw = blockIdx.y * 8 + (blockIdx.x & 7);
    const bool live = ab_raw < a.numA;
    const int ab = live ? ab_raw : a.numA - 1;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63, j = l & 15, q = l >> 4;
    const int Np = a.Np, Fp = a.Fp;

    const int t = *a.t_rd;
    if (a.t_wr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.t_wr = t + a.t_wr_add;

    const size_t sstride = (size_t)a.Bp * Np;
    const float* pseg = a.state + (size_t)(t & 1) * sstride + (size_t)mb * (Np / 16) * 256 + l * 4;
    const float* hseg = a.h_in + (size_t)mb * (Np / 16) * 256 + l * 4;
    const float* xseg = a.xp + (size_t)t * a.Bp * Fp + (size_t)mb * (Fp / 16) * 256 + l * 4;
    const float* brow = a.M + (size_t)ab * 512 + l * 4;
    const size_t bstep = (size_t)a.numA * 512;
    const f16* brow16 = (const f16*)a.M + (size_t)ab * 1024 + l * 8;      // HALF: (c32, ab) = 2 x 512 halves
    const size_t bstep16 = (size_t)a.numA * 1024;
    const int nPH = a.nP + a.nH, nch = nPH + a.nX;
    const int per_wave = (nch - w + NW - 1) / NW, clast = nch - 1;
    const int xlast16 = Fp / 16 - 1;                 // HALF: the last 16-row block of x (Fp / 16 may be odd)

    f32x4 av[G][HALF ? 2 : 1], bv[HALF ? 1 : G][2];
    f16x8 bh[HALF ? G : 1][2];
    auto load_chunk = [&](int i, int g) {
        int c = w + NW * i;
        c = c > clast ? clast : c;
        if (HALF) {
            const float *ap0, *ap1;
            if (c < a.nP) { ap0 = pseg + 512 * c; ap1 = ap0 + 256; }
            else if (c < nPH) { ap0 = hseg + 512 * (c - a.nP); ap1 = ap0 + 256; }
            else {      // (an odd block count: the partner of the last block re-reads it against zero matrix rows)
                const int b0 = 2 * (c - nPH), b1 = b0 + 1 > xlast16 ? xlast16 : b0 + 1;
                ap0 = xseg + 256 * b0; ap1 = xseg + 256 * b1;
            }
            av[g][0] = *(const f32x4*)ap0;
            av[g][HALF ? 1 : 0] = *(const f32x4*)ap1;
            bh[g][0] = *(const f16x8*)(brow16 + (size_t)c * bstep16);
            bh[g][1] = *(const f16x8*)(brow16 + (size_t)c * bstep16 + 512);
        } else {
            const float* ap = c < a.nP ? pseg + 256 * c
                                       : (c < nPH ? hseg + 256 * (c - a.nP) : xseg + 256 * (c - nPH));
            av[g][0] = *(const f32x4*)ap;
            bv[g][0] = *(const f32x4*)(brow + (size_t)c * bstep);
            bv[g][1] = *(const f32x4*)(brow + (size_t)c * bstep + 256);
        }
    };
    constexpr int PF = G - 1;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        av[g][0] = av[g][HALF ? 1 : 0] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (HALF) bh[g][0] = bh[g][1] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        else bv[g][0] = bv[g][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int g = 0; g < PF; ++g) load_chunk(g, g);

    // epilogue operands, issued behind the first GEMM operands
    const int erow = (tid & 255) >> 4, ec = (tid & 15) * 2;
    const int n = ab * ATOMS + ec;
    const int rg = mb * ROWS + erow;
    const f32x2 bs = *(const f32x2*)(a.bias + n);
    const bool vld = a.valid[(size_t)t * a.Bp + rg] != 0;
    const size_t hblk = ((size_t)mb * (Np / 16) + (n >> 4)) * 256;
    const int pos0 = rp_pos(erow, n & 15), pos1 = rp_pos(erow, (n & 15) + 1);

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base < per_wave; base += G) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            load_chunk(base + g + PF, (g + PF) % G);
            __builtin_amdgcn_sched_barrier(0);
            const bool ok = base + g < per_wave;
            if (HALF) {
                const f32x4 a0 = av[g][0], a1 = av[g][HALF ? 1 : 0];
                f16x8 a8 = {(f16)a0[0], (f16)a0[1], (f16)a0[2], (f16)a0[3],
                            (f16)a1[0], (f16)a1[1], (f16)a1[2], (f16)a1[3]};
                if (!ok) a8 = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                acc0 = mfma32h(a8, bh[g][0], acc0);
                acc1 = mfma32h(a8, bh[g][1], acc1);
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float r = ok ? av[g][0][s] : 0.f;
                    acc0 = mfma16(r, bv[g][s >> 1][(s & 1) * 2], acc0);
                    acc1 = mfma16(r, bv[g][s >> 1][(s & 1) * 2 + 1], acc1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const f32x2 pr = {acc0[v], acc1[v]};
        *(f32x2*)(red + (w * ROWS + 4 * q + v) * ATOMS + 2 * j) = pr;
    }
    __syncthreads();
    if (NW > 4 && tid >= 256) return;     // the elementwise epilogue is 256 threads wide

    f32x2 gsum = *(const f32x2*)(red + erow * ATOMS + ec);
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) {
        const f32x2 p2 = *(const f32x2*)(red + (ww * ROWS + erow) * ATOMS + ec);
        gsum[0] += p2[0];
        gsum[1] += p2[1];
    }
    f32x2 hn;
#pragma unroll
    for (int e = 0; e < 2; ++e) hn[e] = (n + e < a.N) ? activate(gsum[e] + bs[e], a.act) : 0.f;

    if (WRITE_OUT && live && rg < a.B) {
        // K.rnn masking: a masked step repeats the previous output (zeros before the first valid
        // step) and keeps the state
        float* orow = a.out + ((size_t)rg * a.T + t) * a.out_width + a.out_off;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (n + e < a.N) {
                float o = hn[e];
                if (!vld) o = (t > 0) ? orow[n + e - (ptrdiff_t)a.out_width] : 0.f;
                orow[n + e] = o;
            }
        }
    }
    if (!live) return;
    if (IS_LAST) {
        // The state buffer holds prev_output * B_U: the step uses the previous output only through
        // (prev_output * B_U) U_k (custom_layers.py:361); the output itself (above) stays unmasked.
        const float* so = a.state + (size_t)(t & 1) * sstride + hblk;
        float* sn = a.state + (size_t)((t + 1) & 1) * sstride + hblk;
        float m0 = 1.f, m1 = 1.f;
        if (rg < a.B) {
            if (n < a.N) m0 = a.drop[(size_t)rg * a.N + n];
            if (n + 1 < a.N) m1 = a.drop[(size_t)rg * a.N + n + 1];
        }
        sn[pos0] = vld ? hn[0] * m0 : so[pos0];
        sn[pos1] = vld ? hn[1] * m1 : so[pos1];
    } else {
        a.h_out[hblk + pos0] = hn[0];
        a.h_out[hblk + pos1] = hn[1];
    }
}

// state[0] = h0 tiled over the rows (custom_layers.py:336-341) or the caller's initial state
// (stateful mode, custom_layers.py:296-318), in the packed A-operand order; frame counters = 0.
__global__ void __launch_bounds__(256)
dense_init_state_kernel(const float* __restrict__ h0, const float* __restrict__ init,
                        float* __restrict__ state, int* tptr, int B, int N, int Np, int Bp,
                        const float* __restrict__ drop_in, float* __restrict__ drop) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { tptr[0] = 0; tptr[16] = 0; }
    if (i >= (size_t)Bp * Np) return;
    const int b = (int)(i / Np), n = (int)(i % Np);
    float v = 0.f;
    if (b < B && n < N) {
        // (the workspace's copy of the mask: the graphs' kernel arguments stay the same call to call)
        const float m = drop_in ? drop_in[(size_t)b * N + n] : 1.f;
        drop[(size_t)b * N + n] = m;
        v = (init ? init[(size_t)b * N + n] : h0[n]) * m;
    }
    state[((size_t)(b >> 4) * (Np / 16) + (n >> 4)) * 256 + rp_pos(b & 15, n & 15)] = v;
}

__global__ void __launch_bounds__(256)
dense_store_state_kernel(const float* __restrict__ state, float* __restrict__ out, int B, int N,
                         int Np) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * N) return;
    const int b = (int)(i / N), n = (int)(i % N);
    out[i] = state[((size_t)(b >> 4) * (Np / 16) + (n >> 4)) * 256 + rp_pos(b & 15, n & 15)];
}

// With a recurrent dropout mask the state buffer holds prev_output * B_U; what a stateful layer carries to
// its next batch is the output itself: a row's last frame repeats the output of its last valid one, a row
// with no valid frame keeps the state it entered with.  (out may be init: same element, same thread.)
__global__ void __launch_bounds__(256)
dense_store_state_unmasked_kernel(const float* __restrict__ h_out, const float* init,
                                  const float* __restrict__ h0, const unsigned char* __restrict__ valid,
                                  float* out, int T, int N, int Bp, int out_width, int out_off) {
    const int b = blockIdx.x;
    int any = 0;
    for (int t = threadIdx.x; t < T; t += 256) any |= valid[(size_t)t * Bp + b];
    any = __syncthreads_or(any);
    const float* last = h_out + ((size_t)b * T + (T - 1)) * out_width + out_off;
    for (int n = threadIdx.x; n < N; n += 256)
        out[(size_t)b * N + n] = any ? last[n] : (init ? init[(size_t)b * N + n] : h0[n]);
}

// Stacked matrix [U; S; W] of one layer -> cell_a operand packing.  Row i of the stack is
// contraction index i: [0, Np) = p (U rows), then Np rows of h (S rows) when nS, then Fp rows of x
// (W rows) when nW; padded rows / columns are zero.
__global__ void __launch_bounds__(256)
pack_dense_kernel(const float* __restrict__ U, const float* __restrict__ S,
                  const float* __restrict__ W, float* __restrict__ M, int N, int F, int Np, int Fp,
                  int L) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)L * Np) return;
    const int i = (int)(idx / Np), n = (int)(idx % Np);
    float v = 0.f;
    if (n < N) {
        if (i < Np) {
            if (i < N) v = U[(size_t)i * N + n];
        } else if (S && i < 2 * Np) {
            if (i - Np < N) v = S[(size_t)(i - Np) * N + n];
        } else if (W) {
            const int f = i - (S ? 2 * Np : Np);
            if (f < F) v = W[(size_t)f * N + n];
        }
    }
    const int c = i >> 4, fi = i & 15, n32 = n & 31;
    M[((size_t)c * (Np / 32) + (n >> 5)) * 512 + (fi >> 3) * 256 + ((fi & 3) * 16 + (n32 >> 1)) * 4 +
      ((fi >> 2) & 1) * 2 + (n32 & 1)] = v;
}

// The same stack as fp16 in the B-operand order of v_mfma_f32_16x16x32_f16 (dense_step_kernel<.., HALF>): block
// (c, ab) = rows 32c..32c+31 x atoms 32ab..32ab+31 as two 512-half halves (even atoms, odd atoms); lane
// (q, j) of a half holds rows 32c + 16b + 4s + q at e = 4b + s.  The x segment is padded to a multiple of 32
// rows (Xp); padded rows / columns are zero.
__global__ void __launch_bounds__(256)
pack_dense16_kernel(const float* __restrict__ U, const float* __restrict__ S,
                    const float* __restrict__ W, f16* __restrict__ M, int N, int F, int Np, int Xp,
                    int L) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)L * Np) return;
    const int i = (int)(idx / Np), n = (int)(idx % Np);
    float v = 0.f;
    if (n < N) {
        if (i < Np) {
            if (i < N) v = U[(size_t)i * N + n];
        } else if (S && i < 2 * Np) {
            if (i - Np < N) v = S[(size_t)(i - Np) * N + n];
        } else if (W) {
            const int f = i - (S ? 2 * Np : Np);
            if (f < F) v = W[(size_t)f * N + n];
        }
    }
    const int c = i >> 5, r = i & 31, b = r >> 4, s = (r & 15) >> 2, q = r & 3;
    const int ab = n >> 5, j = (n & 31) >> 1, hsel = n & 1;
    M[(((size_t)c * (Np / 32) + ab) * 2 + hsel) * 512 + (q * 16 + j) * 8 + 4 * b + s] = (f16)v;
}

__global__ void __launch_bounds__(256)
pack_bias_kernel(const float* __restrict__ b, float* __restrict__ out, int N, int Np, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K * Np) return;
    const int k = i / Np, n = i % Np;
    out[i] = n < N ? b[(size_t)k * N + n] : 0.f;
}

struct DenseLayout {
    int Bp, Fp, Np, numA, nX;
    int half, Xp;            // operand_f16: fp16 matrices, the x rows padded to a multiple of 32 (Xp)
    size_t L0, L1;           // stacked rows of layer 0 / of layers k > 0
    size_t off_bias, params_total;
    size_t off_xp, off_valid, off_h0, off_h1, off_state, off_t, off_drop, ws_total;
    size_t m_off(int k) const {
        return k == 0 ? 0 : (L0 + (size_t)(k - 1) * L1) * Np * (half ? 2 : 4);
    }
};

DenseLayout dense_layout(const drnmf_dense_desc_t* d) {
    DenseLayout D;
    D.Bp = pad_b(d->B);
    D.Fp = pad_f(d->F);
    D.Np = pad_n(d->N);
    D.numA = D.Np / ATOMS;
    D.half = d->operand_f16 != 0;
    D.Xp = D.half ? round_up(D.Fp, 32) : D.Fp;
    D.nX = d->connect_input ? (D.half ? D.Xp / 32 : D.Fp / 16) : 0;
    D.L0 = (size_t)D.Np + (d->connect_input ? D.Xp : 0);
    D.L1 = D.L0 + D.Np;
    D.off_bias = round_up_sz(D.m_off(d->K), 256);
    D.params_total = D.off_bias + round_up_sz((size_t)d->K * D.Np * 4, 256);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += round_up_sz(bytes, 256); return at; };
    D.off_xp = take((size_t)d->T * D.Bp * D.Fp * 4);
    D.off_valid = take((size_t)d->T * D.Bp);
    D.off_h0 = take((size_t)D.Bp * D.Np * 4);
    D.off_h1 = take((size_t)D.Bp * D.Np * 4);
    D.off_state = take((size_t)2 * D.Bp * D.Np * 4);
    D.off_t = take(256);
    D.off_drop = take((size_t)d->B * d->N * 4);
    D.ws_total = o;
    return D;
}

int validate_dense_desc(drnmf_handle_t h, const drnmf_dense_desc_t* d) {
    if (!d) DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "dense desc is NULL");
    if (d->B <= 0 || d->T <= 0 || d->F <= 0 || d->N <= 0 || d->K <= 0)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "B,T,F,N,K must be positive (got %d,%d,%d,%d,%d)",
                   d->B, d->T, d->F, d->N, d->K);
    if (d->activation < DRNMF_ACT_LINEAR || d->activation > DRNMF_ACT_HARD_SIGMOID)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "unknown activation id %d", d->activation);
    if ((int64_t)d->B * d->T * (int64_t)d->N * (d->return_all_hidden ? d->K : 1) >= (1ll << 40))
        DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED, "output tensor too large");
    return DRNMF_OK;
}

template <int G, int NW, bool HALF = false>
void* dense_func(bool last, bool write_out) {
    if (last) return (void*)&dense_step_kernel<G, NW, true, true, HALF>;
    return write_out ? (void*)&dense_step_kernel<G, NW, false, true, HALF>
                     : (void*)&dense_step_kernel<G, NW, false, false, HALF>;
}

}  // namespace

extern "C" size_t drnmf_dense_params_bytes(const drnmf_dense_desc_t* d) {
    if (!d || d->F <= 0 || d->N <= 0 || d->K <= 0) return 0;
    return dense_layout(d).params_total;
}

extern "C" size_t drnmf_dense_workspace_bytes(const drnmf_dense_desc_t* d) {
    if (!d || d->B <= 0 || d->T <= 0 || d->F <= 0 || d->N <= 0 || d->K <= 0) return 0;
    return dense_layout(d).ws_total;
}

extern "C" int32_t drnmf_dense_prepare_params(drnmf_handle_t h, const drnmf_dense_desc_t* d,
                                              const float* U, const float* S, const float* W,
                                              const float* b, void* params, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    int rc = validate_dense_desc(h, d);
    if (rc) return rc;
    if (!U || !b || !params || (d->K > 1 && !S) || (d->connect_input && !W))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "dense_prepare_params: NULL pointer argument");
    if ((uintptr_t)params & 255)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "params must be 256-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const DenseLayout D = dense_layout(d);
    char* base = (char*)params;
    const size_t NN = (size_t)d->N * d->N, FN = (size_t)d->F * d->N;
    for (int k = 0; k < d->K; ++k) {
        const int L = (int)(k == 0 ? D.L0 : D.L1);
        const size_t tot = (size_t)L * D.Np;
        if (D.half)
            hipLaunchKernelGGL(pack_dense16_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                               stream, U + (size_t)k * NN, k > 0 ? S + (size_t)(k - 1) * NN : nullptr,
                               d->connect_input ? W + (size_t)k * FN : nullptr,
                               (f16*)(base + D.m_off(k)), d->N, d->F, D.Np, D.Xp, L);
        else
            hipLaunchKernelGGL(pack_dense_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                               stream, U + (size_t)k * NN, k > 0 ? S + (size_t)(k - 1) * NN : nullptr,
                               d->connect_input ? W + (size_t)k * FN : nullptr,
                               (float*)(base + D.m_off(k)), d->N, d->F, D.Np, D.Fp, L);
    }
    hipLaunchKernelGGL(pack_bias_kernel, dim3((unsigned)((d->K * D.Np + 255) / 256)), dim3(256), 0,
                       stream, b, (float*)(base + D.off_bias), d->N, D.Np, d->K);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

static int32_t dense_forward_impl(drnmf_handle_t h, const drnmf_dense_desc_t* d, const float* x,
                                  float mask_value, const void* params, const float* h0,
                                  const float* initial_state, float* final_state, float* h_out,
                                  void* workspace, size_t workspace_bytes, void* stream_,
                                  const float* drop_u) {
    if (!h) return DRNMF_ERR_INVALID_ARG;
    int rc = validate_dense_desc(h, d);
    if (rc) return rc;
    if (!x || !params || !h_out || !workspace || (!h0 && !initial_state))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "dense_cell_forward: NULL pointer argument");
    const DenseLayout D = dense_layout(d);
    if (workspace_bytes < D.ws_total)
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "dense_cell_forward: workspace %zu < required %zu",
                   workspace_bytes, D.ws_total);
    if (((uintptr_t)workspace & 255) || ((uintptr_t)params & 255))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "workspace/params must be 256-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    const char* pb = (const char*)params;
    float* xp = (float*)(ws + D.off_xp);
    unsigned char* valid = (unsigned char*)(ws + D.off_valid);
    float* hb[2] = {(float*)(ws + D.off_h0), (float*)(ws + D.off_h1)};
    float* state = (float*)(ws + D.off_state);
    int* tA = (int*)(ws + D.off_t);
    int* tB = tA + 16;
    const int K = d->K;

    {
        const size_t rows = (size_t)d->T * D.Bp;
        hipLaunchKernelGGL(pack_input_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                           stream, x, xp, valid, mask_value, d->B, d->T, d->F, D.Bp, D.Fp);
        const size_t tot = (size_t)D.Bp * D.Np;
        hipLaunchKernelGGL(dense_init_state_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256),
                           0, stream, h0, initial_state, state, tA, d->B, d->N, D.Np, D.Bp, drop_u,
                           (float*)(ws + D.off_drop));
        DRNMF_HIP(h, hipGetLastError());
    }

    const dim3 grid(8u * (unsigned)(D.Bp / ROWS), (unsigned)(round_up(D.numA, 8) / 8));
    auto make = [&](int k) {
        DenseArgs a;
        a.M = (const float*)(pb + D.m_off(k));
        a.bias = (const float*)(pb + D.off_bias) + (size_t)k * D.Np;
        a.xp = xp;
        a.h_in = hb[(k + 1) & 1];
        a.h_out = hb[k & 1];
        a.state = state;
        a.valid = valid;
        a.out = h_out;
        // frame counters as in cell_forward.hip: nobody reads a counter in the kernel that writes it
        if (K == 1) { a.t_rd = tA; a.t_wr = nullptr; a.t_wr_add = 0; }
        else if (k == 0) { a.t_rd = tB; a.t_wr = tA; a.t_wr_add = 0; }
        else if (k == K - 1) { a.t_rd = tA; a.t_wr = tB; a.t_wr_add = 1; }
        else { a.t_rd = tA; a.t_wr = nullptr; a.t_wr_add = 0; }
        a.B = d->B; a.T = d->T; a.N = d->N; a.Bp = D.Bp; a.Fp = D.Fp; a.Np = D.Np;
        a.numA = D.numA;
        a.nP = D.half ? D.Np / 32 : D.Np / 16;
        a.nH = k > 0 ? a.nP : 0;
        a.nX = D.nX;
        a.act = d->activation;
        a.out_width = d->return_all_hidden ? d->N * K : d->N;
        a.out_off = d->return_all_hidden ? k * d->N : 0;
        a.drop = (const float*)(ws + D.off_drop);
        return a;
    };
    // waves per workgroup by contraction length (16-wide chunks).  Measured on MI355X, us per
    // layer-step with 4 / 8 / 16 waves: F=513 N=2000 B=64: 17.6 / 17.2 / 19.3; F=257 N=2000 B=32:
    // 14.9 / 13.8 / 16.8; F=257 N=200 B=32: 4.8 / 5.2 / 6.0
    int nw = (D.Np / 16) * 2 + (d->connect_input ? D.Fp / 16 : 0) >= 128 ? 8 : 4;
    if (const char* e = measure_env("DRNMF_DENSE_NW")) {   // tuning aid
        const int v = atoi(e);
        if (v == 4 || v == 8 || v == 16) nw = v;
    }
    auto func = [&](int k) {
        const bool last = k == K - 1, wo = d->return_all_hidden != 0;
        if (D.half)
            return nw == 16 ? dense_func<4, 16, true>(last, wo)
                            : (nw == 8 ? dense_func<4, 8, true>(last, wo) : dense_func<4, 4, true>(last, wo));
        return nw == 16 ? dense_func<4, 16>(last, wo)
                        : (nw == 8 ? dense_func<4, 8>(last, wo) : dense_func<4, 4>(last, wo));
    };

    int fpg = 400 / K;
    fpg = fpg < 1 ? 1 : (fpg > 64 ? 64 : fpg);
    if (fpg > d->T) fpg = d->T;
    const bool use_graph = tune_env("DRNMF_NO_GRAPH") == nullptr;
    if (!use_graph) {
        for (int t = 0; t < d->T; ++t) {
            for (int k = 0; k < K; ++k) {
                DenseArgs a = make(k);
                void* kp[1] = {&a};
                DRNMF_HIP(h, hipLaunchKernel(func(k), grid, dim3(64 * nw), kp, 0, stream));
            }
            if (K == 1) hipLaunchKernelGGL(advance_frame_kernel, dim3(1), dim3(1), 0, stream, tA);
        }
        DRNMF_HIP(h, hipGetLastError());
    } else {
        auto get_graph = [&](int frames, hipGraphExec_t* out) -> int32_t {
            std::vector<uint64_t> key = {
                0xDE05Eull, (uint64_t)d->B, (uint64_t)d->T, (uint64_t)d->F, (uint64_t)d->N,
                (uint64_t)d->K, (uint64_t)d->connect_input, (uint64_t)d->activation,
                (uint64_t)d->return_all_hidden + 2 * (uint64_t)(d->operand_f16 != 0), (uint64_t)(uintptr_t)params,
                (uint64_t)(uintptr_t)h_out, (uint64_t)(uintptr_t)workspace, (uint64_t)frames, (uint64_t)nw};
            // (as the fused path, cell_forward.hip: a hit moves to the back -- eviction is least-recently-used --
            // and is pinned by THIS call, whose sequence number dense_forward_impl advanced)
            for (size_t gi = 0; gi < h->graphs.size(); ++gi)
                if (h->graphs[gi].key == key) {
                    if (gi + 1 != h->graphs.size()) {
                        GraphEntry hit = h->graphs[gi];
                        h->graphs.erase(h->graphs.begin() + (ptrdiff_t)gi);
                        h->graphs.push_back(hit);
                    }
                    h->graphs.back().last_stream = stream;
                    h->graphs.back().pin = h->call_seq;
                    *out = h->graphs.back().exec;
                    return DRNMF_OK;
                }
            {   // bounded cache: the least recently used entry is retired without synchronising (common.h)
                const int32_t erc = graph_cache_make_room(h, stream, 24);
                if (erc) return erc;
            }
            GraphEntry ge;
            ge.key = key;
            DRNMF_HIP(h, hipGraphCreate(&ge.graph, 0));
            hipGraphNode_t last = nullptr;
            auto add = [&](void* f, dim3 g, unsigned block, void** kp) -> hipError_t {
                hipKernelNodeParams p;
                memset(&p, 0, sizeof(p));
                p.func = f;
                p.gridDim = g;
                p.blockDim = dim3(block);
                p.kernelParams = kp;
                hipGraphNode_t node;
                hipError_t e = hipGraphAddKernelNode(&node, ge.graph, last ? &last : nullptr,
                                                     last ? 1 : 0, &p);
                last = node;
                return e;
            };
            for (int rep = 0; rep < frames; ++rep) {
                for (int k = 0; k < K; ++k) {
                    DenseArgs a = make(k);
                    void* kp[1] = {&a};
                    DRNMF_HIP(h, add(func(k), grid, 64 * nw, kp));
                }
                if (K == 1) {
                    int* tp = tA;
                    void* kt[1] = {&tp};
                    DRNMF_HIP(h, add((void*)&advance_frame_kernel, dim3(1), 1, kt));
                }
            }
            DRNMF_HIP(h, hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0));
            ge.last_stream = stream;
            ge.pin = h->call_seq;
            h->graphs.push_back(ge);
            *out = ge.exec;
            return DRNMF_OK;
        };
        ++h->call_seq;       // (a top-level call: entries pinned by an EARLIER call become evictable again)
        hipGraphExec_t ex = nullptr;
        rc = get_graph(fpg, &ex);
        if (rc) return rc;
        int t = 0;
        for (; t + fpg <= d->T; t += fpg) DRNMF_HIP(h, hipGraphLaunch(ex, stream));
        if (t < d->T) {
            rc = get_graph(1, &ex);
            if (rc) return rc;
            for (; t < d->T; ++t) DRNMF_HIP(h, hipGraphLaunch(ex, stream));
        }
    }
    if (final_state && drop_u) {
        hipLaunchKernelGGL(dense_store_state_unmasked_kernel, dim3(d->B), dim3(256), 0, stream, h_out,
                           initial_state, h0, valid, final_state, d->T, d->N, D.Bp,
                           d->return_all_hidden ? d->N * K : d->N,
                           d->return_all_hidden ? (K - 1) * d->N : 0);
        DRNMF_HIP(h, hipGetLastError());
    } else if (final_state) {
        const size_t tot = (size_t)d->B * d->N;
        hipLaunchKernelGGL(dense_store_state_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256),
                           0, stream, state + (size_t)(d->T & 1) * D.Bp * D.Np, final_state, d->B,
                           d->N, D.Np);
        DRNMF_HIP(h, hipGetLastError());
    }
    return DRNMF_OK;
}

extern "C" int32_t drnmf_dense_cell_forward(drnmf_handle_t h, const drnmf_dense_desc_t* d,
                                            const float* x, float mask_value, const void* params,
                                            const float* h0, const float* initial_state,
                                            float* final_state, float* h_out, void* workspace,
                                            size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    return dense_forward_impl(h, d, x, mask_value, params, h0, initial_state, final_state, h_out,
                              workspace, workspace_bytes, stream_, nullptr);
}

// Training phase with recurrent dropout (dropout_U, custom_layers.py:361, 377-384): drop_u [B][N] = the
// mask B_U the caller drew (0 or 1/(1-p)); prev_output * B_U enters every U_k product.
extern "C" int32_t drnmf_dense_cell_forward_dropout(drnmf_handle_t h, const drnmf_dense_desc_t* d,
                                                    const float* x, float mask_value,
                                                    const void* params, const float* h0,
                                                    const float* drop_u, float* h_out,
                                                    void* workspace, size_t workspace_bytes,
                                                    void* stream_) {
    DRNMF_LOCK(h);
    if (h && !drop_u) DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "dense_cell_forward_dropout: NULL mask");
    return dense_forward_impl(h, d, x, mask_value, params, h0, nullptr, nullptr, h_out, workspace,
                              workspace_bytes, stream_, drop_u);
}

// The same for a STATEFUL layer in its training phase (Keras stateful=True under fit, custom_layers.py:
// 296-318 with 377-384): the carried state enters as in drnmf_dense_cell_forward (B_U multiplies it where the
// U_k products read it; what final_state receives is the unmasked output of each row's last valid frame).
extern "C" int32_t drnmf_dense_cell_forward_dropout_stateful(
    drnmf_handle_t h, const drnmf_dense_desc_t* d, const float* x, float mask_value, const void* params,
    const float* h0, const float* initial_state, float* final_state, const float* drop_u, float* h_out,
    void* workspace, size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    if (h && !drop_u)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "dense_cell_forward_dropout_stateful: NULL mask");
    return dense_forward_impl(h, d, x, mask_value, params, h0, initial_state, final_state, h_out,
                              workspace, workspace_bytes, stream_, drop_u);
}
