// Recurrent DR-NMF cell forward on gfx950.
//
// Reference semantics: Masking (enhance.py:253) -> Recurrent.call / K.rnn ->
// SimpleDeepRNN.get_initial_state + step (custom_layers.py:336-375) with the build_alt maps
// (enhance.py:161-204), in the factored ISTA form
//     layer 0 : h  = relu(u0d*p + u0o*(sum(p)-p) + (x Dn_0)*ia_0 + b_0)
//     layer k : h' = relu(h + ((x - h Dn_k^T) Dn_k)*ia_k + b_k + uko*sum(p))
//
// The chain over (t, k) is strictly sequential; only the batch rows and the inside of each
// contraction are parallel.  One layer-step is two dependent skinny GEMMs (B x N x F each) that
// need the whole dictionary, so the dictionary is distributed over the chip by ATOM BLOCKS and
// the only cross-workgroup exchange per layer-step is the reduction of the partial
// reconstructions x^ = sum_blocks h[:, block] Dn[:, block]^T.  On gfx950 a kernel boundary
// (~1.5 us) is cheaper than any in-kernel all-to-all, so the seam is cut THERE:
//
//   cell_layer_kernel (one workgroup = 16 batch rows x 32 atoms, all F):
//       g  = r[16 x F] . Dn_k[F x 32]            MFMA, F split over the 4 waves, LDS reduce
//       h' = relu(...)                           fused epilogue (+ mask select, row sums, output)
//       x^_part = h'[16 x 32] . Dn_{k+1}[F x 32]^T  MFMA, F tiles split over the 4 waves
//   reduce_residual_kernel:  r = x_t - sum_blocks x^_part
//
// so a frame is 2K-1 launches (+1 frame-counter bump), captured once as a hipGraph and replayed
// T times; kernels read the frame index from device memory.  The dictionary slice of a
// workgroup (F x 32 floats = 128-byte rows) is read straight into MFMA operand registers: it is
// not shared between the waves of a workgroup, so an LDS round trip would only add latency.
#include "common.h"

namespace {

constexpr int ROWS = 16;    // batch rows per workgroup (one MFMA M tile)
constexpr int ATOMS = 32;   // atoms per workgroup
constexpr int HT_LD = 36;   // LDS row stride of the h tile (floats): 16-byte pad vs bank conflicts

struct CellArgs {
    const float* Dn;         // [Fp][Np]  this layer's unit-norm dictionary
    const float* Dn_next;    // [Fp][Np]  next layer's (unused when last)
    const float* inv_alpha;  // [Np]
    const float* bias;       // [Np]
    const float* rsrc;       // first layer: xp [T][Bp][Fp]; else r [Bp][Fp]
    const float* h_in;       // [Bp][Np]  previous layer's h (first layer: the state p)
    float* h_out;            // [Bp][Np]  this layer's h (last layer: the state)
    float* state;            // [Bp][Np]
    float* partial;          // [numA][Bp][Fp]
    float* rs_part;          // [2][numA][Bp] row sums of the state per atom block, by frame parity
    float* psum;             // [Bp]  sum(p) of the current frame
    const unsigned char* valid;  // [T][Bp]
    float* out;              // [B][T][out_width]
    const int* tptr;         // device frame counter
    float u0d, u0o, uko;
    int B, T, N, Bp, Fp, Np, numA, nchunks;
    int out_width, out_off, write_out;
};

struct ReduceArgs {
    const float* xp;       // [T][Bp][Fp]
    const float* partial;  // [numA][Bp][Fp]
    float* r;              // [Bp][Fp]
    const int* tptr;
    int n4;                // Bp*Fp/4
    int numA;
};

// G = chunks of 16 bins handled per wave per group (all of a group's operand loads are issued
// before its first MFMA so that one memory round trip covers the group).
template <int G, bool IS_FIRST, bool IS_LAST>
__global__ void __launch_bounds__(256) cell_layer_kernel(const CellArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[4 * ROWS * ATOMS + ROWS * HT_LD];
    float* red = lds;                         // [4][16][32]
    float* htile = lds + 4 * ROWS * ATOMS;    // [16][HT_LD]

    // XCD-aware block -> (atom block, row tile): blocks are dealt round-robin to the 8 XCDs, so
    // the row tiles that share one dictionary slice are given the same blockIdx % 8.
    const int numM = a.Bp / ROWS;
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int m = qb % numM;
    const int ab = (qb / numM) * 8 + xcd;
    if (ab >= a.numA) return;

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63, j = l & 15, q = l >> 4;
    const int t = *a.tptr;
    const int Fp = a.Fp, Np = a.Np;
    const int row0 = m * ROWS, n0 = ab * ATOMS;

    const float* rsrc = IS_FIRST ? a.rsrc + (size_t)t * a.Bp * Fp : a.rsrc;
    const float* arow = rsrc + (size_t)(row0 + j) * Fp + 4 * q;         // + 16*c
    const float* brow = a.Dn + (size_t)(4 * q) * Np + n0 + 2 * j;       // + (16*c + s)*Np
    const float* nrow = a.Dn_next + (size_t)j * Np + n0 + 4 * q;        // + 16*ft*Np + 16*u

    // ---- epilogue operands (issued first: tiny, needed last) -------------------------------
    const int erow = tid >> 4, ec = (tid & 15) * 2;
    const int rg = row0 + erow, n = n0 + ec;
    const f32x2 hp = *(const f32x2*)(a.h_in + (size_t)rg * Np + n);
    const f32x2 ia = *(const f32x2*)(a.inv_alpha + n);
    const f32x2 bs = *(const f32x2*)(a.bias + n);
    float ps;
    if (IS_FIRST) {
        // sum(p) = sum over atom blocks of the row sums left by the previous frame's last layer,
        // added in block order by 16 lanes + a fixed shuffle tree (deterministic)
        const float* rp = a.rs_part + (size_t)(t & 1) * a.numA * a.Bp + rg;
        float s = 0.f;
        for (int b2 = (tid & 15); b2 < a.numA; b2 += 16) s += rp[(size_t)b2 * a.Bp];
        s += __shfl_xor(s, 8, 16);
        s += __shfl_xor(s, 4, 16);
        s += __shfl_xor(s, 2, 16);
        s += __shfl_xor(s, 1, 16);
        ps = s;
        if (ab == 0 && (tid & 15) == 0) a.psum[rg] = ps;
    } else {
        ps = a.psum[rg];
    }

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 nb[G][2];

    // ---- GEMM a:  g[16 x 32] = r[16 x F] . Dn[F x 32], wave w takes chunks c = w (mod 4) ------
    const int per_wave = (a.nchunks - w + 3) >> 2;   // chunks owned by this wave
    for (int base = 0; base < per_wave; base += G) {
        f32x4 av[G];
        f32x2 bv[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (base + g < per_wave) {
                const int c = w + 4 * (base + g);
                av[g] = *(const f32x4*)(arow + 16 * c);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    bv[g][s] = *(const f32x2*)(brow + (size_t)(16 * c + s) * Np);
            }
        }
        if (!IS_LAST && base == 0) {
            // prefetch GEMM b's dictionary operands behind GEMM a's: in flight during GEMM a
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g < per_wave) {
                    const int ft = w + 4 * g;
                    nb[g][0] = *(const f32x4*)(nrow + (size_t)(16 * ft) * Np);
                    nb[g][1] = *(const f32x4*)(nrow + (size_t)(16 * ft) * Np + 16);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (base + g < per_wave) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc0 = mfma16(av[g][s], bv[g][s][0], acc0);
                    acc1 = mfma16(av[g][s], bv[g][s][1], acc1);
                }
            }
        }
    }

    // ---- cross-wave reduction of the 4 F-splits through LDS --------------------------------
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        f32x2 pr = {acc0[v], acc1[v]};
        *(f32x2*)(red + (w * ROWS + 4 * q + v) * ATOMS + 2 * j) = pr;
    }
    __syncthreads();
    f32x2 gsum = *(const f32x2*)(red + (0 * ROWS + erow) * ATOMS + ec);
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) {
        const f32x2 p2 = *(const f32x2*)(red + (ww * ROWS + erow) * ATOMS + ec);
        gsum[0] += p2[0];
        gsum[1] += p2[1];
    }

    // ---- fused update: soft-threshold / non-negativity projection --------------------------
    f32x2 hn;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        float pre;
        if (IS_FIRST) pre = a.u0d * hp[e] + a.u0o * (ps - hp[e]);
        else pre = hp[e] + a.uko * ps;
        pre += gsum[e] * ia[e] + bs[e];
        hn[e] = fmaxf(pre, 0.f);
    }

    const bool row_live = rg < a.B;
    bool vld = true;
    if (IS_LAST || a.write_out) vld = a.valid[(size_t)t * a.Bp + rg] != 0;
    if (a.write_out && row_live) {
        // K.rnn masking: a masked step repeats the previous output (zeros before the first
        // valid step)
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
    if (IS_LAST) {
        // ... and keeps the previous state
        f32x2 st = hn;
        if (!vld) st = IS_FIRST ? hp : *(const f32x2*)(a.state + (size_t)rg * Np + n);
        *(f32x2*)(a.state + (size_t)rg * Np + n) = st;
        float s = st[0] + st[1];
        s += __shfl_xor(s, 8, 16);
        s += __shfl_xor(s, 4, 16);
        s += __shfl_xor(s, 2, 16);
        s += __shfl_xor(s, 1, 16);
        if ((tid & 15) == 0)
            a.rs_part[((size_t)((t + 1) & 1) * a.numA + ab) * a.Bp + rg] = s;
    } else {
        *(f32x2*)(a.h_out + (size_t)rg * Np + n) = hn;
        *(f32x2*)(htile + erow * HT_LD + ec) = hn;
        __syncthreads();

        // ---- GEMM b: x^_part[16 x F] = h'[16 x 32] . Dn_next[F x 32]^T ---------------------
        // contraction slot (u, q, c) <-> local atom 16u + 4q + c on both operands
        const f32x4 ha0 = *(const f32x4*)(htile + j * HT_LD + 4 * q);
        const f32x4 ha1 = *(const f32x4*)(htile + j * HT_LD + 16 + 4 * q);
        float* prow = a.partial + ((size_t)ab * a.Bp + row0 + 4 * q) * Fp + j;
        for (int base = 0; base < per_wave; base += G) {
            if (base > 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (base + g < per_wave) {
                        const int ft = w + 4 * (base + g);
                        nb[g][0] = *(const f32x4*)(nrow + (size_t)(16 * ft) * Np);
                        nb[g][1] = *(const f32x4*)(nrow + (size_t)(16 * ft) * Np + 16);
                    }
                }
            }
            f32x4 xa[G];
#pragma unroll
            for (int g = 0; g < G; ++g) xa[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (base + g < per_wave) xa[g] = mfma16(ha0[c], nb[g][0][c], xa[g]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int g = 0; g < G; ++g)
                    if (base + g < per_wave) xa[g] = mfma16(ha1[c], nb[g][1][c], xa[g]);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (base + g < per_wave) {
                    const int ft = w + 4 * (base + g);
#pragma unroll
                    for (int v = 0; v < 4; ++v) prow[(size_t)v * Fp + 16 * ft] = xa[g][v];
                }
            }
        }
    }
}

// r = x_t - sum_blocks x^_part.  256 threads = 32 float4 outputs x 8 block subsets; the subsets
// are combined through LDS in a fixed order.
__global__ void __launch_bounds__(256) reduce_residual_kernel(const ReduceArgs a) {
    __shared__ f32x4 red[8][32];
    const int el = threadIdx.x & 31, s = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    const int t = *a.tptr;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (e < a.n4) {
        const f32x4* p = (const f32x4*)a.partial + e;
        for (int b = s; b < a.numA; b += 8) acc += p[(size_t)b * a.n4];
    }
    red[s][el] = acc;
    __syncthreads();
    if (s == 0 && e < a.n4) {
        f32x4 tot = red[0][el];
#pragma unroll
        for (int k = 1; k < 8; ++k) tot += red[k][el];
        const f32x4 xv = ((const f32x4*)a.xp)[(size_t)t * a.n4 + e];
        ((f32x4*)a.r)[e] = xv - tot;
    }
}

__global__ void advance_frame_kernel(int* tptr) { *tptr += 1; }

// Masking + relayout: x [B][T][F] -> xp [T][Bp][Fp] (masked frames and all padding zero) and
// valid [T][Bp].  One wave per (t, row).  [K2.0.4-memory: keras.layers.Masking]
__global__ void __launch_bounds__(256)
pack_input_kernel(const float* __restrict__ x, float* __restrict__ xp,
                  unsigned char* __restrict__ valid, float mask_value, int B, int T, int F, int Bp,
                  int Fp) {
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const size_t rowid = (size_t)blockIdx.x * 4 + wv;   // = t*Bp + b
    if (rowid >= (size_t)T * Bp) return;
    const int t = (int)(rowid / Bp), b = (int)(rowid % Bp);
    float* dst = xp + rowid * Fp;
    bool any = false;
    if (b < B) {
        const float* src = x + ((size_t)b * T + t) * F;
        for (int f = l; f < F; f += 64) any |= (src[f] != mask_value);
        any = __any(any);
        for (int f = l; f < Fp; f += 64) dst[f] = (any && f < F) ? src[f] : 0.f;
    } else {
        for (int f = l; f < Fp; f += 64) dst[f] = 0.f;
    }
    if (l == 0) valid[rowid] = any ? 1 : 0;
}

// state = softplus(log_h0) for every row (custom_layers.py:203-206, 336-341); row sums of the
// initial state go to atom block 0 of parity 0; frame counter = 0.
__global__ void __launch_bounds__(256)
init_state_kernel(const float* __restrict__ log_h0, float* __restrict__ state,
                  float* __restrict__ rs_part, int* tptr, int N, int Np, int Bp, int numA) {
    __shared__ float wsum[4];
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int n = tid; n < Np; n += 256) {
        float v = 0.f;
        if (n < N) {
            const float z = log_h0[n];
            v = (z > 20.f) ? z : log1pf(expf(z));
            s += v;
        }
        for (int b = 0; b < Bp; ++b) state[(size_t)b * Np + n] = v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = s;
    __syncthreads();
    const float tot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    for (int i = tid; i < 2 * numA * Bp; i += 256) rs_part[i] = (i < Bp) ? tot : 0.f;
    if (tid == 0) *tptr = 0;
}

struct Workspace {
    size_t off_xp, off_valid, off_r, off_h0, off_h1, off_state, off_partial, off_rs, off_psum,
        off_t, total;
    int Bp, Fp, Np, numA;
};

Workspace workspace_layout(const drnmf_cell_desc_t* d) {
    Workspace W;
    W.Bp = pad_b(d->B);
    W.Fp = pad_f(d->F);
    W.Np = pad_n(d->N);
    W.numA = W.Np / ATOMS;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += round_up_sz(bytes, 256); return at; };
    W.off_xp = take((size_t)d->T * W.Bp * W.Fp * 4);
    W.off_valid = take((size_t)d->T * W.Bp);
    W.off_r = take((size_t)W.Bp * W.Fp * 4);
    W.off_h0 = take((size_t)W.Bp * W.Np * 4);
    W.off_h1 = take((size_t)W.Bp * W.Np * 4);
    W.off_state = take((size_t)W.Bp * W.Np * 4);
    W.off_partial = take((size_t)W.numA * W.Bp * W.Fp * 4);
    W.off_rs = take((size_t)2 * W.numA * W.Bp * 4);
    W.off_psum = take((size_t)W.Bp * 4);
    W.off_t = take(256);
    W.total = o;
    return W;
}

template <int G>
void* layer_func(bool first, bool last) {
    if (first && last) return (void*)&cell_layer_kernel<G, true, true>;
    if (first) return (void*)&cell_layer_kernel<G, true, false>;
    if (last) return (void*)&cell_layer_kernel<G, false, true>;
    return (void*)&cell_layer_kernel<G, false, false>;
}

void* pick_layer_func(int nchunks, bool first, bool last) {
    const int per_wave = (nchunks + 3) / 4;
    if (per_wave <= 3) return layer_func<3>(first, last);
    if (per_wave <= 5) return layer_func<5>(first, last);
    return layer_func<9>(first, last);
}

}  // namespace

extern "C" size_t drnmf_cell_workspace_bytes(const drnmf_cell_desc_t* d) {
    if (!d || d->B <= 0 || d->T <= 0 || d->F <= 0 || d->N <= 0) return 0;
    return workspace_layout(d).total;
}

extern "C" int32_t drnmf_cell_forward(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                      float mask_value, const void* params, const float* log_h0,
                                      float u0_diag, float u0_off, float uk_off, float* h_out,
                                      void* workspace, size_t workspace_bytes, void* stream_) {
    if (!h) return DRNMF_ERR_INVALID_ARG;
    int rc = validate_cell_desc(h, d);
    if (rc) return rc;
    if (!x || !params || !log_h0 || !h_out || !workspace)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "cell_forward: NULL pointer argument");
    const Workspace W = workspace_layout(d);
    if (workspace_bytes < W.total)
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "cell_forward: workspace %zu < required %zu",
                   workspace_bytes, W.total);
    if (((uintptr_t)workspace & 255) || ((uintptr_t)params & 255))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "workspace/params must be 256-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const ParamsLayout L = params_layout(d);
    char* ws = (char*)workspace;
    const char* pb = (const char*)params;
    float* xp = (float*)(ws + W.off_xp);
    unsigned char* valid = (unsigned char*)(ws + W.off_valid);
    float* rbuf = (float*)(ws + W.off_r);
    float* hb[2] = {(float*)(ws + W.off_h0), (float*)(ws + W.off_h1)};
    float* state = (float*)(ws + W.off_state);
    float* partial = (float*)(ws + W.off_partial);
    float* rs_part = (float*)(ws + W.off_rs);
    float* psum = (float*)(ws + W.off_psum);
    int* tptr = (int*)(ws + W.off_t);
    const int K = d->K;

    // ---- per-call prologue ------------------------------------------------------------------
    {
        const size_t rows = (size_t)d->T * W.Bp;
        hipLaunchKernelGGL(pack_input_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                           stream, x, xp, valid, mask_value, d->B, d->T, d->F, W.Bp, W.Fp);
        hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(256), 0, stream, log_h0, state,
                           rs_part, tptr, d->N, W.Np, W.Bp, W.numA);
        DRNMF_HIP(h, hipGetLastError());
    }

    // ---- one frame = 2K-1 launches + counter bump, as a cached hipGraph ----------------------
    std::vector<uint64_t> key = {
        (uint64_t)d->B, (uint64_t)d->T, (uint64_t)d->F, (uint64_t)d->N, (uint64_t)d->K,
        (uint64_t)d->n_D, (uint64_t)d->return_all_hidden, (uint64_t)(uintptr_t)params,
        (uint64_t)(uintptr_t)h_out, (uint64_t)(uintptr_t)workspace};
    {
        uint32_t b0, b1, b2;
        memcpy(&b0, &u0_diag, 4); memcpy(&b1, &u0_off, 4); memcpy(&b2, &uk_off, 4);
        key.push_back(b0); key.push_back(b1); key.push_back(b2);
    }
    const bool use_graph = getenv("DRNMF_NO_GRAPH") == nullptr;

    const unsigned grid_layer = (unsigned)(round_up(W.numA, 8) * (W.Bp / ROWS));
    const int n4 = W.Bp * W.Fp / 4;
    const unsigned grid_reduce = (unsigned)((n4 + 31) / 32);

    auto make_args = [&](int k) {
        CellArgs a;
        const float* Dn_base = (const float*)(pb + L.off_dn);
        const size_t dstride = (size_t)L.Fp * L.Np;
        a.Dn = Dn_base + (d->n_D == 1 ? 0 : (size_t)k * dstride);
        a.Dn_next = Dn_base + (d->n_D == 1 || k + 1 >= K ? 0 : (size_t)(k + 1) * dstride);
        a.inv_alpha = (const float*)(pb + L.off_inv_alpha) + (size_t)k * L.Np;
        a.bias = (const float*)(pb + L.off_bias) + (size_t)k * L.Np;
        a.rsrc = (k == 0) ? xp : rbuf;
        a.h_in = (k == 0) ? state : hb[(k - 1) & 1];
        a.h_out = (k == K - 1) ? state : hb[k & 1];
        a.state = state;
        a.partial = partial;
        a.rs_part = rs_part;
        a.psum = psum;
        a.valid = valid;
        a.out = h_out;
        a.tptr = tptr;
        a.u0d = u0_diag; a.u0o = u0_off; a.uko = uk_off;
        a.B = d->B; a.T = d->T; a.N = d->N; a.Bp = W.Bp; a.Fp = W.Fp; a.Np = W.Np;
        a.numA = W.numA; a.nchunks = W.Fp / 16;
        a.out_width = d->return_all_hidden ? d->N * K : d->N;
        a.out_off = d->return_all_hidden ? k * d->N : 0;
        a.write_out = (d->return_all_hidden || k == K - 1) ? 1 : 0;
        return a;
    };
    ReduceArgs ra;
    ra.xp = xp; ra.partial = partial; ra.r = rbuf; ra.tptr = tptr; ra.n4 = n4; ra.numA = W.numA;

    if (!use_graph) {
        for (int t = 0; t < d->T; ++t) {
            for (int k = 0; k < K; ++k) {
                CellArgs a = make_args(k);
                void* kp[1] = {&a};
                DRNMF_HIP(h, hipLaunchKernel(pick_layer_func(a.nchunks, k == 0, k == K - 1),
                                             dim3(grid_layer), dim3(256), kp, 0, stream));
                if (k < K - 1)
                    hipLaunchKernelGGL(reduce_residual_kernel, dim3(grid_reduce), dim3(256), 0,
                                       stream, ra);
            }
            hipLaunchKernelGGL(advance_frame_kernel, dim3(1), dim3(1), 0, stream, tptr);
        }
        DRNMF_HIP(h, hipGetLastError());
        return DRNMF_OK;
    }

    GraphEntry* entry = nullptr;
    for (auto& g : h->graphs)
        if (g.key == key) { entry = &g; break; }
    if (!entry) {
        if (h->graphs.size() >= 8) {   // bounded cache: drop the oldest
            (void)hipGraphExecDestroy(h->graphs.front().exec);
            (void)hipGraphDestroy(h->graphs.front().graph);
            h->graphs.erase(h->graphs.begin());
        }
        GraphEntry ge;
        ge.key = key;
        DRNMF_HIP(h, hipGraphCreate(&ge.graph, 0));
        hipGraphNode_t last = nullptr;
        auto add = [&](void* func, unsigned grid, unsigned block, void* argp) -> hipError_t {
            hipKernelNodeParams p;
            memset(&p, 0, sizeof(p));
            void* kp[1] = {argp};
            p.func = func;
            p.gridDim = dim3(grid);
            p.blockDim = dim3(block);
            p.sharedMemBytes = 0;
            p.kernelParams = kp;
            p.extra = nullptr;
            hipGraphNode_t node;
            hipError_t e = hipGraphAddKernelNode(&node, ge.graph, last ? &last : nullptr,
                                                 last ? 1 : 0, &p);
            last = node;
            return e;
        };
        for (int k = 0; k < K; ++k) {
            CellArgs a = make_args(k);
            DRNMF_HIP(h, add(pick_layer_func(a.nchunks, k == 0, k == K - 1), grid_layer, 256, &a));
            if (k < K - 1) DRNMF_HIP(h, add((void*)&reduce_residual_kernel, grid_reduce, 256, &ra));
        }
        int* tp = tptr;
        DRNMF_HIP(h, add((void*)&advance_frame_kernel, 1, 1, &tp));
        DRNMF_HIP(h, hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0));
        h->graphs.push_back(ge);
        entry = &h->graphs.back();
    }
    for (int t = 0; t < d->T; ++t) DRNMF_HIP(h, hipGraphLaunch(entry->exec, stream));
    return DRNMF_OK;
}
