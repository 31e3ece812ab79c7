// Frame-parallel sparse-NMF inference on gfx950: K-iteration ISTA (enhance.py:402-456) and
// multiplicative updates with a fixed dictionary (sparseNMF/sparse_nmf_gpu.m:163-173, 210-229;
// enhance.py:838-852).  Frames are independent here, so every iteration is a pair of large-M
// fp32-MFMA GEMMs (gemm_nt.h) whose epilogues carry the whole update:
//     X^ = H W^T      -> residual / ratio written directly        (ISTA: g(X, X^);  MU: max(.,flr))
//     G  = R W        -> H <- max(0, H + (G - lam1)/alph)   or    H <- H * DMH / max(G + sp, flr)
// Row layout: frames are rows (X, V: [n][F]; H: [n][N]); W is [F][N] as in the reference.
#include "gemm_nt.h"
#include "gemm_tn.h"

namespace {

inline int pad4(int v) { return (v + 3) / 4 * 4; }

// Wt[n][f] = W[f][n] (zero-padded to ld = Fp4): the second GEMM contracts over bins
__global__ void __launch_bounds__(256)
transpose_pad_kernel(const float* __restrict__ W, float* __restrict__ Wt, int F, int N, int Fp4) {
    __shared__ float tile[32][33];
    const int f0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int f = f0 + r, n = n0 + tx;
        tile[r][tx] = (f < F && n < N) ? W[(size_t)f * N + n] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, f = f0 + tx;
        if (n < N && f < Fp4) Wt[(size_t)n * Fp4 + f] = tile[tx][r];
    }
}

__global__ void __launch_bounds__(256) fill_kernel(float* p, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------- ISTA epilogues ----------------------------------------------------------------
template <int DIV>   // compile-time divergence: the ED epilogue must not carry the powf paths
struct EpiResidual {   // R = g(X, X^): enhance.py:412 / 431 / 450
    const float* X;
    float* R;
    int F, ldr;
    float beta;
    __device__ f32x2 pre(int64_t row, int col) const { return f32x2{X[row * F + col], 0.f}; }
    __device__ void operator()(int64_t row, int col, float xh, f32x2 pv) const {
        const float x = pv[0];
        float r;
        if (DIV == DRNMF_DIV_ED) r = x - xh;
        else if (DIV == DRNMF_DIV_KL) r = x / xh - 1.f;
        else r = x * powf(xh, beta - 2.f) - powf(xh, beta - 1.f);
        R[row * ldr + col] = r;
    }
};

// Odd bins kept out of the k-tiles (0..2, see drnmf_ista_forward) enter G = R W through
// gemm::Operands::ktail.
struct EpiIstaUpdate {   // H <- max(0, -lam1/alph + H + (1/alph) G): enhance.py:412
    float* H;
    int N;
    float neg_lam_over_alph, inv_alph;
    __device__ f32x2 pre(int64_t row, int col) const { return f32x2{H[row * N + col], 0.f}; }
    __device__ void operator()(int64_t row, int col, float gacc, f32x2 pv) const {
        H[row * N + col] = fmaxf(0.f, neg_lam_over_alph + pv[0] + inv_alph * gacc);
    }
};

// Odd bins of a 2^k+1 STFT (F = 16 j + 1 or 2): X^[row][Fm+i] = H[row,:] . W[Fm+i,:] and its
// residual, one wave per row -- instead of a fifth, almost empty 128-column tile in the X^ GEMM
// (F = 513: 1280 -> 1024 workgroups, 2.5 -> 2 rounds of the chip) and a 17th k-tile in the G GEMM.
template <int DIV>
__global__ void __launch_bounds__(256)
ista_tail_kernel(const float* __restrict__ X, const float* __restrict__ W,
                 const float* __restrict__ H, float* __restrict__ R, int64_t n, int F, int N,
                 int Fm, int nt, int ldr, float beta) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (row >= n) return;
    float s[2] = {0.f, 0.f};
    const float* hrow = H + row * N;
    if ((N & 3) == 0 && (((uintptr_t)H | (uintptr_t)W) & 15) == 0) {
        // 16-byte loads: this pass over H is pure bandwidth (one more read of the activations per
        // iteration), 4-byte loads left it at 3.2 TB/s
        const f32x4* h4 = (const f32x4*)hrow;
        const f32x4* w0 = (const f32x4*)(W + (size_t)Fm * N);
        const f32x4* w1 = (const f32x4*)(W + (size_t)(Fm + (nt > 1 ? 1 : 0)) * N);
        for (int c = l; c < N / 4; c += 64) {
            const f32x4 hv = h4[c], a0 = w0[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[0] = fmaf(hv[e], a0[e], s[0]);
            if (nt > 1) {
                const f32x4 a1 = w1[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) s[1] = fmaf(hv[e], a1[e], s[1]);
            }
        }
    } else {
        for (int c = l; c < N; c += 64) {
            const float hv = hrow[c];
            s[0] = fmaf(hv, W[(size_t)Fm * N + c], s[0]);
            if (nt > 1) s[1] = fmaf(hv, W[(size_t)(Fm + 1) * N + c], s[1]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s[0] += __shfl_xor(s[0], o, 64);
        s[1] += __shfl_xor(s[1], o, 64);
    }
    if (l < nt) {
        const float xh = s[l], x = X[row * F + Fm + l];
        float r;
        if (DIV == DRNMF_DIV_ED) r = x - xh;
        else if (DIV == DRNMF_DIV_KL) r = x / xh - 1.f;
        else r = x * powf(xh, beta - 2.f) - powf(xh, beta - 1.f);
        R[row * ldr + Fm + l] = r;
    }
}

// ---------------- MU epilogues ------------------------------------------------------------------
struct EpiStore {
    float* C;
    int ldc;
    __device__ f32x2 pre(int64_t, int) const { return f32x2{0.f, 0.f}; }
    __device__ void operator()(int64_t row, int col, float v, f32x2) const { C[row * ldc + col] = v; }
};

template <int BC>   // beta class: 2 -> beta == 2, 1 -> beta == 1, 0 -> general
struct EpiLambda {   // lambda = max(W H, flr) and its beta-dependent derivatives
    const float* V;
    float* P1;   // beta==2: lambda;  beta==1: V/lambda;  else: lambda^(beta-1)
    float* P2;   // general beta only: V * lambda^(beta-2)
    float* LAM;  // optional: lambda itself (dictionary training: objective)
    int F, ld;
    float beta, flr;
    __device__ f32x2 pre(int64_t row, int col) const {
        return f32x2{BC == 2 ? 0.f : V[row * F + col], 0.f};
    }
    __device__ void operator()(int64_t row, int col, float acc, f32x2 pv) const {
        const float lam = fmaxf(acc, flr);
        const size_t o = row * ld + col;
        if (LAM) LAM[o] = lam;
        if (BC == 2) {
            P1[o] = lam;
        } else if (BC == 1) {
            P1[o] = pv[0] / lam;
        } else {
            P1[o] = powf(lam, beta - 1.f);
            P2[o] = pv[0] * powf(lam, beta - 2.f);
        }
    }
};

// The same with the objective's divergence term D(V | lambda) summed on the way (gemm_nt.h REDUCE): the
// lambda of sparse_nmf_gpu.m:263 is the one its objective (:267-281) is evaluated on, so the training
// iteration needs no separate pass over V and lambda.  beta == 2: lambda is written once (P1).
template <int BC>
struct EpiLambdaObj {
    const float* V;
    float* P1;
    float* P2;
    float* red_out;      // one partial per workgroup
    int F, ld;
    float beta, flr;
    static constexpr bool REDUCE = true;
    __device__ f32x2 pre(int64_t row, int col) const { return f32x2{V[row * F + col], 0.f}; }
    __device__ float operator()(int64_t row, int col, float acc, f32x2 pv) const {
        const float lam = fmaxf(acc, flr), v = pv[0];
        const size_t o = row * ld + col;
        if (BC == 2) {
            P1[o] = lam;
            return (v - lam) * (v - lam);
        } else if (BC == 1) {
            P1[o] = v / lam;
            return v * logf(v / lam) - v + lam;
        }
        // (two powf per element instead of five: lam^(beta-2) and lam^beta from lam^(beta-1) by one division /
        // product -- with five, the 64 unrolled outputs of a thread spilled 1 KB per lane)
        const float p1 = powf(lam, beta - 1.f);
        P1[o] = p1;
        P2[o] = v * (p1 / lam);
        if (beta == 0.f) return v / lam - logf(v / lam) - 1.f;
        return (powf(v, beta) + (beta - 1.f) * (p1 * lam) - beta * v * p1) / (beta * (beta - 1.f));
    }
};

template <int BC>
struct EpiMuUpdate {   // H <- H * dmh / max(dph + sparsity, flr): sparse_nmf_gpu.m:217-227
    float* H;
    const float* DMH;      // [n][N] (beta != 1)
    const float* colsum;   // [N]    (beta == 1: dph = sum_f w + sparsity, dmh = acc)
    int N;
    float sparsity, flr;
    static constexpr bool EARLY = BC == 1;   // two loaded values per output do not fit early (gemm_nt.h)
    __device__ f32x2 pre(int64_t row, int col) const {
        return f32x2{H[row * N + col], BC == 1 ? colsum[col] : DMH[row * N + col]};
    }
    __device__ void operator()(int64_t row, int col, float acc, f32x2 pv) const {
        if (BC == 1) H[row * N + col] = pv[0] * acc / fmaxf(pv[1] + sparsity, flr);
        else H[row * N + col] = pv[0] * pv[1] / fmaxf(acc + sparsity, flr);
    }
};

// H update with sum(H) -- the sparsity term of the objective -- summed on the way (gemm_nt.h REDUCE)
template <int BC>
struct EpiMuUpdateSum {
    float* H;
    const float* DMH;
    const float* colsum;
    float* red_out;
    int N;
    float sparsity, flr;
    static constexpr bool EARLY = BC == 1;
    static constexpr bool REDUCE = true;
    __device__ f32x2 pre(int64_t row, int col) const {
        return f32x2{H[row * N + col], BC == 1 ? colsum[col] : DMH[row * N + col]};
    }
    __device__ float operator()(int64_t row, int col, float acc, f32x2 pv) const {
        const float hn = BC == 1 ? pv[0] * acc / fmaxf(pv[1] + sparsity, flr)
                                 : pv[0] * pv[1] / fmaxf(acc + sparsity, flr);
        H[row * N + col] = hn;
        return hn;
    }
};

// lambda of the odd bins: max(H[row,:] . W[Fm+i,:], flr), one wave per row
__global__ void __launch_bounds__(256)
mu_tail_kernel(const float* __restrict__ W, const float* __restrict__ H, float* __restrict__ P1,
               int64_t n, int N, int Fm, int nt, int ld, float flr) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63;
    if (row >= n) return;
    float s[2] = {0.f, 0.f};
    const float* hrow = H + row * N;
    if ((N & 3) == 0 && (((uintptr_t)H | (uintptr_t)W) & 15) == 0) {
        // 16-byte loads: this pass over H is pure bandwidth (one more read of the activations per
        // iteration), 4-byte loads left it at 3.2 TB/s
        const f32x4* h4 = (const f32x4*)hrow;
        const f32x4* w0 = (const f32x4*)(W + (size_t)Fm * N);
        const f32x4* w1 = (const f32x4*)(W + (size_t)(Fm + (nt > 1 ? 1 : 0)) * N);
        for (int c = l; c < N / 4; c += 64) {
            const f32x4 hv = h4[c], a0 = w0[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[0] = fmaf(hv[e], a0[e], s[0]);
            if (nt > 1) {
                const f32x4 a1 = w1[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) s[1] = fmaf(hv[e], a1[e], s[1]);
            }
        }
    } else {
        for (int c = l; c < N; c += 64) {
            const float hv = hrow[c];
            s[0] = fmaf(hv, W[(size_t)Fm * N + c], s[0]);
            if (nt > 1) s[1] = fmaf(hv, W[(size_t)(Fm + 1) * N + c], s[1]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s[0] += __shfl_xor(s[0], o, 64);
        s[1] += __shfl_xor(s[1], o, 64);
    }
    if (l < nt) P1[row * ld + Fm + l] = fmaxf(s[l], flr);
}

// column norms of W and sums of the normalised columns: sparse_nmf_gpu.m:163-166
__global__ void __launch_bounds__(256)
mu_colnorm_kernel(const float* __restrict__ W, float* __restrict__ Wn, float* __restrict__ norm,
                  float* __restrict__ colsum, int F, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int f = 0; f < F; ++f) {
        const float w = W[(size_t)f * N + n];
        s = fmaf(w, w, s);
    }
    const float nrm = sqrtf(s);
    float cs = 0.f;
    for (int f = 0; f < F; ++f) {
        const float w = W[(size_t)f * N + n] / nrm;
        Wn[(size_t)f * N + n] = w;
        cs += w;
    }
    norm[n] = nrm;
    colsum[n] = cs;
}

__global__ void __launch_bounds__(256)
mu_scale_h_kernel(float* __restrict__ H, const float* __restrict__ norm, int64_t total, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) H[i] *= norm[i % N];
}

// beta != 2: zeros of V are raised to the smallest positive entry (sparse_nmf_gpu.m:201-205)
__global__ void __launch_bounds__(256)
min_positive_kernel(const float* __restrict__ V, int64_t total, unsigned* out_bits) {
    float m = __int_as_float(0x7f800000);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        const float v = V[i];
        if (v > 0.f && v < m) m = v;
    }
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMin(out_bits, __float_as_uint(m));   // positive floats
}

__global__ void __launch_bounds__(256)
floor_zeros_kernel(const float* __restrict__ V, float* __restrict__ Vf, int64_t total,
                   const unsigned* min_bits) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) {
        const float v = V[i];
        Vf[i] = (v == 0.f) ? __uint_as_float(*min_bits) : v;
    }
}

struct IstaWs {
    size_t off_R, off_Wt, total;
    int Fp4;
};
IstaWs ista_ws(int64_t n, int F, int N) {
    IstaWs w;
    w.Fp4 = pad4(F);
    size_t o = 0;
    w.off_R = o;  o += round_up_sz((size_t)n * w.Fp4 * 4, 256);
    w.off_Wt = o; o += round_up_sz((size_t)N * w.Fp4 * 4, 256);
    w.total = o;
    return w;
}

struct MuWs {
    size_t off_P1, off_P2, off_Wt, off_DMH, off_norm, off_colsum, off_Vf, off_min, total;
    int Fp4;
};
MuWs mu_ws(int64_t n, int F, int N) {
    MuWs w;
    w.Fp4 = pad4(F);
    size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += round_up_sz(b, 256); return at; };
    w.off_P1 = take((size_t)n * w.Fp4 * 4);
    w.off_P2 = take((size_t)n * w.Fp4 * 4);
    w.off_Wt = take((size_t)N * w.Fp4 * 4);
    w.off_DMH = take((size_t)n * N * 4);
    w.off_norm = take((size_t)N * 4);
    w.off_colsum = take((size_t)N * 4);
    w.off_Vf = take((size_t)n * F * 4);
    w.off_min = take(256);
    w.total = o;
    return w;
}

}  // namespace

int head_irm_forward(drnmf_handle_t h, int64_t rows, int F, int r, const float* H, int64_t ld_h,
                     const float* Wn, float* irm, float* ecat, hipStream_t stream);

extern "C" size_t drnmf_ista_workspace_bytes(int64_t n, int32_t F, int32_t N) {
    if (n <= 0 || F <= 0 || N <= 0) return 0;
    return ista_ws(n, F, N).total;
}

extern "C" int32_t drnmf_ista_forward(drnmf_handle_t h, int64_t n, int32_t F, int32_t N, int32_t K,
                                      int32_t divergence, float beta, float lam1, float alph,
                                      const float* X, const float* W, float* H, void* workspace,
                                      size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n <= 0 || F <= 0 || N <= 0 || K < 0)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "ista_forward: bad shape n=%lld F=%d N=%d K=%d",
                   (long long)n, F, N, K);
    if (divergence < DRNMF_DIV_ED || divergence > DRNMF_DIV_BETA)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "ista_forward: unknown divergence %d", divergence);
    if (!X || !W || !H || !workspace)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "ista_forward: NULL pointer argument");
    if (!(alph > 0.f)) DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "ista_forward: alph must be > 0");
    const IstaWs L = ista_ws(n, F, N);
    if (workspace_bytes < L.total)
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "ista_forward: workspace %zu < required %zu",
                   workspace_bytes, L.total);
    hipStream_t stream = (hipStream_t)stream_;
    float* R = (float*)((char*)workspace + L.off_R);
    float* Wt = (float*)((char*)workspace + L.off_Wt);
    const int Fp4 = L.Fp4;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((Fp4 + 31) / 32, (N + 31) / 32), dim3(256), 0,
                       stream, W, Wt, F, N, Fp4);
    if (Fp4 != F) DRNMF_HIP(h, hipMemsetAsync(R, 0, (size_t)n * Fp4 * 4, stream));
    // odd bins (F = 16 j + 1 | 2, the 2^k + 1 STFT sizes) stay out of the GEMM tiles
    const int nt = (F % 16 != 0 && F % 16 <= 2 && F > 16) ? F % 16 : 0;
    const int Fm = F - nt;
    // ... riding on the X^ product's full tiles (gemm_nt.h THIN) when there is one of them behind whole
    // 128-column tiles and the operands allow 16-byte loads, else by ista_tail_kernel
    const char* te = measure_env("DRNMF_THIN");
    const bool thin = nt == 1 && Fm % gemm::BN == 0 && N % 4 == 0 && (((uintptr_t)H | (uintptr_t)W) & 15) == 0 &&
                      !(te && atoi(te) == 0);
    gemm::Operands g1{H, W, n, thin ? F : Fm, N, N, N};          // X^ = H . W^T  (contract atoms)
    gemm::Operands g2{R, Wt, n, N, nt ? Fm : Fp4, Fp4, Fp4, nt};   // G = R . W (contract bins; odd bins: ktail)
    const dim3 tgrid((unsigned)((n + 3) / 4));
    for (int k = 0; k < K; ++k) {
        if (divergence == DRNMF_DIV_ED) {
            DRNMF_HIP(h, gemm::launch(g1, EpiResidual<DRNMF_DIV_ED>{X, R, F, Fp4, beta}, stream));
            if (nt && !thin) hipLaunchKernelGGL(ista_tail_kernel<DRNMF_DIV_ED>, tgrid, dim3(256), 0, stream,
                                       X, W, H, R, n, F, N, Fm, nt, Fp4, beta);
        } else if (divergence == DRNMF_DIV_KL) {
            DRNMF_HIP(h, gemm::launch(g1, EpiResidual<DRNMF_DIV_KL>{X, R, F, Fp4, beta}, stream));
            if (nt && !thin) hipLaunchKernelGGL(ista_tail_kernel<DRNMF_DIV_KL>, tgrid, dim3(256), 0, stream,
                                       X, W, H, R, n, F, N, Fm, nt, Fp4, beta);
        } else {
            DRNMF_HIP(h, gemm::launch(g1, EpiResidual<DRNMF_DIV_BETA>{X, R, F, Fp4, beta}, stream));
            if (nt && !thin) hipLaunchKernelGGL(ista_tail_kernel<DRNMF_DIV_BETA>, tgrid, dim3(256), 0,
                                       stream, X, W, H, R, n, F, N, Fm, nt, Fp4, beta);
        }
        const float c0 = -lam1 / alph, c1 = 1.f / alph;
        DRNMF_HIP(h, gemm::launch(g2, EpiIstaUpdate{H, N, c0, c1}, stream));
    }
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

extern "C" size_t drnmf_mu_workspace_bytes(int64_t n, int32_t F, int32_t N) {
    if (n <= 0 || F <= 0 || N <= 0) return 0;
    const MuWs w = mu_ws(n, F, N);
    // + head scratch for the optional IRM: ecat [2*rp][Fp16]
    return w.total + round_up_sz((size_t)2 * round_up(N / 2 + 1, 16) * pad_f(F) * 4, 256);
}

extern "C" int32_t drnmf_mu_forward(drnmf_handle_t h, int64_t n, int32_t F, int32_t N,
                                    int32_t n_iter, float beta, float sparsity, const float* V,
                                    const float* W, float* Wn, float* H, float* irm,
                                    void* workspace, size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n <= 0 || F <= 0 || N <= 0 || n_iter < 0)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "mu_forward: bad shape n=%lld F=%d N=%d iters=%d",
                   (long long)n, F, N, n_iter);
    if (!V || !W || !Wn || !H || !workspace)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "mu_forward: NULL pointer argument");
    if (irm && (N % 2))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "mu_forward: IRM needs an even number of atoms");
    if (workspace_bytes < drnmf_mu_workspace_bytes(n, F, N))
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "mu_forward: workspace %zu < required %zu",
                   workspace_bytes, drnmf_mu_workspace_bytes(n, F, N));
    hipStream_t stream = (hipStream_t)stream_;
    const MuWs L = mu_ws(n, F, N);
    char* ws = (char*)workspace;
    float* P1 = (float*)(ws + L.off_P1);
    float* P2 = (float*)(ws + L.off_P2);
    float* Wt = (float*)(ws + L.off_Wt);
    float* DMH = (float*)(ws + L.off_DMH);
    float* norm = (float*)(ws + L.off_norm);
    float* colsum = (float*)(ws + L.off_colsum);
    float* Vf = (float*)(ws + L.off_Vf);
    unsigned* minb = (unsigned*)(ws + L.off_min);
    float* ecat = (float*)(ws + L.total);
    const int Fp4 = L.Fp4;
    const float flr = 1e-9f;                                            // sparse_nmf_gpu.m:172
    const int64_t nH = n * (int64_t)N, nV = n * (int64_t)F;

    // normalise the columns of W, rescale H (sparse_nmf_gpu.m:163-166)
    hipLaunchKernelGGL(mu_colnorm_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, W, Wn, norm,
                       colsum, F, N);
    hipLaunchKernelGGL(mu_scale_h_kernel, dim3((unsigned)((nH + 255) / 256)), dim3(256), 0, stream,
                       H, norm, nH, N);
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((Fp4 + 31) / 32, (N + 31) / 32), dim3(256), 0,
                       stream, Wn, Wt, F, N, Fp4);
    const float* Vuse = V;
    if (beta != 2.f) {   // sparse_nmf_gpu.m:201-205
        DRNMF_HIP(h, hipMemsetAsync(minb, 0x7f, 4, stream));   // 0x7f7f7f7f: a huge positive float
        hipLaunchKernelGGL(min_positive_kernel, dim3(1024), dim3(256), 0, stream, V, nV, minb);
        hipLaunchKernelGGL(floor_zeros_kernel, dim3((unsigned)((nV + 255) / 256)), dim3(256), 0,
                           stream, V, Vf, nV, minb);
        Vuse = Vf;
    }
    if (Fp4 != F) {
        DRNMF_HIP(h, hipMemsetAsync(P1, 0, (size_t)n * Fp4 * 4, stream));
        DRNMF_HIP(h, hipMemsetAsync(P2, 0, (size_t)n * Fp4 * 4, stream));
    }
    // beta == 2: odd bins (F = 16 j + 1 | 2) stay out of the GEMM tiles, as in drnmf_ista_forward
    const int nt = (beta == 2.f && F % 16 != 0 && F % 16 <= 2 && F > 16) ? F % 16 : 0;
    const int Fm = F - nt;
    // (the odd bin's lambda: rides on the product's full tiles when the shape allows, as in
    // drnmf_ista_forward, else mu_tail_kernel)
    const char* te = measure_env("DRNMF_THIN");
    const bool thin = nt == 1 && Fm % gemm::BN == 0 && N % 4 == 0 && (((uintptr_t)H | (uintptr_t)Wn) & 15) == 0 &&
                      !(te && atoi(te) == 0);
    gemm::Operands gl{H, Wn, n, thin ? F : Fm, N, N, N};           // W H  (row layout: H . Wn^T)
    auto launch_lambda = [&]() -> hipError_t {
        if (beta == 2.f) {
            hipError_t e = gemm::launch(gl, EpiLambda<2>{Vuse, P1, P2, nullptr, F, Fp4, beta, flr}, stream);
            if (e == hipSuccess && nt && !thin)
                hipLaunchKernelGGL(mu_tail_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0,
                                   stream, Wn, H, P1, n, N, Fm, nt, Fp4, flr);
            return e;
        }
        if (beta == 1.f) return gemm::launch(gl, EpiLambda<1>{Vuse, P1, P2, nullptr, F, Fp4, beta, flr}, stream);
        return gemm::launch(gl, EpiLambda<0>{Vuse, P1, P2, nullptr, F, Fp4, beta, flr}, stream);
    };
    if (beta == 2.f) {
        // dmh = W^T V is loop invariant: one GEMM V . W up front (V copied to a padded buffer so
        // the contraction over bins can use 16-byte loads)
        DRNMF_HIP(h, hipMemcpy2DAsync(P2, (size_t)Fp4 * 4, V, (size_t)F * 4, (size_t)F * 4,
                                      (size_t)n, hipMemcpyDeviceToDevice, stream));
        gemm::Operands gd{P2, Wt, n, N, Fp4, Fp4, Fp4};
        DRNMF_HIP(h, gemm::launch(gd, EpiStore{DMH, N}, stream));
    }
    DRNMF_HIP(h, launch_lambda());                      // lambda before the loop (:173)
    for (int it = 0; it < n_iter; ++it) {
        if (beta != 2.f && beta != 1.f) {
            gemm::Operands gd{P2, Wt, n, N, Fp4, Fp4, Fp4};
            DRNMF_HIP(h, gemm::launch(gd, EpiStore{DMH, N}, stream));
        }
        gemm::Operands gu{P1, Wt, n, N, nt ? Fm : Fp4, Fp4, Fp4, nt};   // odd bins: ktail
        if (beta == 1.f)
            DRNMF_HIP(h, gemm::launch(gu, EpiMuUpdate<1>{H, DMH, colsum, N, sparsity, flr}, stream));
        else
            DRNMF_HIP(h, gemm::launch(gu, EpiMuUpdate<2>{H, DMH, colsum, N, sparsity, flr}, stream));
        DRNMF_HIP(h, launch_lambda());                  // lambda = max(w*h, flr)  (:228)
    }
    if (irm) {
        int rc = head_irm_forward(h, n, F, N / 2, H, N, Wn, irm, ecat, stream);
        if (rc) return rc;
    }
    return DRNMF_OK;
}


// =====================================================================================================
// Dictionary training: multiplicative W and H updates in the column-normalised basis
// (sparseNMF/sparse_nmf_gpu.m:156-298; driven by snmf.py:9-113 and enhance.py:81-135).
// One iteration = H update, lambda, W update (masked columns) + renormalisation, lambda, objective.
// Frame contractions (V^T H, lambda^T H) are split-K TN GEMMs; everything elementwise rides in
// GEMM epilogues or thread-per-column kernels.  Row layout as above (frames are rows).
// =====================================================================================================
namespace {

// split-K count of the W-update statistics' TN products (contraction over the n frames): chosen per shape
// so that tiles x splits fills the chip in whole rounds (gemm_tn::pick_splits; 513 x 1000 on 32k frames: 12)
constexpr int TR_MAX_SPLITS = 64;
static inline int tr_splits(int64_t n, int Fp4, int N) { return gemm_tn::pick_splits(Fp4, N, n, TR_MAX_SPLITS); }

struct EpiPartTN {
    float* P;
    int ld;
    size_t stride;
    __device__ float pre(int, int, int) const { return 0.f; }
    __device__ void operator()(int split, int m, int n, float acc, float) const {
        P[split * stride + (size_t)m * ld + n] = acc;
    }
};

// Odd rows of the W-update statistics.  F = 2^k + 1 bins put ONE row (4 with the padding to Fp4) into a
// fifth 128-row tile of the two TN products NUM = V'^T H, DEN = P1^T H -- a fifth of their time for
// 1/128 of a tile.  The TN products cover the first Mg = F - F % 128 rows; the 1..TN_TAIL_MAX rows
// behind them are dot products of H's columns with single columns of V' / P1, one streaming pass over H
// shared by both statistics: tailp[split][src][j][n], TT_SPLITS row ranges, summed by w_fold_kernel.
constexpr int TN_TAIL_MAX = 4, TT_SPLITS = 512;   // (128 row ranges: 101 us for the 131-MB H, half the CUs idle)
static inline int tn_tail_rows(int F) { return (F > 128 && F % 128 >= 1 && F % 128 <= TN_TAIL_MAX) ? F % 128 : 0; }
// VEC4 (N % 4 == 0): four columns per thread, 16-byte loads of H, eight rows in flight (one column and
// four rows per thread ran the 131 MB of a 32k-frame H at 0.56 TB/s: 234 us)
template <bool VEC4>
__global__ void __launch_bounds__(256)
tn_tail_kernel(const float* __restrict__ A0, const float* __restrict__ A1, const float* __restrict__ H,
               float* __restrict__ tailp, int64_t n, int N, int lda, int Mg, int ntail) {
    constexpr int CW = VEC4 ? 4 : 1, U = 8;
    const int col = (blockIdx.x * 256 + threadIdx.x) * CW;
    if (col >= N) return;
    const int64_t per = (n + TT_SPLITS - 1) / TT_SPLITS;
    const int64_t r0 = blockIdx.y * per;
    int64_t r1 = r0 + per;
    if (r1 > n) r1 = n;
    float a0[TN_TAIL_MAX][CW], a1[TN_TAIL_MAX][CW];
#pragma unroll
    for (int j = 0; j < TN_TAIL_MAX; ++j)
#pragma unroll
        for (int c = 0; c < CW; ++c) a0[j][c] = a1[j][c] = 0.f;
    for (int64_t t = r0; t < r1; t += U) {
        float hv[U][CW];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t tt = t + u < r1 ? t + u : r1 - 1;          // (clamped: weighted by 0 below)
            if (VEC4) {
                const f32x4 v = *(const f32x4*)(H + tt * N + col);
#pragma unroll
                for (int c = 0; c < CW; ++c) hv[u][c] = v[c];
            } else {
                hv[u][0] = H[tt * N + col];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = t + u < r1;
            const int64_t tt = ok ? t + u : r1 - 1;
#pragma unroll
            for (int j = 0; j < TN_TAIL_MAX; ++j) {
                if (j >= ntail) break;
                const float w0 = ok ? A0[tt * lda + Mg + j] : 0.f;       // (wave-uniform address)
                const float w1 = (ok && A1) ? A1[tt * lda + Mg + j] : 0.f;
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    a0[j][c] = fmaf(w0, hv[u][c], a0[j][c]);
                    a1[j][c] = fmaf(w1, hv[u][c], a1[j][c]);
                }
            }
        }
    }
    for (int j = 0; j < ntail; ++j)
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            tailp[(((size_t)blockIdx.y * 2 + 0) * TN_TAIL_MAX + j) * N + col + c] = a0[j][c];
            tailp[(((size_t)blockIdx.y * 2 + 1) * TN_TAIL_MAX + j) * N + col + c] = a1[j][c];
        }
}

// tailp[TT_SPLITS][2][TN_TAIL_MAX][N] -> tailr[2][TN_TAIL_MAX][N] (stored behind the partials): 64 columns x 4
// lanes of row ranges per workgroup, eight partials in flight per lane, the four lanes added in LDS in a
// fixed order.  (Summed by w_fold_kernel's one thread per column, the 512 partials were a 178-us serial chain.)
__global__ void __launch_bounds__(256)
tn_tail_reduce_kernel(const float* __restrict__ tailp, float* __restrict__ tailr, int N, int ntail) {
    __shared__ float part[4][64];
    if ((int)(blockIdx.y % TN_TAIL_MAX) >= ntail) return;                  // (rows the tail pass never wrote)
    const int cl = threadIdx.x & 63, lane = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl, sj = blockIdx.y;                 // sj = src * TN_TAIL_MAX + j
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        constexpr int PER = TT_SPLITS / 4;
        for (int s0 = lane * PER; s0 < (lane + 1) * PER; s0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                a8[u] += tailp[((size_t)(s0 + u) * 2 * TN_TAIL_MAX + sj) * N + col];
        }
    }
    part[lane][cl] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    __syncthreads();
    if (lane == 0 && col < N)
        tailr[(size_t)sj * N + col] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

// column sums over frames of H: hs[n] (beta == 1 W update)
__global__ void __launch_bounds__(256)
colsum_rows_kernel(const float* __restrict__ H, float* __restrict__ part, int64_t n, int N,
                   int splits) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= N) return;
    const int64_t per = (n + splits - 1) / splits;
    const int64_t r0 = blockIdx.y * per;
    int64_t r1 = r0 + per;
    if (r1 > n) r1 = n;
    // eight rows in flight per thread (one dependent load per row ran the 131-MB H of a 32k-frame KL
    // iteration at 0.84 TB/s); the eight chains are added in a fixed order
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int64_t r = r0;
    for (; r + 8 <= r1; r += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = H[(r + u) * N + col];
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[u] += v[u];
    }
    for (; r < r1; ++r) s8[0] += H[r * N + col];
    part[(size_t)blockIdx.y * N + col] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
}

// W update (sparse_nmf_gpu.m:232-262) + L2 renormalisation of every column:
//   dpw = DEN + W * sum_f(NUM * W);  dmw = NUM + W * sum_f(DEN * W);  W <- W * dmw / max(dpw, flr)
// NUM/DEN are sums of the split-K partial [Fp4][N] buffers; for beta == 1, DEN[f][n] = hs[n].
// Three passes over (atom, group of W_FB bins) -- fold + the two per-atom sums; apply + sum of squares;
// scale -- each on (N/256) x (F/W_FB) workgroups with per-group partials summed in a fixed order.  (One
// thread per atom walking all F bins and 2 x 8 partial buffers, twice, ran a 513 x 1000
// dictionary on FOUR workgroups: 0.80 ms of a 3.4-ms iteration, profiles/r04c_snmf_train_kernel_stats.csv.)
constexpr int W_FB = 8;
__global__ void __launch_bounds__(256)
w_fold_kernel(const float* __restrict__ W, float* __restrict__ PN, float* __restrict__ PD,
              const float* __restrict__ hs_part, float* __restrict__ wpart, int F, int N, size_t stride,
              int hs_splits, int beta_is_one, int nsplit, const float* __restrict__ tailp, int Mg) {
    const int n = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (n >= N) return;
    float hs = 0.f;
    if (beta_is_one) hs = ordered_sum<16>(hs_part + n, (size_t)N, hs_splits);
    float sn = 0.f, sd = 0.f;
    const int f0 = g * W_FB;
    // the group's W_FB bins side by side: W_FB (x 2) independent loads in flight per split (bin by bin the
    // fold was one dependent load after the other: 67 us for 50 MB); per (bin, atom) the splits are still
    // added in their fixed order
    float num[W_FB], den[W_FB];
#pragma unroll
    for (int i = 0; i < W_FB; ++i) { num[i] = 0.f; den[i] = beta_is_one ? hs : 0.f; }
    // (four splits' loads in flight; bins outside the products re-read a valid element and add +0)
    for (int s0 = 0; s0 < nsplit; s0 += 4) {
        float vn[4][W_FB], vd[4][W_FB];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int sp = s0 + u < nsplit ? s0 + u : nsplit - 1;
#pragma unroll
            for (int i = 0; i < W_FB; ++i) {
                const int f = f0 + i;
                const size_t o = (size_t)((f < F && f < Mg) ? f : 0) * N + n;
                vn[u][i] = PN[sp * stride + o];
                if (!beta_is_one) vd[u][i] = PD[sp * stride + o];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool su = s0 + u < nsplit;
#pragma unroll
            for (int i = 0; i < W_FB; ++i) {
                const int f = f0 + i;
                const bool ok = su && f < F && f < Mg;
                num[i] += ok ? vn[u][i] : 0.f;
                if (!beta_is_one) den[i] += ok ? vd[u][i] : 0.f;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < W_FB; ++i) {
        const int f = f0 + i;
        if (f >= F) break;
        const size_t o = (size_t)f * N + n;
        if (f >= Mg) {                                 // odd rows: tn_tail_kernel's sums, folded by tn_tail_reduce_kernel
            const float* tailr = tailp + (size_t)TT_SPLITS * 2 * TN_TAIL_MAX * N;
            num[i] += tailr[(size_t)(0 * TN_TAIL_MAX + (f - Mg)) * N + n];
            if (!beta_is_one) den[i] += tailr[(size_t)(1 * TN_TAIL_MAX + (f - Mg)) * N + n];
        }
        PN[o] = num[i];                                // (folded in place: slot 0)
        if (!beta_is_one) PD[o] = den[i];
        const float w = W[o];
        sn = fmaf(num[i], w, sn);
        sd = fmaf(den[i], w, sd);
    }
    wpart[((size_t)g * 3 + 0) * N + n] = sn;
    wpart[((size_t)g * 3 + 1) * N + n] = sd;
    if (g == 0 && beta_is_one) wpart[((size_t)gridDim.y * 3) * N + n] = hs;     // DEN of every bin
}
__global__ void __launch_bounds__(256)
w_apply_kernel(float* __restrict__ W, const float* __restrict__ PN, const float* __restrict__ PD,
               float* __restrict__ wpart, const unsigned char* __restrict__ upd, int F, int N,
               int beta_is_one, float flr) {
    const int n = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y, ng = gridDim.y;
    if (n >= N) return;
    const bool do_upd = upd ? upd[n] != 0 : true;
    const float sn = ordered_sum<16>(wpart + n, (size_t)3 * N, ng);
    const float sd = ordered_sum<16>(wpart + (size_t)N + n, (size_t)3 * N, ng);
    const float hs = beta_is_one ? wpart[((size_t)ng * 3) * N + n] : 0.f;
    float s2 = 0.f;
    float wv[W_FB], nv[W_FB], dv[W_FB];
#pragma unroll
    for (int i = 0; i < W_FB; ++i) {                   // (all loads first: bins past F re-read the group's first)
        const int f = g * W_FB + i;
        const size_t o = (size_t)(f < F ? f : g * W_FB) * N + n;
        wv[i] = W[o];
        nv[i] = PN[o];
        dv[i] = beta_is_one ? hs : PD[o];
    }
#pragma unroll
    for (int i = 0; i < W_FB; ++i) {
        const int f = g * W_FB + i;
        if (f >= F) break;
        float w = wv[i];
        if (do_upd) {
            const float dpw = fmaxf(dv[i] + w * sn, flr);
            const float dmw = nv[i] + w * sd;
            w = w * dmw / dpw;
            W[(size_t)f * N + n] = w;
        }
        s2 = fmaf(w, w, s2);
    }
    wpart[((size_t)g * 3 + 2) * N + n] = s2;
}
__global__ void __launch_bounds__(256)
w_norm_kernel(float* __restrict__ W, const float* __restrict__ wpart, int F, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y, ng = gridDim.y;
    if (n >= N) return;
    const float s2 = ordered_sum<16>(wpart + (size_t)2 * N + n, (size_t)3 * N, ng);
    const float inv = 1.f / sqrtf(s2);
    float wv[W_FB];
#pragma unroll
    for (int i = 0; i < W_FB; ++i) {
        const int f = g * W_FB + i;
        wv[i] = W[(size_t)(f < F ? f : g * W_FB) * N + n];
    }
#pragma unroll
    for (int i = 0; i < W_FB; ++i) {
        const int f = g * W_FB + i;
        if (f < F) W[(size_t)f * N + n] = wv[i] * inv;
    }
}

// objective from the per-workgroup partials of the two fused reductions (EpiLambdaObj, EpiMuUpdateSum)
__global__ void __launch_bounds__(256)
objective_final2_kernel(const float* __restrict__ dpart, int nd, const float* __restrict__ hpart, int nh,
                        float sparsity, float* __restrict__ obj) {
    __shared__ double sd[256], sh[256];
    double d = 0.0, hs = 0.0;
    for (int i = threadIdx.x; i < nd; i += 256) d += (double)dpart[i];
    for (int i = threadIdx.x; i < nh; i += 256) hs += (double)hpart[i];
    sd[threadIdx.x] = d;
    sh[threadIdx.x] = hs;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sd[threadIdx.x] += sd[threadIdx.x + o];
            sh[threadIdx.x] += sh[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        obj[0] = (float)sd[0];
        obj[1] = (float)(sd[0] + (double)sparsity * sh[0]);
    }
}

constexpr int HS_SPLITS = 64;

struct TrWs {
    size_t off_P1, off_P2, off_Vp, off_Wt, off_DMH, off_norm, off_colsum, off_PN, off_PD,
        off_hs, off_obj, off_min, off_wpart, off_tailp, total;
    int Fp4;
};
TrWs tr_ws(int64_t n, int F, int N) {
    TrWs w;
    w.Fp4 = pad4(F);
    size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += round_up_sz(b, 256); return at; };
    const size_t nf = (size_t)n * w.Fp4 * 4;
    w.off_P1 = take(nf); w.off_P2 = take(nf); w.off_Vp = take(nf);
    w.off_Wt = take((size_t)N * w.Fp4 * 4);
    w.off_DMH = take((size_t)n * N * 4);
    w.off_norm = take((size_t)N * 4);
    w.off_colsum = take((size_t)N * 4);
    const int nsplit = tr_splits(n, tn_tail_rows(F) ? F - tn_tail_rows(F) : w.Fp4, N);
    w.off_PN = take((size_t)nsplit * w.Fp4 * N * 4);
    w.off_PD = take((size_t)nsplit * w.Fp4 * N * 4);
    w.off_hs = take((size_t)HS_SPLITS * N * 4);
    w.off_obj = take((size_t)((n + 127) / 128) * (((F + 127) / 128) + ((N + 127) / 128)) * 4);
    w.off_min = take(256);
    w.off_wpart = take(((size_t)((F + W_FB - 1) / W_FB) * 3 + 1) * N * 4);
    w.off_tailp = take((size_t)(TT_SPLITS + 1) * 2 * TN_TAIL_MAX * N * 4);      // partials + their sums
    w.total = o;
    return w;
}

template <int BC>
hipError_t launch_lambda_train(const gemm::Operands& gl, const float* Vp, float* P1, float* P2,
                               float* LAM, int Fp4, float beta, float flr, hipStream_t stream) {
    // V is read from the padded copy (ld = Fp4)
    return gemm::launch(gl, EpiLambda<BC>{Vp, P1, P2, LAM, Fp4, Fp4, beta, flr}, stream);
}
hipError_t lambda_train(const gemm::Operands& gl, const float* Vp, float* P1, float* P2, float* LAM,
                        int Fp4, float beta, float flr, hipStream_t stream) {
    if (beta == 2.f) return launch_lambda_train<2>(gl, Vp, P1, P2, LAM, Fp4, beta, flr, stream);
    if (beta == 1.f) return launch_lambda_train<1>(gl, Vp, P1, P2, LAM, Fp4, beta, flr, stream);
    return launch_lambda_train<0>(gl, Vp, P1, P2, LAM, Fp4, beta, flr, stream);
}

template <int BC>
hipError_t launch_lambda_obj(const gemm::Operands& gl, const float* Vp, float* P1, float* P2, float* dpart,
                             int Fp4, float beta, float flr, hipStream_t stream) {
    return gemm::launch(gl, EpiLambdaObj<BC>{Vp, P1, P2, dpart, Fp4, Fp4, beta, flr}, stream);
}
hipError_t lambda_train_obj(const gemm::Operands& gl, const float* Vp, float* P1, float* P2, float* dpart,
                            int Fp4, float beta, float flr, hipStream_t stream) {
    if (beta == 2.f) return launch_lambda_obj<2>(gl, Vp, P1, P2, dpart, Fp4, beta, flr, stream);
    if (beta == 1.f) return launch_lambda_obj<1>(gl, Vp, P1, P2, dpart, Fp4, beta, flr, stream);
    return launch_lambda_obj<0>(gl, Vp, P1, P2, dpart, Fp4, beta, flr, stream);
}

}  // namespace

extern "C" size_t drnmf_snmf_train_workspace_bytes(int64_t n, int32_t F, int32_t N) {
    if (n <= 0 || F <= 0 || N <= 0) return 0;
    return tr_ws(n, F, N).total;
}

extern "C" int32_t drnmf_snmf_train_init(drnmf_handle_t h, int64_t n, int32_t F, int32_t N, float beta,
                                         const float* V, float* W, float* H, void* workspace,
                                         size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n <= 0 || F <= 0 || N <= 0 || !V || !W || !H || !workspace)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "snmf_train_init: bad argument");
    const TrWs L = tr_ws(n, F, N);
    if (workspace_bytes < L.total)
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "snmf_train_init: workspace %zu < required %zu",
                   workspace_bytes, L.total);
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    float* P1 = (float*)(ws + L.off_P1); float* P2 = (float*)(ws + L.off_P2);
    float* Vp = (float*)(ws + L.off_Vp);
    float* Wt = (float*)(ws + L.off_Wt);
    float* norm = (float*)(ws + L.off_norm); float* colsum = (float*)(ws + L.off_colsum);
    unsigned* minb = (unsigned*)(ws + L.off_min);
    const int Fp4 = L.Fp4;
    const float flr = 1e-9f;
    const int64_t nH = n * (int64_t)N, nV = n * (int64_t)F;
    DRNMF_HIP(h, hipMemsetAsync(ws, 0, L.off_Wt, stream));   // P1, P2, Vp (zero padding)
    const float* Vsrc = V;
    if (beta != 2.f) {   // zeros of V are raised to its smallest positive entry (:201-205)
        DRNMF_HIP(h, hipMemsetAsync(minb, 0x7f, 4, stream));
        hipLaunchKernelGGL(min_positive_kernel, dim3(1024), dim3(256), 0, stream, V, nV, minb);
        hipLaunchKernelGGL(floor_zeros_kernel, dim3((unsigned)((nV + 255) / 256)), dim3(256), 0,
                           stream, V, P1, nV, minb);   // P1 as scratch [n][F]
        Vsrc = P1;
    }
    DRNMF_HIP(h, hipMemcpy2DAsync(Vp, (size_t)Fp4 * 4, Vsrc, (size_t)F * 4, (size_t)F * 4, (size_t)n,
                                  hipMemcpyDeviceToDevice, stream));
    if (beta != 2.f) DRNMF_HIP(h, hipMemsetAsync(P1, 0, (size_t)n * Fp4 * 4, stream));
    // normalise W in place, rescale H (:163-166)
    hipLaunchKernelGGL(mu_colnorm_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, W, W, norm,
                       colsum, F, N);
    hipLaunchKernelGGL(mu_scale_h_kernel, dim3((unsigned)((nH + 255) / 256)), dim3(256), 0, stream,
                       H, norm, nH, N);
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((Fp4 + 31) / 32, (N + 31) / 32), dim3(256), 0,
                       stream, W, Wt, F, N, Fp4);
    gemm::Operands gl{H, W, n, F, N, N, N};
    DRNMF_HIP(h, lambda_train(gl, Vp, P1, P2, nullptr, Fp4, beta, flr, stream));   // (:173)
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

extern "C" int32_t drnmf_snmf_train_step(drnmf_handle_t h, int64_t n, int32_t F, int32_t N, float beta,
                                         float sparsity, float* W, float* H,
                                         const unsigned char* w_update_mask, int32_t update_w,
                                         float* obj, void* workspace, size_t workspace_bytes,
                                         void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n <= 0 || F <= 0 || N <= 0 || !W || !H || !obj || !workspace)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "snmf_train_step: bad argument");
    const TrWs L = tr_ws(n, F, N);
    if (workspace_bytes < L.total)
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "snmf_train_step: workspace %zu < required %zu",
                   workspace_bytes, L.total);
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    float* P1 = (float*)(ws + L.off_P1); float* P2 = (float*)(ws + L.off_P2);
    float* Vp = (float*)(ws + L.off_Vp);
    float* Wt = (float*)(ws + L.off_Wt); float* DMH = (float*)(ws + L.off_DMH);
    float* colsum = (float*)(ws + L.off_colsum);
    float* PN = (float*)(ws + L.off_PN); float* PD = (float*)(ws + L.off_PD);
    float* hs = (float*)(ws + L.off_hs);
    const int nd = (int)((n + 127) / 128) * ((F + 127) / 128);      // (capacity of dpart: tr_ws)
    float* dpart = (float*)(ws + L.off_obj);
    float* hpart = dpart + nd;
    const int Fp4 = L.Fp4;
    const float flr = 1e-9f;
    gemm::Operands gl{H, W, n, F, N, N, N};
    // (nd / nh size the two partial arrays; what the REDUCE launches below actually leave there is one
    // partial per workgroup THEY run -- fewer when the odd column rides on the full tiles)
    const int nd_used = (int)gemm::launch_tiles(gl);
    const int nh_used = (int)gemm::launch_tiles(gemm::Operands{P1, Wt, n, N, Fp4, Fp4, Fp4});
    // ---- H update (:210-229) -------------------------------------------------------------------
    if (beta == 1.f) {
        // dph = sum_f w + sparsity needs the column sums of the CURRENT W
        hipLaunchKernelGGL(colsum_rows_kernel, dim3((N + 255) / 256, 1), dim3(256), 0, stream, W,
                           colsum, (int64_t)F, N, 1);
        gemm::Operands gu{P1, Wt, n, N, Fp4, Fp4, Fp4};
        DRNMF_HIP(h, gemm::launch(gu, EpiMuUpdateSum<1>{H, DMH, colsum, hpart, N, sparsity, flr}, stream));
    } else {
        gemm::Operands gd{beta == 2.f ? Vp : P2, Wt, n, N, Fp4, Fp4, Fp4};
        DRNMF_HIP(h, gemm::launch(gd, EpiStore{DMH, N}, stream));
        gemm::Operands gu{P1, Wt, n, N, Fp4, Fp4, Fp4};
        DRNMF_HIP(h, gemm::launch(gu, EpiMuUpdateSum<2>{H, DMH, colsum, hpart, N, sparsity, flr}, stream));
    }
    // (:228) -- the iteration's LAST lambda is the one the objective is evaluated on (:263, :267-281)
    if (update_w) DRNMF_HIP(h, lambda_train(gl, Vp, P1, P2, nullptr, Fp4, beta, flr, stream));
    else DRNMF_HIP(h, lambda_train_obj(gl, Vp, P1, P2, dpart, Fp4, beta, flr, stream));
    // ---- W update (:232-264) -------------------------------------------------------------------
    if (update_w) {
        const size_t pstr = (size_t)Fp4 * N;
        const int ntail = tn_tail_rows(F), Mg = ntail ? F - ntail : Fp4;
        const int nsplit = tr_splits(n, Mg, N);
        float* tailp = (float*)(ws + L.off_tailp);
        const float* num_src = beta == 2.f ? Vp : (beta == 1.f ? P1 : P2);
        if (ntail) {
            const float* den_src = beta == 1.f ? (const float*)nullptr : (const float*)P1;
            if (N % 4 == 0)
                hipLaunchKernelGGL(tn_tail_kernel<true>, dim3((N / 4 + 255) / 256, TT_SPLITS), dim3(256), 0, stream,
                                   num_src, den_src, H, tailp, n, N, Fp4, Mg, ntail);
            else
                hipLaunchKernelGGL(tn_tail_kernel<false>, dim3((N + 255) / 256, TT_SPLITS), dim3(256), 0, stream,
                                   num_src, den_src, H, tailp, n, N, Fp4, Mg, ntail);
            hipLaunchKernelGGL(tn_tail_reduce_kernel, dim3((N + 63) / 64, 2 * TN_TAIL_MAX), dim3(256), 0, stream,
                               tailp, tailp + (size_t)TT_SPLITS * 2 * TN_TAIL_MAX * N, N, ntail);
        }
        gemm_tn::Operands tn{num_src, H, n, Mg, N, Fp4, N};
        DRNMF_HIP(h, gemm_tn::launch(tn, EpiPartTN{PN, N, pstr}, nsplit, stream));
        if (beta == 1.f) {
            hipLaunchKernelGGL(colsum_rows_kernel, dim3((N + 255) / 256, HS_SPLITS), dim3(256), 0,
                               stream, H, hs, n, N, HS_SPLITS);
        } else {
            gemm_tn::Operands td{P1, H, n, Mg, N, Fp4, N};
            DRNMF_HIP(h, gemm_tn::launch(td, EpiPartTN{PD, N, pstr}, nsplit, stream));
        }
        {
            float* wpart = (float*)(ws + L.off_wpart);
            const dim3 wgrid((N + 255) / 256, (F + W_FB - 1) / W_FB);
            const int b1 = beta == 1.f ? 1 : 0;
            hipLaunchKernelGGL(w_fold_kernel, wgrid, dim3(256), 0, stream, W, PN, PD, hs, wpart, F, N, pstr,
                               HS_SPLITS, b1, nsplit, tailp, ntail ? Mg : F);
            hipLaunchKernelGGL(w_apply_kernel, wgrid, dim3(256), 0, stream, W, PN, PD, wpart, w_update_mask,
                               F, N, b1, flr);
            hipLaunchKernelGGL(w_norm_kernel, wgrid, dim3(256), 0, stream, W, wpart, F, N);
        }
        hipLaunchKernelGGL(transpose_pad_kernel, dim3((Fp4 + 31) / 32, (N + 31) / 32), dim3(256), 0,
                           stream, W, Wt, F, N, Fp4);
        DRNMF_HIP(h, lambda_train_obj(gl, Vp, P1, P2, dpart, Fp4, beta, flr, stream));   // (:263)
    }
    // ---- objective (:267-281): D(V | lambda) and sum(H) were summed in the epilogues above ------
    hipLaunchKernelGGL(objective_final2_kernel, dim3(1), dim3(256), 0, stream, dpart, nd_used, hpart, nh_used,
                       sparsity, obj);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}
