// Training-side pieces around the cell's BPTT: the loss head + mask-head backward and the Adam
// update.  Reference: y_pred = x_raw * mask, loss 'mse' with temporal sample weights
// (enhance.py:1040-1048, 1071-1073, 1152), Adam(lr, clipnorm, decay) (enhance.py:1052-1057).
//
// Gradients are produced UNNORMALISED (loss' = sum_{b,t} w * mean_f (x*mask - y)^2) together with
// (sum, count): data-parallel ranks all-reduce the flat gradient and the two scalars and apply the
// 1/count normalisation afterwards (Keras normalises by batch-level statistics, so averaging
// per-rank normalised gradients would be wrong when ranks hold different numbers of valid frames).
#include "gemm_nt.h"
#include "gemm_tn.h"

namespace {

constexpr int HB_SPLITS = 64;    // upper bound; per shape: gemm_tn::pick_splits
inline int pad4i(int v) { return (v + 3) / 4 * 4; }

__global__ void __launch_bounds__(256)
exp_pad_kernel(const float* __restrict__ kc, const float* __restrict__ kn, float* __restrict__ E,
               int r, int F, int Fp4) {   // E[seg][k][f], ld Fp4, zero padded
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)2 * r * Fp4) return;
    const int f = (int)(i % Fp4);
    const int k = (int)((i / Fp4) % r);
    const int seg = (int)(i / ((size_t)Fp4 * r));
    E[i] = f < F ? expf((seg ? kn : kc)[(size_t)k * F + f]) : 0.f;
}

// per element: err = x*m - y; partial sums of w*err^2/F; dA, dBn of the unnormalised loss
__global__ void __launch_bounds__(256)
loss_grad_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                 const float* __restrict__ A, const float* __restrict__ Bn,
                 const float* __restrict__ y, const float* __restrict__ w, float* __restrict__ dA,
                 float* __restrict__ dBn, float* __restrict__ part, int64_t rows, int F, int Fp4,
                 int square) {
    __shared__ float ssum[4], scnt[4];
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;
    float acc = 0.f, cnt = 0.f;
    if (row < rows) {
        const float wt = w[row];
        const float invF = 1.f / (float)F;
        for (int f = l; f < Fp4; f += 64) {
            float da = 0.f, db = 0.f;
            if (f < F) {
                const size_t o = (size_t)row * F + f;
                const float xv = x[o], m = mask[o], a = A[o], b = Bn[o];
                const float err = xv * m - y[o];
                acc += wt * err * err * invF;
                const float dm = 2.f * wt * err * xv * invF;
                const float S = 1e-7f + a + b;
                const float iS2 = 1.f / (S * S);
                da = dm * b * iS2;                 // d mask / dA  =  Bn / S^2
                db = -dm * (1e-7f + a) * iS2;      // d mask / dBn = -(eps + A) / S^2
                if (square) {                      // A = A0^2 (enhance.py:298-299)
                    da *= 2.f * sqrtf(a);
                    db *= 2.f * sqrtf(b);
                }
            }
            dA[(size_t)row * Fp4 + f] = da;
            dBn[(size_t)row * Fp4 + f] = db;
        }
        cnt = (l == 0 && wt != 0.f) ? 1.f : 0.f;
    }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_xor(acc, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    if (l == 0) { ssum[wv] = acc; scnt[wv] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * (size_t)blockIdx.x + 0] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        part[2 * (size_t)blockIdx.x + 1] = (scnt[0] + scnt[1]) + (scnt[2] + scnt[3]);
    }
}

// SNMF-cost pretraining loss (enhance.py:1023-1035): outputs [x_recon = A + Bn, h], targets [x, x],
// losses ['mse', mean|h|] with weights [0.5, lam1*N/F]  ==  (0.5 |x - x_recon|^2 + lam1 |h|_1) / F
// per frame.  dA = dBn = w (x_recon - x) / F; the |h|_1 part of d hidden is added by l1_add_kernel.
__global__ void __launch_bounds__(256)
snmf_cost_grad_kernel(const float* __restrict__ x, const float* __restrict__ A,
                      const float* __restrict__ Bn, const float* __restrict__ hidden,
                      int64_t ld_h, const float* __restrict__ w, float* __restrict__ dA,
                      float* __restrict__ dBn, float* __restrict__ part, int64_t rows, int F,
                      int Fp4, int N2, float l1_weight) {
    __shared__ float ssum[4], scnt[4];
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;
    float acc = 0.f, cnt = 0.f;
    if (row < rows) {
        const float wt = w[row];
        const float invF = 1.f / (float)F;
        for (int f = l; f < Fp4; f += 64) {
            float d = 0.f;
            if (f < F) {
                const size_t o = (size_t)row * F + f;
                const float err = A[o] + Bn[o] - x[o];
                acc += 0.5f * wt * err * err * invF;
                d = wt * err * invF;
            }
            dA[(size_t)row * Fp4 + f] = d;
            dBn[(size_t)row * Fp4 + f] = d;
        }
        float hs = 0.f;
        for (int n = l; n < N2; n += 64) hs += fabsf(hidden[(size_t)row * ld_h + n]);
        acc += wt * l1_weight * hs / (float)N2;
        cnt = (l == 0 && wt != 0.f) ? 1.f : 0.f;
    }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_xor(acc, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    if (l == 0) { ssum[wv] = acc; scnt[wv] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * (size_t)blockIdx.x + 0] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        part[2 * (size_t)blockIdx.x + 1] = (scnt[0] + scnt[1]) + (scnt[2] + scnt[3]);
    }
}

// d hidden[row][n] += w[row] * coef * sign(h)   (h >= 0 in this model; sign(0) = 0 as Theano's abs)
__global__ void __launch_bounds__(256)
l1_add_kernel(float* __restrict__ d_hidden, const float* __restrict__ hidden, int64_t ld_h,
              const float* __restrict__ w, int64_t rows, int N2, float coef) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * N2) return;
    const int64_t row = i / N2;
    const int n = (int)(i - row * N2);
    const float hv = hidden[(size_t)row * ld_h + n];
    const float sg = hv > 0.f ? 1.f : (hv < 0.f ? -1.f : 0.f);
    d_hidden[i] += w[row] * coef * sg;
}

__global__ void __launch_bounds__(256)
final_sums_kernel(const float* __restrict__ part, int64_t nblocks, float* __restrict__ sums) {
    __shared__ double s0[256], s1[256];
    double a = 0.0, b = 0.0;
    for (int64_t i = threadIdx.x; i < nblocks; i += 256) {
        a += (double)part[2 * i];
        b += (double)part[2 * i + 1];
    }
    s0[threadIdx.x] = a;
    s1[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            s0[threadIdx.x] += s0[threadIdx.x + o];
            s1[threadIdx.x] += s1[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { sums[0] = (float)s0[0]; sums[1] = (float)s1[0]; }
}

struct EpiStoreOff {
    float* C;
    int64_t ldc;
    int coloff;
    __device__ f32x2 pre(int64_t, int) const { return f32x2{0.f, 0.f}; }
    __device__ void operator()(int64_t row, int col, float v, f32x2) const {
        C[row * ldc + coloff + col] = v;
    }
};
struct EpiPart {
    float* P;
    int ld;
    size_t stride;
    __device__ float pre(int, int, int) const { return 0.f; }
    __device__ void operator()(int split, int m, int n, float acc, float) const {
        P[split * stride + (size_t)m * ld + n] = acc;
    }
};

// dK[n][f] = (sum of partials) * exp(K[n][f])
__global__ void __launch_bounds__(256)
dkernel_kernel(const float* __restrict__ P, const float* __restrict__ E, float* __restrict__ dK,
               int r, int F, int Fp4, int splits, size_t stride) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)r * F) return;
    const int n = (int)(i / F), f = (int)(i % F);
    const float g = ordered_sum<8>(P + (size_t)n * Fp4 + f, stride, splits);
    dK[i] = g * E[(size_t)n * Fp4 + f];
}

__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
            float* __restrict__ v, int64_t n, float lr_t, float b1, float b2, float eps,
            float gscale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
}

__global__ void __launch_bounds__(256)
sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out) {
    __shared__ double s[256];
    double a = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        a += (double)g[i] * (double)g[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)s[0];
}

// One optimiser step over the FLAT gradient / moment buffers (drnmf_adam_step_flat): block b updates
// blocks[b].count <= 1024 consecutive elements starting at flat offset blocks[b].flat_off, whose
// parameters live at blocks[b].param (the parameter tensors stay where their owner keeps them).
// Everything the step's scale depends on is read from DEVICE memory -- the all-reduced tail of the
// flat buffer [sum w*mse, #frames with w != 0, #frames, fault] and the 256 partial sums of g^2 -- so
// the host never waits for the gradients: no stream synchronisation, no copy to the host.
//   scale  = 1 / max(count, 1)                      (loss_norm 1, 'keras204': * frames / max(count, 1))
//   clip   : norm = sqrt(sum g^2) * scale > clipnorm  ->  scale *= clipnorm / norm
// A non-zero fault word (a persistent chain of this step timed out on SOME rank: the word is part of
// the all-reduced buffer) skips the update on every rank; report[1] tells the host.
// COUNTED form (step_in != NULL; drnmf_adam_step_flat_counted): the number of steps APPLIED so far lives on the
// device.  t = *step_in + 1, lr_t = lr / (1 + decay (t - 1)) * sqrt(1 - b2^t) / (1 - b1^t) is evaluated here in
// double (Keras' Adam.get_updates), and workgroup 0 leaves *step_out = t -- or *step_in again when the fault
// word skipped the update.  A skipped step is thereby not an iteration ON EVERY RANK AT THE SAME STEP (the
// fault word is all-reduced), whatever the host threads of the ranks have read by then (a host-side
// counter corrected when the fault report happens to be read let the replicas' bias corrections diverge).
__global__ void __launch_bounds__(256)
adam_flat_kernel(const drnmf_adam_block_t* __restrict__ blocks, const float* __restrict__ g,
                 float* __restrict__ m, float* __restrict__ v, const float* __restrict__ scalars,
                 const float* __restrict__ sumsq256, float lr_t, float b1, float b2, float eps,
                 float clipnorm, int loss_norm, float reg_loss, float* __restrict__ report,
                 const float* __restrict__ step_in, float* __restrict__ step_out, double lr_d, double decay_d,
                 double b1_d, double b2_d) {
    __shared__ double red[256];
    __shared__ float lr_sh;
    const float sse = scalars[0], cnt = fmaxf(scalars[1], 1.f), rows = scalars[2];
    const bool fault = scalars[3] != 0.f;
    if (step_in) {                                     // (block-uniform)
        if (threadIdx.x == 0) {
            const double t0 = (double)*step_in, t = t0 + 1.0;
            const double lr0 = decay_d > 0.0 ? lr_d * (1.0 / (1.0 + decay_d * t0)) : lr_d;
            lr_sh = (float)(lr0 * sqrt(1.0 - pow(b2_d, t)) / (1.0 - pow(b1_d, t)));
            if (blockIdx.x == 0) *step_out = fault ? (float)t0 : (float)t;
        }
        __syncthreads();
        lr_t = lr_sh;
    }
    float scale = 1.f / cnt;
    if (loss_norm == 1) scale *= rows / cnt;
    const float scale_loss = scale;
    if (clipnorm > 0.f) {                              // (block-uniform)
        red[threadIdx.x] = (double)sumsq256[threadIdx.x];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        const float norm = sqrtf((float)red[0]) * scale;
        if (norm > clipnorm) scale *= clipnorm / norm;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && report) {
        report[0] = sse * scale_loss + reg_loss;       // the normalised loss of the step
        report[1] = fault ? 1.f : 0.f;
        report[2] = scale;
        report[3] = cnt;
    }
    if (fault) return;
    const drnmf_adam_block_t bk = blocks[blockIdx.x];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = u * 256 + (int)threadIdx.x;
        if (e >= bk.count) break;
        const int64_t i = bk.flat_off + e;
        const float gi = g[i] * scale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        bk.param[e] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

struct LhWs {
    size_t off_E, off_dA, off_dB, off_part, off_P, total;
    int Fp4;
    int64_t nblocks;
};
LhWs lh_layout(int64_t rows, int F, int r) {
    LhWs L;
    L.Fp4 = pad4i(F);
    L.nblocks = (rows + 3) / 4;
    size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += round_up_sz(b, 256); return at; };
    L.off_E = take((size_t)2 * r * L.Fp4 * 4);
    L.off_dA = take((size_t)rows * L.Fp4 * 4);
    L.off_dB = take((size_t)rows * L.Fp4 * 4);
    L.off_part = take((size_t)L.nblocks * 2 * 4);
    L.off_P = take((size_t)gemm_tn::pick_splits(r, L.Fp4, rows, HB_SPLITS) * r * L.Fp4 * 4);
    L.total = o;
    return L;
}

// d hidden = [dA E_c^T | dBn E_n^T],  d kernel = (hidden^T dX) * E  for the two reconstruction layers
int32_t head_backward_gemms(drnmf_handle_t h, int64_t rows, int F, int r, const float* hidden,
                            int64_t ld_h, int h_off, const float* dA, const float* dB,
                            const float* E, float* P, int Fp4, float* d_hidden,
                            float* d_kernel_clean, float* d_kernel_noise, hipStream_t stream) {
    const int N2 = 2 * r;
    const int nsplit = gemm_tn::pick_splits(r, Fp4, rows, HB_SPLITS);
    for (int seg = 0; seg < 2; ++seg) {
        const float* dX = seg ? dB : dA;
        const float* Es = E + (size_t)seg * r * Fp4;
        // d hidden[:, seg*r + n] = sum_f dX[row][f] E[n][f]
        gemm::Operands g1{dX, Es, rows, r, Fp4, Fp4, Fp4};
        DRNMF_HIP(h, gemm::launch(g1, EpiStoreOff{d_hidden, N2, seg * r}, stream));
        // dE[n][f] = sum_rows hidden[row][seg*r + n] dX[row][f];  dK = dE * E
        gemm_tn::Operands t1{hidden + h_off + (size_t)seg * r, dX, rows, r, Fp4, ld_h, Fp4};
        const size_t pstr = (size_t)r * Fp4;
        DRNMF_HIP(h, gemm_tn::launch(t1, EpiPart{P, Fp4, pstr}, nsplit, stream));
        const size_t tot = (size_t)r * F;
        hipLaunchKernelGGL(dkernel_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream,
                           P, Es, seg ? d_kernel_noise : d_kernel_clean, r, F, Fp4, nsplit, pstr);
    }
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

}  // namespace

extern "C" size_t drnmf_loss_head_workspace_bytes(int64_t rows, int32_t F, int32_t r) {
    if (rows <= 0 || F <= 0 || r <= 0) return 0;
    return lh_layout(rows, F, r).total;
}

extern "C" int32_t drnmf_loss_head_backward(drnmf_handle_t h, int64_t rows, int32_t F, int32_t r,
                                            const float* x_raw, const float* hidden, int64_t ld_h,
                                            int32_t h_off, const float* kernel_clean,
                                            const float* kernel_noise, int32_t square,
                                            const float* mask, const float* A, const float* Bn,
                                            const float* y, const float* w, float* sums,
                                            float* d_hidden, float* d_kernel_clean,
                                            float* d_kernel_noise, void* workspace,
                                            size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (rows <= 0 || F <= 0 || r <= 0 || h_off < 0 || ld_h < h_off + 2 * (int64_t)r)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "loss_head_backward: bad shape");
    if (!x_raw || !hidden || !kernel_clean || !kernel_noise || !mask || !A || !Bn || !y || !w ||
        !sums || !d_hidden || !d_kernel_clean || !d_kernel_noise || !workspace)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "loss_head_backward: NULL pointer argument");
    const LhWs L = lh_layout(rows, F, r);
    if (workspace_bytes < L.total)
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "loss_head_backward: workspace %zu < required %zu",
                   workspace_bytes, L.total);
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    float* E = (float*)(ws + L.off_E);
    float* dA = (float*)(ws + L.off_dA);
    float* dB = (float*)(ws + L.off_dB);
    float* part = (float*)(ws + L.off_part);
    float* P = (float*)(ws + L.off_P);
    const int Fp4 = L.Fp4;
    {
        const size_t tot = (size_t)2 * r * Fp4;
        hipLaunchKernelGGL(exp_pad_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                           stream, kernel_clean, kernel_noise, E, r, F, Fp4);
    }
    hipLaunchKernelGGL(loss_grad_kernel, dim3((unsigned)L.nblocks), dim3(256), 0, stream, x_raw,
                       mask, A, Bn, y, w, dA, dB, part, rows, F, Fp4, square);
    hipLaunchKernelGGL(final_sums_kernel, dim3(1), dim3(256), 0, stream, part, L.nblocks, sums);
    DRNMF_HIP(h, hipGetLastError());
    return head_backward_gemms(h, rows, F, r, hidden, ld_h, h_off, dA, dB, E, P, Fp4, d_hidden,
                               d_kernel_clean, d_kernel_noise, stream);
}

extern "C" int32_t drnmf_snmf_cost_head_backward(drnmf_handle_t h, int64_t rows, int32_t F,
                                                 int32_t r, const float* x_raw,
                                                 const float* hidden, int64_t ld_h, int32_t h_off,
                                                 const float* kernel_clean,
                                                 const float* kernel_noise, const float* A,
                                                 const float* Bn, const float* w, float l1_weight,
                                                 float* sums, float* d_hidden,
                                                 float* d_kernel_clean, float* d_kernel_noise,
                                                 void* workspace, size_t workspace_bytes,
                                                 void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (rows <= 0 || F <= 0 || r <= 0 || h_off < 0 || ld_h < h_off + 2 * (int64_t)r)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "snmf_cost_head_backward: bad shape");
    if (!x_raw || !hidden || !kernel_clean || !kernel_noise || !A || !Bn || !w || !sums ||
        !d_hidden || !d_kernel_clean || !d_kernel_noise || !workspace)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "snmf_cost_head_backward: NULL pointer argument");
    const LhWs L = lh_layout(rows, F, r);
    if (workspace_bytes < L.total)
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "snmf_cost_head_backward: workspace %zu < required %zu",
                   workspace_bytes, L.total);
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    float* E = (float*)(ws + L.off_E);
    float* dA = (float*)(ws + L.off_dA);
    float* dB = (float*)(ws + L.off_dB);
    float* part = (float*)(ws + L.off_part);
    float* P = (float*)(ws + L.off_P);
    const int Fp4 = L.Fp4, N2 = 2 * r;
    {
        const size_t tot = (size_t)2 * r * Fp4;
        hipLaunchKernelGGL(exp_pad_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                           stream, kernel_clean, kernel_noise, E, r, F, Fp4);
    }
    hipLaunchKernelGGL(snmf_cost_grad_kernel, dim3((unsigned)L.nblocks), dim3(256), 0, stream,
                       x_raw, A, Bn, hidden + h_off, ld_h, w, dA, dB, part, rows, F, Fp4, N2,
                       l1_weight);
    hipLaunchKernelGGL(final_sums_kernel, dim3(1), dim3(256), 0, stream, part, L.nblocks, sums);
    DRNMF_HIP(h, hipGetLastError());
    const int32_t rc = head_backward_gemms(h, rows, F, r, hidden, ld_h, h_off, dA, dB, E, P, Fp4,
                                           d_hidden, d_kernel_clean, d_kernel_noise, stream);
    if (rc != DRNMF_OK) return rc;
    const int64_t tot = rows * N2;
    hipLaunchKernelGGL(l1_add_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream,
                       d_hidden, hidden + h_off, ld_h, w, rows, N2, l1_weight / (float)N2);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

extern "C" int32_t drnmf_adam_step(drnmf_handle_t h, int64_t n, float* param, const float* grad,
                                   float* m, float* v, float lr_t, float beta1, float beta2,
                                   float eps, float grad_scale, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n <= 0 || !param || !grad || !m || !v)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "adam_step: bad argument");
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream_, param, grad, m, v, n, lr_t, beta1, beta2, eps,
                       grad_scale);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

extern "C" int32_t drnmf_adam_step_flat(drnmf_handle_t h, int64_t n_blocks,
                                        const drnmf_adam_block_t* blocks, const float* flat_grad,
                                        float* flat_m, float* flat_v, const float* scalars4,
                                        const float* sumsq256, float lr_t, float beta1, float beta2,
                                        float eps, float clipnorm, int32_t loss_norm, float reg_loss,
                                        float* report4, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n_blocks <= 0 || n_blocks > 0x7fffffff || !blocks || !flat_grad || !flat_m || !flat_v || !scalars4)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "adam_step_flat: bad argument");
    if (clipnorm > 0.f && !sumsq256)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "adam_step_flat: clipnorm > 0 needs the drnmf_sumsq partials");
    if (loss_norm != 0 && loss_norm != 1)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "adam_step_flat: loss_norm must be 0 (masked mean) or 1 (keras204)");
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream_,
                       blocks, flat_grad, flat_m, flat_v, scalars4, sumsq256, lr_t, beta1, beta2, eps,
                       clipnorm, (int)loss_norm, reg_loss, report4, (const float*)nullptr, (float*)nullptr, 0.0,
                       0.0, 0.0, 0.0);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

extern "C" int32_t drnmf_adam_step_flat_counted(drnmf_handle_t h, int64_t n_blocks,
                                                const drnmf_adam_block_t* blocks, const float* flat_grad,
                                                float* flat_m, float* flat_v, const float* scalars4,
                                                const float* sumsq256, double lr, double decay, double beta1,
                                                double beta2, float eps, float clipnorm, int32_t loss_norm,
                                                float reg_loss, const float* step_in, float* step_out,
                                                float* report4, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n_blocks <= 0 || n_blocks > 0x7fffffff || !blocks || !flat_grad || !flat_m || !flat_v || !scalars4 ||
        !step_in || !step_out || step_in == step_out)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "adam_step_flat_counted: bad argument");
    if (!(lr >= 0.0) || !(decay >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "adam_step_flat_counted: lr >= 0, decay >= 0, 0 <= beta < 1 required");
    if (clipnorm > 0.f && !sumsq256)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "adam_step_flat_counted: clipnorm > 0 needs the drnmf_sumsq partials");
    if (loss_norm != 0 && loss_norm != 1)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "adam_step_flat_counted: loss_norm must be 0 (masked mean) or 1 (keras204)");
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream_,
                       blocks, flat_grad, flat_m, flat_v, scalars4, sumsq256, 0.f, (float)beta1, (float)beta2, eps,
                       clipnorm, (int)loss_norm, reg_loss, report4, step_in, step_out, lr, decay, beta1, beta2);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

extern "C" int32_t drnmf_sumsq(drnmf_handle_t h, int64_t n, const float* g, float* out256,
                               void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n <= 0 || !g || !out256) DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "sumsq: bad argument");
    hipLaunchKernelGGL(sumsq_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream_, g, n, out256);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}
