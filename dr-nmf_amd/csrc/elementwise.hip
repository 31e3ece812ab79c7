// Small elementwise / reduction kernels around the hot path, so that no arithmetic of the product
// runs through torch ops:
//   drnmf_divide_a_by_aplusb  DivideAbyAplusB as a stand-alone layer (custom_layers.py:33-56; inside
//                             the model it is fused into the head kernel)
//   drnmf_add                 x_recon = clean_est + noise_est of model_pretrain (enhance.py:1024-1026)
//   drnmf_loss_forward        validation loss of fit()/evaluate (enhance.py:1152-1157) without gradients
//   drnmf_wav_int16           util.wavwrite's float32 -> int16 conversion (util.py:37-45)
#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
ratio_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out,
             int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = expf(logf(1e-7f + A[i]) - logf(1e-7f + A[i] + B[i]));
}

__global__ void __launch_bounds__(256)
add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
           int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// one wave per row; mode 0: w * mean_F (x*p - y)^2;  mode 1: w * (0.5 mean_F (p + p2 - y)^2 +
// l1_weight * mean_N |hidden|).  part[block] = {sum, #rows with w != 0} of the block's 4 rows.
__global__ void __launch_bounds__(256)
loss_rows_kernel(const float* __restrict__ x, const float* __restrict__ p,
                 const float* __restrict__ p2, const float* __restrict__ y,
                 const float* __restrict__ w, const float* __restrict__ hidden, int64_t ld_h,
                 int N2, float l1_weight, float* __restrict__ part, int64_t rows, int F, int mode) {
    __shared__ float ssum[4], scnt[4];
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wv;
    float acc = 0.f, cnt = 0.f;
    if (row < rows) {
        const float wt = w[row];
        const float invF = 1.f / (float)F;
        for (int f = l; f < F; f += 64) {
            const size_t o = (size_t)row * F + f;
            const float err = mode == 0 ? x[o] * p[o] - y[o] : p[o] + p2[o] - y[o];
            acc += (mode == 0 ? 1.f : 0.5f) * wt * err * err * invF;
        }
        if (mode == 1) {
            float hs = 0.f;
            for (int n = l; n < N2; n += 64) hs += fabsf(hidden[(size_t)row * ld_h + n]);
            acc += wt * l1_weight * hs / (float)N2;
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

// fixed-order fp64 sum of the block partials (deterministic)
__global__ void __launch_bounds__(256)
loss_final_kernel(const float* __restrict__ part, int64_t nblocks, float* __restrict__ sums) {
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

constexpr int AM_BLOCKS = 1024;

__global__ void __launch_bounds__(256)
absmax_part_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
    __shared__ float sm[4];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        m = fmaxf(m, fabsf(x[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

__global__ void __launch_bounds__(256)
absmax_final_kernel(float* __restrict__ part, int nparts) {     // part[nparts] <- max
    __shared__ float sm[256];
    float m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, part[i]);
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[nparts] = sm[0];
}

// numpy: x = x / max|x| if max|x| > 1;  np.int16(x * 32767.0) truncates toward zero
__global__ void __launch_bounds__(256)
to_int16_kernel(const float* __restrict__ x, const float* __restrict__ amax,
                int16_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float m = *amax;
    float v = x[i];
    if (m > 1.f) v = v / m;
    out[i] = (int16_t)(int)(v * 32767.0f);
}

}  // namespace

extern "C" int32_t drnmf_divide_a_by_aplusb(drnmf_handle_t h, int64_t n, const float* A,
                                            const float* B, float* out, void* stream) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n < 0 || (n > 0 && (!A || !B || !out)))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "divide_a_by_aplusb: bad argument");
    if (n == 0) return DRNMF_OK;
    hipLaunchKernelGGL(ratio_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, A, B, out, n);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

extern "C" int32_t drnmf_add(drnmf_handle_t h, int64_t n, const float* a, const float* b,
                             float* out, void* stream) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n < 0 || (n > 0 && (!a || !b || !out))) DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "add: bad argument");
    if (n == 0) return DRNMF_OK;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a, b, out, n);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

extern "C" size_t drnmf_loss_forward_workspace_bytes(int64_t rows) {
    if (rows <= 0) return 0;
    return round_up_sz((size_t)((rows + 3) / 4) * 2 * sizeof(float), 256);
}

extern "C" int32_t drnmf_loss_forward(drnmf_handle_t h, int64_t rows, int32_t F, int32_t mode,
                                      const float* x_raw, const float* pred, const float* pred2,
                                      const float* y, const float* w, const float* hidden,
                                      int64_t ld_h, int32_t N2, float l1_weight, float* sums,
                                      void* workspace, size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (rows <= 0 || F <= 0 || (mode != 0 && mode != 1))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "loss_forward: rows, F must be positive, mode 0 or 1");
    if (!pred || !y || !w || !sums || !workspace || (mode == 0 && !x_raw) ||
        (mode == 1 && (!pred2 || !hidden || N2 <= 0 || ld_h < N2)))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "loss_forward: NULL pointer / bad hidden layout");
    if (workspace_bytes < drnmf_loss_forward_workspace_bytes(rows))
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "loss_forward: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t nblocks = (rows + 3) / 4;
    float* part = (float*)workspace;
    hipLaunchKernelGGL(loss_rows_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream, x_raw, pred,
                       pred2, y, w, hidden, ld_h, N2, l1_weight, part, rows, F, mode);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, stream, part, nblocks, sums);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

extern "C" size_t drnmf_wav_int16_workspace_bytes(void) { return (AM_BLOCKS + 1) * sizeof(float); }

extern "C" int32_t drnmf_wav_int16(drnmf_handle_t h, int64_t n, const float* x, int16_t* out,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n < 0 || (n > 0 && (!x || !out || !workspace)))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "wav_int16: bad argument");
    if (n == 0) return DRNMF_OK;
    if (workspace_bytes < drnmf_wav_int16_workspace_bytes())
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "wav_int16: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    float* part = (float*)workspace;
    int nb = (int)((n + 255) / 256);
    if (nb > AM_BLOCKS) nb = AM_BLOCKS;
    hipLaunchKernelGGL(absmax_part_kernel, dim3(nb), dim3(256), 0, stream, x, n, part);
    hipLaunchKernelGGL(absmax_final_kernel, dim3(1), dim3(256), 0, stream, part, nb);
    hipLaunchKernelGGL(to_int16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x,
                       part + nb, out, n);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}
