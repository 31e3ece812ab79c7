// SDR of bss_eval_sources(xest', xref') with one source (score_audio.m:206) on gfx950.
//
// BSS Eval 3.0 is a third-party toolbox that the reference downloads at setup
// (download_toolboxes.sh) and that is not in its tree: parity is pinned only to the published
// definition (oracle/drnmf_oracle.py:sdr_db).  The estimate, zero-padded by flen-1 samples, is
// projected onto the span of the reference delayed by 0..flen-1 samples:
//     r[a] = sum_n ref[n] ref[n-a],  d[a] = sum_n est[n] ref[n-a]        (drnmf_sdr_corr)
//     C = Toeplitz(r)^-1 d                                               (host, fp64, 512 x 512)
//     s[n] = sum_a C[a] ref[n-a];  SDR = 10 log10(sum s^2 / sum (est - s)^2)   (drnmf_sdr_project)
// Everything accumulates in fp64 (the normal equations of a speech signal are badly conditioned);
// sums are combined in a fixed order, so results are run-to-run identical.  Trailing zero padding
// of est/ref (ragged batches) changes nothing.
#include "common.h"

namespace {

constexpr int CORR_SPLITS = 64;
constexpr int MAX_FLEN = 2048;

// partial[sig][split][2][flen]: thread = lag a, loops over its time segment
__global__ void __launch_bounds__(256)
sdr_corr_kernel(const float* __restrict__ est, const float* __restrict__ ref, int64_t nsampl,
                int flen, double* __restrict__ part) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    const int sp = blockIdx.y, sig = blockIdx.z;
    const float* e = est + (size_t)sig * nsampl;
    const float* s = ref + (size_t)sig * nsampl;
    const int64_t per = (nsampl + CORR_SPLITS - 1) / CORR_SPLITS;
    const int64_t n0 = sp * per;
    int64_t n1 = n0 + per;
    if (n1 > nsampl) n1 = nsampl;
    double ar = 0.0, ad = 0.0;
    if (a < flen) {
        for (int64_t n = n0; n < n1; ++n) {
            const int64_t m = n - a;
            const double sd = m >= 0 ? (double)s[m] : 0.0;
            ar = fma((double)s[n], sd, ar);
            ad = fma((double)e[n], sd, ad);
        }
        double* p = part + (((size_t)sig * CORR_SPLITS + sp) * 2) * flen;
        p[a] = ar;
        p[flen + a] = ad;
    }
}

__global__ void __launch_bounds__(256)
sdr_corr_reduce_kernel(const double* __restrict__ part, int flen, double* __restrict__ r_out,
                       double* __restrict__ d_out) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    const int sig = blockIdx.y;
    if (a >= flen) return;
    double r = 0.0, d = 0.0;
    for (int sp = 0; sp < CORR_SPLITS; ++sp) {
        const double* p = part + (((size_t)sig * CORR_SPLITS + sp) * 2) * flen;
        r += p[a];
        d += p[flen + a];
    }
    r_out[(size_t)sig * flen + a] = r;
    d_out[(size_t)sig * flen + a] = d;
}

// thread = output sample n of the padded length L = nsampl + flen - 1; partial[sig][block][2]
__global__ void __launch_bounds__(256)
sdr_project_kernel(const float* __restrict__ est, const float* __restrict__ ref,
                   const double* __restrict__ coef, int64_t nsampl, int flen,
                   double* __restrict__ part) {
    extern __shared__ double sm[];          // coef[flen] | red[2][256]
    double* cf = sm;
    double* red = sm + flen;
    const int sig = blockIdx.y;
    for (int i = threadIdx.x; i < flen; i += 256) cf[i] = coef[(size_t)sig * flen + i];
    __syncthreads();
    const float* e = est + (size_t)sig * nsampl;
    const float* s = ref + (size_t)sig * nsampl;
    const int64_t L = nsampl + flen - 1;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double st = 0.0, en = 0.0, er = 0.0;
    if (n < L) {
        int a0 = n - (nsampl - 1) > 0 ? (int)(n - (nsampl - 1)) : 0;
        int a1 = n < flen - 1 ? (int)n : flen - 1;
        for (int a = a0; a <= a1; ++a) st = fma(cf[a], (double)s[n - a], st);
        const double ev = n < nsampl ? (double)e[n] : 0.0;
        en = st * st;
        er = (ev - st) * (ev - st);
    }
    red[threadIdx.x] = en;
    red[256 + threadIdx.x] = er;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[threadIdx.x] += red[threadIdx.x + o];
            red[256 + threadIdx.x] += red[256 + threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* p = part + ((size_t)sig * gridDim.x + blockIdx.x) * 2;
        p[0] = red[0];
        p[1] = red[256];
    }
}

__global__ void __launch_bounds__(256)
sdr_final_kernel(const double* __restrict__ part, int nblocks, double* __restrict__ energies,
                 float* __restrict__ out_db) {
    __shared__ double s0[256], s1[256];
    const int sig = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        a += part[((size_t)sig * nblocks + i) * 2 + 0];
        b += part[((size_t)sig * nblocks + i) * 2 + 1];
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
    if (threadIdx.x == 0) {
        energies[2 * sig + 0] = s0[0];
        energies[2 * sig + 1] = s1[0];
        out_db[sig] = (float)(10.0 * log10(s0[0] / s1[0]));
    }
}

size_t corr_bytes(int n_sig, int flen) { return (size_t)n_sig * CORR_SPLITS * 2 * flen * 8; }
int64_t proj_blocks(int64_t nsampl, int flen) { return (nsampl + flen - 1 + 255) / 256; }
size_t proj_bytes(int n_sig, int64_t nsampl, int flen) {
    return (size_t)n_sig * proj_blocks(nsampl, flen) * 2 * 8;
}

}  // namespace

extern "C" size_t drnmf_sdr_workspace_bytes(int32_t n_sig, int64_t nsampl, int32_t flen) {
    if (n_sig <= 0 || nsampl <= 0 || flen <= 0) return 0;
    const size_t a = corr_bytes(n_sig, flen), b = proj_bytes(n_sig, nsampl, flen);
    return round_up_sz(a > b ? a : b, 256);
}

extern "C" int32_t drnmf_sdr_corr(drnmf_handle_t h, int32_t n_sig, int64_t nsampl, int32_t flen,
                                  const float* est, const float* ref, double* r_out,
                                  double* d_out, void* workspace, size_t workspace_bytes,
                                  void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n_sig <= 0 || nsampl <= 0 || flen <= 0 || flen > MAX_FLEN || n_sig > 65535 || !est ||
        !ref || !r_out || !d_out || !workspace)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "sdr_corr: bad argument (flen <= 2048)");
    if (workspace_bytes < drnmf_sdr_workspace_bytes(n_sig, nsampl, flen))
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "sdr_corr: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    double* part = (double*)workspace;
    const unsigned lb = (unsigned)((flen + 255) / 256);
    hipLaunchKernelGGL(sdr_corr_kernel, dim3(lb, CORR_SPLITS, (unsigned)n_sig), dim3(256), 0,
                       stream, est, ref, nsampl, flen, part);
    hipLaunchKernelGGL(sdr_corr_reduce_kernel, dim3(lb, (unsigned)n_sig), dim3(256), 0, stream,
                       part, flen, r_out, d_out);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

extern "C" int32_t drnmf_sdr_project(drnmf_handle_t h, int32_t n_sig, int64_t nsampl,
                                     int32_t flen, const float* est, const float* ref,
                                     const double* coef, double* energies, float* out_db,
                                     void* workspace, size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (n_sig <= 0 || nsampl <= 0 || flen <= 0 || flen > MAX_FLEN || n_sig > 65535 || !est ||
        !ref || !coef || !energies || !out_db || !workspace)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "sdr_project: bad argument (flen <= 2048)");
    if (workspace_bytes < drnmf_sdr_workspace_bytes(n_sig, nsampl, flen))
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "sdr_project: workspace too small");
    hipStream_t stream = (hipStream_t)stream_;
    double* part = (double*)workspace;
    const int64_t nb = proj_blocks(nsampl, flen);
    const size_t shmem = ((size_t)flen + 512) * 8;
    hipLaunchKernelGGL(sdr_project_kernel, dim3((unsigned)nb, (unsigned)n_sig), dim3(256), shmem,
                       stream, est, ref, coef, nsampl, flen, part);
    hipLaunchKernelGGL(sdr_final_kernel, dim3((unsigned)n_sig), dim3(256), 0, stream, part,
                       (int)nb, energies, out_db);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}
