// Measurement probe (not part of libdrnmf): how fast can ONE short kernel launch stream S bytes
// from memory on this chip?  The config-5 cell launches (F = 1025, N = 8000, fp16) each read a
// fresh 16.4 MB dictionary slice; this is the floor for that: the same launch geometry (256-1024
// workgroups, 16-byte loads, everything issued up front), no compute, launches back to back from a
// hipGraph, reading either the same buffer every launch (cache-resident after the first) or a new
// region of a pool larger than the Infinity Cache.
//   stream_probe <MB per launch> <pool MB> <workgroups> <threads> [launches]
// Build: hipcc --offload-arch=gfx950:xnack- -O3 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4;

// wbytes > 0: every launch also dirties wbytes of a scratch buffer with plain stores (does the L2
// still hold the clean read-only region across the kernel boundary?)
template <int U>
__global__ void __launch_bounds__(1024) stream_kernel(const f32x4* __restrict__ src, size_t n16, float* sink,
                                                      f32x4* wbuf, size_t w16) {
    {
        const size_t tid0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (tid0 < w16) wbuf[tid0] = f32x4{1.f, 2.f, 3.f, (float)tid0};
    }
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthr = (size_t)gridDim.x * blockDim.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = tid; i < n16; i += nthr * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + (size_t)u * nthr;
            v[u] = src[j < n16 ? j : i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = acc[0];   // keep the loads
}

int main(int argc, char** argv) {
    const double mb = argc > 1 ? atof(argv[1]) : 16.4;
    const double pool_mb = argc > 2 ? atof(argv[2]) : 1640.0;
    const int wgs = argc > 3 ? atoi(argv[3]) : 256, thr = argc > 4 ? atoi(argv[4]) : 512;
    const int launches = argc > 5 ? atoi(argv[5]) : 400;
    const double wkb = argc > 6 ? atof(argv[6]) : 0.0;
    size_t w16 = (size_t)(wkb * 1024 / 16);
    f32x4* wbuf;
    CK(hipMalloc(&wbuf, (w16 + 1) * 16));
    const size_t n16 = (size_t)(mb * 1e6 / 16), pool16 = (size_t)(pool_mb * 1e6 / 16);
    const size_t regions = pool16 / n16 ? pool16 / n16 : 1;
    f32x4* buf;
    float* sink;
    CK(hipMalloc(&buf, regions * n16 * 16));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, regions * n16 * 16));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    hipGraphNode_t last = nullptr;
    std::vector<const f32x4*> ptrs(launches);
    size_t n = n16;
    for (int i = 0; i < launches; ++i) {
        ptrs[i] = buf + (size_t)(i % regions) * n16;
        void* kp[5] = {&ptrs[i], &n, &sink, &wbuf, &w16};
        hipKernelNodeParams np;
        memset(&np, 0, sizeof(np));
        np.func = (void*)&stream_kernel<8>;
        np.gridDim = dim3(wgs);
        np.blockDim = dim3(thr);
        np.kernelParams = kp;
        hipGraphNode_t node;
        CK(hipGraphAddKernelNode(&node, g, last ? &last : nullptr, last ? 1 : 0, &np));
        last = node;
    }
    hipGraphExec_t ex;
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ex, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double us = best * 1e3 / launches;
    printf("[writes %.0f KB/launch] %.1f MB per launch, pool %.0f MB (%zu regions), %d x %d threads: %.2f us per launch = %.2f TB/s (%.2f TB/s net of a 1.53 us boundary)\n",
           wkb, mb, pool_mb, regions, wgs, thr, us, mb * 1e6 / us / 1e6, mb * 1e6 / (us - 1.53) / 1e6);
    return 0;
}
