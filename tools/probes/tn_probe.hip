// Measurement probe (not part of libdrnmf): the time-batched weight-gradient product
// C[M x N] = A^T B (csrc/gemm_tn.h) at the C2 shape -- contraction over 128 000 frames,
// M = 528 bins, N = 1000 atoms -- with the B operand's row stride either N (layer-major buffers)
// or K*N = 25 000 floats (the [frames][K*N] layout of the all-hidden output).
//   tn_probe <rows> <M> <N> <ldb> <splits> [reps] [zero]
// Build: hipcc --offload-arch=gfx950:xnack- -O3 -std=c++17 -I../../dr-nmf_amd/csrc -I../../include -o tn_probe tn_probe.hip
#include "gemm_tn.h"
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct EpiPart {
    float* P; int ld; size_t stride;
    __device__ float pre(int, int, int) const { return 0.f; }
    __device__ void operator()(int split, int m, int n, float acc, float) const {
        P[split * stride + (size_t)m * ld + n] = acc;
    }
};

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {   // zeros would flatter the clocks
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(h & 0xffffff) * (1.f / 16777216.f);
    }
}

int main(int argc, char** argv) {
    const int64_t rows = atoll(argv[1]);
    const int M = atoi(argv[2]), N = atoi(argv[3]);
    const int64_t ldb = atoll(argv[4]);
    const int splits = atoi(argv[5]);
    const int reps = argc > 6 ? atoi(argv[6]) : 10;
    float *A, *B, *P;
    CK(hipMalloc(&A, rows * M * 4));
    CK(hipMalloc(&B, rows * ldb * 4));
    CK(hipMalloc(&P, (size_t)splits * M * N * 4));
    if (argc > 7) {   // 8th argument: zero operands instead of random ones
        CK(hipMemset(A, 0, rows * M * 4));
        CK(hipMemset(B, 0, rows * ldb * 4));
    } else {
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, A, (size_t)rows * M, 1u);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, B, (size_t)rows * ldb, 2u);
        CK(hipDeviceSynchronize());
    }
    gemm_tn::Operands g{A, B, rows, M, N, M, ldb};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) CK(gemm_tn::launch(g, EpiPart{P, N, (size_t)M * N}, splits, s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) {
        gemm_tn::Operands gi = g;
        gi.B = B + (ldb > N ? (size_t)(i % (ldb / N)) * N : 0);   // another layer's columns each time
        CK(gemm_tn::launch(gi, EpiPart{P, N, (size_t)M * N}, splits, s));
    }
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("rows %lld M %d N %d ldb %lld splits %d: %.3f ms, %.1f TFLOP/s\n", (long long)rows, M, N,
           (long long)ldb, splits, ms, 2.0 * rows * M * N / ms * 1e-9);
    return 0;
}
