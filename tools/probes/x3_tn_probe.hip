// Standalone harness of the split-operand TN product (csrc/gemm_tn_x3.h) at the shapes of the headline's
// time-batched weight gradients (frames x 516 bins, frames x 2000 atoms) and of the dictionary training.
//   hipcc --offload-arch=gfx950:xnack- -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 \
//         -mllvm -pragma-unroll-threshold=262144 -o x3_tn_probe x3_tn_probe.hip && ./x3_tn_probe
#include "../../dr-nmf_amd/csrc/gemm_tn.h"

#include <cmath>
#include <random>
#include <vector>

thread_local int tl_matrix_mode = 0;
thread_local drnmf_handle_t tl_handle = nullptr;
const char* tune_env(const char*) { return nullptr; }
void* x3_scratch_get(hipStream_t, size_t) { return nullptr; }

struct EpiPart {
    float* P; int ld; size_t pstr;
    __device__ float pre(int, int, int) const { return 0.f; }
    __device__ void operator()(int s, int m, int n, float acc, float) const { P[s * pstr + (size_t)m * ld + n] = acc; }
};
__global__ void fold_kernel(const float* P, float* C, size_t n, size_t pstr, int splits) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    for (int s = 0; s < splits; ++s) a += P[s * pstr + i];
    C[i] = a;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int split_div = argc > 1 ? atoi(argv[1]) : 0;      // > 0: splits = pick_splits / split_div (x3 arm only)
    struct Shape { int64_t Kdim; int M, N; const char* name; };
    const Shape shapes[] = {{128000, 516, 2016, "headline weight gradient: 128000 frames, 516 x 2016"},
                            {32768, 516, 1000, "dictionary training statistics: 32768 frames, 516 x 1000"},
                            {16000, 260, 2016, "shipped r = 1000 training step: 16000 frames, 260 x 2016"}};
    std::mt19937 rng(11);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 5;
    for (const Shape& s : shapes) {
        std::vector<float> hA((size_t)s.Kdim * s.M), hB((size_t)s.Kdim * s.N);
        for (auto& v : hA) { const float u = U(rng); v = (u - 0.5f) * u * 2.f; }
        for (auto& v : hB) { const float u = U(rng); v = u < 0.6f ? 0.f : (u - 0.8f) * 3.f; }
        float *A, *B, *P, *C0, *C1;
        const int splits = gemm_tn::pick_splits(s.M, s.N, s.Kdim, 64);
        const size_t pstr = (size_t)s.M * s.N;
        CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4));
        CK(hipMalloc(&P, pstr * splits * 4 * 4)); CK(hipMalloc(&C0, pstr * 4)); CK(hipMalloc(&C1, pstr * 4));
        CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        gemm_tn::Operands g{A, B, s.Kdim, s.M, s.N, s.M, s.N};
        double tf[2];
        for (int mode = 0; mode < 2; ++mode) {
            tl_matrix_mode = mode;
            float* C = mode ? C1 : C0;
            const int sp = (mode && split_div > 0) ? (splits / split_div > 0 ? splits / split_div : 1) : (mode && split_div < 0 ? splits * -split_div : splits);
            auto run = [&]() { return gemm_tn::launch(g, EpiPart{P, s.N, pstr}, sp, 0); };
            CK(run());
            hipLaunchKernelGGL(fold_kernel, dim3((unsigned)((pstr + 255) / 256)), dim3(256), 0, 0, P, C, pstr, pstr, sp);
            std::vector<float> hC(pstr);
            CK(hipMemcpy(hC.data(), C, pstr * 4, hipMemcpyDeviceToHost));
            double mx = 0, ss = 0, ref_mx = 0; int cnt = 0;
            for (int t = 0; t < 300; ++t) {
                const int m = (int)(U(rng) * s.M) % s.M, n = (int)(U(rng) * s.N) % s.N;
                double acc = 0;
                for (int64_t k = 0; k < s.Kdim; ++k) acc += (double)hA[k * s.M + m] * (double)hB[k * s.N + n];
                const double d = (double)hC[(size_t)m * s.N + n] - acc;
                mx = std::fmax(mx, std::fabs(d)); ss += d * d; ref_mx = std::fmax(ref_mx, std::fabs(acc)); ++cnt;
            }
            CK(run());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) CK(run());
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            tf[mode] = 2.0 * s.Kdim * s.M * s.N * reps / (ms * 1e-3) / 1e12;
            printf("%-64s %s (%2d splits): %8.1f us, %6.1f TFLOP/s-eq; vs fp64 max %.2e rms %.2e (of max |ref| %.3g)\n", s.name,
                   mode ? "bf16x3" : "f32   ", sp, ms * 1e3 / reps, tf[mode], mx / ref_mx, std::sqrt(ss / cnt) / ref_mx, ref_mx);
        }
        printf("   -> %.2fx\n", tf[1] / tf[0]);
        (void)hipFree(A); (void)hipFree(B); (void)hipFree(P); (void)hipFree(C0); (void)hipFree(C1);
    }
    return 0;
}
