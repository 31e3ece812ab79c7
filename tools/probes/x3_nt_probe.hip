// Standalone harness of the split-operand NT product (csrc/gemm_nt_x3.h) at the two shapes of a frame-parallel
// ISTA iteration: compiles in seconds, so kernel variants are A/B'd here before the library is rebuilt.
//   hipcc --offload-arch=gfx950:xnack- -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 \
//         -mllvm -pragma-unroll-threshold=262144 [-DX3_TIMELINE] -o x3_nt_probe x3_nt_probe.hip && ./x3_nt_probe
// Prints TFLOP/s (fp32-equivalent: 2 M N K per product) of the fp32 kernel and the split-operand kernel, the
// max / rms difference of their outputs against an fp64 reference on sampled entries, and with -DX3_TIMELINE the
// mean s_memtime phase lengths of wave 0 of the workgroups (k-tile loop: MFMA phase, barrier, stores, barrier).
#include "../../dr-nmf_amd/csrc/gemm_nt.h"

#include <cmath>
#include <random>
#include <vector>

thread_local int tl_matrix_mode = 0;
thread_local drnmf_handle_t tl_handle = nullptr;
const char* tune_env(const char*) { return nullptr; }
static void* g_scratch = nullptr;
static size_t g_scratch_bytes = 0;
void* x3_scratch_get(hipStream_t, size_t bytes) {
    if (bytes > g_scratch_bytes) {
        if (g_scratch) (void)hipFree(g_scratch);
        if (hipMalloc(&g_scratch, bytes) != hipSuccess) return nullptr;
        g_scratch_bytes = bytes;
    }
    return g_scratch;
}

struct EpiStore {
    float* C; int ld;
    __device__ f32x2 pre(int, int) const { return f32x2{0.f, 0.f}; }
    __device__ void operator()(int r, int c, float acc, f32x2) const { C[(int64_t)r * ld + c] = acc; }
};
struct EpiUpd {    // the shape of EpiIstaUpdate: read-modify-write of the output
    float* H; int ld; float c0, c1;
    __device__ f32x2 pre(int r, int c) const { return f32x2{H[(int64_t)r * ld + c], 0.f}; }
    __device__ void operator()(int r, int c, float acc, f32x2 p) const {
        const float v = p[0] + c0 + c1 * acc;
        H[(int64_t)r * ld + c] = v > 0.f ? v : 0.f;
    }
};

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 32768;
    const bool const_a = argc > 2;        // every A value 0.1f: the worst case for correlated rounding
    const bool bf16_exact = argc > 3;     // ... A = 13/128 and B rounded to bf16: only the (hi, hi) product is non-zero,
                                          //     what remains against fp64 is the bf16 MFMA's own accumulation
    const int reps = 10;
    struct Shape { int N, K, ktail; bool upd; const char* name; };
    const Shape shapes[] = {{513, 2000, 0, false, "X^ = H W^T (N = 513 thin, K = 2000)"},
                            {2000, 512, 1, true, "G = R W (N = 2000, K = 512 + 1 tail), read-modify-write epilogue"},
                            {2000, 256, 1, true, "G = R W (N = 2000, K = 256 + 1 tail), read-modify-write epilogue"}};
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& s : shapes) {
        const int lda = (s.K + s.ktail + 3) / 4 * 4, ldb = lda, ldc = (s.N + 3) / 4 * 4;
        std::vector<float> hA((size_t)M * lda), hB((size_t)(s.N + 1) * ldb);
        for (auto& v : hA) { const float u = U(rng); v = bf16_exact ? 0.1015625f : const_a ? 0.1f : (u < 0.3f ? 0.f : u * u * 3.f); }
        for (auto& v : hB) {
            const float u = U(rng); v = u * u * u * u;
            if (bf16_exact) { unsigned b; memcpy(&b, &v, 4); b = (b + 0x7fffu + ((b >> 16) & 1u)) & 0xffff0000u; memcpy(&v, &b, 4); }
        }
        float *A, *B, *C0, *C1;
        CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4));
        CK(hipMalloc(&C0, (size_t)M * ldc * 4)); CK(hipMalloc(&C1, (size_t)M * ldc * 4));
        CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        gemm::Operands g{A, B, M, s.N, s.K, lda, ldb, s.ktail};
        double tf[2] = {0, 0};
        for (int mode = 0; mode < 2; ++mode) {
            tl_matrix_mode = mode;
            float* C = mode ? C1 : C0;
            CK(hipMemset(C, 0, (size_t)M * ldc * 4));
            auto run = [&]() {
                return s.upd ? gemm::launch(g, EpiUpd{C, ldc, -0.0025f, 0.0025f}, 0) : gemm::launch(g, EpiStore{C, ldc}, 0);
            };
            CK(run());                       // (first call: output checked below)
            std::vector<float> hC((size_t)M * ldc);
            CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
            // error against fp64 on sampled entries
            double mx = 0, ss = 0, ref_mx = 0, sm = 0; int cnt = 0;
            for (int t = 0; t < 4000; ++t) {
                const int64_t r = (int64_t)(U(rng) * M) % M; const int c = (int)(U(rng) * s.N) % s.N;
                double acc = 0;
                for (int k = 0; k < s.K + s.ktail; ++k) acc += (double)hA[r * lda + k] * (double)hB[(size_t)c * ldb + k];
                double ref = acc;
                if (s.upd) { ref = 0.0 + (double)-0.0025f + (double)0.0025f * acc; ref = ref > 0 ? ref : 0; }
                const double d = (double)hC[r * ldc + c] - ref;
                mx = std::fmax(mx, std::fabs(d)); ss += d * d; sm += d; ref_mx = std::fmax(ref_mx, std::fabs(ref)); ++cnt;
            }
            for (int w = 0; w < 2; ++w) CK(run());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) CK(run());
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            tf[mode] = 2.0 * M * s.N * (s.K + s.ktail) * reps / (ms * 1e-3) / 1e12;
            printf("%-72s %s: %7.1f us, %6.1f TFLOP/s-eq; vs fp64 max %.2e rms %.2e mean %+.2e (of max |ref| %.3g)\n", s.name,
                   mode ? "bf16x3" : "f32   ", ms * 1e3 / reps, tf[mode], mx / ref_mx, std::sqrt(ss / cnt) / ref_mx, sm / cnt / ref_mx, ref_mx);
        }
        printf("   -> %.2fx\n", tf[1] / tf[0]);
#ifdef X3_TIMELINE
        {
            unsigned long long hT[64][8];
            CK(hipMemcpyFromSymbol(hT, HIP_SYMBOL(gemm::g_x3_timeline), sizeof(hT)));
            double sum[8] = {0}; int n = 0;
            for (int b = 0; b < 64; ++b) if (hT[b][7]) { for (int k = 0; k < 8; ++k) sum[k] += (double)hT[b][k]; ++n; }
            if (n) printf("   timeline (wave 0, mean over %d workgroups, cycles per k-tile over %g k-tiles): MFMA phase %.0f, barrier %.0f, "
                          "stores %.0f, barrier %.0f | prologue %.0f, epilogue %.0f cycles per workgroup\n", n, sum[7] / n,
                          sum[0] / sum[7], sum[1] / sum[7], sum[2] / sum[7], sum[3] / sum[7], sum[4] / n, sum[5] / n);
        }
#endif
        (void)hipFree(A); (void)hipFree(B); (void)hipFree(C0); (void)hipFree(C1);
    }
    return 0;
}
