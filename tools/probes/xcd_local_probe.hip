// Measurement probe (not part of libdrnmf): the small-shape question persist_probe.hip left open.
// A single utterance (BASELINE configs[0]: B = 1, N = 200, K = 10) runs the Gram form of the cell on
// 13 workgroups per launch and is pure launch latency (1.9 us per layer-step).  Would ONE persistent
// kernel whose few workgroups all sit on ONE XCD (they share that XCD's L2, so the grid barrier and
// the exchanged activations never leave it) beat the kernel boundary there?
//   phase = every participant publishes its slice of a small activation matrix (16 rows x N fp32),
//           the participants synchronise, every participant reads the WHOLE matrix and checks it.
// Workgroups are dealt round-robin to the 8 XCDs by id, so the grid is 8 x W and only ids = x (mod 8)
// take part; the others exit at once.  The census of HW_REG_XCC_ID is printed: if the participants
// do not share an XCD the number means nothing.
//   xcd_local_probe <W participants> <N atoms> [phases] [mode]
//     mode 0: sc1 (L1-bypassing) stores / loads + one agent-scope counter (correct on any placement)
//     mode 1: the same exchange over kernel boundaries: one launch of W workgroups per phase (hipGraph)
//     mode 2: mode 0 with empty phases (the barrier alone)
//     mode 3..7: PLAIN producer stores (line kept dirty in the shared L2) + sc1 / nt / sc0 / plain / sc0 sc1 loads
// Build: hipcc --offload-arch=gfx950:xnack- -O3 -o xcd_local_probe xcd_local_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NT = 256, ROWS = 16;
struct Sync { unsigned cnt[32]; unsigned census[8 * 32]; unsigned errors[32]; unsigned timeout[32]; };

__device__ __forceinline__ unsigned ld_rlx(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_rlx(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float val_of(unsigned phase, unsigned idx) { return (float)((phase * 131u + idx * 7u) & 0xffffu); }

__device__ __forceinline__ void st_sc1(float* p, float v) {
    unsigned bits; memcpy(&bits, &v, 4);
    __hip_atomic_store((unsigned*)p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    const unsigned bits = __hip_atomic_load((unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float v; memcpy(&v, &bits, 4); return v;
}

// PLAIN: the producers' stores are plain (the line stays, dirty, in the XCD's L2; the store's vmcnt
// acknowledgement = it reached the L2); AUX: cache-policy bits of the consumers' loads (16 = sc1,
// 2 = nt, 1 = sc0, 0 = plain).  Only valid when every participant shares one XCD (one L2).
template <bool PLAIN, int AUX, bool L2ATOM = false>
__global__ void __launch_bounds__(NT) persistent_kernel(float* E0, float* E1, Sync* S, int W, int total,
                                                        int phases, int empty) {
    if ((blockIdx.x & 7) != 0) return;
    const int b = blockIdx.x >> 3, tid = threadIdx.x;
    if (tid == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
        add_rlx(&S->census[(x & 7) * 32], 1);
    }
    const int per = (total + W - 1) / W;
    unsigned bad = 0;
    for (int p = 0; p < phases; ++p) {
        float* E = (p & 1) ? E1 : E0;
        if (!empty)
            for (int i = b * per + tid; i < (b + 1) * per && i < total; i += NT) {
                if (PLAIN) E[i] = val_of(p, i);
                else st_sc1(E + i, val_of(p, i));
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            // L2ATOM: the arrival is a workgroup-scope atomic (no sc1: executed in the XCD's L2, not
            // sent to memory) -- only valid when every participant shares that L2
            if (L2ATOM) __hip_atomic_fetch_add(&S->cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else add_rlx(&S->cnt[0], 1);
            unsigned spins = 0;
            while (ld_rlx(&S->cnt[0]) < (unsigned)(p + 1) * W) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { S->timeout[0] = 1; break; }
            }
        }
        __syncthreads();
        if (!empty) {      // 16-byte sc1 loads, all in flight together (total <= 16 * NT * 4 floats)
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)E, 0, total * 4, 0x00020000);
            u32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (k * NT + tid) * 16, 0, AUX);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = (k * NT + tid) * 4 + e;
                    float f; unsigned u = v[k][e]; memcpy(&f, &u, 4);
                    if (i < total) bad += f != val_of(p, i);
                }
        }
    }
    if (bad) atomicAdd(&S->errors[0], bad);
}

__global__ void __launch_bounds__(NT) phase_kernel(const float* Ein, float* Eout, Sync* S, int W, int total, int p) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int per = (total + W - 1) / W;
    unsigned bad = 0;
    if (p > 0) {
        typedef __attribute__((ext_vector_type(4))) float f32x4;
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = (k * NT + tid) * 4;
            v[k] = i + 3 < total ? *(const f32x4*)(Ein + i) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = (k * NT + tid) * 4 + e;
                if (i < total && i + 3 - e < total) bad += v[k][e] != val_of(p - 1, i);
            }
    }
    for (int i = b * per + tid; i < (b + 1) * per && i < total; i += NT) Eout[i] = val_of(p, i);
    if (bad) atomicAdd(&S->errors[0], bad);
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 13, N = argc > 2 ? atoi(argv[2]) : 208;
    const int phases = argc > 3 ? atoi(argv[3]) : 4000, mode = argc > 4 ? atoi(argv[4]) : 0;
    int total = ROWS * N;
    float *E0, *E1; Sync* S;
    CK(hipMalloc(&E0, total * 4)); CK(hipMalloc(&E1, total * 4)); CK(hipMalloc(&S, sizeof(Sync)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraphExec_t gexec = nullptr;
    int Wv = W, ph = phases;
    std::vector<int> ps(phases);
    if (mode == 1) {
        hipGraph_t g; CK(hipGraphCreate(&g, 0));
        hipGraphNode_t last = nullptr;
        for (int p = 0; p < phases; ++p) {
            ps[p] = p;
            float* Ein = (p & 1) ? E0 : E1; float* Eout = (p & 1) ? E1 : E0;
            void* kp[6] = {&Ein, &Eout, &S, &Wv, &total, &ps[p]};
            hipKernelNodeParams np; memset(&np, 0, sizeof(np));
            np.func = (void*)&phase_kernel; np.gridDim = dim3(W); np.blockDim = dim3(NT); np.kernelParams = kp;
            hipGraphNode_t node;
            CK(hipGraphAddKernelNode(&node, g, last ? &last : nullptr, last ? 1 : 0, &np));
            last = node;
        }
        CK(hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0));
    }
    float best = 1e30f; unsigned err = 0, tmo = 0;
    for (int r = 0; r < 5; ++r) {
        CK(hipMemsetAsync(S, 0, sizeof(Sync), st));
        CK(hipEventRecord(e0, st));
        if (mode == 1) CK(hipGraphLaunch(gexec, st));
        else {
            int empty = mode == 2 || mode == 9;
            void* kp[7] = {&E0, &E1, &S, &Wv, &total, &ph, &empty};
            void* f = (void*)&persistent_kernel<false, 16>;
            if (mode == 3) f = (void*)&persistent_kernel<true, 16>;
            if (mode == 4) f = (void*)&persistent_kernel<true, 2>;
            if (mode == 5) f = (void*)&persistent_kernel<true, 1>;
            if (mode == 6) f = (void*)&persistent_kernel<true, 0>;
            if (mode == 7) f = (void*)&persistent_kernel<true, 17>;
            if (mode == 8 || mode == 9) f = (void*)&persistent_kernel<true, 16, true>;
            CK(hipLaunchKernel(f, dim3(8 * W), dim3(NT), kp, 0, st));
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        Sync h; CK(hipMemcpy(&h, S, sizeof(Sync), hipMemcpyDeviceToHost));
        err += h.errors[0]; tmo += h.timeout[0];
        if (r == 0 && mode != 1) {
            printf("  participants per XCC:");
            for (int i = 0; i < 8; ++i) printf(" %u", h.census[i * 32]);
            printf("\n");
        }
    }
    printf("W %d, %d x %d fp32 exchanged, mode %d (%s): %.2f us per phase; wrong words %u, timeouts %u\n", W, ROWS,
           N, mode, mode == 1 ? "one launch per phase" : mode == 2 ? "persistent, empty phases: barrier alone" : mode == 3 ? "persistent, PLAIN stores + sc1 loads" : mode == 4 ? "persistent, PLAIN stores + nt loads" : mode == 5 ? "persistent, PLAIN stores + sc0 loads" : mode == 6 ? "persistent, PLAIN stores + plain loads" : mode == 7 ? "persistent, PLAIN stores + sc0 sc1 loads" : mode == 8 ? "persistent, PLAIN stores + sc1 loads, L2 (workgroup-scope) arrival atomics" : mode == 9 ? "barrier alone, L2 arrival atomics" : "persistent, participants on one XCD", best * 1e3f / phases, err, tmo);
    return (err || tmo) ? 2 : 0;
}
