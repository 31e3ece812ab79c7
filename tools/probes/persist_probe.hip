// Measurement probe (not part of libdrnmf): what would ONE phase of the recurrent cell cost if
// the all-to-all between its two GEMMs rode on an in-kernel grid barrier (persistent kernel)
// instead of a kernel boundary?  VERDICT r1 item 4a asks for the measurement on the case that is
// kindest to the persistent form -- tied dictionary, slices resident on chip, NOTHING to stream --
// so the phase body here is only the exchange itself:
//     every workgroup publishes its 2 KB slice of a 512 KB activation matrix (64 rows x 2048
//     atoms fp32, the C2 shape's h), the grid synchronises, every workgroup reads the 64 KB it
//     would contract next (16 rows x 1024 atoms) and checks every word.
// Variants:  -m 0  XCD-hierarchical barrier, plain stores + leader release fence + acquire fences
//                  (MI355X_MICROARCH.md price list row "barrier-xcd")
//            -m 1  same barrier, payload stored write-through (sc1) and read with sc1 loads
//                  (no fences at all: counters only)
//            -m 2  the SAME phase body as one kernel per phase, replayed from a hipGraph
//                  (the kernel-boundary form libdrnmf uses)
//            -m 3  empty phases through the barrier (barrier cost alone)
//            -m 4  empty kernels from a hipGraph (boundary cost alone)
// Build: hipcc --offload-arch=gfx950:xnack- -O3 -o persist_probe persist_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int NWG = 256, NT = 256;
constexpr int ROWS = 64, COLS = 2048;           // exchanged matrix (fp32): 512 KB
constexpr int SLICE = ROWS * COLS / NWG;        // floats published per workgroup (512 = 2 KB)
constexpr int READ = 16 * 1024;                 // floats read per workgroup (64 KB)

struct Sync {
    unsigned census[8 * 32];     // blocks per XCC (stride 32 words = own 128-B line)
    unsigned xcc_cnt[8 * 32];
    unsigned xcc_gen[8 * 32];
    unsigned top[32];
    unsigned flat[32];
    unsigned errors[32];
    unsigned timeout[32];
};

__device__ __forceinline__ unsigned ld_rlx(unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_rlx(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned add_rlx(unsigned* p, unsigned v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned want, unsigned* tmo) {
    for (unsigned spins = 0; ld_rlx(p) < want; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if (spins > (1u << 22)) { st_rlx(tmo, 1); return false; }
    }
    return true;
}

// XCD-hierarchical grid barrier, lane 0 of the block only (caller brackets with __syncthreads).
// FENCES: plain payload -> the XCD's last arriver writes the shared L2 back (release), everyone
// invalidates its L1 / stale L2 lines afterwards (acquire).  !FENCES: sc1 payload, counters only.
template <bool FENCES>
__device__ __forceinline__ void grid_barrier(Sync* S, unsigned epoch, int xcc, unsigned n_here,
                                             unsigned n_xcc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned prev = add_rlx(&S->xcc_cnt[xcc * 32], 1);
    if (prev == epoch * n_here - 1) {                       // last arriver of this XCD: leader
        if (FENCES) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        add_rlx(&S->top[0], 1);
        wait_ge(&S->top[0], epoch * n_xcc, &S->timeout[0]);
        if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        st_rlx(&S->xcc_gen[xcc * 32], epoch);
    } else {
        wait_ge(&S->xcc_gen[xcc * 32], epoch, &S->timeout[0]);
        if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

__device__ __forceinline__ float val_of(unsigned phase, unsigned idx) {
    return (float)((phase * 131u + idx * 7u) & 0xffffu);
}

// publish this block's slice for `phase` (2 floats per thread)
template <bool SC1>
__device__ __forceinline__ void publish(float* E, unsigned phase, int b, int tid) {
    const unsigned i0 = (unsigned)b * SLICE + tid * 2;
    f32x2 v = {val_of(phase, i0), val_of(phase, i0 + 1)};
    if (SC1) {
        unsigned long long bits;
        memcpy(&bits, &v, 8);
        __hip_atomic_store((unsigned long long*)(E + i0), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *(f32x2*)(E + i0) = v;
    }
}

// read the 64 KB this block would contract and check every word
__device__ __forceinline__ unsigned consume(const float* E, unsigned phase, int b, int tid) {
    const unsigned base = (unsigned)((b * 37) % (ROWS * COLS / READ)) * READ;
    unsigned bad = 0;
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = *(const f32x4*)(E + base + (i * NT + tid) * 4);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned idx = base + (i * NT + tid) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) bad += v[i][e] != val_of(phase, idx + e);
    }
    return bad;
}

// sc1 (L1-bypassing) 16-byte loads through the buffer intrinsic: aux bit 4 (16) = sc1
__device__ __forceinline__ unsigned consume_sc1(const float* E, unsigned phase, int b, int tid) {
    const unsigned base = (unsigned)((b * 37) % (ROWS * COLS / READ)) * READ;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)E, 0, ROWS * COLS * 4, 0x00020000);
    unsigned bad = 0;
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned idx = base + (i * NT + tid) * 4;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, idx * 4, 0, 16);
        memcpy(&v[i], &raw, 16);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned idx = base + (i * NT + tid) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) bad += v[i][e] != val_of(phase, idx + e);
    }
    return bad;
}

template <int MODE>   // 0 plain + fences, 1 sc1 + counters only, 3 empty phases
__global__ void __launch_bounds__(NT) persistent_kernel(float* E0, float* E1, Sync* S, int phases) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ unsigned s_xcc, s_here, s_nxcc;
    if (tid == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
        s_xcc = x & 7;
        add_rlx(&S->census[(x & 7) * 32], 1);
        add_rlx(&S->flat[0], 1);
        wait_ge(&S->flat[0], NWG, &S->timeout[0]);          // flat barrier once: census complete
        s_here = ld_rlx(&S->census[(x & 7) * 32]);
        unsigned n = 0;
        for (int i = 0; i < 8; ++i) n += ld_rlx(&S->census[i * 32]) != 0;
        s_nxcc = n;
    }
    __syncthreads();
    const int xcc = s_xcc;
    const unsigned n_here = s_here, n_xcc = s_nxcc;
    unsigned bad = 0;
    for (int p = 0; p < phases; ++p) {
        float* E = (p & 1) ? E1 : E0;                        // double buffer: no WAR hazard
        if (MODE != 3) publish<MODE == 1>(E, (unsigned)p, b, tid);
        __syncthreads();
        if (tid == 0) grid_barrier<MODE == 0>(S, (unsigned)p + 1, xcc, n_here, n_xcc);
        __syncthreads();
        if (MODE == 0) bad += consume(E, (unsigned)p, b, tid);
        if (MODE == 1) bad += consume_sc1(E, (unsigned)p, b, tid);
    }
    if (bad) atomicAdd(&S->errors[0], bad);
}

template <int MODE>   // 2: body, 4: empty
__global__ void __launch_bounds__(NT) phase_kernel(float* Ein, float* Eout, Sync* S, int p) {
    if (MODE == 4) return;
    const int b = blockIdx.x, tid = threadIdx.x;
    unsigned bad = 0;
    if (p > 0) bad = consume(Ein, (unsigned)p - 1, b, tid);      // written by the previous launch
    publish<false>(Eout, (unsigned)p, b, tid);
    if (bad) atomicAdd(&S->errors[0], bad);
}

int main(int argc, char** argv) {
    int mode = 0, phases = 2000, reps = 5;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "-m")) mode = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "-p")) phases = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "-r")) reps = atoi(argv[i + 1]);
    }
    float *E0, *E1;
    Sync* S;
    CK(hipMalloc(&E0, ROWS * COLS * 4));
    CK(hipMalloc(&E1, ROWS * COLS * 4));
    CK(hipMalloc(&S, sizeof(Sync)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipGraphExec_t gexec = nullptr;
    if (mode == 2 || mode == 4) {
        hipGraph_t g;
        CK(hipGraphCreate(&g, 0));
        hipGraphNode_t last = nullptr;
        std::vector<int> ps(phases);
        for (int p = 0; p < phases; ++p) {
            ps[p] = p;
            float* Ein = (p & 1) ? E0 : E1;
            float* Eout = (p & 1) ? E1 : E0;
            void* kp[4] = {&Ein, &Eout, &S, &ps[p]};
            hipKernelNodeParams np;
            memset(&np, 0, sizeof(np));
            np.func = mode == 2 ? (void*)&phase_kernel<2> : (void*)&phase_kernel<4>;
            np.gridDim = dim3(NWG);
            np.blockDim = dim3(NT);
            np.kernelParams = kp;
            hipGraphNode_t node;
            CK(hipGraphAddKernelNode(&node, g, last ? &last : nullptr, last ? 1 : 0, &np));
            last = node;
        }
        CK(hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0));
    }
    float best = 1e30f;
    unsigned err = 0, tmo = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(S, 0, sizeof(Sync), st));
        CK(hipEventRecord(e0, st));
        if (mode == 2 || mode == 4) {
            CK(hipGraphLaunch(gexec, st));
        } else {
            void* kp[4] = {&E0, &E1, &S, &phases};
            void* f = mode == 0 ? (void*)&persistent_kernel<0>
                    : mode == 1 ? (void*)&persistent_kernel<1> : (void*)&persistent_kernel<3>;
            CK(hipLaunchKernel(f, dim3(NWG), dim3(NT), kp, 0, st));
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        Sync h;
        CK(hipMemcpy(&h, S, sizeof(Sync), hipMemcpyDeviceToHost));
        err += h.errors[0];
        tmo += h.timeout[0];
        if (r == 0 && mode != 2 && mode != 4) {
            printf("  census per XCC:");
            for (int i = 0; i < 8; ++i) printf(" %u", h.census[i * 32]);
            printf("\n");
        }
    }
    const char* names[] = {"persistent, plain stores + xcd barrier with fences",
                           "persistent, sc1 stores/loads + xcd barrier (counters only)",
                           "one kernel per phase from a hipGraph (boundary)",
                           "persistent, empty phases (xcd barrier alone, no fences)",
                           "empty kernels from a hipGraph (boundary alone)"};
    printf("mode %d (%s): %d phases, best of %d: %.3f ms = %.2f us per phase; wrong words %u, timeouts %u\n",
           mode, names[mode], phases, reps, best, best * 1e3f / phases, err, tmo);
    return (err || tmo) ? 2 : 0;
}
