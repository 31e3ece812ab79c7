// Measurement probe (not part of libdrnmf), VERDICT r2 item 1: what does one PHASE of the headline
// chain cost when the batch's row tiles run as INDEPENDENT persistent chains?
//
// Batch rows never interact (custom_layers.py:337-338, 346-348), so B = 64 is 4 chains of 16 rows.
// A chain's layer-step is two phases with an all-to-all between them:
//     B-type phase: workgroup (bin tile, atom range) reads its half of h (16 x 1024 fp32 = 64 KB,
//                   written by the chain's other workgroups in the phase before), contracts it with a
//                   64 KB dictionary slice (private: prefetchable), publishes a 16 x 16 residual tile;
//     A-type phase: workgroup (atom block) reads both residual partials (2 x 32 KB), contracts them
//                   with its 64 KB dictionary slice, publishes its 16 x 32 slice of h.
// Every workgroup of a chain takes part in both phases; a per-chain barrier separates them.  The probe
// runs exactly that with synthetic values (every exchanged word is checked), real MFMAs
// (v_mfma_f32_16x16x4_f32, 32 per wave and phase = the C2 shape's count) and a real dictionary
// stream (K = 25 untied layers x 2 packings x 4 MB = 205 MB walked layer by layer).
//
//   chain_probe <chains> <placement> <features> <barrier> <prefetch> [phases]
//     chains     1 | 2 | 4 | 8         workgroups per chain = 256 / chains (one workgroup per CU)
//     placement  0 confined: chain c owns XCDs [c*8/chains, (c+1)*8/chains)
//                1 spread:   every chain has workgroups on all 8 XCDs; the workgroups of different
//                            chains that use the same dictionary slice share an XCD (its L2)
//     features   bit 0 exchange (sc1 stores / loads, checked), bit 1 MFMAs, bit 2 dictionary stream,
//                bit 3 in-kernel timeline of workgroup 0 of chain 0 (adds a full wait per phase)
//     barrier    0 one counter per chain; 1 two levels (groups of 8 workgroups, then the chain)
//     prefetch   0 next phase's dictionary requested between arriving at the barrier and polling
//                1 the polling wave requests its share only after the barrier
// Build: hipcc --offload-arch=gfx950:xnack- -O3 -o chain_probe chain_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int NT = 512, NWV = 8;
constexpr int HSZ = 16 * 2048;          // floats of h per chain (128 KB)
constexpr int RSZ = 2 * 16 * 512;       // floats of the two residual partials per chain (64 KB)
constexpr int SLICE = 16384;            // floats of one dictionary slice (64 KB)
constexpr int KL = 25;                  // untied layers walked

struct Sync {
    unsigned cnt[8 * 64];               // per chain: top counter (own 256-B line)
    unsigned grp[8 * 32 * 64];          // per chain and group of 8 workgroups
    unsigned census[256];               // HW id of every workgroup
    unsigned errors[64];
    unsigned timeout[64];
    unsigned long long tl[8][2][8];
    unsigned flags[8 * 256];            // protocol 1: per chain, one word per workgroup = phases completed     // features bit 3: s_memtime segment sums of workgroup 0 of every chain, by phase type
};

__device__ __forceinline__ unsigned ld_rlx(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_rlx(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float val_of(unsigned phase, unsigned idx) { return (float)((phase * 131u + idx * 7u) & 0xffu); }
__device__ __forceinline__ f32x4 ld4_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16);   // aux 16 = sc1
    f32x4 v; memcpy(&v, &raw, 16); return v;
}
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct Args {
    float* H; float* R;                 // [chains][HSZ], [chains][RSZ]
    const float* D;                     // [KL][2][wpc][SLICE]
    Sync* S;
    int chains, wpc, placement, features, barrier, prefetch, phases, protocol, partial;
};

__global__ void __launch_bounds__(NT) chain_kernel(const Args a) {
    extern __shared__ float lds[];      // (96 KB requested: one workgroup per CU) red[8][16][33]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int xcd = b & 7, jj = b >> 3;                      // observed: block b runs on XCD b % 8
    int chain, i;
    if (a.placement == 0) {
        const int xpc = 8 / a.chains;                        // XCDs per chain
        chain = xcd / xpc;
        i = jj * xpc + (xcd % xpc);
    } else {
        chain = jj % a.chains;
        i = (jj / a.chains) * 8 + xcd;
    }
    if (tid == 0) {
        unsigned x, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        a.S->census[b] = ((x & 7) << 16) | ((hw >> 8) & 0xff);
    }
    const bool ex = a.features & 1, mm = a.features & 2, ds = a.features & 4;
    float* H = a.H + (size_t)chain * HSZ;
    float* R = a.R + (size_t)chain * RSZ;
    __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)H, 0, HSZ * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)R, 0, RSZ * 4, 0x00020000);
    unsigned* cnt = &a.S->cnt[chain * 64];
    unsigned* grp = &a.S->grp[(chain * 32 + (i >> 3)) * 64];
    const unsigned ngrp = (unsigned)a.wpc / 8;

    f32x4 dN[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) dN[g] = f32x4{1.f, 1.f, 1.f, 1.f};
    auto prefetch = [&](int p) {                              // dictionary slice of phase p
        if (!ds) return;
        const int layer = (p >> 1) % KL, type = p & 1;
        const float* src = a.D + ((size_t)(layer * 2 + type) * a.wpc + i) * SLICE;
#pragma unroll
        for (int g = 0; g < 8; ++g) dN[g] = *(const f32x4*)(src + (g * NT + tid) * 4);
    };
    prefetch(0);
    unsigned bad = 0;
    const bool stamp = (a.features & 8) && i == 0 && w == 0;
    unsigned long long seg[2][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}}, t0 = 0, t1;
#define STAMP(k) do { if (stamp) { t1 = __builtin_amdgcn_s_memtime(); seg[type][k] += t1 - t0; t0 = t1; } } while (0)
    if (stamp) t0 = __builtin_amdgcn_s_memtime();
    for (int p = 0; p < a.phases; ++p) {
        const int type = p & 1;                               // 0: A-type (reads R, writes H); 1: B-type
        f32x4 d[8], e[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) { d[g] = dN[g]; e[g] = f32x4{1.f, 1.f, 1.f, 1.f}; }
        if (ex && p > 0) {
            // B-type: the half of h of atom range i & 1; A-type: both residual partials
            const unsigned base = type ? (unsigned)(i & 1) * (HSZ / 2) : 0u;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const unsigned idx = base + (g * NT + tid) * 4;
                e[g] = ld4_sc1(type ? hrs : rrs, idx * 4);
            }
        }
        STAMP(0);                                             // loads issued
        if (stamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        STAMP(1);                                             // wave 0's operands there
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (mm) {
            if (type) {
#pragma unroll
                for (int g = 0; g < 8; ++g)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (s & 1) acc1 = mfma16(e[g][s], d[g][s], acc1);
                        else acc0 = mfma16(e[g][s], d[g][s], acc0);
                    }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 r4 = e[2 * g] + e[2 * g + 1];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc0 = mfma16(r4[s], d[2 * g][s], acc0);
                        acc1 = mfma16(r4[s], d[2 * g + 1][s], acc1);
                    }
                }
            }
        }
        if (ex && p > 0) {
            const unsigned base = type ? (unsigned)(i & 1) * (HSZ / 2) : 0u;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const unsigned idx = base + (g * NT + tid) * 4;
#pragma unroll
                for (int c = 0; c < 4; ++c) bad += e[g][c] != val_of((unsigned)p - 1, idx + c);
            }
        }
        // cross-wave reduce through LDS (A-type: 16 x 32 outputs, B-type: 16 x 16)
        const int j = l & 15, q = l >> 4;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            lds[(w * 16 + 4 * q + v) * 33 + j] = acc0[v];
            lds[(w * 16 + 4 * q + v) * 33 + 16 + j] = acc1[v];
        }
        __syncthreads();
        STAMP(2);                                             // MFMAs, checks, LDS write, workgroup barrier
        if (tid < 256) {
            const int er = tid >> 4, ec = tid & 15;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int ww = 0; ww < NWV; ++ww) {
                s0 += lds[(ww * 16 + er) * 33 + ec];
                s1 += lds[(ww * 16 + er) * 33 + 16 + ec];
            }
            const float z = (s0 + s1) * 0.f;                  // (finite sums: a true dependency, value 0)
            if (ex) {
                if (type) {                                   // residual tile: 256 floats per workgroup
                    const unsigned idx = (unsigned)i * (RSZ / a.wpc) + tid;
                    if (tid < RSZ / a.wpc) {
                        float v = val_of((unsigned)p, idx) + z;
                        unsigned u; memcpy(&u, &v, 4);
                        __hip_atomic_store((unsigned*)(R + idx), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else {                                      // h slice: 512 floats per workgroup (wpc = 64)
                    const unsigned per = HSZ / a.wpc;
                    if (tid * 2 < per) {
                        const unsigned idx = (unsigned)i * per + tid * 2;
                        f32x2 v = {val_of((unsigned)p, idx) + z, val_of((unsigned)p, idx + 1) + z};
                        unsigned long long u; memcpy(&u, &v, 8);
                        __hip_atomic_store((unsigned long long*)(H + idx), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            } else if (z != 0.f) bad++;
        }
        // ---- per-chain barrier ----------------------------------------------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        STAMP(3);                                             // reduce, stores acknowledged, workgroup barrier
        if (tid == 0) {
            if (a.barrier == 0) add_rlx(cnt, 1);
            else {
                const unsigned prev = add_rlx(grp, 1);
                if (prev == (unsigned)(p + 1) * 8 - 1) add_rlx(cnt, 1);
            }
        }
        if (a.prefetch == 0 || w != 0) prefetch(p + 1);
        if (tid == 0) {
            const unsigned want = (unsigned)(p + 1) * (a.barrier == 0 ? (unsigned)a.wpc : ngrp);
            unsigned spins = 0;
            while (ld_rlx(cnt) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { a.S->timeout[0] = 1; break; }
            }
        }
        if (a.prefetch == 1 && w == 0) prefetch(p + 1);
        STAMP(4);                                             // arrive, prefetch issue, poll
        __syncthreads();
        STAMP(5);
    }
    if (stamp && l == 0)
        for (int ty = 0; ty < 2; ++ty)
            for (int k = 0; k < 6; ++k) a.S->tl[chain][ty][k] = seg[ty][k];
    if (bad) atomicAdd(&a.S->errors[0], bad);
}


// ---- protocol 1: per-producer flags + a service wave ---------------------------------------------------
// 9 waves: waves 0-7 load / contract / prefetch, wave 8 reduces the partials, publishes the tile with
// 16-byte write-through stores, waits for their acknowledgement, raises THIS workgroup's flag (one
// word per workgroup, = phases completed) and polls the flags of the producers it depends on with
// ONE load per poll (64 flags = 256 contiguous bytes).  No atomics, no counter serialisation, no
// workgroup-wide store drain.  partial = 1: a B-type phase waits only for the 32 producers of its
// atom range.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ void __launch_bounds__(576) chain_kernel2(const Args a) {
    extern __shared__ float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int xcd = b & 7, jj = b >> 3;
    int chain, i;
    if (a.placement == 0) {
        const int xpc = 8 / a.chains;
        chain = xcd / xpc;
        i = jj * xpc + (xcd % xpc);
    } else {
        chain = jj % a.chains;
        i = (jj / a.chains) * 8 + xcd;
    }
    if (tid == 0) {
        unsigned x, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        a.S->census[b] = ((x & 7) << 16) | ((hw >> 8) & 0xff);
    }
    const bool ex = a.features & 1, mm = a.features & 2, ds = a.features & 4;
    float* H = a.H + (size_t)chain * HSZ;
    float* R = a.R + (size_t)chain * RSZ;
    __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)H, 0, HSZ * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)R, 0, RSZ * 4, 0x00020000);
    unsigned* flags = &a.S->flags[chain * 256];
    __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)flags, 0, 256 * 4, 0x00020000);
    const bool stamp = (a.features & 8) && i == 0;
    unsigned long long seg[2][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}}, t0 = 0, t1;
    unsigned bad = 0;

    if (w == 8) {
        // ---------------- service wave ----------------
        if (stamp) t0 = __builtin_amdgcn_s_memtime();
        for (int p = 0; p < a.phases; ++p) {
            const int type = p & 1;
            if (p > 0) {
                // producers of phase p-1 this phase depends on: all (A-type) or one atom range's (B-type)
                int lo = 0, hi = a.wpc;
                if (type && a.partial) { lo = (i & 1) * (a.wpc / 2); hi = lo + a.wpc / 2; }
                unsigned spins = 0;
                bool done = false;
                while (!done) {
                    done = true;
                    for (int f0 = 0; f0 < a.wpc; f0 += 64) {
                        const int f = f0 + l;
                        const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(frs, f * 4, 0, 16);
                        const bool ok = v >= (unsigned)p || f < lo || f >= hi;
                        if (__builtin_amdgcn_ballot_w64(ok) != ~0ull) done = false;
                    }
                    if (++spins > (1u << 22)) { a.S->timeout[0] = 1; break; }
                }
            }
            STAMP(0);                                         // poll
            wg_barrier();                                     // (B) release the compute waves
            wg_barrier();                                     // (A) their partials are in LDS
            STAMP(1);                                         // compute waves: loads + MFMAs
            // reduce: lane l owns outputs 4l..4l+3 (B-type, 16 x 16) / 8l..8l+7 (A-type, 16 x 32)
            const int nout = type ? 4 : 8;
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                o[k] = 0.f;
                if (k < nout) {
                    const int e = l * nout + k, er = e / (type ? 16 : 32), ec = e % (type ? 16 : 32);
#pragma unroll
                    for (int ww = 0; ww < NWV; ++ww) o[k] += lds[(ww * 16 + er) * 33 + ec];
                }
            }
            if (ex) {
                if (type) {
                    const unsigned idx = (unsigned)i * (RSZ / a.wpc) + l * 4;
                    if (l * 4 < RSZ / a.wpc) {
                        f32x4 v;
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = val_of((unsigned)p, idx + k) + o[k] * 0.f;
                        u32x4 u; memcpy(&u, &v, 16);
                        __builtin_amdgcn_raw_buffer_store_b128(u, rrs, idx * 4, 0, 16);
                    }
                } else {
                    const unsigned per = HSZ / a.wpc;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const unsigned idx = (unsigned)i * per + (hh * 64 + l) * 4;
                        if ((hh * 64 + l) * 4 < per) {
                            f32x4 v;
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = val_of((unsigned)p, idx + k) + o[hh * 4 + k] * 0.f;
                            u32x4 u; memcpy(&u, &v, 16);
                            __builtin_amdgcn_raw_buffer_store_b128(u, hrs, idx * 4, 0, 16);
                        }
                    }
                }
            }
            STAMP(2);                                         // reduce + stores issued
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(3);                                         // stores acknowledged
            if (l == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(p + 1), frs, i * 4, 0, 16);
        }
        if (stamp && l == 0)
            for (int ty = 0; ty < 2; ++ty)
                for (int k = 0; k < 6; ++k) a.S->tl[chain][ty][k] = seg[ty][k];
        return;
    }

    // ---------------- compute waves ----------------
    const int ctid = tid;                                     // 0..511
    f32x4 dN[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) dN[g] = f32x4{1.f, 1.f, 1.f, 1.f};
    auto prefetch = [&](int p) {
        if (!ds) return;
        const int layer = (p >> 1) % KL, type = p & 1;
        const float* src = a.D + ((size_t)(layer * 2 + type) * a.wpc + i) * SLICE;
#pragma unroll
        for (int g = 0; g < 8; ++g) dN[g] = *(const f32x4*)(src + (g * NT + ctid) * 4);
    };
    prefetch(0);
    for (int p = 0; p < a.phases; ++p) {
        const int type = p & 1;
        wg_barrier();                                         // (B)
        f32x4 d[8], e[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) { d[g] = dN[g]; e[g] = f32x4{1.f, 1.f, 1.f, 1.f}; }
        const unsigned base = type ? (unsigned)(i & 1) * (HSZ / 2) : 0u;
        if (ex && p > 0) {
#pragma unroll
            for (int g = 0; g < 8; ++g) e[g] = ld4_sc1(type ? hrs : rrs, (base + (g * NT + ctid) * 4) * 4);
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (mm) {
            if (type) {
#pragma unroll
                for (int g = 0; g < 8; ++g)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (s & 1) acc1 = mfma16(e[g][s], d[g][s], acc1);
                        else acc0 = mfma16(e[g][s], d[g][s], acc0);
                    }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 r4 = e[2 * g] + e[2 * g + 1];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc0 = mfma16(r4[s], d[2 * g][s], acc0);
                        acc1 = mfma16(r4[s], d[2 * g + 1][s], acc1);
                    }
                }
            }
        }
        if (ex && p > 0) {
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) bad += e[g][c] != val_of((unsigned)p - 1, base + (g * NT + ctid) * 4 + c);
        }
        const int j = l & 15, q = l >> 4;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            lds[(w * 16 + 4 * q + v) * 33 + j] = acc0[v];
            lds[(w * 16 + 4 * q + v) * 33 + 16 + j] = acc1[v];
        }
        wg_barrier();                                         // (A)
        prefetch(p + 1);
    }
    if (bad) atomicAdd(&a.S->errors[0], bad);
}

int main(int argc, char** argv) {
    Args a;
    a.chains = argc > 1 ? atoi(argv[1]) : 4;
    a.placement = argc > 2 ? atoi(argv[2]) : 0;
    a.features = argc > 3 ? atoi(argv[3]) : 7;
    a.barrier = argc > 4 ? atoi(argv[4]) : 0;
    a.prefetch = argc > 5 ? atoi(argv[5]) : 0;
    a.phases = argc > 6 ? atoi(argv[6]) : 4900;
    a.protocol = argc > 7 ? atoi(argv[7]) : 0;
    a.partial = argc > 8 ? atoi(argv[8]) : 0;
    a.wpc = 256 / a.chains;
    CK(hipMalloc(&a.H, (size_t)a.chains * HSZ * 4));
    CK(hipMalloc(&a.R, (size_t)a.chains * RSZ * 4));
    const size_t dbytes = (size_t)KL * 2 * a.wpc * SLICE * 4;
    float* D; CK(hipMalloc(&D, dbytes));
    {
        std::vector<float> ones(dbytes / 4, 1.0f);
        CK(hipMemcpy(D, ones.data(), dbytes, hipMemcpyHostToDevice));
    }
    a.D = D;
    CK(hipMalloc(&a.S, sizeof(Sync)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)&chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)&chain_kernel2, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    float best = 1e30f; unsigned err = 0, tmo = 0;
    for (int r = 0; r < 4; ++r) {
        CK(hipMemsetAsync(a.S, 0, sizeof(Sync), st));
        CK(hipEventRecord(e0, st));
        if (a.protocol == 1) hipLaunchKernelGGL(chain_kernel2, dim3(256), dim3(576), 96 * 1024, st, a);
        else hipLaunchKernelGGL(chain_kernel, dim3(256), dim3(NT), 96 * 1024, st, a);
        CK(hipGetLastError());
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        Sync* h = new Sync;
        CK(hipMemcpy(h, a.S, sizeof(Sync), hipMemcpyDeviceToHost));
        err += h->errors[0]; tmo += h->timeout[0];
        if (r == 3 && (a.features & 8)) {
            const char* nm0[6] = {"issue", "operands", "mfma+lds+sync", "reduce+stores+ack", "arrive+prefetch+poll", "exit sync"};
            const char* nm1[6] = {"flag->poll ok", "loads+mfma (compute waves)", "reduce+stores issued", "stores acked", "-", "-"};
            const char** nm = a.protocol == 1 ? nm1 : nm0;
            for (int ty = 0; ty < 2; ++ty) {
                printf("  timeline chain 0 wg 0, %s-type phases (us at 2.4 GHz ticks):", ty ? "B" : "A");
                for (int k = 0; k < 6; ++k) printf(" %s %.2f", nm[k], h->tl[0][ty][k] / 2400.0 / ((a.phases + 1 - ty) / 2));
                printf("\n");
            }
        }
        if (r == 0) {
            // distinct CUs, and does block b sit on XCD b % 8?
            int off = 0, dup = 0;
            std::vector<unsigned> seen;
            for (int b = 0; b < 256; ++b) {
                if ((int)(h->census[b] >> 16) != (b & 7)) ++off;
                for (unsigned s : seen) if (s == h->census[b]) { ++dup; break; }
                seen.push_back(h->census[b]);
            }
            printf("  placement census: %d of 256 workgroups off XCD b%%8, %d share a CU\n", off, dup);
        }
        delete h;
    }
    if (a.protocol == 1) printf("[flags%s] ", a.partial ? ", partial wait" : "");
    printf("chains %d x %d wgs, %s, features %d%s%s%s, barrier %s, prefetch %s: %.3f us per phase = %.2f us per layer-step"
           " (%d phases, best of 4); wrong words %u, timeouts %u\n",
           a.chains, a.wpc, a.placement ? "spread  " : "confined", a.features, (a.features & 1) ? " exch" : "",
           (a.features & 2) ? " mfma" : "", (a.features & 4) ? " dict" : "", a.barrier ? "2-level" : "flat   ",
           a.prefetch ? "late(w0)" : "early   ", best * 1e3f / a.phases, 2 * best * 1e3f / a.phases, a.phases, err, tmo);
    return (err || tmo) ? 2 : 0;
}
