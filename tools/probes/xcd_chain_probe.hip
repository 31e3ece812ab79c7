// Measurement probe (not part of libdrnmf), round 4: what does one PHASE of
// the FACTORED cell cost when every row group of the batch is an independent persistent chain CONFINED TO
// ONE XCD (32 workgroups = that XCD's 32 CUs, exchange through its L2, one counter barrier per phase)?
//
// Round 3's chain_probe measured chains of 64+ workgroups over 2+ XCDs (barrier alone 1.9 us): negative.
// A single-XCD chain synchronises in ~0.7 us and exchanges through a shared L2, but every XCD then
// streams the WHOLE dictionary of a layer per phase (8x the fabric traffic of the launch form, where an
// XCD owns 1/8 of the atoms).  Whether that pays depends on the rows per chain -- this probe runs the
// real per-phase volumes and MFMA counts:
//
//   xcd_chain_probe <chains> <RB> <F> <KS> <CB> <K> <features> [phases]
//     chains   1..8        one per XCD; B = chains * 16 * RB rows
//     RB       1 | 2       16-row blocks per chain (every dictionary operand feeds RB row blocks)
//     F        512 | 256   MFMA bins (the odd STFT bin is outside the tiles)
//     KS, CB   atom ranges and 16-bin column blocks of a B-type workgroup: (F/16/CB) * KS must be 32
//     K        untied layers whose dictionaries (2 packings) are walked: K * 2 * F * 2048 * 4 bytes
//     features bit 0 exchange (checked), bit 1 MFMAs, bit 2 dictionary stream, bit 3 no LDS epilogue,
//              bit 4 in-kernel timeline of chain 0 / workgroup 0
//
//   A-type phase (cell_a): workgroup i owns 64 atoms; 8 waves split the F/16 bin chunks; per chunk and
//       wave: RB*KS exchanged 1-KB loads (residual partials), 4 dictionary loads, 16*RB MFMAs; publishes
//       its 16*RB x 64 slice of h.
//   B-type phase (cell_b): workgroup i owns bin block i / KS (16*CB bins) and atom range i % KS; 8 waves
//       split the range's 16-atom chunks; per chunk: RB exchanged loads (h), CB dictionary loads, 4*RB*CB
//       MFMAs; publishes its 16*RB x 16*CB residual partial.
// Every exchanged word is checked.  Build: hipcc --offload-arch=gfx950:xnack- -O3 -o xcd_chain_probe xcd_chain_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int NP = 2048;                 // atoms (2016 padded to whole 64-atom workgroup slices)
constexpr int WPC = 32;                  // workgroups per chain
constexpr int NWV = 8;                   // compute waves (a ninth wave synchronises)
constexpr int LINE = 1024;               // sync words of a chain sit 4 KB apart
#ifndef SYNC_SCOPE
#define SYNC_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif

__device__ __forceinline__ float val_of(unsigned phase, unsigned idx) { return (float)((phase * 131u + idx * 7u) & 0xffu); }
__device__ __forceinline__ f32x4 ld4_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16);   // aux 16 = sc1
    f32x4 v; memcpy(&v, &raw, 16); return v;
}
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct Args {
    float* H;            // [chains][RB*16*NP]
    float* R;            // [chains][KS][RB*16*F]
    const float* D;      // [K][2][WPC][F*64]
    unsigned* sync;      // [8][LINE]
    unsigned* errors;    // [0] wrong words, [1] timeouts
    unsigned long long* tl;   // [2][8] s_memtime segment sums (features bit 4): compute wave 0 / sync wave of chain 0, workgroup 0
    int chains, F, K, features, phases;
};

template <int RB, int KS, int CB>
__global__ void __launch_bounds__(64 * (NWV + 1)) chain_kernel(const Args a) {
    __shared__ float lds[NWV * RB * 16 * 65];
    __shared__ int ctl[2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int chain = b & 7, i = b >> 3;                       // observed: block b runs on XCD b % 8
    if (chain >= a.chains) return;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const bool ex = a.features & 1, mm = a.features & 2, ds = a.features & 4;
    const int F = a.F, nft = F / 16, NAC = NP / 16;
    const unsigned hsz = RB * 16 * NP, rsz1 = RB * 16 * F;
    float* H = a.H + (size_t)chain * hsz;
    float* R = a.R + (size_t)chain * KS * rsz1;
    __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)H, 0, hsz * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)R, 0, KS * rsz1 * 4, 0x00020000);
    unsigned* line = a.sync + chain * LINE;
    const size_t slice = (size_t)F * 64;
    unsigned bad = 0;

    // B-type geometry
    const int bb = i / KS, ks = i % KS;                        // bin block, atom range
    const int nchB = NAC / KS;                                 // 16-atom chunks of the range
    const int perB = (nchB + NWV - 1) / NWV, perA = (nft + NWV - 1) / NWV;

    const bool stamp = (a.features & 16) && chain == 0 && i == 0;
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1;
#define STAMP(k) do { if (stamp) { t1 = __builtin_amdgcn_s_memtime(); seg[k] += t1 - t0; t0 = t1; } } while (0)
    if (w == NWV) {
        // ---------------- synchronising wave ----------------
        if (stamp) t0 = __builtin_amdgcn_s_memtime();
        for (int p = 0; p < a.phases; ++p) {
            __syncthreads();                                   // (1) partials in LDS
            __syncthreads();                                   // (2) this workgroup's tile is acknowledged
            STAMP(0);                                          // waiting for the compute waves
            if (l == 0) {
                const unsigned old = __hip_atomic_fetch_add(line, 1u, __ATOMIC_RELAXED, SYNC_SCOPE);
                if (stamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); seg[7] += old & 0u; }
                STAMP(1);                                      // arrival atomic returned
                const unsigned want = (unsigned)(p + 1) * WPC;
                unsigned spins = 0;
                while (__hip_atomic_load(line, __ATOMIC_RELAXED, SYNC_SCOPE) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 21)) { a.errors[1] = 1; break; }
                }
                if (stamp) seg[6] += spins;
                STAMP(2);                                      // poll loop
            }
            __syncthreads();                                   // (3) release the compute waves
            STAMP(3);
        }
        if (stamp && l == 0)
            for (int k = 0; k < 8; ++k) a.tl[8 + k] = seg[k];
        return;
    }

    // ---------------- compute waves ----------------
    f32x4 dpre[4];                                             // first chunk's dictionary operands, prefetched
#pragma unroll
    for (int g = 0; g < 4; ++g) dpre[g] = f32x4{1.f, 1.f, 1.f, 1.f};
    auto dict_ptr = [&](int p) {
        const int layer = (p >> 1) % a.K, type = p & 1;
        return a.D + ((size_t)(layer * 2 + type) * WPC + i) * slice;
    };
    auto prefetch = [&](int p) {
        if (!ds) return;
        const float* src = dict_ptr(p);
        const int type = p & 1;
        const int nd = type ? CB : 4;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < nd) dpre[g] = *(const f32x4*)(src + ((size_t)(w * nd + g) * 256 + l * 4));
    };
    prefetch(0);
    const bool cst = stamp && w == 0;
#define CSTAMP(k) do { if (cst) { t1 = __builtin_amdgcn_s_memtime(); seg[k] += t1 - t0; t0 = t1; } } while (0)
    if (cst) t0 = __builtin_amdgcn_s_memtime();
    for (int p = 0; p < a.phases; ++p) {
        const int type = p & 1;
        const float* dsrc = dict_ptr(p);
        const int j = l & 15, q = l >> 4;
        if (type == 0) {
            // ---- A-type: 64 atoms, bins split over the waves ----
            f32x4 acc[RB][4];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[rb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 e[2][RB][KS], d[2][4];
            auto loadA = [&](int ci, int slot) {
                int c = w + NWV * ci;
                c = c < nft ? c : nft - 1;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int k2 = 0; k2 < KS; ++k2) {
                        e[slot][rb][k2] = f32x4{1.f, 1.f, 1.f, 1.f};
                        if (ex && p > 0) e[slot][rb][k2] = ld4_sc1(rrs, (((unsigned)(k2 * RB + rb) * nft + c) * 256 + l * 4) * 4);
                    }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    d[slot][g] = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (ds) d[slot][g] = *(const f32x4*)(dsrc + ((size_t)(c * 4 + g) * 256 + l * 4));
                }
            };
            loadA(0, 0);
            if (ds) {
#pragma unroll
                for (int g = 0; g < 4; ++g) d[0][g] = dpre[g];
            }
            auto stepA = [&](int ci, auto s0tag) {
                constexpr int s0 = decltype(s0tag)::value;
                if (ci + 1 < perA) loadA(ci + 1, s0 ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                const bool ok = w + NWV * ci < nft;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    f32x4 r4 = e[s0][rb][0];
#pragma unroll
                    for (int k2 = 1; k2 < KS; ++k2) r4 += e[s0][rb][k2];
                    if (ex && p > 0 && ok) {
                        const int c = w + NWV * ci;
#pragma unroll
                        for (int k2 = 0; k2 < KS; ++k2)
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc)
                                bad += e[s0][rb][k2][cc] != val_of((unsigned)p - 1, ((unsigned)(k2 * RB + rb) * nft + c) * 256 + l * 4 + cc);
                    }
                    if (!ok) r4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (mm) {
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int t = 0; t < 4; ++t) acc[rb][t] = mfma16(r4[s], d[s0][t][s], acc[rb][t]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            for (int ci = 0; ci < perA; ci += 2) {
                stepA(ci, std::integral_constant<int, 0>{});
                if (ci + 1 < perA) stepA(ci + 1, std::integral_constant<int, 1>{});
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int v = 0; v < 4; ++v) lds[((w * RB + rb) * 16 + 4 * q + v) * 65 + 16 * t + j] = acc[rb][t][v];
        } else {
            // ---- B-type: 16*CB bins, the range's atoms split over the waves ----
            f32x4 acc[RB][CB][2];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[rb][cb][0] = acc[rb][cb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            constexpr int PF = (RB * CB >= 4) ? 2 : 3;
            f32x4 e[PF + 1][RB], d[PF + 1][CB];
            auto loadB = [&](int ci, int slot) {
                int c = w + NWV * ci;
                c = c < nchB ? c : nchB - 1;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    e[slot][rb] = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (ex && p > 0) e[slot][rb] = ld4_sc1(hrs, (((unsigned)rb * NAC + ks * nchB + c) * 256 + l * 4) * 4);
                }
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    d[slot][cb] = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (ds) d[slot][cb] = *(const f32x4*)(dsrc + ((size_t)(c * CB + cb) * 256 + l * 4));
                }
            };
#pragma unroll
            for (int g = 0; g < PF; ++g) loadB(g, g);
            if (ds) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) d[0][cb] = dpre[cb];
            }
            for (int base = 0; base < perB; base += PF + 1) {
#pragma unroll
                for (int g = 0; g < PF + 1; ++g) {
                    const int ci = base + g;
                    loadB(ci + PF, (g + PF) % (PF + 1));
                    __builtin_amdgcn_sched_barrier(0);
                    const bool ok = ci < perB && w + NWV * ci < nchB;
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        f32x4 a4 = e[g][rb];
                        if (ex && p > 0 && ok) {
                            const int c = w + NWV * ci;
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc)
                                bad += a4[cc] != val_of((unsigned)p - 1, ((unsigned)rb * NAC + ks * nchB + c) * 256 + l * 4 + cc);
                        }
                        if (!ok) a4 = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (mm) {
#pragma unroll
                            for (int s = 0; s < 4; ++s)
#pragma unroll
                                for (int cb = 0; cb < CB; ++cb) acc[rb][cb][s & 1] = mfma16(a4[s], d[g][cb][s], acc[rb][cb][s & 1]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        lds[((w * RB + rb) * 16 + 4 * q + v) * 65 + 16 * cb + j] = acc[rb][cb][0][v] + acc[rb][cb][1][v];
        }
        CSTAMP(0);                                             // loads + MFMAs + LDS write
        __syncthreads();                                       // (1)
        CSTAMP(1);
        // ---- cross-wave reduction + publish (first 256 threads) ----
        if (tid < 256 && !(a.features & 8)) {
            const int er = tid >> 4, ec = tid & 15;
            const int ncol = type ? CB : 4;                    // 16-column blocks of this workgroup's tile
            for (int rb = 0; rb < RB; ++rb)
                for (int cbk = 0; cbk < ncol; ++cbk) {
                    float s = 0.f;
#pragma unroll
                    for (int ww = 0; ww < NWV; ++ww) s += lds[((ww * RB + rb) * 16 + er) * 65 + 16 * cbk + ec];
                    const float z = s * 0.f;                   // (finite sums: a true dependency, value 0)
                    if (ex) {
                        if (type == 0) {                       // h block (rb, 4i + cbk)
                            const unsigned idx = ((unsigned)rb * NAC + 4 * i + cbk) * 256 + tid;
                            H[idx] = val_of((unsigned)p, idx) + z;
                        } else {                               // residual partial block (ks, rb, CB*bb + cbk)
                            const unsigned idx = ((unsigned)(ks * RB + rb) * nft + CB * bb + cbk) * 256 + tid;
                            R[idx] = val_of((unsigned)p, idx) + z;
                        }
                    } else if (z != 0.f) bad++;
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CSTAMP(2);                                             // reduce + stores acknowledged
        __syncthreads();                                       // (2)
        CSTAMP(3);
        prefetch(p + 1);
        CSTAMP(4);
        __syncthreads();                                       // (3)
        CSTAMP(5);                                             // waiting for the chain's barrier
    }
    if (cst && l == 0)
        for (int k = 0; k < 8; ++k) a.tl[k] = seg[k];
    if (bad) atomicAdd(&a.errors[0], bad);
}

template <int RB, int KS, int CB>
static float run(const Args& a, hipStream_t st) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipMemsetAsync(a.sync, 0, 8 * LINE * 4, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((chain_kernel<RB, KS, CB>), dim3(8 * WPC), dim3(64 * (NWV + 1)), 0, st, a);
        CK(hipGetLastError());
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    Args a;
    a.chains = argc > 1 ? atoi(argv[1]) : 8;
    const int RB = argc > 2 ? atoi(argv[2]) : 2;
    a.F = argc > 3 ? atoi(argv[3]) : 512;
    const int KS = argc > 4 ? atoi(argv[4]) : 1;
    const int CB = argc > 5 ? atoi(argv[5]) : 1;
    a.K = argc > 6 ? atoi(argv[6]) : 25;
    a.features = argc > 7 ? atoi(argv[7]) : 7;
    a.phases = argc > 8 ? atoi(argv[8]) : 2000;
    if ((a.F / 16 / CB) * KS != WPC) { printf("(F/16/CB)*KS must be %d\n", WPC); return 1; }
    CK(hipMalloc(&a.H, (size_t)8 * RB * 16 * NP * 4));
    CK(hipMalloc(&a.R, (size_t)8 * KS * RB * 16 * a.F * 4));
    CK(hipMemset(a.H, 0, (size_t)8 * RB * 16 * NP * 4));
    CK(hipMemset(a.R, 0, (size_t)8 * KS * RB * 16 * a.F * 4));
    const size_t dbytes = (size_t)a.K * 2 * WPC * a.F * 64 * 4;
    float* D; CK(hipMalloc(&D, dbytes));
    {
        std::vector<float> ones(dbytes / 4, 1.0f);
        CK(hipMemcpy(D, ones.data(), dbytes, hipMemcpyHostToDevice));
    }
    a.D = D;
    CK(hipMalloc(&a.sync, 8 * LINE * 4));
    CK(hipMalloc(&a.errors, 64));
    CK(hipMemset(a.errors, 0, 64));
    CK(hipMalloc(&a.tl, 16 * 8));
    CK(hipMemset(a.tl, 0, 16 * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    float ms = -1.f;
#define CASE(rb, ks, cb) if (RB == rb && KS == ks && CB == cb) ms = run<rb, ks, cb>(a, st)
    CASE(1, 1, 1); CASE(2, 1, 1); CASE(1, 2, 2); CASE(2, 2, 2); CASE(1, 2, 1); CASE(2, 2, 1); CASE(1, 4, 2); CASE(2, 4, 2);
    if (ms < 0) { printf("combination not instantiated\n"); return 1; }
    unsigned err[2];
    CK(hipMemcpy(err, a.errors, 8, hipMemcpyDeviceToHost));
    const double us = ms * 1e3 / a.phases;
    const double flop = 2.0 * a.chains * RB * 16 * a.F * 2016.0;          // per phase, the real atoms
    printf("chains %d x %d wgs (one XCD each), rows/chain %d (B = %d), F %d, KS %d, CB %d, K %d (%.0f MB of dictionaries),"
           " features %d%s%s%s: %.3f us per phase = %.2f us per layer-step, %.1f TFLOP/s = %.1f %% of fp32-MFMA peak;"
           " wrong words %u, timeouts %u\n",
           a.chains, WPC, RB * 16, a.chains * RB * 16, a.F, KS, CB, a.K, dbytes / 1e6, a.features,
           (a.features & 1) ? " exch" : "", (a.features & 2) ? " mfma" : "", (a.features & 4) ? " dict" : "",
           us, 2 * us, flop / us / 1e6, flop / us / 1e6 / 157.3 * 100, err[0], err[1]);
    if (a.features & 16) {
        unsigned long long tl[16];
        CK(hipMemcpy(tl, a.tl, sizeof(tl), hipMemcpyDeviceToHost));
        const double ph = (double)a.phases * 2400.0;          // s_memtime ticks ~ shader clocks ~ 2.4 GHz
        printf("  timeline (us per phase, chain 0 workgroup 0): compute wave 0: work %.2f, barrier(1) %.2f, reduce+ack %.2f,"
               " barrier(2) %.2f, prefetch %.2f, barrier(3) %.2f | sync wave: wait for compute %.2f, arrival atomic %.2f,"
               " poll %.2f (%.1f polls), barrier(3) %.2f\n", tl[0] / ph, tl[1] / ph, tl[2] / ph, tl[3] / ph, tl[4] / ph,
               tl[5] / ph, tl[8] / ph, tl[9] / ph, tl[10] / ph, (double)tl[14] / a.phases, tl[11] / ph);
    }
    return (err[0] || err[1]) ? 2 : 0;
}
