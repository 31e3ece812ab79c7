// Persistent Gram-form chains for small dictionaries: the whole (frame, layer) chain of the forward
// pass (a block of frames) or of the BPTT's sequential pass (all T frames) in ONE launch.
//
// Such shapes -- single-utterance / small-batch serving (BASELINE configs[0]) and the shipped r = 100
// training configuration (params_unfolded_snmf_*.yaml: N = 200, K = 2 | 5, batch 32) -- put a handful
// of workgroups on the chip per layer-step and are nothing but launch latency (3.6-4 us per
// layer-step from a hipGraph).  Batch rows never interact (custom_layers.py:337-338, 346-348), so
// every 16-row tile is an INDEPENDENT chain: its numO <= 32 workgroups (one per 16-atom output tile)
// are dealt to ONE XCD (workgroup ids go round the 8 XCDs: chain m takes the ids = m mod 8 of a grid
// of 8 x numO; chains 8 .. 15 -- the reference predicts in slabs of 250 utterances, enhance.py:1189 --
// take a second round of ids and share the XCDs with the first while both fit its 32 CUs),
// synchronise among themselves only, and exchange their activations through that XCD's L2.  tools/probes/xcd_local_probe.hip (profiles/r03b_xcd_local_probe.txt): barrier 0.5-0.6 us, an
// exchange phase 1.33 us with plain producer stores + L1-bypassing loads (1.62 with write-through
// stores), against 2.4 us with a launch per phase; the same structure over the whole chip does NOT
// pay (profiles/r03a_chain_probe.txt).
//
// Protocol (correct on ANY placement, fast when a chain shares an XCD):
//   * what one workgroup writes and another reads inside the launch goes out through `st_x`: a
//     write-through (sc1) store, or -- once the chain has established that all its workgroups report
//     the same HW_REG_XCC_ID -- a plain store, which stays in the shared L2; the store's vmcnt
//     acknowledgement means it reached that L2.  Consumers always read with sc1 (L1-bypassing) loads.
//   * barrier = one monotonic agent-scope counter per chain.  A ninth wave of every workgroup does
//     nothing but arrive and poll (it has no other loads in flight: a poll issued behind a wave's
//     operand prefetch returns only after the prefetch, in order), the eight others request the next
//     phase's PRIVATE operands (G tile, c_k[t], 1/alpha, validity, stored hiddens) meanwhile.
//   * a chain whose poll exceeds PERSIST_SPIN_LIMIT raises its abort word: every workgroup of the
//     chain leaves at its next barrier and the host-mapped fault word of the handle is set: drnmf_check_status
//     / drnmf_status_take_device report DRNMF_ERR_TIMEOUT instead of a wrong result going out silently.
// Same arithmetic, in the same order, as gram_fwd_kernel / bwd_edge_kernel + gram_bwd_kernel: results
// are bit-identical to the launch-per-layer-step form (tests/test_gpu_dp.py).
#pragma once
#include "cell_gram.h"

namespace {

constexpr int PERSIST_MAX_TILES = 32;           // output tiles per chain (N <= 512)
constexpr int PERSIST_MAX_CHAINS = 16;          // row tiles (B <= 256): one XCD each, a second round sharing them
constexpr unsigned PERSIST_SPIN_LIMIT = 1u << 21;   // polls (~0.5 us each) before a chain gives up
constexpr int PERSIST_LINE_WORDS = 1024;        // a chain's sync words {arrivals, abort, XCC mask} sit 4 KB apart
constexpr int PERSIST_SYNC_BYTES = 16 * PERSIST_LINE_WORDS * 4;   // 16 chains

__device__ __forceinline__ float ld1_sc1(const float* p) {
    const unsigned u = __hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__device__ __forceinline__ void st1_sc1(float* p, float v) {
    unsigned u;
    memcpy(&u, &v, 4);
    __hip_atomic_store((unsigned*)p, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// exchanged store: plain once the chain is known to share one L2, write-through otherwise
__device__ __forceinline__ void st_x(float* p, float v, bool same_xcd) {
    // (an ordinary store: `volatile` would be lowered to sc0 sc1, i.e. write-through again; the
    // "memory"-clobbering s_waitcnt in front of every grid barrier keeps it from sinking past it)
    if (same_xcd) *p = v;
    else st1_sc1(p, v);
}
__device__ __forceinline__ f32x4 ld4_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16);   // aux 16 = sc1
    f32x4 v;
    memcpy(&v, &raw, 16);
    return v;
}
__device__ __forceinline__ void wg_sync() { __syncthreads(); }

// Measurement aid (-DDRNMF_TIMELINE builds only): s_memtime segment sums of wave 0 of workgroup
// (chain 0, tile 0) over all phases of the LAST persistent launch; read with drnmf_debug_persist_timeline.
#ifdef DRNMF_TIMELINE
__device__ unsigned long long g_ptl[2][16];
#define PTL_DECL unsigned long long ptl_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ptl_t0_ = 0, ptl_t1_; \
                 const bool ptl_on_ = blockIdx.x == 0 && threadIdx.x < 64
#define PTL_START do { if (ptl_on_) ptl_t0_ = __builtin_amdgcn_s_memtime(); } while (0)
#define PTL(k) do { if (ptl_on_) { ptl_t1_ = __builtin_amdgcn_s_memtime(); ptl_[k] += ptl_t1_ - ptl_t0_; ptl_t0_ = ptl_t1_; } } while (0)
#define PTL_WAIT do { if (ptl_on_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#define PTL_END(kid, nph) do { if (ptl_on_ && threadIdx.x == 0) { for (int i_ = 0; i_ < 10; ++i_) g_ptl[kid][i_] = ptl_[i_]; g_ptl[kid][15] = (unsigned long long)(nph); } } while (0)
#else
#define PTL_DECL do { } while (0)
#define PTL_START do { } while (0)
#define PTL(k) do { } while (0)
#define PTL_WAIT do { } while (0)
#define PTL_END(kid, nph) do { } while (0)
#endif

// The ninth wave's side of one grid barrier of chain `line`: arrive, poll {arrivals, abort} with ONE
// 8-byte load per poll, publish the verdict in LDS.  ctl[0] = abort, ctl[1] = chain shares one XCD.
__device__ __forceinline__ void persist_arrive_and_wait(unsigned* line, unsigned target, int* ctl,
                                                        unsigned* host_flag, bool learn_xcd) {
    if ((threadIdx.x & 63) == 0) {
        __hip_atomic_fetch_add(line, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int abort = 0;
        for (;;) {
            const unsigned long long v = __hip_atomic_load((unsigned long long*)line, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(v >> 32) != 0u) { abort = 1; break; }
            if ((unsigned)v >= target) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > PERSIST_SPIN_LIMIT) {
                // (never reached when the chain's <= 32 workgroups are resident together, which the
                // host checks before taking this path; should it be, say so instead of hanging)
                __hip_atomic_store(line + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (host_flag) __hip_atomic_store(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                abort = 1;
                break;
            }
        }
        ctl[0] = abort;
        if (learn_xcd) {
            const unsigned mask = __hip_atomic_load(line + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ctl[1] = __builtin_popcount(mask) == 1;
        }
    }
}
__device__ __forceinline__ void persist_census(unsigned* line) {
    if ((threadIdx.x & 63) == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
        __hip_atomic_fetch_or(line + 2, 1u << (x & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (acknowledged before this workgroup's first arrival: whoever sees the barrier complete
        // sees every participant's bit)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// ---------------------------------------------------------------------------------------------
// forward: frames t0 .. t0+nfr-1, K-1 phases per frame (layer 0 rides in the first, gram_fwd_kernel)
struct GramPersistArgs {
    const float* G;          // packed G of layer 0's slot; layer k at G + k * g_stride (0: tied)
    size_t g_stride;
    const float* ia;         // [K][Np]
    const float* Cp;
    float* hb[2];
    float* qb[2];
    float* state;
    float* rs_part;
    float* psum;
    float* psum_all;
    const unsigned char* valid;
    float* out;
    unsigned* bar;           // PERSIST_SYNC_BYTES, zeroed before the launch
    unsigned* host_flag;     // host-mapped word of the handle (timeout report)
    float u0d, u0o, uko;
    int B, T, N, K, Bp, Np, numO, numM, out_width, all_hidden;
    int t0, nfr;
    int cp_mask;             // as GramFwdArgs
    int nwait;               // arrivals a barrier waits for: numO (numO + 1 under DRNMF_PERSIST_FAULT=1, the
                             // fault injection of tests/test_gpu_dp.py: the chains then time out)
};

// NS = chunk slots any wave owns = ceil(Np / 16 / 8) (workgroup-uniform, 1..4): the slots beyond would
// only add 0 * G to the accumulators (N = 200: 2 of 4 slots, i.e. half the matrix-pipe time), and a
// compile-time count keeps the operand loads and MFMAs one straight line.
template <int NS>
__global__ void __launch_bounds__(64 * (NW_G + 1)) gram_persist_kernel(const GramPersistArgs a) {
    // chain = row tile m, dealt to XCD m % 8; a second round of chains (m >= 8: the reference's
    // 250-utterance inference slabs) shares the XCDs with the first
    const int slot = (int)(blockIdx.x >> 3) / a.numO;
    const int m = (blockIdx.x & 7) + 8 * slot;
    if (m >= a.numM) return;
    __shared__ __attribute__((aligned(16))) float red[NW_G * 16 * 17];
    __shared__ float part[32][17];
    __shared__ float ps16[16], psv[16];
    __shared__ int ctl[2];
    // this thread's previous OUTPUT per layer ([K][256] with all_hidden, else [256]; dynamic): a masked
    // step repeats it (K.rnn), and re-reading it from the output tensor would put a global-memory
    // round trip into the epilogue of every phase of a frame in which any of the 16 rows is masked
    // (ragged batches: measured 2.25 against 1.65 us per phase)
    extern __shared__ float oprev[];
    const int ot = (int)(blockIdx.x >> 3) - slot * a.numO;  // output tile (grid = 8 * numO * rounds)
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KL = a.K - 1, nphase = a.nfr * KL;
    unsigned* line = a.bar + PERSIST_LINE_WORDS * m;
    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; }
    if (w == NW_G) {
        // ---- the synchronising wave: as many workgroup barriers per phase as the others ------------
        persist_census(line);
        wg_sync();                                          // (ctl initialised)
        int k = 1;
        for (int p = 0; p < nphase; ++p) {
            if (k == 1) { wg_sync(); wg_sync(); }           // row-sum reduction of a frame's first phase
            wg_sync();                                      // cross-wave reduction
            wg_sync();                                      // (1) the phase's stores are acknowledged
            persist_arrive_and_wait(line, (unsigned)(p + 1) * (unsigned)a.nwait, ctl, a.host_flag, p == 0);
            wg_sync();                                      // (2) release
            if (ctl[0]) return;
            k = (k == a.K - 1) ? 1 : k + 1;
        }
        return;
    }
    wg_sync();
    const int l = tid & 63, j = l & 15, q = l >> 4;
    const int NAC = a.Np / 16;
    const bool ethr = tid < 256;
    const int erow = (tid & 255) >> 4, ecol = tid & 15;
    const int rg = m * 16 + erow, n = ot * 16 + ecol;
    const size_t hoff = ((size_t)m * NAC + ot) * 256 + hp_pos(erow, ecol);
    const size_t cstride = (size_t)a.Bp * a.Np;
    const int clast = NAC - 1;
    const int per_wave = (NAC - w + NW_G - 1) / NW_G;        // <= 4 (NAC <= 32)
    const unsigned abytes = (unsigned)(cstride * 4);
    float* const hb0 = a.hb[0];
    float* const hb1 = a.hb[1];
    float* const qb0 = a.qb[0];
    float* const qb1 = a.qb[1];
    // this thread's own addresses, formed once (vector registers: the scalar file is full and a pointer
    // re-fetched from the kernarg segment inside the epilogue is a scalar-load round trip per phase)
    float* const p_hb0 = hb0 + hoff;
    float* const p_hb1 = hb1 + hoff;
    float* const p_qb0 = qb0 + hoff;
    float* const p_qb1 = qb1 + hoff;
    float* const p_state = a.state + hoff;
    float* const p_rs = a.rs_part + (size_t)ot * a.Bp + rg;           // + parity * numO * Bp
    const size_t rs_par = (size_t)a.numO * a.Bp;
    float* p_out = a.out + ((size_t)rg * a.T + a.t0) * a.out_width + n;   // frame t0; + out_width per frame
    float* p_psall = a.psum_all + (size_t)a.t0 * a.Bp + rg;
    const bool out_live = rg < a.B && n < a.N;

    // Operands that no other workgroup writes inside this launch -- the G tile, c_k[t], 1/alpha, the
    // validity flag -- are requested one phase AHEAD, between the two workgroup barriers of the grid
    // barrier, so that after it only the exchanged activations (sc1, L2 hits) are on the critical
    // path.  This thread's own elements of h, q and the state never leave its registers.
    f32x4 bvN[NS];
    float ckN = 0.f, cnextN = 0.f, iavN = 0.f;
    unsigned char vldN = 1;      // (kept as loaded: a comparison here would wait for every prefetch load)
    const float* Gl = a.G + (size_t)ot * NAC * 256 + l * 4;      // + k * g_stride + 256 * c
    int gc[NS];
#pragma unroll
    for (int g = 0; g < NS; ++g) {
        const int c = w + NW_G * g;
        gc[g] = 256 * (c > clast ? clast : c);
    }
    const float* ial = a.ia + n;                                   // + k * Np
    const float* cpl = a.Cp + hoff;                                // + ((t & cp_mask) * K + k) * cstride
    const unsigned char* vl = a.valid + rg;                        // + t * Bp
    auto prefetch = [&](int fN, int kN) {                          // operands of (frame t0 + fN, layer kN)
        if (fN >= a.nfr) return;
        const int t = a.t0 + fN;
        const float* Gk = Gl + (size_t)kN * a.g_stride;
#pragma unroll
        for (int g = 0; g < NS; ++g) bvN[g] = *(const f32x4*)(Gk + gc[g]);
        if (ethr) {
            iavN = ial[(size_t)kN * a.Np];
            ckN = cpl[(size_t)((t & a.cp_mask) * a.K + kN) * cstride];
            vldN = vl[(size_t)t * a.Bp];
            if (kN == a.K - 1) {
                const int tn = t + 1 < a.T ? t + 1 : t;
                cnextN = cpl[(size_t)((tn & a.cp_mask) * a.K) * cstride];
            }
        }
    };
    prefetch(0, 1);
    // own elements: q of the block's first frame and the state (left by the previous launch / prologue)
    float hcur = 0.f, stcur = 0.f;
    if (ethr) {
        hcur = *((a.t0 & 1) ? p_qb1 : p_qb0);
        stcur = *p_state;
        const int nl = a.all_hidden ? a.K : 1;
        for (int kk = 0; kk < nl; ++kk)
            oprev[kk * 256 + tid] = (a.t0 > 0 && out_live) ? p_out[kk * a.N - (ptrdiff_t)a.out_width] : 0.f;
    }
    bool fast = false;                                             // chain shares one XCD (known after phase 0)
    PTL_DECL;
    PTL_START;

    int f = 0, k = 1;
    for (int p = 0; p < nphase; ++p) {
        const int t = a.t0 + f, par = t & 1;
        const bool first = k == 1, last = k == a.K - 1;
        const float* a_in = first ? (par ? qb1 : qb0) : (((k - 1) & 1) ? hb1 : hb0);
        // ---- exchanged operands (sc1) -----------------------------------------------------------
        __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a_in, 0, abytes, 0x00020000);
        f32x4 av[NS], bv[NS];
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            bv[g] = bvN[g];
            av[g] = ld4_sc1(arsrc, (unsigned)(((size_t)m * NAC * 256 + l * 4) * 4 + gc[g] * 4));
        }
        const float iav = iavN, ck = ckN, cnext = cnextN;
        const bool vld = vldN != 0;
        PTL(0);                                                  // exchanged loads issued
        PTL_WAIT;
        PTL(1);                                                  // wave 0's operands there
        if (first) {
            float rsum = 0.f;
            const int row = tid & 15, pt = tid >> 4;        // 32 parts
            const float* rp = a.rs_part + (size_t)par * rs_par + m * 16 + row;
            if (pt < a.numO) rsum += ld1_sc1(rp + (size_t)pt * a.Bp);      // (numO <= 32: one partial per thread)
            part[tid >> 4][tid & 15] = rsum;
            wg_sync();
            if (tid < 16) {
                float tot = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) tot += part[i][tid];
                ps16[tid] = a.u0o * tot;
                psv[tid] = tot;
            }
            wg_sync();
        }
        // ---- contraction (as gram_contract: waves split the input atoms, fixed-order reduce) ------
        const float addv = first ? ps16[j] : 0.f;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            const bool ok = g < per_wave;
            f32x4 a4 = av[g];
            if (first) {
#pragma unroll
                for (int e = 0; e < 4; ++e) a4[e] = fmaxf(a4[e] + addv, 0.f);
            }
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
                const float a1 = ok ? a4[sI] : 0.f;
                if (sI & 1) acc1 = mfma16(a1, bv[g][sI], acc1);
                else acc0 = mfma16(a1, bv[g][sI], acc0);
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) red[(w * 16 + 4 * q + v) * 17 + j] = acc0[v] + acc1[v];
        PTL(2);                                                  // (row sums,) MFMAs, LDS write
        wg_sync();
        PTL(3);                                                  // slowest wave's partials in LDS
        if (ethr) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW_G; ++ww) s += red[(ww * 16 + erow) * 17 + ecol];
            // ---- update epilogue (as gram_fwd_kernel) -----------------------------------------
            const float ps = psv[erow];                     // sum(p) of this frame (LDS since its first phase)
            if (first && ot == 0 && ecol == 0) {
                a.psum[rg] = ps;
                *p_psall = ps;
            }
            float hprev = hcur;
            if (first) hprev = fmaxf(hprev + a.u0o * ps, 0.f);
            const float pre = hprev - s * iav + ck + a.uko * ps;
            const float hn = fmaxf(pre, 0.f);
            if (out_live) {
                // K.rnn masking: a masked step repeats the previous output (zeros before the first valid)
                if (a.all_hidden) {
                    if (first) {
                        const float o0 = vld ? hprev : oprev[tid];
                        oprev[tid] = o0;
                        p_out[0] = o0;
                    }
                    const float o = vld ? hn : oprev[k * 256 + tid];
                    oprev[k * 256 + tid] = o;
                    p_out[k * a.N] = o;
                } else if (last) {
                    const float o = vld ? hn : oprev[tid];
                    oprev[tid] = o;
                    p_out[0] = o;
                }
            }
            if (last) {
                const float st = vld ? hn : stcur;                // a masked step keeps the state
                stcur = st;
                *p_state = st;
                const float rs = row16_sum(st);
                if (ecol == 0) st_x(p_rs + (size_t)(par ^ 1) * rs_par, rs, fast);
                hcur = (a.u0d - a.u0o) * st + cnext;              // q of the next frame
                st_x(par ? p_qb0 : p_qb1, hcur, fast);
                p_out += a.out_width;
                p_psall += a.Bp;
            } else {
                hcur = hn;
                st_x((k & 1) ? p_hb1 : p_hb0, hn, fast);
            }
        }
        // ---- grid barrier of this chain ---------------------------------------------------------------
        PTL(4);                                                  // reduce, update, stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PTL(5);                                                  // stores acknowledged
        wg_sync();                                               // (1)
        PTL(6);
        if (last) { ++f; k = 1; } else ++k;                      // the next phase
        prefetch(f, k);
        PTL(7);                                                  // next phase's private operands requested
        wg_sync();                                               // (2)
        PTL(8);                                                  // arrival + poll of the ninth wave
        if (ctl[0]) return;
        if (p == 0) fast = ctl[1] != 0;
    }
    PTL_END(0, nphase);
}

// ---------------------------------------------------------------------------------------------
// BPTT sequential pass: frames T-1 .. 0, K phases per frame -- the edge (bwd_edge_kernel: gradient
// w.r.t. the state that entered frame t+1, top of frame t) and the K-1 layer-steps k = K-1 .. 1
// (gram_bwd_kernel) -- then the gradient w.r.t. the initial state.  dz_k, the pending output gradient
// g, d state and the per-row accumulators live in registers; exchanged: dG_k (the contraction's
// operand, ping-pong) and the row-sum partials of dz_0 / uko sum_k dz_k at the frame boundary.
struct GramPersistBwdArgs {
    const float* G;
    size_t g_stride;
    const float* ia;         // [K][Np]
    const float* hall;       // [B][T][K*N]
    const float* d_out;      // [B][T][N]
    float* dz_all;           // [B][T][K*N]
    float* dGp[2];           // packed dG ping-pong (layer k in buffer k & 1)
    float* z0s_part;         // [2][numO][Bp] row sums of dz_0 per output tile, by frame-counter parity
    float* dps_part;         // [2][numO][Bp] uko * sum_{k>=1} row sums of dz_k
    float* dh0_part;         // [numM][Np]
    float* dstate;           // packed, final value (gradient w.r.t. the initial state per row)
    const unsigned char* valid;
    unsigned* bar;
    unsigned* host_flag;
    float u0d, u0o, uko;
    int B, T, N, K, Bp, Np, numO, numM;
    int nwait;               // as GramPersistArgs
};

template <int NS>
__global__ void __launch_bounds__(64 * (NW_G + 1)) gram_persist_bwd_kernel(const GramPersistBwdArgs a) {
    const int slot = (int)(blockIdx.x >> 3) / a.numO;       // (see gram_persist_kernel)
    const int m = (blockIdx.x & 7) + 8 * slot;
    if (m >= a.numM) return;
    __shared__ __attribute__((aligned(16))) float red[NW_G * 16 * 17];
    __shared__ float sm[16][17];
    __shared__ int ctl[2];
    const int ot = (int)(blockIdx.x >> 3) - slot * a.numO;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, nphase = a.T * K;
    unsigned* line = a.bar + PERSIST_LINE_WORDS * m;
    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; }
    if (w == NW_G) {
        persist_census(line);
        wg_sync();
        int k = K;                                          // K = the edge phase of a frame
        for (int p = 0; p < nphase; ++p) {
            if (k != K) wg_sync();                          // cross-wave reduction of a contraction phase
            wg_sync();                                      // (1)
            persist_arrive_and_wait(line, (unsigned)(p + 1) * (unsigned)a.nwait, ctl, a.host_flag, p == 0);
            wg_sync();                                      // (2)
            if (ctl[0]) return;
            k = (k == 1) ? K : k - 1;
        }
        wg_sync();                                          // column sums of d state (after frame 0)
        return;
    }
    wg_sync();
    const int l = tid & 63, j = l & 15, q = l >> 4;
    const int NAC = a.Np / 16;
    const bool ethr = tid < 256;
    const int erow = (tid & 255) >> 4, ecol = tid & 15;
    const int rg = m * 16 + erow, n = ot * 16 + ecol;
    const size_t hoff = ((size_t)m * NAC + ot) * 256 + hp_pos(erow, ecol);
    const int clast = NAC - 1;
    const int per_wave = (NAC - w + NW_G - 1) / NW_G;
    const unsigned abytes = (unsigned)((size_t)a.Bp * a.Np * 4);
    float* const dG0 = a.dGp[0];
    float* const dG1 = a.dGp[1];
    const int KN = K * a.N;
    const bool in = rg < a.B && n < a.N;
    const size_t pstride = (size_t)a.numO * a.Bp;

    // private operands of the NEXT phase, requested between the two workgroup barriers of the grid barrier
    f32x4 bvN[NS];
    float doutN = 0.f, htopN = 0.f, hprevN = 0.f, iapN = 0.f;
    unsigned char vtN = 0;
    const float* Gl = a.G + (size_t)ot * NAC * 256 + l * 4;
    int gc[NS];
#pragma unroll
    for (int g = 0; g < NS; ++g) {
        const int c = w + NW_G * g;
        gc[g] = 256 * (c > clast ? clast : c);
    }
    const float* hrow = a.hall + (size_t)rg * a.T * KN + n;          // + t * KN + k * N (only if `in`)
    const float* drow = a.d_out + (size_t)rg * a.T * a.N + n;        // + t * N
    auto prefetch = [&](int cN, int kN) {                            // phase kN of frame counter cN
        if (cN >= a.T) return;
        const int t = a.T - 1 - cN;
        if (kN == K) {                                               // edge: d_out[t], h_{K-1}[t], validity
            if (ethr) {
                vtN = a.valid[(size_t)t * a.Bp + rg];
                doutN = 0.f;
                htopN = 0.f;
                if (in) {
                    doutN = drow[(size_t)t * a.N];
                    htopN = hrow[(size_t)t * KN + (size_t)(K - 1) * a.N];
                }
            }
            return;
        }
        const float* Gk = Gl + (size_t)kN * a.g_stride;
#pragma unroll
        for (int g = 0; g < NS; ++g) bvN[g] = *(const f32x4*)(Gk + gc[g]);
        if (ethr) {
            iapN = a.ia[(size_t)(kN - 1) * a.Np + n];
            hprevN = in ? hrow[(size_t)t * KN + (size_t)(kN - 1) * a.N] : 0.f;
        }
    };
    prefetch(0, K);
    const float ia_last = ethr ? a.ia[(size_t)(K - 1) * a.Np + n] : 0.f;
    float ds = 0.f, g = 0.f, dz = 0.f, acc_dps = 0.f;                // (the launch version starts from a zeroed workspace)
    unsigned char vn = 0;                                            // validity of frame t+1 (= vt of the frame before)
    bool fast = false;

    // sum over the output tiles of the row-sum partials of parity `par`, in bwd_edge_kernel's order
    auto row_totals = [&](int par, float& s0, float& sp) {
        const float* z0 = a.z0s_part + par * pstride + rg;
        const float* dp_ = a.dps_part + par * pstride + rg;
        float zv[4], pv4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b2 = (tid & 15) + 16 * u;
            const int bc = b2 < a.numO ? b2 : a.numO - 1;
            zv[u] = ld1_sc1(z0 + (size_t)bc * a.Bp);
            pv4[u] = ld1_sc1(dp_ + (size_t)bc * a.Bp);
        }
        s0 = 0.f;
        sp = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = (tid & 15) + 16 * u < a.numO;
            s0 += ok ? zv[u] : 0.f;
            sp += ok ? pv4[u] : 0.f;
        }
        s0 = row16_sum(s0);                                          // (numO <= 32 < 64: no further partials)
        sp = row16_sum(sp);
    };

    int c = 0, k = K;
    for (int p = 0; p < nphase; ++p) {
        const int t = a.T - 1 - c;
        if (k == K) {
            // ---- edge of frame t (bwd_edge_kernel) ---------------------------------------------------
            if (ethr) {
                const float dout = doutN, htop = htopN;
                const unsigned char vt = vtN;
                if (c > 0) {
                    float s0, sp;
                    row_totals((c - 1) & 1, s0, sp);
                    if (vn) ds = bptt_state_grad(a.u0d, a.u0o, dz, s0, sp);      // (dz = dz_0 of frame t+1)
                }
                float dh = 0.f;
                if (vt) { dh = dout + g + ds; g = 0.f; }
                else g += dout;
                dz = htop > 0.f ? dh : 0.f;
                if (in) a.dz_all[((size_t)rg * a.T + t) * KN + (size_t)(K - 1) * a.N + n] = dz;
                st_x((((K - 1) & 1) ? dG1 : dG0) + hoff, dz * ia_last, fast);
                acc_dps = 0.f;
                vn = vt;
            }
        } else {
            // ---- layer-step k of frame t (gram_bwd_kernel) -------------------------------------------
            __amdgpu_buffer_rsrc_t arsrc =
                __builtin_amdgcn_make_buffer_rsrc((void*)((k & 1) ? dG1 : dG0), 0, abytes, 0x00020000);
            f32x4 av[NS], bv[NS];
#pragma unroll
            for (int gg = 0; gg < NS; ++gg) {
                bv[gg] = bvN[gg];
                av[gg] = ld4_sc1(arsrc, (unsigned)(((size_t)m * NAC * 256 + l * 4) * 4 + gc[gg] * 4));
            }
            const float hprev = hprevN, iap = iapN;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int gg = 0; gg < NS; ++gg) {
                const bool ok = gg < per_wave;
#pragma unroll
                for (int sI = 0; sI < 4; ++sI) {
                    const float a1 = ok ? av[gg][sI] : 0.f;
                    if (sI & 1) acc1 = mfma16(a1, bv[gg][sI], acc1);
                    else acc0 = mfma16(a1, bv[gg][sI], acc0);
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) red[(w * 16 + 4 * q + v) * 17 + j] = acc0[v] + acc1[v];
            wg_sync();
            if (ethr) {
                float s = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW_G; ++ww) s += red[(ww * 16 + erow) * 17 + ecol];
                const float dzn = hprev > 0.f ? dz - s : 0.f;
                if (in) a.dz_all[((size_t)rg * a.T + t) * KN + (size_t)(k - 1) * a.N + n] = dzn;
                const float sk = row16_sum(dz), s0 = row16_sum(dzn);
                acc_dps = fmaf(a.uko, sk, acc_dps);
                if (k > 1) {
                    st_x((((k - 1) & 1) ? dG1 : dG0) + hoff, dzn * iap, fast);
                } else if (ecol == 0) {
                    const size_t po = (size_t)(c & 1) * pstride + (size_t)ot * a.Bp + rg;
                    st_x(a.dps_part + po, acc_dps, fast);
                    st_x(a.z0s_part + po, s0, fast);
                }
                dz = dzn;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_sync();                                               // (1)
        if (k == 1) { ++c; k = K; } else --k;
        prefetch(c, k);
        wg_sync();                                               // (2)
        if (ctl[0]) return;
        if (p == 0) fast = ctl[1] != 0;
    }
    // ---- after frame 0: gradient w.r.t. the initial state of every row, summed over the tile's rows ----
    if (ethr) {
        float s0, sp;
        row_totals((a.T - 1) & 1, s0, sp);
        if (vn) ds = bptt_state_grad(a.u0d, a.u0o, dz, s0, sp);
        a.dstate[hoff] = ds;
        sm[erow][ecol] = rg < a.B ? ds : 0.f;
    }
    wg_sync();
    if (tid < 16) {
        float s = 0.f;
        for (int r2 = 0; r2 < 16; ++r2) s += sm[r2][tid];
        a.dh0_part[(size_t)m * a.Np + ot * 16 + tid] = s;
    }
}

// Whether a Gram-form call may run as persistent chains: tile counts, and every chain's workgroups
// resident together on one XCD's 32 CUs (checked once per process against the occupancy API with a
// margin of one workgroup per CU -- the API can answer one too many, MI355X_MICROARCH.md).
static inline int persist_rounds(int numM) { return (numM + 7) / 8; }
// test aid: DRNMF_PERSIST_FAULT=1 makes every barrier wait for one arrival that never comes
static inline int persist_nwait(int numO) {
    const char* e = tune_env("DRNMF_PERSIST_FAULT");
    return numO + ((e && atoi(e) == 1) ? 1 : 0);
}
constexpr int PERSIST_MAX_K = 48;               // [K][256] floats of previous outputs in LDS
static inline size_t persist_fwd_lds(int K, bool all_hidden) { return (size_t)(all_hidden ? K : 1) * 256 * 4; }
static inline bool persist_shape_ok(drnmf_handle_t h, int numM, int numO, int K) {
    if (K < 2 || K > PERSIST_MAX_K || numO > PERSIST_MAX_TILES || numM > PERSIST_MAX_CHAINS) return false;
    if (persist_rounds(numM) * numO > PERSIST_MAX_TILES) return false;   // an XCD's chains fit its 32 CUs
    if (const char* e = tune_env("DRNMF_PERSIST"))
        if (atoi(e) == 0) return false;
    // one workgroup per CU suffices where chain m really lands on XCD m's 32 CUs (the whole MI355X as one
    // device); on a smaller partition (fewer CUs than 8 x 32) every participant of every chain must
    // still fit the device at once.  Occupancy and CU count belong to the handle's device
    // (drnmf_create -> persist_query_occupancy): handles on different devices / partitions differ.
    // (persist_lock_fd: this handle owns the device's cross-process admission, common.h)
    return h->persist_lock_fd >= 0 && h->persist_per_cu >= 1 &&
           numM * numO <= h->persist_n_cu * h->persist_per_cu;
}
// (defined once, in cell_forward.hip)
static inline void persist_query_occupancy_impl(int device, int* per_cu, int* n_cu) {
    int a = 0, b = 0, prev = -1;
    *per_cu = 0;
    *n_cu = 0;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return; }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, gram_persist_kernel<4>, 64 * (NW_G + 1), persist_fwd_lds(PERSIST_MAX_K, true)) != hipSuccess) a = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, gram_persist_bwd_kernel<4>, 64 * (NW_G + 1), 0) != hipSuccess) b = 0;
    *per_cu = a < b ? a : b;
    if (hipDeviceGetAttribute(n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) *n_cu = 0;
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    (void)hipGetLastError();
}

void* pick_persist_fwd(int NAC) {
    switch ((NAC + NW_G - 1) / NW_G) {
        case 1: return (void*)&gram_persist_kernel<1>;
        case 2: return (void*)&gram_persist_kernel<2>;
        case 3: return (void*)&gram_persist_kernel<3>;
        default: return (void*)&gram_persist_kernel<4>;
    }
}
void* pick_persist_bwd(int NAC) {
    switch ((NAC + NW_G - 1) / NW_G) {
        case 1: return (void*)&gram_persist_bwd_kernel<1>;
        case 2: return (void*)&gram_persist_bwd_kernel<2>;
        case 3: return (void*)&gram_persist_bwd_kernel<3>;
        default: return (void*)&gram_persist_bwd_kernel<4>;
    }
}

}  // namespace
