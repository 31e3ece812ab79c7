// Backward pass (BPTT) of the recurrent DR-NMF cell on gfx950.
//
// The reference obtains this by Theano autodiff of the scan (enhance.py:1071-1073, 1152); here it
// is written out.  With dz_k = dh_k * [h_k > 0] and dG_k = dz_k * ia_k (forward: cell_forward.hip):
//     d r_k     = dG_k Dn_k^T                         (contract atoms   -> cell_b_kernel, no x)
//     dh_{k-1}  = dz_k - d r_k Dn_k                   (contract bins    -> bwd_a_kernel)
//     layer 0   : dp = u0d dz_0 + u0o (sum(dz_0) - dz_0) + uko sum_{k>=1} sum(dz_k)
// so the sequential chain has the forward's shape (two skinny launches per layer-step, replayed
// as a per-frame hipGraph in reverse time), and only dz / d r are stored per (t, k).  Everything
// that contracts over frames is time-batched afterwards and off the critical chain:
//     d Dn_k = (R_k^T dz_k) * ia_k - dR_k^T H_{k-1},   R_k = X - H_{k-1} Dn_k^T   (gemm_tn.h, split-K)
//     d b_k = sum dz_k,  d ia_k = sum dz_k * G_k  (G_k recovered from the stored hiddens)
// followed by the parameter maps: d log_D = Dn * (dDn - Dn * colsum(dDn * Dn)), log_alph, log_lam1,
// log_h0.  Masked steps follow K.rnn: no gradient enters a masked frame; the gradient of its
// (repeated) output is handed to the last valid frame and the state gradient passes through.
#include "cell_shared.h"
#include "cell_gram.h"
#include "cell_gram_persist.h"
#include "gemm_nt.h"
#include "gemm_tn.h"

namespace {

constexpr int TN_SPLITS = 64;   // upper bound; the count per shape (and the partial buffers' size): tn_splits()

// Split-K count of the time-batched weight-gradient GEMMs: tiles x splits workgroups run in rounds
// of 512 (2 per CU); pick the count that minimises rounds / splits (8 splits at the C2 shape made
// 640 workgroups = 2 rounds, 6 make 480 = 1 round: 3.8 -> 2.6 ms per GEMM).
static int tn_splits(int M, int N, int64_t Kdim) { return gemm_tn::pick_splits(M, N, Kdim, TN_SPLITS); }
constexpr int CR_SPLITS = 256;

struct EdgeArgs {
    const float* hall;       // [B][T][K*N]
    const float* d_out;      // [B][T][N]
    float* dz_all;           // [B][T][K*N]
    const float* ia_last;    // [Np] of layer K-1
    float* dstate;           // packed [Bp][Np]
    float* gq;               // packed
    float* dzp_top;          // packed dz buffer of layer K-1
    float* dGp_top;
    const float* dzp0;       // packed dz buffer of layer 0
    float* dz0s_part;        // [2][numA][Bp]
    float* dps_part;         // [2][numA][Bp]
    float* dh0_part;         // [numM][Np]
    const unsigned char* valid;
    const int* c_rd;
    int* c_wr;
    float u0d, u0o;
    int B, T, N, K, Bp, Np, numA;
    // odd bins (kept out of the MFMA tiles as in the forward): d r_tail of layer K-1 is the sum over
    // atom blocks of these partial dots of dG with the tail rows of the dictionary
    const float* Dtail_top;  // [MAX_TAIL][Np] of layer K-1
    float* dq_out;           // [MAX_TAIL][Bp][numA]
    int ntail;
    int nparts, ppb;         // row-sum partials per row: numA (1 per 32-atom block), or in the Gram
                             // form one per 16-atom output tile (2 per block)
};

__global__ void __launch_bounds__(256) bwd_edge_kernel(const EdgeArgs a) {
    __shared__ float sm[ROWS][ATOMS + 1];
    const int m = blockIdx.x >> 3;                          // grid layout: see cell_a_kernel
    const int ab = blockIdx.y * 8 + (blockIdx.x & 7);
    if (ab >= a.numA) return;
    const int tid = threadIdx.x;
    const int c = *a.c_rd;
    if (a.c_wr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.c_wr = c;
    const int t = a.T - 1 - c;
    const int NAC = a.Np / 16, ac0 = ab * 2;
    const int erow = tid >> 4, ec = (tid & 15) * 2;
    const int rg = m * ROWS + erow, n = ab * ATOMS + ec;
    const size_t hoff = ((size_t)m * NAC + ac0 + (ec >> 4)) * 256 + hp_pos(erow, ec & 15);
    const int KN = a.K * a.N;
    const size_t pstride = (size_t)a.nparts * a.Bp;

    // Every load of this kernel depends only on the frame counter: all of them are requested here,
    // in one round trip (the two parts below used to pay theirs one after the other, and the
    // partial-sum loop one per 16 partials: 6.8 us per launch for an elementwise kernel).
    f32x2 ds = *(const f32x2*)(a.dstate + hoff);
    const bool top = t >= 0;
    const bool lrow = rg < a.B;
    const bool pair = lrow && n + 1 < a.N && (a.N & 1) == 0;      // 8-byte accesses (n is even)
    f32x2 g = {0.f, 0.f}, dout2 = {0.f, 0.f}, h2 = {0.f, 0.f};
    unsigned char vt = 0;
    const size_t ro = top ? ((size_t)rg * a.T + t) : 0;
    f32x2 ia = {0.f, 0.f}, dtl[MAX_TAIL];
#pragma unroll
    for (int i = 0; i < MAX_TAIL; ++i) dtl[i] = f32x2{0.f, 0.f};
    if (top) {
        vt = a.valid[(size_t)t * a.Bp + rg];
        g = *(const f32x2*)(a.gq + hoff);
        ia = *(const f32x2*)(a.ia_last + n);
#pragma unroll
        for (int i = 0; i < MAX_TAIL; ++i)
            if (i < a.ntail) dtl[i] = *(const f32x2*)(a.Dtail_top + (size_t)i * a.Np + n);
        if (pair) {
            dout2 = *(const f32x2*)(a.d_out + ro * a.N + n);
            h2 = *(const f32x2*)(a.hall + ro * KN + (size_t)(a.K - 1) * a.N + n);
        } else if (lrow) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (n + e < a.N) {
                    dout2[e] = a.d_out[ro * a.N + n + e];
                    h2[e] = a.hall[ro * KN + (size_t)(a.K - 1) * a.N + n + e];
                }
        }
    }
    if (c > 0) {
        // ---- bottom of frame t+1: gradient w.r.t. the state that entered it -------------------
        const int par = (c - 1) & 1;
        const float* z0 = a.dz0s_part + par * pstride + rg;
        const float* dp_ = a.dps_part + par * pstride + rg;
        const f32x2 dz0 = *(const f32x2*)(a.dzp0 + hoff);
        const unsigned char vn = a.valid[(size_t)(t + 1) * a.Bp + rg];
        float zv[4], pv4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {           // the first 64 partials: branch-free, clamped index
            const int b2 = (tid & 15) + 16 * u;
            const int bc = b2 < a.nparts ? b2 : a.nparts - 1;
            zv[u] = z0[(size_t)bc * a.Bp];
            pv4[u] = dp_[(size_t)bc * a.Bp];
        }
        float s0 = 0.f, sp = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = (tid & 15) + 16 * u < a.nparts;
            s0 += ok ? zv[u] : 0.f;
            sp += ok ? pv4[u] : 0.f;
        }
        for (int b2 = (tid & 15) + 64; b2 < a.nparts; b2 += 16) {
            s0 += z0[(size_t)b2 * a.Bp];
            sp += dp_[(size_t)b2 * a.Bp];
        }
        s0 = row16_sum(s0);
        sp = row16_sum(sp);
        if (vn) {
            ds[0] = bptt_state_grad(a.u0d, a.u0o, dz0[0], s0, sp);
            ds[1] = bptt_state_grad(a.u0d, a.u0o, dz0[1], s0, sp);
        }
    }
    if (top) {
        // ---- top of frame t -------------------------------------------------------------------
        const bool v = vt != 0;
        f32x2 dz = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float dout = dout2[e];
            float dh = 0.f;
            if (v) { dh = dout + g[e] + ds[e]; g[e] = 0.f; }
            else g[e] += dout;
            dz[e] = h2[e] > 0.f ? dh : 0.f;
        }
        if (pair) {
            st_save(a.dz_all + ro * KN + (size_t)(a.K - 1) * a.N + n, dz);
        } else if (lrow) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (n + e < a.N) a.dz_all[ro * KN + (size_t)(a.K - 1) * a.N + n + e] = dz[e];
        }
        *(f32x2*)(a.gq + hoff) = g;
        *(f32x2*)(a.dstate + hoff) = ds;
        *(f32x2*)(a.dzp_top + hoff) = dz;
        f32x2 dG = {dz[0] * ia[0], dz[1] * ia[1]};
        if (a.dGp_top) *(f32x2*)(a.dGp_top + hoff) = dG;   // (Gram chain only, see scale_pack_kernel)
#pragma unroll
        for (int i = 0; i < MAX_TAIL; ++i) {
            if (i >= a.ntail) continue;
            const f32x2 dt = dtl[i];
            const float sq = row16_sum(dG[0] * dt[0] + dG[1] * dt[1]);
            if ((tid & 15) == 0) a.dq_out[((size_t)i * a.Bp + rg) * a.numA + ab] = sq;
        }
        const int par = c & 1;
        float s = dz[0] + dz[1];
        s = row16_sum(s);
        if ((tid & 15) == 0) {
            for (int pp = 0; pp < a.ppb; ++pp)
                if (ab * a.ppb + pp < a.nparts)
                    a.dps_part[par * pstride + (size_t)(ab * a.ppb + pp) * a.Bp + rg] = 0.f;
            if (a.K == 1) a.dz0s_part[par * pstride + (size_t)ab * a.Bp + rg] = s;
        }
    } else {
        // ---- after frame 0: ds = gradient w.r.t. the initial state of every row ---------------
        *(f32x2*)(a.dstate + hoff) = ds;
        sm[erow][ec] = rg < a.B ? ds[0] : 0.f;
        sm[erow][ec + 1] = rg < a.B ? ds[1] : 0.f;
        __syncthreads();
        if (tid < ATOMS) {
            float s = 0.f;
            for (int r2 = 0; r2 < ROWS; ++r2) s += sm[r2][tid];
            a.dh0_part[(size_t)m * a.Np + ab * ATOMS + tid] = s;
        }
    }
}

struct BwdAArgs {
    const float* Dn;         // packed dictionary of layer k
    const float* ia_prev;    // [Np] 1/alpha of layer k-1
    const float* drpart;     // [KS][Bp][Fp] packed partials of d r_k
    const float* dzp_in;     // packed dz_k
    float* dzp_out;          // packed dz_{k-1}
    float* dGp_out;          // packed dG_{k-1}
    const float* hall;
    float* dz_all;
    float* dR;               // [B*T][Fp] row-major d r_k of this layer (MFMA bin tiles in tile_unpermute order)
    float* dz0s_part;
    float* dps_part;
    const int* c_rd;
    int* c_wr;
    float uko;
    int k, B, T, N, K, Bp, Fp, Np, numA, nchunks;
    // odd bins: d r_tail = sum of dq_in over atom blocks; its rank-1 term is added to the MFMA
    // result, it is copied into dR, and the partials of layer k-1 are produced for the next launch
    const float* Dtail;      // [MAX_TAIL][Np] of layer k
    const float* Dtail_prev; // of layer k-1
    const float* dq_in;      // qred = 0: [MAX_TAIL][Bp][numA] partials of layer k (numA <= 64, added here);
                             // qred = 1: [MAX_TAIL][Bp], summed by the cell_b launch in between
    float* dq_out;           // [MAX_TAIL][Bp][numA] partials of layer k-1
    int ntail;
};

// (leading scalar arguments: preloaded into SGPRs, see cell_b_kernel)
// K0 (KL / beta cell only, layer 0 = a full ISTA step from the state p): the result is d p itself --
// no relu mask of a previous layer, nothing to store in dz_all -- left in dzp_out for bwd_edge_kernel.
template <int G, int KS, bool QRED = false, bool K0 = false>
__global__ void __launch_bounds__(256)
bwd_a_kernel(const float* drpart_, const float* Dn_, const int* c_rd_, int Bp_, int Fp_, int Np_,
             int numA_, int nchunks_, const BwdAArgs a_in) {
    BwdAArgs a = a_in;
    a.drpart = drpart_; a.Dn = Dn_; a.c_rd = c_rd_; a.Bp = Bp_; a.Fp = Fp_; a.Np = Np_;
    a.numA = numA_; a.nchunks = nchunks_;
    __shared__ __attribute__((aligned(16))) float red[4 * ROWS * ATOMS];
    const int m = blockIdx.x >> 3;                          // grid layout: see cell_a_kernel
    const int ab_raw = blockIdx.y * 8 + (blockIdx.x & 7);
    const bool live = ab_raw < a.numA;
    const int ab = live ? ab_raw : a.numA - 1;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63, j = l & 15, q = l >> 4;
    const int Fp = a.Fp, Np = a.Np;
    const int row0 = m * ROWS, n0 = ab * ATOMS;
    const size_t pstride = (size_t)a.Bp * Fp;
    const int NAC = Np / 16, nft = Fp / 16, ac0 = ab * 2;
    const float* arow = a.drpart + (size_t)m * nft * 256 + l * 4;
    const float* brow = a.Dn + (size_t)ab * 512 + l * 4;     // the cell_a packing (common.h)
    const size_t bstep = (size_t)NAC * 256;

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int per_wave = (a.nchunks - w + 3) >> 2;
    const int clast = a.nchunks - 1;
    f32x4 av[G][KS];
    f32x4 bv[G][2];
    auto load_chunk = [&](int i, int g) {      // chunk i of this wave -> slot g
        int c = w + 4 * i;
        c = c > clast ? clast : c;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            av[g][ks] = *(const f32x4*)(arow + 256 * c + (size_t)ks * pstride);
        bv[g][0] = *(const f32x4*)(brow + (size_t)c * bstep);
        bv[g][1] = *(const f32x4*)(brow + (size_t)c * bstep + 256);
    };
    // same software pipeline as the forward's cell_a_kernel: G rotating operand slots, loads PF =
    // G-1 chunks ahead of the MFMAs, clamped (never-loaded slots zeroed), last group peeled when
    // every wave owns whole groups
    constexpr int PF = G - 1;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        bv[g][0] = bv[g][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) av[g][ks] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the d r copy (below): this workgroup's chunk c = ab, requested ahead of the operand stream
    const bool dr_mine = live && ab < a.nchunks && (ab & 3) == w;
    f32x4 drv[KS];
    if (dr_mine) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            drv[ks] = *(const f32x4*)(arow + 256 * ab + (size_t)ks * pstride);
    }
#pragma unroll
    for (int g = 0; g < PF; ++g) load_chunk(g, g);
    __builtin_amdgcn_sched_barrier(0);   // struct-dependent code stays behind the first operand loads

    const int cnt = *a.c_rd;
    if (a.c_wr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.c_wr = cnt + 1;
    const int t = a.T - 1 - cnt;
    const int erow = tid >> 4, ec = (tid & 15) * 2;
    const int rg = row0 + erow, n = n0 + ec;
    const size_t hoff = ((size_t)m * NAC + ac0 + (ec >> 4)) * 256 + hp_pos(erow, ec & 15);
    const int KN = a.K * a.N;
    const f32x2 dzk = *(const f32x2*)(a.dzp_in + hoff);
    const f32x2 ia = *(const f32x2*)(a.ia_prev + n);
    // this row's running sum(dz_k) term (read-modify-write once per launch): its load goes out here,
    // with everything else -- at its point of use it was a dependent round trip, behind a wait for every
    // store of the epilogue, on the tail of each launch
    const size_t po = (size_t)(cnt & 1) * a.numA * a.Bp + (size_t)ab * a.Bp + rg;
    float dps_old = 0.f;
    if ((tid & 15) == 0) dps_old = a.dps_part[po];
    f32x2 hprev = {0.f, 0.f};
    if (K0) hprev = f32x2{1.f, 1.f};
    else {
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (rg < a.B && n + e < a.N)
                hprev[e] = a.hall[((size_t)rg * a.T + t) * KN + (size_t)(a.k - 1) * a.N + n + e];
    }

    f32x2 dt[MAX_TAIL], dtp[MAX_TAIL];
    float qs[MAX_TAIL], qv[MAX_TAIL][4];
#pragma unroll
    for (int i = 0; i < MAX_TAIL; ++i) {
        dt[i] = *(const f32x2*)(a.Dtail + (size_t)i * Np + n);
        dtp[i] = *(const f32x2*)(a.Dtail_prev + (size_t)i * Np + n);
        qs[i] = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) qv[i][u] = 0.f;
        if (i >= a.ntail) continue;
        if (QRED) {
            qs[i] = a.dq_in[(size_t)i * a.Bp + rg];
        } else {
            const float* qp = a.dq_in + ((size_t)i * a.Bp + rg) * a.numA;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b2 = (tid & 15) + 16 * u;
                if (b2 < a.numA) qv[i][u] = qp[b2];
            }
        }
    }

    // The summed d r_k of this row tile goes out row-major for the weight gradients.  Every
    // workgroup of the row tile holds all of it; chunk c is stored by atom block c mod numA so
    // that no single workgroup carries the whole copy (it used to be block 0: the launch then
    // waited for that one straggler).
    float* const dr_row = a.dR + ((size_t)(row0 + j) * a.T + t) * Fp + 4 * q;   // (tile_unpermute order)
    const bool dr_lane = row0 + j < a.B;
    auto compute_chunk = [&](int base, int g) {
        f32x4 r4 = av[g][0];
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) r4 += av[g][ks];
        const bool ok = base + g < per_wave;
        if (!ok) r4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc0 = mfma16(r4[s], bv[g][s >> 1][(s & 1) * 2], acc0);
            acc1 = mfma16(r4[s], bv[g][s >> 1][(s & 1) * 2 + 1], acc1);
        }
    };
    const bool exact = (a.nchunks % (4 * G)) == 0;
    int base = 0;
    for (; base + (exact ? G : 0) < per_wave; base += G) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            load_chunk(base + g + PF, (g + PF) % G);
            __builtin_amdgcn_sched_barrier(0);
            compute_chunk(base, g);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (exact) {
        load_chunk(base + PF, PF % G);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            compute_chunk(base, g);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // Every load of this kernel was requested ahead of the last operand chunk and loads return in
    // order: all of them are here.  Said explicitly, because the compiler otherwise waits for the
    // early epilogue operands (tail rows, partials) where it first uses them -- behind the epilogue's
    // conditional stores, with a count that is safe for the path issuing the FEWEST stores, i.e. on
    // the common path it sat out nearly every store's acknowledgement (s_waitcnt vmcnt(1) before the
    // last tail partial).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
    // the d r copy, outside the MFMA loop (a wave-uniform branch inside it costs several %): the
    // chunks c = ab (mod numA) are stored by the wave that owns them.  The first one was requested
    // before the operand stream (its data is there by now: loads return in order) -- re-read here
    // its cache-hit latency sat in front of the cross-wave reduction of every launch; further
    // chunks (fewer atom blocks than chunks: small dictionaries) are re-read.
#ifndef DRNMF_EXP_NODR
    if (dr_mine) {
        f32x4 r4 = drv[0];
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) r4 += drv[ks];
        if (dr_lane) st_save(dr_row + 16 * ab, r4);
    }
#endif
    if (live) {
        for (int c = ab + a.numA; c < a.nchunks; c += a.numA) {
            if ((c & 3) != w) continue;
            f32x4 r4 = *(const f32x4*)(arow + 256 * c);
#pragma unroll
            for (int ks = 1; ks < KS; ++ks) r4 += *(const f32x4*)(arow + 256 * c + (size_t)ks * pstride);
            if (dr_lane) st_save(dr_row + 16 * c, r4);
        }
    }

#pragma unroll
    for (int v = 0; v < 4; ++v) {
        f32x2 pr = {acc0[v], acc1[v]};
        *(f32x2*)(red + (w * ROWS + 4 * q + v) * ATOMS + 2 * j) = pr;
    }
    __syncthreads();
    f32x2 gsum = *(const f32x2*)(red + (0 * ROWS + erow) * ATOMS + ec);
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) {
        const f32x2 p2 = *(const f32x2*)(red + (ww * ROWS + erow) * ATOMS + ec);
        gsum[0] += p2[0];
        gsum[1] += p2[1];
    }

    if (!live) return;
#pragma unroll
    for (int i = 0; i < MAX_TAIL; ++i) {
        if (i >= a.ntail) continue;
        float sq = qs[i];                                     // d r_tail[row] of layer k
        if (!QRED) {
            sq = (qv[i][0] + qv[i][1]) + (qv[i][2] + qv[i][3]);
            const float* qp = a.dq_in + ((size_t)i * a.Bp + rg) * a.numA;
            for (int b2 = (tid & 15) + 64; b2 < a.numA; b2 += 16) sq += qp[b2];
            sq = row16_sum(sq);
        }
        gsum[0] = fmaf(sq, dt[i][0], gsum[0]);
        gsum[1] = fmaf(sq, dt[i][1], gsum[1]);
        if (ab_raw == 0 && (tid & 15) == 0 && rg < a.B)
            a.dR[((size_t)rg * a.T + t) * Fp + 16 * a.nchunks + i] = sq;
    }
    if (a.ntail > 0 && ab_raw == 0 && rg < a.B) {             // the rest of the tail tile is padding
        const int c0 = a.ntail + (tid & 15);
        if (c0 < 16) a.dR[((size_t)rg * a.T + t) * Fp + 16 * a.nchunks + c0] = 0.f;
    }
    f32x2 dzn;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float dh = dzk[e] - gsum[e];
        dzn[e] = hprev[e] > 0.f ? dh : 0.f;
    }
#ifndef DRNMF_EXP_NODZ
    if (!K0 && rg < a.B) {     // one 8-byte store where the pair is whole and aligned (n is even)
        float* dzo = a.dz_all + ((size_t)rg * a.T + t) * KN + (size_t)(a.k - 1) * a.N + n;
        if (n + 1 < a.N && (a.N & 1) == 0) st_save(dzo, dzn);
        else {
            if (n < a.N) dzo[0] = dzn[0];
            if (n + 1 < a.N) dzo[1] = dzn[1];
        }
    }
#endif
    st_xchg(a.dzp_out + hoff, dzn);
    f32x2 dG = {dzn[0] * ia[0], dzn[1] * ia[1]};
    if (a.dGp_out) *(f32x2*)(a.dGp_out + hoff) = dG;       // (nullptr: cell_b reads dz, scale_pack_kernel)
    if (a.k >= 2) {
#pragma unroll
        for (int i = 0; i < MAX_TAIL; ++i) {
            if (i >= a.ntail) continue;
            const float sq = row16_sum(dG[0] * dtp[i][0] + dG[1] * dtp[i][1]);
            if ((tid & 15) == 0) a.dq_out[((size_t)i * a.Bp + rg) * a.numA + ab] = sq;
        }
    }
    float s = dzk[0] + dzk[1], s0 = dzn[0] + dzn[1];
    s = row16_sum(s);
    s0 = row16_sum(s0);
    if ((tid & 15) == 0) {
        a.dps_part[po] = dps_old + a.uko * s;
        if (a.k == 1) a.dz0s_part[po] = s0;
    }
}

struct BwdAParams {
    void* p[9];
    explicit BwdAParams(BwdAArgs& a)
        : p{&a.drpart, &a.Dn, &a.c_rd, &a.Bp, &a.Fp, &a.Np, &a.numA, &a.nchunks, &a} {}
};

template <int KS>
void* bwd_a_func(int per_wave, bool qred) {
    if (qred) return per_wave <= 2 ? (void*)&bwd_a_kernel<2, KS, true> : (void*)&bwd_a_kernel<4, KS, true>;
    if (per_wave <= 2) return (void*)&bwd_a_kernel<2, KS>;
    return (void*)&bwd_a_kernel<4, KS>;
}
void* pick_bwd_a(int nchunks, int KS, bool qred = false, bool k0 = false) {
    const int per_wave = (nchunks + 3) / 4;
    if (k0)          // (the KL / beta cell: one atom range, no odd-bin side path)
        return per_wave <= 2 ? (void*)&bwd_a_kernel<2, 1, false, true> : (void*)&bwd_a_kernel<4, 1, false, true>;
    switch (KS) {
        case 1: return bwd_a_func<1>(per_wave, qred);
        case 2: return bwd_a_func<2>(per_wave, qred);
        case 4: return bwd_a_func<4>(per_wave, qred);
        default: return bwd_a_func<8>(per_wave, qred);
    }
}

// The chain's d r_k = (dz_k * ia_k) Dn_k^T needs dz scaled per atom; scaling the dictionary's
// columns once per step instead lets cell_b contract the packed dz itself, and bwd_a / bwd_edge
// stop storing a second packed copy (dG) of every layer-step: out = Dp with column n times ia[n],
// same packing (block (f/16, n/16), element ((n%16)/4 * 16 + f%16) * 4 + n%4).
__global__ void __launch_bounds__(256)
scale_pack_kernel(const float* __restrict__ Dp, const float* __restrict__ ia, float* __restrict__ out,
                  size_t total, int Np) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const int n = (int)((p >> 8) % (size_t)(Np / 16)) * 16 + (int)((p & 255) >> 6) * 4 + (int)(p & 3);
    out[p] = Dp[p] * ia[n];
}

// ---------------- time-batched phase --------------------------------------------------------------
__global__ void __launch_bounds__(256)
unpack_dn_kernel(const float* __restrict__ Dp, float* __restrict__ Dn, int Fp, int Np) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)Fp * Np) return;
    const int f = (int)(i / Np), n = (int)(i % Np);
    Dn[i] = Dp[((size_t)(f >> 4) * (Np / 16) + (n >> 4)) * 256 + (((n & 15) >> 2) * 16 + (f & 15)) * 4 +
               (n & 3)];
}

struct EpiResid {   // R = X - H Dn^T
    const float* X;
    float* R;
    int F, ldr;
    __device__ f32x2 pre(int64_t row, int col) const { return f32x2{X[row * F + col], 0.f}; }
    __device__ void operator()(int64_t row, int col, float xh, f32x2 pv) const {
        R[row * ldr + col] = pv[0] - xh;
    }
};
struct EpiStore {   // C = acc
    float* C;
    int ldc;
    __device__ f32x2 pre(int64_t, int) const { return f32x2{0.f, 0.f}; }
    __device__ void operator()(int64_t row, int col, float acc, f32x2) const {
        C[row * ldc + col] = acc;
    }
};
// (perm_rows: the A operand's columns below it are in the chain kernels' saved order, common.h
// tile_unpermute -- output row m of the product is bin tile_unpermute(m))
struct EpiP1 {      // partial[split] = acc * ia[n]
    float* P;
    const float* ia;
    int Np;
    size_t stride;
    int perm_rows;
    __device__ float pre(int, int, int n) const { return ia[n]; }
    __device__ void operator()(int split, int m, int n, float acc, float pv) const {
        if (m < perm_rows) m = tile_unpermute(m);
        P[split * stride + (size_t)m * Np + n] = acc * pv;
    }
};
struct EpiP2 {      // partial[split] -= acc
    float* P;
    int Np;
    size_t stride;
    int perm_rows;
    __device__ float pre(int split, int m, int n) const {
        if (m < perm_rows) m = tile_unpermute(m);
        return P[split * stride + (size_t)m * Np + n];
    }
    __device__ void operator()(int split, int m, int n, float acc, float pv) const {
        if (m < perm_rows) m = tile_unpermute(m);
        P[split * stride + (size_t)m * Np + n] = pv - acc;
    }
};

// d log_D (+)= Dn * (dDn - Dn * c),  c[n] = sum_f dDn[f][n] Dn[f][n],  dDn = sum of the partials.
// N % 4 == 0: two elementwise passes over (16-bin groups) x (64 atom quads) workgroups, 16-byte
// accesses -- pass 1 adds the splits up (into split 0) and leaves each bin group's share of c in
// `cpart`, pass 2 adds the shares in a fixed order and applies.  (The single-kernel form below
// ran 63 workgroups for a whole layer: 0.4 ms at C2.)
__global__ void __launch_bounds__(256)
dlogd_sum_kernel(float* __restrict__ P, const float* __restrict__ Dn, float* __restrict__ cpart,
                 int F, int N, int Np, int splits, size_t stride, int bpt) {
    // bpt = bins per thread: 4 (16 bins per workgroup), or 1 for narrow dictionaries, whose handful
    // of workgroups would otherwise walk 64 split partials x 4 bins each (N = 200: 17 workgroups, 78 us)
    __shared__ f32x4 cs[4][64];
    const int qn = threadIdx.x & 63, fl = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + qn) * 4;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        for (int i = 0; i < bpt; ++i) {
            const int f = (blockIdx.y * 4 + fl) * bpt + i;
            if (f >= F) break;
            const size_t o = (size_t)f * Np + n;
            // (eight splits' loads in flight, added in split order: one dependent 16-byte load per split
            // made N = 200 layers wait 20 us here)
            f32x4 g = *(const f32x4*)(P + o);
            for (int s0 = 1; s0 < splits; s0 += 8) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = *(const f32x4*)(P + (size_t)(s0 + u < splits ? s0 + u : splits - 1) * stride + o);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (s0 + u < splits) g += v[u];
            }
            *(f32x4*)(P + o) = g;
            const f32x4 dn = *(const f32x4*)(Dn + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = fmaf(g[e], dn[e], c[e]);
        }
    }
    cs[fl][qn] = c;
    __syncthreads();
    if (fl == 0 && n < N)
        *(f32x4*)(cpart + (size_t)blockIdx.y * Np + n) = (cs[0][qn] + cs[1][qn]) + (cs[2][qn] + cs[3][qn]);
}
__global__ void __launch_bounds__(256)
dlogd_apply_kernel(const float* __restrict__ P, const float* __restrict__ Dn,
                   const float* __restrict__ cpart, float* __restrict__ dlogD, int F, int N, int Np,
                   int ngroups, int accumulate, int bpt) {
    const int qn = threadIdx.x & 63, fl = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + qn) * 4;
    if (n >= N) return;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int g0 = 0; g0 < ngroups; g0 += 16) {          // (16 loads in flight, added in group order)
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            v[u] = *(const f32x4*)(cpart + (size_t)(g0 + u < ngroups ? g0 + u : ngroups - 1) * Np + n);
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (g0 + u < ngroups) c += v[u];
    }
    for (int i = 0; i < bpt; ++i) {
        const int f = (blockIdx.y * 4 + fl) * bpt + i;
        if (f >= F) break;
        const size_t o = (size_t)f * Np + n;
        const f32x4 g = *(const f32x4*)(P + o), dn = *(const f32x4*)(Dn + o);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dn[e] * (g[e] - dn[e] * c[e]);
        f32x4* op = (f32x4*)(dlogD + (size_t)f * N + n);
        *op = accumulate ? *op + v : v;
    }
}
// General N: workgroup = 32 atoms x 8 bin groups (bins f = fg mod 8); the column sums c are combined
// through LDS in a fixed order (deterministic).
__global__ void __launch_bounds__(256)
dlogd_kernel(const float* __restrict__ P, const float* __restrict__ Dn, float* __restrict__ dlogD,
             int F, int N, int Np, int splits, size_t stride, int accumulate) {
    __shared__ float cs[8][32];
    const int ln = threadIdx.x & 31, fg = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + ln;
    const bool ok = n < N;
    float c = 0.f;
    if (ok) {
        for (int f = fg; f < F; f += 8) {
            const float g = ordered_sum<8>(P + (size_t)f * Np + n, stride, splits);
            c = fmaf(g, Dn[(size_t)f * Np + n], c);
        }
    }
    cs[fg][ln] = c;
    __syncthreads();
    if (!ok) return;
    c = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) c += cs[i][ln];
    for (int f = fg; f < F; f += 8) {
        const float g = ordered_sum<8>(P + (size_t)f * Np + n, stride, splits);
        const float dn = Dn[(size_t)f * Np + n];
        const float v = dn * (g - dn * c);
        float* o = dlogD + (size_t)f * N + n;
        *o = accumulate ? *o + v : v;
    }
}

// per-atom sums over all frames of layer k: Sb = sum dz, Sgi = sum dz * (h_k - base) (= ia * sum dz G)
// and, for F = 16 j + 1 (the 2^k + 1 STFT sizes), the ODD BIN's row of the two weight-gradient
// products -- S1 = sum_t r_k[t][Fm] dz_k[t][n], S2 = sum_t d r_k[t][Fm] h_{k-1}[t][n] -- which would
// otherwise cost the TN GEMMs a fifth, 1/128-full row of output tiles (M = 528: 5 x 128).  dz and
// h_{k-1} are streamed here anyway.
constexpr int CR_SLOTS = 4;
struct ColRedArgs {
    const float* hall;
    const float* dz_all;
    const float* bias;       // [Np] of layer k
    const float* psum_all;   // [T][Bp]
    const unsigned char* seen;
    const float* log_h0;
    const float* init;       // stateful training: the state every sequence ENTERS with, [B][N] (a constant of
                             // the gradient, custom_layers.py:296-318), instead of softplus(log_h0); or NULL
    float* part;             // [CR_SPLITS][CR_SLOTS][Np]
    const float* rt;         // r_k[.][Fm]  (row stride ldr) or nullptr: no odd-bin row
    const float* drt;        // d r_k[.][Fm] (row stride ldr), k >= 1
    float u0d, u0o, uko;
    int k, B, T, N, K, Bp, Np, ldr;
};
__global__ void __launch_bounds__(256) colreduce_kernel(const ColRedArgs a) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int sp = blockIdx.y;
    if (n >= a.N) return;
    const int64_t BT = (int64_t)a.B * a.T;
    const int64_t per = (BT + CR_SPLITS - 1) / CR_SPLITS;
    const int64_t r0 = sp * per;
    int64_t r1 = r0 + per;
    if (r1 > BT) r1 = BT;
    const int KN = a.K * a.N;
    const float bk = a.bias[n];
    float h0v = 0.f;
    if (a.k == 0) {
        const float z = a.log_h0[n];
        h0v = (z > 20.f) ? z : log1pf(expf(z));
    }
    const bool tail = a.rt != nullptr, tail2 = tail && a.k >= 1;
    float sb = 0.f, sg = 0.f, s1 = 0.f, s2 = 0.f;
    int b = (int)(r0 / a.T), t = (int)(r0 % a.T) - 1;
    // dz is sparse (it inherits the zeros of h): its loads decide everything else, so eight of
    // them are in flight at a time (one dependent load per row made this kernel latency-bound:
    // 1.7 ms per layer at the C2 shape)
    constexpr int U = 8;
    const float* dzp = a.dz_all + (size_t)a.k * a.N + n;
    const float* hpp = a.hall + (size_t)(a.k >= 1 ? a.k - 1 : 0) * a.N + n;
    for (int64_t bt0 = r0; bt0 < r1; bt0 += U) {
        float dzv[U], hpv[U], rtv[U], drv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t bt = bt0 + u < r1 ? bt0 + u : r1 - 1;
            dzv[u] = dzp[bt * KN];
            if (tail) rtv[u] = a.rt[bt * a.ldr];
            if (tail2) {
                hpv[u] = hpp[bt * KN];
                drv[u] = a.drt[bt * a.ldr];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t bt = bt0 + u;
            if (bt >= r1) break;
            if (++t == a.T) { t = 0; ++b; }
            if (tail2) s2 = fmaf(drv[u], hpv[u], s2);
            const float dz = dzv[u];
            if (dz == 0.f) continue;
            if (tail) s1 = fmaf(rtv[u], dz, s1);
            const float ps = a.psum_all[(size_t)t * a.Bp + b];
            const float hk = a.hall[bt * KN + (size_t)a.k * a.N + n];
            float base;
            if (a.k == 0) {
                const float p = a.seen[(size_t)t * a.Bp + b]
                                    ? a.hall[(bt - 1) * KN + (size_t)(a.K - 1) * a.N + n]
                                    : (a.init ? a.init[(size_t)b * a.N + n] : h0v);
                base = a.u0d * p + a.u0o * (ps - p) + bk;
            } else {
                base = (tail2 ? hpv[u] : a.hall[bt * KN + (size_t)(a.k - 1) * a.N + n]) + bk +
                       a.uko * ps;
            }
            sb += dz;
            sg = fmaf(dz, hk - base, sg);
        }
    }
    float* o = a.part + (size_t)sp * CR_SLOTS * a.Np + n;
    o[0] = sb;
    o[(size_t)a.Np] = sg;
    o[(size_t)2 * a.Np] = s1;
    o[(size_t)3 * a.Np] = s2;
}

// Same sums, four atoms per thread (N % 4 == 0): the dz / h_{k-1} streams come in 16-byte loads,
// 4 KB contiguous per row and workgroup (the scalar kernel above moved 2 GB per layer at 1.4 TB/s).
// The rarely taken branch -- dz != 0 -- still gathers h_k / p with scalar loads.
__global__ void __launch_bounds__(256) colreduce4_kernel(const ColRedArgs a) {
    const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int sp = blockIdx.y;
    if (n >= a.N) return;
    const int64_t BT = (int64_t)a.B * a.T;
    const int64_t per = (BT + CR_SPLITS - 1) / CR_SPLITS;
    const int64_t r0 = sp * per;
    int64_t r1 = r0 + per;
    if (r1 > BT) r1 = BT;
    const int KN = a.K * a.N;
    const f32x4 bk = *(const f32x4*)(a.bias + n);
    f32x4 h0v = {0.f, 0.f, 0.f, 0.f};
    if (a.k == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = a.log_h0[n + e];
            h0v[e] = (z > 20.f) ? z : log1pf(expf(z));
        }
    }
    const bool tail = a.rt != nullptr, tail2 = tail && a.k >= 1;
    f32x4 sb = {0.f, 0.f, 0.f, 0.f}, sg = sb, s1 = sb, s2 = sb;
    int b = (int)(r0 / a.T), t = (int)(r0 % a.T) - 1;
    constexpr int U = 4;
    const float* dzp = a.dz_all + (size_t)a.k * a.N + n;
    const float* hpp = a.hall + (size_t)(a.k >= 1 ? a.k - 1 : 0) * a.N + n;
    for (int64_t bt0 = r0; bt0 < r1; bt0 += U) {
        f32x4 dzv[U], hpv[U];
        float rtv[U], drv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t bt = bt0 + u < r1 ? bt0 + u : r1 - 1;
            dzv[u] = *(const f32x4*)(dzp + bt * KN);
            if (tail) rtv[u] = a.rt[bt * a.ldr];
            if (tail2) {
                hpv[u] = *(const f32x4*)(hpp + bt * KN);
                drv[u] = a.drt[bt * a.ldr];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t bt = bt0 + u;
            if (bt >= r1) break;
            if (++t == a.T) { t = 0; ++b; }
            if (tail2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) s2[e] = fmaf(drv[u], hpv[u][e], s2[e]);
            }
            const f32x4 dz = dzv[u];
            if (dz[0] == 0.f && dz[1] == 0.f && dz[2] == 0.f && dz[3] == 0.f) continue;
            const float ps = a.psum_all[(size_t)t * a.Bp + b];
            const f32x4 hk = *(const f32x4*)(a.hall + bt * KN + (size_t)a.k * a.N + n);
            f32x4 base;
            if (a.k == 0) {
                f32x4 p = h0v;
                if (a.seen[(size_t)t * a.Bp + b])
                    p = *(const f32x4*)(a.hall + (bt - 1) * KN + (size_t)(a.K - 1) * a.N + n);
                else if (a.init)
                    p = *(const f32x4*)(a.init + (size_t)b * a.N + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) base[e] = a.u0d * p[e] + a.u0o * (ps - p[e]) + bk[e];
            } else {
                const f32x4 hp = tail2 ? hpv[u]
                                       : *(const f32x4*)(a.hall + bt * KN + (size_t)(a.k - 1) * a.N + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) base[e] = hp[e] + bk[e] + a.uko * ps;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (dz[e] == 0.f) continue;          // (exactly the scalar kernel's terms and order)
                if (tail) s1[e] = fmaf(rtv[u], dz[e], s1[e]);
                sb[e] += dz[e];
                sg[e] = fmaf(dz[e], hk[e] - base[e], sg[e]);
            }
        }
    }
    float* o = a.part + (size_t)sp * CR_SLOTS * a.Np + n;
    *(f32x4*)o = sb;
    *(f32x4*)(o + (size_t)a.Np) = sg;
    *(f32x4*)(o + (size_t)2 * a.Np) = s1;
    *(f32x4*)(o + (size_t)3 * a.Np) = s2;
}

// The same sums for NARROW dictionaries (N / 4 <= 64 atom quads): the 256 threads of a workgroup are
// 64 quads x 4 row lanes -- lane rl takes the rows r0 + rl, r0 + rl + 4, ... of the split -- and the four
// partial sums meet in LDS in a fixed order.  (With one thread per quad an N = 200 layer ran 50 threads
// per workgroup through 62 dependent rows each: 72 us.)
__global__ void __launch_bounds__(256) colreduce4_rows_kernel(const ColRedArgs a) {
    __shared__ f32x4 acc[4][4][64];
    const int qn = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = qn * 4;
    const int sp = blockIdx.y;
    const bool live = n < a.N;
    const int64_t BT = (int64_t)a.B * a.T;
    const int64_t per = (BT + CR_SPLITS - 1) / CR_SPLITS;
    const int64_t r0 = sp * per;
    int64_t r1 = r0 + per;
    if (r1 > BT) r1 = BT;
    const int KN = a.K * a.N;
    f32x4 sb = {0.f, 0.f, 0.f, 0.f}, sg = sb, s1 = sb, s2 = sb;
    if (live) {
        const f32x4 bk = *(const f32x4*)(a.bias + n);
        f32x4 h0v = {0.f, 0.f, 0.f, 0.f};
        if (a.k == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z = a.log_h0[n + e];
                h0v[e] = (z > 20.f) ? z : log1pf(expf(z));
            }
        }
        const bool tail = a.rt != nullptr, tail2 = tail && a.k >= 1;
        const float* dzp = a.dz_all + (size_t)a.k * a.N + n;
        const float* hpp = a.hall + (size_t)(a.k >= 1 ? a.k - 1 : 0) * a.N + n;
        constexpr int U = 4;
        for (int64_t bt0 = r0 + rl; bt0 < r1; bt0 += 4 * U) {
            f32x4 dzv[U], hpv[U];
            float rtv[U], drv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int64_t bt = bt0 + 4 * u;
                bt = bt < r1 ? bt : r1 - 1;
                dzv[u] = *(const f32x4*)(dzp + bt * KN);
                if (tail) rtv[u] = a.rt[bt * a.ldr];
                if (tail2) {
                    hpv[u] = *(const f32x4*)(hpp + bt * KN);
                    drv[u] = a.drt[bt * a.ldr];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t bt = bt0 + 4 * u;
                if (bt >= r1) break;
                if (tail2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) s2[e] = fmaf(drv[u], hpv[u][e], s2[e]);
                }
                const f32x4 dz = dzv[u];
                if (dz[0] == 0.f && dz[1] == 0.f && dz[2] == 0.f && dz[3] == 0.f) continue;
                const int b = (int)(bt / a.T), t = (int)(bt - (int64_t)b * a.T);
                const float ps = a.psum_all[(size_t)t * a.Bp + b];
                const f32x4 hk = *(const f32x4*)(a.hall + bt * KN + (size_t)a.k * a.N + n);
                f32x4 base;
                if (a.k == 0) {
                    f32x4 p = h0v;
                    if (a.seen[(size_t)t * a.Bp + b])
                        p = *(const f32x4*)(a.hall + (bt - 1) * KN + (size_t)(a.K - 1) * a.N + n);
                    else if (a.init)
                        p = *(const f32x4*)(a.init + (size_t)b * a.N + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) base[e] = a.u0d * p[e] + a.u0o * (ps - p[e]) + bk[e];
                } else {
                    const f32x4 hp = tail2 ? hpv[u]
                                           : *(const f32x4*)(a.hall + bt * KN + (size_t)(a.k - 1) * a.N + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) base[e] = hp[e] + bk[e] + a.uko * ps;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (dz[e] == 0.f) continue;
                    if (tail) s1[e] = fmaf(rtv[u], dz[e], s1[e]);
                    sb[e] += dz[e];
                    sg[e] = fmaf(dz[e], hk[e] - base[e], sg[e]);
                }
            }
        }
    }
    acc[0][rl][qn] = sb; acc[1][rl][qn] = sg; acc[2][rl][qn] = s1; acc[3][rl][qn] = s2;
    __syncthreads();
    if (rl == 0 && live) {
        float* o = a.part + (size_t)sp * CR_SLOTS * a.Np + n;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
            *(f32x4*)(o + (size_t)sl * a.Np) = (acc[sl][0][qn] + acc[sl][1][qn]) + (acc[sl][2][qn] + acc[sl][3][qn]);
    }
}

// stage 1 of the scalar gradients: the CR_SPLITS partial sums of every atom, added in split order
// (one thread per atom; slot 0 of `part` receives the totals).  With an odd-bin row (Ptail != null)
// its gradient ia[n] S1[n] - S2[n] goes where the GEMM partials of that row would have been: split 0
// of P, zeros in the other splits (dlogd_kernel adds the splits up).
__global__ void __launch_bounds__(256)
colreduce_fold_kernel(float* __restrict__ part, int N, int Np, float* __restrict__ Ptail,
                      const float* __restrict__ ia, int splits, size_t pstride) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float sb = 0.f, sg = 0.f, s1 = 0.f, s2 = 0.f;
    static_assert(CR_SPLITS % 32 == 0, "colreduce folds walk the splits eight at a time");
    for (int s0 = 0; s0 < CR_SPLITS; s0 += 8) {          // (32 loads in flight, added in split order)
        float v[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* p = part + (size_t)(s0 + u) * CR_SLOTS * Np + n;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = p[(size_t)k * Np];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { sb += v[u][0]; sg += v[u][1]; s1 += v[u][2]; s2 += v[u][3]; }
    }
    part[n] = sb;
    part[(size_t)Np + n] = sg;
    if (Ptail) {
        Ptail[n] = ia[n] * s1 - s2;
        for (int s = 1; s < splits; ++s) Ptail[s * pstride + n] = 0.f;
    }
}

// The same for narrow dictionaries: 64 atoms x 4 split lanes per workgroup (one thread per atom walked
// 256 splits x 4 slots of dependent loads: 30 us for N = 200); fixed combination order.
__global__ void __launch_bounds__(256)
colreduce_fold_lanes_kernel(float* __restrict__ part, int N, int Np, float* __restrict__ Ptail,
                            const float* __restrict__ ia, int splits, size_t pstride) {
    __shared__ float acc[4][4][64];
    const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    float sb = 0.f, sg = 0.f, s1 = 0.f, s2 = 0.f;
    if (n < N)
        for (int s0 = sl; s0 < CR_SPLITS; s0 += 32) {   // (eight of the lane's splits = 32 loads in flight)
            float v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* p = part + (size_t)(s0 + 4 * u) * CR_SLOTS * Np + n;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[u][k] = p[(size_t)k * Np];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { sb += v[u][0]; sg += v[u][1]; s1 += v[u][2]; s2 += v[u][3]; }
        }
    acc[0][sl][c] = sb; acc[1][sl][c] = sg; acc[2][sl][c] = s1; acc[3][sl][c] = s2;
    __syncthreads();
    if (sl != 0 || n >= N) return;
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = (acc[k][0][c] + acc[k][1][c]) + (acc[k][2][c] + acc[k][3][c]);
    part[n] = t[0];
    part[(size_t)Np + n] = t[1];
    if (Ptail) {
        Ptail[n] = ia[n] * t[2] - t[3];
        for (int s = 1; s < splits; ++s) Ptail[s * pstride + n] = 0.f;
    }
}

// d log_alph / d log_lam1 of layer k from the per-atom sums: ia = exp(-log_alph), b = -lam * ia
//   d log_alph[n] = -Sgi[n] - b[n] Sb[n];   d log_lam1 = sum_n b[n] Sb[n]
__global__ void __launch_bounds__(256)
scalar_grads_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                    float* __restrict__ d_alph, float* __restrict__ d_lam, int N, int Np,
                    int alph_len, int acc_alph, int acc_lam) {
    __shared__ float sa[256], sl[256];
    float ta = 0.f, tl = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float sb = part[n], sg = part[(size_t)Np + n];      // folded by colreduce_fold_kernel
        const float b = bias[n];
        const float va = -sg - b * sb;
        tl += b * sb;
        if (alph_len > 1) d_alph[n] = acc_alph ? d_alph[n] + va : va;
        else ta += va;
    }
    sa[threadIdx.x] = ta;
    sl[threadIdx.x] = tl;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sa[threadIdx.x] += sa[threadIdx.x + o];
            sl[threadIdx.x] += sl[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (alph_len == 1) d_alph[0] = acc_alph ? d_alph[0] + sa[0] : sa[0];
        d_lam[0] = acc_lam ? d_lam[0] + sl[0] : sl[0];
    }
}

__global__ void __launch_bounds__(256)
dlogh0_kernel(const float* __restrict__ dh0_part, const float* __restrict__ log_h0,
              float* __restrict__ d_log_h0, int N, int Np, int numM, int stateful) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float s = ordered_sum<8>(dh0_part + n, (size_t)Np, numM);
    // (stateful training: the sequences entered with a supplied state, a constant: log_h0 was not used)
    d_log_h0[n] = stateful ? 0.f : s / (1.f + expf(-log_h0[n]));   // d softplus = sigmoid
}

// ---- KL / beta cell (ista_kl / ista_beta run recurrently, cell_forward.hip) ----------------------------
// r = g(x, x^) with g = x / x^ - 1 (KL) or x x^(beta-2) - x^(beta-1) (enhance.py:431, 450); the chain
// needs d x^ = d r * dg/dx^.  With s = -dg/dx^ the launches around it keep the Euclidean cell's signs
// (there s = 1):  s_KL = x / x^2,  s_beta = (beta-1) x^(beta-2) - (beta-2) x x^(beta-3).
__device__ __forceinline__ float ista_g(int div, float beta, float xv, float xe) {
    if (div == DRNMF_DIV_KL) return xv / xe - 1.f;
    return xv * powf(xe, beta - 2.f) - powf(xe, beta - 1.f);
}
__device__ __forceinline__ float ista_neg_dg(int div, float beta, float xv, float xe) {
    if (div == DRNMF_DIV_KL) return xv / (xe * xe);
    return (beta - 1.f) * powf(xe, beta - 2.f) - (beta - 2.f) * xv * powf(xe, beta - 3.f);
}
// d r partial of the current (frame, layer), packed [Bp][Fp] (one atom range), times s(x_t, x^_{t,k})
// (a masked frame carries no gradient, and its x^ may be 0 / 0 -- the forward ran the iteration on
// x_t = 0 there and discarded the result: the factor is forced to 0, not multiplied by 0)
__global__ void __launch_bounds__(256)
dgdx_scale_kernel(const float* __restrict__ xp, const float* __restrict__ xhat,
                  float* __restrict__ drpart, const int* c_rd, const unsigned char* __restrict__ valid,
                  int T, int k, int K, int div, float beta, int F, int Fp, int Bp) {
    const int t = T - 1 - *c_rd;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one block = one 16 x 16 tile (rp_pos order)
    if (i >= (size_t)Bp * Fp) return;
    const int nft = Fp / 16;
    const int ft = (int)(blockIdx.x % nft), m = (int)(blockIdx.x / nft);
    const int pos = threadIdx.x;
    const int f = 16 * ft + 4 * (pos & 3) + (pos >> 6);
    const int row = m * 16 + ((pos >> 2) & 15);
    const size_t slab = (size_t)Bp * Fp;
    float sc = 0.f;
    if (f < F && valid[(size_t)t * Bp + row])
        sc = ista_neg_dg(div, beta, xp[(size_t)t * slab + i], xhat[((size_t)t * K + k) * slab + i]);
    const float dr = drpart[i];
    drpart[i] = sc != 0.f ? dr * sc : 0.f;
}
// R_k = g(X, X^_k) row-major [B*T][Fp] (padding bins 0) from the packed x^ of layer k
__global__ void __launch_bounds__(256)
unpack_g_kernel(const float* __restrict__ x, const float* __restrict__ xhat, float* __restrict__ R,
                const unsigned char* __restrict__ valid, int B, int T, int F, int Fp, int Bp, int k,
                int K, int div, float beta) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * T * Fp) return;
    const int f = (int)(i % Fp);
    const size_t bt = i / Fp;
    const int t = (int)(bt % T), b = (int)(bt / T);
    float g = 0.f;
    if (f < F && valid[(size_t)t * Bp + b]) {       // (masked frames: no gradient, and x^ may be 0 / 0)
        const float xe = xhat[((size_t)t * K + k) * Bp * Fp + ((size_t)(b >> 4) * (Fp / 16) + (f >> 4)) * 256 +
                              rp_pos(b & 15, f & 15)];
        g = ista_g(div, beta, x[bt * F + f], xe);
    }
    R[i] = g;
}
// state that entered frame t of row b: the previous (possibly repeated) output once a valid frame has
// been seen, else softplus(log_h0)
__global__ void __launch_bounds__(256)
state_matrix_kernel(const float* __restrict__ hall, const unsigned char* __restrict__ seen,
                    const float* __restrict__ log_h0, float* __restrict__ P, int B, int T, int N,
                    int K, int Bp, const float* __restrict__ init) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * T * N) return;
    const int n = (int)(i % N);
    const size_t bt = i / N;
    const int t = (int)(bt % T), b = (int)(bt / T);
    float v;
    if (seen[(size_t)t * Bp + b]) v = hall[(bt - 1) * K * N + (size_t)(K - 1) * N + n];
    else if (init) v = init[(size_t)b * N + n];
    else {
        const float z = log_h0[n];
        v = (z > 20.f) ? z : log1pf(expf(z));
    }
    P[i] = v;
}

struct BwdWs {
    size_t off_dstate, off_gq, off_dzp0, off_dzp1, off_dGp0, off_dGp1, off_drpart, off_z0s, off_dps,
        off_cnt, off_dh0, off_dzall, off_dR, off_xpad, off_R, off_dn, off_dpia, off_dnia, off_P, off_cr, off_dq,
        off_dqsum, off_pst, total;
};
BwdWs bwd_layout(const drnmf_cell_desc_t* d, const Workspace& W) {
    BwdWs L;
    size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += round_up_sz(b, 256); return at; };
    const size_t hp = (size_t)W.Bp * W.Np * 4;
    const int64_t BT = (int64_t)d->B * d->T;
    L.off_dstate = take(hp); L.off_gq = take(hp);
    L.off_dzp0 = take(hp); L.off_dzp1 = take(hp); L.off_dGp0 = take(hp); L.off_dGp1 = take(hp);
    L.off_drpart = take((size_t)MAX_KS * W.Bp * W.Fp * 4);
    const int nparts = W.gram ? W.numO : W.numA;
    L.off_z0s = take((size_t)2 * nparts * W.Bp * 4);
    L.off_dps = take((size_t)2 * nparts * W.Bp * 4);
    L.off_cnt = take(256 + 65536);   // frame counters + the persistent chains' sync lines
    L.off_dh0 = take((size_t)(W.Bp / ROWS) * W.Np * 4);
    L.off_dzall = take((size_t)BT * d->K * d->N * 4);
    // d r_k of every layer from the sequential pass; the Gram form recomputes one layer at a time
    L.off_dR = take((size_t)(W.gram ? 1 : d->K) * BT * W.Fp * 4);
    L.off_xpad = take((size_t)BT * W.Fp * 4);
    L.off_R = take((size_t)BT * W.Fp * 4);
    L.off_dn = take((size_t)W.Fp * W.Np * 4);
    L.off_dpia = take(W.gram ? 0 : (size_t)d->K * W.Fp * W.Np * 4);
    L.off_dnia = take(W.gram ? (size_t)W.Fp * W.Np * 4 : 0);
    {
        const bool odd = (d->F % 16 == 1) && d->F > 16 && d->divergence == DRNMF_DIV_ED;
        L.off_P = take((size_t)tn_splits(odd ? d->F - 1 : W.Fp, d->N, BT) * W.Fp * W.Np * 4);
    }
    L.off_cr = take((size_t)CR_SPLITS * CR_SLOTS * W.Np * 4);
    L.off_dq = take((size_t)2 * MAX_TAIL * W.Bp * W.numA * 4);
    L.off_dqsum = take((size_t)MAX_TAIL * W.Bp * 4);
    L.off_pst = take(d->divergence != DRNMF_DIV_ED ? (size_t)BT * d->N * 4 : 0);   // KL / beta: state matrix
    L.total = o;
    return L;
}

}  // namespace

extern "C" size_t drnmf_cell_backward_workspace_bytes(const drnmf_cell_desc_t* d) {
    if (!d || d->B <= 0 || d->T <= 0 || d->F <= 0 || d->N <= 0 || d->K <= 0) return 0;
    return bwd_layout(d, workspace_layout(d)).total;
}

static int32_t cell_backward_impl(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                  const void* params, const float* log_h0, float u0_diag,
                                  float u0_off, float uk_off, const float* hall,
                                  const float* d_out, const void* fwd_workspace,
                                  size_t fwd_workspace_bytes, void* bwd_workspace,
                                  size_t bwd_workspace_bytes, float* d_log_D,
                                  float* d_log_alph, float* d_log_lam1, float* d_log_h0,
                                  void* stream_, float* prof_ms, float beta = 0.f,
                                  const float* initial_state = nullptr) {
    if (!h) return DRNMF_ERR_INVALID_ARG;
    // (No implicit look at the handle's fault word here: a host-side read-and-clear at the entry of the
    // NEXT call races with the stream-ordered drnmf_status_take_device of a training step -- the host
    // could consume the word before the device-side guard of the fused Adam launch has seen it.  Faults
    // are reported by drnmf_check_status / drnmf_status_take_device only.)
    int rc = validate_cell_desc(h, d);
    if (rc) return rc;
    // operand_f16: the forward ran on fp16 matrix-core operands; its BPTT is computed in fp32 from
    // the stored hiddens and the fp32 dictionary packings kept in the same prepared block
    // (rounding treated as the identity: mixed-precision training)
    // KL / beta cell (drnmf_cell_backward_ista): every layer is a full ISTA step from its input (layer
    // 0 from the state), no U term -- the same chain with one more layer-step per frame, the residual
    // map's derivative between its two launches, and u0_diag = 1, u0_off = uk_off = 0
    const bool nonlin = d->divergence != DRNMF_DIV_ED;
    if (nonlin) { u0_diag = 1.f; u0_off = 0.f; uk_off = 0.f; }
    const int kmin = nonlin ? 0 : 1;                     // last layer-step of a frame's chain
    if (!d->return_all_hidden)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG,
                   "cell_backward needs the forward run with return_all_hidden=1 (all K hiddens)");
    if (!x || !params || !log_h0 || !hall || !d_out || !fwd_workspace || !bwd_workspace ||
        !d_log_D || !d_log_alph || !d_log_lam1 || !d_log_h0)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "cell_backward: NULL pointer argument");
    const Workspace W = workspace_layout(d);
    const BwdWs L = bwd_layout(d, W);
    if (nonlin && (W.off_xhat == 0 || W.KS != 1 || W.gram || d->operand_f16))
        DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED, "cell_backward (KL / beta): needs the fp32 training forward "
                   "of drnmf_cell_forward_ista (return_all_hidden = 1)");
    if (fwd_workspace_bytes < W.total || bwd_workspace_bytes < L.total)
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "cell_backward: workspace too small (fwd %zu/%zu, bwd "
                   "%zu/%zu)", fwd_workspace_bytes, W.total, bwd_workspace_bytes, L.total);
    if (((uintptr_t)bwd_workspace & 255) || ((uintptr_t)fwd_workspace & 255))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "workspaces must be 256-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const ParamsLayout PL = params_layout(d);
    const char* pb = (const char*)params;
    const char* fw = (const char*)fwd_workspace;
    char* bw = (char*)bwd_workspace;
    const int K = d->K, N = d->N, F = d->F, T = d->T, B = d->B;
    const int64_t BT = (int64_t)B * T;
    // MFMA bin tiles: the odd bins of a 2^k+1 STFT are handled by rank-1 terms, as in the forward
    const int numM = W.Bp / ROWS, nft = W.nft_main;
    const bool half = d->operand_f16 != 0;
    const float* Dp_base = (const float*)(pb + (half ? PL.off_dn32 : PL.off_dn));
    const size_t dstride = (size_t)PL.Fp * PL.Np;
    const float* DpA_base = half ? Dp_base + (size_t)d->n_D * dstride : (const float*)(pb + PL.off_dnA);
    auto Dp_of = [&](int k) { return Dp_base + (d->n_D == 1 ? 0 : (size_t)k * dstride); };
    auto ia_of = [&](int k) { return (const float*)(pb + PL.off_inv_alpha) + (size_t)k * PL.Np; };
    auto b_of = [&](int k) { return (const float*)(pb + PL.off_bias) + (size_t)k * PL.Np; };
    const unsigned char* valid = (const unsigned char*)(fw + W.off_valid);
    const unsigned char* seen = (const unsigned char*)(fw + W.off_seen);
    const float* psum_all = (const float*)(fw + W.off_psum_all);

    float* dstate = (float*)(bw + L.off_dstate);
    float* gq = (float*)(bw + L.off_gq);
    float* dzp[2] = {(float*)(bw + L.off_dzp0), (float*)(bw + L.off_dzp1)};
    float* dGp[2] = {(float*)(bw + L.off_dGp0), (float*)(bw + L.off_dGp1)};
    float* drpart = (float*)(bw + L.off_drpart);
    float* z0s = (float*)(bw + L.off_z0s);
    float* dps = (float*)(bw + L.off_dps);
    int* cA = (int*)(bw + L.off_cnt);
    int* cB = cA + 16;
    float* dh0_part = (float*)(bw + L.off_dh0);
    float* dz_all = (float*)(bw + L.off_dzall);
    float* dR_all = (float*)(bw + L.off_dR);
    float* xpad = (float*)(bw + L.off_xpad);
    float* Rk = (float*)(bw + L.off_R);
    float* Dn_rm = (float*)(bw + L.off_dn);
    float* P = (float*)(bw + L.off_P);
    float* crp = (float*)(bw + L.off_cr);

    // measurement aid (drnmf_cell_backward_profile): HIP events at the phase boundaries
    hipEvent_t pev[3] = {nullptr, nullptr, nullptr};
    if (prof_ms) {
        for (auto& e : pev) DRNMF_HIP(h, hipEventCreate(&e));
        DRNMF_HIP(h, hipEventRecord(pev[0], stream));
    }
    // ---- sequential pass: T replays of the per-frame graph in reverse time ---------------------
    DRNMF_HIP(h, hipMemsetAsync(bw, 0, L.off_dh0, stream));   // dstate .. counters
    const dim3 grid_a(8u * (unsigned)numM, (unsigned)(round_up(W.numA, 8) / 8));
    const dim3 grid_b(8u * (unsigned)(numM / W.RB), (unsigned)(round_up(nft * W.KS, 8) / 8));

    EdgeArgs ea;
    ea.hall = hall; ea.d_out = d_out; ea.dz_all = dz_all; ea.ia_last = ia_of(K - 1);
    ea.dstate = dstate; ea.gq = gq; ea.dzp_top = dzp[(K - 1) & 1]; ea.dGp_top = W.gram ? dGp[(K - 1) & 1] : nullptr;
    ea.dzp0 = nonlin ? dzp[1] : dzp[0];      // (KL / beta: layer 0's launch leaves d p in dzp[(0 - 1) & 1])
    ea.dz0s_part = z0s; ea.dps_part = dps; ea.dh0_part = dh0_part;
    ea.valid = valid;
    const bool edge_only = K == 1 && !nonlin;            // a frame of the ED cell with K = 1: the edge alone
    ea.c_rd = edge_only ? cA : cB;
    ea.c_wr = edge_only ? nullptr : cA;
    ea.u0d = u0_diag; ea.u0o = u0_off;
    ea.B = B; ea.T = T; ea.N = N; ea.K = K; ea.Bp = W.Bp; ea.Np = W.Np; ea.numA = W.numA;
    auto tail_of = [&](int k) {
        return (const float*)(pb + PL.off_tail) + (d->n_D == 1 ? 0 : (size_t)k * MAX_TAIL * PL.Np);
    };
    float* dq = (float*)(bw + L.off_dq);
    const size_t dqstride = (size_t)MAX_TAIL * W.Bp * W.numA;
    ea.Dtail_top = tail_of(K - 1);
    ea.dq_out = dq + (size_t)((K - 1) & 1) * dqstride;   // partials of layer k live in buffer k & 1
    ea.ntail = W.gram ? 0 : W.ntail;
    ea.nparts = W.gram ? W.numO : W.numA;
    ea.ppb = W.gram ? 2 : 1;
    auto make_g = [&](int k) {       // Gram form: one launch per layer-step (cell_gram.h)
        GramBwdArgs g;
        memset(&g, 0, sizeof(g));
        g.G = (const float*)(pb + PL.off_gram) + (d->n_D == 1 ? 0 : (size_t)k * PL.Np * PL.Np);
        g.dGp_in = dGp[k & 1];
        g.dzp_in = dzp[k & 1];
        g.ia_prev = ia_of(k - 1);
        g.dzp_out = dzp[(k - 1) & 1];
        g.dGp_out = dGp[(k - 1) & 1];
        g.hall = hall; g.dz_all = dz_all;
        g.dz0s_part = z0s; g.dps_part = dps;
        g.c_rd = cA;
        g.c_wr = (k == 1) ? cB : nullptr;
        g.uko = uk_off;
        g.k = k; g.B = B; g.T = T; g.N = N; g.K = K; g.Bp = W.Bp; g.Np = W.Np; g.numO = W.numO;
        return g;
    };
    const dim3 grid_g(8u * (unsigned)numM, (unsigned)(round_up(W.numO, 8) / 8));

    // (the fp16-operand forward counts 32-atom chunks; the fp32 kernels of this pass 16-atom ones)
    const int nch_ks_b = half ? W.Np / 16 : W.nch_ks;
    const bool qred = qred_wanted(W.numA, W.ntail, nft, W.KS, W.RB);
    float* DpIa = (float*)(bw + L.off_dpia);
    if (!W.gram) {
        const size_t tot = (size_t)W.Fp * W.Np;
        for (int k = kmin; k < K; ++k)
            hipLaunchKernelGGL(scale_pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                               stream, Dp_of(k), ia_of(k), DpIa + (size_t)k * tot, tot, W.Np);
    }
    auto make_b = [&](int k) {
        CellBArgs b;
        b.Dn_next = DpIa + (size_t)k * W.Fp * W.Np;    // Dn_k with column n scaled by ia_k[n]
        b.h = dzp[k & 1];
        b.xp = nullptr;
        b.rpart = drpart;
        b.t_rd = cA;
        b.Bp = W.Bp; b.Fp = W.Fp; b.Np = W.Np; b.nft = nft; b.KS = W.KS; b.nch_ks = nch_ks_b;
        b.q_in = dq + (size_t)(k & 1) * dqstride;          // partials of layer k (bwd_edge / bwd_a)
        b.qsum = (float*)(bw + L.off_dqsum);
        b.numA = W.numA; b.ntail = W.ntail;
        b.logKS = 0;
        while ((1 << b.logKS) < W.KS) ++b.logKS;
        return b;
    };
    auto make_a = [&](int k) {
        BwdAArgs a;
        a.Dn = DpA_base + (d->n_D == 1 ? 0 : (size_t)k * dstride);
        a.ia_prev = ia_of(k >= 1 ? k - 1 : 0);
        a.drpart = drpart;
        a.dzp_in = dzp[k & 1];
        a.dzp_out = dzp[(k - 1) & 1];
        a.dGp_out = nullptr;
        a.hall = hall; a.dz_all = dz_all;
        a.dR = dR_all + (size_t)k * BT * W.Fp;
        a.dz0s_part = z0s; a.dps_part = dps;
        a.c_rd = cA;
        a.c_wr = (k == kmin) ? cB : nullptr;
        a.uko = uk_off;
        a.k = k; a.B = B; a.T = T; a.N = N; a.K = K; a.Bp = W.Bp; a.Fp = W.Fp; a.Np = W.Np;
        a.numA = W.numA; a.nchunks = nft;
        a.Dtail = tail_of(k);
        a.Dtail_prev = tail_of(k >= 1 ? k - 1 : 0);
        a.dq_in = qred ? (const float*)(bw + L.off_dqsum) : dq + (size_t)(k & 1) * dqstride;
        a.dq_out = dq + (size_t)((k - 1) & 1) * dqstride;
        a.ntail = W.ntail;
        return a;
    };

    // Gram form with few tiles: the whole sequential pass as ONE launch of independent persistent
    // chains, one per row tile (cell_gram_persist.h); same arithmetic as the graphs below.
    const bool persist = W.gram && persist_shape_ok(h, numM, W.numO, K) && persist_admit(h, stream);
    if (persist) {
        GramPersistBwdArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.G = (const float*)(pb + PL.off_gram);
        pa.g_stride = d->n_D == 1 ? 0 : (size_t)PL.Np * PL.Np;
        pa.ia = (const float*)(pb + PL.off_inv_alpha);
        pa.hall = hall; pa.d_out = d_out; pa.dz_all = dz_all;
        pa.dGp[0] = dGp[0]; pa.dGp[1] = dGp[1];
        pa.z0s_part = z0s; pa.dps_part = dps; pa.dh0_part = dh0_part; pa.dstate = dstate;
        pa.valid = valid;
        pa.bar = (unsigned*)(bw + L.off_cnt + 256);            // (zeroed by the memset above)
        pa.host_flag = h->persist_flag;
        pa.u0d = u0_diag; pa.u0o = u0_off; pa.uko = uk_off;
        pa.B = B; pa.T = T; pa.N = N; pa.K = K; pa.Bp = W.Bp; pa.Np = W.Np; pa.numO = W.numO; pa.numM = numM;
        pa.nwait = persist_nwait(W.numO);
        void* kp[1] = {&pa};
        DRNMF_HIP(h, hipLaunchKernel(pick_persist_bwd(W.Np / 16), dim3(8u * (unsigned)(W.numO * persist_rounds(numM))),
                                     dim3(64 * (NW_G + 1)), kp, 0, stream));
        persist_mark(h, stream);
    }
    uint32_t beta_bits;
    memcpy(&beta_bits, &beta, 4);
    std::vector<uint64_t> key = {(W.gram ? 0xB00Cull : 0xB00Bull) + ((uint64_t)d->divergence << 16) +
                                     ((uint64_t)beta_bits << 32), (uint64_t)B, (uint64_t)T, (uint64_t)F, (uint64_t)N,
                                 (uint64_t)K, (uint64_t)d->n_D, (uint64_t)(uintptr_t)params,
                                 (uint64_t)(uintptr_t)hall, (uint64_t)(uintptr_t)d_out,
                                 (uint64_t)(uintptr_t)fwd_workspace,
                                 (uint64_t)(uintptr_t)bwd_workspace};
    {
        uint32_t b0, b1, b2;
        memcpy(&b0, &u0_diag, 4); memcpy(&b1, &u0_off, 4); memcpy(&b2, &uk_off, 4);
        key.push_back(b0); key.push_back(b1); key.push_back(b2);
    }
    // several frames per graph, as in the forward (cell_forward.hip)
    int fpg_max = 800 / (2 * K - 1);         // ~800 kernel nodes per graph
    fpg_max = fpg_max < 1 ? 1 : (fpg_max > 64 ? 64 : fpg_max);
    if (fpg_max > T) fpg_max = T;
    auto get_graph = [&](int fpg, hipGraphExec_t* out) -> int32_t {
    std::vector<uint64_t> gkey = key;
    gkey.push_back((uint64_t)fpg);
    GraphEntry* entry = nullptr;
    for (auto& g : h->graphs)
        if (g.key == gkey) { entry = &g; break; }
    if (!entry) {
        {   // bounded cache: the oldest entry is retired without synchronising (common.h)
            const int32_t erc = graph_cache_make_room(h, stream, 24);
            if (erc) return erc;
        }
        GraphEntry ge;
        ge.key = gkey;
        DRNMF_HIP(h, hipGraphCreate(&ge.graph, 0));
        hipGraphNode_t last = nullptr;
        auto add = [&](void* func, dim3 grid, unsigned block, void** kp) -> hipError_t {
            hipKernelNodeParams p;
            memset(&p, 0, sizeof(p));
            p.func = func; p.gridDim = grid; p.blockDim = dim3(block);
            p.sharedMemBytes = 0; p.kernelParams = kp; p.extra = nullptr;
            hipGraphNode_t node;
            hipError_t e = hipGraphAddKernelNode(&node, ge.graph, last ? &last : nullptr,
                                                 last ? 1 : 0, &p);
            last = node;
            return e;
        };
        for (int rep = 0; rep < fpg; ++rep) {
            void* ke[1] = {&ea};
            DRNMF_HIP(h, add((void*)&bwd_edge_kernel, grid_a, 256, ke));
            for (int k = K - 1; k >= kmin; --k) {
                if (W.gram) {
                    GramBwdArgs g = make_g(k);
                    void* kg[1] = {&g};
                    DRNMF_HIP(h, add(pick_gram_bwd(W.Np / 16), grid_g, 64 * NW_G, kg));
                    continue;
                }
                CellBArgs b = make_b(k);
                DRNMF_HIP(h, add(pick_b_func(nch_ks_b, W.RB, false, qred), grid_b, 64 * NW_B, CellBParams(b).p));
                if (nonlin) {                          // d x^ = d r * dg/dx^ (x_t, x^_{t,k}), in place
                    const float* xpp = (const float*)(fw + W.off_xp);
                    const float* xh = (const float*)(fw + W.off_xhat);
                    float* drp = drpart;
                    const int* crd = cA;
                    const unsigned char* vp = valid;
                    int Tv = T, kv = k, Kv = K, dv = d->divergence, Fv = F, Fpv = W.Fp, Bpv = W.Bp;
                    float bt = beta;
                    void* ks[13] = {&xpp, &xh, &drp, &crd, &vp, &Tv, &kv, &Kv, &dv, &bt, &Fv, &Fpv, &Bpv};
                    DRNMF_HIP(h, add((void*)&dgdx_scale_kernel,
                                     dim3((unsigned)((size_t)W.Bp * W.Fp / 256)), 256, ks));
                }
                BwdAArgs a = make_a(k);
                DRNMF_HIP(h, add(pick_bwd_a(nft, W.KS, qred, nonlin && k == 0), grid_a, 256, BwdAParams(a).p));
            }
            if (edge_only) {
                int* cp = cA;
                void* kc[1] = {&cp};
                DRNMF_HIP(h, add((void*)&advance_frame_kernel, dim3(1), 1, kc));
            }
        }
        DRNMF_HIP(h, hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0));
        h->graphs.push_back(ge);
        entry = &h->graphs.back();
    }
    entry->last_stream = stream;
    *out = entry->exec;
    return DRNMF_OK;
    };
    if (!persist) {
        hipGraphExec_t exec_n = nullptr, exec_1 = nullptr;
        int32_t grc = get_graph(fpg_max, &exec_n);
        if (grc) return grc;
        int t = 0;
        for (; t + fpg_max <= T; t += fpg_max) DRNMF_HIP(h, hipGraphLaunch(exec_n, stream));
        if (t < T) {
            grc = get_graph(1, &exec_1);
            if (grc) return grc;
            for (; t < T; ++t) DRNMF_HIP(h, hipGraphLaunch(exec_1, stream));
        }
    }
    if (!persist) hipLaunchKernelGGL(bwd_edge_kernel, grid_a, dim3(256), 0, stream, ea);   // t = -1
    hipLaunchKernelGGL(dlogh0_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, dh0_part,
                       log_h0, d_log_h0, N, W.Np, numM, initial_state != nullptr ? 1 : 0);
    DRNMF_HIP(h, hipGetLastError());
    if (prof_ms) DRNMF_HIP(h, hipEventRecord(pev[1], stream));

    // ---- time-batched phase ----------------------------------------------------------------------
    // (Round 5 ran these products in blocks of 256 / 512 frames on a side stream BESIDE the sequential pass --
    // gemm_tn over row segments of every utterance, one set of split partials per layer -- and the headline
    // step went from 1014 to 1091 / 1063 ms: profiles/r05_overlap_negative.txt, the patch beside it.)
    DRNMF_HIP(h, hipMemsetAsync(xpad, 0, (size_t)BT * W.Fp * 4, stream));
    DRNMF_HIP(h, hipMemcpy2DAsync(xpad, (size_t)W.Fp * 4, x, (size_t)F * 4, (size_t)F * 4,
                                  (size_t)BT, hipMemcpyDeviceToDevice, stream));
    if (W.Fp != F) DRNMF_HIP(h, hipMemsetAsync(Rk, 0, (size_t)BT * W.Fp * 4, stream));
    const size_t pstr = (size_t)W.Fp * W.Np;
    const int KN = K * N;
    for (int k = 0; k < K; ++k) {
        const size_t tot = (size_t)W.Fp * W.Np;
        hipLaunchKernelGGL(unpack_dn_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                           stream, Dp_of(k), Dn_rm, W.Fp, W.Np);
        const float* Aop = xpad;
        const int perm_all = 16 * nft;       // the chain kernels' saved copies: every MFMA bin tile
        int perm1 = 0;
        if (nonlin) {
            // R_k = g(X, X^_k) of EVERY layer, layer 0 included, from the x^ the forward kept
            const size_t tot2 = (size_t)BT * W.Fp;
            hipLaunchKernelGGL(unpack_g_kernel, dim3((unsigned)((tot2 + 255) / 256)), dim3(256), 0, stream,
                               x, (const float*)(fw + W.off_xhat), Rk, valid, B, T, F, W.Fp, W.Bp, k, K,
                               d->divergence, beta);
            Aop = Rk;
        } else if (k >= 1 && W.off_rsave != 0) {
            // the forward left r_k of every frame in its workspace (cell_a_kernel, Rsave)
            Aop = (const float*)(fw + W.off_rsave) + (size_t)(k - 1) * BT * W.Fp;
            perm1 = perm_all;
        } else if (k >= 1) {
            gemm::Operands g1{hall + (size_t)(k - 1) * N, Dn_rm, BT, F, N, KN, W.Np};
            DRNMF_HIP(h, gemm::launch(g1, EpiResid{x, Rk, F, W.Fp}, stream));
            Aop = Rk;
        }
        // M = Fp: the padded bins of R / X / dR are zero, and whole 4-column groups keep the loads
        // vectorised.  F = 16 j + 1: M = F - 1, the odd bin's row comes from colreduce_kernel.
        const bool odd = (F % 16 == 1) && F > 16 && !nonlin;   // (KL / beta: no odd-bin side path at all)
        const int Mg = odd ? F - 1 : W.Fp;
        gemm_tn::Operands t1{Aop, dz_all + (size_t)k * N, BT, Mg, N, W.Fp, KN};
        const int nsplit = tn_splits(Mg, N, BT);
        const float* dRk_tail = nullptr;
        DRNMF_HIP(h, gemm_tn::launch(t1, EpiP1{P, ia_of(k), W.Np, pstr, perm1}, nsplit, stream));
        if (k >= 1) {
            const float* dRk = dR_all + (size_t)k * BT * W.Fp;
            if (W.gram) {
                // the Gram chain never forms d r_k = (dz_k / alpha_k) Dn_k^T: one more frame-parallel
                // product, against the dictionary scaled by 1/alpha_k
                float* DnIa = (float*)(bw + L.off_dnia);
                hipLaunchKernelGGL(unpack_scaled_kernel, dim3((unsigned)((tot + 255) / 256)),
                                   dim3(256), 0, stream, Dp_of(k), ia_of(k), DnIa, W.Fp, W.Np);
                gemm::Operands g2{dz_all + (size_t)k * N, DnIa, BT, W.Fp, N, KN, W.Np};
                DRNMF_HIP(h, gemm::launch(g2, EpiStore{dR_all, W.Fp}, stream));
                dRk = dR_all;
            }
            gemm_tn::Operands t2{dRk, hall + (size_t)(k - 1) * N, BT, Mg, N, W.Fp, KN};
            DRNMF_HIP(h, gemm_tn::launch(t2, EpiP2{P, W.Np, pstr, W.gram ? 0 : perm_all}, nsplit, stream));
            dRk_tail = dRk + (F - 1);
        } else if (nonlin) {
            // layer 0 of the KL / beta cell contracts the STATE p_t: - d x^_0^T P, P row-major [B*T][N]
            float* Pst = (float*)(bw + L.off_pst);
            const size_t totp = (size_t)BT * N;
            hipLaunchKernelGGL(state_matrix_kernel, dim3((unsigned)((totp + 255) / 256)), dim3(256), 0,
                               stream, hall, seen, log_h0, Pst, B, T, N, K, W.Bp, initial_state);
            gemm_tn::Operands t2{dR_all, Pst, BT, Mg, N, W.Fp, N};
            DRNMF_HIP(h, gemm_tn::launch(t2, EpiP2{P, W.Np, pstr, perm_all}, nsplit, stream));
        }
        ColRedArgs ca;
        ca.rt = odd ? Aop + (F - 1) : nullptr;
        ca.drt = dRk_tail;
        ca.ldr = W.Fp;
        ca.hall = hall; ca.dz_all = dz_all; ca.bias = b_of(k); ca.psum_all = psum_all;
        ca.seen = seen; ca.log_h0 = log_h0; ca.init = initial_state; ca.part = crp;
        ca.u0d = u0_diag; ca.u0o = u0_off; ca.uko = uk_off;
        ca.k = k; ca.B = B; ca.T = T; ca.N = N; ca.K = K; ca.Bp = W.Bp; ca.Np = W.Np;
        if (N % 4 == 0 && N / 4 <= 64)
            hipLaunchKernelGGL(colreduce4_rows_kernel, dim3(1, CR_SPLITS), dim3(256), 0, stream, ca);
        else if (N % 4 == 0)   // (Np, K*N, the buffers' bases: multiples of 4 floats then)
            hipLaunchKernelGGL(colreduce4_kernel, dim3((N / 4 + 255) / 256, CR_SPLITS), dim3(256), 0,
                               stream, ca);
        else
            hipLaunchKernelGGL(colreduce_kernel, dim3((N + 255) / 256, CR_SPLITS), dim3(256), 0,
                               stream, ca);
        const int ka = d->n_alph == 1 ? 0 : k, kl = d->n_lam == 1 ? 0 : k;
        if (N <= 512)
            hipLaunchKernelGGL(colreduce_fold_lanes_kernel, dim3((N + 63) / 64), dim3(256), 0, stream,
                               crp, N, W.Np, odd ? P + (size_t)(F - 1) * W.Np : (float*)nullptr,
                               ia_of(k), nsplit, pstr);
        else
            hipLaunchKernelGGL(colreduce_fold_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, crp,
                               N, W.Np, odd ? P + (size_t)(F - 1) * W.Np : (float*)nullptr, ia_of(k),
                               nsplit, pstr);
        const int kd = d->n_D == 1 ? 0 : k;
        if (N % 4 == 0) {
            // (cpart: the split >= 2 slots of the column-reduction partials, dead after the fold)
            float* cpart = crp + (size_t)2 * CR_SLOTS * W.Np;
            // 16 bins per workgroup, or 4 where that leaves fewer than 128 workgroups
            const int bpt = ((N / 4 + 63) / 64) * ((F + 15) / 16) < 128 ? 1 : 4;
            const int ngroups = (F + 4 * bpt - 1) / (4 * bpt);
            const dim3 dgrid((N / 4 + 63) / 64, ngroups);
            hipLaunchKernelGGL(dlogd_sum_kernel, dgrid, dim3(256), 0, stream, P, Dn_rm, cpart, F, N,
                               W.Np, nsplit, pstr, bpt);
            hipLaunchKernelGGL(dlogd_apply_kernel, dgrid, dim3(256), 0, stream, P, Dn_rm, cpart,
                               d_log_D + (size_t)kd * F * N, F, N, W.Np, ngroups,
                               (d->n_D == 1 && k > 0) ? 1 : 0, bpt);
        } else {
            hipLaunchKernelGGL(dlogd_kernel, dim3((N + 31) / 32), dim3(256), 0, stream, P, Dn_rm,
                               d_log_D + (size_t)kd * F * N, F, N, W.Np, nsplit, pstr,
                               (d->n_D == 1 && k > 0) ? 1 : 0);
        }
        hipLaunchKernelGGL(scalar_grads_kernel, dim3(1), dim3(256), 0, stream, crp, b_of(k),
                           d_log_alph + (size_t)ka * d->alph_len, d_log_lam1 + kl, N, W.Np,
                           d->alph_len, (d->n_alph == 1 && k > 0) ? 1 : 0,
                           (d->n_lam == 1 && k > 0) ? 1 : 0);
    }
    DRNMF_HIP(h, hipGetLastError());
    if (prof_ms) {
        DRNMF_HIP(h, hipEventRecord(pev[2], stream));
        DRNMF_HIP(h, hipStreamSynchronize(stream));
        DRNMF_HIP(h, hipEventElapsedTime(&prof_ms[0], pev[0], pev[1]));
        DRNMF_HIP(h, hipEventElapsedTime(&prof_ms[1], pev[1], pev[2]));
        prof_ms[2] = (float)T * (float)(W.gram ? K : (nonlin ? 3 * K + 1 : 2 * K - 1)) + 1.f;   // launches of the sequential pass
        for (auto& e : pev) (void)hipEventDestroy(e);
    }
    return DRNMF_OK;
}

extern "C" int32_t drnmf_cell_backward(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                       const void* params, const float* log_h0, float u0_diag,
                                       float u0_off, float uk_off, const float* hall,
                                       const float* d_out, const void* fwd_workspace,
                                       size_t fwd_workspace_bytes, void* bwd_workspace,
                                       size_t bwd_workspace_bytes, float* d_log_D,
                                       float* d_log_alph, float* d_log_lam1, float* d_log_h0,
                                       void* stream_) {
    DRNMF_LOCK(h);
    if (h && d && d->divergence != DRNMF_DIV_ED)
        DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED,
                   "cell_backward is the reference's (Euclidean) cell; KL / beta: drnmf_cell_backward_ista");
    return cell_backward_impl(h, d, x, params, log_h0, u0_diag, u0_off, uk_off, hall, d_out,
                              fwd_workspace, fwd_workspace_bytes, bwd_workspace, bwd_workspace_bytes,
                              d_log_D, d_log_alph, d_log_lam1, d_log_h0, stream_, nullptr);
}

// Stateful training (custom_layers.py:296-318; Keras Recurrent stateful=True under fit / train_on_batch):
// the forward was drnmf_cell_forward_stateful with return_all_hidden = 1, every sequence entered with
// initial_state[b] -- a CONSTANT of the gradient (Keras does not backpropagate into the previous batch).
// Same BPTT; the state that enters a row's first valid frame is initial_state[b] instead of
// softplus(log_h0) wherever the weight gradients need it, and d_log_h0 = 0.  initial_state NULL = drnmf_cell_backward.
extern "C" int32_t drnmf_cell_backward_stateful(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                                const void* params, const float* log_h0, float u0_diag,
                                                float u0_off, float uk_off, const float* initial_state,
                                                const float* hall, const float* d_out,
                                                const void* fwd_workspace, size_t fwd_workspace_bytes,
                                                void* bwd_workspace, size_t bwd_workspace_bytes,
                                                float* d_log_D, float* d_log_alph, float* d_log_lam1,
                                                float* d_log_h0, void* stream_) {
    DRNMF_LOCK(h);
    if (h && d && d->divergence != DRNMF_DIV_ED)
        DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED, "cell_backward_stateful is the reference's (Euclidean) cell");
    if (h && initial_state && ((uintptr_t)initial_state & 15))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "cell_backward_stateful: initial_state must be 16-byte aligned");
    return cell_backward_impl(h, d, x, params, log_h0, u0_diag, u0_off, uk_off, hall, d_out,
                              fwd_workspace, fwd_workspace_bytes, bwd_workspace, bwd_workspace_bytes,
                              d_log_D, d_log_alph, d_log_lam1, d_log_h0, stream_, nullptr, 0.f, initial_state);
}

extern "C" int32_t drnmf_cell_backward_ista(drnmf_handle_t h, const drnmf_cell_desc_t* d,
                                            const float* x, const void* params, const float* log_h0,
                                            float beta, const float* hall, const float* d_out,
                                            const void* fwd_workspace, size_t fwd_workspace_bytes,
                                            void* bwd_workspace, size_t bwd_workspace_bytes,
                                            float* d_log_D, float* d_log_alph, float* d_log_lam1,
                                            float* d_log_h0, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (!d || (d->divergence != DRNMF_DIV_KL && d->divergence != DRNMF_DIV_BETA))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG,
                   "cell_backward_ista: d->divergence must be DRNMF_DIV_KL or DRNMF_DIV_BETA");
    return cell_backward_impl(h, d, x, params, log_h0, 1.f, 0.f, 0.f, hall, d_out, fwd_workspace,
                              fwd_workspace_bytes, bwd_workspace, bwd_workspace_bytes, d_log_D,
                              d_log_alph, d_log_lam1, d_log_h0, stream_, nullptr, beta);
}

// ... and of a stateful KL / beta cell (drnmf_cell_forward_ista with an initial state, return_all_hidden = 1)
extern "C" int32_t drnmf_cell_backward_ista_stateful(drnmf_handle_t h, const drnmf_cell_desc_t* d,
                                                     const float* x, const void* params, const float* log_h0,
                                                     float beta, const float* initial_state, const float* hall,
                                                     const float* d_out, const void* fwd_workspace,
                                                     size_t fwd_workspace_bytes, void* bwd_workspace,
                                                     size_t bwd_workspace_bytes, float* d_log_D,
                                                     float* d_log_alph, float* d_log_lam1, float* d_log_h0,
                                                     void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (!d || (d->divergence != DRNMF_DIV_KL && d->divergence != DRNMF_DIV_BETA))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG,
                   "cell_backward_ista_stateful: d->divergence must be DRNMF_DIV_KL or DRNMF_DIV_BETA");
    if (initial_state && ((uintptr_t)initial_state & 15))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "cell_backward_ista_stateful: initial_state must be 16-byte aligned");
    return cell_backward_impl(h, d, x, params, log_h0, 1.f, 0.f, 0.f, hall, d_out, fwd_workspace,
                              fwd_workspace_bytes, bwd_workspace, bwd_workspace_bytes, d_log_D,
                              d_log_alph, d_log_lam1, d_log_h0, stream_, nullptr, beta, initial_state);
}

extern "C" int32_t drnmf_cell_backward_profile(
    drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x, const void* params,
    const float* log_h0, float u0_diag, float u0_off, float uk_off, const float* hall,
    const float* d_out, const void* fwd_workspace, size_t fwd_workspace_bytes, void* bwd_workspace,
    size_t bwd_workspace_bytes, float* d_log_D, float* d_log_alph, float* d_log_lam1,
    float* d_log_h0, void* stream_, float* out_ms_host) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (!out_ms_host) DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "cell_backward_profile: NULL out_ms_host");
    if (d && d->divergence != DRNMF_DIV_ED)
        DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED, "cell_backward_profile brackets the Euclidean cell's BPTT");
    return cell_backward_impl(h, d, x, params, log_h0, u0_diag, u0_off, uk_off, hall, d_out,
                              fwd_workspace, fwd_workspace_bytes, bwd_workspace, bwd_workspace_bytes,
                              d_log_D, d_log_alph, d_log_lam1, d_log_h0, stream_, out_ms_host);
}
