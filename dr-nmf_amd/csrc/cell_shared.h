// Pieces of the recurrent cell shared by the forward (cell_forward.hip) and backward
// (cell_backward.hip) translation units: tile geometry, the bin-tile contraction kernel
// (x^ = h Dn^T in the forward, d r = dG Dn^T in the backward) and the workspace layout.
#pragma once
#include "common.h"

// Measurement aid, compiled only with -DDRNMF_TIMELINE (build.py: DRNMF_TIMELINE=1): s_memtime stamps
// of wave 0 of every workgroup of the LAST launch of cell_b / cell_a, read back with
// drnmf_debug_timeline (tools/timeline.py).
#ifdef DRNMF_TIMELINE
__device__ unsigned long long g_timeline[2][1024][8];
#define DRNMF_STAMP(kid, slot)                                                               \
    do {                                                                                     \
        if (threadIdx.x == 0)                                                                \
            g_timeline[kid][(blockIdx.y * gridDim.x + blockIdx.x) & 1023][slot] =           \
                __builtin_amdgcn_s_memtime();                                                \
    } while (0)
#else
#define DRNMF_STAMP(kid, slot) do { } while (0)
#endif

namespace {

constexpr int ROWS = 16;    // batch rows per workgroup (one MFMA M tile)
constexpr int ATOMS = 32;   // atoms per cell_a workgroup
constexpr int MAX_KS = 8;
constexpr int MAX_SPLIT = 8;   // sub-batches of a large inference batch (Workspace::split)
constexpr int NW_B = 8;    // waves per cell_b workgroup (more requests in flight per CU)
constexpr int NW_A = 4;    // waves per cell_a workgroup (8 measured slower, twice: 224k vs 232k, later 259k vs 277k frames/s)

struct CellBArgs {
    const void* Dn_next;     // packed dictionary of the next layer (fp32 Dp, or the fp16 DpB packing)
    const float* h;          // [Bp][Np]  this layer's h (fp16 mode: the fp16 copy Hp16, see below)
    const float* xp;         // [Bp][Fp] packed x of the CURRENT frame (republished by layer 0's cell_a,
                             // so this kernel needs no frame index); NULL: rpart[ks] = +acc (backward)
    float* rpart;            // [KS][Bp][Fp]
    const int* t_rd;
    int Bp, Fp, Np, nft, KS, logKS, nch_ks;   // nch_ks = 16-atom chunks per atom range
#ifdef DRNMF_MEASURE
    int ablate = 0;          // measurement aid (DRNMF_ABLATE_B), see cell_b_kernel
#endif
    // odd bins: the per-atom-block partial dot products the producing cell_a / bwd_a launch left
    // ([MAX_TAIL][Bp][numA]) are summed HERE, by the first workgroup of every row tile group, into
    // qsum [MAX_TAIL][Bp]: the consuming launch then loads one value per row instead of numA (at
    // N = 8000, numA = 250, those loads were 1.7 us of texture-path time on the tail of every cell_a)
    const float* q_in = nullptr;
    float* qsum = nullptr;
    int numA = 0, ntail = 0;
};

// x^ partial of one (row tile group, bin tile, atom range) and the residual partial
//   rpart[ks] = (ks == 0 ? x_t : 0) - h[16*RB x range] . Dn_next[16 bins x range]^T.
// GB = 16-atom chunks per wave per group.  RB = 16-row blocks per workgroup: large batches (the
// reference predicts in slabs of 250 utterances, enhance.py:1189-1193) reuse every dictionary
// operand for RB row blocks, which divides the operand traffic per flop by up to (1 + RB) / 2RB.
// HALF (BASELINE config 5): both operands are STORED as fp16 and enter the matrix cores through
// v_mfma_f32_16x16x32_f16 -- one MFMA per 32-atom chunk, a lane's operand is one 16-byte load --
// with fp32 accumulation; one atom range (KS = 1), the residual goes out as fp16 in cell_a's
// operand order.  Tile-packed fp16 buffers (1 KB blocks of 512 halves, lane l = q*16 + row at l*16 B):
//     Hp16[m][n/32][q][row][e]   atom 32 (n/32) + 8q + e          (cell_b's A operand)
//     Rp16[m][f/32][q][row][e]   bin  32 (f/32) + 16 (e/4) + 4 (e%4) + q   (cell_a's A operand)
// The arguments are passed as individual scalars (not as one struct) so that the command
// processor can preload them into SGPRs (-amdgpu-kernarg-preload-count, build.py): the kernel
// then starts without a dependent scalar load from the kernarg segment.
// (Two 16-bin column blocks per workgroup -- 0.25 instead of 0.375 KB of operands per MFMA at RB = 2 -- were built
// and measured in round 5 for the large batches and LOST at every size: B = 250 / 512 / 1024 680 -> 613, 846 -> 833,
// 981 -> 967 k frames/s (116 instead of 80 VGPRs, half the workgroups): profiles/r05_cb2_sweep_negative.txt.)
template <int GB, int RB = 1, int NW = 8, bool HALF = false, bool QRED = false>
__global__ void __launch_bounds__(64 * NW)
cell_b_kernel(const void* Dn_next, const float* h_in, const float* xp, float* rpart, int Fp_,
              int Np_, int nft_, int KS_, int nch_ks_, int Bp_, DRNMF_ABLATE_PARAM   // <= 15 dwords: preloaded
              const float* q_in, float* qsum, int numA, int ntail) {
    // (-DDRNMF_MEASURE builds only -- ablate_, DRNMF_ABLATE_B: bit 0 = every dictionary load reads chunk 0,
    // bit 1 = every activation load reads chunk 0; the stream in question then costs nothing, results are garbage)
#ifdef DRNMF_MEASURE
    const CellBArgs a{Dn_next, h_in, xp, rpart, nullptr, Bp_, Fp_, Np_, nft_, KS_,
                      __builtin_ctz((unsigned)KS_), nch_ks_, ablate_, q_in, qsum, numA, ntail};
#else
    const CellBArgs a{Dn_next, h_in, xp, rpart, nullptr, Bp_, Fp_, Np_, nft_, KS_,
                      __builtin_ctz((unsigned)KS_), nch_ks_, q_in, qsum, numA, ntail};
#endif
    __shared__ float qred[MAX_TAIL][NW];
    __shared__ __attribute__((aligned(16))) float red[NW * RB * 16 * 17];   // row stride 17: the
    // epilogue threads read (row, bin 4s+q) with row fastest -- stride 16 would be an 8-way bank conflict
    // 2-D grid (x = 8 * row tile group + XCD slot, y = octet of (bin tile, atom range)): see
    // cell_a_kernel.  KS is a power of two.  Padded blocks redo the last tile with the store
    // predicated off.
    DRNMF_STAMP(0, 0);
    const int m = blockIdx.x >> 3;
    const int rest_raw = blockIdx.y * 8 + (blockIdx.x & 7);
    const bool live = rest_raw < a.nft * a.KS;
    const int rest = live ? rest_raw : a.nft * a.KS - 1;
    const int ft = rest >> a.logKS, ks = rest & (a.KS - 1);

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int l = tid & 63, j = l & 15, q = l >> 4;
    const int Np = a.Np, Fp = a.Fp;
    const int cbase = ks * a.nch_ks;                       // first chunk of this atom range
    int nch = a.Np / (HALF ? 32 : 16) - cbase;             // chunks left in the matrix
    if (nch > a.nch_ks) nch = a.nch_ks;
    if (nch < 1) nch = 1;                                  // (never: KS*nch_ks covers Np/16)
    int per_wave = (nch - w + NW - 1) / NW;
    const int clast = nch - 1;

    const int NAC = HALF ? Np / 32 : Np / 16;              // chunks per row of blocks
    // blocks (mb, cbase+c) of Hp and (ft, cbase+c) of Dp: lane (j, q) = lane q*16 + j reads the
    // float4 of atoms 16c + 4q + {0..3} of row / bin j at lane*16 bytes -> one contiguous 1 KB
    // block per wave instruction, consecutive lanes on consecutive addresses (fp16: 8 halves of the
    // 32-atom chunk, same 16 bytes per lane)
    const float* arow = a.h + ((size_t)m * RB * NAC + cbase) * 256 + l * 4;              // + 256*c
    const size_t astep = (size_t)NAC * 256;                                              // per row block
    const float* brow = (const float*)a.Dn_next + ((size_t)ft * NAC + cbase) * 256 + l * 4;   // [q][bin][e]
    const f16* arow16 = (const f16*)a.h + ((size_t)m * RB * NAC + cbase) * 512 + l * 8;
    const size_t astep16 = (size_t)NAC * 512;
    const f16* brow16 = (const f16*)a.Dn_next + ((size_t)ft * NAC + cbase) * 512 + l * 8;

    // branch-free operand loads (clamped chunk index, zeroed A operand when out of range)
    // ALLB (fp16, GB = 32: long contractions, F = 1025 / N = 8000): every dictionary operand of the
    // wave -- the stream that comes from HBM -- is requested before the first MFMA (32 x 4 VGPRs),
    // the activations (L2) follow in a rotating window of AW slots.  With both streams in one
    // rotating window the wave stalled on each HBM round trip with its L1 path idle: the two
    // latencies added up instead of overlapping.
    constexpr bool ALLB = HALF && GB == 32;
    constexpr int AW = 8;
    f32x4 av[HALF ? 1 : GB][RB], bv[HALF ? 1 : GB];
    f16x8 ah[HALF ? (ALLB ? AW : GB) : 1][RB], bh[HALF ? GB : 1];
    auto load_chunk = [&](int base, int g) {
        int c = w + NW * (base + g);
        c = c > clast ? clast : c;
        if (HALF) {
            const int ca = DRNMF_ABLATED(ablate_, 2, c), cb = DRNMF_ABLATED(ablate_, 1, c);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) ah[g][rb] = *(const f16x8*)(arow16 + rb * astep16 + 512 * ca);
            bh[g] = *(const f16x8*)(brow16 + 512 * cb);
        } else {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) av[g][rb] = *(const f32x4*)(arow + rb * astep + 256 * c);
            bv[g] = *(const f32x4*)(brow + 256 * c);
        }
    };
    DRNMF_STAMP(0, 1);
    // software pipeline: loads run PF chunks ahead of the MFMAs (see cell_a_kernel)
    constexpr int PF0 = HALF ? AW - 1 : (RB > 1 ? 3 : 6);   // (8 = every chunk of the C2 shape up front: no gain)
    constexpr int PF = GB < PF0 ? GB : PF0;
#pragma unroll
    for (int g = 0; g < PF; ++g) load_chunk(0, g);

    // odd-bin partials (see CellBArgs): workgroup r < 16*RB of a row tile group sums the numA
    // partials of its row r.  Issued behind the first operand loads (the arguments involved are not
    // among the preloaded ones: nothing ahead of the operand loads may wait for the kernarg fetch),
    // one load per thread, summed after the MFMA loop.
    // QRED is instantiated for numA > 64 only (N > 2048; such shapes have >= 16*RB workgroups per
    // row tile group): below that the consumer adds its 4 prefetched partials itself and this
    // kernel is exactly the one measured before (the extra arguments are then never fetched).
    const bool qwg = QRED && a.ntail > 0 && rest_raw < 16 * RB;      // (workgroup-uniform)
    const int rgq = (m * RB + (rest_raw >> 4)) * 16 + (rest_raw & 15);
    float qv1[MAX_TAIL];
#pragma unroll
    for (int i = 0; i < MAX_TAIL; ++i) {
        qv1[i] = 0.f;
        if (QRED && qwg && i < a.ntail && tid < a.numA)
            qv1[i] = a.q_in[((size_t)i * a.Bp + rgq) * a.numA + tid];
    }
    // x_t element for the epilogue: issued behind the operand loads, consumed at the very end
    // thread p of the first 256 owns the element at position p of the output block (rp_pos order:
    // p = (q*16 + row)*4 + s <-> bin 4s + q), so that the x_t load and the store are contiguous
    const int ep = tid & 255;
    const int erow = (ep >> 2) & 15, ecol = 4 * (ep & 3) + (ep >> 6);
    // (buffers keep Fp/16 tiles per row tile even when only a.nft of them are MFMA tiles)
    const size_t eoff = ((size_t)m * RB * (Fp / 16) + ft) * 256 + ep;
    const size_t estep = (size_t)(Fp / 16) * 256;                                        // per row block
    float xv[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        xv[rb] = 0.f;
        if (a.xp != nullptr && ks == 0) xv[rb] = a.xp[eoff + rb * estep];
    }

    f32x4 acc[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb][0] = acc[rb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute_chunk = [&](int base, int g) {
        const bool ok = base + g < per_wave;
        // two independent accumulator chains per row block hide the dependent MFMA latency
        if (HALF) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                f16x8 a8 = ah[g][rb];
                if (!ok) a8 = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                acc[rb][g & 1] = mfma32h(a8, bh[g], acc[rb][g & 1]);
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float a1 = ok ? av[g][rb][s] : 0.f;
                acc[rb][s & 1] = mfma16(a1, bv[g][s], acc[rb][s & 1]);
            }
        }
    };
    if (ALLB) {
        // (the prologue filled A and B slots 0 .. AW-2; the rest of the dictionary operands now)
#pragma unroll
        for (int g = AW - 1; g < GB; ++g) {
            int c = w + NW * g;
            c = c > clast ? clast : c;
            bh[g] = *(const f16x8*)(brow16 + 512 * DRNMF_ABLATED(ablate_, 1, c));
        }
#pragma unroll
        for (int i = 0; i < GB; ++i) {
            {
                int c = w + NW * (i + AW - 1);
                c = c > clast ? clast : c;
                const int ca = DRNMF_ABLATED(ablate_, 2, c);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    ah[(i + AW - 1) % AW][rb] = *(const f16x8*)(arow16 + rb * astep16 + 512 * ca);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bool ok = i < per_wave;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                f16x8 a8 = ah[i % AW][rb];
                if (!ok) a8 = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                acc[rb][i & 1] = mfma32h(a8, bh[i], acc[rb][i & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) DRNMF_STAMP(0, 2);
        }
        DRNMF_STAMP(0, 3);
    } else if (HALF) {
        // rotating operand slots (as cell_a_kernel): chunk i lives in slot i mod GB and its loads run
        // GB-1 chunks ahead of its MFMA, so the stream never drains between groups -- the
        // dictionary of this mode comes from HBM (K untied layers exceed the Infinity Cache)
        auto load_slot = [&](int i, int g) {
            int c = w + NW * i;
            c = c > clast ? clast : c;
            const int ca = DRNMF_ABLATED(ablate_, 2, c), cb = DRNMF_ABLATED(ablate_, 1, c);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) ah[g][rb] = *(const f16x8*)(arow16 + rb * astep16 + 512 * ca);
            bh[g] = *(const f16x8*)(brow16 + 512 * cb);
        };
        // (the prologue above filled slots 0 .. PF-1 = GB-1 or fewer; top up to GB-1)
#pragma unroll
        for (int g = PF; g < GB - 1; ++g) load_slot(g, g);
        for (int base = 0; base < per_wave; base += GB) {
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                load_slot(base + g + GB - 1, (g + GB - 1) % GB);
                __builtin_amdgcn_sched_barrier(0);
                compute_chunk(base, g);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
    for (int base = 0; base < per_wave; base += GB) {
        if (base > 0) {
#pragma unroll
            for (int g = 0; g < PF; ++g) load_chunk(base, g);
        }
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            if (g + PF < GB) load_chunk(base, g + PF);
            __builtin_amdgcn_sched_barrier(0);
            compute_chunk(base, g);
            __builtin_amdgcn_sched_barrier(0);
            if (base == 0 && g == 0) DRNMF_STAMP(0, 2);
        }
    }
    DRNMF_STAMP(0, 3);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            red[((w * RB + rb) * 16 + 4 * q + v) * 17 + j] = acc[rb][0][v] + acc[rb][1][v];
    if (QRED && qwg) {
#pragma unroll
        for (int i = 0; i < MAX_TAIL; ++i) {
            float sq = qv1[i];
            if (i < a.ntail)
                for (int b2 = tid + 64 * NW; b2 < a.numA; b2 += 64 * NW)     // numA > 64 NW only
                    sq += a.q_in[((size_t)i * a.Bp + rgq) * a.numA + b2];
            sq = row16_sum(sq);                       // fixed reduction tree: deterministic
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (l == 0) qred[i][w] = sq;
        }
    }
    __syncthreads();
    DRNMF_STAMP(0, 4);
    if (QRED && qwg && tid < a.ntail) {
        float tot = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) tot += qred[tid][ww];
        a.qsum[(size_t)tid * a.Bp + rgq] = tot;
    }
    if (tid >= 256) return;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) s += red[((ww * RB + rb) * 16 + erow) * 17 + ecol];
        if (!live) continue;
        if (HALF) {
            // fp16 residual in cell_a's operand order: chunk ft/2, slot (q = bin%4, e = 4 (ft%2) + bin/4)
            f16* r16 = (f16*)a.rpart + ((size_t)(m * RB + rb) * (Fp / 32) + (ft >> 1)) * 512 +
                       ((ecol & 3) * 16 + erow) * 8 + (ft & 1) * 4 + (ecol >> 2);
            *r16 = (f16)((a.xp != nullptr) ? xv[rb] - s : s);
        } else {
            st_xchg(a.rpart + (size_t)ks * a.Bp * Fp + eoff + rb * estep, (a.xp != nullptr) ? xv[rb] - s : s);
        }
    }
    DRNMF_STAMP(0, 5);
}

// kernelParams array of cell_b_kernel for hipLaunchKernel / hipGraphAddKernelNode
struct CellBParams {
    void* p[15];
    explicit CellBParams(CellBArgs& b)
        : p{&b.Dn_next, &b.h, &b.xp, &b.rpart, &b.Fp, &b.Np, &b.nft, &b.KS, &b.nch_ks, &b.Bp,
#ifdef DRNMF_MEASURE
            &b.ablate,
#endif
            &b.q_in, &b.qsum, &b.numA, &b.ntail} {}
};

__global__ void advance_frame_kernel(int* tptr) { *tptr += 1; }

// Masking + relayout: x [B][T][F] -> xp [T][Bp][Fp] (masked frames and all padding zero) and
// valid [T][Bp].  One wave per (t, row).  [K2.0.4-memory: keras.layers.Masking]
// xp16 (fp16 operand mode, or NULL): the same frames once more as fp16 in Rp16 order (cell_b_kernel), the
// first layer's MFMA operand -- it then reads half the bytes and runs the same operand path as the other layers
// (reading the fp32 blocks, converting and republishing them in the kernel: 160 VGPRs against 125; the
// first-layer launch at F = 1025, N = 8000: 20.4 -> 17.2 us).
__global__ void __launch_bounds__(256)
pack_input_kernel(const float* __restrict__ x, float* __restrict__ xp,
                  unsigned char* __restrict__ valid, float mask_value, int B, int T, int F, int Bp,
                  int Fp, f16* __restrict__ xp16 = nullptr) {
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const size_t rowid = (size_t)blockIdx.x * 4 + wv;   // = t*Bp + b
    if (rowid >= (size_t)T * Bp) return;
    const int t = (int)(rowid / Bp), b = (int)(rowid % Bp);
    const int nft = Fp / 16;
    // tile-packed destination: Rp[t][b/16][f/16][rp_pos(b%16, f%16)]
    float* dst = xp + (size_t)t * Bp * Fp + (size_t)(b >> 4) * nft * 256;
    // Rp16[m][f/32][q = f%4][row][e = 4 ((f/16)%2) + (f%16)/4]   (Fp % 32 == 0 in that mode)
    f16* dst16 = xp16 ? xp16 + (size_t)t * Bp * Fp + (size_t)(b >> 4) * (Fp / 32) * 512 + (b & 15) * 8 : nullptr;
    auto pos16 = [](int f) { return (size_t)(f >> 5) * 512 + (f & 3) * 128 + ((f >> 4) & 1) * 4 + ((f & 15) >> 2); };
    bool any = false;
    if (b < B) {
        const float* src = x + ((size_t)b * T + t) * F;
        for (int f = l; f < F; f += 64) any |= (src[f] != mask_value);
        any = __any(any);
        for (int f = l; f < Fp; f += 64) {
            const float v = (any && f < F) ? src[f] : 0.f;
            dst[(size_t)(f >> 4) * 256 + rp_pos(b & 15, f & 15)] = v;
            if (dst16) dst16[pos16(f)] = (f16)v;
        }
    } else {
        for (int f = l; f < Fp; f += 64) {
            dst[(size_t)(f >> 4) * 256 + rp_pos(b & 15, f & 15)] = 0.f;
            if (dst16) dst16[pos16(f)] = (f16)0.f;
        }
    }
    if (l == 0) valid[rowid] = any ? 1 : 0;
}


// Whether a call takes the Gram form.  A Gram launch streams 8 * numM * Np^2 bytes from the L2
// (every 16 x 16 output tile reads 16 rows of h and 16 rows of G); measured on MI355X
// (tools/profile_shape.py with DRNMF_GRAM=0|1; F = 257, K = 5, B = 32; us per layer-step, factored
// pair -> Gram launch): N = 200 6.0 -> 4.2, N = 512 6.1 -> 4.4, N = 1000 6.3 -> 5.7, N = 2000 7.7 ->
// 8.9; N = 200 wins 1.5-1.6x at every batch from 1 to 250.  Rule: numM * Np^2 <= 3e6 (the crossover
// sits near 25 MB of operand traffic per launch), independent of T and K.
constexpr int GRAM_TB = 64;          // frames per block of the hoisted c_k products (ring of 2 blocks)
static inline bool gram_wanted(const drnmf_cell_desc_t* d) {
    if (!gram_eligible(d)) return false;
    const int Np = pad_n(d->N), Bp = round_up(d->B, ROWS);
    // (no dependence on T: a time prefix of a batch must take the same form)
    bool want = (int64_t)(Bp / ROWS) * Np * Np <= GRAM_MAX_WORK;
    if (const char* e = tune_env("DRNMF_GRAM")) {            // tuning aid: force the choice
        if (atoi(e) == 0) want = false;
        if (atoi(e) == 1) want = true;
    }
    return want;
}

// The consumer of cell_b's odd-bin sums (cell_a / bwd_a) and cell_b itself must agree on who adds
// the partials: cell_b when there are more than 64 atom blocks and enough workgroups to cover the
// 16*RB rows of a tile group.
static inline bool qred_wanted(int numA, int ntail, int nft, int KS, int RB) {
    return ntail > 0 && numA > 64 && nft * KS >= 16 * RB;
}

struct Workspace {
    size_t off_xp, off_valid, off_seen, off_psum_all, off_rpart, off_h0, off_h1, off_state, off_rs,
        off_psum, off_t, total;
    int Bp, Fp, Np, numA, KS, nch_ks;
    int RB;                // 16-row blocks per workgroup (1 or 2); Bp is a multiple of 16*RB
    int RBa;               // the same for cell_a alone (= RB; the fp16 mode may block cell_a only)
    int nft_main, ntail;   // forward: bins 16*nft_main .. F-1 (at most MAX_TAIL) are handled outside the MFMA tiles
    size_t off_qpart, off_xtail, off_xcur, off_qsum;
    size_t off_h16_0, off_h16_1, off_r16;   // fp16 operand mode: Hp16 ping-pong, Rp16 (cell_b_kernel)
    size_t off_xp16;                         // ... and the packed input once more as fp16 (pack_input_kernel)
    bool half;
    // Gram form (cell_gram.h): packed c, ring of 2 x GRAM_TB frames [.][K][Bp][Np]; one block of x
    // padded row-major [B][GRAM_TB][Fp]; q ping-pong
    bool gram;
    int numO;              // output tiles of 16 atoms (= row-sum partials per row)
    size_t off_cp, off_xpad, off_q0, off_q1;
    bool cp_full;          // the hoisted c_k of ALL T frames are resident (one product up front, one
                           // persistent launch for the whole sequence); false: ring of two blocks
    int cp_frames;         // frames of hoisted c_k the workspace holds: 2 GRAM_TB (a ring of two blocks)
                           // or, for the shapes the persistent chains serve, all T (one product up
                           // front, one launch for the whole sequence)
    size_t off_rsave;      // training forward (all hidden layers, fp32, factored): residuals r_k, k >= 1,
                           // row-major [K-1][B*T][Fp] (MFMA bin tiles in tile_unpermute order) for the
                           // BPTT's weight gradients; 0 = absent
    size_t off_xhat;       // KL / beta cell, training forward: x^_k = h_in Dn_k^T of every (frame, layer),
                           // tile-packed [T][K][Bp][Fp] (the BPTT needs dg/dx^ there); 0 = absent
    // Large inference batches (the reference predicts in slabs of 250 utterances, enhance.py:1189-1193):
    // batch rows never interact, so the batch runs as `split` independent sub-batches of `split_rows`
    // rows (the last one takes the rest) on the caller's stream + side streams of the handle -- the
    // launch boundaries, prologues and epilogues of one sub-batch's chain fill with the other's MFMA
    // loops (tools/two_stream_probe.py: B = 256 +13 % with two, B = 512 +15 % with four).  Sub-batch s
    // owns the workspace bytes [s * split_bytes, (s + 1) * split_bytes).
    int split, split_rows;
    size_t split_bytes;
};

Workspace workspace_layout(const drnmf_cell_desc_t* d, bool allow_split = true) {
    Workspace W;
    W.gram = gram_wanted(d);
    W.half = d->operand_f16 != 0;
    W.Fp = pad_f_mode(d->F, W.half);
    W.Np = pad_n(d->N);
    W.numA = W.Np / ATOMS;
    // (the KL / beta cell applies a nonlinear map to the COMPLETE x^ of every bin: no odd-bin side
    // path, a single atom range)
    const bool nonlin = d->divergence != DRNMF_DIV_ED;
    W.ntail = (!nonlin && d->F % 16 != 0 && d->F % 16 <= MAX_TAIL && d->F > 16) ? d->F % 16 : 0;
    W.nft_main = W.ntail ? d->F / 16 : W.Fp / 16;
    // Row blocks per workgroup.  Measured on MI355X (F=513, N=2000, K=25; frames/s with 1 / 2 / 4
    // row blocks): B=128 343k / 332k / 206k, B=256 431k / 486k / 399k, B=512 526k / 572k / 537k,
    // B=1024 577k / 636k / 595k -- two blocks pay once every CU holds >= 2 such workgroups; four
    // leave too few waves per CU to hide the operand latency (round 4, sub-batches on side streams, four
    // against the rule below: B=512 639k / 848k, B=1024 640k / 982k: profiles/r04j_rb4_sweep.txt).
    W.RB = 1;
    {
        const int groups = round_up(d->B, ROWS * 2) / (ROWS * 2);
        // (a SUB-BATCH of a split call -- allow_split == false -- shares the chip with its siblings'
        // launches: one such workgroup per CU is enough, and the halved operand traffic per MFMA is what
        // counts; measured at B = 250 as 2 x 125 rows: 686-695 k frames/s with one row block, 723 k with two)
        const int need = allow_split ? 500 : 250;
        if (groups * W.numA >= need && groups * W.nft_main * 2 >= need) W.RB = 2;
    }
    if (const char* e = tune_env("DRNMF_RB")) {   // tuning aid: force the row blocking
        const int v = atoi(e);
        if (v == 1 || v == 2) W.RB = v;     // (four: measured in round 4, never chosen, removed: DESIGN.md 4.2)
    }
    if (W.gram) W.RB = 1;   // (one row block per workgroup: gram_wanted() counts 16-row tiles)
    // fp16 operand mode: cell_b keeps one row block per workgroup (its 16 x 16 output tiles with
    // the whole contraction are what fills the chip), cell_a may still carry two, which halves the
    // number of times a dictionary slice is pulled out of the L2
    W.RBa = W.RB;
    if (W.half && W.RB == 1) {
        const int groups = round_up(d->B, ROWS * 2) / (ROWS * 2);
        // measured at F=1025, N=8000, B=64 (once the odd-bin partials were out of cell_a's tail): one row
        // block 10.4 us per launch, two 9.6, four 10.0
        if (groups * W.numA >= 256) W.RBa = 2;
        if (const char* e = tune_env("DRNMF_RBA")) {
            const int v = atoi(e);
            if (v == 1 || v == 2) W.RBa = v;
        }
    }
    W.Bp = round_up(d->B, ROWS * (W.RBa > W.RB ? W.RBa : W.RB));
    // atom ranges per (row tile group, bin tile) in cell_b: enough workgroups to cover the 256
    // CUs (row-blocked kernels are instantiated for KS <= 2 only)
    const int tiles = (W.Bp / (ROWS * W.RB)) * W.nft_main;
    const int nchN = W.Np / 16;
    int KS = 1;
    // (measured, frames/s with 2 / 4 / 8 ranges at N = 2000: B=16 F=257 481k / 514k / 495k; B=16 F=513
    // 87k / 90k / 76k; B=32 F=257 931k / 991k / 943k and the training step 48.3 / 46.4 / 48.4 ms;
    // B=32 F=513 166k / 171k / -; B=48 242k / 231k / -; B=64 291k / 279k / -: never more than 4, and
    // the smallest count that gives ~192 workgroups -- each extra range is another residual partial
    // for every cell_a workgroup to read)
    const bool blocked = W.RB > 1 || (!W.half && W.RBa > 1);     // (row-blocked cell_a / cell_b: KS <= 2)
    while (KS < (blocked ? 2 : 4) && tiles * KS < 192 && nchN / (KS * 2) >= 4) KS *= 2;
    if (const char* e = tune_env("DRNMF_KS")) {   // tuning aid: force the number of atom ranges
        const int v = atoi(e);
        if ((v == 1 || v == 2 || v == 4 || v == 8) && (!blocked || v <= 2) && nchN / v >= 1) KS = v;
    }
    if (nonlin) KS = 1;
    if (W.half) KS = 1;        // the fp16 residual is stored once, already rounded (cell_b_kernel)
    W.KS = KS;
    W.nch_ks = W.half ? W.Np / 32 : (nchN + KS - 1) / KS;   // fp16: 32-atom chunks
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += round_up_sz(bytes, 256); return at; };
    W.off_xp = take((size_t)d->T * W.Bp * W.Fp * 4);
    W.off_valid = take((size_t)d->T * W.Bp);
    W.off_seen = take((size_t)d->T * W.Bp);          // any valid frame before t (backward)
    W.off_psum_all = take((size_t)d->T * W.Bp * 4);  // sum(p) of every frame (backward)
    W.off_rpart = take((size_t)MAX_KS * W.Bp * W.Fp * 4);
    W.off_h0 = take((size_t)W.Bp * W.Np * 4);
    W.off_h1 = take((size_t)W.Bp * W.Np * 4);
    W.off_state = take((size_t)W.Bp * W.Np * 4);
    W.numO = W.Np / 16;
    W.off_rs = take((size_t)2 * (W.gram ? W.numO : round_up(W.numA, 4)) * W.Bp * 4);
    W.off_psum = take((size_t)W.Bp * 4);
    W.off_qpart = take((size_t)2 * W.numA * MAX_TAIL * W.Bp * 4);
    W.off_xtail = take((size_t)MAX_TAIL * W.Bp * 4);
    W.off_qsum = take((size_t)MAX_TAIL * W.Bp * 4);
    W.off_xcur = take((size_t)W.Bp * W.Fp * 4);
    W.off_h16_0 = take(W.half ? (size_t)W.Bp * W.Np * 2 : 0);
    W.off_h16_1 = take(W.half ? (size_t)W.Bp * W.Np * 2 : 0);
    W.off_r16 = take(W.half ? (size_t)W.Bp * W.Fp * 2 : 0);
    W.off_xp16 = take(W.half ? (size_t)d->T * W.Bp * W.Fp * 2 : 0);
    W.cp_full = W.gram && W.numO <= 32 && W.Bp / ROWS <= 16 && d->T > 2 * GRAM_TB &&
                (size_t)d->T * d->K * W.Bp * W.Np * 4 <= ((size_t)1 << 30);
    if (const char* e = tune_env("DRNMF_CP_FULL"))      // tuning aid: 0 keeps the ring of two blocks
        if (atoi(e) == 0) W.cp_full = false;
    W.cp_frames = W.cp_full ? d->T : 2 * GRAM_TB;
    W.off_cp = take(W.gram ? (size_t)W.cp_frames * d->K * W.Bp * W.Np * 4 : 0);
    W.off_xpad = take(W.gram ? (size_t)d->B * (W.cp_full ? d->T : GRAM_TB) * W.Fp * 4 : 0);
    W.off_q0 = take(W.gram ? (size_t)W.Bp * W.Np * 4 : 0);
    W.off_q1 = take(W.gram ? (size_t)W.Bp * W.Np * 4 : 0);
    W.off_rsave = 0;
    if (d->return_all_hidden && !W.half && !W.gram && d->divergence == DRNMF_DIV_ED && d->K >= 2) {
        // (the padding bins of every row must read as zero in the gradient GEMMs: when Fp > F the
        // forward clears the buffer once per call)
        W.off_rsave = take((size_t)(d->K - 1) * d->B * d->T * W.Fp * 4);
    }
    W.off_xhat = 0;
    if (nonlin && d->return_all_hidden) {
        (void)take(256);                                      // (an offset of 0 means "absent")
        W.off_xhat = take((size_t)d->T * d->K * W.Bp * W.Fp * 4);
    }
    W.off_t = take(256 + 65536);     // frame counters (256 B) + the persistent chains' sync lines (cell_gram_persist.h)
    W.total = o;
    W.split = 1;
    W.split_rows = d->B;
    W.split_bytes = 0;
    if (allow_split && !W.gram && !d->return_all_hidden && d->divergence == DRNMF_DIV_ED) {
        // Sub-batches of 64, 128 or 256 rows (whole groups of two 16-row blocks; the last one takes the
        // rest).  Measured, cell + head, T = 400, k frames/s by number of sub-batches
        // (profiles/r04f_split_sweep.txt): B = 80: 286 unsplit / 343 with two (B = 64: 318 / 254 -- no split);
        // 192: 573 (2) / 626 (3) / 623 (4); 250: 722 (2) / 647 (3); 320: 633 (2) / 718 (3) / 685 (4);
        // 512: 789 (2) / 812 (3) / 844 (4); 768: 899 (2) / 935 (3) / 932 (4); 1024: 964 (3) / 973 (4) /
        // 800 (5) / 774 (6); 640 as 5 x 128: 681, 224 as 4 x 64: 604 -- sub-batches that are not a multiple of
        // 64 rows lose, and so do five or six of them.  Beyond 256 rows one
        // sub-batch's h no longer fits an XCD's 4 MB L2 next to its dictionary slices (B = 1024 unsplit:
        // cell_b at 36 % L2 hit rate).
        int R = d->B < 80 ? d->B : (d->B <= 192 ? 64 : (d->B <= 512 ? 128 : 256));
        int S = (d->B + R - 1) / R;
        if (S > MAX_SPLIT) { S = MAX_SPLIT; R = round_up((d->B + S - 1) / S, 64); }
        if (const char* e = tune_env("DRNMF_SPLIT")) {      // tuning aid: force the number of sub-batches
            const int v = atoi(e);
            if (v >= 1 && v <= MAX_SPLIT) { S = v; R = round_up((d->B + S - 1) / S, 32); }
        }
        while (S > 1 && d->B < 32 * S) --S;
        if (S * R < d->B) R = round_up((d->B + S - 1) / S, 32);     // (the sub-batches must cover the batch)
        if (S > 1) {
            W.split = S;
            W.split_rows = R;
            drnmf_cell_desc_t ds = *d;
            ds.B = W.split_rows;
            W.split_bytes = round_up_sz(workspace_layout(&ds, false).total, 256);
            if (W.split_bytes * S > W.total) W.total = W.split_bytes * S;
        }
    }
    return W;
}

template <int RB, bool HALF, bool QRED>
void* pick_b_func_rb(int nch_ks) {
    const int per_wave = (nch_ks + 7) / 8;
    if (per_wave <= 2) return (void*)&cell_b_kernel<2, RB, 8, HALF, QRED>;
    if (per_wave <= 4) return (void*)&cell_b_kernel<4, RB, 8, HALF, QRED>;
    if constexpr (HALF && RB == 1) {
        // (a deeper rotating window for BOTH streams is slower: 16 slots 11.3, 24 slots 13.7 us per
        // cell_b launch against 10.4 with 8 and 9.8 with the dictionary operands up front; 16 waves
        // with every operand of every chunk requested up front: 19 us -- the per-CU miss path
        // degrades when it is flooded)
        if (per_wave > 8 && per_wave <= 32 && !measure_env("DRNMF_NO_ALLB"))
            return (void*)&cell_b_kernel<32, 1, 8, true, QRED>;     // all dictionary operands up front
    }
    return (void*)&cell_b_kernel<8, RB, 8, HALF, QRED>;
}
void* pick_b_func(int nch_ks, int RB = 1, bool half = false, bool qred = false) {
    // fp16: nch_ks counts 32-atom chunks; eight rotating operand slots where a wave owns that many
    if (qred) {
        if (half) return RB == 2 ? pick_b_func_rb<2, true, true>(nch_ks) : pick_b_func_rb<1, true, true>(nch_ks);
        return RB == 2 ? pick_b_func_rb<2, false, true>(nch_ks) : pick_b_func_rb<1, false, true>(nch_ks);
    }
    if (half) return RB == 2 ? pick_b_func_rb<2, true, false>(nch_ks) : pick_b_func_rb<1, true, false>(nch_ks);
    if (RB == 2) return pick_b_func_rb<2, false, false>(nch_ks);
    return pick_b_func_rb<1, false, false>(nch_ks);
}

}  // namespace
