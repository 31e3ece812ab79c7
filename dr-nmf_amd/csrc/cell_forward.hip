// Recurrent DR-NMF cell forward on gfx950.
//
// Reference semantics: Masking (enhance.py:253) -> Recurrent.call / K.rnn ->
// SimpleDeepRNN.get_initial_state + step (custom_layers.py:336-375) with the build_alt maps
// (enhance.py:161-204), in the factored ISTA form
//     layer 0 : h  = relu(u0d*p + u0o*(sum(p)-p) + (x Dn_0)*ia_0 + b_0)
//     layer k : h' = relu(h + ((x - h Dn_k^T) Dn_k)*ia_k + b_k + uko*sum(p))
//
// The chain over (t, k) is strictly sequential; only the batch rows and the inside of each
// contraction are parallel.  One layer-step is two dependent skinny GEMMs (B x N x F each) with
// an all-to-all between them (x^ contracts over every atom, g over every bin).  On gfx950 a
// kernel boundary (~2 us) is cheaper than any in-kernel all-to-all, so each GEMM is one launch
// sized to the whole chip and the exchange rides on the boundary:
//
//   cell_a_kernel  (workgroup = 16 batch rows x 32 atoms, all bins; bins split over 4 waves):
//       r  = sum_ks rpart[ks]                  (x_t for layer 0)
//       g  = r[16 x F] . Dn_k[F x 32]          MFMA + LDS reduce over the 4 waves
//       h' = relu(...)                         fused epilogue (+ mask select, row sums, output)
//   cell_b_kernel  (workgroup = 16 rows x 16 bins x one of KS atom ranges; atoms over 4 waves):
//       x^ = h'[16 x N/KS] . Dn_{k+1}[16 x N/KS]^T   MFMA + LDS reduce
//       rpart[ks] = (ks==0 ? x_t : 0) - x^     (the KS partials are summed by the next cell_a)
//
// Internal layouts are TILE-PACKED: every matrix the two kernels exchange (dictionary, h, state,
// residual partials, packed input) is stored as 1 KB blocks of 16 x 16 floats, so that each MFMA
// operand load instruction of a wave covers contiguous full cache lines (a row-major layout makes
// every load touch 16 half-used lines and the kernels stall on VMEM issue):
//     Dp[ft][ac][q][f%16][e]   Hp[m][ac][hp_pos(row%16, n%16)]   Rp[m][ft][rp_pos(row%16, f%16)]
// with ft = f/16, ac = n/16, m = row/16; hp_pos / rp_pos (common.h) order a block so that lane
// q*16 + row finds its four operand values at lane*16 bytes (cell_a contracts bin 16c + 4s + q in
// its s-th MFMA, cell_b atom 16c + 4q + s); cell_a / bwd_a read the dictionary from a second
// packing, DpA (common.h).
//
// A frame is 2K-1 launches, built once as a hipGraph and replayed T times; kernels read the frame
// index from device memory.  Dictionary operands are read straight into MFMA operand registers:
// they are not shared between the waves of a workgroup, so an LDS round trip would only add
// latency; all of a wave's operand loads are issued before its first MFMA.
#include "cell_shared.h"
#include "cell_gram.h"
#include "cell_gram_persist.h"
#include "gemm_nt.h"

#include <type_traits>

namespace {

struct CellAArgs {
    const void* Dn;          // packed unit-norm dictionary of this layer (fp32 Dp, or the fp16 DpA packing)
    const float* inv_alpha;  // [Np]
    const float* bias;       // [Np]
    const float* rsrc;       // first layer: xp [T][Bp][Fp]; else rpart [KS][Bp][Fp]
    const float* h_in;       // [Bp][Np]  previous layer's h (first layer: the state p)
    float* h_out;            // [Bp][Np]  this layer's h (last layer: the state)
    f16* h16_out;            // fp16 mode: the same h as Hp16, cell_b's MFMA operand (cell_shared.h)
    float* state;            // [Bp][Np]
    float* rs_part;          // [2][Bp][numA rounded up to 4] row sums of the state per atom block, by frame parity
    float* psum;             // [Bp]  sum(p) of the current frame
    float* psum_all;         // [T][Bp] the same, kept for the backward pass
    const unsigned char* valid;  // [T][Bp]
    float* out;              // [B][T][out_width]
    const int* t_rd;         // frame counter to read
    int* t_wr;               // counter to publish (or null)
    int t_wr_add;
    float u0d, u0o, uko;
    const float* Dtail;      // [MAX_TAIL][Np] tail-bin rows of this layer's dictionary
    const float* Dtail_next; // the same of layer k+1 (for the next x^ tail partials)
    const float* q_in;       // tail-bin x^ of the previous layer.  qred = 0: its per-atom-block partials
                             // [MAX_TAIL][Bp][numA] (numA <= 64: four per lane, added here);
                             // qred = 1: [MAX_TAIL][Bp], summed by the cell_b launch in between
    float* q_out;            // same, produced for the next layer
    float* xtail;            // [MAX_TAIL][Bp] tail bins of x_t (published by the first layer)
    float* xcur;             // [Bp][Fp] packed x_t, republished by the first layer for cell_b
    int B, T, N, Bp, Fp, Np, numA, nchunks, KS, ntail;
    int tail_tile;           // 16-bin tile index of the odd bins (= MFMA tiles of 16 bins)
#ifdef DRNMF_MEASURE
    int ablate;              // measurement aid (DRNMF_ABLATE_A): bit 0 = dictionary loads read chunk 0,
                             // bit 1 = residual loads read chunk 0 (fp16 mode; results are garbage)
#endif
    float* Rsave;            // training (all-hidden, fp32, k >= 1): this layer's residual r_k of every
                             // frame, row-major [B*T][Fp] with the columns of every MFMA bin tile in
                             // the saved order (common.h tile_unpermute), for the weight gradients of
                             // the BPTT (saves its recomputation as a frame-parallel GEMM); else NULL
    int out_width, out_off, write_out;
    const void* Dn_pf;       // fp16 mode: the NEXT layer's dictionary (the cell_b launch behind this one reads it),
                             // pulled into the L2s by this launch's fifth wave; NULL: no prefetch
    int pf_tiles;            // MFMA bin tiles of that cell_b launch (its workgroup (x, y) reads tile 8 y + x % 8)
    int pf_sleep;            // the prefetching wave starts 8 * 64 * pf_sleep cycles into the launch
    const f16* x16;          // fp16 mode, first layer: the packed input as fp16, Rp16 order [T][Bp][Fp] (pack_input_kernel)
};

// G = operand slots (16-bin chunks in flight per wave, prefetch distance G-1).  RB = 16-row blocks per
// workgroup (see cell_b_kernel): every dictionary operand feeds RB row blocks.
// HALF: residual and dictionary are stored as fp16 and enter the matrix cores through
// v_mfma_f32_16x16x32_f16 (two MFMAs per 32-bin chunk, 16-byte operands per lane; KS = 1; the
// first layer converts x_t from the fp32 input blocks), fp32 accumulation; state, sums, update and
// the odd bins stay fp32; h goes out in fp32 (next epilogue) AND as fp16 (cell_b's operand).
// The fp16 dictionary exists in ONE packing, cell_b's (atoms contiguous per lane: params.hip), so that
// the cell_b / cell_a pair of a layer reads the same bytes (the second read is an Infinity-Cache hit,
// the prepared block is half the size).  This kernel contracts over BINS: a wave stages the two 1-KB
// blocks of its 32-bin chunk (16 bins x 32 atoms each, 16 bytes per lane as they lie in memory)
// through a wave-private LDS buffer and reads them back with ds_read_b64_tr_b16, which hands lane
// (atom j, slot q) the four bins 4e + q (e = 0..3) of atom j out of a [4 bins][4 atoms] gather per 16 lanes
// -- exactly the slot order of the fp16 residual Rp16 (cell_shared.h).
// What the operand addresses need is passed as leading scalar arguments (preloaded into SGPRs by
// the command processor, see cell_b_kernel); the rest of the struct is fetched by scalar loads that
// are not on the path to the first operand load.
// QRED: the odd-bin x^ of the previous layer arrives already summed (cell_b, more than 64 atom blocks);
// a template parameter, not a runtime switch: the epilogue-operand block of the headline
// instantiation is sensitive to every extra register and branch (a runtime `if` cost 3 % there).
// (A 64-row form that shares the dictionary chunk between the waves through LDS, and one with four row
// blocks per wave in registers, were built, measured and rejected in round 4: profiles/r04j_ldsb_sweep.txt,
// r04j_rb4_sweep.txt)
template <int G, int KS, int RB, bool IS_FIRST, bool IS_LAST, bool ALL_HIDDEN, bool HALF = false,
          bool LATE = false, bool QRED = false>
__global__ void __launch_bounds__(64 * (NW_A + (HALF ? 1 : 0)))
cell_a_kernel(const float* rsrc_, const void* Dn_, const int* t_rd_, int Bp_, int Fp_, int Np_,
              int numA_, int nchunks_, const CellAArgs a_in) {
    CellAArgs a = a_in;
    a.rsrc = rsrc_; a.Dn = Dn_; a.t_rd = t_rd_; a.Bp = Bp_; a.Fp = Fp_; a.Np = Np_;
    a.numA = numA_; a.nchunks = nchunks_;
    constexpr bool WRITE_OUT = IS_LAST || ALL_HIDDEN;
    DRNMF_STAMP(1, 0);
    __shared__ __attribute__((aligned(16))) float red[NW_A * RB * ROWS * ATOMS];   // [NW][RB][16][32]

    // 2-D grid (x = 8 * row tile group + XCD slot, y = atom block octet): workgroups are dealt
    // round-robin to the 8 XCDs by linear id = x + y * gridDim.x, gridDim.x % 8 == 0, so the row
    // tiles that share one dictionary slice (atom block 8y + x%8) land on the same XCD / L2 AND
    // are dispatched next to each other (with the atom block as the fast index, a per-XCD
    // dictionary share above the 4 MB L2 -- F=1025, N=8000 -- is evicted between the row tiles
    // that use it: 26.6k -> 28.5k frames/s there, C2 unchanged).  Blocks past numA redo the last atom
    // block with every store predicated off: no early exit and no division, so all kernel
    // arguments arrive in ONE scalar-load round trip ahead of the operand loads.
    const int mb0 = (blockIdx.x >> 3) * RB;       // first 16-row block of this workgroup
    const int ab_raw = blockIdx.y * 8 + (blockIdx.x & 7);
    const bool live = ab_raw < a.numA;
    const int ab = live ? ab_raw : a.numA - 1;

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int l = tid & 63, j = l & 15, q = l >> 4;
    const int Fp = a.Fp, Np = a.Np;
    const int n0 = ab * ATOMS;
    const size_t pstride = (size_t)a.Bp * Fp;     // one r partial

    // ---- fp16 mode: a FIFTH wave that only prefetches ------------------------------------------------
    // The cell_b launch behind this one reads the next layer's dictionary for the first time -- with untied
    // layers past the Infinity Cache (BASELINE configs[4]: 0.8 GB) an HBM miss per line, ~2.8 us to first
    // data and ~3 TB/s under that flood.  Its workgroups with blockIdx.x % 8 = x read the bin tiles
    // ft = x (mod 8), 16 bins x all atoms = Np/32 contiguous 1-KB blocks each, 2 MB per XCD out of a 4-MB L2;
    // workgroups are dealt round-robin to the XCDs, so the workgroups of THIS launch with the same x sit on
    // the same XCD and pull exactly that share through their L2 while the four working waves (whose own
    // dictionary was left in the Infinity Cache by the previous cell_b) keep the matrix pipes.  The loads
    // have no consumer (s_endpgm waits for them); this wave takes no part in anything else but the barrier.
    // A separate prefetch kernel on a side stream, paced by a progress counter, LOST: 8.78 -> 9.05-12.7 us per
    // launch (profiles/r05_prefetch_side_stream_negative.txt).
    if (HALF && w == NW_A) {
        if (IS_FIRST && !IS_LAST) {
            // first layer: republish x_t (fp32) at a frame-independent address -- cell_b subtracts x^ from it
            // without a dependent frame-index load.  One 1-KB block per workgroup of the row group (the working
            // waves read the fp16 copy of x_t, pack_input_kernel; they used to read these fp32 blocks, convert
            // and republish them: 160 VGPRs instead of 125)
            const int t5 = *a.t_rd;
            const size_t g0 = (size_t)mb0 * (Fp / 16) * 256 + l * 4;
            const float* src = a.rsrc + (size_t)t5 * pstride + g0;
            float* dst = a.xcur + g0;
            const int nblk = RB * (Fp / 16);           // (the row blocks of a group are contiguous)
            for (int b = ab_raw; b < nblk; b += 8 * (int)gridDim.y)
                *(f32x4*)(dst + (size_t)b * 256) = *(const f32x4*)(src + (size_t)b * 256);
        }
        for (int i = 0; i < a.pf_sleep; ++i) __builtin_amdgcn_s_sleep(8);      // (see make_a; DRNMF_PF_SLEEP)
        if (!IS_LAST && a.Dn_pf != nullptr) {
            const int x = blockIdx.x & 7;
            const int nac32 = Np >> 5;
            const int ntile = (a.pf_tiles - x + 7) >> 3;                 // bin tiles of this XCD's share
            const int wg = (blockIdx.x >> 3) * gridDim.y + blockIdx.y;   // this workgroup among the XCD's
            const int nwg = (gridDim.x >> 3) * gridDim.y;
            // block b of the share = (tile b / nac32, column b % nac32), b = wg, wg + nwg, ...
            int tj = 0, col = wg;
            while (col >= nac32) { col -= nac32; ++tj; }
            const int dj = nwg / nac32, dc = nwg - dj * nac32;
            const char* base = (const char*)a.Dn_pf + (size_t)l * 16;
            // (ONE destination register quad, tied through every load and the final wait: the compiler does
            // not know the loads are asynchronous, and a quad it considered dead would be handed to the next
            // address -- which the returning data then overwrites)
            f32x4 sink = {0.f, 0.f, 0.f, 0.f};
            while (tj < ntile) {
                const char* p = base + ((size_t)(x + 8 * tj) * nac32 + col) * 1024;
                asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(p) : "memory");
                tj += dj;
                col += dc;
                if (col >= nac32) { col -= nac32; ++tj; }
            }
            __syncthreads();           // (the one barrier of the kernel: not held up by the loads)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");
            return;
        }
        __syncthreads();
        return;
    }

    int t = 0;
    if (IS_FIRST) {
        t = *a.t_rd;
        if (a.t_wr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.t_wr = t + a.t_wr_add;
    }
    const float* rsrc = IS_FIRST ? a.rsrc + (size_t)t * pstride : a.rsrc;
    const int NAC = Np / 16, nft = Fp / 16, ac0 = ab * 2;
    // A: block (mb, c) of Rp, lane (row j, slot q) reads the float4 {s = 0..3} <-> f = 16c+4s+q
    const float* arow = rsrc + (size_t)mb0 * nft * 256 + l * 4;                   // + 256*c
    const size_t astep = (size_t)nft * 256;                                        // per row block
    // B: blocks (c, ac0 | ac0+1) of Dp, rows 4s+q, atoms 2j, 2j+1 of the 32
    // fp32: the cell_a packing (common.h): block (c, ab) of 512 floats, lane l reads 2 x 16 bytes
    const float* brow = (const float*)a.Dn + (size_t)ab * 512 + l * 4;
    // fp16: blocks (2c | 2c+1, ab) of 512 halves of the one fp16 packing (params.hip), 16 bytes per lane
    const f16* brow16 = (const f16*)a.Dn + (size_t)ab * 512 + l * 8;
    const size_t bstep = HALF ? (size_t)NAC * 512 : (size_t)NAC * 256;             // per chunk c
    const size_t bstep_blk = (size_t)NAC * 256;                                     // fp16: block 2c -> 2c+1
    // fp16 residual Rp16 (cell_shared.h): block (mb, c) of 512 halves; first layer: x_t in the same order
    const f16* arow16 = (IS_FIRST ? a.x16 + (size_t)t * pstride : (const f16*)rsrc) +
                        (size_t)mb0 * (Fp / 32) * 512 + l * 8;
    const size_t astep16 = (size_t)(Fp / 32) * 512;

    // training: the summed residual of this row tile also goes out row-major for the BPTT's weight
    // gradients (Rsave).  Every workgroup of the row tile holds all of it; chunk c is stored by atom
    // block c mod numA, by the wave that owns it.  Its (re-)load is issued HERE, ahead of the
    // operand stream, and the store follows the MFMA loop: by then the data is there (in-order return)
    // and the stores drain under the reduction and the epilogue.
#ifdef DRNMF_EXP_NORSAVE
    constexpr bool RSAVE = false;
#else
    constexpr bool RSAVE = ALL_HIDDEN && !IS_FIRST && !HALF;
#endif
    const bool rs_mine = RSAVE && a.Rsave != nullptr && live && ab < a.nchunks &&
                         (ab & (NW_A - 1)) == w;
    f32x4 rsv[RSAVE ? RB : 1][RSAVE ? KS : 1];
    if (RSAVE && rs_mine) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                rsv[rb][ks] = *(const f32x4*)(arow + rb * astep + 256 * ab + (size_t)ks * pstride);
    }

    // ---- GEMM operands first (critical path), wave w takes chunks c = w (mod 4) -------------
    // Loads are branch-free (out-of-range chunks are clamped to the last chunk and their A
    // operand zeroed) so that the compiler can retire them with counted vmcnt waits and the
    // MFMAs start as soon as the first chunk lands.
    f32x4 acc[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb][0] = acc[rb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (fp16 mode: a.nchunks counts 32-bin chunks)
    const int per_wave = (a.nchunks - w + NW_A - 1) / NW_A;   // chunks owned by this wave (>= 0)
    const int clast = a.nchunks - 1;
    // fp32: KS residual partials per chunk (x_t: one).  fp16: one fp16 block per 32-bin chunk (ah) -- the
    // residual, or for the first layer the fp16 copy of x_t
    constexpr int NP = HALF ? 1 : (IS_FIRST ? 1 : KS);
    constexpr bool A16 = HALF;

    f32x4 av[A16 ? 1 : G][RB][NP];
    f32x4 bv[HALF ? 1 : G][2];      // {s = 2i: atoms a0 a1, s = 2i+1: a0 a1} for i = 0, 1
    f16x8 ah[A16 ? G : 1][RB];
    f16x8 bh[HALF ? G : 1][2];
    auto load_chunk = [&](int i, int g) {      // chunk i of this wave -> slot g
        int c = w + NW_A * i;
        c = c > clast ? clast : c;
        const int cb = HALF ? DRNMF_ABLATED(a.ablate, 1, c) : c;
        if (A16) {
            const int ca = DRNMF_ABLATED(a.ablate, 2, c);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) ah[g][rb] = *(const f16x8*)(arow16 + rb * astep16 + 512 * ca);
        } else {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int ks = 0; ks < NP; ++ks)
                    av[g][rb][ks] = *(const f32x4*)(arow + rb * astep + 256 * c + (size_t)ks * pstride);
        }
        if (HALF) {
            bh[g][0] = *(const f16x8*)(brow16 + (size_t)cb * bstep);                // bins 32c .. 32c+15
            bh[g][1] = *(const f16x8*)(brow16 + (size_t)cb * bstep + bstep_blk);    // bins 32c+16 .. 32c+31
        } else {
            bv[g][0] = *(const f32x4*)(brow + (size_t)c * bstep);
            bv[g][1] = *(const f32x4*)(brow + (size_t)c * bstep + 256);
        }
    };
    // Software pipeline: the texture path of a CU moves 64 B/clk and is shared by the 4 waves, so
    // issuing one chunk's operand loads for all waves takes about as long as one wave's MFMAs on
    // that chunk.  Loads run PF chunks ahead of the MFMAs (enough to cover the L2/fabric latency)
    // and the two streams overlap instead of adding up.
    constexpr int PF = G - 1;   // prefetch distance 2..4 measured equivalent, 6 slower (C2 shape)
    // Slots past a wave's last chunk are never loaded but still multiplied (by a zeroed residual,
    // branch-free): their dictionary registers must not hold Inf/NaN garbage (one fp16 bit
    // pattern in 32 does).
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (!HALF) bv[g][0] = bv[g][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (HALF) bh[g][0] = bh[g][1] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (A16) ah[g][rb] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < NP; ++ks)
                if (!A16) av[g][rb][ks] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int g = 0; g < PF; ++g) load_chunk(g, g);
    // everything below depends on struct fields that are still being fetched by scalar loads:
    // keep it (and the wait for those loads) behind the first operand loads
    __builtin_amdgcn_sched_barrier(0);
    DRNMF_STAMP(1, 1);

    // ---- epilogue operands (tiny, needed last).  Their loads are issued behind the FIRST chunk's
    // MFMAs (load_epilogue_operands below): issued ahead of the GEMM loop they sat between the
    // prefetched operand chunks and the first MFMA in the in-order vmcnt count, so that the first
    // MFMA waited for them (and for all prefetched chunks) instead of for chunk 0 alone.
    const int erow = (tid & 255) >> 4, ec = (tid & 15) * 2;
    const int n = n0 + ec;
    const size_t hoff0 = ((size_t)mb0 * NAC + ac0 + (ec >> 4)) * 256 + hp_pos(erow, ec & 15);
    const size_t hstep = (size_t)NAC * 256;                                        // per row block
    f32x2 ia, bs;
    f32x2 hp[RB];
    float ps[RB];
    bool vld[RB];
    // tail bins (F = 16*nchunks + ntail): residual r_tail = x_tail - sum over atom blocks of the
    // partial dot products left by the previous layer.  Only the LOADS are issued; they are
    // reduced after the MFMA loop (consuming them earlier would force an in-order vmcnt wait on
    // every operand load issued so far).
    float xt[RB][MAX_TAIL];
    float qs[QRED ? RB : 1][MAX_TAIL];
    float qv[QRED ? 1 : RB][MAX_TAIL][4];
    f32x2 dt[MAX_TAIL], dtn[MAX_TAIL];   // tail rows of this layer's and the next layer's dictionary
    auto load_epilogue_operands = [&]() {
    ia = *(const f32x2*)(a.inv_alpha + n);
    bs = *(const f32x2*)(a.bias + n);
    if (!IS_FIRST && WRITE_OUT) {
        t = *a.t_rd;
        if (a.t_wr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.t_wr = t + a.t_wr_add;
    }
#pragma unroll
    for (int i = 0; i < MAX_TAIL; ++i) {
        dt[i] = *(const f32x2*)(a.Dtail + (size_t)i * Np + n);
        dtn[i] = *(const f32x2*)(a.Dtail_next + (size_t)i * Np + n);   // (used at the very end: a
                                            // load issued there would sit on the critical path)
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int rg = (mb0 + rb) * ROWS + erow;
        hp[rb] = *(const f32x2*)(a.h_in + hoff0 + rb * hstep);
        if (IS_FIRST) {
            // sum(p) = sum over atom blocks of the row sums left by the previous frame's last
            // layer, added in block order by 16 lanes + a fixed shuffle tree (deterministic)
            // A row's partials are contiguous ([parity][row][atom block], padded to a multiple of four with
            // zeros): 16-byte loads, 16 lanes cover 64 blocks.  (They used to lie [block][row]: one 4-byte load
            // per block and 16 cache lines per load instruction -- at N = 8000, 250 blocks, that made this
            // launch 17.3 us against the other layers' 8.9; N = 2000: 6.2 against 5.3.)
            const int numAp = (a.numA + 3) & ~3;
            const float* rp = a.rs_part + ((size_t)(t & 1) * a.Bp + rg) * numAp;
            float s = 0.f;
            for (int b0 = (tid & 15) * 4; b0 < numAp; b0 += 256) {
                f32x4 v4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int b2 = b0 + 64 * u;
                    v4[u] = *(const f32x4*)(rp + (b2 < numAp ? b2 : b0));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    s += (b0 + 64 * u < numAp) ? (v4[u][0] + v4[u][1]) + (v4[u][2] + v4[u][3]) : 0.f;
            }
            s = row16_sum(s);
            ps[rb] = s;
            if (ab_raw == 0 && (tid & 15) == 0) {
                a.psum[rg] = s;
                a.psum_all[(size_t)t * a.Bp + rg] = s;
            }
        } else {
            ps[rb] = a.psum[rg];
        }
        vld[rb] = true;
        if (WRITE_OUT) vld[rb] = a.valid[(size_t)t * a.Bp + rg] != 0;
#pragma unroll
        for (int i = 0; i < MAX_TAIL; ++i) {
            xt[rb][i] = 0.f;
            if (QRED) qs[rb][i] = 0.f;
            if (i >= a.ntail) continue;
            if (IS_FIRST) {
                // packed input: bin 16*nchunks + i sits in tile nchunks
                xt[rb][i] = rsrc[((size_t)(mb0 + rb) * nft + a.tail_tile) * 256 + rp_pos(erow, i)];
            } else {
                xt[rb][i] = a.xtail[(size_t)i * a.Bp + rg];
                if (QRED) {
                    qs[rb][i] = a.q_in[(size_t)i * a.Bp + rg];
                } else {
                    const float* qp = a.q_in + ((size_t)i * a.Bp + rg) * a.numA;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {   // atom blocks (tid&15) + 16u: numA <= 64 here
                        const int b2 = (tid & 15) + 16 * u;
                        qv[rb][i][u] = b2 < a.numA ? qp[b2] : 0.f;
                    }
                }
            }
        }
    }
    };

    // ---- GEMM:  g[16*RB x 32] = r[16*RB x F] . Dn[F x 32] ---------------------------------------
    // fp16: wave-private staging for the transposed dictionary reads, two slots of two padded blocks
    // (288 bytes per atom octet q: the [4 bins][4 atoms] gathers of a 32-lane half then cover all 64 banks)
    constexpr int TRQ = 144, TRBLK = 4 * TRQ, TRSLOT = 2 * TRBLK;          // halves
    __shared__ __attribute__((aligned(16))) f16 trbuf[HALF ? NW_A * 2 * TRSLOT : 8];
    f16* const trw = trbuf + (HALF ? w * 2 * TRSLOT : 0);
    // this lane's gather (native lane i = l & 15: bin row i >> 2, atom quad i & 3; lane group l >> 4 = slot q):
    // bins 4 (i >> 2) + q, atoms 16 a + 4 (i & 3) .. + 3 -> octet 2a + ((i & 3) >> 1), halves 4 (i & 1) ..
    const int tr_rd = (((l & 3) >> 1) * TRQ) + (4 * ((l & 15) >> 2) + (l >> 4)) * 8 + (l & 1) * 4;
    const int tr_wr = (l >> 4) * TRQ + (l & 15) * 8;
    int tr_par = 0;
    f16x8 bt0, bt1;     // the chunk's two operands: atoms j / 16 + j of the block (j = l & 15), 8 bins per lane
    auto transpose_chunk = [&](int g) {
        f16* slot = trw + tr_par * TRSLOT;
        tr_par ^= 1;
        *(f16x8*)(slot + tr_wr) = bh[g][0];
        *(f16x8*)(slot + TRBLK + tr_wr) = bh[g][1];
        // (wave-private: the LDS serves one wave's instructions in order; the fence keeps the compiler
        // from moving the gathers, which touch other lanes' stores, above them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        using s16x4 = __attribute__((__vector_size__(4 * sizeof(short)))) short;
        using s16x8 = __attribute__((__vector_size__(8 * sizeof(short)))) short;
        using lds_s16x4 = __attribute__((address_space(3))) s16x4;
        const s16x4 t00 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(slot + tr_rd));
        const s16x4 t01 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(slot + TRBLK + tr_rd));
        const s16x4 t10 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(slot + 2 * TRQ + tr_rd));
        const s16x4 t11 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(slot + TRBLK + 2 * TRQ + tr_rd));
        bt0 = __builtin_bit_cast(f16x8, (s16x8)__builtin_shufflevector(t00, t01, 0, 1, 2, 3, 4, 5, 6, 7));
        bt1 = __builtin_bit_cast(f16x8, (s16x8)__builtin_shufflevector(t10, t11, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto compute_chunk = [&](int base, int g) {
        const bool ok = base + g < per_wave;
        if (HALF) {
            transpose_chunk(g);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                f16x8 a8 = ah[g][rb];
                if (!ok) a8 = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                acc[rb][0] = mfma32h(a8, bt0, acc[rb][0]);
                acc[rb][1] = mfma32h(a8, bt1, acc[rb][1]);
            }
            return;
        }
        f32x4 r4[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            r4[rb] = av[g][rb][0];
#pragma unroll
            for (int ks = 1; ks < NP; ++ks) r4[rb] += av[g][rb][ks];
            if (IS_FIRST && ab_raw == 0 && ok) {
                // republish this row tile's x_t chunk at a frame-independent address (cell_b
                // reads it without a dependent frame-index load)
                const int c = w + NW_A * (base + g);
                *(f32x4*)(a.xcur + (size_t)(mb0 + rb) * nft * 256 + 256 * c + l * 4) = r4[rb];
            }
            if (!ok) r4[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                acc[rb][0] = mfma16(r4[rb][s], bv[g][s >> 1][(s & 1) * 2], acc[rb][0]);
                acc[rb][1] = mfma16(r4[rb][s], bv[g][s >> 1][(s & 1) * 2 + 1], acc[rb][1]);
            }
        }
    };
    // sched_barriers pin the interleave (the scheduler would otherwise regroup loads and MFMAs);
    // the MFMAs retire their operands with counted vmcnt waits in issue order.  The G operand
    // slots rotate: chunk i lives in slot i mod G and its loads are issued PF = G-1 chunks ahead,
    // so the register footprint does not grow with the number of chunks and the pipeline never
    // drains between groups.
    // Loads past a wave's last chunk are clamped (they re-read the last chunk: an L1/L2 hit, no
    // branch -- a wave-uniform branch around them or around the MFMAs costs 1-7 %).  When every
    // wave owns a whole number of groups (all 2^k+1 STFT sizes from 512 up: nchunks % 16 == 0)
    // the last group is peeled and issues only the one load that is still needed.
    const bool exact = (a.nchunks % (NW_A * G)) == 0;
    auto full_group = [&](int base, auto epi_tag) {      // every chunk's load runs PF chunks ahead
#pragma unroll
        for (int g = 0; g < G; ++g) {
            load_chunk(base + g + PF, (g + PF) % G);
            __builtin_amdgcn_sched_barrier(0);
            compute_chunk(base, g);
            __builtin_amdgcn_sched_barrier(0);
            if (decltype(epi_tag)::value && g == 0) {
                load_epilogue_operands();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto last_group = [&](int base) {                     // a whole last group: one load is left
        load_chunk(base + PF, PF % G);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            compute_chunk(base, g);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Where the epilogue operand loads go (LATE, picked per shape by pick_a_func).  With two
    // groups per wave (F = 513) issuing them behind the first chunk's MFMAs measures 310 -> 316 k
    // frames/s at the headline shape (N = 200: +1.4 %); a single group (F = 257: -1 %) and long
    // contractions (F = 1025: -5 %) keep them ahead of the loop, where they overlap the initial
    // load latency.  Both orders in ONE kernel behind a runtime switch cost up to 35 % at other
    // shapes, hence the template parameter.  Measured, not derived: the same order without the
    // (unreached) single-group branch below compiles to a kernel that gains nothing (309.9 k).
    if (LATE) {
        if (exact && per_wave <= G) {      // (not reached with the shapes pick_a_func sends here)
            load_chunk(PF, PF % G);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                __builtin_amdgcn_sched_barrier(0);
                compute_chunk(0, g);
                __builtin_amdgcn_sched_barrier(0);
                if (g == 0) load_epilogue_operands();
            }
        } else {
            full_group(0, std::true_type{});
            int base = G;
            for (; base + (exact ? G : 0) < per_wave; base += G) full_group(base, std::false_type{});
            if (exact && base < per_wave) last_group(base);
        }
    } else {
        load_epilogue_operands();
        int base = 0;
        for (; base + (exact ? G : 0) < per_wave; base += G) full_group(base, std::false_type{});
        if (exact) last_group(base);
    }

    // (Rsave: the data arrived long ago -- the loads are older than every operand load the MFMA loop
    // waited for -- and the stores drain under the reduction and the epilogue)
    if (RSAVE && rs_mine) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            f32x4 r4 = rsv[rb][0];
#pragma unroll
            for (int ks = 1; ks < KS; ++ks) r4 += rsv[rb][ks];
            const int row = (mb0 + rb) * ROWS + j;
            if (row < a.B) {
                st_save(a.Rsave + ((size_t)row * a.T + t) * Fp + 16 * ab + 4 * q, r4);   // (tile_unpermute order)
            }
        }
    }

    // ---- cross-wave reduction of the 4 F-splits through LDS --------------------------------
    DRNMF_STAMP(1, 3);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float* rr = red + ((w * RB + rb) * ROWS + 4 * q + v) * ATOMS;
            if (HALF) {     // (transposed dictionary reads: accumulator a holds atoms 16 a + j)
                rr[j] = acc[rb][0][v];
                rr[16 + j] = acc[rb][1][v];
            } else {        // (fp32 cell_a packing: atoms 2 j + a)
                f32x2 pr = {acc[rb][0][v], acc[rb][1][v]};
                *(f32x2*)(rr + 2 * j) = pr;
            }
        }
    __syncthreads();
    DRNMF_STAMP(1, 4);
    if (NW_A > 4 && tid >= 256) return;   // the elementwise epilogue is 256 threads wide

#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int rg = (mb0 + rb) * ROWS + erow;
        const size_t hoff = hoff0 + rb * hstep;
        f32x2 gsum = *(const f32x2*)(red + ((0 * RB + rb) * ROWS + erow) * ATOMS + ec);
#pragma unroll
        for (int ww = 1; ww < NW_A; ++ww) {
            const f32x2 p2 = *(const f32x2*)(red + ((ww * RB + rb) * ROWS + erow) * ATOMS + ec);
            gsum[0] += p2[0];
            gsum[1] += p2[1];
        }

        DRNMF_STAMP(1, 6);
#pragma unroll
        for (int i = 0; i < MAX_TAIL; ++i) {   // rank-1 update per tail bin
            if (i >= a.ntail) continue;
            float rt = xt[rb][i];
            if (IS_FIRST) {
                if (ab_raw == 0 && (tid & 15) == 0) a.xtail[(size_t)i * a.Bp + rg] = xt[rb][i];
            } else {
                if (QRED) {
                    rt -= qs[rb][i];
                } else {
                    float sq = (qv[rb][i][0] + qv[rb][i][1]) + (qv[rb][i][2] + qv[rb][i][3]);
                    const float* qp = a.q_in + ((size_t)i * a.Bp + rg) * a.numA;
                    for (int b2 = (tid & 15) + 64; b2 < a.numA; b2 += 16)   // (numA > 64 with too few
                        sq += qp[b2];                                       // cell_b workgroups: rare)
                    rt -= row16_sum(sq);
                }
                if (ALL_HIDDEN && !HALF && a.Rsave != nullptr && ab_raw == 0 && (tid & 15) == 0 &&
                    rg < a.B)
                    a.Rsave[((size_t)rg * a.T + t) * Fp + 16 * a.tail_tile + i] = rt;
            }
            gsum[0] = fmaf(rt, dt[i][0], gsum[0]);
            gsum[1] = fmaf(rt, dt[i][1], gsum[1]);
        }

        DRNMF_STAMP(1, 7);
        // ---- fused update: soft-threshold / non-negativity projection ----------------------
        f32x2 hn;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float pre;
            if (IS_FIRST) pre = a.u0d * hp[rb][e] + a.u0o * (ps[rb] - hp[rb][e]);
            else pre = hp[rb][e] + a.uko * ps[rb];
            pre += gsum[e] * ia[e] + bs[e];
            hn[e] = fmaxf(pre, 0.f);
        }

        const bool row_live = live && rg < a.B;
#ifdef DRNMF_EXP_NOOUT
        if (WRITE_OUT && row_live && IS_LAST) {
#else
        if (WRITE_OUT && row_live) {
#endif
            // K.rnn masking: a masked step repeats the previous output (zeros before the first
            // valid step)
            float* orow = a.out + ((size_t)rg * a.T + t) * a.out_width + a.out_off;
            if (vld[rb] && n + 1 < a.N && ((a.N | a.out_width | a.out_off) & 1) == 0) {
                st_save(orow + n, hn);          // one 8-byte store (n is even): the usual case
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (n + e < a.N) {
                        float o = hn[e];
                        if (!vld[rb]) o = (t > 0) ? orow[n + e - (ptrdiff_t)a.out_width] : 0.f;
                        orow[n + e] = o;
                    }
                }
            }
        }
        if (!live) continue;
        if (IS_LAST) {
            // ... and keeps the previous state
            f32x2 st = hn;
            if (!vld[rb]) st = IS_FIRST ? hp[rb] : *(const f32x2*)(a.state + hoff);
            *(f32x2*)(a.state + hoff) = st;
            float s = st[0] + st[1];
            s = row16_sum(s);
            if ((tid & 15) == 0)
                a.rs_part[((size_t)((t + 1) & 1) * a.Bp + rg) * ((a.numA + 3) & ~3) + ab] = s;
        } else {
            st_xchg(a.h_out + hoff, hn);
            if (HALF)   // Hp16 block (mb, ab): slot q = ec/8, e = ec%8 (cell_shared.h)
                *(f16x2*)(a.h16_out + ((size_t)(mb0 + rb) * (Np / 32) + ab) * 512 +
                          ((ec >> 3) * 16 + erow) * 8 + (ec & 7)) = f16x2{(f16)hn[0], (f16)hn[1]};
            // tail bins of the next layer's x^: partial dot product over this block's 32 atoms
#pragma unroll
            for (int i = 0; i < MAX_TAIL; ++i) {
                if (i >= a.ntail) continue;
                float sq = hn[0] * dtn[i][0] + hn[1] * dtn[i][1];
                sq = row16_sum(sq);
                if ((tid & 15) == 0) a.q_out[((size_t)i * a.Bp + rg) * a.numA + ab] = sq;
            }
        }
    }
    if (RSAVE && a.Rsave != nullptr && live) {
        // fewer atom blocks than chunks (small dictionaries; those shapes run the Gram form unless
        // it is switched off): the remaining chunks of this atom block, read here
        for (int c = ab + a.numA; c < a.nchunks; c += a.numA) {
            if ((c & (NW_A - 1)) != w) continue;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float* src = arow + rb * astep + 256 * c;
                f32x4 r4 = *(const f32x4*)src;
#pragma unroll
                for (int ks = 1; ks < KS; ++ks) r4 += *(const f32x4*)(src + (size_t)ks * pstride);
                const int row = (mb0 + rb) * ROWS + j;
                if (row < a.B) {
                    st_save(a.Rsave + ((size_t)row * a.T + t) * Fp + 16 * c + 4 * q, r4);
                }
            }
        }
    }
    DRNMF_STAMP(1, 5);
}

// kernelParams array of cell_a_kernel
struct CellAParams {
    void* p[9];
    explicit CellAParams(CellAArgs& a)
        : p{&a.rsrc, &a.Dn, &a.t_rd, &a.Bp, &a.Fp, &a.Np, &a.numA, &a.nchunks, &a} {}
};

__global__ void noop_kernel() {}

// stateful mode (custom_layers.py:296-318; Keras Recurrent stateful=True): the state entering frame
// 0 is supplied by the caller.  One wave per row: pack the row into Hp and leave its sum in atom
// block 0 of parity 0 (sum(p) adds the blocks).
__global__ void __launch_bounds__(256)
load_state_kernel(const float* __restrict__ init, float* __restrict__ state,
                  float* __restrict__ rs_part, int* tptr, int B, int N, int Np, int Bp, int numA,
                  int by_row = 0) {
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wv;
    if (blockIdx.x == 0 && threadIdx.x == 0) { tptr[0] = 0; tptr[16] = 0; }
    if (b >= Bp) return;
    float s = 0.f;
    for (int n = l; n < Np; n += 64) {
        const float v = (b < B && n < N) ? init[(size_t)b * N + n] : 0.f;
        s += v;
        state[((size_t)(b >> 4) * (Np / 16) + (n >> 4)) * 256 + hp_pos(b & 15, n & 15)] = v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (l == 0) {
        // rs_part [2][numA][Bp], or by_row: [2][Bp][numA] (the factored cell_a; numA padded by the caller)
        for (int a = 0; a < 2 * numA; ++a)
            rs_part[by_row ? ((size_t)(a / numA) * Bp + b) * numA + a % numA : (size_t)a * Bp + b] =
                (a == 0) ? s : 0.f;
    }
}

__global__ void __launch_bounds__(256)
store_state_kernel(const float* __restrict__ state, float* __restrict__ out, int B, int N, int Np) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * N) return;
    const int b = (int)(i / N), n = (int)(i % N);
    out[i] = state[((size_t)(b >> 4) * (Np / 16) + (n >> 4)) * 256 + hp_pos(b & 15, n & 15)];
}

// seen[t][row] = 1 iff some frame before t was valid (then the state entering frame t is the last
// valid output, else softplus(log_h0)); used by the backward pass
__global__ void __launch_bounds__(256)
seen_kernel(const unsigned char* __restrict__ valid, unsigned char* __restrict__ seen, int T,
            int Bp) {
    // one wave per row, 64 frames per step: an exclusive prefix-OR from the ballot of the step (a
    // thread per row walking T dependent loads took 53 us at T = 500)
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (row >= Bp) return;
    bool carry = false;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + l;
        const bool v = t < T && valid[(size_t)t * Bp + row] != 0;
        const unsigned long long m = __ballot(v);
        const unsigned long long below = m & ((1ull << l) - 1ull);            // lanes < l
        if (t < T) seen[(size_t)t * Bp + row] = (carry || below != 0ull) ? 1 : 0;
        carry = carry || m != 0ull;
    }
}

// state = softplus(log_h0) for every row (custom_layers.py:203-206, 336-341); row sums of the
// initial state go to atom block 0 of parity 0; frame counter = 0.
__global__ void __launch_bounds__(256)
init_state_kernel(const float* __restrict__ log_h0, float* __restrict__ state,
                  float* __restrict__ rs_part, int* tptr, int N, int Np, int Bp, int numA,
                  int by_row = 0) {
    __shared__ float wsum[4];
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int n = tid; n < Np; n += 256) {
        float v = 0.f;
        if (n < N) {
            const float z = log_h0[n];
            v = (z > 20.f) ? z : log1pf(expf(z));
            s += v;
        }
        for (int b = 0; b < Bp; ++b)   // tile-packed Hp[b/16][n/16][b%16][n%16]
            state[((size_t)(b >> 4) * (Np / 16) + (n >> 4)) * 256 + hp_pos(b & 15, n & 15)] = v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = s;
    __syncthreads();
    const float tot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    // (block 0 of parity 0 of every row; by_row: [2][Bp][numA], see load_state_kernel)
    const int rs_stride = by_row ? numA : 1;
    for (int i = tid; i < 2 * numA * Bp; i += 256)
        rs_part[i] = (i < Bp * rs_stride && i % rs_stride == 0) ? tot : 0.f;
    if (tid == 0) { tptr[0] = 0; tptr[16] = 0; }
}

// KL / beta cell: r = g(x_t, x^) on the packed [Bp][Fp] residual buffer, in place (x^ was left
// there by cell_b in its no-input mode).  g = x/x^ - 1 (KL) or x x^(beta-2) - x^(beta-1)
// (enhance.py:431, 450); padded bins are forced to 0 (0/0 otherwise).  Also the frame-counter
// hand-over of the frame's first kernel (see cell_forward_impl).
// xsave (training forward): x^ of this (frame, layer) is kept, packed as it is, at
// xsave + t * xsave_tstride (the pointer already carries the layer's offset) for the BPTT.
__global__ void __launch_bounds__(256)
resid_div_kernel(const float* __restrict__ xp, float* __restrict__ r, const int* t_rd, int* t_wr,
                 int div, float beta, int F, int Fp, int Bp, float* __restrict__ xsave,
                 size_t xsave_tstride) {
    const int t = *t_rd;
    if (t_wr && blockIdx.x == 0 && threadIdx.x == 0) *t_wr = t;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one block = one 16 x 16 tile
    if (i >= (size_t)Bp * Fp) return;
    const int ft = (int)(blockIdx.x % (Fp / 16));
    const int pos = threadIdx.x;                                   // rp_pos order: ((c&3)*16 + row)*4 + (c>>2)
    const int c = 4 * (pos & 3) + (pos >> 6);
    const int f = 16 * ft + c;
    const float xv = xp[(size_t)t * Bp * Fp + i], xe = r[i];
    if (xsave) xsave[(size_t)t * xsave_tstride + i] = xe;
    float g = 0.f;
    if (f < F) {
        if (div == DRNMF_DIV_KL) g = xv / xe - 1.f;
        else g = xv * powf(xe, beta - 2.f) - powf(xe, beta - 1.f);
    }
    r[i] = g;
}

template <int G, int KS, int RB, bool AH, bool HALF, bool LATE = false>
void* a_func(bool first, bool last, bool qred = false) {
    if (qred && !LATE) {        // (first layers never read the previous layer's odd bins)
        if (last) return (void*)&cell_a_kernel<G, KS, RB, false, true, AH, HALF, false, true>;
        if (!first) return (void*)&cell_a_kernel<G, KS, RB, false, false, AH, HALF, false, true>;
    }
    if (first && last) return (void*)&cell_a_kernel<G, KS, RB, true, true, AH, HALF, LATE>;
    if (first) return (void*)&cell_a_kernel<G, KS, RB, true, false, AH, HALF, LATE>;
    if (last) return (void*)&cell_a_kernel<G, KS, RB, false, true, AH, HALF, LATE>;
    return (void*)&cell_a_kernel<G, KS, RB, false, false, AH, HALF, LATE>;
}

template <int KS, int RB, bool AH, bool HALF>
void* a_func_g(int per_wave, bool first, bool last, bool qred = false) {
    // G operand slots; operand registers G * (4*KS*RB + 8)
    if (per_wave <= 2) return a_func<2, KS, RB, AH, HALF>(first, last, qred);
    // (8 operand slots measured no better at F=1025, N=8000: 26.9 vs 26.0 us per launch; not instantiated)
    // two groups per wave (F = 513), one row block, fp32: epilogue operand loads behind the first chunk
    bool late = RB == 1 && !HALF && per_wave > 4 && per_wave <= 8;
    if (const char* e = measure_env("DRNMF_LATE"))   // tuning aid: 0 = never, 2 = whenever instantiated
        late = atoi(e) == 2 ? (RB == 1 && !HALF && per_wave > 2) : (late && atoi(e) != 0);
    if (late && !qred) return a_func<4, KS, RB, AH, HALF, (RB == 1 && !HALF)>(first, last);
    return a_func<4, KS, RB, AH, HALF>(first, last, qred);
}

template <bool AH, bool HALF>
void* pick_a_func_ah(int per_wave, int KS, int RB, bool first, bool last, bool qred) {
    // row-blocked variants exist for KS <= 2 (workspace_layout never pairs RB > 1 with more)
    if (RB == 2) return KS == 1 ? a_func_g<1, 2, AH, HALF>(per_wave, first, last, qred)
                                : a_func_g<2, 2, AH, HALF>(per_wave, first, last, qred);
    switch (KS) {
        case 1: return a_func_g<1, 1, AH, HALF>(per_wave, first, last, qred);
        case 2: return a_func_g<2, 1, AH, HALF>(per_wave, first, last, qred);
        case 4: return a_func_g<4, 1, AH, HALF>(per_wave, first, last, qred);
        default: return a_func_g<8, 1, AH, HALF>(per_wave, first, last, qred);
    }
}

void* pick_a_func(int nchunks, int KS, int RB, bool first, bool last, bool all_hidden, bool half,
                  bool qred = false) {
    const int per_wave = (nchunks + NW_A - 1) / NW_A;
    if (half) {   // KS = 1, nchunks counts 32-bin chunks
        if (all_hidden)   // (training: every hidden layer goes out in fp32 for the fp32 BPTT)
            return RB == 2 ? a_func_g<1, 2, true, true>(per_wave, first, last, qred)
                           : a_func_g<1, 1, true, true>(per_wave, first, last, qred);
        return RB == 2 ? a_func_g<1, 2, false, true>(per_wave, first, last, qred)
                       : a_func_g<1, 1, false, true>(per_wave, first, last, qred);
    }
    return all_hidden ? pick_a_func_ah<true, false>(per_wave, KS, RB, first, last, qred)
                      : pick_a_func_ah<false, false>(per_wave, KS, RB, first, last, qred);
}

}  // namespace

void persist_query_occupancy(int device, int* per_cu, int* n_cu) {
    persist_query_occupancy_impl(device, per_cu, n_cu);
}

#ifdef DRNMF_TIMELINE
extern "C" int32_t drnmf_debug_persist_timeline(void* out_host, size_t bytes) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_ptl),
                               bytes < sizeof(g_ptl) ? bytes : sizeof(g_ptl)) == hipSuccess ? 0 : -3;
}
extern "C" int32_t drnmf_debug_timeline(void* out_host, size_t bytes) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_timeline),
                               bytes < sizeof(g_timeline) ? bytes : sizeof(g_timeline)) == hipSuccess
               ? 0 : -3;
}
#endif

extern "C" int32_t drnmf_cell_launches_per_frame(const drnmf_cell_desc_t* d) {
    if (!d || d->B <= 0 || d->T <= 0 || d->F <= 0 || d->N <= 0 || d->K <= 0) return 0;
    if (d->divergence != DRNMF_DIV_ED) return 3 * d->K;
    return gram_wanted(d) ? d->K - 1 : 2 * d->K - 1;
}

extern "C" size_t drnmf_cell_workspace_bytes(const drnmf_cell_desc_t* d) {
    if (!d || d->B <= 0 || d->T <= 0 || d->F <= 0 || d->N <= 0) return 0;
    return workspace_layout(d).total;
}

// ---- Gram form (cell_gram.h): K-1 launches per frame --------------------------------------------
static int32_t cell_forward_gram(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                 float mask_value, const void* params, const float* log_h0,
                                 float u0_diag, float u0_off, float uk_off, float* h_out,
                                 void* workspace, hipStream_t stream, const float* initial_state) {
    const Workspace W = workspace_layout(d);
    const ParamsLayout L = params_layout(d);
    char* ws = (char*)workspace;
    const char* pb = (const char*)params;
    const int K = d->K, NAC = W.Np / 16, numO = W.numO, numM = W.Bp / ROWS;
    unsigned char* valid = (unsigned char*)(ws + W.off_valid);
    float* hb[2] = {(float*)(ws + W.off_h0), (float*)(ws + W.off_h1)};
    float* state = (float*)(ws + W.off_state);
    float* rs_part = (float*)(ws + W.off_rs);
    float* qb[2] = {(float*)(ws + W.off_q0), (float*)(ws + W.off_q1)};
    float* Cp = (float*)(ws + W.off_cp);
    float* xpad = (float*)(ws + W.off_xpad);
    int* tA = (int*)(ws + W.off_t);
    int* tB = tA + 16;
    const size_t cstride = (size_t)W.Bp * W.Np;
    auto ia_of = [&](int k) { return (const float*)(pb + L.off_inv_alpha) + (size_t)k * L.Np; };
    auto b_of = [&](int k) { return (const float*)(pb + L.off_bias) + (size_t)k * L.Np; };
    auto G_of = [&](int k) {
        return (const float*)(pb + L.off_gram) + (d->n_D == 1 ? 0 : (size_t)k * L.Np * L.Np);
    };
    auto DnT_of = [&](int k) {
        return (const float*)(pb + L.off_dnT) + (d->n_D == 1 ? 0 : (size_t)k * L.Np * L.Fp);
    };
    // ---- prologue: validity flags, initial state, c_k for every frame and layer -------------------
    {
        const size_t rows = (size_t)d->T * W.Bp;
        hipLaunchKernelGGL(pack_input_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                           stream, x, (float*)(ws + W.off_xp), valid, mask_value, d->B, d->T, d->F,
                           W.Bp, W.Fp);
        hipLaunchKernelGGL(seen_kernel, dim3((W.Bp + 3) / 4), dim3(256), 0, stream, valid,
                           (unsigned char*)(ws + W.off_seen), d->T, W.Bp);
        if (initial_state)
            hipLaunchKernelGGL(load_state_kernel, dim3((W.Bp + 3) / 4), dim3(256), 0, stream,
                               initial_state, state, rs_part, tA, d->B, d->N, W.Np, W.Bp,
                               (numO + 1) / 2);
        else
            hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(256), 0, stream, log_h0, state,
                               rs_part, tA, d->N, W.Np, W.Bp, (numO + 1) / 2);
        DRNMF_HIP(h, hipGetLastError());
        // padded batch rows of c are never produced by the GEMM: keep them finite
        if (W.Bp != d->B)
            DRNMF_HIP(h, hipMemsetAsync(Cp, 0, (size_t)W.cp_frames * K * cstride * 4, stream));
    }
    // c_k of one block of GRAM_TB frames into its ring slot (stream-ordered behind the chain of the
    // block that used the slot before)
    const bool cp_full = W.cp_full;       // every frame's c_k resident: one product up front
    const int cp_mask = cp_full ? 0x7fffffff : 2 * GRAM_TB - 1;
    auto compute_block = [&](int j) -> int32_t {
        if (cp_full && j > 0) return DRNMF_OK;
        const int t0 = cp_full ? 0 : j * GRAM_TB;
        if (t0 >= d->T) return DRNMF_OK;
        const int tbc = cp_full ? d->T : (d->T - t0 < GRAM_TB ? d->T - t0 : GRAM_TB);
        const size_t tot = (size_t)d->B * tbc * W.Fp;
        hipLaunchKernelGGL(gather_block_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                           stream, x, xpad, d->B, d->T, d->F, W.Fp, t0, tbc);
        DRNMF_HIP(h, hipGetLastError());
        float* slot = Cp + (size_t)(t0 & cp_mask) * K * cstride;
        if (d->n_D == K && K > 1) {      // untied: all layers in one product (EpiCPackAll)
            gemm::Operands g{xpad, DnT_of(0), (int64_t)d->B * tbc, K * W.Np, W.Fp, W.Fp, W.Fp};
            EpiCPackAll epi{slot, ia_of(0), b_of(0), tbc, K, NAC, W.Np, cstride};
            DRNMF_HIP(h, gemm::launch(g, epi, stream));
            return DRNMF_OK;
        }
        if (d->n_D == 1 && K > 1) {      // tied: one product, K epilogue writes (EpiCPackTied)
            gemm::Operands g{xpad, DnT_of(0), (int64_t)d->B * tbc, W.Np, W.Fp, W.Fp, W.Fp};
            EpiCPackTied epi{slot, ia_of(0), b_of(0), tbc, K, NAC, W.Np, cstride};
            DRNMF_HIP(h, gemm::launch(g, epi, stream));
            return DRNMF_OK;
        }
        for (int k = 0; k < K; ++k) {
            gemm::Operands g{xpad, DnT_of(k), (int64_t)d->B * tbc, W.Np, W.Fp, W.Fp, W.Fp};
            EpiCPack epi{slot + (size_t)k * cstride, ia_of(k), b_of(k), tbc, K, NAC, cstride};
            DRNMF_HIP(h, gemm::launch(g, epi, stream));
        }
        return DRNMF_OK;
    };
    {
        int32_t rc = compute_block(0);
        if (rc) return rc;
        rc = compute_block(1);
        if (rc) return rc;
        const size_t tot = cstride;
        hipLaunchKernelGGL(gram_init_q_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                           stream, state, Cp, qb[0], u0_diag, u0_off, tot);
        DRNMF_HIP(h, hipGetLastError());
    }
    const dim3 grid(8u * (unsigned)numM, (unsigned)(round_up(numO, 8) / 8));
    auto make = [&](int k, int par) {
        GramFwdArgs a;
        memset(&a, 0, sizeof(a));
        a.G = G_of(k);
        a.a_in = (k == 1) ? qb[par] : hb[(k - 1) & 1];
        a.ia = ia_of(k);
        a.Cp = Cp;
        a.h_out = hb[k & 1];
        a.state = state;
        a.qnext = qb[par ^ 1];
        a.rs_part = rs_part;
        a.psum = (float*)(ws + W.off_psum);
        a.psum_all = (float*)(ws + W.off_psum_all);
        a.valid = valid;
        a.out = h_out;
        if (K == 2) {                    // one launch per frame: the two counters alternate
            a.t_rd = par ? tB : tA; a.t_wr = par ? tA : tB; a.t_wr_add = 1;
        } else if (k == 1) { a.t_rd = tB; a.t_wr = tA; a.t_wr_add = 0; }
        else if (k == K - 1) { a.t_rd = tA; a.t_wr = tB; a.t_wr_add = 1; }
        else { a.t_rd = tA; a.t_wr = nullptr; a.t_wr_add = 0; }
        a.u0d = u0_diag; a.u0o = u0_off; a.uko = uk_off;
        a.B = d->B; a.T = d->T; a.N = d->N; a.K = K; a.k = k; a.Bp = W.Bp; a.Np = W.Np;
        a.numO = numO;
        a.out_width = d->return_all_hidden ? d->N * K : d->N;
        a.all_hidden = d->return_all_hidden ? 1 : 0;
        a.par = par;
        a.cp_mask = cp_mask;
        return a;
    };
    std::vector<uint64_t> key = {
        0x6A4Dull, (uint64_t)d->B, (uint64_t)d->T, (uint64_t)d->F, (uint64_t)d->N, (uint64_t)K,
        (uint64_t)d->n_D, (uint64_t)d->return_all_hidden, (uint64_t)(uintptr_t)params,
        (uint64_t)(uintptr_t)h_out, (uint64_t)(uintptr_t)workspace};
    {
        uint32_t b0, b1, b2;
        memcpy(&b0, &u0_diag, 4); memcpy(&b1, &u0_off, 4); memcpy(&b2, &uk_off, 4);
        key.push_back(b0); key.push_back(b1); key.push_back(b2);
    }
    auto get_graph = [&](int frames, hipGraphExec_t* out) -> int32_t {
        std::vector<uint64_t> gkey = key;
        gkey.push_back((uint64_t)frames);
        for (auto& g : h->graphs)
            if (g.key == gkey) { g.last_stream = stream; *out = g.exec; return DRNMF_OK; }
        const int32_t erc = graph_cache_make_room(h, stream, 24);
        if (erc) return erc;
        GraphEntry ge;
        ge.key = gkey;
        DRNMF_HIP(h, hipGraphCreate(&ge.graph, 0));
        hipGraphNode_t last = nullptr;
        for (int rep = 0; rep < frames; ++rep) {
            for (int k = 1; k < K; ++k) {
                GramFwdArgs a = make(k, rep & 1);
                void* kp[1] = {&a};
                hipKernelNodeParams p;
                memset(&p, 0, sizeof(p));
                p.func = pick_gram_fwd(NAC, k == 1, k == K - 1);
                p.gridDim = grid;
                p.blockDim = dim3(64 * NW_G);
                p.kernelParams = kp;
                hipGraphNode_t node;
                DRNMF_HIP(h, hipGraphAddKernelNode(&node, ge.graph, last ? &last : nullptr,
                                                   last ? 1 : 0, &p));
                last = node;
            }
        }
        DRNMF_HIP(h, hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0));
        ge.last_stream = stream;
        h->graphs.push_back(ge);
        *out = ge.exec;
        return DRNMF_OK;
    };
    // Few tiles per row tile: every row tile runs as an independent persistent chain on its own XCD
    // (cell_gram_persist.h), one launch per block of frames.  DRNMF_PERSIST=0, a shape outside its
    // limits, or persistent launches of another stream still in flight on this handle keep the
    // launch-per-layer-step graphs.
    if (persist_shape_ok(h, numM, numO, K) && persist_admit(h, stream)) {
        unsigned* bar = (unsigned*)(ws + W.off_t + 256);
        const int tb = cp_full ? d->T : GRAM_TB;          // frames per launch
        for (int j = 0; j * tb < d->T; ++j) {
            const int t0 = j * tb;
            const int t1 = (j + 1) * tb < d->T ? (j + 1) * tb : d->T;
            GramPersistArgs a;
            memset(&a, 0, sizeof(a));
            a.G = (const float*)(pb + L.off_gram);
            a.g_stride = d->n_D == 1 ? 0 : (size_t)L.Np * L.Np;
            a.ia = (const float*)(pb + L.off_inv_alpha);
            a.Cp = Cp;
            a.hb[0] = hb[0]; a.hb[1] = hb[1];
            a.qb[0] = qb[0]; a.qb[1] = qb[1];
            a.state = state;
            a.rs_part = rs_part;
            a.psum = (float*)(ws + W.off_psum);
            a.psum_all = (float*)(ws + W.off_psum_all);
            a.valid = valid;
            a.out = h_out;
            a.bar = bar;
            a.host_flag = h->persist_flag;
            a.u0d = u0_diag; a.u0o = u0_off; a.uko = uk_off;
            a.B = d->B; a.T = d->T; a.N = d->N; a.K = K; a.Bp = W.Bp; a.Np = W.Np; a.numO = numO;
            a.numM = numM;
            a.out_width = d->return_all_hidden ? d->N * K : d->N;
            a.all_hidden = d->return_all_hidden ? 1 : 0;
            a.t0 = t0; a.nfr = t1 - t0;
            a.cp_mask = cp_mask;
            a.nwait = persist_nwait(numO);
            DRNMF_HIP(h, hipMemsetAsync(bar, 0, PERSIST_SYNC_BYTES, stream));   // arrivals, abort, XCC masks
            void* kp[1] = {&a};
            DRNMF_HIP(h, hipLaunchKernel(pick_persist_fwd(NAC), dim3(8u * (unsigned)(numO * persist_rounds(numM))),
                                         dim3(64 * (NW_G + 1)), kp,
                                         persist_fwd_lds(K, d->return_all_hidden != 0), stream));
            int32_t rc = compute_block(j + 2);
            if (rc) return rc;
        }
        persist_mark(h, stream);
        return DRNMF_OK;
    }
    // an even number of frames per graph (the frame parity of every node is then static) that
    // divides the block length
    int fpg = 2;
    while (fpg * 2 <= GRAM_TB && fpg * 2 * (K - 1) <= 800) fpg *= 2;
    hipGraphExec_t ex = nullptr;
    for (int j = 0; j * GRAM_TB < d->T; ++j) {
        const int t1 = (j + 1) * GRAM_TB < d->T ? (j + 1) * GRAM_TB : d->T;
        int t = j * GRAM_TB;
        if (t1 - t >= fpg) {
            int32_t rc = get_graph(fpg, &ex);
            if (rc) return rc;
            for (; t + fpg <= t1; t += fpg) DRNMF_HIP(h, hipGraphLaunch(ex, stream));
        }
        if (t1 - t >= 2) {
            int32_t rc = get_graph(2, &ex);
            if (rc) return rc;
            for (; t + 2 <= t1; t += 2) DRNMF_HIP(h, hipGraphLaunch(ex, stream));
        }
        if (t < t1) {                    // (t is even here: a single frame of parity 0)
            int32_t rc = get_graph(1, &ex);
            if (rc) return rc;
            DRNMF_HIP(h, hipGraphLaunch(ex, stream));
        }
        int32_t rc = compute_block(j + 2);
        if (rc) return rc;
    }
    return DRNMF_OK;
}

// What one sub-batch of a split call still has to replay once its prologue is enqueued: the caller
// interleaves the graph launches of all sub-batches (see the split branch of cell_forward_impl).
struct FwdPlan {
    hipGraphExec_t exec_n = nullptr, exec_1 = nullptr;
    int fpg = 1, n_full = 0, n_rem = 0;
};

static int32_t cell_forward_impl(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                 float mask_value, const void* params, const float* log_h0,
                                 float u0_diag, float u0_off, float uk_off, float* h_out,
                                 void* workspace, size_t workspace_bytes, void* stream_,
                                 int profile_frames, float* out_us,
                                 const float* initial_state = nullptr, float* final_state = nullptr,
                                 bool allow_split = true, FwdPlan* plan = nullptr) {
    if (!h) return DRNMF_ERR_INVALID_ARG;
    // (No implicit look at the handle's fault word here: a host-side read-and-clear at the entry of the
    // NEXT call races with the stream-ordered drnmf_status_take_device of a training step -- the host
    // could consume the word before the device-side guard of the fused Adam launch has seen it.  Faults
    // are reported by drnmf_check_status / drnmf_status_take_device only.)
    int rc = validate_cell_desc(h, d);
    if (rc) return rc;
    if (!plan) ++h->call_seq;        // (a top-level call: the graphs it takes are pinned until it returns)
    if (d->divergence != DRNMF_DIV_ED)
        DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED,
                   "cell_forward is the reference's (Euclidean) cell; KL / beta: drnmf_cell_forward_ista");
    if (!x || !params || !log_h0 || !h_out || !workspace)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "cell_forward: NULL pointer argument");
    const Workspace W = workspace_layout(d, allow_split);
    if (workspace_bytes < W.total)
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "cell_forward: workspace %zu < required %zu",
                   workspace_bytes, W.total);
    if (((uintptr_t)workspace & 255) || ((uintptr_t)params & 255))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "workspace/params must be 256-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    auto store_final = [&](const Workspace& Wl) -> int32_t {
        if (!final_state) return DRNMF_OK;
        const size_t tot = (size_t)d->B * d->N;
        hipLaunchKernelGGL(store_state_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream,
                           (const float*)((char*)workspace + Wl.off_state), final_state, d->B, d->N, Wl.Np);
        DRNMF_HIP(h, hipGetLastError());
        return DRNMF_OK;
    };
    if (W.gram) {
        if (profile_frames > 0)
            DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED,
                       "cell_profile brackets the factored launches; this call takes the Gram form");
        rc = cell_forward_gram(h, d, x, mask_value, params, log_h0, u0_diag, u0_off, uk_off,
                               h_out, workspace, stream, initial_state);
        return rc ? rc : store_final(W);
    }
    if (W.split > 1 && profile_frames <= 0) {
        // ---- independent sub-batches on the caller's stream + side streams (Workspace::split) ----
        for (int i = 0; i + 1 < W.split; ++i) {
            if (!h->side_stream[i]) DRNMF_HIP(h, hipStreamCreateWithFlags(&h->side_stream[i], hipStreamNonBlocking));
            if (!h->join_ev[i]) DRNMF_HIP(h, hipEventCreateWithFlags(&h->join_ev[i], hipEventDisableTiming));
        }
        if (!h->fork_ev) DRNMF_HIP(h, hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
        DRNMF_HIP(h, hipEventRecord(h->fork_ev, stream));
        const size_t width = (size_t)d->N * (d->return_all_hidden ? d->K : 1);
        // (every sub-batch may take two graphs from the cache; entries taken by THIS call are pinned -- an
        // insertion for a later sub-batch never retires an executable an earlier one's plan holds -- and
        // cache hits cost nothing: room is made only when an entry is actually inserted)
        int32_t first_err = DRNMF_OK;
        // (from here on a failing HIP call is RECORDED and the code falls through to the join loop: side
        // streams may already be writing h_out / the workspace, and the caller's stream must not run ahead
        // of them -- ADVICE r4)
        auto note = [&](hipError_t e, const char* what) {
            if (e == hipSuccess || first_err) return;
            first_err = DRNMF_ERR_HIP;
            snprintf(h->err, sizeof(h->err), "cell_forward (split): %s failed: %s", what, hipGetErrorString(e));
        };
        FwdPlan plans[MAX_SPLIT];
        hipStream_t sts[MAX_SPLIT];
        int nsub = 0;
        for (int sidx = 0; sidx < W.split && !first_err; ++sidx) {
            const int b0 = sidx * W.split_rows;
            if (b0 >= d->B) break;
            drnmf_cell_desc_t ds = *d;
            ds.B = (d->B - b0 < W.split_rows) ? d->B - b0 : W.split_rows;
            hipStream_t st = sidx == 0 ? stream : h->side_stream[sidx - 1];
            sts[sidx] = st;
            nsub = sidx + 1;
            if (sidx > 0) {
                note(hipStreamWaitEvent(st, h->fork_ev, 0), "hipStreamWaitEvent");
                if (first_err) break;
            }
            // prologue (input packing, initial state) enqueued; the frame graphs come back as a plan
            const int32_t src = cell_forward_impl(
                h, &ds, x + (size_t)b0 * d->T * d->F, mask_value, params, log_h0, u0_diag, u0_off, uk_off,
                h_out + (size_t)b0 * d->T * width, (char*)workspace + (size_t)sidx * W.split_bytes,
                W.split_bytes, (void*)st, 0, nullptr,
                initial_state ? initial_state + (size_t)b0 * d->N : nullptr, nullptr, false, &plans[sidx]);
            if (src && !first_err) first_err = src;
            if (src) plans[sidx] = FwdPlan{};
        }
        // Round-robin over the sub-batches, one graph launch (a block of frames) each: a host that
        // enqueued one sub-batch's whole sequence first would fill the hardware queue with it (T = 2000:
        // ~100 k packets) and the side streams would only start when the first was nearly done --
        // measured: the 250 x 2000 slab at 445 k frames/s, against 656 k with 200-frame sequences.
        {
            int most = 0;
            for (int i = 0; i < nsub; ++i) most = plans[i].n_full > most ? plans[i].n_full : most;
            for (int c = 0; c < most && !first_err; ++c)
                for (int i = 0; i < nsub && !first_err; ++i)
                    if (c < plans[i].n_full) note(hipGraphLaunch(plans[i].exec_n, sts[i]), "hipGraphLaunch");
            most = 0;
            for (int i = 0; i < nsub; ++i) most = plans[i].n_rem > most ? plans[i].n_rem : most;
            for (int c = 0; c < most && !first_err; ++c)
                for (int i = 0; i < nsub && !first_err; ++i)
                    if (c < plans[i].n_rem) note(hipGraphLaunch(plans[i].exec_1, sts[i]), "hipGraphLaunch");
        }
        for (int sidx = 0; sidx < nsub; ++sidx) {
            const int b0 = sidx * W.split_rows;
            hipStream_t st = sts[sidx];
            if (final_state && !first_err) {
                drnmf_cell_desc_t ds = *d;
                ds.B = (d->B - b0 < W.split_rows) ? d->B - b0 : W.split_rows;
                const Workspace Ws = workspace_layout(&ds, false);
                const size_t tot = (size_t)ds.B * d->N;
                hipLaunchKernelGGL(store_state_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st,
                                   (const float*)((char*)workspace + (size_t)sidx * W.split_bytes + Ws.off_state),
                                   final_state + (size_t)b0 * d->N, ds.B, d->N, Ws.Np);
                note(hipGetLastError(), "store_state_kernel launch");
            }
            // (join also after a failed sub-call: the caller's stream must not run ahead of side work)
            if (sidx > 0) {
                const hipError_t e1 = hipEventRecord(h->join_ev[sidx - 1], st);
                note(e1, "hipEventRecord");
                if (e1 == hipSuccess) note(hipStreamWaitEvent(stream, h->join_ev[sidx - 1], 0), "hipStreamWaitEvent");
            }
        }
        return first_err;
    }
    const ParamsLayout L = params_layout(d);
    char* ws = (char*)workspace;
    const char* pb = (const char*)params;
    float* xp = (float*)(ws + W.off_xp);
    unsigned char* valid = (unsigned char*)(ws + W.off_valid);
    float* rpart = (float*)(ws + W.off_rpart);
    float* hb[2] = {(float*)(ws + W.off_h0), (float*)(ws + W.off_h1)};
    f16* hb16[2] = {(f16*)(ws + W.off_h16_0), (f16*)(ws + W.off_h16_1)};
    f16* r16 = (f16*)(ws + W.off_r16);
    float* state = (float*)(ws + W.off_state);
    float* rs_part = (float*)(ws + W.off_rs);
    float* psum = (float*)(ws + W.off_psum);
    int* tA = (int*)(ws + W.off_t);      // frame index read by every kernel of a frame
    int* tB = tA + 16;                   // next frame index, published by the last kernel
    const int K = d->K;
    const bool half = d->operand_f16 != 0;

    // ---- per-call prologue ------------------------------------------------------------------
    {
        const size_t rows = (size_t)d->T * W.Bp;
        hipLaunchKernelGGL(pack_input_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                           stream, x, xp, valid, mask_value, d->B, d->T, d->F, W.Bp, W.Fp,
                           half ? (f16*)(ws + W.off_xp16) : (f16*)nullptr);
        hipLaunchKernelGGL(seen_kernel, dim3((W.Bp + 3) / 4), dim3(256), 0, stream, valid,
                           (unsigned char*)(ws + W.off_seen), d->T, W.Bp);
        if (initial_state)
            hipLaunchKernelGGL(load_state_kernel, dim3((W.Bp + 3) / 4), dim3(256), 0, stream,
                               initial_state, state, rs_part, tA, d->B, d->N, W.Np, W.Bp, round_up(W.numA, 4), 1);
        else
            hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(256), 0, stream, log_h0, state,
                               rs_part, tA, d->N, W.Np, W.Bp, round_up(W.numA, 4), 1);
        DRNMF_HIP(h, hipGetLastError());
        // fp16 mode: cell_b writes the MFMA bin tiles of Rp16 only; the padding of the last 32-bin
        // chunk (and the odd-bin tile) must be finite: it meets zero dictionary slots
        if (half) DRNMF_HIP(h, hipMemsetAsync(r16, 0, (size_t)W.Bp * W.Fp * 2, stream));
        // training: the saved residuals' padding bins (past the odd bins) are never written
        if (W.off_rsave != 0 && W.ntail > 0)
            DRNMF_HIP(h, hipMemsetAsync(ws + W.off_rsave, 0,
                                        (size_t)(K - 1) * d->B * d->T * W.Fp * 4, stream));
    }

    // ---- one frame = 2K-1 launches, as a cached hipGraph -------------------------------------
    // measurement aid: DRNMF_ABLATE=1 launches the same grids but every workgroup exits at once
    // (launch/boundary floor of the frame graph; results are garbage)
    const bool ablate = measure_env("DRNMF_ABLATE") != nullptr;
    std::vector<uint64_t> key = {
        (uint64_t)d->B, (uint64_t)d->T, (uint64_t)d->F, (uint64_t)d->N, (uint64_t)d->K,
        (uint64_t)d->n_D, (uint64_t)d->return_all_hidden + 2 * (uint64_t)(d->operand_f16 != 0),
        (uint64_t)(uintptr_t)params,
        (uint64_t)(uintptr_t)h_out, (uint64_t)(uintptr_t)workspace, (uint64_t)ablate,
        // the layout choices baked into the nodes: a sub-batch of a split call and a direct call of the same
        // B take different row blockings (workspace_layout's `need`), and the tuning variables move them too
        (uint64_t)W.RB | ((uint64_t)W.RBa << 8) | ((uint64_t)W.KS << 16) | ((uint64_t)W.nch_ks << 24)};
    {
        uint32_t b0, b1, b2;
        memcpy(&b0, &u0_diag, 4); memcpy(&b1, &u0_off, 4); memcpy(&b2, &uk_off, 4);
        key.push_back(b0); key.push_back(b1); key.push_back(b2);
    }
    const bool use_graph = tune_env("DRNMF_NO_GRAPH") == nullptr && profile_frames <= 0;
    // frames per graph: the kernels advance the device-side frame counters themselves, so a
    // graph may hold several frames' worth of nodes (fewer graph launches from the host)
    // (measured at the C2 shape: 1 / 2 / 4 / 8 / 20 frames per graph -> 251.1k / 253.9k / 255.6k /
    // 256.1k / 256.3k frames/s).  The remainder T mod FPG runs on a one-frame graph.
    int fpg_max = 800 / (2 * d->K - 1);      // ~800 kernel nodes per graph
    fpg_max = fpg_max < 1 ? 1 : (fpg_max > 64 ? 64 : fpg_max);
    if (const char* e = tune_env("DRNMF_FPG")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 64) fpg_max = v;
    }
    if (fpg_max > d->T) fpg_max = d->T;

    const int numM = W.Bp / (ROWS * W.RB), nft = W.nft_main;   // MFMA bin tiles (tail bins handled apart)
    const dim3 grid_a(8u * (unsigned)(W.Bp / (ROWS * W.RBa)), (unsigned)(round_up(W.numA, 8) / 8));
    const dim3 grid_b(8u * (unsigned)numM, (unsigned)(round_up(nft * W.KS, 8) / 8));
    const bool qred = qred_wanted(W.numA, W.ntail, nft, W.KS, W.RB);
    // per stored layer: the fp32 packing of cell_b (Fp*Np*4 bytes; cell_a's lies at off_dnA), or the ONE
    // fp16 packing both kernels read (Fp*Np*2 bytes)
    const char* Dn_base = pb + L.off_dn;
    const size_t dstride = (size_t)L.Fp * L.Np * (half ? 2 : 4);
    auto Dn_of = [&](int k) { return Dn_base + (d->n_D == 1 ? 0 : (size_t)k * dstride); };
    auto DnB_of = [&](int k) { return Dn_of(k); };
    auto DnA_of = [&](int k) {      // cell_a's operand: the fp16 packing again, or the fp32 cell_a packing
        return half ? Dn_of(k) : pb + L.off_dnA + (d->n_D == 1 ? 0 : (size_t)k * dstride);
    };
    auto tail_of = [&](int k) {
        return (const float*)(pb + L.off_tail) + (d->n_D == 1 ? 0 : (size_t)k * MAX_TAIL * L.Np);
    };

    // fp16 mode, untied layers: the cell_a launch ahead of a cell_b prefetches that cell_b's dictionary
    // (cell_a_kernel's fifth wave; DRNMF_PF=0 switches it off: measurement aid)
    const bool pf_on = half && d->n_D == K && K > 1 && !(measure_env("DRNMF_PF") && atoi(measure_env("DRNMF_PF")) == 0);
    auto make_a = [&](int k) {
        CellAArgs a;
        a.Dn = DnA_of(k);
        a.inv_alpha = (const float*)(pb + L.off_inv_alpha) + (size_t)k * L.Np;
        a.bias = (const float*)(pb + L.off_bias) + (size_t)k * L.Np;
        a.rsrc = (k == 0) ? xp : (half ? (const float*)r16 : rpart);
        a.h_in = (k == 0) ? state : hb[(k - 1) & 1];
        a.h_out = (k == K - 1) ? state : hb[k & 1];
        a.h16_out = hb16[k & 1];
        a.state = state;
        a.rs_part = rs_part;
        a.psum = psum;
        a.psum_all = (float*)(ws + W.off_psum_all);
        a.valid = valid;
        a.out = h_out;
        // frame counters: the first kernel of a frame reads tB and republishes it as tA; the last
        // publishes tB = t+1; nobody reads a counter in the kernel that writes it.  K == 1 has a
        // single kernel per frame and uses the separate advance kernel instead.
        if (K == 1) { a.t_rd = tA; a.t_wr = nullptr; a.t_wr_add = 0; }
        else if (k == 0) { a.t_rd = tB; a.t_wr = tA; a.t_wr_add = 0; }
        else if (k == K - 1) { a.t_rd = tA; a.t_wr = tB; a.t_wr_add = 1; }
        else { a.t_rd = tA; a.t_wr = nullptr; a.t_wr_add = 0; }
        a.u0d = u0_diag; a.u0o = u0_off; a.uko = uk_off;
        a.B = d->B; a.T = d->T; a.N = d->N; a.Bp = W.Bp; a.Fp = W.Fp; a.Np = W.Np;
        a.numA = W.numA; a.nchunks = half ? (nft + 1) / 2 : nft; a.KS = W.KS; a.ntail = W.ntail;
        a.tail_tile = nft;
        a.Dtail = tail_of(k);
        a.Dtail_next = tail_of(k + 1 < K ? k + 1 : k);
        float* qp = (float*)(ws + W.off_qpart);
        const size_t qstride = (size_t)W.numA * MAX_TAIL * W.Bp;
        a.q_in = qred ? (const float*)(ws + W.off_qsum)   // layer k-1's partials, summed by cell_b
                      : qp + (size_t)((k + 1) & 1) * qstride;
        a.q_out = qp + (size_t)(k & 1) * qstride;
        a.xtail = (float*)(ws + W.off_xtail);
        a.xcur = (float*)(ws + W.off_xcur);
        a.out_width = d->return_all_hidden ? d->N * K : d->N;
        a.out_off = d->return_all_hidden ? k * d->N : 0;
        a.write_out = (d->return_all_hidden || k == K - 1) ? 1 : 0;
#ifdef DRNMF_MEASURE
        a.ablate = tune_env("DRNMF_ABLATE_A") ? atoi(tune_env("DRNMF_ABLATE_A")) : 0;
#endif
        a.Rsave = (W.off_rsave != 0 && k >= 1)
                      ? (float*)(ws + W.off_rsave) + (size_t)(k - 1) * d->B * d->T * W.Fp : nullptr;
        // fp16 mode, a cell_b launch behind this one: its dictionary goes into the L2s meanwhile
        a.Dn_pf = nullptr;
        a.pf_tiles = nft;
        // (the prefetching wave starts ~2 us into the launch: the working waves' first fetches are the
        // latency-critical part.  Delay in units of 512 cycles, us per launch at the config-5 shape:
        // 0 8.58-8.62, 2 8.58, 4 8.55, 8 8.47, 12 8.42, 16 8.54, 24 9.28, 32 10.2; no prefetch 8.82)
        a.pf_sleep = measure_env("DRNMF_PF_SLEEP") ? atoi(measure_env("DRNMF_PF_SLEEP")) : 10;
        a.x16 = half ? (const f16*)(ws + W.off_xp16) : nullptr;
        if (pf_on && k + 1 < K) a.Dn_pf = DnB_of(k + 1);
        return a;
    };
    auto make_b = [&](int k) {   // between layer k and k+1
        CellBArgs b;
        b.Dn_next = DnB_of(k + 1);
        b.h = half ? (const float*)hb16[k & 1] : hb[k & 1];
        b.xp = (const float*)(ws + W.off_xcur);
        b.rpart = half ? (float*)r16 : rpart;
        b.t_rd = tA;
        b.Bp = W.Bp; b.Fp = W.Fp; b.Np = W.Np; b.nft = nft; b.KS = W.KS;
        b.logKS = 0;
        while ((1 << b.logKS) < W.KS) ++b.logKS;
        b.nch_ks = W.nch_ks;
#ifdef DRNMF_MEASURE
        b.ablate = tune_env("DRNMF_ABLATE_B") ? atoi(tune_env("DRNMF_ABLATE_B")) : 0;
#endif
        b.q_in = (const float*)(ws + W.off_qpart) + (size_t)(k & 1) * W.numA * MAX_TAIL * W.Bp;
        b.qsum = (float*)(ws + W.off_qsum);
        b.numA = W.numA; b.ntail = W.ntail;
        return b;
    };

    if (!use_graph) {
        // plain launches; in profile mode every launch is bracketed by HIP events on the stream
        const int T_run = profile_frames > 0 ? (profile_frames < d->T ? profile_frames : d->T)
                                             : d->T;
        std::vector<hipEvent_t> ev;
        std::vector<int> kind;   // 0 = cell_a middle layer, 1 = cell_b, 2 = other
        auto mark = [&](int k_) -> hipError_t {
            if (profile_frames <= 0) return hipSuccess;
            hipEvent_t e;
            hipError_t er = hipEventCreate(&e);
            if (er != hipSuccess) return er;
            ev.push_back(e);
            kind.push_back(k_);
            return hipEventRecord(e, stream);
        };
        for (int t = 0; t < T_run; ++t) {
            for (int k = 0; k < K; ++k) {
                CellAArgs a = make_a(k);
                CellAParams kpa(a);
                void** kp = kpa.p;
                DRNMF_HIP(h, mark((k > 0 && k < K - 1) ? 0 : 2));
                DRNMF_HIP(h, hipLaunchKernel(pick_a_func(a.nchunks, W.KS, W.RBa, k == 0, k == K - 1,
                                                         d->return_all_hidden != 0, half, qred),
                                             grid_a, dim3(64 * (NW_A + (half ? 1 : 0))), kp, 0, stream));
                if (k < K - 1) {
                    CellBArgs b = make_b(k);
                    CellBParams kb(b);
                    DRNMF_HIP(h, mark(1));
                    DRNMF_HIP(h, hipLaunchKernel(pick_b_func(W.nch_ks, W.RB, half, qred), grid_b, dim3(64 * NW_B), kb.p, 0,
                                                 stream));
                }
            }
            if (K == 1) {
                DRNMF_HIP(h, mark(2));
                hipLaunchKernelGGL(advance_frame_kernel, dim3(1), dim3(1), 0, stream, tA);
            }
        }
        DRNMF_HIP(h, mark(3));
        DRNMF_HIP(h, hipGetLastError());
        if (profile_frames > 0) {
            DRNMF_HIP(h, hipStreamSynchronize(stream));
            double sum[3] = {0, 0, 0};
            long cnt[3] = {0, 0, 0};
            const size_t per_frame = (ev.size() - 1) / (size_t)T_run;
            const size_t skip = T_run > 1 ? per_frame : 0;          // first frame = warm-up
            for (size_t i = skip; i + 1 < ev.size(); ++i) {
                float ms = 0.f;
                DRNMF_HIP(h, hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
                if (kind[i] < 2) { sum[kind[i]] += ms; cnt[kind[i]]++; }
                sum[2] += ms;
            }
            const int frames_timed = T_run > 1 ? T_run - 1 : 1;
            if (out_us) {
                out_us[0] = cnt[0] ? (float)(sum[0] / cnt[0] * 1e3) : 0.f;
                out_us[1] = cnt[1] ? (float)(sum[1] / cnt[1] * 1e3) : 0.f;
                out_us[2] = (float)(sum[2] / frames_timed * 1e3);
            }
            for (auto e : ev) (void)hipEventDestroy(e);
        }
        return store_final(W);
    }

    auto get_graph = [&](int fpg, hipGraphExec_t* out) -> int32_t {
    std::vector<uint64_t> gkey = key;
    gkey.push_back((uint64_t)fpg);
    GraphEntry* entry = nullptr;
    for (size_t gi = 0; gi < h->graphs.size(); ++gi)
        if (h->graphs[gi].key == gkey) {
            // (a hit moves to the back: eviction is least-recently-used)
            if (gi + 1 != h->graphs.size()) {
                GraphEntry hit = h->graphs[gi];
                h->graphs.erase(h->graphs.begin() + (ptrdiff_t)gi);
                h->graphs.push_back(hit);
            }
            entry = &h->graphs.back();
            break;
        }
    if (!entry) {
        {   // bounded cache: the least recently used entry is retired without synchronising (params.hip)
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
            p.func = func;
            p.gridDim = grid;
            p.blockDim = dim3(block);
            p.sharedMemBytes = 0;
            p.kernelParams = kp;
            p.extra = nullptr;
            hipGraphNode_t node;
            hipError_t e = hipGraphAddKernelNode(&node, ge.graph, last ? &last : nullptr,
                                                 last ? 1 : 0, &p);
            last = node;
            return e;
        };
        for (int rep = 0; rep < fpg; ++rep) {
            for (int k = 0; k < K; ++k) {
                CellAArgs a = make_a(k);
                CellAParams ka_(a);
                void** ka = ka_.p;
                DRNMF_HIP(h, add(ablate ? (void*)&noop_kernel : pick_a_func(a.nchunks, W.KS, W.RBa, k == 0, k == K - 1, d->return_all_hidden != 0, half, qred), grid_a, 64 * (NW_A + (half ? 1 : 0)), ka));
                if (k < K - 1) {
                    CellBArgs b = make_b(k);
                    DRNMF_HIP(h, add(ablate ? (void*)&noop_kernel : pick_b_func(W.nch_ks, W.RB, half, qred), grid_b, 64 * NW_B, CellBParams(b).p));
                }
            }
            if (K == 1) {
                int* tp = tA;
                void* kt[1] = {&tp};
                DRNMF_HIP(h, add((void*)&advance_frame_kernel, dim3(1), 1, kt));
            }
        }
        DRNMF_HIP(h, hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0));
        h->graphs.push_back(ge);
        entry = &h->graphs.back();
    }
    entry->last_stream = stream;
    entry->pin = h->call_seq;
    *out = entry->exec;
    return DRNMF_OK;
    };
    hipGraphExec_t exec_n = nullptr, exec_1 = nullptr;
    int32_t grc = get_graph(fpg_max, &exec_n);
    if (grc) return grc;
    if (plan) {
        plan->exec_n = exec_n;
        plan->fpg = fpg_max;
        plan->n_full = d->T / fpg_max;
        plan->n_rem = d->T % fpg_max;
        if (plan->n_rem) {
            grc = get_graph(1, &exec_1);
            if (grc) return grc;
            plan->exec_1 = exec_1;
        }
        return DRNMF_OK;
    }
    int t = 0;
    for (; t + fpg_max <= d->T; t += fpg_max) DRNMF_HIP(h, hipGraphLaunch(exec_n, stream));
    if (t < d->T) {
        grc = get_graph(1, &exec_1);     // (may evict; exec_n is not used again)
        if (grc) return grc;
        for (; t < d->T; ++t) DRNMF_HIP(h, hipGraphLaunch(exec_1, stream));
    }
    return store_final(W);
}

extern "C" int32_t drnmf_cell_forward(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                      float mask_value, const void* params, const float* log_h0,
                                      float u0_diag, float u0_off, float uk_off, float* h_out,
                                      void* workspace, size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    return cell_forward_impl(h, d, x, mask_value, params, log_h0, u0_diag, u0_off, uk_off, h_out,
                             workspace, workspace_bytes, stream_, 0, nullptr);
}

extern "C" int32_t drnmf_cell_forward_stateful(drnmf_handle_t h, const drnmf_cell_desc_t* d,
                                               const float* x, float mask_value, const void* params,
                                               const float* log_h0, float u0_diag, float u0_off,
                                               float uk_off, const float* initial_state,
                                               float* final_state, float* h_out, void* workspace,
                                               size_t workspace_bytes, void* stream_) {
    DRNMF_LOCK(h);
    return cell_forward_impl(h, d, x, mask_value, params, log_h0, u0_diag, u0_off, uk_off, h_out,
                             workspace, workspace_bytes, stream_, 0, nullptr, initial_state, final_state);
}

extern "C" int32_t drnmf_cell_profile(drnmf_handle_t h, const drnmf_cell_desc_t* d, const float* x,
                                      float mask_value, const void* params, const float* log_h0,
                                      float u0_diag, float u0_off, float uk_off, float* h_out,
                                      void* workspace, size_t workspace_bytes, void* stream_,
                                      int32_t frames, float* out_us_host) {
    DRNMF_LOCK(h);
    if (frames <= 0 || !out_us_host) {
        if (h) snprintf(h->err, sizeof(h->err), "cell_profile: frames must be > 0, out non-NULL");
        return DRNMF_ERR_INVALID_ARG;
    }
    return cell_forward_impl(h, d, x, mask_value, params, log_h0, u0_diag, u0_off, uk_off, h_out,
                             workspace, workspace_bytes, stream_, frames, out_us_host);
}

extern "C" int32_t drnmf_cell_forward_ista(drnmf_handle_t h, const drnmf_cell_desc_t* d,
                                           const float* x, float mask_value, const void* params,
                                           const float* log_h0, float beta,
                                           const float* initial_state, float* final_state,
                                           float* h_out, void* workspace, size_t workspace_bytes,
                                           void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    int rc = validate_cell_desc(h, d);
    if (rc) return rc;
    if (d->divergence != DRNMF_DIV_KL && d->divergence != DRNMF_DIV_BETA)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG,
                   "cell_forward_ista: d->divergence must be DRNMF_DIV_KL or DRNMF_DIV_BETA");
    if (!x || !params || !log_h0 || !h_out || !workspace)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "cell_forward_ista: NULL pointer argument");
    const Workspace W = workspace_layout(d);
    if (workspace_bytes < W.total)
        DRNMF_FAIL(h, DRNMF_ERR_WORKSPACE, "cell_forward_ista: workspace %zu < required %zu",
                   workspace_bytes, W.total);
    if (((uintptr_t)workspace & 255) || ((uintptr_t)params & 255))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "workspace/params must be 256-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    const ParamsLayout L = params_layout(d);
    char* ws = (char*)workspace;
    const char* pb = (const char*)params;
    float* xp = (float*)(ws + W.off_xp);
    unsigned char* valid = (unsigned char*)(ws + W.off_valid);
    float* rpart = (float*)(ws + W.off_rpart);
    float* hb[2] = {(float*)(ws + W.off_h0), (float*)(ws + W.off_h1)};
    float* state = (float*)(ws + W.off_state);
    float* rs_part = (float*)(ws + W.off_rs);
    float* psum = (float*)(ws + W.off_psum);
    int* tA = (int*)(ws + W.off_t);
    int* tB = tA + 16;
    const int K = d->K;

    {   // prologue as the reference cell's (pack + mask, initial state, counters = 0)
        const size_t rows = (size_t)d->T * W.Bp;
        hipLaunchKernelGGL(pack_input_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                           stream, x, xp, valid, mask_value, d->B, d->T, d->F, W.Bp, W.Fp);
        if (initial_state)
            hipLaunchKernelGGL(load_state_kernel, dim3((W.Bp + 3) / 4), dim3(256), 0, stream,
                               initial_state, state, rs_part, tA, d->B, d->N, W.Np, W.Bp, round_up(W.numA, 4), 1);
        else
            hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(256), 0, stream, log_h0, state,
                               rs_part, tA, d->N, W.Np, W.Bp, round_up(W.numA, 4), 1);
        DRNMF_HIP(h, hipMemsetAsync(psum, 0, (size_t)W.Bp * 4, stream));   // read (times 0) by cell_a
        if (d->return_all_hidden) {      // the BPTT reads these (times 0) as the reference cell's does
            DRNMF_HIP(h, hipMemsetAsync(ws + W.off_psum_all, 0, (size_t)d->T * W.Bp * 4, stream));
            hipLaunchKernelGGL(seen_kernel, dim3((W.Bp + 3) / 4), dim3(256), 0, stream, valid,
                               (unsigned char*)(ws + W.off_seen), d->T, W.Bp);
        }
        DRNMF_HIP(h, hipGetLastError());
    }

    const int numM = W.Bp / (ROWS * W.RB), nft = W.nft_main;      // = Fp / 16: no odd-bin side path
    const dim3 grid_a(8u * (unsigned)numM, (unsigned)(round_up(W.numA, 8) / 8));
    const dim3 grid_b(8u * (unsigned)numM, (unsigned)(round_up(nft * W.KS, 8) / 8));
    const dim3 grid_r((unsigned)((size_t)W.Bp * W.Fp / 256));
    const size_t dstride = (size_t)L.Fp * L.Np * 4;
    auto Dn_of = [&](int k) { return pb + L.off_dn + (d->n_D == 1 ? 0 : (size_t)k * dstride); };
    auto DnA_of = [&](int k) { return pb + L.off_dnA + (d->n_D == 1 ? 0 : (size_t)k * dstride); };
    auto tail_of = [&](int k) {
        return (const float*)(pb + L.off_tail) + (d->n_D == 1 ? 0 : (size_t)k * MAX_TAIL * L.Np);
    };
    // layer k of a frame: cell_b (x^ = h_in Dn_k^T) -> resid_div (r = g(x_t, x^)) -> cell_a as a
    // MIDDLE layer of the reference cell (h' = relu(h_in + (r Dn_k) / alpha_k + b_k), uko = 0);
    // h_in = the state for k = 0.  Frame counters: the first resid kernel of a frame reads tB and
    // republishes it as tA, the last cell_a reads tA and publishes tB = t + 1.
    auto h_in_of = [&](int k) { return k == 0 ? state : hb[(k - 1) & 1]; };
    auto make_b = [&](int k) {
        CellBArgs b;
        b.Dn_next = Dn_of(k);
        b.h = h_in_of(k);
        b.xp = nullptr;                 // rpart = + x^
        b.rpart = rpart;
        b.t_rd = tA;
        b.Bp = W.Bp; b.Fp = W.Fp; b.Np = W.Np; b.nft = nft; b.KS = W.KS;
        b.logKS = 0;
        b.nch_ks = W.nch_ks;
        return b;
    };
    auto make_a = [&](int k) {
        CellAArgs a;
        memset(&a, 0, sizeof(a));
        a.Dn = DnA_of(k);
        a.inv_alpha = (const float*)(pb + L.off_inv_alpha) + (size_t)k * L.Np;
        a.bias = (const float*)(pb + L.off_bias) + (size_t)k * L.Np;
        a.rsrc = rpart;
        a.h_in = h_in_of(k);
        a.h_out = (k == K - 1) ? state : hb[k & 1];
        a.state = state;
        a.rs_part = rs_part;
        a.psum = psum;
        a.psum_all = (float*)(ws + W.off_psum_all);
        a.valid = valid;
        a.out = h_out;
        a.t_rd = tA;
        a.t_wr = (k == K - 1) ? tB : nullptr;
        a.t_wr_add = 1;
        a.u0d = 0.f; a.u0o = 0.f; a.uko = 0.f;
        a.B = d->B; a.T = d->T; a.N = d->N; a.Bp = W.Bp; a.Fp = W.Fp; a.Np = W.Np;
        a.numA = W.numA; a.nchunks = nft; a.KS = W.KS; a.ntail = 0; a.tail_tile = nft;
        a.Dtail = tail_of(k);
        a.Dtail_next = tail_of(k);
        float* qp = (float*)(ws + W.off_qpart);
        a.q_in = qp; a.q_out = qp;
        a.xtail = (float*)(ws + W.off_xtail);
        a.xcur = (float*)(ws + W.off_xcur);
        a.out_width = d->return_all_hidden ? d->N * K : d->N;
        a.out_off = d->return_all_hidden ? k * d->N : 0;
        a.write_out = (d->return_all_hidden || k == K - 1) ? 1 : 0;
        return a;
    };
    const int div = d->divergence;
    int F = d->F, Fp = W.Fp, Bp = W.Bp;

    // one frame's launches, through `emit(func, grid, block, kernelParams)`
    auto frame = [&](auto&& emit) -> int32_t {
        for (int k = 0; k < K; ++k) {
            CellBArgs b = make_b(k);
            DRNMF_HIP(h, emit(pick_b_func(W.nch_ks, W.RB, false), grid_b, 64 * NW_B, CellBParams(b).p));
            const float* xpp = xp;
            float* rp = rpart;
            const int* trd = (k == 0) ? tB : tA;
            int* twr = (k == 0) ? tA : nullptr;
            int dv = div;
            float bt = beta;
            float* xsv = W.off_xhat ? (float*)(ws + W.off_xhat) + (size_t)k * Bp * Fp : nullptr;
            size_t xst = (size_t)K * Bp * Fp;
            void* kr[11] = {&xpp, &rp, &trd, &twr, &dv, &bt, &F, &Fp, &Bp, &xsv, &xst};
            DRNMF_HIP(h, emit((void*)&resid_div_kernel, grid_r, 256, kr));
            CellAArgs a = make_a(k);
            CellAParams ka(a);
            // (never the IS_FIRST variant: layer 0 is a full ISTA step from the state)
            DRNMF_HIP(h, emit(pick_a_func(nft, W.KS, W.RB, false, k == K - 1,
                                          d->return_all_hidden != 0, false),
                              grid_a, 64 * NW_A, ka.p));
        }
        return DRNMF_OK;
    };

    const bool use_graph = tune_env("DRNMF_NO_GRAPH") == nullptr;
    if (!use_graph) {
        for (int t = 0; t < d->T; ++t) {
            rc = frame([&](void* f, dim3 g, unsigned blk, void** kp) {
                return hipLaunchKernel(f, g, dim3(blk), kp, 0, stream);
            });
            if (rc) return rc;
        }
        DRNMF_HIP(h, hipGetLastError());
    } else {
        int fpg = 600 / (3 * K);
        fpg = fpg < 1 ? 1 : (fpg > 64 ? 64 : fpg);
        if (fpg > d->T) fpg = d->T;
        uint32_t bbits;
        memcpy(&bbits, &beta, 4);
        auto get_graph = [&](int frames, hipGraphExec_t* out) -> int32_t {
            std::vector<uint64_t> key = {
                0x157AULL, (uint64_t)d->B, (uint64_t)d->T, (uint64_t)d->F, (uint64_t)d->N,
                (uint64_t)d->K, (uint64_t)d->n_D, (uint64_t)d->return_all_hidden,
                (uint64_t)d->divergence, (uint64_t)bbits, (uint64_t)(uintptr_t)params,
                (uint64_t)(uintptr_t)h_out, (uint64_t)(uintptr_t)workspace, (uint64_t)frames};
            for (auto& g : h->graphs)
                if (g.key == key) { *out = g.exec; return DRNMF_OK; }
            if (h->graphs.size() >= 24) {
                DRNMF_HIP(h, hipDeviceSynchronize());
                (void)hipGraphExecDestroy(h->graphs.front().exec);
                (void)hipGraphDestroy(h->graphs.front().graph);
                h->graphs.erase(h->graphs.begin());
            }
            GraphEntry ge;
            ge.key = key;
            DRNMF_HIP(h, hipGraphCreate(&ge.graph, 0));
            hipGraphNode_t last = nullptr;
            auto add = [&](void* f, dim3 g, unsigned blk, void** kp) -> hipError_t {
                hipKernelNodeParams p;
                memset(&p, 0, sizeof(p));
                p.func = f;
                p.gridDim = g;
                p.blockDim = dim3(blk);
                p.kernelParams = kp;
                hipGraphNode_t node;
                hipError_t e = hipGraphAddKernelNode(&node, ge.graph, last ? &last : nullptr,
                                                     last ? 1 : 0, &p);
                last = node;
                return e;
            };
            for (int rep = 0; rep < frames; ++rep) {
                int32_t r2 = frame(add);
                if (r2) return r2;
            }
            DRNMF_HIP(h, hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0));
            h->graphs.push_back(ge);
            *out = ge.exec;
            return DRNMF_OK;
        };
        hipGraphExec_t ex = nullptr;
        rc = get_graph(fpg, &ex);
        if (rc) return rc;
        int t = 0;
        for (; t + fpg <= d->T; t += fpg) DRNMF_HIP(h, hipGraphLaunch(ex, stream));
        if (t < d->T) {
            rc = get_graph(1, &ex);
            if (rc) return rc;
            for (; t < d->T; ++t) DRNMF_HIP(h, hipGraphLaunch(ex, stream));
        }
    }
    if (final_state) {
        const size_t tot = (size_t)d->B * d->N;
        hipLaunchKernelGGL(store_state_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                           stream, state, final_state, d->B, d->N, W.Np);
        DRNMF_HIP(h, hipGetLastError());
    }
    return DRNMF_OK;
}
