// C[M x N] = epilogue(A . Bt^T) with SPLIT OPERANDS on the bf16 matrix pipe (matrix mode DRNMF_MATRIX_BF16X3,
// include/drnmf.h).  Included at the end of gemm_nt.h: same Operands, same epilogue functors, same THIN /
// ktail / NB / REDUCE conventions; gemm::launch() hands a product over to launch_x3() when the mode is on.
//
// The fp32 matrix pipe of gfx950 runs at 1/16 of the bf16 rate.  An fp32 value is EXACTLY the sum of three
// bf16 planes (hi = rne(x), mid = rne(x - hi), lo = x - hi - mid: 3 x 8 significand bits + the signs), so
//   a . b = ah bh + (ah bm + am bh) + (ah bl + al bh + am bm) + O(2^-26 |a b|)
// -- six v_mfma_f32_32x32x16_bf16 (32 cycles per SIMD each, fp32 accumulate) replace eight
// v_mfma_f32_32x32x2_f32 (64 cycles each) per 32 x 32 x 16 block: 2.67x the matrix rate, at an error per
// product below the rounding of the fp32 accumulation both forms share.
//
//   Bt (the dictionary side: the SAME operand for every row tile, and constant over the iterations of the
//      callers' loops) is split ONCE per product by a pre-pass into handle-owned scratch,
//        B3[row tile][k-tile][128 rows][plane][4 pieces of 8 slots] bf16 -- the 24 KB LDS image of the
//        (row tile, k-tile) operand tile, swizzle included; rows padded to 128 and slots to 32 with zeros,
//      and staged global -> registers -> LDS as a LINEAR copy of 16-byte pieces: no VALU work, no bounds
//      selects, no address tables;
//   A  (frames x contraction: activations) is split on its way from the staging registers to LDS
//      (v_cvt_pk_bf16_f32, shift / and, subtract: 5.5 VALU operations per element) in the shadow of the
//      second half of the k-tile's MFMAs;
//   LDS, per operand tile row: [hi | mid | lo] x 64 bytes = 192 bytes, no padding; the four 16-byte pieces
//      of a plane are XOR-swizzled with bits 2..3 of the row, which makes the fragment reads (ds_read_b128,
//      32 rows x one piece) and both kinds of staging writes conflict-free.  48 KB per k-tile, ONE buffer,
//      two workgroups per CU: one's store phase between its two barriers runs under the other's MFMAs.
//
// History of the numbers (frame-parallel ISTA, 32768 x 513 x 2000, TFLOP/s fp32-equivalent; fp32 pipe 117-120):
// both operands split in the kernel behind the MFMAs 147, the same split interleaved with the MFMAs 149 (PMC:
// matrix pipes 55 % busy at 1.73 GHz -- the chip clocks down under the VALU + MFMA load -- and a third of
// the LDS cycles bank conflicts of the padded 208-byte rows), XCD-aware tile order +-0, Bt pre-split + swizzled
// unpadded rows 180, priority in the MFMA phase 186 (pipes 64 % busy at 1.76 GHz): profiles/r06_x3_steps.txt,
// profiles/r06_x3_pmc.txt.
#pragma once

#include <utility>

namespace gemm {

using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
constexpr int X3_ROW = 192;                       // bytes per tile row: 3 planes x 32 slots x 2
constexpr int X3_OPER = BM * X3_ROW;              // bytes per operand tile

// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}): indices that MUST be
// compile-time constants (register arrays indexed through a lambda parameter end up in scratch)
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {   // low half rne(a), high half rne(b)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float lo_f32(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float hi_f32(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }
// two consecutive contraction slots -> one dword per plane.  Round-to-nearest planes (v_cvt_pk_bf16_f32): the
// residuals carry random signs.  Planes by truncation (and / subtract, the same instruction count) leave every
// dropped term with the sign of the product: 2.2e-6 instead of 3.8e-7 rms on H W^T (tools/probes/x3_nt_probe.hip).
__device__ __forceinline__ void split2(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned h = pk_bf16(x0, x1);
    const float r0 = x0 - lo_f32(h), r1 = x1 - hi_f32(h);
    const unsigned m = pk_bf16(r0, r1);
    const float s0 = r0 - lo_f32(m), s1 = r1 - hi_f32(m);
    p0 = h;
    p1 = m;
    p2 = pk_bf16(s0, s1);
}
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                   c, 0, 0, 0);
}

// Pre-pass: Bt [N][ldb] fp32 -> B3 (layout above).  One thread per 8 slots.
static __global__ void __launch_bounds__(256)
x3_split_rows_kernel(const float* __restrict__ Bt, int64_t ldb, int N, int K, int KT, int64_t total,
                     u32x4* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int piece = (int)(idx & 3);
    const int64_t rk = idx >> 2;
    const int kt = (int)(rk % KT);
    const int64_t row = rk / KT;
    const int k = kt * 32 + piece * 8;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (row < N) {                               // (K % 4 == 0: a 4-slot group is inside or outside)
        if (k < K) v0 = *(const f32x4*)(Bt + row * ldb + k);
        if (k + 4 < K) v1 = *(const f32x4*)(Bt + row * ldb + k + 4);
    }
    u32x4 p0, p1, p2;
    unsigned a, b, c;
    split2(v0[0], v0[1], a, b, c); p0[0] = a; p1[0] = b; p2[0] = c;
    split2(v0[2], v0[3], a, b, c); p0[1] = a; p1[1] = b; p2[1] = c;
    split2(v1[0], v1[1], a, b, c); p0[2] = a; p1[2] = b; p2[2] = c;
    split2(v1[2], v1[3], a, b, c); p0[3] = a; p1[3] = b; p2[3] = c;
    const int r = (int)(row & (BN - 1));
    u32x4* o = out + ((row / BN) * KT + kt) * (X3_OPER / 16) + r * 12 + (piece ^ ((r >> 2) & 3));
    o[0] = p0;
    o[4] = p1;
    o[8] = p2;
}

// -DX3_TIMELINE (tools/probes/x3_nt_probe.hip only): s_memtime phase sums of wave 0 of workgroups 0..63
#ifdef X3_TIMELINE
__device__ unsigned long long g_x3_timeline[64][8];
#define X3_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define X3_ADD(slot, a, b) tl_sum[slot] += (b) - (a)
#else
#define X3_T(var) do { } while (0)
#define X3_ADD(slot, a, b) do { } while (0)
#endif

template <class Epi, bool THIN>
__global__ void __launch_bounds__(256, 2) gemm_nt_x3_kernel(const Operands g, const Epi epi) {
    __shared__ __attribute__((aligned(16))) float lds[2 * X3_OPER / 4];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int li = l & 31, kk = l >> 5;
    const int wm = w >> 1, wn = w & 1;

    // Workgroup -> tile, XCD-aware (g.per_xcd > 0): consecutive workgroup ids go round the 8 XCDs, each with
    // its own L2, and the tiles_n column tiles of a row tile read the same 128 rows of A -- XCD x takes the
    // contiguous range [x * per_xcd, (x + 1) * per_xcd) of the row-major tile order, so a row tile's A operand
    // crosses the fabric once instead of once per XCD that holds one of its column tiles.
    const int tiles_n = (g.N + BN - 1) / BN;
    int lin = (int)blockIdx.x;
    if (g.per_xcd) {
        lin = (int)(blockIdx.x & 7) * g.per_xcd + (int)(blockIdx.x >> 3);
        if ((int64_t)lin >= ((g.M + BM - 1) / BM) * tiles_n) return;
    }
    const int64_t tm = lin / tiles_n;
    const int tn = lin % tiles_n;
    const int64_t m0 = tm * BM;
    const int n0 = tn * BN;

    // staging map of A: thread -> (row = tid/8 + 32 i, 4 floats at k = (tid%8) * 4)
    const int srow = tid >> 3, sk = (tid & 7) * 4;
    // ... of B3: a linear copy, thread -> 16-byte pieces tid + 256 i (i < 6) of the tile's 1536
    // (by LDS-DMA into a second B buffer instead -- global_load_lds_dwordx4, no staging registers, no
    // ds_write_b128: measured equal, 163.6 / 144.9 against 169.9 / 140.2 TFLOP/s-eq for the two ISTA products,
    // profiles/r06_x3_steps.txt; the compiler makes every LDS read behind an LDS-DMA wait for it, which pins the
    // DMA behind the tile's last fragment read)
    const u32x4* B3 = (const u32x4*)g.B3 + (int64_t)tn * g.kt3 * (X3_OPER / 16) + tid;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

    f32x4 ra[4], rt = {0.f, 0.f, 0.f, 0.f};
    u32x4 rb[6];
    float tacc[4] = {0.f, 0.f, 0.f, 0.f};
    bool mine[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mine[i] = THIN && (tiles_n >= 4 ? i == tn : i % tiles_n == tn);
    auto raw4 = [&](const float* base, int64_t row, int64_t nrows, int64_t ld, int k) {
        const int64_t rr = row < nrows ? row : nrows - 1;
        const int kc = k < g.K ? k : 0;
        return *(const f32x4*)(base + rr * ld + kc);
    };
    // A is fetched TWO tiles ahead (gload_a(kt + 2) right behind the split of tile kt + 1, into the registers
    // that split just freed): a k-tile of this kernel is ~0.5 us of MFMAs, less than an HBM round trip under
    // load, and with one tile of lookahead the split waited for its operand whenever A did not sit in the
    // Infinity Cache (M = 16384: 185, M = 32768: 160 TFLOP/s-eq).  B3 (cache-resident) one tile ahead.
    auto gload_a = [&](int kt) __attribute__((always_inline)) {
        if (THIN) rt = raw4(g.Bt, g.N, g.N + 1, g.ldb, kt * BK + sk);
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = raw4(g.A, m0 + srow + 32 * i, g.M, g.lda, kt * BK + sk);
    };
    auto gload_b = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) rb[i] = B3[kt * (X3_OPER / 16) + 256 * i];
    };
    // piece m = 0..7 of the staged A tile's split: two consecutive slots of staging slice m & 3 (half m >> 2):
    // 9 VALU operations (+ 2 selects unless FULL: every row of the tile inside M and the k-tile inside K)
    u32x2 pa[4][3];
    auto split_piece = [&](auto m_tag, auto full_tag, int k0) __attribute__((always_inline)) {
        constexpr int m = decltype(m_tag)::value;
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int i = m & 3, j = m >> 2;
        const bool kok = k0 + sk < g.K;
        const bool ok = FULL || (kok && m0 + srow + 32 * i < g.M);
        const float x0 = ok ? ra[i][2 * j] : 0.f, x1 = ok ? ra[i][2 * j + 1] : 0.f;
        unsigned q0, q1, q2;
        split2(x0, x1, q0, q1, q2);
        pa[i][0][j] = q0; pa[i][1][j] = q1; pa[i][2][j] = q2;
        if (THIN) {
            const float t0 = (FULL || kok) ? rt[2 * j] : 0.f, t1 = (FULL || kok) ? rt[2 * j + 1] : 0.f;
            tacc[i] += x0 * t0 + x1 * t1;
        }
    };
    const int a_dst = srow * X3_ROW + ((((tid & 7) >> 1) ^ ((srow >> 2) & 3)) << 4) + (tid & 1) * 8;
    auto store_staged = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) *(u32x2*)((char*)lds + a_dst + 32 * i * X3_ROW + 64 * p) = pa[i][p];
#pragma unroll
        for (int i = 0; i < 6; ++i) *(u32x4*)((char*)lds + X3_OPER + 16 * tid + 4096 * i) = rb[i];
    };

    // One k-tile: two steps of 16 contraction slots; per step 12 fragments (ds_read_b128: 2 row blocks x 3
    // planes of A, the same of B) and 6 products x 4 output tiles = 24 MFMAs.  Lane group kk of step st takes
    // piece 2 st + kk of every plane (any slot order does: A and B share it).
    const int fsw = (li >> 2) & 3;
    const int foff0 = ((kk ^ fsw) << 4), foff1 = (((2 + kk) ^ fsw) << 4);
    const char* Ar = (const char*)lds + (wm * 64 + li) * X3_ROW;
    const char* Br = (const char*)lds + X3_OPER + (wn * 64 + li) * X3_ROW;
    constexpr int PA[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};   // decreasing magnitude
    auto ktile = [&](auto stage_tag, auto full_tag, int kt_next, auto nb_tag) __attribute__((always_inline)) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        constexpr int NB = decltype(nb_tag)::value;
        u32x4 fa[2][2][3], fb[2][2][3];          // [step][row / column block][plane]
        auto fetch = [&](auto st_tag) __attribute__((always_inline)) {
            constexpr int st = decltype(st_tag)::value;
            const int fo = st ? foff1 : foff0;
            if (NB == 0) return;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                fa[st][0][p] = *(const u32x4*)(Ar + 64 * p + fo);
                fb[st][0][p] = *(const u32x4*)(Br + 64 * p + fo);
                fa[st][1][p] = *(const u32x4*)(Ar + 32 * X3_ROW + 64 * p + fo);
                if (NB == 2) fb[st][1][p] = *(const u32x4*)(Br + 32 * X3_ROW + 64 * p + fo);
            }
        };
        auto product = [&](auto st_tag, auto qt_tag) __attribute__((always_inline)) {
            constexpr int st = decltype(st_tag)::value, q = decltype(qt_tag)::value >> 2, t = decltype(qt_tag)::value & 3;
            if (NB == 2 || (NB == 1 && (t & 1) == 0))
                acc[t >> 1][t & 1] = mfma_bf16(fa[st][t >> 1][PA[q]], fb[st][t & 1][PB[q]], acc[t >> 1][t & 1]);
        };
        // (priority over the other workgroup's waves while this one feeds the matrix pipe: they are in their
        // store phase or would only interleave with it -- +2 .. 6 %)
        __builtin_amdgcn_s_setprio(3);
        fetch(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        if (STAGE) gload_b(kt_next);
        fetch(std::integral_constant<int, 1>{});
        // step 0: the 6 loads of the next B3 tile and the 12 fragment reads of step 1 between its MFMAs
        static_for<24>([&](auto qt) __attribute__((always_inline)) { product(std::integral_constant<int, 0>{}, qt); });
        if (NB == 2) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (STAGE && i < 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // step 1: one piece of A's split behind each of its first 8 MFMAs, fenced (sched_group_barrier left
        // the VALU operations in one lump behind the MFMAs); then the loads of the A tile after next
        static_for<24>([&](auto qt) __attribute__((always_inline)) {
            product(std::integral_constant<int, 1>{}, qt);
            if constexpr (STAGE && decltype(qt)::value < 8) {
                split_piece(qt, full_tag, kt_next * BK);
                if (NB == 2) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (STAGE && decltype(qt)::value == 8) {
                gload_a(kt_next + 1);         // (clamped addresses: a tile past the end is loaded and never used)
                if (NB == 2) __builtin_amdgcn_sched_barrier(0);
            }
        });
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nkt = (g.K + BK - 1) / BK;
    // (wave-uniform: tn, wn and N are)
    const int nb = n0 + wn * 64 + 32 < g.N ? 2 : (n0 + wn * 64 < g.N ? 1 : 0);
#ifdef X3_TIMELINE
    unsigned long long tl_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    X3_T(t_start);
    gload_a(0);
    gload_b(0);
    static_for<8>([&](auto m) __attribute__((always_inline)) { split_piece(m, std::false_type{}, 0); });
    gload_a(1);
    store_staged();
    __syncthreads();
    {
        // tile kt in LDS: fetch tile kt+1 into the staging registers, contract (splitting the staged A tile
        // on the way), and once every wave is done reading store the staged tile in its place
        auto body = [&](auto nb_tag) __attribute__((always_inline)) {
            int kt = 0;
            if (m0 + BM <= g.M)        // every k-tile but the last is then whole: no selects
                for (; (kt + 2) * BK <= g.K && kt + 1 < nkt; ++kt) {
                    X3_T(t0);
                    ktile(std::true_type{}, std::true_type{}, kt + 1, nb_tag);
                    X3_T(t1);
                    __syncthreads();
                    X3_T(t2);
                    store_staged();
#ifdef X3_TIMELINE
                    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
#endif
                    X3_T(t3);
                    __syncthreads();
                    X3_T(t4);
                    X3_ADD(0, t0, t1); X3_ADD(1, t1, t2); X3_ADD(2, t2, t3); X3_ADD(3, t3, t4); X3_ADD(7, 0ull, 1ull);
                }
            for (; kt + 1 < nkt; ++kt) {
                ktile(std::true_type{}, std::false_type{}, kt + 1, nb_tag);
                __syncthreads();
                store_staged();
                __syncthreads();
            }
        };
        if (nb == 2) body(std::integral_constant<int, 2>{});
        else if (nb == 1) body(std::integral_constant<int, 1>{});
        else body(std::integral_constant<int, 0>{});
    }

    X3_T(t_loop_end);
    // Last k-tile and epilogue (gemm_nt.h).  Register v of lane l holds row (v&3) + 8*(v>>2) + 4*(l>>5),
    // column l&31.
    constexpr bool EARLY = epi_early<Epi>::value;
    constexpr bool RED = epi_reduce<Epi>::value;
    float red = 0.f;
    const int M32 = (int)g.M, m032 = (int)m0;
    auto rowof = [&](int a, int v) { return m032 + wm * 64 + a * 32 + (v & 3) + 8 * (v >> 2) + 4 * kk; };
    auto colof = [&](int b) { return n0 + wn * 64 + b * 32 + li; };
    auto finish = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;   // every row of the tile is inside M
        f32x2 pv[EARLY ? 2 : 1][EARLY ? 2 : 1][16];
        float ta[2] = {0.f, 0.f}, tb[2] = {0.f, 0.f};
        const bool tk = kk < g.ktail;
        if (g.ktail) {   // raw loads (clamped addresses); zeroed after the MFMAs
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t r = m0 + wm * 64 + i * 32 + li;
                const int c = n0 + wn * 64 + i * 32 + li;
                ta[i] = g.A[(r < g.M ? r : g.M - 1) * g.lda + g.K + (tk ? kk : 0)];
                tb[i] = g.Bt[(int64_t)(c < g.N ? c : g.N - 1) * g.ldb + g.K + (tk ? kk : 0)];
            }
        }
        if (EARLY) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    int col = colof(b);
                    col = col < g.N ? col : g.N - 1;
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        int row = rowof(a, v);
                        if (!FULL) row = row < M32 ? row : M32 - 1;
                        pv[a][b][v] = epi.pre(row, col);
                    }
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        // (the last tile under the epilogue's early loads -- 128 registers of `pre` values beside the
        // accumulators: fragments are fetched per product, 16 registers at a time, instead of per step)
        if (nb > 0) {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const int fo = st ? foff1 : foff0;
                    const u32x4 a0 = *(const u32x4*)(Ar + 64 * PA[q] + fo);
                    const u32x4 a1 = *(const u32x4*)(Ar + 32 * X3_ROW + 64 * PA[q] + fo);
                    const u32x4 b0 = *(const u32x4*)(Br + 64 * PB[q] + fo);
                    acc[0][0] = mfma_bf16(a0, b0, acc[0][0]);
                    acc[1][0] = mfma_bf16(a1, b0, acc[1][0]);
                    if (nb == 2) {
                        const u32x4 b1 = *(const u32x4*)(Br + 32 * X3_ROW + 64 * PB[q] + fo);
                        acc[0][1] = mfma_bf16(a0, b1, acc[0][1]);
                        acc[1][1] = mfma_bf16(a1, b1, acc[1][1]);
                    }
                }
        }
        if (g.ktail && nb > 0) {      // the odd contraction columns in exact fp32 (gemm_nt.h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!(tk && m0 + wm * 64 + i * 32 + li < g.M)) ta[i] = 0.f;
                if (!(tk && n0 + wn * 64 + i * 32 + li < g.N)) tb[i] = 0.f;
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[0], tb[0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[0], tb[1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[1], tb[0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[1], tb[1], acc[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int col = colof(b);
                if (col >= g.N) continue;
                if (!EARLY) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        int row = rowof(a, v);
                        if (!FULL) row = row < M32 ? row : M32 - 1;
                        pv[0][0][v] = epi.pre(row, col);
                    }
                }
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    int row = rowof(a, v);
                    if (EARLY) asm volatile("" : "+v"(row));
                    if (FULL || row < M32) {
                        if constexpr (RED) red += epi(row, col, acc[a][b][v], pv[EARLY ? a : 0][EARLY ? b : 0][v]);
                        else epi(row, col, acc[a][b][v], pv[EARLY ? a : 0][EARLY ? b : 0][v]);
                    }
                }
            }
    };
    if (m032 + BM <= M32) finish(std::true_type{});
    else finish(std::false_type{});
    if (THIN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!mine[i]) continue;
            float t = tacc[i];
            t += __shfl_xor(t, 1, 64);          // the eight k-lanes (tid & 7) of staging row tid >> 3
            t += __shfl_xor(t, 2, 64);
            t += __shfl_xor(t, 4, 64);
            const int row = m032 + srow + 32 * i;
            if ((tid & 7) == 0 && row < M32) {
                const f32x2 pv = epi.pre(row, g.N);
                if constexpr (RED) red += epi(row, g.N, t, pv);
                else epi(row, g.N, t, pv);
            }
        }
    }
#ifdef X3_TIMELINE
    if (tid == 0 && blockIdx.x < 64) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        unsigned long long loop = 0;
        for (int k = 0; k < 4; ++k) loop += tl_sum[k];
        tl_sum[4] = (t_loop_end - t_start) - loop;
        tl_sum[5] = t_end - t_loop_end;
        for (int k = 0; k < 8; ++k) g_x3_timeline[blockIdx.x][k] = tl_sum[k];
    }
#endif
    if constexpr (RED) {
        __syncthreads();                       // every wave is done with the staged tiles
        lds[tid] = red;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) lds[tid] += lds[tid + o];
            __syncthreads();
        }
        if (tid == 0) epi.red_out[lin] = lds[0];
    }
}

// g: as gemm::launch() prepared it (thin column already taken off N).  *taken = false: not run (no scratch)
template <class Epi>
inline hipError_t launch_x3(const Operands& g_in, bool thin, const Epi& epi, hipStream_t stream, bool* taken) {
    Operands g = g_in;
    *taken = false;
    const int KT = (g.K + BK - 1) / BK;
    const int64_t Npad = ((int64_t)g.N + BN - 1) / BN * BN;
    void* scratch = x3_scratch_get(stream, (size_t)Npad * KT * X3_ROW);
    if (!scratch) return hipSuccess;
    const int64_t total = Npad * KT * 4;
    hipLaunchKernelGGL(x3_split_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, g.Bt, g.ldb,
                       g.N, g.K, KT, total, (u32x4*)scratch);
    g.B3 = scratch;
    g.kt3 = KT;
    int64_t tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    const char* xe = measure_env("DRNMF_NT_XCD");     // (0: identity order)
    if (tiles >= 64 && !(xe && atoi(xe) == 0)) {          // (a handful of tiles: nothing to share)
        g.per_xcd = (int)((tiles + 7) / 8);
        tiles = (int64_t)g.per_xcd * 8;
    }
    if (thin)
        hipLaunchKernelGGL((gemm_nt_x3_kernel<Epi, true>), dim3((unsigned)tiles), dim3(256), 0, stream, g, epi);
    else
        hipLaunchKernelGGL((gemm_nt_x3_kernel<Epi, false>), dim3((unsigned)tiles), dim3(256), 0, stream, g, epi);
    *taken = true;
    return hipGetLastError();
}

}  // namespace gemm
