// C[M x N] (+)= A^T B with SPLIT OPERANDS on the bf16 matrix pipe (matrix mode DRNMF_MATRIX_BF16X3): the
// time-batched weight gradients and the dictionary-training statistics.  Included at the end of gemm_tn.h: same
// Operands, split / XCD map and epilogue functors; gemm_tn::launch() hands a product over when the mode is on.
// The arithmetic is gemm_nt_x3.h's (three bf16 planes per fp32 value, six v_mfma_f32_32x32x16_bf16 per product,
// fp32 accumulate).  What differs is the staging: both operands are activations (no side is constant, nothing
// to pre-split) and both are contraction-major in memory ([frame][column]), so
//   * a thread stages 16 bytes = 4 columns of one frame, splits them (5.5 VALU operations per element, in the
//     shadow of the second half of the k-tile's MFMAs) and stores 8 bytes per plane at [plane][frame][column]
//     -- the tile lies in LDS as it lies in memory, no register transpose;
//   * the MFMA operand (8 consecutive frames of one column per lane) comes out of LDS through
//     ds_read_b64_tr_b16: 16 lanes read a [4 frames][16 columns] block and receive it transposed
//     (tools/probes/tr16_probe.hip), two reads per operand.
// LDS: [operand][plane][32 frames][128 columns] bf16 = 48 KB, one buffer, two workgroups per CU; the four
// 64-byte column chunks of a frame row are XOR-swizzled with the frame index (mod 4): transposing reads (4 frames
// x 64 bytes per 32 lanes) and staging writes (128 contiguous bytes per 16 lanes) are conflict-free.
#pragma once

#include <utility>

namespace gemm_tn {

using gemm::u32x4;
using gemm::u32x2;
using gemm::split2;
using gemm::mfma_bf16;
using gemm::static_for;
constexpr int X3_PLANE = BK * BM * 2;      // bytes: 32 frames x 128 columns x bf16
constexpr int X3_OPER = 3 * X3_PLANE;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <class Epi>
__global__ void __launch_bounds__(256, 2) gemm_tn_x3_kernel(const Operands g, const Epi epi) {
    __shared__ __attribute__((aligned(16))) float lds[2 * X3_OPER / 4];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int li = l & 31, kk = l >> 5;
    const int wm = w >> 1, wn = w & 1;
    // (split, tile) of this workgroup: gemm_tn.h
    const int tiles_n = (g.N + BN - 1) / BN;
    const int tiles = ((g.M + BM - 1) / BM) * tiles_n;
    const int splits = g.splits, total = tiles * splits;
    const int per_xcd = (total + 7) / 8;
    const int lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || lin >= total) return;
    const int split = lin / tiles, tile = lin % tiles;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int64_t nkt = (g.Kdim + BK - 1) / BK;
    const int64_t per = (nkt + splits - 1) / splits;
    const int64_t kt0 = split * per;
    int64_t kt1 = kt0 + per;
    if (kt1 > nkt) kt1 = nkt;

    const int kr = tid >> 5, c4 = (tid & 31) * 4;   // staging: frame kr + 8 i, 4 columns at c4

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

    f32x4 ra[4], rb[4];
    auto raw4 = [&](const float* base, int64_t k, int64_t ld, int c, int C) {
        const int64_t kr2 = k < g.Kdim ? k : g.Kdim - 1;
        const int cc = c < C ? c : 0;
        return *(const f32x4*)(base + kr2 * ld + cc);
    };
    auto gload = [&](int64_t kt) __attribute__((always_inline)) {
        const int64_t k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = raw4(g.A, k0 + kr + 8 * i, g.lda, m0 + c4, g.M);
            rb[i] = raw4(g.B, k0 + kr + 8 * i, g.ldb, n0 + c4, g.N);
        }
    };
    // piece p = 0..15 of the staged tile's split: two columns of one staged 16-byte group (operand p & 1,
    // staging slice (p >> 1) & 3, half p >> 3), in the order the loads were issued
    u32x2 pa[4][3], pb[4][3];
    const bool aok = m0 + c4 < g.M, bok = n0 + c4 < g.N;
    auto split_piece = [&](auto p_tag, int64_t kt) __attribute__((always_inline)) {
        constexpr int p = decltype(p_tag)::value;
        constexpr int i = (p >> 1) & 3, j = p >> 3;
        const bool kok = kt * BK + kr + 8 * i < g.Kdim;
        unsigned q0, q1, q2;
        if ((p & 1) == 0) {
            const bool ok = kok && aok;
            split2(ok ? ra[i][2 * j] : 0.f, ok ? ra[i][2 * j + 1] : 0.f, q0, q1, q2);
            pa[i][0][j] = q0; pa[i][1][j] = q1; pa[i][2][j] = q2;
        } else {
            const bool ok = kok && bok;
            split2(ok ? rb[i][2 * j] : 0.f, ok ? rb[i][2 * j + 1] : 0.f, q0, q1, q2);
            pb[i][0][j] = q0; pb[i][1][j] = q1; pb[i][2][j] = q2;
        }
    };
    // (frame kr + 8 i: its index mod 4 is kr & 3)
    const int s_dst = kr * 256 + ((((c4 * 2) >> 6) ^ (kr & 3)) << 6) + ((c4 * 2) & 63);
    auto store_staged = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                *(u32x2*)((char*)lds + s_dst + 8 * i * 256 + p * X3_PLANE) = pa[i][p];
                *(u32x2*)((char*)lds + X3_OPER + s_dst + 8 * i * 256 + p * X3_PLANE) = pb[i][p];
            }
    };
    // transposing fragment read (header): this lane's address inside a plane for column block c (32 columns
    // = one 64-byte chunk), frames fb .. fb + 3 (fb a multiple of 4)
    const int tr_row = ((l >> 2) & 3) * 256 + 32 * ((l >> 4) & 1) + 8 * (l & 3);
    const int tr_sw = (l >> 2) & 3;
    auto tr8 = [&](const char* plane, int chunk, int fb) __attribute__((always_inline)) {
        const char* p = plane + fb * 256 + tr_row + ((chunk ^ tr_sw) << 6);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 256));
        u32x4 r;
        const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
        r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
        return r;
    };
    constexpr int PA[6] = {0, 0, 1, 0, 2, 1}, PB[6] = {0, 1, 0, 2, 0, 1};   // decreasing magnitude
    // NA (0, 1, 2): 32-row halves of this wave's 64 output rows inside M (gemm_tn.h)
    auto ktile = [&](auto stage_tag, int64_t kt_next, auto na_tag) __attribute__((always_inline)) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        constexpr int NA = decltype(na_tag)::value;
        u32x4 fa[2][2][3], fb[2][2][3];          // [step][row / column block][plane]
        auto fetch = [&](auto st_tag) __attribute__((always_inline)) {
            constexpr int st = decltype(st_tag)::value;
            if (NA == 0) return;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                fa[st][0][p] = tr8((const char*)lds + p * X3_PLANE, 2 * wm, 16 * st + 8 * kk);
                fb[st][0][p] = tr8((const char*)lds + X3_OPER + p * X3_PLANE, 2 * wn, 16 * st + 8 * kk);
                if (NA == 2) fa[st][1][p] = tr8((const char*)lds + p * X3_PLANE, 2 * wm + 1, 16 * st + 8 * kk);
                fb[st][1][p] = tr8((const char*)lds + X3_OPER + p * X3_PLANE, 2 * wn + 1, 16 * st + 8 * kk);
            }
        };
        auto product = [&](auto st_tag, auto qt_tag) __attribute__((always_inline)) {
            constexpr int st = decltype(st_tag)::value, q = decltype(qt_tag)::value >> 2, t = decltype(qt_tag)::value & 3;
            if (NA == 2 || (NA == 1 && (t >> 1) == 0))
                acc[t >> 1][t & 1] = mfma_bf16(fa[st][t >> 1][PA[q]], fb[st][t & 1][PB[q]], acc[t >> 1][t & 1]);
        };
        __builtin_amdgcn_s_setprio(3);        // (gemm_nt_x3.h)
        fetch(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        if (STAGE) gload(kt_next);
        fetch(std::integral_constant<int, 1>{});
        static_for<24>([&](auto qt) __attribute__((always_inline)) { product(std::integral_constant<int, 0>{}, qt); });
        if (NA == 2) {       // step 0: the 8 global loads of the next tile and the 24 reads of step 1 between its MFMAs
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (STAGE && i < 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // step 1: one piece of the split behind each of its first 16 MFMAs, fenced (gemm_nt_x3.h)
        static_for<24>([&](auto qt) __attribute__((always_inline)) {
            product(std::integral_constant<int, 1>{}, qt);
            if constexpr (STAGE && decltype(qt)::value < 16) {
                split_piece(qt, kt_next);
                if (NA == 2) __builtin_amdgcn_sched_barrier(0);
            }
        });
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    const int na = m0 + wm * 64 + 32 < g.M ? 2 : (m0 + wm * 64 < g.M ? 1 : 0);   // (wave-uniform)
    if (kt0 < kt1) {
        gload(kt0);
        static_for<16>([&](auto p) __attribute__((always_inline)) { split_piece(p, kt0); });
        store_staged();
        __syncthreads();
        auto body = [&](auto na_tag) __attribute__((always_inline)) {
            for (int64_t kt = kt0; kt + 1 < kt1; ++kt) {
                ktile(std::true_type{}, kt + 1, na_tag);
                __syncthreads();
                store_staged();
                __syncthreads();
            }
            ktile(std::false_type{}, 0, na_tag);
        };
        if (na == 2) body(std::integral_constant<int, 2>{});
        else if (na == 1) body(std::integral_constant<int, 1>{});
        else body(std::integral_constant<int, 0>{});
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn * 64 + b * 32 + li;
            if (n >= g.N) continue;
            float pv[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                int m = m0 + wm * 64 + a * 32 + (v & 3) + 8 * (v >> 2) + 4 * kk;
                m = m < g.M ? m : g.M - 1;
                pv[v] = epi.pre(split, m, n);
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * 64 + a * 32 + (v & 3) + 8 * (v >> 2) + 4 * kk;
                if (m < g.M) epi(split, m, n, acc[a][b][v], pv[v]);
            }
        }
}

template <class Epi>
inline hipError_t launch_x3(const Operands& gg, const Epi& epi, dim3 grid, hipStream_t stream) {
    hipLaunchKernelGGL((gemm_tn_x3_kernel<Epi>), grid, dim3(256), 0, stream, gg, epi);
    return hipGetLastError();
}

}  // namespace gemm_tn
