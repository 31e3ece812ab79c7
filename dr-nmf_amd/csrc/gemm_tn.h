// C[M x N] (+)= A^T B with A [Kdim][M], B [Kdim][N] both row-major: the contraction runs over the
// ROW index (frames).  This is the shape of every time-batched weight gradient of the cell
// (d Dn_k = R_k^T dG_k - dR_k^T H_{k-1}, contraction over B*T rows): large Kdim, small output, so
// the contraction is additionally split over gridDim.y and each split hands its partial tile to
// the epilogue functor (partial buffers are summed afterwards in a fixed order: deterministic).
//
//   block tile 128 x 128, BK = 32 rows, 256 threads = 2 x 2 waves, v_mfma_f32_32x32x2_f32
//   both tiles land in LDS exactly as they sit in memory ([k][m], [k][n]: coalesced 512-byte rows);
//   MFMA fragments are single-dword reads of 32 consecutive columns (conflict-free).
#pragma once
#include "common.h"
#include "gemm_nt.h"      // (gemm_nt_x3.h: the split-operand helpers gemm_tn_x3.h shares)

#include <type_traits>

namespace gemm_tn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int BM = 128, BN = 128, BK = 32;

struct Operands {
    const float* A;   // [Kdim][lda], columns = M
    const float* B;   // [Kdim][ldb], columns = N
    int64_t Kdim;
    int M, N;
    int64_t lda, ldb;
    int splits = 1;   // set by launch()
};

// VEC: lda, ldb, M, N multiples of 4 and 16-byte aligned bases -> branch-free float4 loads
template <bool VEC>
__device__ __forceinline__ f32x4 load4(const float* base, int64_t k, int64_t Kend, int64_t ld,
                                       int c, int C) {
    if (VEC) {
        const bool ok = k < Kend && c < C;
        const int64_t kr = k < Kend ? k : Kend - 1;
        const int cc = c < C ? c : 0;
        f32x4 v = *(const f32x4*)(base + kr * ld + cc);
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
        return v;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k < Kend) {
        const float* p = base + k * ld + c;
        if (c + 0 < C) v[0] = p[0];
        if (c + 1 < C) v[1] = p[1];
        if (c + 2 < C) v[2] = p[2];
        if (c + 3 < C) v[3] = p[3];
    }
    return v;
}

// Epi: float pre(int split, int m, int n) const;  void operator()(int split, int m, int n, float acc, float pre) const
template <class Epi, bool VEC>
__global__ void __launch_bounds__(256) gemm_tn_kernel(const Operands g, const Epi epi) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BK * BM];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int li = l & 31, kk = l >> 5;
    const int wm = w >> 1, wn = w & 1;
    // Workgroup -> (split, tile), XCD-aware: consecutive workgroup ids go round the 8 XCDs, and the
    // tiles of one split read the same rows of A and B, so XCD x takes the contiguous range
    // [x * per, (x+1) * per) of the split-major order: a split's operands then live in ONE L2
    // instead of being fetched into all eight (2.7 GB -> 0.8 GB per C2 weight-gradient GEMM).
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

    const int kr = tid >> 5, c4 = (tid & 31) * 4;   // staging: row kr + 8*i, 4 columns at c4

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

    // (vector path: the out-of-range zeroing is applied when the staged tile is written to LDS,
    // not right behind the load, so that the loads stay in flight under the MFMAs -- gemm_nt.h)
    f32x4 ra[4], rb[4];
    auto raw4 = [&](const float* base, int64_t k, int64_t ld, int c, int C) {
        const int64_t kr2 = k < g.Kdim ? k : g.Kdim - 1;
        const int cc = c < C ? c : 0;
        return *(const f32x4*)(base + kr2 * ld + cc);
    };
    auto gload = [&](int64_t kt) {
        const int64_t k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (VEC) {
                ra[i] = raw4(g.A, k0 + kr + 8 * i, g.lda, m0 + c4, g.M);
                rb[i] = raw4(g.B, k0 + kr + 8 * i, g.ldb, n0 + c4, g.N);
            } else {
                ra[i] = load4<false>(g.A, k0 + kr + 8 * i, g.Kdim, g.lda, m0 + c4, g.M);
                rb[i] = load4<false>(g.B, k0 + kr + 8 * i, g.Kdim, g.ldb, n0 + c4, g.N);
            }
        }
    };
    auto swrite = [&](int buf, int64_t kt) {
        float* As = lds + buf * 2 * BK * BM;
        float* Bs = As + BK * BM;
        const int64_t k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 va = ra[i], vb = rb[i];
            if (VEC) {
                const bool kok = k0 + kr + 8 * i < g.Kdim;
                if (!(kok && m0 + c4 < g.M)) va = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!(kok && n0 + c4 < g.N)) vb = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            *(f32x4*)(As + (kr + 8 * i) * BM + c4) = va;
            *(f32x4*)(Bs + (kr + 8 * i) * BN + c4) = vb;
        }
    };

    // One k-tile = 16 steps of 4 MFMAs.  Everything else is pinned between them (gemm_nt.h has the
    // measurements behind this): the fragments of step s+1 (two ds_read2_b32) behind the 1st and
    // 3rd MFMA of step s; the 8 global loads of the NEXT k-tile in steps 0..3; its 8 LDS writes in
    // steps 12..15.  STAGE = false: the last k-tile of the split.
    // NA (0, 1 or 2) = 32-row halves of this wave's 64 output rows that lie inside M: the last tile row
    // of an output with 4 j + few rows (dictionary training: F = 513 padded to 516 = four tile rows + 4)
    // keeps the matrix pipes of the waves that own nothing but padding idle (gemm_nt.h has the column
    // counterpart).  NA < 2 leaves the scheduling to the compiler.
    auto ktile = [&](int buf, auto stage_tag, int64_t kt_next, auto na_tag) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        constexpr int NA = decltype(na_tag)::value;
        const float* As = lds + buf * 2 * BK * BM + kk * BM + wm * 64 + li;
        const float* Bs = lds + buf * 2 * BK * BM + BK * BM + kk * BN + wn * 64 + li;
        float fa0[2], fa1[2], fb0[2], fb1[2];
        auto fetch = [&](int s) {
            if (NA == 0) return;
            fa0[s & 1] = As[2 * s * BM];
            if (NA == 2) fa1[s & 1] = As[2 * s * BM + 32];
            fb0[s & 1] = Bs[2 * s * BN];
            fb1[s & 1] = Bs[2 * s * BN + 32];
        };
        fetch(0);
        __builtin_amdgcn_sched_barrier(0);
        if (STAGE) gload(kt_next);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s < 15) fetch(s + 1);
            if (NA == 2) {
                const float a0 = fa0[s & 1], a1 = fa1[s & 1], b0 = fb0[s & 1], b1 = fb1[s & 1];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            } else if (NA == 1) {
                const float a0 = fa0[s & 1], b0 = fb0[s & 1], b1 = fb1[s & 1];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            }
        }
        if (STAGE) swrite(buf ^ 1, kt_next);
        if (NA < 2) return;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (s < 15) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // ds_read2 of step s+1
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (STAGE && s < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // global load
                if (STAGE && s >= 12) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // LDS write
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    const int na = m0 + wm * 64 + 32 < g.M ? 2 : (m0 + wm * 64 < g.M ? 1 : 0);   // (wave-uniform)
    if (kt0 < kt1) {
        gload(kt0);
        swrite(0, kt0);
        __syncthreads();
        if (na == 2) {
            for (int64_t kt = kt0; kt + 1 < kt1; ++kt) {
                ktile((int)((kt - kt0) & 1), std::true_type{}, kt + 1, std::integral_constant<int, 2>{});
                __syncthreads();
            }
            ktile((int)((kt1 - 1 - kt0) & 1), std::false_type{}, 0, std::integral_constant<int, 2>{});
        } else if (na == 1) {
            for (int64_t kt = kt0; kt + 1 < kt1; ++kt) {
                ktile((int)((kt - kt0) & 1), std::true_type{}, kt + 1, std::integral_constant<int, 1>{});
                __syncthreads();
            }
            ktile((int)((kt1 - 1 - kt0) & 1), std::false_type{}, 0, std::integral_constant<int, 1>{});
        } else {
            for (int64_t kt = kt0; kt + 1 < kt1; ++kt) {
                ktile((int)((kt - kt0) & 1), std::true_type{}, kt + 1, std::integral_constant<int, 0>{});
                __syncthreads();
            }
        }
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

// Split-K count for an M x N output contracted over Kdim: tiles x splits workgroups run in rounds of
// 512 (2 per CU); the count that minimises rounds / splits -- weighted by 1 + 0.004 s for the partial
// tiles every split writes and the combine pass reads back -- with each split keeping >= 192
// contraction steps.  A handful of output tiles (small dictionaries: F = 257, N = 200 is four) thus
// still spreads over the chip.
inline int pick_splits(int M, int N, int64_t Kdim, int max_splits) {
    const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= max_splits; ++s) {
        if (s > 1 && Kdim / s < 192) break;
        const double cost = (double)((tiles * s + 511) / 512) / s * (1.0 + 0.004 * s);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return best;
}

template <class Epi>
inline hipError_t launch_x3(const Operands& gg, const Epi& epi, dim3 grid, hipStream_t stream);   // gemm_tn_x3.h

template <class Epi>
inline hipError_t launch(const Operands& g, const Epi& epi, int splits, hipStream_t stream) {
    // the 4-wide column groups must not straddle M / N: pad the operand or take the scalar path
    const bool vec = (g.lda % 4 == 0) && (g.ldb % 4 == 0) && (g.M % 4 == 0) && (g.N % 4 == 0) &&
                     (((uintptr_t)g.A & 15) == 0) && (((uintptr_t)g.B & 15) == 0);
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    Operands gg = g;
    gg.splits = splits;
    const int per_xcd = (tiles * splits + 7) / 8;
    dim3 grid((unsigned)(per_xcd * 8));
    if (vec && tl_matrix_mode == DRNMF_MATRIX_BF16X3) return launch_x3(gg, epi, grid, stream);
    if (vec)
        hipLaunchKernelGGL((gemm_tn_kernel<Epi, true>), grid, dim3(256), 0, stream, gg, epi);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<Epi, false>), grid, dim3(256), 0, stream, gg, epi);
    return hipGetLastError();
}

}  // namespace gemm_tn

#include "gemm_tn_x3.h"
