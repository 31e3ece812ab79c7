// C[M x N] = epilogue(A[M x K] . Bt[N x K]^T) in exact fp32 on the gfx950 matrix cores
// (v_mfma_f32_32x32x2_f32).  Used by the frame-parallel paths (ISTA, multiplicative updates)
// where M = frames is large: both operands are K-contiguous ("NT"), staged through LDS in
// 128 x 32 tiles (register-staged double buffer: next tile's global loads fly under this tile's
// MFMAs), read back as 16-byte fragments along K, and the epilogue functor is applied straight
// from the accumulators (residuals, soft-threshold and multiplicative updates never round-trip
// through HBM as separate passes).
//
//   block tile 128 x 128, BK = 32, 256 threads = 2 x 2 waves, wave tile 64 x 64 = 2 x 2 MFMA tiles
//   LDS row stride 36 floats (144 B): conflict-free for the ds_read_b128 lane groups
//   MFMA contraction slots: lane group kk = l>>5, fragment element e  <->  k = 8c + 4kk + e
#pragma once
#include "common.h"

#include <type_traits>

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace gemm {

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = 36;

struct Operands {
    const float* A;   // [M][lda], K contiguous
    const float* Bt;  // [N][ldb], K contiguous
    int64_t M;
    int N, K;
    int64_t lda, ldb;
};

// VEC: lda, ldb, K multiples of 4, base pointers 16-byte aligned -> one branch-free float4 load
// (out-of-range rows / k are clamped to a valid address and the result zeroed by a select).
template <bool VEC>
__device__ __forceinline__ f32x4 load4(const float* base, int64_t row, int64_t nrows, int64_t ld,
                                       int k, int K) {
    if (VEC) {
        const bool ok = row < nrows && k < K;
        const int64_t rr = row < nrows ? row : nrows - 1;
        const int kc = k < K ? k : 0;
        f32x4 v = *(const f32x4*)(base + rr * ld + kc);
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
        return v;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < nrows) {
        const float* p = base + row * ld + k;
        if (k + 0 < K) v[0] = p[0];
        if (k + 1 < K) v[1] = p[1];
        if (k + 2 < K) v[2] = p[2];
        if (k + 3 < K) v[3] = p[3];
    }
    return v;
}

// Epi: struct with   f32x2 pre(int64_t row, int col) const        -- values the update needs from memory
//                    void operator()(int64_t row, int col, float acc, f32x2 pre) const
// The epilogue first issues all 16 `pre` loads of a 32x32 tile, then applies: one memory round trip
// per tile instead of one per element.
template <class Epi, bool VEC>
__global__ void __launch_bounds__(256) gemm_nt_kernel(const Operands g, const Epi epi) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BM * LDS_LD];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int li = l & 31, kk = l >> 5;
    const int wm = w >> 1, wn = w & 1;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int64_t tm = blockIdx.x / tiles_n;
    const int tn = blockIdx.x % tiles_n;
    const int64_t m0 = tm * BM;
    const int n0 = tn * BN;

    // staging map: thread -> (row = tid/8 + 32*i, 4 floats at k = (tid%8)*4)
    const int srow = tid >> 3, sk = (tid & 7) * 4;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

    // The loads of tile kt+1 are issued before tile kt's MFMAs and first consumed after them
    // (swrite).  In the vector path the out-of-range zeroing is therefore applied in swrite, not
    // at the load: a select right behind the load made the compiler wait for every load (vmcnt)
    // ahead of the MFMA phase, i.e. the global-load latency was exposed once per k-tile.
    f32x4 ra[4], rb[4];
    auto raw4 = [&](const float* base, int64_t row, int64_t nrows, int64_t ld, int k) {
        const int64_t rr = row < nrows ? row : nrows - 1;
        const int kc = k < g.K ? k : 0;
        return *(const f32x4*)(base + rr * ld + kc);
    };
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (VEC) {
                ra[i] = raw4(g.A, m0 + srow + 32 * i, g.M, g.lda, k0 + sk);
                rb[i] = raw4(g.Bt, n0 + srow + 32 * i, g.N, g.ldb, k0 + sk);
            } else {
                ra[i] = load4<false>(g.A, m0 + srow + 32 * i, g.M, g.lda, k0 + sk, g.K);
                rb[i] = load4<false>(g.Bt, n0 + srow + 32 * i, g.N, g.ldb, k0 + sk, g.K);
            }
        }
    };
    auto swrite = [&](int buf, int k0) {
        float* As = lds + buf * 2 * BM * LDS_LD;
        float* Bs = As + BM * LDS_LD;
        const bool kok = k0 + sk < g.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 va = ra[i], vb = rb[i];
            if (VEC) {
                if (!(kok && m0 + srow + 32 * i < g.M)) va = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!(kok && n0 + srow + 32 * i < g.N)) vb = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            *(f32x4*)(As + (srow + 32 * i) * LDS_LD + sk) = va;
            *(f32x4*)(Bs + (srow + 32 * i) * LDS_LD + sk) = vb;
        }
    };

    const int nkt = (g.K + BK - 1) / BK;
    gload(0);
    swrite(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) gload((kt + 1) * BK);   // in flight during this tile's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        const float* As = lds + buf * 2 * BM * LDS_LD + (wm * 64 + li) * LDS_LD + 4 * kk;
        const float* Bs = lds + buf * 2 * BM * LDS_LD + BM * LDS_LD + (wn * 64 + li) * LDS_LD +
                          4 * kk;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 a0 = *(const f32x4*)(As + 8 * c);
            const f32x4 a1 = *(const f32x4*)(As + 32 * LDS_LD + 8 * c);
            const f32x4 b0 = *(const f32x4*)(Bs + 8 * c);
            const f32x4 b1 = *(const f32x4*)(Bs + 32 * LDS_LD + 8 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nkt) swrite(buf ^ 1, (kt + 1) * BK);
        __syncthreads();
    }

    // epilogue: register v of lane l holds row (v&3) + 8*(v>>2) + 4*(l>>5), column l&31.
    // Row indices are formed in 32 bits (launch() refuses M >= 2^31): the functors' row * ld then
    // is one 32 x 32 -> 64-bit multiply-add instead of a 64 x 32-bit product per element.
    const int M32 = (int)g.M, m032 = (int)m0;
    auto tile_epilogue = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;   // every row of the tile is inside M
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int col = n0 + wn * 64 + b * 32 + li;
                if (col >= g.N) continue;
                f32x2 pv[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    int row = m032 + wm * 64 + a * 32 + (v & 3) + 8 * (v >> 2) + 4 * kk;
                    if (!FULL) row = row < M32 ? row : M32 - 1;
                    pv[v] = epi.pre(row, col);
                }
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = m032 + wm * 64 + a * 32 + (v & 3) + 8 * (v >> 2) + 4 * kk;
                    if (FULL || row < M32) epi(row, col, acc[a][b][v], pv[v]);
                }
            }
    };
    if (m032 + BM <= M32) tile_epilogue(std::true_type{});
    else tile_epilogue(std::false_type{});
}

template <class Epi>
inline hipError_t launch(const Operands& g, const Epi& epi, hipStream_t stream) {
    const bool vec = (g.lda % 4 == 0) && (g.ldb % 4 == 0) && (g.K % 4 == 0) &&
                     (((uintptr_t)g.A & 15) == 0) && (((uintptr_t)g.Bt & 15) == 0);
    const int64_t tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    if (tiles <= 0 || tiles > 0x7fffffff || g.M > 0x7fffff00) return hipErrorInvalidValue;
    if (vec)
        hipLaunchKernelGGL((gemm_nt_kernel<Epi, true>), dim3((unsigned)tiles), dim3(256), 0, stream,
                           g, epi);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<Epi, false>), dim3((unsigned)tiles), dim3(256), 0,
                           stream, g, epi);
    return hipGetLastError();
}

}  // namespace gemm
