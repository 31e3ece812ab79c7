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
    int ktail = 0;    // extra contraction columns K .. K+ktail-1 (<= 2), see below
    int thin = 0;     // 1: output column N (row N of Bt) exists BEHIND the N tiled columns, see THIN below
    // split-operand mode only (gemm_nt_x3.h; set by launch()): Bt already split into bf16 planes
    const void* B3 = nullptr;
    int kt3 = 0;
    int per_xcd = 0;  // tiles per XCD of the XCD-aware workgroup -> tile map (0: identity)
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

// Epi: struct with   f32x2 pre(int64_t row, int col) const   -- per-element values from memory (raw loads)
//                    void operator()(int64_t row, int col, float acc, f32x2 pre) const
// All of a workgroup's pre loads (its 128 x 128 outputs) are issued BEFORE the MFMAs of the last
// k-tile and first consumed after them: the epilogue's memory round trip -- with the short
// contractions here (K = 512: 16 k-tiles) a tenth of a workgroup's life when it was paid once per
// 32 x 32 tile after the loop -- hides under matrix work.  `pre` therefore returns what it loaded
// untouched (arithmetic on a loaded value would pull its wait in front of the MFMAs).
// `static constexpr bool EARLY = false` in the functor keeps the per-tile order (register budget:
// the kernel must stay at two waves per SIMD, a spilled loaded value waits for its load).
//
// Operands::ktail (0..2): contraction indices K .. K+ktail-1 are taken from the same arrays but
// outside the k-tiles -- one extra MFMA step per output tile whose two k slots are those columns,
// fragments fetched straight from global memory with the early loads.  This is how the odd bins of
// a 2^k+1 STFT (K = 513 = 16 tiles + 1) ride along without a 17th, almost empty k-tile.
// `static constexpr bool REDUCE = true` in the functor: its operator() RETURNS a float per output element;
// the kernel sums them per thread (fixed order), per workgroup (LDS tree, fixed order) and stores ONE
// partial per workgroup at epi.red_out[blockIdx.x] (blockIdx.x < launch_tiles(g)) -- a grid-wide sum (the objective of the dictionary
// training) rides on a GEMM's epilogue instead of costing its own pass over the outputs; deterministic.
template <class E, class = void> struct epi_reduce : std::false_type {};
template <class E> struct epi_reduce<E, std::void_t<decltype(E::REDUCE)>>
    : std::integral_constant<bool, E::REDUCE> {};

template <class E, class = void> struct epi_early : std::true_type {};
template <class E> struct epi_early<E, std::void_t<decltype(E::EARLY)>>
    : std::integral_constant<bool, E::EARLY> {};

// THIN (launch() sets it up: VEC, no ktail, N = 128 j + 1 -- the 513 / 257 / 1025 bins of a 2^k + 1 STFT):
// the one column behind the last full tile is not given a fifth, 1/128-full column of workgroups (a
// fifth of the launch, even with its MFMAs switched off: NB below).  Every A value passes through the
// registers of ONE thread of every column tile's workgroup on its way to LDS (gload -> swrite); the
// thread multiplies them with the matching 16 bytes of Bt's row N -- one extra 16-byte load and 16 FMAs
// per thread and k-tile, no extra pass over A; the eight k-lanes of a row are added after the loop and
// the workgroup of column tile tn stores the rows of staging slice i = tn (srow + 32 i).
template <class Epi, bool VEC, bool THIN = false>
__global__ void __launch_bounds__(256, 2) gemm_nt_kernel(const Operands g, const Epi epi) {
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
    f32x4 ra[4], rb[4], rt = {0.f, 0.f, 0.f, 0.f};
    float tacc[4] = {0.f, 0.f, 0.f, 0.f};
    // staging slices whose thin-column products this workgroup owns (workgroup-uniform)
    bool mine[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mine[i] = THIN && (tiles_n >= 4 ? i == tn : i % tiles_n == tn);
    auto raw4 = [&](const float* base, int64_t row, int64_t nrows, int64_t ld, int k) {
        const int64_t rr = row < nrows ? row : nrows - 1;
        const int kc = k < g.K ? k : 0;
        return *(const f32x4*)(base + rr * ld + kc);
    };
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (VEC) {
                // (row N of Bt first: swrite's first use of it then waits for the oldest load only)
                if (THIN && i == 0) rt = raw4(g.Bt, g.N, g.N + 1, g.ldb, k0 + sk);
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
            if (THIN) {      // (all four slices, branch-free: a workgroup-uniform `if (mine[i])` splits the
                f32x4 vt = rt;   // k-tile's basic block and with it the MFMA / LDS-write interleave)
                if (!kok) vt = f32x4{0.f, 0.f, 0.f, 0.f};
                tacc[i] += (va[0] * vt[0] + va[1] * vt[1]) + (va[2] * vt[2] + va[3] * vt[3]);
            }
        }
    };
    // One k-tile = 4 chunks of 8 contraction slots = 64 MFMAs per wave.  Everything else a k-tile
    // needs is pinned between them with sched_group_barrier (left to itself the compiler issued the
    // LDS reads in bursts and waited for them in front of the MFMAs, twice per k-tile):
    //   chunk 0: fragments of chunk 1 (one ds_read_b128 behind every 2nd MFMA), then the 8 global
    //            loads of the NEXT k-tile (one behind every MFMA)
    //   chunk 1, 2: fragments of the next chunk in the first half, 8 bare MFMAs for them to land
    //   chunk 3: the 8 LDS writes of the next k-tile (global loads issued ~3000 cycles earlier)
    // STAGE = false (last k-tile): reads and MFMAs only.
    // NB (0, 1 or 2) = 32-column halves of this wave's 64 output columns that lie inside N: the last
    // column tile of a 2^k+1-wide output (F = 513: column 512 alone) keeps 3 of its 4 waves' MFMAs off the
    // matrix pipes instead of running a full 128-column tile for one column (the staging loads, LDS writes
    // and barriers are still shared by all four waves).  NB < 2 leaves the scheduling to the compiler.
    auto ktile = [&](int buf, auto stage_tag, int k_next, auto nb_tag) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        constexpr int NB = decltype(nb_tag)::value;
        const float* As = lds + buf * 2 * BM * LDS_LD + (wm * 64 + li) * LDS_LD + 4 * kk;
        const float* Bs = lds + buf * 2 * BM * LDS_LD + BM * LDS_LD + (wn * 64 + li) * LDS_LD +
                          4 * kk;
        f32x4 fa0[2], fa1[2], fb0[2], fb1[2];
        auto fetch = [&](int c) {
            if (NB == 0) return;
            fa0[c & 1] = *(const f32x4*)(As + 8 * c);
            fb0[c & 1] = *(const f32x4*)(Bs + 8 * c);
            fa1[c & 1] = *(const f32x4*)(As + 32 * LDS_LD + 8 * c);
            if (NB == 2) fb1[c & 1] = *(const f32x4*)(Bs + 32 * LDS_LD + 8 * c);
        };
        fetch(0);
        __builtin_amdgcn_sched_barrier(0);
        if (STAGE) gload(k_next);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 3) fetch(c + 1);
            if (NB == 2) {
                const f32x4 a0 = fa0[c & 1], a1 = fa1[c & 1], b0 = fb0[c & 1], b1 = fb1[c & 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b1[e], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[1][1], 0, 0, 0);
                }
            } else if (NB == 1) {
                const f32x4 a0 = fa0[c & 1], a1 = fa1[c & 1], b0 = fb0[c & 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[0][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b0[e], acc[1][0], 0, 0, 0);
                }
            }
        }
        if (STAGE) swrite(buf ^ 1, k_next);
        if (NB < 2) return;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    // 2 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // 1 LDS read of the next chunk
            }
            if (STAGE && c == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 global load
                }
                if (THIN) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // (row N of Bt)
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);       // reads land under these
            }
        }
        if (STAGE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // 1 LDS write
            }
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nkt = (g.K + BK - 1) / BK;
    // (wave-uniform: tn, wn and N are)
    const int nb = n0 + wn * 64 + 32 < g.N ? 2 : (n0 + wn * 64 < g.N ? 1 : 0);
    gload(0);
    swrite(0, 0);
    __syncthreads();
    if (nb == 2) {
        for (int kt = 0; kt + 1 < nkt; ++kt) {
            ktile(kt & 1, std::true_type{}, (kt + 1) * BK, std::integral_constant<int, 2>{});
            __syncthreads();
        }
    } else if (nb == 1) {
        for (int kt = 0; kt + 1 < nkt; ++kt) {
            ktile(kt & 1, std::true_type{}, (kt + 1) * BK, std::integral_constant<int, 1>{});
            __syncthreads();
        }
    } else {
        for (int kt = 0; kt + 1 < nkt; ++kt) {
            ktile(kt & 1, std::true_type{}, (kt + 1) * BK, std::integral_constant<int, 0>{});
            __syncthreads();
        }
    }

    // Last k-tile and epilogue.  Register v of lane l holds row (v&3) + 8*(v>>2) + 4*(l>>5),
    // column l&31.  Row indices are formed in 32 bits (launch() refuses M >= 2^31): the functors'
    // row * ld then is one 32 x 32 -> 64-bit multiply-add instead of a 64 x 32-bit product.
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
        if (nb == 2) ktile((nkt - 1) & 1, std::false_type{}, 0, std::integral_constant<int, 2>{});
        else if (nb == 1) ktile((nkt - 1) & 1, std::false_type{}, 0, std::integral_constant<int, 1>{});
        if (g.ktail && nb > 0) {
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
                    // opaque copy: the store addresses are formed again here instead of 64 64-bit
                    // load addresses staying live across the MFMAs (that cost the second wave per SIMD)
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
    if constexpr (RED) {
        __syncthreads();                       // every wave is done with the staged tiles
        lds[tid] = red;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) lds[tid] += lds[tid + o];
            __syncthreads();
        }
        if (tid == 0) epi.red_out[blockIdx.x] = lds[0];
    }
}

inline bool vec_ok(const Operands& g) {
    return (g.lda % 4 == 0) && (g.ldb % 4 == 0) && (g.K % 4 == 0) && (((uintptr_t)g.A & 15) == 0) &&
           (((uintptr_t)g.Bt & 15) == 0);
}
// N = 128 j + 1: the odd column rides on the full tiles' staging (THIN above)
inline bool thin_applies(const Operands& g) {
    const char* te = measure_env("DRNMF_THIN");                  // measurement aid: 0 = a tile column of its own
    return vec_ok(g) && g.ktail == 0 && g.N > BN && g.N % BN == 1 && !(te && atoi(te) == 0);
}
// workgroups of launch(g, ...) = partials a REDUCE epilogue leaves at red_out[0 .. launch_tiles)
inline int64_t launch_tiles(const Operands& g) {
    const int n = thin_applies(g) ? g.N - 1 : g.N;
    return ((g.M + BM - 1) / BM) * ((n + BN - 1) / BN);
}

// gemm_nt_x3.h: the same product with split operands (matrix mode DRNMF_MATRIX_BF16X3)
template <class Epi>
inline hipError_t launch_x3(const Operands& g, bool thin, const Epi& epi, hipStream_t stream, bool* taken);

template <class Epi>
inline hipError_t launch(const Operands& g_in, const Epi& epi, hipStream_t stream) {
    Operands g = g_in;
    const bool vec = vec_ok(g);
    const bool thin = thin_applies(g);
    if (thin) { g.N -= 1; g.thin = 1; }
    const int64_t tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    if (tiles <= 0 || tiles > 0x7fffffff || g.M > 0x7fffff00 || g.ktail < 0 || g.ktail > 2)
        return hipErrorInvalidValue;
    if (vec && tl_matrix_mode == DRNMF_MATRIX_BF16X3) {
        bool taken = false;
        const hipError_t e = launch_x3(g, thin, epi, stream, &taken);
        if (taken || e != hipSuccess) return e;
    }
    if (thin)
        hipLaunchKernelGGL((gemm_nt_kernel<Epi, true, true>), dim3((unsigned)tiles), dim3(256), 0, stream,
                           g, epi);
    else if (vec)
        hipLaunchKernelGGL((gemm_nt_kernel<Epi, true>), dim3((unsigned)tiles), dim3(256), 0, stream,
                           g, epi);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<Epi, false>), dim3((unsigned)tiles), dim3(256), 0,
                           stream, g, epi);
    return hipGetLastError();
}

}  // namespace gemm

#include "gemm_nt_x3.h"
