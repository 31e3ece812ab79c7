// Shared host/device helpers for libdrnmf.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/drnmf.h"

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// Stores of the training chain's SAVED copies (all-hidden output, residual copy, dz / d r copies): read
// by the time-batched phase only, never by the next launch -- nontemporal (streamed through the L2
// instead of sitting there dirty until the end-of-kernel write-back).  Same-box A/B at the headline
// shape (tools/ab_builds.sh, profiles/r03e_ab_stores.txt): ordinary stores 1041-1045 ms per step,
// nontemporal 1018-1022, write-through (sc0 sc1) 1060-1063.  -DDRNMF_EXP_STSAVE=0|2: the other two.
#ifndef DRNMF_EXP_STSAVE
#define DRNMF_EXP_STSAVE 1
#endif
__device__ __forceinline__ void st_save(float* p, float v) {
#if DRNMF_EXP_STSAVE == 1
    __builtin_nontemporal_store(v, p);
#elif DRNMF_EXP_STSAVE == 2
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_save(float* p, f32x2 v) {
#if DRNMF_EXP_STSAVE == 1
    __builtin_nontemporal_store(v, (f32x2*)p);
#elif DRNMF_EXP_STSAVE == 2
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#else
    *(f32x2*)p = v;
#endif
}
__device__ __forceinline__ void st_save(float* p, f32x4 v) {
#if DRNMF_EXP_STSAVE == 1
    __builtin_nontemporal_store(v, (f32x4*)p);
#elif DRNMF_EXP_STSAVE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#else
    *(f32x4*)p = v;
#endif
}
// Column order of the saved residual / d r copies inside every 16-bin MFMA tile: position 4q + s holds
// bin 4s + q -- a lane of the chain kernels owns the bins {q, q+4, q+8, q+12} of its row (Rp packing),
// which are then ONE 16-byte store.  An involution; the weight-gradient GEMMs undo it on their output
// row index (tile_unpermute).
__host__ __device__ __forceinline__ int tile_unpermute(int m) {
    return (m & ~15) | ((m & 3) << 2) | ((m >> 2) & 3);
}
// Exchange stores (read by the NEXT launch): ordinary.  Nontemporal (-DDRNMF_EXP_STX=2) gains 0.3 % on
// the headline training step (1013-1017 ms against 1018-1022) and nothing on its inference forward, but
// costs large batches their L2 hits: B = 250 inference 8.60 -> 9.00 us per launch, B = 128 6.06 -> 6.13
// (tools/ab_shapes.sh, profiles/r03e_ab_stores.txt).  1: write-through (slower everywhere).
#ifndef DRNMF_EXP_STX
#define DRNMF_EXP_STX 0
#endif
__device__ __forceinline__ void st_xchg(float* p, float v) {
#if DRNMF_EXP_STX == 1
    asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#elif DRNMF_EXP_STX == 2
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_xchg(float* p, f32x2 v) {
#if DRNMF_EXP_STX == 1
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#elif DRNMF_EXP_STX == 2
    __builtin_nontemporal_store(v, (f32x2*)p);
#else
    *(f32x2*)p = v;
#endif
}

// v_mfma_f32_16x16x4_f32: D[i][j] += sum_{k<4} A[i][k] B[k][j]
//   lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
//   result register v of lane l is D[i = 4*(l>>4) + v][j = l&15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// sum_{q < cnt} p[q * stride] in index order with `U` loads in flight (a plain loop over a run-time count
// compiles to one dependent load per term: 65 group partials = 19 us for 260 KB in snmf.hip's w_norm_kernel).  The adds
// keep their order, out-of-range terms add +0: the sum is bit-identical to the plain loop's.
template <int U>
__device__ __forceinline__ float ordered_sum(const float* __restrict__ p, size_t stride, int cnt) {
    float acc = 0.f;
    for (int q0 = 0; q0 < cnt; q0 += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u < cnt ? q0 + u : cnt - 1;
            v[u] = p[(size_t)q * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += q0 + u < cnt ? v[u] : 0.f;
    }
    return acc;
}

// Sum over the 16 lanes of a DPP row (lanes 16i .. 16i+15), result in every lane, by four row
// rotations on the VALU (v_add_f32 ... row_ror:8/4/2/1).  __shfl_xor(x, o, 16) compiles to
// ds_bpermute_b32 -- an LDS round trip per step: eight of them in a row were ~0.2 us on the tail of
// every cell_a launch.  Every lane adds the same operand pairs (only commuted), so all 16 results
// are bit-identical.
__device__ __forceinline__ float row16_sum(float v) {
#define DRNMF_ROR(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(                 \
        0, __builtin_bit_cast(int, (x)), 0x120 + (n), 0xf, 0xf, false))
    v += DRNMF_ROR(v, 8);
    v += DRNMF_ROR(v, 4);
    v += DRNMF_ROR(v, 2);
    v += DRNMF_ROR(v, 1);
#undef DRNMF_ROR
    return v;
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline size_t round_up_sz(size_t v, size_t m) { return (v + m - 1) / m * m; }

// padded dims used by every kernel: batch rows to 16 (one MFMA M-tile), bins to 16 (one MFMA
// tile along F), atoms to 32 (one workgroup's atom block)
static inline int pad_b(int B) { return round_up(B, 16); }
static inline int pad_f(int F) { return round_up(F, 16); }
// fp16-operand mode contracts 32 bins per MFMA (v_mfma_f32_16x16x32_f16): whole 32-bin chunks
static inline int pad_f_mode(int F, bool half) { return round_up(F, half ? 32 : 16); }
static inline int pad_n(int N) { return round_up(N, 32); }

struct GraphEntry {
    std::vector<uint64_t> key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t last_stream = nullptr;   // stream of the most recent replay (eviction waits on it)
    uint64_t pin = 0;                    // drnmf_handle_s::call_seq of the call that last took it: never
                                         // evicted by that same call (its sub-batches' plans hold the exec)
};

struct drnmf_handle_s {
    // Calls on ONE handle from several host threads are serialised here (every entry point of the ABI
    // that takes a handle holds it while it validates and ENQUEUES -- the graph cache, the side streams
    // and fork / join events of split calls, the FFT tables and the error string are per-handle state);
    // handles are independent of each other.  Recursive: drnmf_destroy -> drnmf_comm_destroy.
    std::recursive_mutex mu;
    int device = 0;
    int matrix_mode = DRNMF_MATRIX_F32;  // drnmf_set_matrix_mode: how the frame-parallel products contract
    // DRNMF_MATRIX_BF16X3 only: device scratch for the Bt operand of a frame-parallel product, split into
    // its bf16 planes by a pre-pass (gemm_nt_x3.h) -- one buffer per stream that ran such a product, grown
    // on demand; outgrown buffers are parked (work already enqueued may still read them) and freed by
    // drnmf_destroy.  The one device allocation the library makes on its own (include/drnmf.h).
    struct X3Scratch { hipStream_t stream; void* ptr; size_t bytes; };
    std::vector<X3Scratch> x3_scratch;
    std::vector<void*> x3_parked;
    char err[512] = {0};
    std::vector<GraphEntry> graphs;      // least recently used first
    uint64_t call_seq = 0;               // top-level forward calls so far (GraphEntry::pin)
    // graphs dropped from the bounded cache: destroyed once the work that was enqueued when they
    // were retired has completed (graph_cache_insert, no device-wide synchronisation)
    struct Retired { GraphEntry g; hipEvent_t done; };
    std::vector<Retired> retired;
    // STFT window / twiddle tables (stft.hip): filled once per FFT size 2^6 .. 2^12
    bool fft_ready[7] = {false, false, false, false, false, false, false};
    hipEvent_t fft_event[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // RCCL communicator (comm.hip): one rank per handle / GPU
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    // persistent small-shape chains (cell_gram_persist.h): a host-mapped word the kernels raise when
    // a chain times out, and the stream / completion event of the most recent call that launched
    // them (two such launches in flight on DIFFERENT streams could each hold CUs the other's
    // workgroups need: the second call takes the launch-per-layer-step graphs instead)
    unsigned* persist_flag = nullptr;
    hipEvent_t persist_done = nullptr;
    hipStream_t persist_stream = nullptr;
    bool persist_pending = false;
    // occupancy of the two persistent kernels and the CU count of THIS handle's device / partition,
    // queried at drnmf_create (params.hip); 0 = the persistent chains are never taken
    int persist_per_cu = 0, persist_n_cu = 0;
    // Cross-PROCESS admission of the persistent chains: two processes on one GPU (several ranks of a
    // test job, a second tenant) are not coordinated by the per-handle stream admission below -- their
    // chains could each hold CUs the other's workgroups need.  The first handle on a device that
    // acquires an exclusive, non-blocking flock on /tmp/drnmf_persist_<pci bus id>.lock keeps it for its
    // lifetime and may take the chains; every other handle (other processes, a second handle of this
    // process) runs the launch-per-layer-step graphs, which compute the same bits.  The kernel drops the
    // lock when its owner dies.  -1: not the owner.
    int persist_lock_fd = -1;
    char persist_reason[256] = {0};      // why this handle is (not) admitted to the persistent chains
    // side streams + fork / join events of the sub-batch split of large inference batches
    // (cell_shared.h Workspace::split), created at first use, destroyed with the handle
    hipStream_t side_stream[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork_ev = nullptr, join_ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

// Tuning / measurement aids are read from the environment ONCE per process (at
// the first library call that consults one) into a snapshot; drnmf_reload_env() retakes it.  A size
// query and the call it sizes therefore see the same values even if the environment changes in
// between.  Returns NULL when the variable was not set.
const char* tune_env(const char* name);
// MEASUREMENT aids exist only in a -DDRNMF_MEASURE build (build.py: DRNMF_MEASURE=1), as
// the s_memtime stamps exist only in a -DDRNMF_TIMELINE one: the operand-redirecting `ablate` arguments of the
// chain kernels, empty-kernel graphs, the fifth wave's start delay, per-kernel variant overrides.  The product
// library carries none of them -- not in a kernel signature, not in an address computation, not as an
// environment read (profiles/r06_ab_no_aids.txt: the same numbers with and without).
#ifdef DRNMF_MEASURE
static inline const char* measure_env(const char* name) { return tune_env(name); }
#define DRNMF_ABLATE_PARAM int ablate_,
#define DRNMF_ABLATED(word, bit, c) (((word) & (bit)) ? 0 : (c))
#else
static inline const char* measure_env(const char*) { return nullptr; }
#define DRNMF_ABLATE_PARAM
#define DRNMF_ABLATED(word, bit, c) (c)
#endif

// cell_gram_persist.h support (params.hip)
int32_t persist_check_flag(drnmf_handle_t h);             // DRNMF_ERR_TIMEOUT once after a chain gave up
bool persist_admit(drnmf_handle_t h, hipStream_t stream); // false: another stream's persistent launches may still run
void persist_mark(drnmf_handle_t h, hipStream_t stream);  // after a call's persistent launches
void persist_query_occupancy(int device, int* per_cu, int* n_cu);   // cell_forward.hip, at drnmf_create

// Bounded graph cache shared by the forward / backward / dense cells.  Evicting an entry must not
// synchronise the device (ABI contract: calls only enqueue): the evicted executable may still be
// replaying on `stream`, so it is parked with an event recorded on that stream and destroyed by a
// later call once the event has completed.
int32_t graph_cache_make_room(drnmf_handle_t h, hipStream_t stream, size_t max_entries);

extern char g_create_err[512];

// The matrix mode of the handle whose entry point this thread is inside (drnmf_set_matrix_mode): read by
// gemm::launch / gemm_tn::launch, which see operands and a stream but no handle.
extern thread_local int tl_matrix_mode;
extern thread_local drnmf_handle_t tl_handle;
struct MatrixModeScope {
    int prev;
    drnmf_handle_t prev_h;
    explicit MatrixModeScope(drnmf_handle_t h) : prev(tl_matrix_mode), prev_h(tl_handle) {
        if (h) { tl_matrix_mode = h->matrix_mode; tl_handle = h; }
    }
    ~MatrixModeScope() { tl_matrix_mode = prev; tl_handle = prev_h; }
};
// >= bytes of device scratch for split operands, private to (the current entry point's handle, stream);
// NULL without a handle / device or when the allocation fails (the caller then takes the fp32 kernels)
void* x3_scratch_get(hipStream_t stream, size_t bytes);

#define DRNMF_LOCK(h)                                        \
    std::unique_lock<std::recursive_mutex> handle_lock_;     \
    if (h) handle_lock_ = std::unique_lock<std::recursive_mutex>((h)->mu); \
    MatrixModeScope matrix_mode_scope_(h)

#define DRNMF_FAIL(h, code, ...)                                  \
    do {                                                          \
        if (h) snprintf((h)->err, sizeof((h)->err), __VA_ARGS__); \
        return (code);                                            \
    } while (0)

#define DRNMF_HIP(h, expr)                                                               \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            DRNMF_FAIL(h, DRNMF_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                 \
                       hipGetErrorString(e_), __FILE__, __LINE__);                       \
        }                                                                                \
    } while (0)

// Position of element (row r, column c), r, c in 0..15, inside a 1 KB block of a tile-packed
// activation / residual buffer.  Both orders put the four values one MFMA lane (row r, k-group q)
// needs in one 16-byte piece at lane * 16 bytes (lane = q*16 + r): consecutive lanes read
// consecutive memory (lane order matters to the texture path: +1.3 % per operand on the headline).
//   hp_pos: lane's values = atoms 4q..4q+3 (cell_b contracts consecutive atoms per MFMA)
//   rp_pos: lane's values = bins q, 4+q, 8+q, 12+q (cell_a contracts bin 4s+q in its s-th MFMA)
__host__ __device__ static inline int hp_pos(int r, int c) { return ((c >> 2) * 16 + r) * 4 + (c & 3); }
__host__ __device__ static inline int rp_pos(int r, int c) { return ((c & 3) * 16 + r) * 4 + (c >> 2); }

constexpr int MAX_TAIL = 2;   // STFT sizes are 2^k + 1: the odd bin(s) must not cost a whole 16-bin tile

using f16 = _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

// v_mfma_f32_16x16x32_f16 (gfx950): D[i][j] += sum_{k<32} A[i][k] B[k][j]; lane l supplies the 8
// k-slots (q = l>>4, e = 0..7) of row i = l&15 of A and of column j = l&15 of B -- slot (q, e) of A
// meets slot (q, e) of B, which is all the packings below rely on; D as in mfma16.
__device__ __forceinline__ f32x4 mfma32h(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// params block layout (see drnmf_prepare_params).  off_dn: per stored layer either the fp32
// tile-packed dictionary Dp[ft][ac][f%16][n%16] or, with operand_f16, two fp16 packings of the same
// bytes in total (cell_a's, then cell_b's; params.hip; Fp is then a multiple of 32).  off_tail: fp32 rows of the tail bins
// [n_D][MAX_TAIL][Np] (fp16-rounded values in operand_f16 mode).  off_dnA (fp32 mode only): a
// second packing of the dictionary for cell_a / bwd_a, whose lanes need 2 atoms x 4 bins per chunk:
// DpA[ft][n/32][s/2][q][j][s%2][a] = Dn[16 ft + 4s + q][32 (n/32) + 2j + a], so that the operand
// comes in two 16-byte loads per lane over contiguous 1 KB (8-byte loads run at half the
// texture-path rate).
// Gram form of the cell (cell_gram.h): h S_k with S_k = I - G_k diag(1/alpha_k), G_k = Dn_k^T Dn_k
// -- the matrix the reference itself materialises (enhance.py:172-181) -- is ONE launch per
// layer-step instead of the factored pair.  Worth it when the batch is small (each launch is
// latency, not work) and N x N per stored layer is affordable; the prepared block then also holds
//   off_gram: G packed as cell_b's operand, [n_D][Np/16][Np/16][q][o%16][e] = G[o][16 ac + 4q + e]
//   off_dnT:  Dn^T row-major [n_D][Np][Fp] (operand of the frame-parallel x Dn_k product)
//   off_dn_rm: scratch, Dn row-major [n_D][round_up(Fp,32)][Np] (zero rows behind Fp): the operand of ONE
//              split-by-layer TN product for all Gram matrices
// for fp32 Euclidean descriptors with N <= GRAM_MAX_N (a property of the descriptor's F, N, K only:
// the same block serves every batch size).
// The bound is the one gram_wanted() (cell_shared.h) applies to a single row tile: a dictionary wider
// than that can never take the Gram form, so its block must not carry (and every prepare_params
// must not rebuild) G_k and Dn_k^T -- 0.5 GB and n_D extra GEMMs per training step at N = 2000,
// K = 25.  DRNMF_GRAM=1 (tuning aid: force the form) widens eligibility to GRAM_MAX_N.
constexpr int GRAM_MAX_N = 4096;
constexpr int64_t GRAM_MAX_WORK = 3000000;      // (row tiles) * Np^2 of gram_wanted()
static inline bool gram_eligible(const drnmf_cell_desc_t* d) {
    if (d->operand_f16 || d->divergence != DRNMF_DIV_ED || d->K < 2 || d->N > GRAM_MAX_N) return false;
    const int64_t Np = pad_n(d->N);
    if (Np * Np <= GRAM_MAX_WORK) return true;
    const char* e = tune_env("DRNMF_GRAM");
    return e && atoi(e) == 1;
}

struct ParamsLayout {
    int Fp, Np;
    size_t off_dn, off_colnorm, off_inv_alpha, off_bias, off_tail, off_dnA, total;
    size_t off_gram, off_dnT, off_dn_rm;   // (0 unless gram_eligible)
    size_t off_dn32;      // operand_f16 only: the fp32 cell_b packing followed by the fp32 cell_a
                          // packing of every stored layer -- the BPTT of a model whose FORWARD
                          // runs on fp16 operands is computed in fp32 (mixed-precision training)
};
static inline ParamsLayout params_layout(const drnmf_cell_desc_t* d) {
    ParamsLayout L;
    L.Fp = pad_f_mode(d->F, d->operand_f16 != 0);
    L.Np = pad_n(d->N);
    size_t o = 0;
    // (fp32: cell_b's packing, cell_a's at off_dnA; operand_f16: the ONE fp16 packing both kernels read)
    L.off_dn = o;        o += round_up_sz((size_t)d->n_D * L.Fp * L.Np * (d->operand_f16 ? 2 : 4), 256);
    L.off_colnorm = o;   o += round_up_sz((size_t)d->n_D * L.Np * sizeof(float), 256);
    L.off_inv_alpha = o; o += round_up_sz((size_t)d->K * L.Np * sizeof(float), 256);
    L.off_bias = o;      o += round_up_sz((size_t)d->K * L.Np * sizeof(float), 256);
    L.off_tail = o;      o += round_up_sz((size_t)d->n_D * MAX_TAIL * L.Np * sizeof(float), 256);
    L.off_dnA = o;
    if (!d->operand_f16) o += (size_t)d->n_D * L.Fp * L.Np * sizeof(float);
    L.off_dn32 = 0;
    if (d->operand_f16) {
        L.off_dn32 = o;
        o += (size_t)2 * d->n_D * L.Fp * L.Np * sizeof(float);
    }
    L.off_gram = L.off_dnT = L.off_dn_rm = 0;
    if (gram_eligible(d)) {
        L.off_gram = o;  o += (size_t)d->n_D * L.Np * L.Np * sizeof(float);
        L.off_dnT = o;   o += (size_t)d->n_D * L.Np * L.Fp * sizeof(float);
        L.off_dn_rm = o; o += (size_t)d->n_D * round_up(L.Fp, 32) * L.Np * sizeof(float);
    }
    L.total = o;
    return L;
}

int validate_cell_desc(drnmf_handle_t h, const drnmf_cell_desc_t* d);
