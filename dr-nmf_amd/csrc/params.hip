// Handle management + parameter maps (log-domain SNMF parameters -> prepared per-layer block).
// Reference: build_alt's maps_from_alt (enhance.py:161-204), evaluated in
// SimpleDeepRNN.build (custom_layers.py:234-287).
#include "common.h"
#include "gemm_tn.h"

#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/file.h>
#include <unistd.h>

#include <map>
#include <mutex>
#include <string>

char g_create_err[512] = {0};

// ---- tuning-variable snapshot (common.h) ---------------------------------------------------------
namespace {
const char* const kTuneNames[] = {
    "DRNMF_ABLATE", "DRNMF_ABLATE_A", "DRNMF_ABLATE_B", "DRNMF_CP_FULL", "DRNMF_DENSE_NW", "DRNMF_FPG",
    "DRNMF_GRAM", "DRNMF_KS", "DRNMF_LATE", "DRNMF_NO_ALLB", "DRNMF_NO_GRAPH",
    "DRNMF_PERSIST", "DRNMF_PERSIST_FAULT", "DRNMF_RB", "DRNMF_RBA", "DRNMF_SPLIT",
    "DRNMF_THIN", "DRNMF_PF", "DRNMF_PF_SLEEP", "DRNMF_NT_XCD"};
std::mutex g_tune_mu;
// (values are never erased or overwritten in place: a pointer handed out stays valid for the
// process lifetime; a reload appends a new generation)
std::vector<std::map<std::string, std::string>*> g_tune_gen;
void tune_take_locked() {
    auto* m = new std::map<std::string, std::string>();
    for (const char* n : kTuneNames)
        if (const char* v = getenv(n)) (*m)[n] = v;
    g_tune_gen.push_back(m);
}
}  // namespace

const char* tune_env(const char* name) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    if (g_tune_gen.empty()) tune_take_locked();
    const auto& m = *g_tune_gen.back();
    auto it = m.find(name);
    return it == m.end() ? nullptr : it->second.c_str();
}

extern "C" int32_t drnmf_reload_env(void) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    tune_take_locked();
    return DRNMF_OK;
}

extern "C" int32_t drnmf_version(void) { return DRNMF_VERSION; }

thread_local int tl_matrix_mode = DRNMF_MATRIX_F32;
thread_local drnmf_handle_t tl_handle = nullptr;

void* x3_scratch_get(hipStream_t stream, size_t bytes) {
    drnmf_handle_t h = tl_handle;          // (the entry point that got us here holds h->mu)
    if (!h || h->device < 0) return nullptr;
    for (auto& s : h->x3_scratch) {
        if (s.stream != stream) continue;
        if (s.bytes >= bytes) return s.ptr;
        void* p = nullptr;
        const size_t want = bytes + bytes / 2;
        if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        h->x3_parked.push_back(s.ptr);
        s.ptr = p;
        s.bytes = want;
        return p;
    }
    void* p = nullptr;
    const size_t want = bytes < ((size_t)16 << 20) ? ((size_t)16 << 20) : bytes;
    if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    h->x3_scratch.push_back({stream, p, want});
    return p;
}

extern "C" int32_t drnmf_set_matrix_mode(drnmf_handle_t h, int32_t mode) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (mode != DRNMF_MATRIX_F32 && mode != DRNMF_MATRIX_BF16X3)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "set_matrix_mode: unknown mode %d", mode);
    h->matrix_mode = mode;
    return DRNMF_OK;
}

extern "C" int32_t drnmf_get_matrix_mode(drnmf_handle_t h) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    return h->matrix_mode;
}

extern "C" int32_t drnmf_create(drnmf_handle_t* out, int32_t device) {
    if (!out) return DRNMF_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        snprintf(g_create_err, sizeof(g_create_err), "no HIP device visible (%s)",
                 e == hipSuccess ? "count=0" : hipGetErrorString(e));
        return DRNMF_ERR_HIP;
    }
    if (device < 0 || device >= n) {
        snprintf(g_create_err, sizeof(g_create_err), "device %d out of range [0,%d)", device, n);
        return DRNMF_ERR_INVALID_ARG;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        snprintf(g_create_err, sizeof(g_create_err), "hipGetDeviceProperties: %s",
                 hipGetErrorString(e));
        return DRNMF_ERR_HIP;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_create_err, sizeof(g_create_err),
                 "device %d is %s; libdrnmf is built for gfx950 (MI355X) only", device,
                 prop.gcnArchName);
        return DRNMF_ERR_UNSUPPORTED;
    }
    drnmf_handle_t h = new (std::nothrow) drnmf_handle_s();
    if (!h) return DRNMF_ERR_HIP;
    h->device = device;
    // (mapped + coherent: a kernel's system-scope store is seen by the host without a synchronise)
    // (4 KB: the fault word in the first 64 bytes, the report ring of drnmf_host_report_ring behind it)
    if (hipHostMalloc((void**)&h->persist_flag, 4096, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess)
        memset(h->persist_flag, 0, 4096);
    else {
        h->persist_flag = nullptr;
        (void)hipGetLastError();
    }
    // occupancy of the persistent chain kernels on THIS device / partition (cell_forward.hip)
    persist_query_occupancy(device, &h->persist_per_cu, &h->persist_n_cu);
    {   // cross-process admission of the persistent chains (common.h)
        char bus[64] = {0}, path[128];
        if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, device) != hipSuccess) {
            (void)hipGetLastError();
            snprintf(bus, sizeof(bus), "dev%d", device);
        }
        for (char* c = bus; *c; ++c)
            if (!((*c >= '0' && *c <= '9') || (*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z'))) *c = '_';
        snprintf(path, sizeof(path), "/tmp/drnmf_persist_%s.lock", bus);
        // (a predictable name in a shared directory: never follow a link another user planted there, and widen
        // the mode only of a regular, singly linked file this user owns -- ADVICE r5)
        int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0666);
        struct stat st;
        if (fd >= 0 && (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_nlink != 1)) {
            close(fd);
            fd = -1;
            errno = ELOOP;
        }
        if (fd >= 0) {
            // (world-writable whatever the creator's umask: the next user of the box must be able to open it)
            if (st.st_uid == geteuid()) (void)fchmod(fd, 0666);
            if (flock(fd, LOCK_EX | LOCK_NB) == 0) {
                h->persist_lock_fd = fd;
                snprintf(h->persist_reason, sizeof(h->persist_reason), "admitted (holds %s)", path);
            } else {
                snprintf(h->persist_reason, sizeof(h->persist_reason),
                         "not admitted: %s is held by another handle / process on this GPU (%s)", path,
                         strerror(errno));
                close(fd);
            }
        } else {
            snprintf(h->persist_reason, sizeof(h->persist_reason), "not admitted: open(%s) failed: %s", path,
                     strerror(errno));
        }
        if (h->persist_lock_fd >= 0 && !h->persist_flag)
            snprintf(h->persist_reason, sizeof(h->persist_reason),
                     "not admitted: no host-mapped fault word (hipHostMalloc failed)");
        else if (h->persist_lock_fd >= 0 && h->persist_per_cu < 1)
            snprintf(h->persist_reason, sizeof(h->persist_reason),
                     "not admitted: the persistent kernels do not fit a CU of this device / partition");
    }
    *out = h;
    return DRNMF_OK;
}

// drnmf_check_status: the handle's asynchronous fault word (raised by a persistent chain that gave
// up), read AND cleared.  Meaningful after the caller has synchronised the stream of the call in
// question: that call itself returned DRNMF_OK when it was enqueued.
extern "C" int32_t drnmf_check_status(drnmf_handle_t h) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    return persist_check_flag(h);
}

// 1: this handle may run the persistent small-shape chains (it owns the device's cross-process lock and
// has a fault word); 0: it always takes the launch-per-layer-step graphs (same results).
extern "C" int32_t drnmf_persist_admitted(drnmf_handle_t h) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    return (h->persist_lock_fd >= 0 && h->persist_flag != nullptr && h->persist_per_cu >= 1) ? 1 : 0;
}

extern "C" const char* drnmf_persist_admit_reason(drnmf_handle_t h) {
    if (!h) return "no handle";
    return h->persist_reason[0] ? h->persist_reason : "not admitted: a handle without a device";
}

// A small ring of 4-float slots in host-mapped, coherent memory that kernels may write (the report of
// drnmf_adam_step_flat): the host reads a slot after waiting for an event recorded behind the kernel --
// no device-to-host copy, no stream synchronisation.  Owned by the handle.
extern "C" int32_t drnmf_host_report_ring(drnmf_handle_t h, float** ring_host, int32_t* slots) {
    DRNMF_LOCK(h);
    if (!h || !ring_host || !slots) return DRNMF_ERR_INVALID_ARG;
    if (!h->persist_flag) DRNMF_FAIL(h, DRNMF_ERR_HIP, "host_report_ring: no host-mapped memory on this handle");
    *ring_host = (float*)((char*)h->persist_flag + 256);
    *slots = (4096 - 256) / 16;
    return DRNMF_OK;
}

// Stream-ordered variant for callers that do not synchronise (the training step): a one-thread kernel
// takes the fault word (read + clear) and ADDS 1.0f to *dst when it was raised.
__global__ void status_take_kernel(unsigned* flag, float* dst) {
    const unsigned v = __hip_atomic_exchange(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (v != 0u) *dst += 1.0f;
}
extern "C" int32_t drnmf_status_take_device(drnmf_handle_t h, float* dst_device, void* stream_) {
    DRNMF_LOCK(h);
    if (!h || !dst_device) return DRNMF_ERR_INVALID_ARG;
    if (!h->persist_flag) return DRNMF_OK;       // (no fault word: the persistent chains are never taken)
    hipLaunchKernelGGL(status_take_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, h->persist_flag,
                       dst_device);
    DRNMF_HIP(h, hipGetLastError());
    return DRNMF_OK;
}

// A handle bound to NO device: size queries, descriptor validation and argument checks behave as on a
// real handle; whatever would touch the GPU fails with DRNMF_ERR_HIP.  For hosts without a GPU (the
// CPU-side sanitizer run of the ABI's host half, tests/test_sanitize.py); no product path creates one.
extern "C" int32_t drnmf_create_unbound(drnmf_handle_t* out) {
    if (!out) return DRNMF_ERR_INVALID_ARG;
    drnmf_handle_t h = new (std::nothrow) drnmf_handle_s();
    if (!h) return DRNMF_ERR_HIP;
    h->device = -1;
    *out = h;
    return DRNMF_OK;
}

int32_t persist_check_flag(drnmf_handle_t h) {
    if (!h->persist_flag || *(volatile unsigned*)h->persist_flag == 0u) return DRNMF_OK;
    *(volatile unsigned*)h->persist_flag = 0u;
    DRNMF_FAIL(h, DRNMF_ERR_TIMEOUT,
               "a persistent small-shape launch of an earlier call on this handle timed out waiting "
               "for its own workgroups (not resident together: other persistent kernels on this GPU?); "
               "that call's result is invalid -- rerun it, or set DRNMF_PERSIST=0");
}

bool persist_admit(drnmf_handle_t h, hipStream_t stream) {
    if (!h->persist_flag) return false;                  // no way to report a timeout: never take the path
    if (!h->persist_pending || h->persist_stream == stream) return true;
    if (hipEventQuery(h->persist_done) == hipSuccess) { h->persist_pending = false; return true; }
    (void)hipGetLastError();                              // hipErrorNotReady is not an error
    return false;
}

void persist_mark(drnmf_handle_t h, hipStream_t stream) {
    if (!h->persist_done && hipEventCreateWithFlags(&h->persist_done, hipEventDisableTiming) != hipSuccess) {
        h->persist_done = nullptr;
        (void)hipGetLastError();
        return;
    }
    if (hipEventRecord(h->persist_done, stream) == hipSuccess) {
        h->persist_stream = stream;
        h->persist_pending = true;
    }
}

extern "C" int32_t drnmf_comm_destroy(drnmf_handle_t h);

extern "C" int32_t drnmf_destroy(drnmf_handle_t h) {
    if (!h) return DRNMF_ERR_INVALID_ARG;
    (void)drnmf_comm_destroy(h);
    for (auto& r : h->retired) {      // (destroy is the one call that may wait)
        (void)hipEventSynchronize(r.done);
        (void)hipEventDestroy(r.done);
        if (r.g.exec) (void)hipGraphExecDestroy(r.g.exec);
        if (r.g.graph) (void)hipGraphDestroy(r.g.graph);
    }
    for (auto& g : h->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    for (auto& e : h->fft_event)
        if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < 7; ++i) {
        if (h->side_stream[i]) { (void)hipStreamSynchronize(h->side_stream[i]); (void)hipStreamDestroy(h->side_stream[i]); }
        if (h->join_ev[i]) (void)hipEventDestroy(h->join_ev[i]);
    }
    if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
    if (h->persist_done) (void)hipEventDestroy(h->persist_done);
    if (h->persist_flag) (void)hipHostFree(h->persist_flag);
    if (h->persist_lock_fd >= 0) close(h->persist_lock_fd);      // (releases the flock)
    if (!h->x3_scratch.empty() || !h->x3_parked.empty()) {
        // (work that reads the scratch may still be in flight on THIS handle's device, which need not be current)
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (h->device >= 0 && cur != h->device) (void)hipSetDevice(h->device);
        (void)hipDeviceSynchronize();
        for (auto& s : h->x3_scratch) (void)hipFree(s.ptr);
        for (void* p : h->x3_parked) (void)hipFree(p);
        if (cur >= 0 && cur != h->device) (void)hipSetDevice(cur);
    }
    delete h;
    return DRNMF_OK;
}

int32_t graph_cache_make_room(drnmf_handle_t h, hipStream_t stream, size_t max_entries) {
    // reap retired graphs whose last replay has completed (a query, never a wait)
    for (size_t i = 0; i < h->retired.size();) {
        if (hipEventQuery(h->retired[i].done) == hipSuccess) {
            (void)hipEventDestroy(h->retired[i].done);
            if (h->retired[i].g.exec) (void)hipGraphExecDestroy(h->retired[i].g.exec);
            if (h->retired[i].g.graph) (void)hipGraphDestroy(h->retired[i].g.graph);
            h->retired.erase(h->retired.begin() + i);
        } else {
            ++i;
        }
    }
    (void)hipGetLastError();          // hipErrorNotReady of the query is not an error
    while (h->graphs.size() >= max_entries) {
        // the least recently used entry that the CURRENT call does not hold (a split call's sub-batches
        // each hold up to two executables until their launches are enqueued); all held: grow instead
        size_t victim = 0;
        while (victim < h->graphs.size() && h->graphs[victim].pin == h->call_seq && h->call_seq != 0) ++victim;
        if (victim == h->graphs.size()) break;
        drnmf_handle_s::Retired r;
        r.g = h->graphs[victim];
        h->graphs.erase(h->graphs.begin() + (ptrdiff_t)victim);
        DRNMF_HIP(h, hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
        // the evicted executable was last replayed on r.g.last_stream (this call's stream when it
        // never ran): everything enqueued there so far precedes the event
        DRNMF_HIP(h, hipEventRecord(r.done, r.g.last_stream ? r.g.last_stream : stream));
        h->retired.push_back(r);
    }
    return DRNMF_OK;
}

extern "C" const char* drnmf_last_error(drnmf_handle_t h) { return h ? h->err : g_create_err; }

int validate_cell_desc(drnmf_handle_t h, const drnmf_cell_desc_t* d) {
    if (!d) DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "desc is NULL");
    if (d->B <= 0 || d->T <= 0 || d->F <= 0 || d->N <= 0 || d->K <= 0)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "B,T,F,N,K must be positive (got %d,%d,%d,%d,%d)",
                   d->B, d->T, d->F, d->N, d->K);
    if ((d->n_D != 1 && d->n_D != d->K) || (d->n_alph != 1 && d->n_alph != d->K) ||
        (d->n_lam != 1 && d->n_lam != d->K))
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "n_D/n_alph/n_lam must be 1 or K");
    if (d->alph_len != 1 && d->alph_len != d->N)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "alph_len must be 1 or N");
    if (d->operand_f16 != 0 && d->operand_f16 != 1)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "operand_f16 must be 0 or 1");
    if (d->divergence < DRNMF_DIV_ED || d->divergence > DRNMF_DIV_BETA)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "divergence must be DRNMF_DIV_ED, _KL or _BETA");
    if (d->divergence != DRNMF_DIV_ED && d->operand_f16)
        DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED, "the KL / beta cell runs on fp32 operands");
    if ((int64_t)d->B * d->T * (int64_t)d->N * (d->return_all_hidden ? d->K : 1) >= (1ll << 40))
        DRNMF_FAIL(h, DRNMF_ERR_UNSUPPORTED, "output tensor too large");
    return DRNMF_OK;
}

// Dp (cell_b's fp32 packing) -> Dn row-major [Fr][Np] (Fr = round_up(Fp, 32), zero rows behind Fp) and its
// transpose [Np][Fp]; blockIdx.y = stored layer
__global__ void __launch_bounds__(256)
unpack_both_kernel(const float* __restrict__ Dp, float* __restrict__ Dn_rm,
                   float* __restrict__ DnT, int Fp, int Np, int Fr) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)Fr * Np) return;
    const int f = (int)(i / Np), n = (int)(i % Np);
    const size_t lstride = (size_t)Fp * Np;
    Dp += blockIdx.y * lstride;
    float v = 0.f;
    if (f < Fp) {
        v = Dp[((size_t)(f >> 4) * (Np / 16) + (n >> 4)) * 256 + (((n & 15) >> 2) * 16 + (f & 15)) * 4 + (n & 3)];
        DnT[blockIdx.y * lstride + (size_t)n * Fp + f] = v;
    }
    Dn_rm[blockIdx.y * (size_t)Fr * Np + i] = v;
}

struct EpiGramPack {   // G_k[o][i] = sum_f Dn_k[f][o] Dn_k[f][i] -> cell_b operand packing (common.h); split = layer k
    float* Gp;
    int NAC;
    size_t gstride;
    __device__ float pre(int, int, int) const { return 0.f; }
    __device__ void operator()(int k, int o, int i, float acc, float) const {
        Gp[k * gstride + ((size_t)(o >> 4) * NAC + (i >> 4)) * 256 + (((i & 15) >> 2) * 16 + (o & 15)) * 4 +
           (i & 3)] = acc;
    }
};

extern "C" size_t drnmf_params_bytes(const drnmf_cell_desc_t* d) {
    if (!d || d->F <= 0 || d->N <= 0 || d->K <= 0 || d->n_D <= 0) return 0;
    return params_layout(d).total;
}

// One thread per (stored layer, atom column): pass 1 accumulates sum_f exp(log_D)^2 in f order,
// pass 2 writes exp(log_D)/sqrt(sum) -- enhance.py:177-178 / 190-191 -- into the TILE-PACKED
// dictionary (1 KB blocks of 16 bins x 16 atoms; see cell_forward.hip).  Padded rows/columns are
// written as zeros.  HALF = false: fp32 Dp[ft][ac][f%16][n%16].  HALF = true: ONE fp16 packing for
// v_mfma_f32_16x16x32_f16 (a lane's operand = 8 k-slots = 16 bytes; Fp % 32 == 0), 1 KB blocks of 512
// halves, lane l = q*16 + j reads at l*16 bytes:
//   block (f/16, n/32): [q][j = f%16][e] = Dn[f][32 (n/32) + 8q + e]
// Fp*Np halves per layer: cell_b (contracts atoms) takes a lane's 16 bytes as they are; cell_a (contracts
// bins) reads the SAME blocks and transposes them on the way through LDS (ds_read_b64_tr_b16,
// cell_forward.hip) -- a layer's dictionary is fetched from HBM once per layer-step, by cell_b, and the
// cell_a launch behind it finds it in the Infinity Cache.  (Rounds 2-4 kept a second packing for cell_a:
// 1.6 GB instead of 0.8 GB at F = 1025, N = 8000, K = 50, every layer-step reading both from HBM.)  With HALF
// the bins from 16*(F/16) up when F % 16 <= MAX_TAIL (the odd bins handled outside the matrix cores) are ZERO.
// tail[layer][i][n] = Dn[16*(F/16) + i][n] for the (at most MAX_TAIL) bins past the last full tile.
template <bool HALF>
__global__ void __launch_bounds__(256)
prep_dict_kernel(const float* __restrict__ log_D, void* __restrict__ Dn_,
                 float* __restrict__ colnorm, float* __restrict__ tail, float* __restrict__ DnA_,
                 int F, int N, int Fp, int Np, int f_mfma) {
    // workgroup = 64 atoms x 4 interleaved bin groups (a thread per atom walking all F bins twice
    // left a small dictionary -- N = 200, K = 5 -- on five workgroups: 147 us per training step)
    __shared__ float ssum[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int fq = threadIdx.x >> 6;
    const int layer = blockIdx.y;
    const int NAC = Np / 16;
    const size_t lstride = (size_t)Fp * Np;
    // fp32 Dp block (ft, ac): [(q*16 + f%16)*4 + e] = Dn[f][16 ac + 4q + e] -- lane l = q*16 + f%16 of
    // cell_b reads its four atoms at l*16 bytes: consecutive lanes, consecutive 16 bytes
    float* dn = (float*)Dn_ + (size_t)layer * lstride + (size_t)(n >> 4) * 256 +
                ((n & 15) >> 2) * 64 + (n & 3);
    f16* dB = (f16*)Dn_ + (size_t)layer * lstride;
    float* dnA = HALF ? nullptr : DnA_ + (size_t)layer * lstride;
    auto put = [&](int f, float v) {   // (v by value: zeroed past f_mfma in the fp16 packings)
        if (!HALF) {
            dn[(size_t)(f >> 4) * NAC * 256 + (f & 15) * 4] = v;
            // cell_a packing (common.h): block (ft, n/32), s = fi/4, q = fi%4, j = (n%32)/2, a = n%2
            const int fi = f & 15, n32 = n & 31;
            dnA[((size_t)(f >> 4) * (Np / 32) + (n >> 5)) * 512 + (fi >> 3) * 256 +
                ((fi & 3) * 16 + (n32 >> 1)) * 4 + ((fi >> 2) & 1) * 2 + (n32 & 1)] = v;
        } else {
            if (f >= f_mfma) v = 0.f;      // odd bins: outside the matrix cores (tail rows below)
            const int n32 = n & 31;
            // block (f/16, n/32): slot q = n32/8, e = n32%8
            dB[((size_t)(f >> 4) * (Np / 32) + (n >> 5)) * 512 + ((n32 >> 3) * 16 + (f & 15)) * 8 +
               (n32 & 7)] = (f16)v;
        }
    };
    const int t0 = (F / 16) * 16;
    const bool live = n < N, inb = n < Np;   // (Np is a multiple of 32: the last block may overhang)
    float* tl = tail + (size_t)layer * MAX_TAIL * Np + n;
    const float* ld = log_D + (size_t)layer * F * N;
    // (eight of a thread's bins in flight, used in bin order: one dependent load per bin left the
    // N = 200, K = 5 dictionaries of a training step at 47 us)
    float s = 0.f;
    if (live)
        for (int f0 = fq; f0 < F; f0 += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int f = f0 + 4 * u;
                v[u] = ld[(size_t)(f < F ? f : fq) * N + n];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (f0 + 4 * u >= F) break;
                const float e = expf(v[u]);
                s = fmaf(e, e, s);
            }
        }
    ssum[fq][threadIdx.x & 63] = s;
    __syncthreads();
    const int c = threadIdx.x & 63;
    const float nrm = live ? sqrtf((ssum[0][c] + ssum[1][c]) + (ssum[2][c] + ssum[3][c])) : 1.f;
    if (!inb) return;
    if (fq == 0) colnorm[(size_t)layer * Np + n] = nrm;
    for (int f0 = fq; f0 < Fp; f0 += 32) {
        float lv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = f0 + 4 * u;
            lv[u] = (live && f < F) ? ld[(size_t)f * N + n] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = f0 + 4 * u;
            if (f >= Fp) break;
            float v = (live && f < F) ? expf(lv[u]) / nrm : 0.f;
            put(f, v);
            if (HALF) v = (float)(f16)v;
            if (f >= t0 && f - t0 < MAX_TAIL) tl[(size_t)(f - t0) * Np] = v;
        }
    }
    if (fq == 0)
        for (int i = 0; i < MAX_TAIL; ++i)
            if (t0 + i >= Fp) tl[(size_t)i * Np] = 0.f;
}

// 1/alpha[n] and b[n] = -lam/alpha[n] per layer (enhance.py:187-194, 201-203).  Padded atoms get
// a bias of -1e30 so that relu() pins them to exactly zero.
__global__ void __launch_bounds__(256)
prep_scalars_kernel(const float* __restrict__ log_alph, const float* __restrict__ log_lam1,
                    float* __restrict__ inv_alpha, float* __restrict__ bias, int N, int Np,
                    int n_alph, int alph_len, int n_lam) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (n >= Np) return;
    float ia = 0.f, b = -1e30f;
    if (n < N) {
        const float* la = log_alph + (size_t)(n_alph == 1 ? 0 : k) * alph_len;
        const float al = expf(la[alph_len == 1 ? 0 : n]);
        const float lam = expf(log_lam1[n_lam == 1 ? 0 : k]);
        ia = 1.0f / al;
        b = -lam / al;
    }
    inv_alpha[(size_t)k * Np + n] = ia;
    bias[(size_t)k * Np + n] = b;
}

extern "C" int32_t drnmf_prepare_params(drnmf_handle_t h, const drnmf_cell_desc_t* d,
                                        const float* log_D, const float* log_alph,
                                        const float* log_lam1, void* params, void* stream_) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    int rc = validate_cell_desc(h, d);
    if (rc) return rc;
    if (!log_D || !log_alph || !log_lam1 || !params)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "prepare_params: NULL pointer argument");
    hipStream_t stream = (hipStream_t)stream_;
    const ParamsLayout L = params_layout(d);
    char* base = (char*)params;
    dim3 g1((L.Np + 63) / 64, d->n_D);
    // bins from f_mfma up are the odd bins the cell keeps out of the matrix cores (workspace_layout:
    // ntail): zero in the fp16 MFMA packings, present in the tail rows
    const bool has_tail = d->divergence == DRNMF_DIV_ED && d->F % 16 != 0 &&
                          d->F % 16 <= MAX_TAIL && d->F > 16;
    const int f_mfma = has_tail ? (d->F / 16) * 16 : L.Fp;
    if (d->operand_f16) {
        // the fp32 packings first (they write the unrounded tail rows, which the fp16 pass then
        // replaces by the fp16-rounded values the forward kernels use)
        float* dn32 = (float*)(base + L.off_dn32);
        hipLaunchKernelGGL(prep_dict_kernel<false>, g1, dim3(256), 0, stream, log_D, (void*)dn32,
                           (float*)(base + L.off_colnorm), (float*)(base + L.off_tail),
                           dn32 + (size_t)d->n_D * L.Fp * L.Np, d->F, d->N, L.Fp, L.Np, f_mfma);
        hipLaunchKernelGGL(prep_dict_kernel<true>, g1, dim3(256), 0, stream, log_D,
                           (void*)(base + L.off_dn), (float*)(base + L.off_colnorm),
                           (float*)(base + L.off_tail), (float*)nullptr, d->F, d->N, L.Fp, L.Np,
                           f_mfma);
    } else
        hipLaunchKernelGGL(prep_dict_kernel<false>, g1, dim3(256), 0, stream, log_D,
                           (void*)(base + L.off_dn), (float*)(base + L.off_colnorm),
                           (float*)(base + L.off_tail), (float*)(base + L.off_dnA), d->F, d->N,
                           L.Fp, L.Np, f_mfma);
    dim3 g2((L.Np + 255) / 256, d->K);
    hipLaunchKernelGGL(prep_scalars_kernel, g2, dim3(256), 0, stream, log_alph, log_lam1,
                       (float*)(base + L.off_inv_alpha), (float*)(base + L.off_bias), d->N, L.Np,
                       d->n_alph, d->alph_len, d->n_lam);
    DRNMF_HIP(h, hipGetLastError());
    if (gram_eligible(d)) {
        // G_k = Dn_k^T Dn_k per stored layer (2 N^2 F flops each) + the transposed dictionary: ONE TN
        // product over the layers stacked along the contraction, its split s = the k-tiles of layer s
        // (a launch pair per layer was 5 x 31 us of a 9.5-ms training step at N = 200)
        const int Fr = round_up(L.Fp, 32);
        float* Dn_rm = (float*)(base + L.off_dn_rm);
        hipLaunchKernelGGL(unpack_both_kernel, dim3((unsigned)(((size_t)Fr * L.Np + 255) / 256), d->n_D),
                           dim3(256), 0, stream, (const float*)(base + L.off_dn), Dn_rm,
                           (float*)(base + L.off_dnT), L.Fp, L.Np, Fr);
        gemm_tn::Operands g{Dn_rm, Dn_rm, (int64_t)d->n_D * Fr, L.Np, L.Np, L.Np, L.Np};
        EpiGramPack epi{(float*)(base + L.off_gram), L.Np / 16, (size_t)L.Np * L.Np};
        DRNMF_HIP(h, gemm_tn::launch(g, epi, d->n_D, stream));
    }
    return DRNMF_OK;
}
