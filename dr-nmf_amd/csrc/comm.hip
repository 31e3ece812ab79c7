// Data-parallel collective behind the C ABI: one RCCL communicator per handle (one rank per GPU /
// process), used for the ONE all-reduce per optimiser step of the flat gradient buffer and for the
// parameter broadcast that makes the replicas identical (the reference
// itself is single-device, enhance.py:579).
//
// librccl is resolved at run time (dlopen by SONAME): a process that already holds RCCL -- a
// PyTorch-ROCm host ships its own copy next to its HIP runtime -- must keep using that one, and a
// plain C host gets /opt/rocm's.  Only the C entry points below are used; their prototypes are
// RCCL's public ones (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclAllReduce, ncclBroadcast,
// ncclCommDestroy, ncclGetErrorString).
#include <dlfcn.h>

#include "common.h"

namespace {

struct UniqueId { char internal[DRNMF_COMM_ID_BYTES]; };   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*get_unique_id_fn)(UniqueId*);
typedef int (*comm_init_rank_fn)(void**, int, UniqueId, int);
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*broadcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*comm_destroy_fn)(void*);
typedef const char* (*error_string_fn)(int);
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0;   // ncclDataType_t / ncclRedOp_t values of rccl.h

struct Rccl {
    void* lib = nullptr;
    get_unique_id_fn get_unique_id = nullptr;
    comm_init_rank_fn comm_init_rank = nullptr;
    all_reduce_fn all_reduce = nullptr;
    broadcast_fn broadcast = nullptr;
    comm_destroy_fn comm_destroy = nullptr;
    error_string_fn error_string = nullptr;
    char why[256] = {0};
};

void fill_rccl(Rccl& R) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        R.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (R.lib) break;
    }
    if (!R.lib) {
        snprintf(R.why, sizeof(R.why), "librccl.so.1 not loadable: %s", dlerror());
        return;
    }
    R.get_unique_id = (get_unique_id_fn)dlsym(R.lib, "ncclGetUniqueId");
    R.comm_init_rank = (comm_init_rank_fn)dlsym(R.lib, "ncclCommInitRank");
    R.all_reduce = (all_reduce_fn)dlsym(R.lib, "ncclAllReduce");
    R.broadcast = (broadcast_fn)dlsym(R.lib, "ncclBroadcast");
    R.comm_destroy = (comm_destroy_fn)dlsym(R.lib, "ncclCommDestroy");
    R.error_string = (error_string_fn)dlsym(R.lib, "ncclGetErrorString");
    if (!R.get_unique_id || !R.comm_init_rank || !R.all_reduce || !R.broadcast ||
        !R.comm_destroy || !R.error_string) {
        snprintf(R.why, sizeof(R.why), "librccl lacks an expected nccl* entry point");
        R.lib = nullptr;
    }
}

// Lazy, once: a function-local static initialised by a lambda -- the language runs it on exactly
// one thread and makes the others wait, so two host threads making their first comm call never see
// a half-filled table.
Rccl* rccl() {
    static Rccl R;
    static const bool once = [] { fill_rccl(R); return true; }();
    (void)once;
    return &R;
}

#define DRNMF_RCCL(h, R, expr)                                                            \
    do {                                                                                  \
        const int r_ = (expr);                                                            \
        if (r_ != 0)                                                                      \
            DRNMF_FAIL(h, DRNMF_ERR_RCCL, "%s failed: %s", #expr, (R)->error_string(r_)); \
    } while (0)

}  // namespace

extern "C" int32_t drnmf_comm_unique_id(drnmf_handle_t h, void* id_out_host) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (!id_out_host) DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "comm_unique_id: NULL id buffer");
    Rccl* R = rccl();
    if (!R->lib) DRNMF_FAIL(h, DRNMF_ERR_RCCL, "%s", R->why);
    UniqueId id;
    DRNMF_RCCL(h, R, R->get_unique_id(&id));
    memcpy(id_out_host, &id, sizeof(id));
    return DRNMF_OK;
}

extern "C" int32_t drnmf_comm_init(drnmf_handle_t h, const void* id_host, int32_t rank,
                                   int32_t world) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (!id_host || world < 1 || rank < 0 || rank >= world)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "comm_init: need an id and 0 <= rank (%d) < world (%d)",
                   rank, world);
    if (h->comm) DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "comm_init: the handle already owns a communicator");
    Rccl* R = rccl();
    if (!R->lib) DRNMF_FAIL(h, DRNMF_ERR_RCCL, "%s", R->why);
    DRNMF_HIP(h, hipSetDevice(h->device));
    UniqueId id;
    memcpy(&id, id_host, sizeof(id));
    void* comm = nullptr;
    DRNMF_RCCL(h, R, R->comm_init_rank(&comm, world, id, rank));
    h->comm = comm;
    h->comm_rank = rank;
    h->comm_world = world;
    return DRNMF_OK;
}

extern "C" int32_t drnmf_comm_destroy(drnmf_handle_t h) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (!h->comm) return DRNMF_OK;
    Rccl* R = rccl();
    void* comm = h->comm;
    h->comm = nullptr;
    h->comm_rank = 0;
    h->comm_world = 1;
    if (R->lib) DRNMF_RCCL(h, R, R->comm_destroy(comm));
    return DRNMF_OK;
}

extern "C" int32_t drnmf_comm_info(drnmf_handle_t h, int32_t* rank, int32_t* world) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (rank) *rank = h->comm_rank;
    if (world) *world = h->comm ? h->comm_world : 1;
    return DRNMF_OK;
}

extern "C" int32_t drnmf_allreduce_grads(drnmf_handle_t h, float* flat, int64_t n, void* stream) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (!flat || n <= 0) DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "allreduce_grads: empty buffer");
    if (!h->comm)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "allreduce_grads: no communicator (drnmf_comm_init)");
    Rccl* R = rccl();
    DRNMF_RCCL(h, R, R->all_reduce(flat, flat, (size_t)n, NCCL_FLOAT32, NCCL_SUM, h->comm,
                                   (hipStream_t)stream));
    return DRNMF_OK;
}

extern "C" int32_t drnmf_broadcast_params(drnmf_handle_t h, float* buf, int64_t n, int32_t root,
                                          void* stream) {
    DRNMF_LOCK(h);
    if (!h) return DRNMF_ERR_INVALID_ARG;
    if (!buf || n <= 0) DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "broadcast_params: empty buffer");
    if (!h->comm)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "broadcast_params: no communicator (drnmf_comm_init)");
    if (root < 0 || root >= h->comm_world)
        DRNMF_FAIL(h, DRNMF_ERR_INVALID_ARG, "broadcast_params: root %d outside [0,%d)", root,
                   h->comm_world);
    Rccl* R = rccl();
    DRNMF_RCCL(h, R, R->broadcast(buf, buf, (size_t)n, NCCL_FLOAT32, root, h->comm,
                                  (hipStream_t)stream));
    return DRNMF_OK;
}
