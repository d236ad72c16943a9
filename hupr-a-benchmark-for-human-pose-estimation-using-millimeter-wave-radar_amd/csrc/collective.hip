// (e) Data-parallel exchange step over RCCL / xGMI — the one collective of the path (the reference has no distributed
// layer at all: single device string, tools/base.py:14).
//
// librccl is bound at run time (dlopen + dlsym), not at link time: the process that hosts this library normally has
// PyTorch's own copy of librccl.so.1 mapped already, and two different RCCL images in one process would each bring
// their own topology / IPC state.  hupr_comm_load() therefore first asks the loader for the copy that is already
// resident (RTLD_NOLOAD on the soname) and only then for a fresh one (explicit path, or the soname through the normal
// search path, e.g. /opt/rocm/lib).
//
// Stream semantics: hupr_allreduce_bucket / hupr_broadcast_bucket enqueue on `stream` and return; they are legal
// inside a hipGraph capture of that stream (RCCL collectives are capturable), which is how the engine replays a whole
// data-parallel training step as one graph.
#include <dlfcn.h>
#include <string.h>

#include "hupr_common.h"

namespace {

// The slice of rccl.h this file needs (values per /opt/rocm/include/rccl/rccl.h:40-43,448-468).
typedef struct { char internal[HUPR_COMM_ID_BYTES]; } rccl_unique_id;
typedef void* rccl_comm_t;
enum { RCCL_SUM = 0, RCCL_FLOAT32 = 7, RCCL_BFLOAT16 = 9 };

struct Api {
    void* handle = nullptr;
    int (*get_unique_id)(rccl_unique_id*) = nullptr;
    int (*comm_init_rank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
    int (*comm_destroy)(rccl_comm_t) = nullptr;
    int (*comm_count)(const rccl_comm_t, int*) = nullptr;
    int (*comm_user_rank)(const rccl_comm_t, int*) = nullptr;
    int (*all_reduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*broadcast)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    const char* (*error_string)(int) = nullptr;
};
Api g_api;

template <typename F> bool bind(void* h, const char* name, F& out) {
    out = reinterpret_cast<F>(dlsym(h, name));
    return out != nullptr;
}

int ensure_loaded() {
    if (g_api.handle) return HUPR_OK;
    return hupr_comm_load(nullptr);
}

int rccl_fail(const char* what, int rc) {
    return hupr::fail(HUPR_ERR_COMM, "%s: %s (ncclResult %d)", what,
                      g_api.error_string ? g_api.error_string(rc) : "?", rc);
}

}  // namespace

extern "C" int hupr_comm_load(const char* path_or_null) {
    if (g_api.handle) return HUPR_OK;
    void* h = nullptr;
    if (path_or_null && path_or_null[0]) h = dlopen(path_or_null, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return hupr::fail(HUPR_ERR_COMM, "hupr_comm_load: cannot load librccl.so.1: %s", dlerror());
    Api a;
    a.handle = h;
    if (!bind(h, "ncclGetUniqueId", a.get_unique_id) || !bind(h, "ncclCommInitRank", a.comm_init_rank) ||
        !bind(h, "ncclCommDestroy", a.comm_destroy) ||
        !bind(h, "ncclCommCount", a.comm_count) || !bind(h, "ncclCommUserRank", a.comm_user_rank) || !bind(h, "ncclAllReduce", a.all_reduce) ||
        !bind(h, "ncclBroadcast", a.broadcast) || !bind(h, "ncclGetErrorString", a.error_string))
        return hupr::fail(HUPR_ERR_COMM, "hupr_comm_load: librccl lacks a required symbol: %s", dlerror());
    g_api = a;
    return HUPR_OK;
}

extern "C" int hupr_comm_unique_id(void* id_out) {
    HUPR_REQUIRE(id_out != nullptr, "hupr_comm_unique_id: null output");
    if (int rc = ensure_loaded()) return rc;
    rccl_unique_id id;
    if (int rc = g_api.get_unique_id(&id)) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id_out, id.internal, HUPR_COMM_ID_BYTES);
    return HUPR_OK;
}

extern "C" int hupr_comm_init_rank(hupr_comm_t* comm_out, const void* id, int n_ranks, int rank) {
    HUPR_REQUIRE(comm_out != nullptr && id != nullptr, "hupr_comm_init_rank: null pointer");
    HUPR_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "hupr_comm_init_rank: rank %d of %d", rank, n_ranks);
    if (int rc = ensure_loaded()) return rc;
    rccl_unique_id uid;
    memcpy(uid.internal, id, HUPR_COMM_ID_BYTES);
    rccl_comm_t c = nullptr;
    if (int rc = g_api.comm_init_rank(&c, n_ranks, uid, rank)) return rccl_fail("ncclCommInitRank", rc);
    *comm_out = c;
    return HUPR_OK;
}

extern "C" int hupr_comm_destroy(hupr_comm_t comm) {
    if (comm == nullptr) return HUPR_OK;
    if (int rc = ensure_loaded()) return rc;
    if (int rc = g_api.comm_destroy(comm)) return rccl_fail("ncclCommDestroy", rc);
    return HUPR_OK;
}

extern "C" int hupr_comm_info(hupr_comm_t comm, int* n_ranks_out, int* rank_out) {
    HUPR_REQUIRE(comm != nullptr && n_ranks_out != nullptr && rank_out != nullptr, "hupr_comm_info: null pointer");
    if (int rc = ensure_loaded()) return rc;
    if (int rc = g_api.comm_count(comm, n_ranks_out)) return rccl_fail("ncclCommCount", rc);
    if (int rc = g_api.comm_user_rank(comm, rank_out)) return rccl_fail("ncclCommUserRank", rc);
    return HUPR_OK;
}

extern "C" int hupr_allreduce_bucket(hupr_comm_t comm, void* bucket, size_t count, int dtype, hupr_stream_t stream) {
    HUPR_REQUIRE(comm != nullptr, "hupr_allreduce_bucket: null communicator");
    HUPR_REQUIRE(dtype == HUPR_COMM_F32 || dtype == HUPR_COMM_BF16, "hupr_allreduce_bucket: dtype %d", dtype);
    if (count == 0) return HUPR_OK;
    HUPR_REQUIRE(bucket != nullptr, "hupr_allreduce_bucket: null bucket");
    if (int rc = ensure_loaded()) return rc;
    const int dt = dtype == HUPR_COMM_F32 ? RCCL_FLOAT32 : RCCL_BFLOAT16;
    if (int rc = g_api.all_reduce(bucket, bucket, count, dt, RCCL_SUM, comm, hupr::as_stream(stream)))
        return rccl_fail("ncclAllReduce", rc);
    return HUPR_OK;
}

extern "C" int hupr_broadcast_bucket(hupr_comm_t comm, void* bucket, size_t count, int dtype, int root,
                                     hupr_stream_t stream) {
    HUPR_REQUIRE(comm != nullptr, "hupr_broadcast_bucket: null communicator");
    HUPR_REQUIRE(dtype == HUPR_COMM_F32 || dtype == HUPR_COMM_BF16, "hupr_broadcast_bucket: dtype %d", dtype);
    if (count == 0) return HUPR_OK;
    HUPR_REQUIRE(bucket != nullptr && root >= 0, "hupr_broadcast_bucket: bad argument");
    if (int rc = ensure_loaded()) return rc;
    const int dt = dtype == HUPR_COMM_F32 ? RCCL_FLOAT32 : RCCL_BFLOAT16;
    if (int rc = g_api.broadcast(bucket, bucket, count, dt, root, comm, hupr::as_stream(stream)))
        return rccl_fail("ncclBroadcast", rc);
    return HUPR_OK;
}
