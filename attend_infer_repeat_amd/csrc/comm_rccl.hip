// RCCL gradient all-reduce as a C-ABI entry point that can be CAPTURED into the step's hipGraph.
//
// The reference is single-process; the data-parallel path of this package shards the batch over one process
// per GPU and sums ONE flat fp32 gradient bucket per step.  Round 1 issued that collective through
// torch.distributed between two graphs (forward+backward | update); here the engine's stream calls ncclAllReduce itself,
// so with world > 1 the whole step -- noise, forward, backward, all-reduce over xGMI, both RMSProp updates -- is still one
// graph replay with no host round trip.
//
// RCCL is bound at run time (dlopen / dlsym) against the instance the process already holds -- torch imports its own
// librccl.so next to its own libamdhip64.so, and a second copy of either would not share streams or IPC handles -- so the
// package has no build-time dependency on RCCL headers and single-GPU use never touches the library.
#include <dlfcn.h>
#include <string.h>
#include "air_common.h"

namespace {
typedef struct { char internal[128]; } nccl_unique_id;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_get_unique_id)(nccl_unique_id *);
typedef int (*fn_comm_init_rank)(void **, int, nccl_unique_id, int);
typedef int (*fn_comm_destroy)(void *);
typedef int (*fn_comm_count)(void *, int *);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*fn_error_string)(int);
enum { NCCL_FLOAT32 = 7, NCCL_SUM = 0 };                      // ncclDataType_t / ncclRedOp_t values of rccl.h

struct RcclApi {
    void *handle;
    fn_get_unique_id get_unique_id;
    fn_comm_init_rank comm_init_rank;
    fn_comm_destroy comm_destroy;
    fn_comm_count comm_count;
    fn_all_reduce all_reduce;
    fn_error_string error_string;
};
RcclApi g_api = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
char g_last_error[256] = "";

void set_error(const char *what, int code) {
    const char *msg = (g_api.error_string && code > 0) ? g_api.error_string(code) : "";
    snprintf(g_last_error, sizeof(g_last_error), "%s (rccl result %d%s%s)", what, code, msg[0] ? ": " : "", msg);
}

int bind_rccl() {
    if (g_api.handle) return AIR_OK;
    const char *names[] = {"librccl.so", "librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }      // the instance already in the process
    if (!h) for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { snprintf(g_last_error, sizeof(g_last_error), "librccl.so not found: %s", dlerror()); return AIR_E_UNSUPPORTED; }
    RcclApi a;
    a.handle = h;
    a.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    a.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    a.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    a.comm_count = (fn_comm_count)dlsym(h, "ncclCommCount");
    a.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    a.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.comm_count || !a.all_reduce) {
        snprintf(g_last_error, sizeof(g_last_error), "librccl.so lacks an expected symbol");
        return AIR_E_UNSUPPORTED;
    }
    g_api = a;
    return AIR_OK;
}
}  // namespace

extern "C" const char *air_comm_last_error(void) { return g_last_error; }

// 128 opaque bytes that rank 0 creates and every rank passes to air_comm_init (exchange them by any means: the Python side
// uses torch.distributed.broadcast_object_list on whatever backend is up)
extern "C" int air_comm_unique_id(void *id_out_128_bytes) {
    AIR_REQUIRE(id_out_128_bytes, AIR_E_NULL);
    int st = bind_rccl();
    if (st) return st;
    nccl_unique_id id;
    const int r = g_api.get_unique_id(&id);
    if (r != 0) { set_error("ncclGetUniqueId failed", r); return AIR_E_UNSUPPORTED; }
    memcpy(id_out_128_bytes, &id, sizeof(id));
    return AIR_OK;
}

// collective call on every rank; the calling thread's current HIP device is the rank's GPU
extern "C" int air_comm_init(void **comm_out, int world_size, int rank, const void *id_128_bytes) {
    AIR_REQUIRE(comm_out && id_128_bytes, AIR_E_NULL);
    AIR_REQUIRE(world_size > 0 && rank >= 0 && rank < world_size, AIR_E_SHAPE);
    int st = bind_rccl();
    if (st) return st;
    nccl_unique_id id;
    memcpy(&id, id_128_bytes, sizeof(id));
    void *comm = nullptr;
    const int r = g_api.comm_init_rank(&comm, world_size, id, rank);
    if (r != 0) { set_error("ncclCommInitRank failed", r); return AIR_E_UNSUPPORTED; }
    *comm_out = comm;
    return AIR_OK;
}

// 0 when the library can be bound in this process (dlopen + the symbols above); nothing collective happens here, so every
// rank can call it and AGREE on the outcome before any of them enters the blocking air_comm_init
extern "C" int air_comm_available(void) { return bind_rccl(); }

// number of ranks the communicator spans, as RCCL itself reports it (ncclCommCount)
extern "C" int air_comm_count(void *comm, int *count_out) {
    AIR_REQUIRE(comm && count_out, AIR_E_NULL);
    if (!g_api.handle) return AIR_E_UNSUPPORTED;
    const int r = g_api.comm_count(comm, count_out);
    if (r != 0) { set_error("ncclCommCount failed", r); return AIR_E_UNSUPPORTED; }
    return AIR_OK;
}

extern "C" int air_comm_destroy(void *comm) {
    if (!comm) return AIR_OK;
    if (!g_api.handle) return AIR_E_UNSUPPORTED;
    const int r = g_api.comm_destroy(comm);
    if (r != 0) { set_error("ncclCommDestroy failed", r); return AIR_E_UNSUPPORTED; }
    return AIR_OK;
}

// buf[0:n] <- sum over ranks of buf[0:n] (fp32, in place) on `stream`.  No allocation, no host synchronisation: legal
// between air_graph_begin_capture and air_graph_end_capture, where it becomes a node of the step's graph.
extern "C" int air_allreduce_sum(float *buf, size_t n, void *comm, void *stream) {
    AIR_REQUIRE(buf && comm, AIR_E_NULL);
    AIR_REQUIRE(n > 0, AIR_E_SHAPE);
    if (!g_api.handle) return AIR_E_UNSUPPORTED;
    const int r = g_api.all_reduce(buf, buf, n, NCCL_FLOAT32, NCCL_SUM, comm, air_stream(stream));
    if (r != 0) { set_error("ncclAllReduce failed", r); return AIR_E_UNSUPPORTED; }
    return AIR_OK;
}

// join `stream` to whatever `event` marks (hipStreamWaitEvent): forks / joins a side stream inside a capture, so that a
// gradient slice that is final early can be all-reduced while the rest of the backward still runs
extern "C" int air_stream_wait_event(void *stream, void *event) {
    AIR_REQUIRE(event, AIR_E_NULL);
    return (int)hipStreamWaitEvent(air_stream(stream), (hipEvent_t)event, 0);
}
