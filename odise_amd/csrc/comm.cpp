// comm.cpp — the one exchange step of the data-parallel inference path: an RCCL all-gather of fixed-size per-image
// prediction records over xGMI, owned by this library (no torch device tensors, no torch NCCL process group).
//
// Replaces detectron2's pickled `comm.gather` of per-image predictions reached from the evaluation loop
// (/root/reference odise/evaluation/evaluator.py:144 -> d2 DatasetEvaluator.evaluate) and, for semantic evaluation, the gather of
// confusion matrices (odise/evaluation/d2_evaluator.py:63): both become one collective on device buffers.
//
// Process model: one process per GPU (torch.distributed.run / d2 launch); the Python launcher obtains the 128-byte unique id on
// rank 0 (odise_hip_comm_unique_id), broadcasts it over its CPU rendezvous (gloo / TCPStore) and every rank calls
// odise_hip_comm_init.  Collectives run on a SECOND HIP stream of the context: the exchange of step i waits (event) for step i's
// kernels only, so step i+1's kernels overlap it; odise_hip_comm_wait joins the host.  A world of one rank takes the same code path
// (RCCL communicator of size 1), which is what bench.py --gpus 1 and the single-GPU tests exercise.
//
// RCCL is bound at run time (dlopen of librccl.so.1): libodise_hip.so itself has no link-time dependency on it, single-GPU users
// who never call odise_hip_comm_* do not need it in their library path.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "common.h"

namespace odise {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl g_rccl;

static int rccl_load() {
    if (g_rccl.handle) return ODISE_OK;
    // An RCCL that is ALREADY in the process wins: with torch imported first the process runs torch's bundled HIP runtime (same SONAME as
    // /opt/rocm's) and torch's own librccl.so is loaded with it - the communicator must come from the RCCL that was
    // built against the runtime it runs on, not from a second copy out of the system directory.
    const char* names[] = {getenv("ODISE_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    if (!(names[0] && *names[0])) {
        for (const char* n : {"librccl.so", "librccl.so.1"}) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (h) break;
        }
    }
    for (const char* n : names) {
        if (h) break;
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) {
        set_error("comm: cannot load librccl.so.1 (%s); set ODISE_RCCL_LIB or add the ROCm lib directory to LD_LIBRARY_PATH", dlerror());
        return ODISE_ERR_STATE;
    }
    Rccl r;
    r.handle = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce || !r.GetErrorString) {
        set_error("comm: librccl does not export the expected nccl* entry points");
        dlclose(h);
        return ODISE_ERR_STATE;
    }
    g_rccl = r;
    return ODISE_OK;
}

#define ODISE_CHECK_NCCL(expr)                                                                                  \
    do {                                                                                                        \
        ncclResult_t _r = (expr);                                                                               \
        if (_r != ncclSuccess) {                                                                                \
            ::odise::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));   \
            return ODISE_ERR_HIP;                                                                               \
        }                                                                                                       \
    } while (0)

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;   // the exchange stream
    hipEvent_t produced = nullptr;  // compute stream -> exchange stream
    hipEvent_t done = nullptr;      // exchange stream -> host / compute stream
};

void comm_release(odise_hip_ctx* ctx) {
    Comm* c = (Comm*)ctx->comm;
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->produced) (void)hipEventDestroy(c->produced);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    ctx->comm = nullptr;
}

}  // namespace odise

using namespace odise;

extern "C" int odise_hip_comm_unique_id(void* id128) {
    ODISE_REQUIRE(id128, "comm_unique_id: null buffer");
    if (int rc = rccl_load()) return rc;
    ncclUniqueId id;
    ODISE_CHECK_NCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == ODISE_COMM_ID_BYTES, "unique id size");
    memcpy(id128, &id, sizeof(id));
    return ODISE_OK;
}

static int comm_init_impl(odise_hip_ctx* ctx, const void* id128, int rank, int world) {
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));
    Comm* c = new Comm();
    ctx->comm = c;   // owned by the context from here on: the caller releases it on any failure below
    c->rank = rank;
    c->world = world;
    ODISE_CHECK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    ODISE_CHECK_HIP(hipEventCreateWithFlags(&c->produced, hipEventDisableTiming));
    ODISE_CHECK_HIP(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ODISE_CHECK_NCCL(g_rccl.CommInitRank(&c->comm, world, id, rank));
    return ODISE_OK;
}

extern "C" int odise_hip_comm_init(odise_hip_ctx* ctx, const void* id128, int rank, int world) {
    ODISE_REQUIRE(ctx && id128, "comm_init: null argument");
    ODISE_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d / world %d", rank, world);
    if (int rc = rccl_load()) return rc;
    comm_release(ctx);
    const int rc = comm_init_impl(ctx, id128, rank, world);
    if (rc != ODISE_OK) comm_release(ctx);   // never leave a half-built communicator behind: comm_of() only tests ctx->comm
    return rc;
}

extern "C" int odise_hip_comm_destroy(odise_hip_ctx* ctx) {
    ODISE_REQUIRE(ctx, "comm_destroy: null context");
    comm_release(ctx);
    return ODISE_OK;
}

extern "C" int odise_hip_comm_info(odise_hip_ctx* ctx, int* rank, int* world) {
    ODISE_REQUIRE(ctx, "comm_info: null context");
    Comm* c = (Comm*)ctx->comm;
    if (rank) *rank = c ? c->rank : 0;
    if (world) *world = c ? c->world : 0;
    return ODISE_OK;
}

static int comm_of(odise_hip_ctx* ctx, Comm** out, const char* who) {
    ODISE_REQUIRE(ctx, "%s: null context", who);
    *out = (Comm*)ctx->comm;
    if (!*out) {
        set_error("%s: call odise_hip_comm_init first", who);
        return ODISE_ERR_STATE;
    }
    ODISE_CHECK_HIP(hipSetDevice(ctx->device));   // callers may be worker threads
    return ODISE_OK;
}

extern "C" int odise_hip_allgather_predictions(odise_hip_ctx* ctx, const int32_t* local, int64_t count, int32_t* all) {
    Comm* c = nullptr;
    if (int rc = comm_of(ctx, &c, "allgather_predictions")) return rc;
    ODISE_REQUIRE(local && all && count > 0, "allgather_predictions: bad argument");
    ODISE_CHECK_HIP(hipEventRecord(c->produced, ctx->stream));          // everything the compute stream has queued so far
    ODISE_CHECK_HIP(hipStreamWaitEvent(c->stream, c->produced, 0));
    ODISE_CHECK_NCCL(g_rccl.AllGather(local, all, (size_t)count, ncclInt32, c->comm, c->stream));
    ODISE_CHECK_HIP(hipEventRecord(c->done, c->stream));
    return ODISE_OK;
}

// Uneven shards (the last batch of a dataset whose size is not a multiple of world x batch: InferenceSampler's shards differ by one image,
// odise/data/build.py:145-151): every rank contributes n_records <= max_records records, max_records identical on every rank (each rank derives
// it from the dataset size, distributed.shard_range - no size exchange).  This rank's slice of `all` is assembled on the exchange stream - its
// records, then -1 rows up to max_records - and gathered in place.
extern "C" int odise_hip_allgather_records(odise_hip_ctx* ctx, const int32_t* local, int n_records, int max_records, int64_t record_len, int32_t* all) {
    Comm* c = nullptr;
    if (int rc = comm_of(ctx, &c, "allgather_records")) return rc;
    ODISE_REQUIRE(all && record_len > 0 && max_records >= 1 && n_records >= 0 && n_records <= max_records && (local || n_records == 0),
                  "allgather_records: bad argument (n_records %d, max_records %d)", n_records, max_records);
    const size_t slice = (size_t)max_records * record_len;
    int32_t* mine = all + (size_t)c->rank * slice;
    ODISE_CHECK_HIP(hipEventRecord(c->produced, ctx->stream));
    ODISE_CHECK_HIP(hipStreamWaitEvent(c->stream, c->produced, 0));
    if (n_records > 0 && local != mine)
        ODISE_CHECK_HIP(hipMemcpyAsync(mine, local, (size_t)n_records * record_len * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
    if (n_records < max_records)
        ODISE_CHECK_HIP(hipMemsetAsync(mine + (size_t)n_records * record_len, 0xFF, (size_t)(max_records - n_records) * record_len * sizeof(int32_t), c->stream));
    ODISE_CHECK_NCCL(g_rccl.AllGather(mine, all, slice, ncclInt32, c->comm, c->stream));   // in place: sendbuff = recvbuff + rank * count
    ODISE_CHECK_HIP(hipEventRecord(c->done, c->stream));
    return ODISE_OK;
}

extern "C" int odise_hip_allreduce_sum_i64(odise_hip_ctx* ctx, int64_t* data, int64_t count) {
    Comm* c = nullptr;
    if (int rc = comm_of(ctx, &c, "allreduce_sum_i64")) return rc;
    ODISE_REQUIRE(data && count > 0, "allreduce_sum_i64: bad argument");
    ODISE_CHECK_HIP(hipEventRecord(c->produced, ctx->stream));
    ODISE_CHECK_HIP(hipStreamWaitEvent(c->stream, c->produced, 0));
    ODISE_CHECK_NCCL(g_rccl.AllReduce(data, data, (size_t)count, ncclInt64, ncclSum, c->comm, c->stream));
    ODISE_CHECK_HIP(hipEventRecord(c->done, c->stream));
    return ODISE_OK;
}

extern "C" int odise_hip_comm_wait(odise_hip_ctx* ctx, int block_host) {
    Comm* c = nullptr;
    if (int rc = comm_of(ctx, &c, "comm_wait")) return rc;
    if (block_host) ODISE_CHECK_HIP(hipEventSynchronize(c->done));
    else ODISE_CHECK_HIP(hipStreamWaitEvent(ctx->stream, c->done, 0));  // later compute-stream work may overwrite the gathered buffers
    return ODISE_OK;
}
