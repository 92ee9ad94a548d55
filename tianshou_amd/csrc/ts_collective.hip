// ts_collective.hip -- the data-parallel exchange of the update step: in-place sum all-reduce of a flat fp32 buffer
// over RCCL (xGMI inside a node), one communicator per process (one process per GPU).
//
// The reference has no distributed path (its only multi-GPU mechanism is single-process nn.DataParallel,
// tianshou/utils/net/common.py:473-515); SURVEY 8(b)/(e) make the all-reduce part of the boundary so that a host in
// any language can drive the data-parallel update without torch.distributed: rank 0 calls ts_allreduce_unique_id,
// ships the 128 bytes to the other ranks through whatever rendezvous the host already has, every rank calls
// ts_allreduce_init, and then ts_allreduce on the stream the gradient kernels run on (no host sync: the collective
// is ordered after ts_ppo_grad and before ts_ppo_apply by the stream).
//
// RCCL is resolved at run time (dlopen): libtsengine.so itself stays loadable on hosts without RCCL, and inside a
// PyTorch process the already-loaded librccl is reused instead of mapping a second copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "ts_common.h"

struct ts_comm {
    ncclComm_t comm;
    int rank, world, device;
};

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    const char* (*get_error_string)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names)          // a copy the process already mapped (PyTorch ships its own)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    Rccl r;
    r.handle = h;
    r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(h, "ncclAllReduce"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    r.get_error_string = reinterpret_cast<decltype(r.get_error_string)>(dlsym(h, "ncclGetErrorString"));
    if (r.get_unique_id && r.comm_init_rank && r.all_reduce && r.comm_destroy) g_rccl = r;
}

int need_rccl() {
    std::call_once(g_once, load_rccl);
    TS_REQUIRE(g_rccl.handle != nullptr, TS_ERR_UNSUPPORTED, "RCCL (librccl.so.1) could not be loaded: %s", dlerror());
    return TS_OK;
}

int rccl_fail(const char* what, ncclResult_t r) {
    return ts::fail(TS_ERR_HIP, "%s failed: %s", what, g_rccl.get_error_string ? g_rccl.get_error_string(r) : "RCCL error");
}

}  // namespace

extern "C" {

int ts_allreduce_unique_id(uint8_t* h_id128) {
    static_assert(sizeof(ncclUniqueId) == 128, "the boundary ships the RCCL unique id as 128 bytes");
    TS_REQUIRE(h_id128 != nullptr, TS_ERR_INVALID_ARG, "ts_allreduce_unique_id: NULL output");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.get_unique_id(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    std::memcpy(h_id128, &id, sizeof(id));
    return TS_OK;
}

int ts_allreduce_init(const uint8_t* h_id128, int64_t rank, int64_t world, int device, ts_comm** out) {
    TS_REQUIRE(h_id128 && out, TS_ERR_INVALID_ARG, "ts_allreduce_init: NULL argument");
    TS_REQUIRE(world >= 1 && rank >= 0 && rank < world, TS_ERR_INVALID_ARG, "ts_allreduce_init: rank %lld of %lld",
               (long long)rank, (long long)world);
    if (int rc = need_rccl()) return rc;
    TS_HIP_CHECK(hipSetDevice(device));
    ncclUniqueId id;
    std::memcpy(&id, h_id128, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.comm_init_rank(&comm, (int)world, id, (int)rank);
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    *out = new ts_comm{comm, (int)rank, (int)world, device};
    return TS_OK;
}

int ts_allreduce(ts_comm* comm, float* buf, int64_t n, ts_stream_t stream) {
    TS_REQUIRE(comm != nullptr, TS_ERR_INVALID_ARG, "ts_allreduce: communicator is NULL");
    TS_REQUIRE(n >= 0 && (buf != nullptr || n == 0), TS_ERR_INVALID_ARG, "ts_allreduce: bad buffer");
    if (n == 0) return TS_OK;
    const ncclResult_t r = g_rccl.all_reduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, comm->comm, ts::as_stream(stream));
    if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
    return TS_OK;
}

int ts_allreduce_destroy(ts_comm* comm) {
    if (!comm) return TS_OK;
    if (g_rccl.comm_destroy) g_rccl.comm_destroy(comm->comm);
    delete comm;
    return TS_OK;
}

}  // extern "C"
