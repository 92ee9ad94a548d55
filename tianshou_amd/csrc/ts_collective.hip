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

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "ts_common.h"

struct ts_small_comm;
struct ts_comm {
    ncclComm_t comm;                   // NULL: no RCCL communicator (only the one-shot path below is available)
    int rank, world, device;
    ts_small_comm* small;              // optional one-shot path for payloads up to its capacity (not owned)
};
extern "C" int ts_allreduce_small(ts_small_comm* comm, float* buf, int64_t n, ts_stream_t stream);
extern "C" int64_t ts_allreduce_small_capacity(const ts_small_comm* comm);

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*comm_count)(const ncclComm_t, int*) = nullptr;
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
    r.comm_count = reinterpret_cast<decltype(r.comm_count)>(dlsym(h, "ncclCommCount"));
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
    *out = new ts_comm{comm, (int)rank, (int)world, device, nullptr};
    return TS_OK;
}

int ts_allreduce(ts_comm* comm, float* buf, int64_t n, ts_stream_t stream) {
    TS_REQUIRE(comm != nullptr, TS_ERR_INVALID_ARG, "ts_allreduce: communicator is NULL");
    TS_REQUIRE(n >= 0 && (buf != nullptr || n == 0), TS_ERR_INVALID_ARG, "ts_allreduce: bad buffer");
    if (n == 0) return TS_OK;
    if (comm->small && n <= ts_allreduce_small_capacity(comm->small)) return ts_allreduce_small(comm->small, buf, n, stream);
    TS_REQUIRE(comm->comm != nullptr, TS_ERR_UNSUPPORTED,
               "ts_allreduce: %lld floats exceed the one-shot path and this communicator has no RCCL side", (long long)n);
    const ncclResult_t r = g_rccl.all_reduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, comm->comm, ts::as_stream(stream));
    if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
    return TS_OK;
}

int ts_allreduce_ranks(const ts_comm* comm, int64_t* world, int64_t* rccl_ranks) {
    TS_REQUIRE(comm && world && rccl_ranks, TS_ERR_INVALID_ARG, "ts_allreduce_ranks: NULL argument");
    *world = comm->world;
    *rccl_ranks = 0;
    if (comm->comm && g_rccl.comm_count) {
        int n = 0;
        const ncclResult_t r = g_rccl.comm_count(comm->comm, &n);
        if (r != ncclSuccess) return rccl_fail("ncclCommCount", r);
        *rccl_ranks = n;
    }
    return TS_OK;
}

int ts_allreduce_destroy(ts_comm* comm) {
    if (!comm) return TS_OK;
    if (comm->comm && g_rccl.comm_destroy) g_rccl.comm_destroy(comm->comm);
    delete comm;
    return TS_OK;
}


}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// One-shot all-reduce for small payloads (<= 64 KB: the 44 KB gradient + loss parts of a PPO minibatch step sits on the
// critical path of a ~55 us kernel).  No ring, no proxy thread: every rank owns one device buffer that the
// other ranks map through HIP IPC (xGMI peer access inside a node).  Per call a single workgroup
//   1. copies its payload into its own buffer (slot = call parity) with system-scope write-through stores,
//   2. drains them (s_waitcnt vmcnt(0)) and publishes the call number in its flag word (system-scope release),
//   3. polls the flag words of the peers (relaxed system-scope loads, s_sleep between polls, bounded),
//   4. sums the W payloads in rank order 0 .. W-1 (the same order on every rank: all replicas obtain bit-identical
//      results, and for W = 2 the result equals any other all-reduce bit for bit) and writes the result in place.
// Two slots are enough: a rank that starts call e + 1 has finished reading call e everywhere it matters -- nobody
// overwrites slot (e & 1) before call e + 2, which it only enters after seeing every peer's flag reach e + 1.
// The same agent / system-scope hand-off idiom as the single-pass GAE scan (ts_returns.hip), across devices.
// ------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int SMALL_MAX_WORLD = 8;
constexpr int64_t SMALL_HEADER_FLOATS = 64;          // flag (u64), error (u32), padding: two 128-byte lines

struct SmallPeers { float* p[SMALL_MAX_WORLD]; };

__device__ __forceinline__ void store_sys(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long load_sys(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(1024) void small_allreduce_kernel(float* __restrict__ buf, int64_t n, SmallPeers peers, int rank,
                                                               int world, unsigned long long epoch, int64_t cap,
                                                               unsigned long long spin_limit) {
    const int tid = threadIdx.x;
    float* mine = peers.p[rank];
    const int64_t slot_off = SMALL_HEADER_FLOATS + (int64_t)(epoch & 1) * cap;
    const int64_t n2 = (n + 1) / 2;                                  // 8-byte granules (cap is even, buffers are padded)
    // 1. publish the payload
    for (int64_t i = tid; i < n2; i += 1024) {
        const float a = buf[2 * i], b = 2 * i + 1 < n ? buf[2 * i + 1] : 0.f;
        const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
        store_sys(reinterpret_cast<unsigned long long*>(mine + slot_off) + i, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int s_bad;
    if (tid == 0) {
        s_bad = 0;
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    // 3. wait for the peers' flags (one lane per peer)
    if (tid < world && tid != rank) {
        const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(peers.p[tid]);
        unsigned long long spins = 0;
        while (load_sys(flag) < epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > spin_limit) { s_bad = 1; break; }
        }
    }
    __syncthreads();
    if (s_bad) {                                                     // a peer never arrived: flag the error, leave buf alone
        if (tid == 0) reinterpret_cast<unsigned*>(mine)[2] = 1u;
        return;
    }
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // system scope
    __syncthreads();
    // 4. sum in rank order
    for (int64_t i = tid; i < n2; i += 1024) {
        float s0 = 0.f, s1 = 0.f;
        for (int r = 0; r < world; ++r) {
            const unsigned long long v = load_sys(reinterpret_cast<const unsigned long long*>(peers.p[r] + slot_off) + i);
            const float a = __uint_as_float((unsigned)(v & 0xffffffffull)), b = __uint_as_float((unsigned)(v >> 32));
            s0 = r == 0 ? a : s0 + a;
            s1 = r == 0 ? b : s1 + b;
        }
        buf[2 * i] = s0;
        if (2 * i + 1 < n) buf[2 * i + 1] = s1;
    }
}

}  // namespace

struct ts_small_comm {
    int rank, world, device;
    int64_t cap;                       // floats per slot (even)
    float* local;                      // header + 2 slots, owned
    float* peer[SMALL_MAX_WORLD];      // peer[rank] == local
    bool opened[SMALL_MAX_WORLD];
    unsigned long long epoch;
};

extern "C" {

int ts_allreduce_small_create(int device, int64_t max_floats, ts_small_comm** out, uint8_t* h_handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the boundary ships the IPC handle as 64 bytes");
    TS_REQUIRE(out && h_handle64, TS_ERR_INVALID_ARG, "ts_allreduce_small_create: NULL argument");
    TS_REQUIRE(max_floats >= 1 && max_floats <= 16384, TS_ERR_INVALID_ARG,
               "ts_allreduce_small_create: 1 .. 16384 floats (64 KB) per call, got %lld", (long long)max_floats);
    TS_HIP_CHECK(hipSetDevice(device));
    const int64_t cap = (max_floats + 1) / 2 * 2;
    const size_t bytes = sizeof(float) * (size_t)(SMALL_HEADER_FLOATS + 2 * cap);
    float* p = nullptr;
    // fine-grained device memory: peers (and other processes on this device) observe the write-through stores
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&p), bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        p = nullptr;
        TS_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), bytes));
    }
    TS_HIP_CHECK(hipMemset(p, 0, bytes));
    TS_HIP_CHECK(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    TS_HIP_CHECK(hipIpcGetMemHandle(&h, p));
    std::memcpy(h_handle64, &h, sizeof(h));
    ts_small_comm* c = new ts_small_comm{};
    c->rank = -1; c->world = 0; c->device = device; c->cap = cap; c->local = p; c->epoch = 0;
    *out = c;
    return TS_OK;
}

int ts_allreduce_small_connect(ts_small_comm* comm, const uint8_t* h_handles, int64_t rank, int64_t world) {
    TS_REQUIRE(comm && h_handles, TS_ERR_INVALID_ARG, "ts_allreduce_small_connect: NULL argument");
    TS_REQUIRE(world >= 1 && world <= SMALL_MAX_WORLD && rank >= 0 && rank < world, TS_ERR_INVALID_ARG,
               "ts_allreduce_small_connect: rank %lld of %lld (at most %d ranks)", (long long)rank, (long long)world,
               SMALL_MAX_WORLD);
    TS_REQUIRE(comm->rank < 0, TS_ERR_INVALID_ARG, "ts_allreduce_small_connect: already connected");
    TS_HIP_CHECK(hipSetDevice(comm->device));
    for (int r = 0; r < world; ++r) {
        if (r == rank) { comm->peer[r] = comm->local; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, h_handles + 64 * r, sizeof(h));
        void* p = nullptr;
        TS_HIP_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        comm->peer[r] = static_cast<float*>(p);
        comm->opened[r] = true;
    }
    comm->rank = (int)rank; comm->world = (int)world;
    return TS_OK;
}

int ts_allreduce_small(ts_small_comm* comm, float* buf, int64_t n, ts_stream_t stream) {
    TS_REQUIRE(comm != nullptr && comm->rank >= 0, TS_ERR_INVALID_ARG, "ts_allreduce_small: communicator not connected");
    TS_REQUIRE(n >= 0 && n <= comm->cap && (buf != nullptr || n == 0), TS_ERR_INVALID_ARG,
               "ts_allreduce_small: %lld floats (capacity %lld)", (long long)n, (long long)comm->cap);
    if (n == 0 || comm->world == 1) return TS_OK;
    SmallPeers peers{};
    for (int r = 0; r < comm->world; ++r) peers.p[r] = comm->peer[r];
    static const unsigned long long spin_limit = [] {
        const char* e = getenv("TS_SMALL_ALLREDUCE_SPINS");
        return e ? strtoull(e, nullptr, 10) : 20000000ull;          // x (s_sleep 8 + a fabric load) ~ several seconds
    }();
    ++comm->epoch;
    hipLaunchKernelGGL(small_allreduce_kernel, dim3(1), dim3(1024), 0, ts::as_stream(stream), buf, n, peers, comm->rank,
                       comm->world, comm->epoch, comm->cap, spin_limit);
    TS_LAUNCH_CHECK();
    return TS_OK;
}

int64_t ts_allreduce_small_capacity(const ts_small_comm* comm) { return comm ? comm->cap : 0; }

int ts_allreduce_attach_small(ts_comm* comm, ts_small_comm* small) {
    TS_REQUIRE(comm != nullptr, TS_ERR_INVALID_ARG, "ts_allreduce_attach_small: communicator is NULL");
    TS_REQUIRE(small == nullptr || (small->rank == comm->rank && small->world == comm->world), TS_ERR_INVALID_ARG,
               "ts_allreduce_attach_small: rank / world of the two communicators differ");
    comm->small = small;
    return TS_OK;
}

int ts_allreduce_from_small(ts_small_comm* small, ts_comm** out) {
    TS_REQUIRE(small != nullptr && small->rank >= 0 && out != nullptr, TS_ERR_INVALID_ARG,
               "ts_allreduce_from_small: a connected one-shot communicator is needed");
    *out = new ts_comm{nullptr, small->rank, small->world, small->device, small};
    return TS_OK;
}

int ts_allreduce_small_status(ts_small_comm* comm, ts_stream_t stream) {
    TS_REQUIRE(comm != nullptr, TS_ERR_INVALID_ARG, "ts_allreduce_small_status: communicator is NULL");
    unsigned flag = 0;
    TS_HIP_CHECK(hipMemcpyAsync(&flag, reinterpret_cast<unsigned*>(comm->local) + 2, sizeof(flag), hipMemcpyDeviceToHost,
                                ts::as_stream(stream)));
    TS_HIP_CHECK(hipStreamSynchronize(ts::as_stream(stream)));
    TS_REQUIRE(flag == 0, TS_ERR_HIP, "ts_allreduce_small: a peer did not arrive within the spin limit (rank %d of %d)",
               comm->rank, comm->world);
    return TS_OK;
}

int ts_allreduce_small_destroy(ts_small_comm* comm) {
    if (!comm) return TS_OK;
    (void)hipSetDevice(comm->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < SMALL_MAX_WORLD; ++r)
        if (comm->opened[r]) (void)hipIpcCloseMemHandle(comm->peer[r]);
    if (comm->local) (void)hipFree(comm->local);
    delete comm;
    return TS_OK;
}

}  // extern "C"
