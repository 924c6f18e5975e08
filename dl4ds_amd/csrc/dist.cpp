// RCCL communicator for the data-parallel gradient all-reduce / initial broadcast.
// Horovod call sites replaced: dl4ds/training/base.py:97-107 (init, local rank -> device),
// supervised.py:365 (DistributedOptimizer: average all-reduce of every gradient), supervised.py:369 and
// cgan.py:633-637 (broadcast of variables and optimiser slots from rank 0), cgan.py:608-611.
//
// xGMI is point-to-point (7 links x ~153 GB/s per GPU); the gradient arena of the headline model is 0.82 MB,
// i.e. latency-bound.  The supervised trainer sends the arena as a few buckets (Graph::plan_buckets), each launched
// on the side stream the moment the backward pass has finished the ops that own it -- the tail of the network
// first -- so the collectives run under the remaining backward kernels (Horovod's DistributedOptimizer does the same
// with its fusion buffer); the CGAN trainer, whose discriminator is back-propagated twice, sends each arena once.
// The 1/world average is folded into the Adam kernel (adam.hip).
#include "dist.h"
#include "runtime.h"
#include "prof.h"
#include <rccl/rccl.h>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <strings.h>
#include <cstdlib>
#include <thread>

namespace {
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;
// one "bucket is final" event per bucket launch (a ring, far longer than the number of buckets in flight): a wait captures
// the record it follows, so a single re-recorded event would be correct too -- separate events keep that from being a
// property every future edit has to preserve
constexpr int kEvRing = 128;
hipEvent_t g_ev_ring[kEvRing] = {};
int g_ev_next = 0;
hipEvent_t g_ev_done = nullptr;
// DL4DS_DIST_STANDIN=1 (profiling aid, profiles/rccl_overlap_*.txt): a 1-rank communicator's in-place all-reduce launches
// NO device kernel (RCCL returns early), and RCCL refuses two ranks on one GPU, so a single-GPU trace cannot show where the
// collectives sit.  With this switch every bucket launch is followed, on the communication stream, by a stand-in kernel
// that reads and rewrites the bucket (values unchanged): the trace then shows the stream / event topology -- what runs
// concurrently with the backward pass and what Adam waits for.  It is NOT a collective and says so in its name.
__global__ void standin_for_rccl_allreduce_kernel(float* buf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = buf[i];
        asm volatile("" : "+v"(v));
        buf[i] = v;
    }
}
void launch_standin(float* buf, size_t n, hipStream_t cs) {
    static const bool on = test_env("DL4DS_DIST_STANDIN") != nullptr;
    if (!on || g_world != 1) return;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 64);      // few workgroups: a collective occupies few CUs
    DL4DS_LAUNCH(standin_for_rccl_allreduce_kernel, dim3(blocks), dim3(256), 0, cs, buf, n);
    HIP_CHECK(hipGetLastError());
}
hipEvent_t next_ready_event() {
    hipEvent_t& e = g_ev_ring[g_ev_next];
    g_ev_next = (g_ev_next + 1) % kEvRing;
    if (!e) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return e;
}
float* g_small = nullptr;            // device scratch of the host-side reductions
constexpr int kSmallFloats = 1024;

#define NCCL_CHECK(expr)                                                                         \
    do {                                                                                         \
        ncclResult_t _r = (expr);                                                                \
        if (_r != ncclSuccess)                                                                   \
            throw Dl4dsError(std::string("RCCL error: ") + ncclGetErrorString(_r) + " (" #expr ")"); \
    } while (0)
}  // namespace

void dist_unique_id(char id128[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId id;
    NCCL_CHECK(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, 128);
}

void dist_init(int rank, int world, const char id128[128]) {
    rt_ensure_init();
    DL4DS_REQUIRE(g_comm == nullptr, "dist already initialised");
    DL4DS_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank/world");
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    NCCL_CHECK(ncclCommInitRank(&g_comm, world, id, rank));
    g_rank = rank;
    g_world = world;
    HIP_CHECK(hipEventCreateWithFlags(&g_ev_done, hipEventDisableTiming));
}

void dist_world(int& rank, int& world) {
    rank = g_rank;
    world = g_world;
}

void dist_allreduce_grads(float* buf, size_t n, hipStream_t stream) {
    if (g_comm == nullptr || n == 0) return;      // (a 1-rank communicator still runs the collective: smoke-tests the path)
    hipStream_t cs = rt().comm_stream;
    hipEvent_t ready = next_ready_event();
    HIP_CHECK(hipEventRecord(ready, stream));
    HIP_CHECK(hipStreamWaitEvent(cs, ready, 0));
    NCCL_CHECK(ncclAllReduce(buf, buf, n, ncclFloat32, ncclSum, g_comm, cs));
    HIP_CHECK(hipEventRecord(g_ev_done, cs));
    HIP_CHECK(hipStreamWaitEvent(stream, g_ev_done, 0));
}

bool dist_active() { return g_comm != nullptr; }

void dist_allreduce_bucket_async(float* buf, size_t n, hipStream_t stream, hipStream_t aux) {
    if (g_comm == nullptr || n == 0) return;
    hipStream_t cs = rt().comm_stream;
    hipEvent_t ready = next_ready_event();
    HIP_CHECK(hipEventRecord(ready, stream));
    HIP_CHECK(hipStreamWaitEvent(cs, ready, 0));
    if (aux) {
        hipEvent_t ready_aux = next_ready_event();
        HIP_CHECK(hipEventRecord(ready_aux, aux));
        HIP_CHECK(hipStreamWaitEvent(cs, ready_aux, 0));
    }
    NCCL_CHECK(ncclAllReduce(buf, buf, n, ncclFloat32, ncclSum, g_comm, cs));
    launch_standin(buf, n, cs);
}

// Host-side wait for `stream` with a watchdog.  With more than one rank, work queued behind a collective only completes
// when EVERY rank has entered that collective: a rank that crashed, skipped a step or called the collectives in another
// order would leave the others in hipStreamSynchronize for ever.  Poll instead, surface RCCL's asynchronous errors and
// give up after DL4DS_COLLECTIVE_TIMEOUT_S (default 1800 s) with a message that names the call.
void dist_stream_sync(hipStream_t stream, const char* what) {
    // (DL4DS_FORCE_WATCHDOG=1: take the polling path with a 1-rank communicator too -- the only way to exercise it on one GPU)
    static const bool force = test_env("DL4DS_FORCE_WATCHDOG") != nullptr;
    if (g_comm == nullptr || (g_world <= 1 && !force)) {
        HIP_CHECK(hipStreamSynchronize(stream));
        device_error_check(what);
        return;
    }
    static const double limit = [] {
        const char* e = getenv("DL4DS_COLLECTIVE_TIMEOUT_S");
        const double v = e ? std::atof(e) : 0.0;
        return v > 0.0 ? v : 1800.0;
    }();
    const auto t0 = std::chrono::steady_clock::now();
    for (long spin = 0;; ++spin) {
        const hipError_t e = hipStreamQuery(stream);
        // hipErrorNotReady is a status, but HIP records it as the thread's sticky "last error": clear it, or the next
        // HIP_CHECK(hipGetLastError()) after a kernel launch would report it
        if (e == hipErrorNotReady) (void)hipGetLastError();
        if (e == hipSuccess) { device_error_check(what); return; }
        if (e != hipErrorNotReady) HIP_CHECK(e);
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > 2e-3) {                                   // short waits stay a pure spin
            ncclResult_t async = ncclSuccess;
            if (ncclCommGetAsyncError(g_comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress)
                throw Dl4dsError(std::string(what) + ": RCCL reported an asynchronous error on rank " + std::to_string(g_rank) +
                                 " of " + std::to_string(g_world) + ": " + ncclGetErrorString(async));
            if (waited > limit)
                throw Dl4dsError(std::string(what) + ": rank " + std::to_string(g_rank) + " of " + std::to_string(g_world) +
                                 " waited " + std::to_string((long)waited) + " s for work queued behind an RCCL collective -- "
                                 "another rank has died, skipped a step or entered the collectives in a different order "
                                 "(DL4DS_COLLECTIVE_TIMEOUT_S sets this limit)");
            std::this_thread::sleep_for(std::chrono::microseconds(waited < 0.05 ? 20 : 200));
        }
    }
}

void dist_allreduce_wait(hipStream_t stream) {
    if (g_comm == nullptr) return;
    HIP_CHECK(hipEventRecord(g_ev_done, rt().comm_stream));
    HIP_CHECK(hipStreamWaitEvent(stream, g_ev_done, 0));
}

void dist_broadcast(float* buf, size_t n, int root, hipStream_t stream) {
    if (g_comm == nullptr || n == 0) return;      // (a 1-rank communicator still runs the collective: smoke-tests the path)
    NCCL_CHECK(ncclBroadcast(buf, buf, n, ncclFloat32, root, g_comm, stream));
}

int dist_expected_world() {
    // (same reading as dl4ds_amd.parallel.allow_unsynced: unset, empty, "0", "false", "no", "off" mean NO)
    const char* allow = getenv("DL4DS_ALLOW_UNSYNCED");
    if (allow && allow[0] && std::strcmp(allow, "0") != 0 && strcasecmp(allow, "false") != 0 && strcasecmp(allow, "no") != 0 &&
        strcasecmp(allow, "off") != 0)
        return 1;
    const char* w = std::getenv("WORLD_SIZE");
    const int n = w ? std::atoi(w) : 1;
    return n > 1 ? n : 1;
}

void dist_require_ready(const char* what) {
    const int expect = dist_expected_world();
    if (g_comm != nullptr) {
        if (expect > 1 && expect != g_world)
            throw Dl4dsError(std::string(what) + ": the launcher started WORLD_SIZE=" + std::to_string(expect) +
                             " ranks but the RCCL communicator has " + std::to_string(g_world));
        return;
    }
    if (expect > 1)
        throw Dl4dsError(std::string(what) + ": this process is rank " + (std::getenv("RANK") ? std::getenv("RANK") : "?") +
                         " of WORLD_SIZE=" + std::to_string(expect) +
                         " but no RCCL communicator exists (dl4ds_dist_init / dl4ds_amd.parallel.init_from_env was not "
                         "called): refusing to train an unsynchronised replica.  Set DL4DS_ALLOW_UNSYNCED=1 to run "
                         "independent replicas on purpose.");
}

void dist_allreduce_host(float* host, int n, int op) {
    if (g_comm == nullptr || n <= 0) return;
    DL4DS_REQUIRE(n <= kSmallFloats, "dist_allreduce_host: at most 1024 values");
    DL4DS_REQUIRE(op >= 0 && op <= 2, "dist_allreduce_host: op must be 0 (sum), 1 (max) or 2 (min)");
    hipStream_t s = rt().stream;
    if (!g_small) HIP_CHECK(hipMalloc((void**)&g_small, kSmallFloats * sizeof(float)));
    HIP_CHECK(hipMemcpyAsync(g_small, host, n * sizeof(float), hipMemcpyHostToDevice, s));
    const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
    NCCL_CHECK(ncclAllReduce(g_small, g_small, n, ncclFloat32, ops[op], g_comm, s));
    HIP_CHECK(hipMemcpyAsync(host, g_small, n * sizeof(float), hipMemcpyDeviceToHost, s));
    dist_stream_sync(s, "dl4ds_dist_allreduce_host");
}

void dist_barrier() {
    dist_stream_sync(rt().stream, "dl4ds_dist_barrier");
    float one = 1.f;
    dist_allreduce_host(&one, 1, 0);
}

void dist_comm_info(int& nranks, int& rank, int& device) {
    nranks = 0; rank = 0; device = -1;
    if (g_comm == nullptr) return;
    NCCL_CHECK(ncclCommCount(g_comm, &nranks));
    NCCL_CHECK(ncclCommUserRank(g_comm, &rank));
    NCCL_CHECK(ncclCommCuDevice(g_comm, &device));
}

void dist_broadcast_i64(long* host_value, int root) {
    if (g_comm == nullptr) return;
    static_assert(sizeof(long) == 8, "long is 64-bit on this platform");
    hipStream_t s = rt().stream;
    if (!g_small) HIP_CHECK(hipMalloc((void**)&g_small, kSmallFloats * sizeof(float)));
    HIP_CHECK(hipMemcpyAsync(g_small, host_value, 8, hipMemcpyHostToDevice, s));
    NCCL_CHECK(ncclBroadcast(g_small, g_small, 1, ncclInt64, root, g_comm, s));
    HIP_CHECK(hipMemcpyAsync(host_value, g_small, 8, hipMemcpyDeviceToHost, s));
    dist_stream_sync(s, "dl4ds_dist_broadcast (optimizer.iterations)");
}

void dist_finalize() {
    if (g_small) { (void)hipFree(g_small); g_small = nullptr; }
    if (g_comm) {
        (void)ncclCommDestroy(g_comm);
        g_comm = nullptr;
    }
    g_rank = 0;
    g_world = 1;
}
