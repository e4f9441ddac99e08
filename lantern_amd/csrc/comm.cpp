// comm.cpp -- transports of the work-sharded build's exchange step (see comm.hpp).
#include "comm.hpp"
#include "abi_guard.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

namespace lgpu {

// ---- RCCL, bound at run time ---------------------------------------------------------------------------
namespace {
struct RcclApi
{
    void *so = nullptr;
    decltype(&ncclGetUniqueId)    GetUniqueId = nullptr;
    decltype(&ncclCommInitRank)   CommInitRank = nullptr;
    decltype(&ncclCommDestroy)    CommDestroy = nullptr;
    decltype(&ncclCommAbort)      CommAbort = nullptr;
    decltype(&ncclBroadcast)      Broadcast = nullptr;
    decltype(&ncclAllGather)      AllGather = nullptr;
    decltype(&ncclGroupStart)     GroupStart = nullptr;
    decltype(&ncclGroupEnd)       GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string                   why;
};

RcclApi *rccl_api()
{
    static RcclApi    api;
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if(api.so || !api.why.empty()) return &api;
    // LANTERN_GPU_RCCL_LIB names the library to bind instead (a site's own RCCL build; the test double of tests/fake_rccl/, which
    // runs this transport with several in-process ranks on one device).  Otherwise: a host that already mapped RCCL (PyTorch
    // bundles librccl.so.1 under the same SONAME) gets that copy.
    const char *override_path = std::getenv("LANTERN_GPU_RCCL_LIB");
    if(override_path && *override_path) {
        api.so = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
        if(!api.so) {
            api.why = std::string("lantern_gpu: cannot load LANTERN_GPU_RCCL_LIB=") + override_path + ": " + (dlerror() ? dlerror() : "unknown error");
            return &api;
        }
    }
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for(const char *n : names) {
        if(api.so) break;
        api.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if(!api.so) {
        api.why = std::string("lantern_gpu: cannot load librccl.so.1: ") + (dlerror() ? dlerror() : "unknown error");
        return &api;
    }
#define BIND(field, sym)                                                         \
    api.field = (decltype(api.field))dlsym(api.so, sym);                        \
    if(!api.field) { api.why = std::string("lantern_gpu: librccl lacks ") + sym; api.so = nullptr; return &api; }
    BIND(GetUniqueId, "ncclGetUniqueId")
    BIND(CommInitRank, "ncclCommInitRank")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(CommAbort, "ncclCommAbort")
    BIND(Broadcast, "ncclBroadcast")
    BIND(AllGather, "ncclAllGather")
    BIND(GroupStart, "ncclGroupStart")
    BIND(GroupEnd, "ncclGroupEnd")
    BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    return &api;
}
}  // namespace

// ---- in-process hub: a generation barrier over which W threads exchange host pointers -------------------
struct LocalHub
{
    std::mutex              mu;
    std::condition_variable cv;
    int                     world = 1, arrived = 0;
    uint64_t                generation = 0;
    std::vector<void *>     bufs;
    bool                    broken = false;

    bool barrier(double timeout_s)
    {
        std::unique_lock<std::mutex> lk(mu);
        if(broken) return false;
        const uint64_t gen = generation;
        if(++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
            return true;
        }
        const bool ok = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return generation != gen || broken; });
        if(!ok || broken) { broken = true; cv.notify_all(); return false; }
        return true;
    }
};

static int hub_allgatherv(void *ctx, void *host_buf, const size_t *off, const size_t *cnt, int world, int rank)
{
    Comm     *c = (Comm *)ctx;
    LocalHub *h = c->hub.get();
    {
        std::lock_guard<std::mutex> g(h->mu);
        h->bufs[ (size_t)rank ] = host_buf;
    }
    if(!h->barrier(c->timeout_s)) return 1;
    for(int r = 0; r < world; ++r)
        if(r != rank && cnt[ r ]) std::memcpy((char *)host_buf + off[ r ], (const char *)h->bufs[ (size_t)r ] + off[ r ], cnt[ r ]);
    return h->barrier(c->timeout_s) ? 0 : 1;  // nobody may reuse its buffer before everyone has copied
}

bool Comm::wait(hipStream_t st)
{
    if(!rccl) {
        if(hipStreamSynchronize(st) != hipSuccess) { err = "lantern_gpu: HIP failure waiting for the stream"; return false; }
        return true;
    }
    // a peer that died or diverged would leave the collective kernel spinning for ever: poll against a deadline
    const auto t0 = std::chrono::steady_clock::now();
    for(unsigned spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(st);
        if(q == hipSuccess) return true;
        if(q != hipErrorNotReady) { err = std::string("lantern_gpu: HIP failure waiting for a collective: ") + hipGetErrorString(q); return false; }
        if(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
            err = "lantern_gpu: collective timed out (a peer rank is missing or diverged)";
            if(nccl_comm) { (void)rccl_api()->CommAbort((ncclComm_t)nccl_comm); nccl_comm = nullptr; }
            return false;
        }
        if(spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

bool Comm::allgatherv_device(void *d_buf, const size_t *off, const size_t *cnt, hipStream_t st)
{
    size_t extent = 0, total = 0;
    for(int r = 0; r < world; ++r) {
        extent = std::max(extent, off[ r ] + cnt[ r ]);
        total += cnt[ r ];
    }
    collectives += 1;
    bytes_exchanged += total - cnt[ rank ];  // bytes this rank receives
    if(world == 1 || total == 0) return true;
    if(rccl) {
        RcclApi *api = rccl_api();
        if(!nccl_comm) { err = "lantern_gpu: the RCCL communicator was aborted"; return false; }
        ncclResult_t rc = api->GroupStart();
        for(int r = 0; r < world && rc == ncclSuccess; ++r) {
            if(cnt[ r ] == 0) continue;
            char *seg = (char *)d_buf + off[ r ];
            rc = api->Broadcast(seg, seg, cnt[ r ], ncclInt8, r, (ncclComm_t)nccl_comm, st);
        }
        const ncclResult_t rc2 = api->GroupEnd();
        if(rc == ncclSuccess) rc = rc2;
        if(rc != ncclSuccess) { err = std::string("lantern_gpu: RCCL all-gather failed: ") + api->GetErrorString(rc); return false; }
        return true;
    }
    // host transport: stage own segment, exchange, copy the peers' segments back
    if(hipStreamSynchronize(st) != hipSuccess) { err = "lantern_gpu: HIP failure before the exchange"; return false; }
    if(stage.size() < extent) stage.resize(extent);
    if(cnt[ rank ] && hipMemcpy(stage.data() + off[ rank ], (char *)d_buf + off[ rank ], cnt[ rank ], hipMemcpyDeviceToHost) != hipSuccess) {
        err = "lantern_gpu: HIP failure staging the exchange";
        return false;
    }
    if(fn(fn_ctx, stage.data(), off, cnt, world, rank) != 0) { err = "lantern_gpu: the caller's all-gather failed or timed out"; return false; }
    for(int r = 0; r < world; ++r) {
        if(r == rank || cnt[ r ] == 0) continue;
        if(hipMemcpyAsync((char *)d_buf + off[ r ], stage.data() + off[ r ], cnt[ r ], hipMemcpyHostToDevice, st) != hipSuccess) {
            err = "lantern_gpu: HIP failure unstaging the exchange";
            return false;
        }
    }
    // the staging buffer is reused by the next exchange
    if(hipStreamSynchronize(st) != hipSuccess) { err = "lantern_gpu: HIP failure after the exchange"; return false; }
    return true;
}

bool Comm::allgatherv_host(void *h_buf, const size_t *off, const size_t *cnt)
{
    if(world == 1) return true;
    if(!rccl) {
        if(fn(fn_ctx, h_buf, off, cnt, world, rank) != 0) { err = "lantern_gpu: the caller's all-gather failed or timed out"; return false; }
        return true;
    }
    size_t extent = 0;
    for(int r = 0; r < world; ++r) extent = std::max(extent, off[ r ] + cnt[ r ]);
    void *d = nullptr;
    if(hipMalloc(&d, extent ? extent : 16) != hipSuccess) { err = "lantern_gpu: out of device memory (exchange)"; return false; }
    bool ok = hipMemcpy((char *)d + off[ rank ], (char *)h_buf + off[ rank ], cnt[ rank ], hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && allgatherv_device(d, off, cnt, nullptr) && wait(nullptr);
    ok = ok && hipMemcpy(h_buf, d, extent, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    if(!ok && err.empty()) err = "lantern_gpu: HIP failure in the metadata exchange";
    return ok;
}

}  // namespace lgpu

// =====================================================================================================
// C ABI
// =====================================================================================================
using namespace lgpu;

#define CLEAR(e) do { if(e) *(e) = nullptr; } while(0)
#define FAIL(e, msg) do { if(e) *(e) = (msg); } while(0)

static thread_local std::string g_comm_err;
static const char *keep_err(const std::string &s)
{
    g_comm_err = s;
    return g_comm_err.c_str();
}

extern "C" {

void lantern_gpu_comm_unique_id(char *id128, usearch_error_t *e)
try {
    CLEAR(e);
    int ndev = 0;
    if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { FAIL(e, "lantern_gpu: no HIP device available (this library has no CPU fallback)"); return; }
    RcclApi *api = rccl_api();
    if(!api->so) { FAIL(e, keep_err(api->why)); return; }
    ncclUniqueId       id;
    const ncclResult_t rc = api->GetUniqueId(&id);
    if(rc != ncclSuccess) { FAIL(e, keep_err(std::string("lantern_gpu: ncclGetUniqueId: ") + api->GetErrorString(rc))); return; }
    static_assert(sizeof(id) == LANTERN_GPU_COMM_ID_BYTES, "unique id size");
    std::memcpy(id128, &id, sizeof(id));
}
LANTERN_ABI_CATCH_VOID(e)

lantern_gpu_comm_t *lantern_gpu_comm_init_rccl(int rank, int world, const char *id128, usearch_error_t *e)
try {
    CLEAR(e);
    if(world < 1 || rank < 0 || rank >= world || !id128) { FAIL(e, "lantern_gpu: bad rank / world / id"); return nullptr; }
    // the device check comes first: a host without a GPU never maps librccl (573 MB, and a process that later imports
    // PyTorch would otherwise end up with ROCm's RCCL under torch's bundled HIP stack)
    int ndev = 0;
    if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { FAIL(e, "lantern_gpu: no HIP device available (this library has no CPU fallback)"); return nullptr; }
    RcclApi *api = rccl_api();
    if(!api->so) { FAIL(e, keep_err(api->why)); return nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t         nc = nullptr;
    const ncclResult_t rc = api->CommInitRank(&nc, world, id, rank);  // binds the calling thread's current HIP device
    if(rc != ncclSuccess) { FAIL(e, keep_err(std::string("lantern_gpu: ncclCommInitRank: ") + api->GetErrorString(rc))); return nullptr; }
    Comm *c = new Comm();
    c->rank = rank;
    c->world = world;
    c->rccl = true;
    c->nccl_comm = nc;
    return (lantern_gpu_comm_t *)c;
}
LANTERN_ABI_CATCH(e)

lantern_gpu_comm_t *lantern_gpu_comm_init_host(int rank, int world, lantern_gpu_allgatherv_fn fn, void *ctx, usearch_error_t *e)
try {
    CLEAR(e);
    if(world < 1 || rank < 0 || rank >= world || !fn) { FAIL(e, "lantern_gpu: bad rank / world / callback"); return nullptr; }
    Comm *c = new Comm();
    c->rank = rank;
    c->world = world;
    c->fn = fn;
    c->fn_ctx = ctx;
    return (lantern_gpu_comm_t *)c;
}
LANTERN_ABI_CATCH(e)

void lantern_gpu_comm_init_local(int world, lantern_gpu_comm_t **out, usearch_error_t *e)
try {
    CLEAR(e);
    if(world < 1 || !out) { FAIL(e, "lantern_gpu: bad world size"); return; }
    auto hub = std::make_shared<LocalHub>();
    hub->world = world;
    hub->bufs.assign((size_t)world, nullptr);
    for(int r = 0; r < world; ++r) {
        Comm *c = new Comm();
        c->rank = r;
        c->world = world;
        c->hub = hub;
        c->fn = hub_allgatherv;
        c->fn_ctx = c;
        out[ r ] = (lantern_gpu_comm_t *)c;
    }
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_comm_free(lantern_gpu_comm_t *h)
try {
    Comm *c = (Comm *)h;
    if(!c) return;
    if(c->rccl && c->nccl_comm) (void)rccl_api()->CommDestroy((ncclComm_t)c->nccl_comm);
    delete c;
}
LANTERN_ABI_CATCH_VOID(nullptr)

int lantern_gpu_comm_rank(lantern_gpu_comm_t *h) { return h ? ((Comm *)h)->rank : 0; }
int lantern_gpu_comm_world(lantern_gpu_comm_t *h) { return h ? ((Comm *)h)->world : 1; }

void lantern_gpu_comm_set_timeout(lantern_gpu_comm_t *h, double seconds)
try {
    if(h && seconds > 0) ((Comm *)h)->timeout_s = seconds;
}
LANTERN_ABI_CATCH_VOID(nullptr)

void lantern_gpu_comm_stats(lantern_gpu_comm_t *h, uint64_t *bytes_received, uint64_t *collectives)
try {
    Comm *c = (Comm *)h;
    if(bytes_received) *bytes_received = c ? c->bytes_exchanged : 0;
    if(collectives) *collectives = c ? c->collectives : 0;
}
LANTERN_ABI_CATCH_VOID(nullptr)

void lantern_gpu_comm_allgatherv_host(lantern_gpu_comm_t *h, void *host_buf, const size_t *offsets, const size_t *counts, usearch_error_t *e)
try {
    CLEAR(e);
    Comm *c = (Comm *)h;
    if(!c || !host_buf || !offsets || !counts) { FAIL(e, "lantern_gpu: bad arguments"); return; }
    if(!c->allgatherv_host(host_buf, offsets, counts)) FAIL(e, c->err.c_str());
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_comm_allgatherv_device(lantern_gpu_comm_t *h, void *device_buf, const size_t *offsets, const size_t *counts, void *stream,
                                        usearch_error_t *e)
try {
    CLEAR(e);
    Comm *c = (Comm *)h;
    if(!c || !device_buf || !offsets || !counts) { FAIL(e, "lantern_gpu: bad arguments"); return; }
    if(!c->allgatherv_device(device_buf, offsets, counts, (hipStream_t)stream) || !c->wait((hipStream_t)stream)) FAIL(e, c->err.c_str());
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_shard_range(size_t n, int world, int rank, size_t *begin, size_t *end)
try {
    if(world < 1) world = 1;
    if(begin) *begin = n * (size_t)rank / (size_t)world;
    if(end) *end = n * ((size_t)rank + 1) / (size_t)world;
}
LANTERN_ABI_CATCH_VOID(nullptr)

}  // extern "C"
