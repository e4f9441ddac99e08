// fake_rccl.cpp -- a TEST DOUBLE of librccl for boxes with one GPU: the nine entry points lantern_amd/csrc/comm.cpp binds
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclCommAbort, ncclBroadcast, ncclAllGather, ncclGroupStart, ncclGroupEnd,
// ncclGetErrorString), implemented across the THREADS of one process with device-to-device copies on one device.
//
// Why: the RCCL transport of the sharded builds (comm.cpp: communicator bring-up, the grouped-broadcast all-gather-v, segment
// arithmetic from the owners' counts, the deadline poll) has only ever run with ONE rank -- real RCCL refuses two ranks on one
// device and no multi-GPU box was available to builder or driver.  With LANTERN_GPU_RCCL_LIB pointing here, the same code path runs
// at worlds 2, 3 and 8 with ragged shards (tests/test_gpu_fake_rccl.py).  What this double does NOT cover: xGMI, RCCL's own
// topology / protocol choices, inter-process bring-up (NCCL_SOCKET_IFNAME, IPC handles).
//
// Semantics kept: ncclCommInitRank blocks until all `world` ranks of an id arrived; collectives must be issued by every rank in
// the same order (each is matched by its sequence number and is a rendezvous); operations between GroupStart / GroupEnd are
// deferred to GroupEnd; an operation is complete on return (stronger than RCCL's stream ordering, which the caller's
// Comm::wait then finds already satisfied).  Buffers are device pointers of the one shared device.
//
//   hipcc -O2 -shared -fPIC -o librccl_fake.so fake_rccl.cpp      (tests/test_gpu_fake_rccl.py builds it)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct World
{
    int                       world = 0;
    std::mutex                mu;
    std::condition_variable   cv;
    int                       arrived = 0;
    uint64_t                  generation = 0;
    bool                      broken = false;
    std::vector<const void *> slot;  // per rank: the pointer it publishes for the collective in flight
    int                       joined = 0;

    bool barrier()
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
        const bool ok = cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != gen || broken; });
        if(!ok || broken) { broken = true; cv.notify_all(); return false; }
        return true;
    }
};

struct FakeComm
{
    std::shared_ptr<World> w;
    int                    rank = 0;
};

struct Op
{
    int         kind;  // 0 broadcast, 1 all-gather
    const void *send;
    void       *recv;
    size_t      bytes;
    int         root;
    FakeComm   *comm;
    hipStream_t st;
};

std::mutex                                    g_mu;
std::map<std::string, std::shared_ptr<World>> g_worlds;
std::atomic<uint64_t>                         g_next_id{ 1 };
thread_local int                              t_group_depth = 0;
thread_local std::vector<Op>                  t_pending;

size_t type_bytes(ncclDataType_t t)
{
    switch(t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 1;
    }
}

ncclResult_t run(const Op &op)
{
    World *w = op.comm->w.get();
    const int r = op.comm->rank;
    // what this rank contributes must be complete before a peer reads it
    if(hipStreamSynchronize(op.st) != hipSuccess) return ncclUnhandledCudaError;
    {
        std::lock_guard<std::mutex> g(w->mu);
        w->slot[ (size_t)r ] = op.send;
    }
    if(!w->barrier()) return ncclSystemError;
    bool ok = true;
    if(op.kind == 0) {
        const void *src = w->slot[ (size_t)op.root ];
        if(src != op.recv && op.bytes) ok = hipMemcpyAsync(op.recv, src, op.bytes, hipMemcpyDeviceToDevice, op.st) == hipSuccess;
    } else {
        for(int p = 0; p < w->world && ok; ++p) {
            char *dst = (char *)op.recv + (size_t)p * op.bytes;
            if(w->slot[ (size_t)p ] != dst && op.bytes) ok = hipMemcpyAsync(dst, w->slot[ (size_t)p ], op.bytes, hipMemcpyDeviceToDevice, op.st) == hipSuccess;
        }
    }
    ok = ok && hipStreamSynchronize(op.st) == hipSuccess;
    if(!w->barrier()) return ncclSystemError;  // nobody republishes before everyone has copied
    return ok ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t submit(const Op &op)
{
    if(t_group_depth > 0) { t_pending.push_back(op); return ncclSuccess; }
    return run(op);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if(!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof(*id));
    const uint64_t n = g_next_id.fetch_add(1);
    std::memcpy(id->internal, "FAKERCCL", 8);
    std::memcpy(id->internal + 8, &n, 8);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if(!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto &slot = g_worlds[ std::string(id.internal, sizeof(id.internal)) ];
        if(!slot) {
            slot = std::make_shared<World>();
            slot->world = nranks;
            slot->slot.assign((size_t)nranks, nullptr);
        }
        if(slot->world != nranks) return ncclInvalidArgument;
        w = slot;
    }
    FakeComm *c = new FakeComm();
    c->w = w;
    c->rank = rank;
    if(!w->barrier()) { delete c; return ncclSystemError; }  // bring-up is a rendezvous, as with the real library
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete (FakeComm *)comm;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm)
{
    FakeComm *c = (FakeComm *)comm;
    if(c) {
        {
            std::lock_guard<std::mutex> g(c->w->mu);
            c->w->broken = true;
        }
        c->w->cv.notify_all();
        delete c;
    }
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    ++t_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if(t_group_depth <= 0) return ncclInvalidUsage;
    if(--t_group_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_pending);
    for(const Op &op : ops) {
        const ncclResult_t r = run(op);
        if(r != ncclSuccess && rc == ncclSuccess) rc = r;
    }
    return rc;
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = (FakeComm *)comm;
    if(!c || root < 0 || root >= c->w->world) return ncclInvalidArgument;
    return submit(Op{ 0, sendbuff, recvbuff, count * type_bytes(datatype), root, c, stream });
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    FakeComm *c = (FakeComm *)comm;
    if(!c) return ncclInvalidArgument;
    return submit(Op{ 1, sendbuff, recvbuff, sendcount * type_bytes(datatype), 0, c, stream });
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch(r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake rccl: HIP failure";
    case ncclSystemError: return "fake rccl: a rank is missing (rendezvous timed out or the communicator was aborted)";
    case ncclInvalidArgument: return "fake rccl: invalid argument";
    case ncclInvalidUsage: return "fake rccl: invalid usage";
    default: return "fake rccl: error";
    }
}

}  // extern "C"
