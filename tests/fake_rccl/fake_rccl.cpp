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
// Two modes.  Default: the ranks are THREADS of one process and exchange device pointers (device-to-device copies).
// FAKE_RCCL_MULTIPROCESS=1: the ranks are PROCESSES (what `bench.py --gpus N` launches) that meet in a shared-memory file named by the
// unique id (/dev/shm/fake_rccl_<id>): a segment goes device -> shared memory -> device, in pieces of at most 32 MB, between barriers on
// atomics in the file's header.  Same entry points, same ordering rules.
//
//   hipcc -O2 -shared -fPIC -o librccl_fake.so fake_rccl.cpp      (tests/test_gpu_fake_rccl.py builds it)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
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

// ---- multi-process mode: the world lives in a shared-memory file ----
constexpr size_t kShmPiece = (size_t)32 << 20, kShmHeader = 4096;
struct ShmHeader
{
    std::atomic<uint32_t> magic;       // set by the creator once the header is initialised
    std::atomic<uint32_t> world;
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> broken;
};
struct ShmWorld
{
    ShmHeader *h = nullptr;
    char      *data = nullptr;  // kShmPiece * world bytes
    size_t     bytes = 0;
    int        world = 0;
    std::string path;
    ~ShmWorld() { if(h) munmap((void *)h, bytes); }

    bool barrier()
    {
        if(h->broken.load()) return false;
        const uint32_t gen = h->generation.load();
        if(h->arrived.fetch_add(1) + 1 == (uint32_t)world) {
            h->arrived.store(0);
            h->generation.fetch_add(1);
            return true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        for(unsigned spin = 0; h->generation.load() == gen; ++spin) {
            if(h->broken.load()) return false;
            if(spin > 200) std::this_thread::sleep_for(std::chrono::microseconds(20));
            if(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) { h->broken.store(1); return false; }
        }
        return true;
    }
};

struct FakeComm
{
    std::shared_ptr<World>    w;    // thread mode
    std::shared_ptr<ShmWorld> shm;  // process mode
    int                       rank = 0;
    int                       world() const { return shm ? shm->world : w->world; }
};

bool multiprocess()
{
    const char *e = std::getenv("FAKE_RCCL_MULTIPROCESS");
    return e && std::atoi(e) != 0;
}

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

// process mode: every segment travels device -> shared memory -> device in pieces of at most kShmPiece
ncclResult_t run_shm(const Op &op)
{
    ShmWorld *w = op.comm->shm.get();
    const int r = op.comm->rank, W = w->world;
    if(hipStreamSynchronize(op.st) != hipSuccess) return ncclUnhandledCudaError;
    if(op.bytes == 0) return ncclSuccess;  // (every rank sees the same size: nobody waits for anybody)
    bool ok = true;
    for(size_t done = 0; done < op.bytes; done += kShmPiece) {
        const size_t n = op.bytes - done < kShmPiece ? op.bytes - done : kShmPiece;
        if(op.kind == 0) {
            if(r == op.root && n) ok = ok && hipMemcpy(w->data, (const char *)op.send + done, n, hipMemcpyDeviceToHost) == hipSuccess;
            if(!w->barrier()) return ncclSystemError;
            if(!(r == op.root && op.send == op.recv) && n) ok = ok && hipMemcpy((char *)op.recv + done, w->data, n, hipMemcpyHostToDevice) == hipSuccess;
        } else {
            if(n) ok = ok && hipMemcpy(w->data + (size_t)r * kShmPiece, (const char *)op.send + done, n, hipMemcpyDeviceToHost) == hipSuccess;
            if(!w->barrier()) return ncclSystemError;
            for(int p = 0; p < W && ok && n; ++p)
                ok = hipMemcpy((char *)op.recv + (size_t)p * op.bytes + done, w->data + (size_t)p * kShmPiece, n, hipMemcpyHostToDevice) == hipSuccess;
        }
        if(!w->barrier()) return ncclSystemError;  // the area is reused by the next piece
    }
    return ok ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t run(const Op &op)
{
    if(op.comm->shm) return run_shm(op);
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
    uint64_t n = g_next_id.fetch_add(1);
    if(multiprocess()) n = (n << 40) ^ ((uint64_t)getpid() << 20) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();  // unique across processes
    std::memcpy(id->internal, "FAKERCCL", 8);
    std::memcpy(id->internal + 8, &n, 8);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if(!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if(multiprocess()) {
        uint64_t n = 0;
        std::memcpy(&n, id.internal + 8, 8);
        char path[ 128 ];
        std::snprintf(path, sizeof(path), "/dev/shm/fake_rccl_%016llx", (unsigned long long)n);
        const size_t bytes = kShmHeader + kShmPiece * (size_t)nranks;
        int  fd = open(path, O_RDWR | O_CREAT | O_EXCL, 0600);
        const bool creator = fd >= 0;
        if(!creator) fd = open(path, O_RDWR, 0600);
        if(fd < 0) return ncclSystemError;
        if(creator && ftruncate(fd, (off_t)bytes) != 0) { close(fd); return ncclSystemError; }
        if(!creator) {  // wait until the creator has sized the file
            struct stat sb;
            for(int i = 0; i < 100000; ++i) {
                if(fstat(fd, &sb) == 0 && (size_t)sb.st_size >= bytes) break;
                usleep(100);
            }
        }
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if(m == MAP_FAILED) return ncclSystemError;
        auto sw = std::make_shared<ShmWorld>();
        sw->h = (ShmHeader *)m;
        sw->data = (char *)m + kShmHeader;
        sw->bytes = bytes;
        sw->world = nranks;
        sw->path = path;
        if(creator) {
            sw->h->world.store((uint32_t)nranks);
            sw->h->arrived.store(0);
            sw->h->generation.store(0);
            sw->h->broken.store(0);
            sw->h->magic.store(0xFA4Eu);
        } else {
            for(int i = 0; i < 1200000 && sw->h->magic.load() != 0xFA4Eu; ++i) usleep(100);
            if(sw->h->magic.load() != 0xFA4Eu || sw->h->world.load() != (uint32_t)nranks) return ncclSystemError;
        }
        FakeComm *c = new FakeComm();
        c->shm = sw;
        c->rank = rank;
        if(!sw->barrier()) { delete c; return ncclSystemError; }  // bring-up is a rendezvous
        if(rank == 0) unlink(path);  // everyone has it mapped: the name can go
        *comm = (ncclComm_t)c;
        return ncclSuccess;
    }
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
        if(c->shm) c->shm->h->broken.store(1);
        else {
            {
                std::lock_guard<std::mutex> g(c->w->mu);
                c->w->broken = true;
            }
            c->w->cv.notify_all();
        }
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
    if(!c || root < 0 || root >= c->world()) return ncclInvalidArgument;
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
