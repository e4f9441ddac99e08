// comm.hpp -- the exchange step of the work-sharded index build.
//
// One process (or thread) per GPU; every rank holds a replica of the graph and of the vector block in its own
// HBM.  The only exchanged data are (1) the vector shards at the start, (2) per batch, the top-M neighbour
// lists the ranks selected for their share of the new nodes, (3) per batch, the adjacency rows the ranks
// re-pruned.  All three are in-place "all-gather with per-rank sizes" operations on a device buffer:
//
//   RCCL transport   ncclBroadcast x world inside one ncclGroupStart/End (= all-gather-v) on the index's HIP
//                    stream; data never leaves the devices (xGMI).  librccl.so.1 is dlopen'ed on first use so
//                    that hosts which never shard (a PostgreSQL backend) do not map it.
//   host transport   the caller supplies the all-gather over a HOST buffer (MPI, gloo, sockets); the library
//                    stages D2H / H2D around it.  Also used by the built-in in-process hub
//                    (lantern_gpu_comm_init_local: one thread per rank), which the single-GPU test box uses to
//                    run a world of 2-3 ranks against one device.
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "../../include/lantern_gpu.h"

namespace lgpu {

struct LocalHub;  // in-process rendezvous shared by the ranks of lantern_gpu_comm_init_local

struct Comm
{
    int  rank = 0, world = 1;
    bool rccl = false;
    // RCCL
    void *nccl_comm = nullptr;
    // host transport
    lantern_gpu_allgatherv_fn fn = nullptr;
    void                     *fn_ctx = nullptr;
    std::shared_ptr<LocalHub> hub;
    std::vector<char>         stage;
    double      timeout_s = 180.0;  // a collective that does not complete in this time is an error (never a hang)
    uint64_t    bytes_exchanged = 0, collectives = 0;
    std::string err;

    // In-place all-gather on a DEVICE buffer: on entry bytes [off[rank], off[rank] + cnt[rank]) are valid,
    // on return (after wait()) all world segments are.  Enqueued on `st` for RCCL, synchronous for the host transport.
    bool allgatherv_device(void *d_buf, const size_t *off, const size_t *cnt, hipStream_t st);
    // The same over a HOST buffer (small metadata: shard sizes).
    bool allgatherv_host(void *h_buf, const size_t *off, const size_t *cnt);
    // deadline-bounded hipStreamSynchronize
    bool wait(hipStream_t st);
};

}  // namespace lgpu
