"""One rank of tests/test_gpu_sharded_build.py::test_two_gloo_processes_share_one_build.

    python tests/gloo_build_rank.py RANK WORLD OUT_DIR [rows]     (MASTER_ADDR / MASTER_PORT in the environment)

Builds ONE index with lantern_gpu_add_sharded together with its peers; the exchange runs over torch.distributed's
gloo backend through the library's host transport.  Every rank writes its replica's checksum; rank 0 also builds the
same rows alone and writes that checksum.  With `rows` as a fourth argument the build is lantern_gpu_add_row_sharded
(test_two_gloo_processes_share_one_row_sharded_build): the replicas' checksums must agree with each other, and every rank
writes the labels its index returns for the rows themselves as queries.
"""
import os
import sys

import numpy as np


def main():
    rank, world, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    by_rows = len(sys.argv) > 4 and sys.argv[4] == "rows"
    import torch  # noqa: F401  -- before the HIP library: one HIP runtime per process (lantern_amd/capi.py note)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lantern_amd import capi, sharded

    n, d = 2400, 96
    rng = np.random.default_rng(13)
    base = rng.standard_normal((n, d), dtype=np.float32)
    labels = np.arange(n, dtype=np.uint64) + 1
    comm = sharded.host_comm()
    comm.set_timeout(120)
    lo, hi = capi.shard_range(n, world, rank)
    ix = capi.GpuIndex("l2sq", d, M=8, ef_construction=40, ef=32, seed=21)
    ix.set_add_batch(256, 8)
    if by_rows:
        ix.add_row_sharded(comm, labels[lo:hi], base[lo:hi])
        found, _, _ = ix.search_batch(base, 1, 64)
        np.save(os.path.join(out_dir, f"self{rank}.npy"), found[:, 0])
    else:
        ix.add_sharded(comm, labels[lo:hi], base[lo:hi])
    np.save(os.path.join(out_dir, f"sum{rank}.npy"), np.array([ix.checksum()], dtype=np.uint64))
    if rank == 0:
        ref = capi.GpuIndex("l2sq", d, M=8, ef_construction=40, ef=32, seed=21)
        ref.set_add_batch(256, 8)
        ref.add_many(labels, base)
        ref.flush()
        np.save(os.path.join(out_dir, "ref.npy"), np.array([ref.checksum()], dtype=np.uint64))
        if by_rows:
            found, _, _ = ref.search_batch(base, 1, 64)
            np.save(os.path.join(out_dir, "self_ref.npy"), found[:, 0])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
