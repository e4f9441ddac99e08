"""The N>1 path on CPU: world_size-2 gloo processes run lantern_amd/sharded.py (split, per-rank search on
a replica, gather, max-over-ranks).  The per-rank search engine here is the oracle -- the sharding logic
is what is under test; the GPU search itself is covered by the -m gpu parity tests."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lantern_amd import sharded
    from oracle import binding as oracle

    rng = np.random.default_rng(1)
    base = rng.standard_normal((600, 16), dtype=np.float32)
    queries = rng.standard_normal((101, 16), dtype=np.float32)  # ragged: 51 + 50
    ix = oracle.OracleIndex("l2sq", 16, M=8, ef_construction=32, ef=32, seed=5)
    ix.add_many(np.arange(600) + 1, base)  # every rank builds the same replica (deterministic)

    def search(qs, k):
        lab, dst, _, _, _ = ix.search_batch(qs, k)
        return lab, dst

    lab, dst = sharded.sharded_search(search, queries, 5)
    t = sharded.max_over_ranks(1.0 + rank)
    if rank == 0:
        np.save(os.path.join(out_dir, "lab.npy"), lab)
        np.save(os.path.join(out_dir, "dst.npy"), dst)
        np.save(os.path.join(out_dir, "t.npy"), np.array([t]))
    dist.barrier()
    dist.destroy_process_group()


def test_split_range_is_balanced_and_covers():
    from lantern_amd import sharded

    for n in (0, 1, 7, 8, 101, 8192):
        for w in (1, 2, 3, 8):
            r = sharded.split_range(n, w)
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1


def test_world_size_2_gloo_sharded_search(tmp_path):
    import torch.multiprocessing as mp

    from oracle import binding as oracle

    oracle.build()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    lab = np.load(tmp_path / "lab.npy")
    dst = np.load(tmp_path / "dst.npy")
    assert float(np.load(tmp_path / "t.npy")[0]) == 2.0  # max over ranks of (1, 2)
    rng = np.random.default_rng(1)
    base = rng.standard_normal((600, 16), dtype=np.float32)
    queries = rng.standard_normal((101, 16), dtype=np.float32)
    ix = oracle.OracleIndex("l2sq", 16, M=8, ef_construction=32, ef=32, seed=5)
    ix.add_many(np.arange(600) + 1, base)
    ref_lab, ref_dst, _, _, _ = ix.search_batch(queries, 5)
    assert np.array_equal(lab, ref_lab) and np.array_equal(dst, ref_dst)
