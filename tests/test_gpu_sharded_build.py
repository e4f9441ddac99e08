"""Work-sharded index build on the device: a world of 2-3 ranks builds ONE graph, every rank ends
up with a replica that is bit-identical to the single-GPU build of the same rows with the same batch plan -- which
tests/test_gpu_parity.py pins edge-for-edge on the CPU oracle.  The test box has one GPU, so the ranks share it:
threads of this process over the in-process hub, and two gloo processes over the host transport.  The RCCL
transport refuses two ranks on one device, so it is exercised at world size 1 here (library binding, communicator,
grouped broadcasts on the index stream) and at 2/4/8 ranks by `bench.py --gpus N`.
"""
import os
import socket
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LABEL0 = 1


@pytest.fixture(scope="module")
def capi():
    from lantern_amd import capi

    capi.lib()
    assert capi.device_count() > 0, "no HIP device: the gpu tests need a real MI355X"
    return capi


def rand_rows(rng, n, d, metric):
    if metric == "hamming":
        return rng.integers(0, 2**32, size=(n, d), dtype=np.uint32)
    return rng.standard_normal((n, d), dtype=np.float32)


def single_build(capi, metric, base, labels, M, efc, plan, quant="f32"):
    ix = capi.GpuIndex(metric, base.shape[1], M=M, ef_construction=efc, ef=32, seed=21, quantization=quant)
    ix.set_add_batch(*plan)
    ix.add_many(labels, base)
    ix.flush()
    return ix


def graphs_equal(a, b):
    return (a["entry_slot"] == b["entry_slot"] and a["max_level"] == b["max_level"] and
            all(np.array_equal(a[k], b[k]) for k in ("levels", "labels", "upper_off", "nbr0", "upper_nbr")))


def threaded_world(capi, world, metric, base, labels, cuts, M, efc, plan, quant="f32"):
    """`world` ranks as threads on one GPU over the in-process hub; cuts[r]:cuts[r+1] is rank r's shard."""
    comms = capi.Comm.local_world(world)
    out, errs = [None] * world, []

    def run(r):
        try:
            comms[r].set_timeout(120)
            ix = capi.GpuIndex(metric, base.shape[1], M=M, ef_construction=efc, ef=32, seed=21, quantization=quant)
            ix.set_add_batch(*plan)
            ix.add_sharded(comms[r], labels[cuts[r]:cuts[r + 1]], base[cuts[r]:cuts[r + 1]])
            out[r] = ix
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errs.append((r, repr(e)))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    return out, comms


# shapes: short rows (generic kernels, G = 16), 768-d f32 (register-resident connect / re-prune), cosine with the
# column-slab re-prune, hamming, f16 storage; plans small enough that most batches are split (b >= 8 per rank)
@pytest.mark.parametrize("metric,n,d,M,efc,plan,quant", [
    ("l2sq", 3000, 64, 8, 40, (256, 8), "f32"),
    ("l2sq", 2500, 768, 16, 64, (512, 8), "f32"),
    ("cos", 1500, 1100, 12, 40, (256, 8), "f32"),
    ("hamming", 2000, 24, 6, 32, (128, 8), "f32"),
    ("l2sq", 1500, 768, 16, 48, (256, 8), "f16"),
])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_build_is_bit_identical_to_single_gpu_build(capi, metric, n, d, M, efc, plan, quant, world):
    rng = np.random.default_rng(n + d)
    base = rand_rows(rng, n, d, metric)
    labels = np.arange(n, dtype=np.uint64) + LABEL0
    ref = single_build(capi, metric, base, labels, M, efc, plan, quant)
    g_ref = ref.export_graph()
    cuts = [capi.shard_range(n, world, r)[0] for r in range(world)] + [n]
    ixs, comms = threaded_world(capi, world, metric, base, labels, cuts, M, efc, plan, quant)
    for r, ix in enumerate(ixs):
        assert len(ix) == n
        assert graphs_equal(ix.export_graph(), g_ref), f"rank {r}'s replica differs from the single-GPU graph"
        assert ix.checksum() == ref.checksum()
    # the work really was shared: the ranks' distance evaluations add up to (about) the single-GPU count, and
    # nobody did all of it; exchanges happened
    single = ref.counters()["add_dist_evals"]
    parts = [ix.counters()["add_dist_evals"] for ix in ixs]
    assert max(parts) < 0.8 * single
    assert all(c.stats()["collectives"] > 2 and c.stats()["bytes_received"] > 0 for c in comms)
    # and the replicas answer queries identically
    q = rand_rows(rng, 16, d, metric)
    l0, d0, _ = ref.search_batch(q, 5)
    for ix in ixs:
        l1, d1, _ = ix.search_batch(q, 5)
        assert np.array_equal(l0, l1) and np.array_equal(d0, d1)


def test_sharded_build_matches_oracle_edge_for_edge(capi, oracle):
    # direct pin on the CPU oracle (not only via the single-GPU build)
    n, d, M, efc, plan = 1800, 48, 8, 40, (128, 8)
    rng = np.random.default_rng(5)
    base = rng.standard_normal((n, d), dtype=np.float32)
    labels = np.arange(n, dtype=np.uint64) + LABEL0
    ora = oracle.OracleIndex("l2sq", d, M=M, ef_construction=efc, ef=32, seed=21, sum_mode=oracle.SUM_WAVE64)
    ora.add_planned(labels, base, max_batch=plan[0], min_ratio=plan[1])
    ixs, _ = threaded_world(capi, 2, "l2sq", base, labels, [0, 900, n], M, efc, plan)
    go = ora.export_graph()
    for ix in ixs:
        assert graphs_equal(ix.export_graph(), go)


def test_ragged_and_empty_shards(capi):
    # shard sizes need not be balanced; a rank may contribute nothing and still takes its share of the work
    n, d = 1200, 32
    rng = np.random.default_rng(9)
    base = rng.standard_normal((n, d), dtype=np.float32)
    labels = np.arange(n, dtype=np.uint64) + LABEL0
    ref = single_build(capi, "l2sq", base, labels, 8, 32, (128, 8))
    ixs, _ = threaded_world(capi, 3, "l2sq", base, labels, [0, 1000, 1000, n], 8, 32, (128, 8))
    for ix in ixs:
        assert ix.checksum() == ref.checksum()


def test_sharded_append_to_existing_replicas(capi):
    # add_sharded on non-empty (identical) replicas appends; replicas of different sizes are refused
    n, d = 1600, 40
    rng = np.random.default_rng(11)
    base = rng.standard_normal((n, d), dtype=np.float32)
    labels = np.arange(n, dtype=np.uint64) + LABEL0
    ref = capi.GpuIndex("l2sq", d, M=8, ef_construction=32, ef=32, seed=21)
    ref.set_add_batch(128, 8)
    ref.add_many(labels[:600], base[:600])
    ref.flush()
    ref.add_many(labels[600:], base[600:])
    ref.flush()
    comms = capi.Comm.local_world(2)
    out, errs = [None, None], []

    def run(r):
        try:
            ix = capi.GpuIndex("l2sq", d, M=8, ef_construction=32, ef=32, seed=21)
            ix.set_add_batch(128, 8)
            ix.add_many(labels[:600], base[:600])
            ix.flush()
            lo, hi = (600, 1100) if r == 0 else (1100, n)
            ix.add_sharded(comms[r], labels[lo:hi], base[lo:hi])
            out[r] = ix
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert out[0].checksum() == ref.checksum() == out[1].checksum()


def test_missing_peer_times_out_instead_of_hanging(capi):
    comms = capi.Comm.local_world(2)
    comms[0].set_timeout(1.0)
    ix = capi.GpuIndex("l2sq", 8, M=4, ef_construction=16, ef=16, seed=1)
    rows = np.random.default_rng(0).standard_normal((64, 8), dtype=np.float32)
    with pytest.raises(capi.LanternGpuError, match="all-gather failed or timed out"):
        ix.add_sharded(comms[0], np.arange(64, dtype=np.uint64) + 1, rows)  # rank 1 never shows up


def test_rccl_transport_world_of_one(capi):
    """The RCCL binding end to end on the one GPU of this box: dlopen, unique id, ncclCommInitRank, the grouped
    in-place broadcasts on the index's stream, the deadline-bounded wait.  (RCCL refuses a second rank on the
    same device; the multi-rank exchange itself is the same code path with world > 1.)"""
    n, d = 1500, 64
    rng = np.random.default_rng(2)
    base = rng.standard_normal((n, d), dtype=np.float32)
    labels = np.arange(n, dtype=np.uint64) + LABEL0
    ref = single_build(capi, "l2sq", base, labels, 8, 40, (128, 8))
    uid = capi.Comm.unique_id()
    assert len(uid) == capi.COMM_ID_BYTES and any(uid)
    comm = capi.Comm.rccl(0, 1, uid)
    assert (comm.rank, comm.world) == (0, 1)
    # the exchange primitive on a device buffer (degenerate at world 1, but it goes through RCCL's group calls)
    from lantern_amd import hip

    buf = hip.Buffer.from_numpy(np.arange(256, dtype=np.uint8))
    comm.allgatherv_device(buf.ptr, [0], [256])
    assert np.array_equal(buf.download(256, np.uint8), np.arange(256, dtype=np.uint8))
    ix = capi.GpuIndex("l2sq", d, M=8, ef_construction=40, ef=32, seed=21)
    ix.set_add_batch(128, 8)
    ix.add_sharded(comm, labels, base)
    assert ix.checksum() == ref.checksum()
    comm.free()


# ---- two PROCESSES on the one GPU, exchange over torch.distributed (gloo) through the host transport -----------
# The ranks are fresh interpreters (tests/gloo_build_rank.py): they import torch BEFORE the HIP library so that each
# process has one HIP runtime; this pytest process never imports torch.
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_gloo_processes_share_one_build(capi, tmp_path):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=root)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "gloo_build_rank.py"), str(r), "2", str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    ref = np.load(tmp_path / "ref.npy")[0]
    assert np.load(tmp_path / "sum0.npy")[0] == ref == np.load(tmp_path / "sum1.npy")[0]
