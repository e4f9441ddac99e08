"""The exchange step of the work-sharded build without a device: the in-place all-gather with per-rank
sizes over (a) the in-process hub, one thread per rank, and (b) two gloo processes through the library's host
transport -- the same C entry points lantern_gpu_add_sharded drives (lantern_amd/csrc/comm.cpp); plus the balanced
split and the failure behaviour (a missing peer is an error string after the deadline, never a hang).  The build
itself on top of this exchange is covered by the -m gpu tests (tests/test_gpu_sharded_build.py)."""
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from lantern_amd import build, capi

    build.build()
    capi.lib()
    return capi


def test_shard_range_is_a_balanced_partition(capi):
    for n in (0, 1, 7, 8, 101, 8192, 1_000_003):
        for w in (1, 2, 3, 8):
            r = [capi.shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1


def fill(rank, offsets, counts):
    extent = max(o + c for o, c in zip(offsets, counts))
    buf = np.zeros(extent, dtype=np.uint8)
    buf[offsets[rank]: offsets[rank] + counts[rank]] = (np.arange(counts[rank]) * 7 + rank * 31 + 1) % 251
    return buf


def expected(world, offsets, counts):
    out = np.zeros(max(o + c for o, c in zip(offsets, counts)), dtype=np.uint8)
    for r in range(world):
        out[offsets[r]: offsets[r] + counts[r]] = (np.arange(counts[r]) * 7 + r * 31 + 1) % 251
    return out


@pytest.mark.parametrize("world,counts", [(2, [5, 9]), (3, [16, 0, 3]), (4, [1, 1, 1, 1]), (3, [100_000, 70_001, 3])])
def test_in_process_hub_allgatherv(capi, world, counts):
    offsets = list(np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(int))
    comms = capi.Comm.local_world(world)
    assert [c.rank for c in comms] == list(range(world)) and all(c.world == world for c in comms)
    got, errs = {}, []

    def run(c):
        try:
            for _ in range(3):  # the hub is reusable: one rendezvous per exchange
                buf = fill(c.rank, offsets, counts)
                c.allgatherv_host(buf, offsets, counts)
                got[c.rank] = buf
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=run, args=(c,)) for c in comms]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    want = expected(world, offsets, counts)
    assert all(np.array_equal(got[r], want) for r in range(world))


def test_missing_peer_is_an_error_after_the_deadline(capi):
    comms = capi.Comm.local_world(2)
    comms[0].set_timeout(0.5)
    t0 = time.time()
    with pytest.raises(capi.LanternGpuError, match="all-gather failed or timed out"):
        comms[0].allgatherv_host(np.zeros(8, dtype=np.uint8), [0, 4], [4, 4])
    assert time.time() - t0 < 10


def test_rccl_transport_needs_a_device(capi):
    if capi.device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(capi.LanternGpuError, match="no HIP device|cannot load librccl"):
        capi.Comm.rccl(0, 1, bytes(capi.COMM_ID_BYTES))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


RANK_SCRIPT = r"""
import os, sys
import numpy as np
import torch.distributed as dist
rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dist.init_process_group("gloo", rank=rank, world_size=world)
from lantern_amd import capi, sharded
comm = sharded.host_comm()
assert (comm.rank, comm.world) == (rank, world)
counts = [1000, 37]
offsets = [0, 1000]
buf = np.zeros(1037, dtype=np.uint8)
buf[offsets[rank]: offsets[rank] + counts[rank]] = (np.arange(counts[rank]) * 7 + rank * 31 + 1) % 251
comm.allgatherv_host(buf, offsets, counts)
np.save(os.path.join(out, f"buf{rank}.npy"), buf)
dist.barrier()
dist.destroy_process_group()
"""


def test_world_size_2_gloo_host_transport(capi, tmp_path):
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", RANK_SCRIPT, str(r), "2", str(tmp_path)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    want = expected(2, [0, 1000], [1000, 37])
    assert np.array_equal(np.load(tmp_path / "buf0.npy"), want) and np.array_equal(np.load(tmp_path / "buf1.npy"), want)
