"""The RCCL transport of the sharded builds (lantern_amd/csrc/comm.cpp) with MORE THAN ONE RANK, on a one-GPU box: the test double
tests/fake_rccl/fake_rccl.cpp implements the nine ncclXxx entry points comm.cpp binds across the threads of one process, and
LANTERN_GPU_RCCL_LIB points the library at it.  Real RCCL refuses two ranks on one device, so until a multi-GPU box runs
`bench.py --gpus N` this is the only execution the multi-rank RCCL code path gets: communicator bring-up per rank, the metadata
all-gather through the device path, the grouped in-place broadcasts with ragged (and empty) segments on the index streams -- with the
ranks as threads of one process (run_world.py), and with the ranks as the PROCESSES `bench.py --gpus N` launches (the double's
shared-memory mode), i.e. the driver's own multi-GPU command end to end through the RCCL path."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests", "fake_rccl")


@pytest.fixture(scope="module")
def fake_lib():
    from lantern_amd import build, capi

    build.build()
    assert capi.device_count() > 0
    so = os.path.join(HERE, "librccl_fake.so")
    src = os.path.join(HERE, "fake_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call([build._hipcc(), "-O2", "-shared", "-fPIC", "-o", so, src])
    return so


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_builds_through_the_rccl_transport(fake_lib, world):
    env = dict(os.environ, LANTERN_GPU_RCCL_LIB=fake_lib)
    p = subprocess.run([sys.executable, os.path.join(HERE, "run_world.py"), str(world)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    line = next((json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")), None)
    assert p.returncode == 0 and line, (p.stdout[-2000:], p.stderr[-2000:])
    assert line["uid_is_the_doubles"], "comm.cpp did not bind the test double"
    assert not line["comm_errors"] and line["rccl_ranks_seen"] == world
    assert not line["allgatherv_errors"] and all(line["allgatherv_ok"])
    ws = line["work_sharded"]
    assert not ws["errors"], ws["errors"]
    assert all(r["checksum"] == ws["reference_checksum"] and r["size"] == 2400 for r in ws["ranks"]), "a replica differs from the one-GPU graph"
    assert max(r["add_dist_evals"] for r in ws["ranks"]) < 0.9 * ws["reference_add_dist_evals"], "the work was not shared"
    rs = line["row_sharded"]
    assert not rs["errors"], rs["errors"]
    assert len({r["checksum"] for r in rs["ranks"]}) == 1 and all(r["size"] == 2400 for r in rs["ranks"])
    assert all(abs(r["recall"] - rs["one_gpu_recall"]) <= 0.03 for r in rs["ranks"]), rs
    assert all(s["collectives"] > 4 and s["bytes_received"] > 0 for s in line["stats"])
    if world >= 3:
        assert line["shards"][1] == 0, "the empty shard"


@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_ranks_bring_up_and_build_through_the_rccl_path(fake_lib, ranks):
    """`bench.py --gpus N` as the driver launches it (N = 2 and the full node's 8), with the double in its multi-process mode: the unique id travels over the
    rendezvous, every rank's ncclCommInitRank runs under its deadline thread, the probe all-gather and the verdict exchange pass, the
    collective build runs through the RCCL transport (grouped broadcasts on the index stream) and leaves identical replicas.  On a
    one-GPU box the ranks share the device, which real RCCL refuses -- hence the explicit override for the double."""
    env = dict(os.environ, LANTERN_GPU_RCCL_LIB=fake_lib, FAKE_RCCL_MULTIPROCESS="1", LANTERN_BENCH_RCCL_ON_SHARED_DEVICE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--rows", "120000", "--dim", "128", "--steps", "3", "--queries", "2048",
           "--no-cpu", "--no-pmc", "--no-secondary", "--build-quality-rows", "0", "--truth-queries", "256"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    line = next((json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")), None)
    assert p.returncode == 0 and line, (p.stdout[-2000:], p.stderr[-3000:])
    cb = line["collective_build"]
    assert line["n_gpus"] == ranks and cb["transport_used"] == "rccl" and cb["rccl_ranks_seen"] == ranks, cb
    assert cb["replicas_identical"] and cb["collectives"] > 4 and all(b > 0 for b in cb["bytes_received_per_rank"]), cb
    assert line["value"] > 0 and line["recall_at_10"] > 0.3
