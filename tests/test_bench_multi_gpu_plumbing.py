"""bench.py --gpus N without a device: the command re-executes itself under torch.distributed.run, the ranks find each
other through the file rendezvous, reduce a timing, and rank 0 prints ONE line claiming the right GPU count.  (--dry-run
stops short of touching a device; the measuring path itself is covered on the GPU.)  Also: the rendezvous primitives and
the world / --gpus consistency check."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, env=None, timeout=240):
    e = dict(os.environ)
    e.update(env or {})
    e.pop("WORLD_SIZE", None) if env is None else None
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=e)


def test_plain_command_with_gpus_2_spawns_two_ranks_and_prints_one_line():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--rows", "2000", "--dim", "32", "--no-cpu", "--dist-backend", "gloo"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1] and d["dry_run"] is True
    assert d["config"]["global_queries_per_step"] == 2 * d["config"]["queries_per_step_per_gpu"]
    assert abs(d["ms_per_step"] - 2.0) < 1e-6  # max over ranks of (rank + 1) ms
    for key in ("metric", "value", "unit", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--dry-run"], cwd=ROOT, capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=2 but --gpus 1" in (p.stderr + p.stdout)


def test_file_rendezvous_collectives(tmp_path):
    from lantern_amd.rendezvous import FileRendezvous, RendezvousTimeout

    world = 3
    out = [None] * world

    def rank_main(r):
        rdv = FileRendezvous(r, world, path=str(tmp_path / "rdv"), timeout=30)
        rdv.barrier()
        got = rdv.allgather(f"rank{r}".encode())
        uid = rdv.broadcast(b"U" * 128 if r == 0 else None, 0)
        mx = rdv.max_float(0.5 * (r + 1))
        buf = np.zeros(10 + 20 + 5, dtype=np.uint8)
        offs, cnts = [0, 10, 30], [10, 20, 5]
        buf[offs[r]:offs[r] + cnts[r]] = r + 1
        for _ in range(6):  # repeated exchanges: old files are deleted, the directory stays bounded
            rdv.allgatherv(buf, offs, cnts)
        out[r] = (got, uid, mx, buf.copy(), len(os.listdir(rdv.path)))

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for r in range(world):
        got, uid, mx, buf, nfiles = out[r]
        assert got == [b"rank0", b"rank1", b"rank2"] and uid == b"U" * 128 and mx == 1.5
        assert buf.tolist() == [1] * 10 + [2] * 20 + [3] * 5
        assert nfiles <= 3 * world
    lone = FileRendezvous(0, 2, path=str(tmp_path / "lonely"), timeout=0.3)
    with pytest.raises(RendezvousTimeout):
        lone.barrier()


def test_eight_ranks_rendezvous_shard_the_rows_and_print_one_line():
    """The first 8-GPU run should be boring: eight ranks find each other, every rank synthesises only ITS shard (the shards tile
    the row range, host memory per rank is an eighth of the set), the timing reduction is the max over ranks, and rank 0 prints
    ONE line with n_gpus = 8.  (No device is touched: --dry-run.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--dry-run", "--rows", "80000", "--dim", "64", "--no-cpu", "--dist-backend", "files"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == list(range(8))
    shards = d["dry_run_shards"]
    assert [s["rank"] for s in shards] == list(range(8)) and [s["local_rank"] for s in shards] == list(range(8))
    assert shards[0]["rows"][0] == 0 and shards[-1]["rows"][1] == 80000 and d["rows_covered"] == 80000
    assert all(a["rows"][1] == b["rows"][0] for a, b in zip(shards, shards[1:]))  # the shards tile the range
    assert all(s["host_bytes"] == (s["rows"][1] - s["rows"][0]) * 64 * 4 for s in shards)  # a rank holds its shard only
    assert len({s["checksum"] for s in shards}) == 8  # ... and draws it from its own stream
    assert abs(d["ms_per_step"] - 8.0) < 1e-6 and d["config"]["global_queries_per_step"] == 8 * d["config"]["queries_per_step_per_gpu"]
