"""One rank of a work-sharded index build over RCCL, as its own process:

    python -m lantern_amd.sharded_build --rank R --world W --rendezvous DIR [--device D] [--rows N --dim d ...]

No PyTorch in here: the process binds ROCm's HIP runtime and ROCm's RCCL only.  Rank 0 draws the RCCL unique id and
publishes it as DIR/uid (atomic rename); the peers pick it up from there (same node).  Every rank generates the
benchmark's synthetic rows, keeps its shard, and calls lantern_gpu_add_sharded; the result line (JSON on stdout)
carries the build time, this replica's graph checksum and the exchange statistics.  bench.py --gpus N launches one
of these per GPU AFTER its own measurement, so a failure here can never cost the benchmark line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def publish_uid(path: str, uid: bytes):
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(uid)
    os.replace(tmp, path)


def await_uid(path: str, nbytes: int, timeout: float) -> bytes:
    t0 = time.time()
    while time.time() - t0 < timeout:
        if os.path.exists(path):
            data = open(path, "rb").read()
            if len(data) == nbytes:
                return data
        time.sleep(0.05)
    raise TimeoutError(f"no RCCL unique id at {path} after {timeout} s")


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--rank", type=int, required=True)
    p.add_argument("--world", type=int, required=True)
    p.add_argument("--rendezvous", required=True, help="a directory all ranks of the node can see")
    p.add_argument("--device", type=int, default=None, help="HIP device (default: rank)")
    p.add_argument("--rows", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--metric", default="l2sq")
    p.add_argument("--M", type=int, default=16)
    p.add_argument("--efc", type=int, default=128)
    p.add_argument("--ef", type=int, default=64)
    p.add_argument("--add-batch", type=int, default=8192)
    p.add_argument("--quant", default="f32")
    p.add_argument("--data", default="gaussian")
    p.add_argument("--data-scale", type=float, default=1.0)
    p.add_argument("--timeout", type=float, default=120.0, help="deadline of every collective")
    a = p.parse_args()

    from lantern_amd import capi, hip, synth

    ndev = capi.device_count()
    assert ndev > 0, "no HIP device"
    dev = a.device if a.device is not None else a.rank % ndev
    hip.set_device(dev)

    t0 = time.time()
    base = synth.base_rows(a.data, a.rows, a.dim)
    if a.data_scale != 1.0:
        base *= np.float32(a.data_scale)
    lo, hi = capi.shard_range(a.rows, a.world, a.rank)
    shard = np.ascontiguousarray(base[lo:hi])
    labels = np.arange(lo, hi, dtype=np.uint64) + 1
    del base
    t_gen = time.time() - t0

    uid_path = os.path.join(a.rendezvous, "uid")
    if a.rank == 0:
        uid = capi.Comm.unique_id()
        publish_uid(uid_path, uid)
    else:
        uid = await_uid(uid_path, capi.COMM_ID_BYTES, 300.0)
    comm = capi.Comm.rccl(a.rank, a.world, uid)
    comm.set_timeout(a.timeout)

    ix = capi.GpuIndex(a.metric, a.dim, M=a.M, ef_construction=a.efc, ef=a.ef, seed=42, quantization=a.quant)
    ix.reserve(a.rows)
    ix.set_add_batch(a.add_batch, 16)
    # a first exchange as the start barrier (datagen times differ between ranks), then the timed collective build
    token = np.zeros(8 * a.world, dtype=np.uint8)
    comm.allgatherv_host(token, [8 * r for r in range(a.world)], [8] * a.world)
    hip.synchronize()
    t0 = time.time()
    ix.add_sharded(comm, labels, shard)
    hip.synchronize()
    t_build = time.time() - t0
    out = {
        "rank": a.rank, "world": a.world, "device": dev, "rows": a.rows, "dim": a.dim,
        "seconds": t_build, "vectors_per_s": a.rows / t_build, "checksum": f"{ix.checksum():016x}",
        "size": len(ix), "exchange": comm.stats(), "counters": ix.counters(), "datagen_seconds": t_gen,
        "transport": "RCCL all-gather-v (grouped ncclBroadcast) on the index stream",
    }
    print("SHARDED_BUILD " + json.dumps(out), flush=True)
    comm.free()


if __name__ == "__main__":
    main()
