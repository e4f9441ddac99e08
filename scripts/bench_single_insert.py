#!/usr/bin/env python
"""ldb_aminsert's unit of work: ONE vector added to an existing index (usearch_add + flush: the insert is linked before the call
returns, as usearch_add_external's is; insert.c:32-46, hnsw.c:aminsert).  100k x 128 f32 L2sq, M=16 ef_construction=128: wall time
per insert through the host ABI, and the CPU port's on one thread on the same graph (a backend is single-threaded).

    python scripts/bench_single_insert.py [--rows 100000 --dim 128 --inserts 500] > profiles/<name>.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--rows", type=int, default=100_000)
    p.add_argument("--dim", type=int, default=128)
    p.add_argument("--metric", default="l2sq")
    p.add_argument("--inserts", type=int, default=500)
    p.add_argument("--no-cpu", action="store_true")
    a = p.parse_args()
    from lantern_amd import capi, hip

    rng = np.random.default_rng(1)
    base = rng.standard_normal((a.rows, a.dim), dtype=np.float32)
    extra = np.random.default_rng(3).standard_normal((a.inserts + 20, a.dim), dtype=np.float32)
    ix = capi.GpuIndex(a.metric, a.dim, M=16, ef_construction=128, ef=64, seed=42)
    ix.reserve(a.rows + a.inserts + 20)
    ix.add_many(np.arange(a.rows, dtype=np.uint64) + 1, base)
    ix.flush()
    hip.synchronize()
    lat = []
    for i, v in enumerate(extra):
        t0 = time.perf_counter()
        ix.add(a.rows + 1 + i, v)
        ix.flush()
        if i >= 20:
            lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e6
    out = {"config": f"{a.rows}x{a.dim} f32 {a.metric} M=16 ef_construction=128, one usearch_add + flush per vector",
           "inserts": int(lat.size), "us_per_insert_wall": {"mean": float(lat.mean()), "p50": float(np.median(lat)), "p99": float(np.percentile(lat, 99))},
           "inserts_per_s_single_stream": float(1e6 / lat.mean())}
    if not a.no_cpu:
        from oracle import binding as oracle

        ora = oracle.OracleIndex(a.metric, a.dim, M=16, ef_construction=128, ef=64, seed=42, sum_mode=oracle.SUM_WAVE64)
        ora.add_planned(np.arange(a.rows, dtype=np.uint64) + 1, base, 8192, 16)  # the device's batch plan: the same graph, the same generator state
        clat = []
        for i, v in enumerate(extra):
            t0 = time.perf_counter()
            ora.add(a.rows + 1 + i, v)
            if i >= 20:
                clat.append(time.perf_counter() - t0)
        clat = np.array(clat) * 1e6
        out["cpu_port_us_per_insert_1_thread"] = {"mean": float(clat.mean()), "p50": float(np.median(clat))}
        g1, o1 = ix.export_graph(), ora.export_graph()
        out["graph_identical_to_cpu_port_after_inserts"] = bool(np.array_equal(g1["nbr0"], o1["nbr0"]) and np.array_equal(g1["levels"], o1["levels"]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
