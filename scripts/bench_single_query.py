#!/usr/bin/env python
"""BASELINE config[1]: 100k x 128 f32 L2sq, M=16 ef=64 k=10, queries issued ONE AT A TIME through usearch_search_ef
(the call a PostgreSQL backend makes per scan: scan.c:220-228).  Reports the per-query wall latency through the host ABI,
the search kernel's own time for a lone query (HIP events on a stream), and the CPU port's per-query latency on one
thread on the same graph (a PostgreSQL backend is single-threaded: utils.c:66).

    python scripts/bench_single_query.py [--rows 100000 --dim 128 --queries 2000] > profiles/<name>.json
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
    p.add_argument("--queries", type=int, default=2000)
    p.add_argument("--ef", type=int, default=64)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--no-cpu", action="store_true")
    a = p.parse_args()
    from lantern_amd import capi, hip

    base = np.random.default_rng(1).standard_normal((a.rows, a.dim), dtype=np.float32)  # C2 seeds 1 / 2
    queries = np.random.default_rng(2).standard_normal((a.queries, a.dim), dtype=np.float32)
    ix = capi.GpuIndex(a.metric, a.dim, M=16, ef_construction=128, ef=a.ef, seed=42)
    ix.reserve(a.rows)
    t0 = time.time()
    ix.add_many(np.arange(a.rows, dtype=np.uint64) + 1, base)
    ix.flush()
    hip.synchronize()
    t_build = time.time() - t0
    for q in queries[:50]:
        ix.search(q, a.k)
    lat = []
    res = []
    for q in queries:
        t0 = time.perf_counter()
        lab, dst = ix.search(q, a.k)
        lat.append(time.perf_counter() - t0)
        res.append(lab.copy())
    lat = np.array(lat) * 1e6
    # the kernel alone: one query per launch on a stream, HIP events
    stream = hip.Stream()
    dq = hip.Buffer.from_numpy(hip.padded_rows(queries, False))
    d_lab, d_dst = hip.Buffer(a.k * 8), hip.Buffer(a.k * 4)
    d_D, d_E = hip.Buffer(8), hip.Buffer(8)
    row_bytes = hip.padded_rows(queries[:1], False).shape[1] * 4
    shapes = {}
    for name, waves in (("classic_8_waves", 8), ("latency_bound_shape", 0)):  # 0 = automatic: the lone-query shape of walk_spec.hpp
        ix.set_search_shape(waves, 0)
        kern, Ds, Es = [], [], []
        for i in range(min(a.queries, 500)):
            s, e = hip.Event(), hip.Event()
            s.record(stream.handle)
            ix.search_batch_device(dq.ptr + i * row_bytes, 1, a.k, a.ef, 0, d_lab.ptr, d_dst.ptr, None, None, d_D.ptr, d_E.ptr, stream.handle)
            e.record(stream.handle)
            stream.synchronize()
            kern.append(s.elapsed_ms(e) * 1e3)
            Ds.append(int(d_D.download(1, np.uint64)[0]))
            Es.append(int(d_E.download(1, np.uint64)[0]))
        shapes[name] = {"us_per_query_kernel": {"mean": float(np.mean(kern)), "p50": float(np.median(kern))},
                        "us_per_hop_kernel": float(np.mean(kern) / max(np.mean(Es), 1))}
    ix.set_search_shape(0, 0)
    blab, _, _ = ix.search_batch(queries, a.k)
    same = float(np.mean([np.array_equal(r, b[:len(r)]) for r, b in zip(res, blab)]))
    out = {"config": f"{a.rows}x{a.dim} f32 {a.metric} M=16 efc=128 ef={a.ef} k={a.k}, one query per usearch_search_ef call",
           "queries": a.queries, "us_per_query_wall": {"mean": float(lat.mean()), "p50": float(np.median(lat)), "p99": float(np.percentile(lat, 99))},
           "qps_single_stream": float(1e6 / lat.mean()),
           "kernel_only": shapes,
           "hops_per_query": float(np.mean(Es)), "dist_evals_per_query": float(np.mean(Ds)),
           "identical_to_batch_search": same, "build_vectors_per_s": a.rows / t_build}
    if not a.no_cpu:
        from oracle import binding as oracle

        native = oracle.build_native() and oracle.use_native(True)
        g = ix.export_graph()
        ora = oracle.OracleIndex.from_graph(a.metric, base, g, 16, 128, a.ef, 42, oracle.SUM_FAST)
        ora.search_batch(queries[:100], a.k, a.ef, 1)
        t0 = time.perf_counter()
        _, _, slots, _, _ = ora.search_batch(queries, a.k, a.ef, 1)
        cpu_us = (time.perf_counter() - t0) / a.queries * 1e6
        out["cpu_port_us_per_query_1_thread"] = cpu_us
        out["cpu_port_build"] = "gcc -O3 -march=native" if native else "gcc -O3 -march=x86-64-v3"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
