#!/usr/bin/env python
"""The serving path at the HEADLINE shape (1M x 768 f32 L2sq, M=16 ef=64 k=10): the scan-side service at 1 .. 1024 connections
(lantern-scan-load --port against ONE resident index), lantern_gpu_search_batch_lane with 8192-query host batches on 1 / 2 / 4
lanes, and one backend's usearch_search_ef latency beside the CPU port's on the same graph.  One JSON object per line.

    python scripts/scan_load_headline.py [--rows 1000000 --dim 768] > profiles/r05_scan_load_1Mx768.jsonl
"""
import argparse
import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # the launcher gives the runtime a hardware queue per service lane
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--rows", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--connections", default="1,16,64,256,1024")
    p.add_argument("--seconds", type=float, default=3.0)
    p.add_argument("--max-batch", type=int, default=1024)
    p.add_argument("--max-wait-us", type=int, default=200)
    a = p.parse_args()
    from lantern_amd import capi, hip, synth

    base = synth.base_rows("gaussian", a.rows, a.dim)
    ix = capi.GpuIndex("l2sq", a.dim, M=16, ef_construction=128, ef=64, seed=42)
    ix.reserve(a.rows)
    t0 = time.time()
    ix.add_many(np.arange(a.rows, dtype=np.uint64) + 1, base)
    ix.flush()
    hip.synchronize()
    print(json.dumps({"index": f"{a.rows}x{a.dim} f32 l2sq M=16 efc=128 ef=64", "build_seconds": time.time() - t0, "lanes_env": os.environ.get("LANTERN_SCAN_LANES")}), flush=True)
    queries = np.random.default_rng(4).standard_normal((8192 * 4, a.dim), dtype=np.float32)
    # ---- one backend: usearch_search_ef, one query per call
    for q in queries[:30]:
        ix.search(q, 10)
    lat = []
    for q in queries[:1000]:
        t0 = time.perf_counter()
        ix.search(q, 10)
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e6
    one = {"leg": "one backend, usearch_search_ef per call", "us_per_call": {"mean": float(lat.mean()), "p50": float(np.median(lat)), "p99": float(np.percentile(lat, 99))}}
    try:
        from oracle import binding as oracle

        native = oracle.build_native() and oracle.use_native(True)
        ora = oracle.OracleIndex.from_graph("l2sq", base, ix.export_graph(), 16, 128, 64, 42, oracle.SUM_FAST)
        ora.search_batch(queries[:50], 10, 64, 1)
        t0 = time.perf_counter()
        ora.search_batch(queries[:1000], 10, 64, 1)
        one["cpu_port_us_per_query_1_thread"] = (time.perf_counter() - t0) / 1000 * 1e6
        one["cpu_port_build"] = "native" if native else "x86-64-v3"
        del ora
    except Exception as ex:  # noqa: BLE001
        one["cpu_error"] = repr(ex)
    print(json.dumps(one), flush=True)
    # ---- host-buffer batches on lanes
    nq = 8192
    batches = [np.ascontiguousarray(queries[i * nq:(i + 1) * nq]) for i in range(4)]
    dq = hip.Buffer.from_numpy(hip.padded_rows(batches[0], False))
    lab, dist, slot = hip.Buffer(nq * 80), hip.Buffer(nq * 40), hip.Buffer(nq * 40)
    for _ in range(2):
        ix.search_batch_device(dq.ptr, nq, 10, 64, 0, lab.ptr, dist.ptr, slot.ptr)
    hip.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ix.search_batch_device(dq.ptr, nq, 10, 64, 0, lab.ptr, dist.ptr, slot.ptr)
    hip.synchronize()
    resident = nq * 10 / (time.perf_counter() - t0)
    ix.search_batch(batches[0], 10, 64)
    t0 = time.perf_counter()
    for i in range(8):
        ix.search_batch(batches[i % 4], 10, 64)
    dt = time.perf_counter() - t0
    print(json.dumps({"leg": "lantern_gpu_search_batch, one call at a time", "queries_per_s": nq * 8 / dt, "device_resident_queries_per_s": resident,
                      "over_device_resident": nq * 8 / dt / resident}), flush=True)
    for lanes in (1, 2, 4):
        per = 6
        for ln in range(lanes):
            ix.search_batch_lane(ln, batches[ln % 4], 10, 64)
        go = threading.Barrier(lanes + 1)

        def worker(ln):
            go.wait()
            for i in range(per):
                ix.search_batch_lane(ln, batches[(ln + i) % 4], 10, 64)

        ts = [threading.Thread(target=worker, args=(ln,)) for ln in range(lanes)]
        [t.start() for t in ts]
        go.wait()
        t0 = time.perf_counter()
        [t.join() for t in ts]
        dt = time.perf_counter() - t0
        print(json.dumps({"leg": f"lantern_gpu_search_batch_lane, {lanes} lanes x {per} host batches of {nq}", "queries_per_s": nq * per * lanes / dt,
                          "over_device_resident": nq * per * lanes / dt / resident}), flush=True)
    # ---- the scan service
    tool = os.path.join(ROOT, "lantern_amd", "lib", "lantern-scan-load")
    srv = capi.ScanServer(index=ix, max_batch=a.max_batch, max_wait_us=a.max_wait_us)
    for c in [int(x) for x in a.connections.split(",")]:
        for extra in ([], ["--client-threads", "8"]) if c >= 64 else ([],):
            before, tb = srv.stats(), srv.timing()
            pr = subprocess.run([tool, "--port", str(srv.port), "--dim", str(a.dim), "--rows", str(a.rows), "--connections", str(c), "--seconds", str(a.seconds),
                                 "--warmup-seconds", "1"] + extra, capture_output=True, text=True, timeout=300)
            after, ta = srv.stats(), srv.timing()
            line = next((json.loads(l) for l in pr.stdout.splitlines() if l.startswith("{")), {"error": (pr.stderr or pr.stdout)[-300:]})
            req, bat = after["requests"] - before["requests"], after["batches"] - before["batches"]
            line.pop("service", None)
            dn = max(ta["requests"] - tb["requests"], 1)
            line["server_side_us"] = {k2: (ta[k2] * ta["requests"] - tb[k2] * tb["requests"]) / dn for k2 in ("wait_for_batch_us", "batch_closed_to_answer_us", "answer_to_socket_us")}
            line.update({"leg": "scan service", "mean_batch": req / max(bat, 1), "over_device_resident": line.get("queries_per_s", 0) / resident,
                         "max_batch": a.max_batch, "max_wait_us": a.max_wait_us})
            print(json.dumps(line), flush=True)
    srv.stop()


if __name__ == "__main__":
    main()
