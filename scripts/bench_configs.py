#!/usr/bin/env python
"""Secondary measurements of the other BASELINE.json configs (parity-test cases, not the bench line):
single-query latency (config[1]), the 1024-query cosine batch and the dense MFMA contraction (config[2]),
the hamming check set, and a 1536-d build (config[4] shape, one GPU).  Prints one JSON object per config.

    python scripts/bench_configs.py > gpurun_out/configs.jsonl
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import capi, hip  # noqa: E402

HBM = 8000.0


def build(metric, base, M=16, efc=128, ef=64):
    ix = capi.GpuIndex(metric, base.shape[1], M=M, ef_construction=efc, ef=ef, seed=42)
    ix.reserve(base.shape[0])
    hip.synchronize()
    t0 = time.time()
    ix.add_many(np.arange(base.shape[0], dtype=np.uint64) + 1, base)
    ix.flush()
    hip.synchronize()
    return ix, time.time() - t0


def batch_qps(ix, queries, k, ef, steps=10, waves=4, ham=False):
    nq = queries.shape[0]
    rows = ix.device_query_rows(queries)
    dq = hip.Buffer.from_numpy(rows)
    lab, dist, slot = hip.Buffer(nq * k * 8), hip.Buffer(nq * k * 4), hip.Buffer(nq * k * 4)
    Dv, Ev = hip.Buffer(nq * 8), hip.Buffer(nq * 8)
    st = hip.Stream()
    ix.set_search_shape(waves)
    go = lambda: ix.search_batch_device(dq.ptr, nq, k, ef, 0, lab.ptr, dist.ptr, slot.ptr, None, Dv.ptr, Ev.ptr, st.handle, query_stride=rows.strides[0])
    for _ in range(2):
        go()
    hip.synchronize()
    e0, e1 = hip.Event(), hip.Event()
    e0.record(st.handle)
    for _ in range(steps):
        go()
    e1.record(st.handle)
    hip.synchronize()
    ms = e0.elapsed_ms(e1) / steps
    D, E = Dv.download(nq, np.uint64).astype(float), Ev.download(nq, np.uint64).astype(float)
    return nq / ms * 1e3, ms, D, E, slot.download((nq, k), np.uint32)


def recall(found, truth):
    return float(np.mean([len(set(f.tolist()) & set(t.tolist())) / truth.shape[1] for f, t in zip(found, truth)]))


def main():
    out = []
    # ---- config[1]: 100k x 128 f32 L2sq, M=16 ef=64 k=10, single-query search --------------------------
    base = np.random.default_rng(1).standard_normal((100_000, 128), dtype=np.float32)
    queries = np.random.default_rng(2).standard_normal((10_000, 128), dtype=np.float32)
    ix, tb = build("l2sq", base)
    for q in queries[:20]:
        ix.search(q, 10)
    t0 = time.perf_counter()
    for q in queries[:500]:
        ix.search(q, 10)
    lat = (time.perf_counter() - t0) / 500
    qps, ms, D, E, slot = batch_qps(ix, queries, 10, 64)
    truth, _ = ix.exact_search(queries[:1024], 10)
    bytes_q = D * 128 * 4 + E * 128 + 128 * 4
    out.append({"config": "100k x 128 f32 l2sq M=16 ef=64 k=10", "single_query_latency_us": lat * 1e6, "single_query_qps": 1 / lat,
                "batch_10000_qps": qps, "batch_ms": ms, "recall_at_10": recall(slot[:1024], truth), "build_vectors_per_s": 100_000 / tb,
                "dist_evals_per_query": D.mean(), "algorithmic_GBps_batch": bytes_q.sum() / ms / 1e6,
                "note": "single-query = usearch_search_ef through the host ABI (H2D query, one 8-wave workgroup, D2H result): launch/latency-bound"})
    print(json.dumps(out[-1]), flush=True)
    del ix
    # ---- hamming check set: 100k x 768 bits ------------------------------------------------------------
    words = np.random.default_rng(9).integers(0, 2**32, size=(100_000, 24), dtype=np.uint32)
    hq = np.random.default_rng(10).integers(0, 2**32, size=(4096, 24), dtype=np.uint32)
    ix, tb = build("hamming", words)
    qps, ms, D, E, slot = batch_qps(ix, hq, 10, 64, ham=True)
    truth, _ = ix.exact_search(hq[:1024], 10)
    out.append({"config": "hamming 100k x 768 bits M=16 ef=64 k=10", "batch_4096_qps": qps, "recall_at_10": recall(slot[:1024], truth),
                "build_vectors_per_s": 100_000 / tb, "dist_evals_per_query": D.mean()})
    print(json.dumps(out[-1]), flush=True)
    del ix
    # ---- config[2]: 1M x 768 f32 cosine, 1024-query batch; dense MFMA contraction ----------------------
    base = np.random.default_rng(3).standard_normal((1_000_000, 768), dtype=np.float32)
    queries = np.random.default_rng(4).standard_normal((1024, 768), dtype=np.float32)
    ix, tb = build("cos", base)
    res = {}
    for waves in (4, 8):
        qps, ms, D, E, slot = batch_qps(ix, queries, 10, 64, waves=waves)
        bytes_q = D * 768 * 4 + E * 128 + 768 * 4
        res[f"waves{waves}"] = {"qps": qps, "ms": ms, "algorithmic_GBps": bytes_q.sum() / ms / 1e6, "frac_of_hbm_peak": bytes_q.sum() / ms / 1e6 / HBM}
    hip.synchronize()
    t0 = time.perf_counter()
    truth, _ = ix.exact_search(queries, 10)
    t_exact = time.perf_counter() - t0
    out.append({"config": "1M x 768 f32 cosine M=16 ef=64 k=10, 1024-query batch", "search": res, "recall_at_10": recall(slot, truth),
                "build_vectors_per_s": 1_000_000 / tb, "dist_evals_per_query": D.mean(),
                "exact_search_1024x1M_s": t_exact, "dense_contraction_TFLOPs": 2 * 1024 * 1e6 * 768 / 1e12,
                "note": "exact_search wall time includes H2D of the queries, norms, 16 chunks of MFMA tile + top-k select, re-rank, D2H"})
    print(json.dumps(out[-1]), flush=True)
    del ix
    # ---- config[4] shape on one GPU: 1536-d build (200k rows) -------------------------------------------
    base = np.random.default_rng(7).standard_normal((200_000, 1536), dtype=np.float32)
    queries = np.random.default_rng(8).standard_normal((1000, 1536), dtype=np.float32)
    ix, tb = build("l2sq", base)
    qps, ms, D, E, slot = batch_qps(ix, queries, 10, 64)
    truth, _ = ix.exact_search(queries, 10)
    out.append({"config": "200k x 1536 f32 l2sq build M=16 efc=128 (config[4] row width, 1 GPU)", "build_vectors_per_s": 200_000 / tb,
                "build_seconds": tb, "recall_at_10_ef64": recall(slot, truth), "batch_1000_qps": qps})
    print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
