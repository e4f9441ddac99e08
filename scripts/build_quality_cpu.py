#!/usr/bin/env python
"""Build-quality comparison WITHOUT a device: the batch-synchronous plan the device builds with (batches of up to 8192, never
more than size / 16) against the reference's strictly sequential build (one usearch_add per tuple, build.c:83-135), both by the
CPU port (oracle/, the reference's summation flags), searched by the port against exact truth.

The planned CPU build IS the device's graph: tests/test_gpu_build_parity_production_batch.py pins the device build edge for edge
to `add_planned(8192, 16)` at 100k-160k rows, and the device search to the port's on the same graph bit for bit.  That makes the
1M x 768 comparison affordable to commit (a sequential build of the headline set takes ~15 minutes of one core: the rate falls
from 6 k vectors/s at 4k rows to ~1 k at 1M) without holding a GPU box for it.

    python scripts/build_quality_cpu.py --rows 1000000 --dim 768 --data gaussian [--metric l2sq] [--threads 8]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import synth  # noqa: E402
from oracle import binding as oracle  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--rows", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--data", default="gaussian")
    p.add_argument("--metric", default="l2sq")
    p.add_argument("--threads", type=int, default=8)
    p.add_argument("--queries", type=int, default=1000)
    p.add_argument("--M", type=int, default=16)
    p.add_argument("--efc", type=int, default=128)
    p.add_argument("--ef", type=int, default=64)
    a = p.parse_args()
    oracle.build()
    base = synth.base_rows(a.data, a.rows, a.dim)
    q = synth.query_maker(a.data, a.dim)(np.random.default_rng(4), a.queries)
    labels = np.arange(a.rows, dtype=np.uint64) + 1
    truth, _ = oracle.bruteforce(base, q, 10, a.metric, oracle.SUM_FAST, a.threads)
    out = {"set": f"{a.rows}x{a.dim} f32 {a.metric} {a.data} (base seed {synth.BASE_SEED}, query seed 4), M={a.M} ef_construction={a.efc} ef={a.ef}, {a.queries} queries"}
    for name in ("planned", "sequential"):
        ix = oracle.OracleIndex(a.metric, a.dim, M=a.M, ef_construction=a.efc, ef=a.ef, seed=42, sum_mode=oracle.SUM_FAST)
        ix.reserve(a.rows)
        t0 = time.time()
        if name == "planned":
            ix.set_build_threads(a.threads)
            ix.add_planned(labels, base, 8192, 16)
        else:
            ix.add_many(labels, base)
        t = time.time() - t0
        _, _, slots, D, E = ix.search_batch(q, 10, a.ef, a.threads)
        g = ix.export_graph()
        out[name] = {"recall_at_10": oracle.recall_at_k(slots, truth), "build_seconds": t, "dist_evals_per_query": float(D.mean()), "expansions_per_query": float(E.mean()),
                     "mean_out_degree_level0": float((g["nbr0"] != 0xFFFFFFFF).sum(axis=1).mean())}
        print(name, json.dumps(out[name]), file=sys.stderr, flush=True)
        del ix, g
    out["abs_diff"] = abs(out["planned"]["recall_at_10"] - out["sequential"]["recall_at_10"])
    out["bar"] = 0.005
    print(json.dumps(out))


if __name__ == "__main__":
    main()
