#!/usr/bin/env python
"""Freeze the oracle's behaviour on small seeded cases into tests/golden/oracle_regression.json.

    python scripts/make_oracle_golden.py          # rewrites the fixture

The reference pins this path only on the tiny exact cases of tests/golden/lantern_expected.json; everything larger is
arbitrated by the oracle.  This fixture pins the ORACLE: an edit to oracle/*.c that changes a
graph, a result list or a distance bit shows up as a diff here, not as a silent change of what the GPU path is compared
against.  Inputs are regenerated from the seeds; outputs are stored exactly (ids as integers, f32 distances as hex bits).
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as oracle  # noqa: E402

CASES = [
    # name, metric, n, d, M, efc, ef, k, sum mode, batch plan (None = sequential usearch_add), storage
    ("l2sq_seq", "l2sq", 600, 24, 6, 32, 32, 5, "SUM_SEQ", None, "f32"),
    ("l2sq_wave_planned", "l2sq", 900, 40, 8, 40, 48, 10, "SUM_WAVE64", (64, 8), "f32"),
    ("cos_wave", "cos", 500, 33, 4, 24, 24, 5, "SUM_WAVE64", (32, 4), "f32"),
    ("cos_wave_768", "cos", 300, 768, 8, 32, 32, 5, "SUM_WAVE64", (16, 4), "f32"),
    ("hamming", "hamming", 700, 6, 6, 32, 32, 5, "SUM_SEQ", (64, 8), "f32"),
    ("l2sq_f16", "l2sq", 400, 50, 6, 32, 32, 5, "SUM_WAVE64_F16", (32, 8), "f16"),
    ("l2sq_i8", "l2sq", 400, 50, 6, 32, 32, 5, "SUM_I8", (32, 8), "i8"),
]


def rows(rng, n, d, metric):
    if metric == "hamming":
        return rng.integers(0, 2**32, size=(n, d), dtype=np.uint32)
    return rng.standard_normal((n, d), dtype=np.float32)


def run_case(name, metric, n, d, M, efc, ef, k, mode, plan, storage):
    rng = np.random.default_rng(int(hashlib.sha256(name.encode()).hexdigest()[:8], 16))
    base, queries = rows(rng, n, d, metric), rows(rng, 16, d, metric)
    if storage == "f16":
        base, queries = oracle.round_f16(base), oracle.round_f16(queries)
    if storage == "i8":
        base, queries = oracle.quantize_i8(base * np.float32(0.4)), oracle.quantize_i8(queries * np.float32(0.4))
    ix = oracle.OracleIndex(metric, d, M=M, ef_construction=efc, ef=ef, seed=11, sum_mode=getattr(oracle, mode))
    labels = np.arange(n, dtype=np.uint64) + 1
    if plan is None:
        ix.add_many(labels, base)
    else:
        ix.add_planned(labels, base, max_batch=plan[0], min_ratio=plan[1])
    g = ix.export_graph()
    lab, dist, slots, D, E = ix.search_batch(queries, k)
    h = hashlib.sha256()
    for key in ("levels", "nbr0", "upper_off", "upper_nbr"):
        h.update(np.ascontiguousarray(g[key]).tobytes())
    return {
        "graph_sha256": h.hexdigest(), "entry_slot": int(g["entry_slot"]), "max_level": int(g["max_level"]),
        "slots": slots.astype(np.int64).tolist(),
        "dist_bits": [[f"{int(x):08x}" for x in row] for row in dist.view(np.uint32)],
        "dist_evals": D.astype(np.int64).tolist(), "expansions": E.astype(np.int64).tolist(),
    }


def main():
    oracle.build()
    out = {"_about": "oracle regression fixture; regenerate with scripts/make_oracle_golden.py", "cases": {}}
    for c in CASES:
        out["cases"][c[0]] = {"params": {"metric": c[1], "n": c[2], "d": c[3], "M": c[4], "efc": c[5], "ef": c[6], "k": c[7], "sum_mode": c[8],
                                         "plan": c[9], "storage": c[10]}, "expect": run_case(*c)}
    path = os.path.join(ROOT, "tests", "golden", "oracle_regression.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
