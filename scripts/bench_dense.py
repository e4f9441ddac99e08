#!/usr/bin/env python
"""The dense batched-query x candidate contraction on its own (BASELINE config[2]: 1024 queries x 1M x 768,
cosine): exact k-NN through k_dense_f32 (fp32 MFMA) + k_select + k_rerank.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import capi, hip  # noqa: E402

n, d, nq, k = 1_000_000, 768, 1024, 10
metric = sys.argv[1] if len(sys.argv) > 1 else "cos"
base = np.random.default_rng(3).standard_normal((n, d), dtype=np.float32)
queries = np.random.default_rng(4).standard_normal((nq, d), dtype=np.float32)
ix = capi.GpuIndex(metric, d, M=4, ef_construction=8)
g = {"levels": np.zeros(n, np.uint8), "nbr0": np.full((n, 8), 0xFFFFFFFF, np.uint32), "upper_off": np.full(n, 0xFFFFFFFF, np.uint32),
     "upper_nbr": np.zeros((0, 4), np.uint32), "labels": None, "entry_slot": 0, "max_level": 0}
ix.import_graph(base, g)
ix.exact_search(queries, k)
hip.synchronize()
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    slots, dists = ix.exact_search(queries, k)
hip.synchronize()
dt = (time.perf_counter() - t0) / reps
flops = 2.0 * nq * n * d
print(json.dumps({"config": f"exact k-NN {nq} x {n} x {d} f32 {metric}", "seconds_per_call_wall": dt, "algorithmic_TFLOP": flops / 1e12,
                  "TFLOPs_wall": flops / dt / 1e12, "peak_fp32_matrix_TFLOPs": 157.3}))
