#!/usr/bin/env python
"""The PLAIN-OUTPUT variant of k_dense_f32 (the 1024 x 65 536 distance matrix written out: lantern_gpu_distance_matrix with
exact_order = 0, the path PQ's nearest-centroid search and the first columns of an exact k-NN take) on its own.  Run under
`rocprofv3 --kernel-trace --stats`: the launches of k_dense_f32<metric, false> are what is timed (the call's H2D / D2H copies of
470 MB are not the kernel's).  Prints the wall time per call for orientation."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import capi  # noqa: E402

metric = sys.argv[1] if len(sys.argv) > 1 else "l2sq"
a = np.random.default_rng(4).standard_normal((1024, 768), dtype=np.float32)
b = np.random.default_rng(3).standard_normal((65536, 768), dtype=np.float32)
capi.distance_matrix(a, b, metric, exact_order=False)
t0 = time.perf_counter()
for _ in range(6):
    m = capi.distance_matrix(a, b, metric, exact_order=False)
dt = (time.perf_counter() - t0) / 6
ref = ((a[:4, None, :].astype(np.float64) - b[None, :64, :].astype(np.float64)) ** 2).sum(-1) if metric == "l2sq" else None
err = float(np.abs(m[:4, :64] - ref).max() / ref.max()) if ref is not None else None
print(json.dumps({"config": f"distance matrix 1024 x 65536 x 768 f32 {metric}, plain output", "seconds_per_call_wall_incl_copies": dt, "GFLOP_per_launch": 2 * 1024 * 65536 * 768 / 1e9,
                  "max_rel_err_sample": err}))
