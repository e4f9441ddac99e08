#!/usr/bin/env python
"""How much of a lone query's hop is memory?  The same walk (lone-query shape, 128-d) on an index that fits the 4 MB L2 of one
XCD (4 000 rows = 2 MB) beside the 100 000-row index of BASELINE config[1]: what is left of a hop when rows cost an L2 hit."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lantern_amd import capi  # noqa: E402
from profile_spec_hops import measure  # noqa: E402

os.environ["LANTERN_GPU_TWIN"] = "0"
res = {}
for n in (4000, 100_000):
    base = np.random.default_rng(1).standard_normal((n, 128), dtype=np.float32)
    q = np.random.default_rng(2).standard_normal((2000, 128), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", 128, M=16, ef_construction=128, ef=64, seed=42)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    ix.flush()
    m = measure(ix, q, 1, 300)
    m["us_per_hop"] = m["us_per_launch"] / m["hops_per_query"]
    res[f"{n}x128 lone query"] = m
    del ix
print(json.dumps(res, indent=1))
