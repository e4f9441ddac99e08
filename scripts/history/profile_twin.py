#!/usr/bin/env python
"""Sections of a round of the two-nodes-per-round walk (walk_twin.hpp) beside those of a hop of the one-node walk, lone query,
100k x 128 (BASELINE config[1]): LANTERN_GPU_TWIN=0|1 around scripts/profile_spec_hops.py's `measure`."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lantern_amd import capi  # noqa: E402
from profile_spec_hops import measure  # noqa: E402

res = {}
base = np.random.default_rng(1).standard_normal((100_000, 128), dtype=np.float32)
q = np.random.default_rng(2).standard_normal((2000, 128), dtype=np.float32)
ix = capi.GpuIndex("l2sq", 128, M=16, ef_construction=128, ef=64, seed=42)
ix.add_many(np.arange(base.shape[0], dtype=np.uint64) + 1, base)
ix.flush()
for twin in ("0", "1"):
    os.environ["LANTERN_GPU_TWIN"] = twin
    res[f"100kx128 lone query, twin={twin}"] = measure(ix, q, 1, 300)
print(json.dumps(res, indent=1))
