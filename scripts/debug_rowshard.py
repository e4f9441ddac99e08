"""debug aid: one-rank row-sharded build of a small set (argv: n d)"""
import sys
import numpy as np
from lantern_amd import capi

n, d = int(sys.argv[1]), int(sys.argv[2])
rows = np.random.default_rng(1).standard_normal((n, d), dtype=np.float32)
comms = capi.Comm.local_world(1)
ix = capi.GpuIndex("l2sq", d, M=16, ef_construction=128, seed=3)
ix.add_row_sharded(comms[0], np.arange(n, dtype=np.uint64) + 1, rows)
lab, _, _ = ix.search_batch(rows[:min(n, 64)], 1, 64)
print("ok", n, d, float((lab[:, 0] == np.arange(min(n, 64)) + 1).mean()))
