#!/usr/bin/env python
"""Calibration of the cache model (lantern_amd/tools/cache_model.c) on launches whose true traffic is known, against the counters:
the distance kernel in the WALK's launch shape (k_gather_walkshape) over 1M x 768 rows with (a) every row once per launch in slot
order -- a stream, no reuse: fabric bytes == algorithmic bytes, the control for the FETCH_SIZE conversion -- and (b) 17.6 M uniformly
random rows -- reuse by chance only.  Prints one JSON line with the model's figures; run it under
    rocprofv3 --kernel-include-regex k_gather --pmc FETCH_SIZE -- python scripts/calibrate_cache_model.py
    rocprofv3 --kernel-include-regex k_gather --pmc TCC_HIT_sum TCC_MISS_sum -- python scripts/calibrate_cache_model.py
(bash scripts/gpu.sh calibrate) for the counters of the same launches: launches 1-3 are (a), 4-6 are (b)."""
import json
import os
import sys

import numpy as np

os.environ["LANTERN_GPU_GATHER_WALKSHAPE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_cache_model as cm  # noqa: E402
from lantern_amd import capi  # noqa: E402


def main():
    rows, dim, per, nq = int(os.environ.get("CAL_ROWS", "1000000")), 768, 2144, 8192  # CAL_ROWS=10000000: a 30.7 GB table (does translation traffic grow?)
    rng = np.random.default_rng(3)
    base = rng.standard_normal((rows, dim), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", dim, M=4, ef_construction=8, seed=1)
    g = {"levels": np.zeros(rows, np.uint8), "nbr0": np.full((rows, 8), 0xFFFFFFFF, np.uint32), "upper_off": np.full(rows, 0xFFFFFFFF, np.uint32),
         "upper_nbr": np.zeros((0, 4), np.uint32), "labels": None, "entry_slot": 0, "max_level": 0}
    ix.import_graph(base, g)
    q = rng.standard_normal(dim, dtype=np.float32)
    seq = np.arange(rows, dtype=np.uint32)
    rnd = rng.integers(0, rows, size=per * nq, dtype=np.uint32)
    for _ in range(3):
        ix.distance_gather(q, seq)
    for _ in range(3):
        ix.distance_gather(q, rnd)
    # the model on the same draws: 1536 persistent workgroups take the draws interleaved; as a trace: nq walkers' lists of `per` rows
    tr = rnd.reshape(per, nq).T.copy()  # draw i goes to group i mod ngroups: column-major over "queries"
    ct = np.full(nq, per, dtype=np.uint32)
    cold, warm = cm.replay([tr, tr], [ct, ct], walkers=1536, row_bytes=dim * 4, list0_bytes=128, listu_bytes=64)
    print(json.dumps({"calibration": True, "rows": rows, "row_bytes": dim * 4,
                      "stream": {"rows_per_launch": rows, "algorithmic_bytes": rows * dim * 4, "launches": [1, 2, 3]},
                      "random": {"rows_per_launch": per * nq, "algorithmic_bytes": per * nq * dim * 4, "launches": [4, 5, 6],
                                 "model_fabric_bytes": warm["fabric_bytes"], "model_dram_bytes": warm["dram_bytes"],
                                 "model_fabric_over_algorithmic": warm["fabric_bytes"] / (per * nq * dim * 4),
                                 "model_dram_over_algorithmic": warm["dram_bytes"] / (per * nq * dim * 4)}}))


if __name__ == "__main__":
    main()
