#!/usr/bin/env python
"""Read-only gather ceiling: the distance kernel on its own -- k_gather: metric(query, row[slots[i]]) -- over
as many RANDOM row ids as one 8192-query search launch evaluates (2144 x 8192 = 17.6 M rows of 3 KB on 1M x 768), with
no list maintenance, no visited set, no dependent hops.  Run under rocprofv3 (kernel trace, then FETCH_SIZE and TCC hit /
miss passes) it gives the fabric traffic and the time of a pure random-row stream to set beside k_search's.

    python scripts/bench_gather_ceiling.py [--rows 1000000 --dim 768 --evals 17563648]

CAUTION when reading a rocprofv3 table of this command: the FIRST k_gather launch is the 1000-row self-check below, so "average per launch"
over all 6 launches is 5/6 of a full launch's figure.  profiles/r03_gather_ceiling.md was read that way (8.02 ms, 46 GB, "6.73 TB/s"); per
FULL launch it is 9.58 ms and 55.3 GB = 5.64 TB/s algorithmic (profiles/r06_cache_model_calibration.md; scripts/calibrate_cache_model.py
reads the launches one by one).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import capi, hip  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--rows", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--evals", type=int, default=2144 * 8192)
    p.add_argument("--reps", type=int, default=5)
    a = p.parse_args()
    rng = np.random.default_rng(3)
    base = rng.standard_normal((a.rows, a.dim), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", a.dim, M=4, ef_construction=8, seed=1)
    g = {"levels": np.zeros(a.rows, np.uint8), "nbr0": np.full((a.rows, 8), 0xFFFFFFFF, np.uint32), "upper_off": np.full(a.rows, 0xFFFFFFFF, np.uint32),
         "upper_nbr": np.zeros((0, 4), np.uint32), "labels": None, "entry_slot": 0, "max_level": 0}
    ix.import_graph(base, g)  # rows only
    slots = rng.integers(0, a.rows, size=a.evals, dtype=np.uint32)  # uniform: no hub rows, no reuse beyond chance
    q = rng.standard_normal(a.dim, dtype=np.float32)
    out = ix.distance_gather(q, slots[:1000])
    ref = ((base[slots[:1000]] - q) ** 2).sum(1)
    assert np.allclose(out, ref, rtol=1e-4)
    wall = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        ix.distance_gather(q, slots)
        wall.append(time.perf_counter() - t0)
    nbytes = a.evals * a.dim * 4
    print(json.dumps({"rows": a.rows, "dim": a.dim, "row_evals_per_launch": a.evals, "bytes_per_launch": nbytes,
                      "host_wall_s_incl_copies": wall, "note": "kernel time and fabric traffic: see the rocprofv3 passes of this command (k_gather)"}))


if __name__ == "__main__":
    main()
