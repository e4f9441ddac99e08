#!/usr/bin/env python
"""A/B of the walk's candidate-list placement (walk.hpp search_level_reg vs search_level; LANTERN_GPU_LDS_LIST=1 forces the
LDS form): the latency-bound batch shapes, the bandwidth-bound batch and the build.  Same index, same queries, same answers;
only the time differs.  Prints one JSON object."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_configs import batch_qps, build  # noqa: E402


def both(fn):
    out = {}
    for name, v in (("registers", "0"), ("lds", "1")):
        os.environ["LANTERN_GPU_LDS_LIST"] = v
        out[name] = fn()
    os.environ.pop("LANTERN_GPU_LDS_LIST", None)
    return out


def main():
    n = int(os.environ.get("ROWS768", "1000000"))
    res = {}
    for metric in ("cos", "l2sq"):
        base = np.random.default_rng(3).standard_normal((n, 768), dtype=np.float32)
        q = np.random.default_rng(4).standard_normal((8192, 768), dtype=np.float32)
        builds = {}
        ix = None
        for name, v in (("lds", "1"), ("registers", "0")):
            os.environ["LANTERN_GPU_LDS_LIST"] = v
            del ix
            ix, tb = build(metric, base)
            builds[name] = n / tb
        os.environ.pop("LANTERN_GPU_LDS_LIST", None)
        res[f"{metric} {n}x768 build vectors/s"] = builds
        for nq, waves in ((1024, 0), (1024, 8), (256, 0), (8192, 0)):
            def run():
                qps, ms, D, E, slot = batch_qps(ix, q[:nq], 10, 64, waves=waves)
                by = float((D * 768 * 4 + E * 128 + 768 * 4).sum())
                return {"qps": qps, "ms": ms, "frac_of_hbm_peak": by / ms / 1e6 / 8000.0, "slot_checksum": int(slot.astype(np.uint64).sum())}
            res[f"{metric} {nq} queries waves={waves or 'auto'}"] = both(run)
        del ix
    base = np.random.default_rng(1).standard_normal((100_000, 128), dtype=np.float32)
    q = np.random.default_rng(2).standard_normal((10_000, 128), dtype=np.float32)
    ix, _ = build("l2sq", base)
    for nq in (64, 1024, 10_000):
        res[f"l2sq 100kx128 {nq} queries"] = both(lambda: dict(zip(("qps", "ms"), batch_qps(ix, q[:nq], 10, 64, waves=0)[:2])))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
