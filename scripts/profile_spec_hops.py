#!/usr/bin/env python
"""Where a hop of the LATENCY-BOUND walk (lantern_amd/csrc/walk_spec.hpp) spends its time: the instrumented kernel
(lantern_gpu_spec_profile) on a lone query (BASELINE config[1], the 3 + 8-wave shape) and on 1024-query batches at 768-d in the
shapes the launcher can choose from.  Cycles per hop by wave role and section; where neighbour lists came from.

    python scripts/profile_spec_hops.py > profiles/r03_spec_hop_phases.json
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import capi, hip  # noqa: E402


def measure(ix, queries, nq_per_launch, launches, k=10, ef=64):
    dq = hip.Buffer.from_numpy(hip.padded_rows(queries, False))
    row_bytes = hip.padded_rows(queries[:1], False).shape[1] * 4
    lab, dst = hip.Buffer(nq_per_launch * k * 8), hip.Buffer(nq_per_launch * k * 4)
    st = hip.Stream()
    ix.set_search_shape(0, 0)
    out = {}
    for prof in (False, True):  # timing without the clock reads, then the sections
        ix.spec_profile(prof, read=True)
        e0, e1 = hip.Event(), hip.Event()
        for i in range(3):
            ix.search_batch_device(dq.ptr, nq_per_launch, k, ef, 0, lab.ptr, dst.ptr, None, None, None, None, st.handle)
        st.synchronize()
        ix.spec_profile(prof, read=True)
        e0.record(st.handle)
        for i in range(launches):
            off = (i * nq_per_launch) % max(1, queries.shape[0] - nq_per_launch + 1)
            ix.search_batch_device(dq.ptr + off * row_bytes, nq_per_launch, k, ef, 0, lab.ptr, dst.ptr, None, None, None, None, st.handle)
        e1.record(st.handle)
        st.synchronize()
        us = e0.elapsed_ms(e1) * 1e3 / launches
        if not prof:
            out["us_per_launch"] = us
        else:
            p = ix.spec_profile(False, read=True)
            hops = max(p["row"]["hops"], 1)
            out["us_per_launch_instrumented"] = us
            out["hops_per_query"] = hops / (launches * nq_per_launch)
            out["cycles_per_hop"] = {role: {k2: v / hops for k2, v in sec.items() if k2 not in ("hops", "list_source")} for role, sec in p.items()}
            out["rounds_per_query"] = hops / (launches * nq_per_launch)
            out["speculative_nodes_completed_per_round"] = p["list"]["list_source"] / hops  # (the two-nodes-per-round walk: walk_twin.hpp; else 0)
            tot = max(p["visit"]["list_source"] + p["row"]["list_source"] + p["fill"]["list_source"], 1)
            out["neighbour_list_source"] = {"staged_with_previous_hop": p["visit"]["list_source"] / tot, "lds_cache": p["row"]["list_source"] / tot,
                                            "hbm": p["fill"]["list_source"] / tot}
    return out


def main():
    res = {}
    base = np.random.default_rng(1).standard_normal((100_000, 128), dtype=np.float32)
    q = np.random.default_rng(2).standard_normal((2000, 128), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", 128, M=16, ef_construction=128, ef=64, seed=42)
    ix.add_many(np.arange(base.shape[0], dtype=np.uint64) + 1, base)
    ix.flush()
    res["100kx128 lone query (3 role + 8 row waves)"] = measure(ix, q, 1, 300)
    del ix
    n = int(os.environ.get("ROWS768", "300000"))
    base = np.random.default_rng(3).standard_normal((n, 768), dtype=np.float32)
    q = np.random.default_rng(4).standard_normal((4096, 768), dtype=np.float32)
    for metric in ("l2sq", "cos"):
        ix = capi.GpuIndex(metric, 768, M=16, ef_construction=128, ef=64, seed=42)
        ix.reserve(n)
        ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
        ix.flush()
        res[f"{n}x768 {metric} lone query"] = measure(ix, q, 1, 200)
        for spec, waves in (("1", "4"), ("1", "8"), ("2", "11"), ("2", "7")):
            os.environ["LANTERN_GPU_SPEC"], os.environ["LANTERN_GPU_SPEC_WAVES"] = spec, waves
            res[f"{n}x768 {metric} 1024 queries, spec {spec}, {waves} waves"] = measure(ix, q, 1024, 4)
        del os.environ["LANTERN_GPU_SPEC"], os.environ["LANTERN_GPU_SPEC_WAVES"]
        del ix
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
