#!/usr/bin/env python
"""Where a hop's time goes: the instrumented walk kernel (lantern_gpu_search_phase_profile) on the latency-bound shapes --
a lone query (BASELINE config[1]) and a 1024-query batch (config[2]) -- and on the bandwidth-bound 8192-query batch.
Prints shader-clock cycles per hop by phase (thread 0's view: a phase ends at the barrier that closes it)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import capi, hip  # noqa: E402


def measure(ix, queries, nq_per_launch, waves, launches, k=10, ef=64):
    d = queries.shape[1]
    dq = hip.Buffer.from_numpy(hip.padded_rows(queries, False))
    row_bytes = hip.padded_rows(queries[:1], False).shape[1] * 4
    lab, dst = hip.Buffer(nq_per_launch * k * 8), hip.Buffer(nq_per_launch * k * 4)
    Dv, Ev = hip.Buffer(nq_per_launch * 8), hip.Buffer(nq_per_launch * 8)
    st = hip.Stream()
    ix.set_search_shape(waves, 0)
    ix.phase_profile(True, read=True)
    hops = evals = 0
    e0, e1 = hip.Event(), hip.Event()
    e0.record(st.handle)
    for i in range(launches):
        off = (i * nq_per_launch) % max(1, queries.shape[0] - nq_per_launch + 1)
        ix.search_batch_device(dq.ptr + off * row_bytes, nq_per_launch, k, ef, 0, lab.ptr, dst.ptr, None, None, Dv.ptr, Ev.ptr, st.handle)
        if nq_per_launch == 1:
            st.synchronize()
            hops += int(Ev.download(1, np.uint64)[0])
            evals += int(Dv.download(1, np.uint64)[0])
    e1.record(st.handle)
    hip.synchronize()
    if nq_per_launch > 1:
        hops = int(Ev.download(nq_per_launch, np.uint64).sum()) * launches
        evals = int(Dv.download(nq_per_launch, np.uint64).sum()) * launches
    ph = ix.phase_profile(False, read=True)
    nqs = launches * nq_per_launch
    out = {"queries": nqs, "waves": waves, "hops_per_query": hops / nqs, "evals_per_query": evals / nqs,
           "cycles_per_hop": {k2: ph[k2] / hops for k2 in ("pop", "list_arrival", "visited_compact", "first_barrier", "distances", "merge")},
           "descent_cycles_per_query": ph["descent"] / nqs, "cycles_per_query": ph["query"] / nqs,
           "wall_us_per_launch": e0.elapsed_ms(e1) * 1e3 / launches}
    out["implied_clock_GHz_if_query_is_launch"] = out["cycles_per_query"] / out["wall_us_per_launch"] / 1e3 if nq_per_launch == 1 else None
    return out


def main():
    res = {}
    base = np.random.default_rng(1).standard_normal((100_000, 128), dtype=np.float32)
    q = np.random.default_rng(2).standard_normal((2000, 128), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", 128, M=16, ef_construction=128, ef=64, seed=42)
    ix.add_many(np.arange(base.shape[0], dtype=np.uint64) + 1, base)
    ix.flush()
    for w in (8, 4, 1):
        res[f"100kx128 single query, {w} waves"] = measure(ix, q, 1, w, 300)
    del ix
    n = int(os.environ.get("ROWS768", "300000"))
    base = np.random.default_rng(3).standard_normal((n, 768), dtype=np.float32)
    q = np.random.default_rng(4).standard_normal((8192, 768), dtype=np.float32)
    metric = os.environ.get("METRIC768", "l2sq")
    ix = capi.GpuIndex(metric, 768, M=16, ef_construction=128, ef=64, seed=42)
    ix.reserve(n)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    ix.flush()
    res[f"{n}x768 single query, 8 waves"] = measure(ix, q, 1, 8, 200)
    res[f"{n}x768 1024 queries, 4 waves"] = measure(ix, q, 1024, 4, 5)
    res[f"{n}x768 1024 queries, 8 waves"] = measure(ix, q, 1024, 8, 5)
    res[f"{n}x768 8192 queries, 4 waves"] = measure(ix, q, 8192, 4, 3)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
