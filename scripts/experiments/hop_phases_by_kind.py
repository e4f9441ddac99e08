#!/usr/bin/env python
"""Where a hop's time goes in the bandwidth-bound shape (8192 queries, four waves) by storage kind, on the clustered set: the
instrumented walk (lantern_gpu_search_phase_profile), shader-clock cycles per hop by phase, thread 0's view.  Question: the i8 walk
runs at 0.6 of its own random-gather rate where f32 and f16 run at theirs -- which phase carries the difference?"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lantern_amd import capi, hip, synth  # noqa: E402

n, nq, k, ef = int(os.environ.get("ROWS", "500000")), 8192, 10, 64
res = {}
# (the instrumented walk exists for f32 rows only: a 192-d f32 row has the 768 bytes, the 16-lane groups and the three steps of a 768-d i8 row)
for kind, scale, d in (("f32", 1.0, 768), ("f32", 1.0, 192)):
    base = synth.base_rows("clustered", n, d, 3)
    queries = synth.query_maker("clustered", d)(np.random.default_rng(4), nq)
    ix = capi.GpuIndex("l2sq", d, M=16, ef_construction=128, ef=ef, seed=42, quantization=kind)
    ix.reserve(n)
    ix.set_add_batch(32768, 16)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, base * np.float32(scale))
    ix.flush()
    rows = ix.device_query_rows(queries * np.float32(scale))
    dq = hip.Buffer.from_numpy(rows)
    lab, dst, Dv, Ev = hip.Buffer(nq * k * 8), hip.Buffer(nq * k * 4), hip.Buffer(nq * 8), hip.Buffer(nq * 8)
    st = hip.Stream()

    def launch():
        ix.search_batch_device(dq.ptr, nq, k, ef, 0, lab.ptr, dst.ptr, None, None, Dv.ptr, Ev.ptr, st.handle, query_stride=rows.strides[0])

    for _ in range(2):
        launch()
    hip.synchronize()
    e0, e1 = hip.Event(), hip.Event()
    e0.record(st.handle)
    for _ in range(3):
        launch()
    e1.record(st.handle)
    hip.synchronize()
    plain_ms = e0.elapsed_ms(e1) / 3
    ix.phase_profile(True, read=True)
    e0.record(st.handle)
    for _ in range(3):
        launch()
    e1.record(st.handle)
    hip.synchronize()
    prof_ms = e0.elapsed_ms(e1) / 3
    ph = ix.phase_profile(False, read=True)
    hops, evals = int(Ev.download(nq, np.uint64).sum()) * 3, int(Dv.download(nq, np.uint64).sum()) * 3
    res[f"{kind} {d}-d"] = {"launch_ms": round(plain_ms, 4), "launch_ms_instrumented": round(prof_ms, 4), "hops_per_query": hops / 3 / nq, "evals_per_hop": evals / hops,
                 "row_bytes": ix.row_bytes(),
                 "cycles_per_hop": {p: round(ph[p] / hops, 1) for p in ("pop", "list_arrival", "visited_compact", "first_barrier", "distances", "merge")},
                 "descent_cycles_per_query": round(ph["descent"] / 3 / nq, 1), "cycles_per_query": round(ph["query"] / 3 / nq, 1)}
    del ix
print(json.dumps(res, indent=1))
