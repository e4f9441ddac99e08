#!/usr/bin/env python
"""Host time of each lantern_gpu_search_batch_device_strided call when the launches are short (1024 queries, ~0.3 ms of kernel): does a
call return at once (the kernel queued), or does something on the host side wait?  Seen in bench.py's wall clock with --queries 1024:
some runs 0.5 ms per step, some 3.8 ms, the kernel 0.49 ms in both.
Result (round 6): 7 - 11 us of host time per call with or without events around it, 40 launches = 40 x the kernel time: the library's
call path does not wait; the bench's occasional slow wall clock at that batch size is not reproduced here (the driver's command uses
8192-query steps, where the host runs ahead either way)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from lantern_amd import capi, hip  # noqa: E402

n, d, nq, k = 200_000, 768, 1024, 10
rng = np.random.default_rng(1)
base = rng.standard_normal((n, d), dtype=np.float32)
ix = capi.GpuIndex("cos", d, M=16, ef_construction=128, ef=64, seed=42)
ix.reserve(n)
ix.set_add_batch(8192, 16)
ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
ix.flush()
rows = ix.device_query_rows(rng.standard_normal((nq, d), dtype=np.float32))
dq = hip.Buffer.from_numpy(rows)
lab, dist, slot, D, E = hip.Buffer(nq * k * 8), hip.Buffer(nq * k * 4), hip.Buffer(nq * k * 4), hip.Buffer(nq * 8), hip.Buffer(nq * 8)
st = hip.Stream()
out = []
for with_events in (True, False, True):
    for trial in range(3):
        hip.synchronize()
        ev = [(hip.Event(), hip.Event()) for _ in range(40)]
        per_call = []
        t_all = time.perf_counter()
        for s, e in ev:
            t0 = time.perf_counter()
            if with_events:
                s.record(st.handle)
            ix.search_batch_device(dq.ptr, nq, k, 64, 0, lab.ptr, dist.ptr, slot.ptr, None, D.ptr, E.ptr, st.handle, query_stride=rows.strides[0])
            if with_events:
                e.record(st.handle)
            per_call.append((time.perf_counter() - t0) * 1e6)
        t_issue = time.perf_counter() - t_all
        hip.synchronize()
        t_total = time.perf_counter() - t_all
        kern = [s.elapsed_ms(e) for s, e in ev] if with_events else []
        out.append({"events": with_events, "trial": trial, "issue_ms": round(t_issue * 1e3, 3), "total_ms": round(t_total * 1e3, 3),
                    "kernel_ms_mean": round(float(np.mean(kern)), 4) if kern else None,
                    "host_us_per_call": {"p50": round(float(np.median(per_call)), 1), "max": round(float(np.max(per_call)), 1), "first": round(per_call[0], 1)}})
print(json.dumps(out, indent=1))
