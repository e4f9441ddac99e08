#!/usr/bin/env python
"""One usearch_search_ef per launch (BASELINE config[1]'s form) at 128-d and 768-d: kernel time per call and per hop by HIP events.
The A/B companion of scripts/experiments/ab_lib.sh for changes that touch the latency-bound walk (walk_spec.hpp) or the descent."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lantern_amd import capi, hip  # noqa: E402


def one(n, d, nq=400, ef=64, k=10):
    base = np.random.default_rng(1).standard_normal((n, d), dtype=np.float32)
    queries = np.random.default_rng(2).standard_normal((nq, d), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", d, M=16, ef_construction=128, ef=ef, seed=42)
    ix.reserve(n)
    ix.set_add_batch(8192, 16)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    ix.flush()
    st = hip.Stream()
    rows = ix.device_query_rows(queries)
    dq = hip.Buffer.from_numpy(rows)
    d_lab, d_dst, d_D, d_E = hip.Buffer(k * 8), hip.Buffer(k * 4), hip.Buffer(8), hip.Buffer(8)
    kern, Es = [], []
    for i in range(nq):
        s, e = hip.Event(), hip.Event()
        s.record(st.handle)
        ix.search_batch_device(dq.ptr + i * rows.strides[0], 1, k, ef, 0, d_lab.ptr, d_dst.ptr, None, None, d_D.ptr, d_E.ptr, st.handle, query_stride=rows.strides[0])
        e.record(st.handle)
        st.synchronize()
        if i >= 50:
            kern.append(s.elapsed_ms(e) * 1e3)
            Es.append(int(d_E.download(1, np.uint64)[0]))
    return {"rows": n, "dim": d, "kernel_us_mean": round(float(np.mean(kern)), 2), "kernel_us_p50": round(float(np.median(kern)), 2),
            "hops": round(float(np.mean(Es)), 1), "us_per_hop": round(float(np.mean(kern) / np.mean(Es)), 3), "checksum": ix.checksum()}


if __name__ == "__main__":
    print(json.dumps({"lone_query": [one(100_000, 128), one(100_000, 768)]}))
