#!/usr/bin/env python
"""How often is the node a hop expands the list's FRONT (known before the previous hop's distances arrive) rather than one of the
previous hop's new rows?  The kill criterion of a speculative fetch of the front's adjacency list during the distance phase: from
the instrumented walk's own trace (lantern_gpu_search_row_trace), node h+1 is "new" when it appears among the rows evaluated in hop h."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lantern_amd import capi, hip, synth  # noqa: E402

out = {}
for data, d in (("clustered", 192), ("clustered", 768), ("gaussian", 768)):
    n, nq, k, ef, cap = 300_000, 512, 10, 64, 8192
    base = synth.base_rows(data, n, d, 3)
    queries = synth.query_maker(data, d)(np.random.default_rng(4), nq)
    ix = capi.GpuIndex("l2sq", d, M=16, ef_construction=128, ef=ef, seed=42)
    ix.reserve(n)
    ix.set_add_batch(32768, 16)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    ix.flush()
    rows = ix.device_query_rows(queries)
    dq = hip.Buffer.from_numpy(rows)
    lab, dist = hip.Buffer(nq * k * 8), hip.Buffer(nq * k * 4)
    ix.set_search_shape(4)
    ix.row_trace_begin(nq, cap)
    ix.search_batch_device(dq.ptr, nq, k, ef, 0, lab.ptr, dist.ptr, None, None, None, None, query_stride=rows.strides[0])
    hip.synchronize()
    trace, counts = ix.row_trace_end()
    hits = total = 0
    by_third = [[0, 0], [0, 0], [0, 0]]
    for q in range(nq):
        t = trace[q, :counts[q]]
        marks = np.nonzero((t >> 30) == 2)[0]  # level-0 lists, in expansion order
        for j in range(len(marks) - 1):
            new_rows = t[marks[j] + 1:marks[j + 1]]
            nxt = t[marks[j + 1]] & 0x3FFFFFFF
            hit = nxt not in new_rows[(new_rows >> 30) == 0]
            hits += hit
            total += 1
            b = by_third[min(2, 3 * j // max(1, len(marks) - 1))]
            b[0] += hit
            b[1] += 1
    out[f"{data} {d}-d"] = {"hops": total, "next_node_is_the_front": round(hits / total, 4),
                            "by_third_of_the_walk": [round(h / max(1, c), 3) for h, c in by_third]}
    del ix
print(json.dumps(out, indent=1))
