#!/usr/bin/env python
"""Concurrent single-query clients through the scan-side service (lantern_scan_server_*): BASELINE config[1]'s shape
(100k x 128 f32 L2sq, M=16 ef=64 k=10), every client thread issues one query at a time -- the way PostgreSQL backends
do -- and the server coalesces them.  Prints one JSON line: aggregate QPS, mean latency, batch statistics, and the
one-at-a-time rate of usearch_search_ef on the same index for comparison.  The clients are Python threads (the GIL is
released while a request is in flight), so the client side, not the device, bounds the rate reported here."""
import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # the launcher gives the runtime a hardware queue per service lane
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lantern_amd import capi  # noqa: E402

n, d, k, ef = 100_000, 128, 10, 64
clients, per = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(1)
base = rng.standard_normal((n, d), dtype=np.float32)
ix = capi.GpuIndex("l2sq", d, M=16, ef_construction=128, ef=ef, seed=42)
ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
ix.flush()
queries = np.random.default_rng(2).standard_normal((clients * per, d), dtype=np.float32)
want, _, _ = ix.search_batch(queries, k)
t0 = time.perf_counter()
for q in queries[:200]:
    ix.search(q, k)
single = 200 / (time.perf_counter() - t0)

srv = capi.ScanServer(index=ix, max_batch=256, max_wait_us=150)
ok, lat = [], []
start = threading.Barrier(clients + 1)


def session(t):
    c = capi.ScanClient(srv.host, srv.port)
    start.wait()
    good, acc = 0, 0.0
    for i in range(per):
        qi = t * per + i
        a = time.perf_counter()
        lab, _ = c.search(queries[qi], k)
        acc += time.perf_counter() - a
        good += int(np.array_equal(lab, want[qi]))
    ok.append(good)
    lat.append(acc / per)
    c.close()


ts = [threading.Thread(target=session, args=(t,)) for t in range(clients)]
[t.start() for t in ts]
start.wait()
t0 = time.perf_counter()
[t.join() for t in ts]
dt = time.perf_counter() - t0
st = srv.stats()
srv.stop()
print(json.dumps({"config": f"{n} x {d} f32 l2sq M=16 ef={ef} k={k}: {clients} clients x {per} single queries through the scan server",
                  "aggregate_qps": clients * per / dt, "mean_latency_us": float(np.mean(lat)) * 1e6, "identical_results": sum(ok) == clients * per,
                  "batches": st["batches"], "mean_batch": st["requests"] / max(st["batches"], 1), "largest_batch": st["largest_batch"],
                  "one_at_a_time_usearch_search_ef_qps": single, "window_us": 150}))
