"""bench_cpu.py -- the `cpu_baseline` leg of bench.py / bench_secondary.py: the CPU port (oracle/hnsw.c, a restatement of the usearch
path -- the reference binary itself cannot be built here) timed on this host's cores on the SAME graph.

Procedure: warm-up excluded; (a) 1 thread -- a PostgreSQL backend (utils.c:66); (b) all usable cores, one
query per thread -- the external indexer's model (server.rs:317-359); THREE timed repetitions per leg, the MEDIAN reported with
min / max beside it.  Samples are sized from a probe so that the whole leg stays near its budget.

This module is a baseline harness only: nothing under lantern_amd/ imports it, and the numbers it returns are never `value`.
"""
from __future__ import annotations

import os
import time

import numpy as np

REPS = 3


def usable_cores() -> int:
    """Threads this process may really use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def _stats(rates):
    return {"median": float(np.median(rates)), "min": float(np.min(rates)), "max": float(np.max(rates)), "repetitions": len(rates)}


def search_rates(ora, queries, k, ef, seconds, cores=None, reps=REPS):
    """Queries/s of `ora.search_batch` on 1 thread and on `cores` threads: `reps` timed repetitions each, median + min / max.
    `seconds` is the budget of the whole leg (about 30 % for the 1-thread repetitions, 70 % for the all-cores ones).
    Returns (dict, slots of the last all-cores repetition's first len(queries) answers)."""
    cores = cores or usable_cores()
    nq = queries.shape[0]
    probe = min(32, nq)
    ora.search_batch(queries[:probe], k, ef, 1)  # warm-up: first touch of the visited set, page faults of the graph
    t0 = time.perf_counter()
    ora.search_batch(queries[:probe], k, ef, 1)
    per_q = (time.perf_counter() - t0) / probe
    n1 = int(max(16, min(nq, (seconds * 0.3 / reps) / max(per_q, 1e-9))))
    r1 = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ora.search_batch(queries[:n1], k, ef, 1)
        r1.append(n1 / (time.perf_counter() - t0))
    # all cores: a short probe sizes the sample (parallel efficiency is host-dependent).  The sample cycles through the given
    # queries: many threads need >10^4 queries to reach steady state (each first faults in its own visited-set array).
    pq = np.ascontiguousarray(np.tile(queries, (max(1, (cores * 32) // nq + 1), 1))[: cores * 32])
    ora.search_batch(pq, k, ef, cores)  # warm-up of every thread's own state
    t0 = time.perf_counter()
    ora.search_batch(pq, k, ef, cores)
    qps_probe = pq.shape[0] / (time.perf_counter() - t0)
    want = int(seconds * 0.7 / reps * qps_probe)
    tile = int(max(1, min(64, -(-want // nq))))
    tiled = np.ascontiguousarray(np.tile(queries, (tile, 1)))
    nall = tiled.shape[0]
    rall, slots = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        _, _, slots, _, _ = ora.search_batch(tiled, k, ef, cores)
        rall.append(nall / (time.perf_counter() - t0))
    out = {"value": float(np.median(rall)), "unit": "queries/s", "cores": cores, "kind": "port",
           "all_cores": _stats(rall), "value_1_thread": float(np.median(r1)), "one_thread": _stats(r1),
           "us_per_query_1_thread": 1e6 / float(np.median(r1)),
           "procedure": f"warm-up excluded, {reps} timed repetitions per leg, median reported (BASELINE.md section 3)",
           "sample": f"{nall} queries per repetition ({nq} of this workload cycled x{tile}) on {cores} threads, one query per thread "
                     f"(server.rs:317-359 model); {n1} queries per repetition on 1 thread (a PostgreSQL backend, utils.c:66); same graph"}
    return out, slots[:nq] if slots is not None else None


def port_build_note(native: bool) -> str:
    return ("gcc -O3 -march=native + the reference's -fassociative-math flags" if native
            else "gcc -O3 -march=x86-64-v3 + the reference's -fassociative-math flags")
