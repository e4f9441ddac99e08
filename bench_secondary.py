"""bench_secondary.py -- the `secondary` array of bench.py's ONE JSON line (rank 0, one GPU).

The headline times device-resident 8192-query batches on the prescribed i.i.d. Gaussian set.  These entries put the other
ways the same hot path is called under the same (driver's) clock:

  c2_one_query_per_call     BASELINE config[1]: 100k x 128 f32 L2sq, M=16 ef=64 k=10, ONE usearch_search_ef per call -- the
                            reference's actual calling pattern (lantern_hnsw/src/hnsw/scan.c:220-228); wall and kernel time
  c3_cosine_1024_batches    BASELINE config[2]: 1M x 768 f32 cosine, 1024-query batches, one and two launches in flight
  c3_dense_exact_knn        BASELINE config[2]'s dense contraction: the exact k-NN of 1024 queries x 1M x 768 cosine through k_dense_f32
                            (fp32 MFMA): TFLOP/s of steady full-chunk launches and of the whole call against the 157.3 TFLOP/s
                            fp32-matrix peak, and the matrix pipe's busy fraction from an in-run rocprofv3 counter pass
  (the clustered set is no longer a secondary entry: it is the line's co-headline, bench.py clustered_coheadline)
  headline_host_buffers     the headline index through lantern_gpu_search_batch / _lane: queries and answers in HOST memory
                            (PCIe both ways inside the timed region) -- what a caller of the C ABI gets
  headline_scan_service     the headline index behind the scan-side service at 256 connections (one query per request, the
                            way PostgreSQL backends scan: scan.c:167-338), driven by lantern-scan-load

Every entry: `workload`, `value` + `unit`, `ms_per_step`, `recall_at_10`, `roofline` (algorithmic bytes of SURVEY 8d with D / E
counted on the device; counter bytes where exactly one launch is in flight and the pass is affordable), `cpu_baseline` (the CPU
port on the same graph, on ONE thread -- a PostgreSQL backend, utils.c:66 -- and on all cores; median of three repetitions each).  The oracle is used as the baseline only.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0
STREAMING_GBS = 6290.0  # MI355X_MICROARCH.md: measured float4 copy
GATHER_GBS = 5640.0     # profiles/r06_cache_model_calibration.md: uniformly random 3 KiB rows in the walk's launch shape, algorithmic bytes (r03 - r05: 6730)


def fractions(achieved_gbs: float) -> dict:
    return {"frac": achieved_gbs / HBM_PEAK_GBS, "frac_of_streaming": achieved_gbs / STREAMING_GBS, "frac_of_gather_ceiling": achieved_gbs / GATHER_GBS}


def _recall(found, truth, k):
    return float(np.mean([len(set(f.tolist()) & set(t.tolist())) / k for f, t in zip(found, truth)]))


def _build(capi, hip, metric, base, a):
    ix = capi.GpuIndex(metric, base.shape[1], M=a.M, ef_construction=a.efc, ef=a.ef, seed=42)
    ix.reserve(base.shape[0])
    ix.set_add_batch(a.add_batch, 16)
    hip.synchronize()
    t0 = time.perf_counter()
    ix.add_many(np.arange(base.shape[0], dtype=np.uint64) + 1, base)
    ix.flush()
    hip.synchronize()
    return ix, time.perf_counter() - t0


def _cpu_port(ix, base, queries, metric, a, seconds=4.0):
    """The CPU port on the identical graph, 1 thread and all cores: three timed repetitions each, median reported (bench_cpu.py)."""
    import bench_cpu
    from oracle import binding as oracle

    native = oracle.build_native() and oracle.use_native(True)
    ora = oracle.OracleIndex.from_graph(metric, base, ix.export_graph(), a.M, a.efc, a.ef, 42, oracle.SUM_FAST)
    rates, slots = bench_cpu.search_rates(ora, queries, a.k, a.ef, seconds)
    rates["build"] = bench_cpu.port_build_note(native)
    return rates, slots


def _over_cpu(value, cpu):
    return {"gpu_over_cpu_1_thread": value / cpu["value_1_thread"], "gpu_over_cpu_all_cores": value / cpu["value"]}


def _traffic_of(a, measure_traffic, ix, **shape):
    """Fabric-side bytes per launch of a secondary leg's own launch shape: ONE in-run rocprofv3 --pmc FETCH_SIZE pass over a re-execution
    of that shape (bench.py --pmc-child; same seeds, graph checksum checked).  Writes are not counted here (< 0.01 % of the reads in
    the headline's two-pass measurement).  (None, None) when rocprofv3 is absent or --no-pmc."""
    if measure_traffic is None:
        return None, None
    import copy

    b = copy.copy(a)
    b.n, b.dim, b.metric = shape["rows"], shape["dim"], shape["metric"]
    b.queries, b.query_batches, b.base_seed, b.query_seed, b.pmc_steps = shape["queries"], shape["query_batches"], shape["base_seed"], shape["query_seed"], shape["pmc_steps"]
    b.data, b.quant, b.pq_subvectors, b.data_scale = "gaussian", "f32", 0, 1.0
    det = measure_traffic(b, f"{ix.checksum():016x}", counters=("FETCH_SIZE",))
    if det and det.get("hbm_bytes_per_launch"):
        return det["hbm_bytes_per_launch"], det["source"]
    return None, (det or {}).get("passes")


FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md / SURVEY 8d: dense fp32 matrix (v_mfma_f32_32x32x2_f32: 256 CUs x 256 flop/cycle x 2.4 GHz)
DENSE_CHUNK = 65536             # columns of a full launch of the exact k-NN (index.cpp exact_knn_device_impl)


def _graphless_index(capi, metric, base):
    """Rows in HBM without a graph: all the exact k-NN needs."""
    n = base.shape[0]
    ix = capi.GpuIndex(metric, base.shape[1], M=4, ef_construction=8)
    g = {"levels": np.zeros(n, np.uint8), "nbr0": np.full((n, 8), 0xFFFFFFFF, np.uint32), "upper_off": np.full(n, 0xFFFFFFFF, np.uint32),
         "upper_nbr": np.zeros((0, 4), np.uint32), "labels": None, "entry_slot": 0, "max_level": 0}
    ix.import_graph(base, g)
    return ix


def dense_child(a, capi, hip):
    """`bench.py --dense-child`: the exact k-NN leg alone, for a rocprofv3 counter pass over k_dense_f32 (c3_dense_exact_knn)."""
    base = np.random.default_rng(3).standard_normal((a.n, a.dim), dtype=np.float32)
    queries = np.random.default_rng(4).standard_normal((1024, a.dim), dtype=np.float32)
    ix = _graphless_index(capi, "cos", base)
    for _ in range(3):
        slots, _ = ix.exact_search(queries, a.k)
    hip.synchronize()
    print(json.dumps({"dense_child": True, "calls": 3, "slot_checksum": int(slots.astype(np.uint64).sum())}), flush=True)


def c3_dense_exact_knn(a, capi, hip, base, keep=None):
    """BASELINE config[2] / north_star: "batched IP as MFMA GEMM, rocprof MFMA util".  The one dense contraction on the path: exact k-NN
    of 1024 queries x N x d (cosine) = k_dense_f32 (fp32 MFMA, csrc/bruteforce.hip) + fused top-k + exact re-rank; the reference's
    own dense site is the PQ k-means assignment (product_quantization.c:80-124), which runs on the same kernel.
    Algorithmic flops = 2 d per REQUIRED (query, row) pair (padding earns nothing)."""
    import bench_pmc

    n, d, nq, k = base.shape[0], base.shape[1], 1024, a.k
    queries = np.random.default_rng(4).standard_normal((nq, d), dtype=np.float32)  # C3: base seed 3, queries seed 4
    ix = (keep or {}).get("cos_index") or _graphless_index(capi, "cos", base)
    ix.exact_search(queries, k)  # first call: first touch of the row block, clock ramp (its launches are the "cold" ones)
    hip.synchronize()
    calls = 6
    capi.dense_profile(True)
    t0 = time.perf_counter()
    for _ in range(calls):
        slots, dists = ix.exact_search(queries, k)
    hip.synchronize()
    wall = (time.perf_counter() - t0) / calls
    recs = capi.dense_profile(False)
    full = [r["ms"] for r in recs if r["rows"] == nq and r["cols"] == DENSE_CHUNK and r["ms"] > 0]
    other = [r for r in recs if not (r["rows"] == nq and r["cols"] == DENSE_CHUNK)]
    flops_call = 2.0 * nq * n * d
    flops_full = 2.0 * nq * DENSE_CHUNK * d
    full_ms = float(np.mean(full)) if full else None
    per_pos = {}  # mean duration by position of the launch within its call (the first launches of a call follow the norms / memset kernels)
    pos = 0
    for r in recs:
        pos = 0 if (r["cols"] < DENSE_CHUNK and not r["fused"]) else pos + 1
        if r["rows"] == nq and r["cols"] == DENSE_CHUNK and r["ms"] > 0:
            per_pos.setdefault(pos, []).append(r["ms"])
    steady = flops_full / (full_ms * 1e-3) / 1e12 if full_ms else None
    kernel_ms_per_call = float(sum(r["ms"] for r in recs if r["ms"] > 0)) / calls
    # the same calls on the CPU port's brute force (one thread and all cores are both memory-bound f32 dot products)
    from oracle import binding as oracle

    import bench_cpu

    cores = bench_cpu.usable_cores()
    oracle.build_native() and oracle.use_native(True)
    sample = max(8, min(2 * cores, 64))
    t0 = time.perf_counter()
    truth, _ = oracle.bruteforce(base, queries[:sample], k, "cos", oracle.SUM_FAST, cores)
    cpu_s = (time.perf_counter() - t0) / sample
    agree = float(np.mean([len(set(x.tolist()) & set(y.tolist())) / k for x, y in zip(slots[:sample], truth)]))
    roof = {"bound": "mfma_fp32", "algorithmic_flops_per_call": flops_call, "algorithmic_flops_per_full_launch": flops_full,
            "avg_launch_ms": full_ms, "avg_launch_ms_basis": f"HIP events around the {len(full)} steady full-chunk launches (1024 x {DENSE_CHUNK} x {d}) of {calls} calls after a warm-up call",
            "achieved": steady, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": (steady / FP32_MATRIX_PEAK_TFLOPS) if steady else None,
            "full_launch_ms": {"mean": full_ms, "median": float(np.median(full)) if full else None, "min": float(np.min(full)) if full else None,
                               "max": float(np.max(full)) if full else None, "by_position_in_call": {str(k_): round(float(np.mean(v)), 4) for k_, v in sorted(per_pos.items())}},
            "launches_per_call": len(recs) / calls, "other_launches_ms": sorted({(r["cols"], round(r["ms"], 3)) for r in other})[:6],
            "contraction_ms_per_call": kernel_ms_per_call, "achieved_contraction_per_call": flops_call / (kernel_ms_per_call * 1e-3) / 1e12 if kernel_ms_per_call else None,
            "call_level": {"seconds_per_call_wall": wall, "achieved": flops_call / wall / 1e12, "frac": flops_call / wall / 1e12 / FP32_MATRIX_PEAK_TFLOPS,
                           "includes": "host query padding + H2D, row norms, the partial last chunk, fused top-k selection, exact re-rank, D2H of the answers"},
            "kernel": "k_dense_f32<cos, fused top-k> (csrc/bruteforce.hip: v_mfma_f32_32x32x2_f32, buffer_load ... lds, persistent, software-pipelined)",
            "traffic": None, "mfma_busy": None}
    if a.no_pmc or not bench_pmc.rocprof():
        roof["mfma_busy_note"] = "no counter pass (--no-pmc or rocprofv3 absent); the last committed one: profiles/r04_dense_mfma.md (0.892)"
    else:
        child = [sys.executable, os.path.join(ROOT, "bench.py"), "--dense-child", "--rows", str(n), "--dim", str(d), "--k", str(k)]
        r = bench_pmc.run_pass(child, "k_dense_f32", "k_dense_f32", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], "dense_child")
        if "error" in r:
            roof["mfma_busy_note"] = "counter pass failed: " + r["error"]
        else:
            v = r["values"]
            m, sq, gui = (np.array(v[c], dtype=np.float64) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
            if len(m) and len(m) == len(gui) == len(sq):
                big = gui >= 0.8 * gui.max()  # the full-chunk launches (a partial chunk is a third of one)
                simds, xcds, ses = 1024.0, 8.0, 32.0  # the counters are sums over 256 CUs x 4 SIMDs | 8 XCDs | 32 shader engines (profiles/r03_dense_mfma.md)
                busy_gui = (m[big] / simds) / (gui[big] / xcds)
                busy_sq = (m[big] / simds) / (sq[big] / ses)
                roof["mfma_busy"] = float(np.median(busy_gui))
                roof["mfma_busy_detail"] = {"definition": "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs, median over the full-chunk launches of the pass",
                                            "over_sq_busy_cycles": float(np.median(busy_sq)), "launches_counted": int(big.sum()), "launches_in_pass": int(len(m)),
                                            "mean_per_full_launch": {"SQ_VALU_MFMA_BUSY_CYCLES": float(m[big].mean()), "SQ_BUSY_CYCLES": float(sq[big].mean()), "GRBM_GUI_ACTIVE": float(gui[big].mean())},
                                            "effective_clock_GHz": float(np.median(gui[big] / xcds) / (full_ms * 1e-3) / 1e9) if full_ms else None,
                                            "command": r["command"], "seconds": r["seconds"]}
            else:
                roof["mfma_busy_note"] = f"counter rows do not line up ({len(m)} / {len(sq)} / {len(gui)})"
    return {"name": "c3_dense_exact_knn",
            "workload": f"BASELINE config[2]'s dense contraction: exact k-NN of {nq} queries x {n} x {d} f32 cos (k={k}) through k_dense_f32 (fp32 MFMA) + fused top-k + exact re-rank",
            "value": nq / wall, "unit": "queries/s", "ms_per_step": wall * 1e3, "step": f"one exact k-NN call of {nq} queries (host buffers)",
            "recall_at_10": 1.0, "recall_note": f"exact by construction; top-{k} overlap with the CPU port's brute force on {sample} queries: {agree}",
            "roofline": roof,
            "cpu_baseline": {"value": 1.0 / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
                             "sample": f"{sample} queries of this workload on {cores} threads (one query per thread), the CPU port's brute force over the same {n} rows"},
            "gpu_over_cpu": (nq / wall) * cpu_s}


class _Batches:
    """B resident query batches with their outputs; step(i) searches batch i mod B on stream i mod S."""

    def __init__(self, hip, ix, queries, nq, a, streams=1):
        self.hip, self.ix, self.nq, self.a = hip, ix, nq, a
        self.B = max(1, queries.shape[0] // nq)
        self.S = streams
        self.streams = [hip.Stream() for _ in range(streams)]
        self.lanes = []
        for i in range(self.B):
            qi = queries[i * nq:(i + 1) * nq]
            self.lanes.append({"dq": hip.Buffer.from_numpy(hip.padded_rows(qi, False)), "lab": hip.Buffer(nq * a.k * 8), "dist": hip.Buffer(nq * a.k * 4),
                               "slot": hip.Buffer(nq * a.k * 4), "D": hip.Buffer(nq * 8), "E": hip.Buffer(nq * 8)})

    def step(self, i):
        L, a = self.lanes[i % self.B], self.a
        self.ix.search_batch_device(L["dq"].ptr, self.nq, a.k, a.ef, 0, L["lab"].ptr, L["dist"].ptr, L["slot"].ptr, None, L["D"].ptr, L["E"].ptr,
                                    self.streams[i % self.S].handle)

    def timed(self, steps, warmup=None):
        hip = self.hip
        for i in range(self.B if warmup is None else warmup):
            self.step(i)
        hip.synchronize()
        ev = [(hip.Event(), hip.Event()) for _ in range(steps)]
        t0 = time.perf_counter()
        for i, (s, e) in enumerate(ev):
            st = self.streams[i % self.S].handle
            s.record(st)
            self.step(i)
            e.record(st)
        hip.synchronize()
        elapsed = time.perf_counter() - t0
        return elapsed, [s.elapsed_ms(e) for s, e in ev]

    def algorithmic_bytes(self, row_bytes, list_bytes, steps):
        per = []
        for L in self.lanes:
            D = L["D"].download(self.nq, np.uint64).astype(np.float64)
            E = L["E"].download(self.nq, np.uint64).astype(np.float64)
            per.append((float((D * row_bytes + E * list_bytes + row_bytes).sum()), float(D.mean()), float(E.mean())))
        return float(np.mean([per[i % self.B][0] for i in range(steps)])), per[0][1], per[0][2]


def c2_one_query_per_call(a, capi, hip, measure_traffic=None):
    n, d, nq = 100_000, 128, 2000
    base = np.random.default_rng(1).standard_normal((n, d), dtype=np.float32)  # seeds 1 / 2
    queries = np.random.default_rng(2).standard_normal((nq, d), dtype=np.float32)
    ix, t_build = _build(capi, hip, "l2sq", base, a)
    for q in queries[:50]:
        ix.search(q, a.k, a.ef)
    lat, found = [], []
    t_all = time.perf_counter()
    for q in queries:
        t0 = time.perf_counter()
        lab, _ = ix.search(q, a.k, a.ef)
        lat.append(time.perf_counter() - t0)
        found.append(lab.astype(np.int64) - 1)
    t_all = time.perf_counter() - t_all
    lat = np.array(lat) * 1e6
    # the kernel alone: one query per launch, HIP events on the launch stream
    st = hip.Stream()
    dq = hip.Buffer.from_numpy(hip.padded_rows(queries, False))
    d_lab, d_dst, d_D, d_E = hip.Buffer(a.k * 8), hip.Buffer(a.k * 4), hip.Buffer(8), hip.Buffer(8)
    row_bytes = hip.padded_rows(queries[:1], False).shape[1] * 4
    kern, Ds, Es = [], [], []
    for i in range(500):
        s, e = hip.Event(), hip.Event()
        s.record(st.handle)
        ix.search_batch_device(dq.ptr + i * row_bytes, 1, a.k, a.ef, 0, d_lab.ptr, d_dst.ptr, None, None, d_D.ptr, d_E.ptr, st.handle)
        e.record(st.handle)
        st.synchronize()
        kern.append(s.elapsed_ms(e) * 1e3)
        Ds.append(int(d_D.download(1, np.uint64)[0]))
        Es.append(int(d_E.download(1, np.uint64)[0]))
    truth, _ = ix.exact_search(queries[:512], a.k)
    cpu, _ = _cpu_port(ix, base, queries, "l2sq", a, seconds=3.0)
    bytes_q = float(np.mean(Ds)) * d * 4 + float(np.mean(Es)) * (2 * a.M * 4) + d * 4
    gbs = bytes_q / (float(np.mean(kern)) * 1e-6) / 1e9
    traffic, src = _traffic_of(a, measure_traffic, ix, rows=n, dim=d, metric="l2sq", queries=1, query_batches=64, base_seed=1, query_seed=2, pmc_steps=64)
    return {"name": "c2_one_query_per_call",
            "workload": f"BASELINE config[1]: HNSW search {n}x{d} f32 l2sq M={a.M} ef_construction={a.efc} ef={a.ef} k={a.k}, {nq} x ONE usearch_search_ef per call (scan.c:220-228)",
            "value": nq / t_all, "unit": "queries/s", "ms_per_step": float(lat.mean()) / 1e3, "step": "one usearch_search_ef call (host query in, host answer out)",
            "us_per_call_wall": {"mean": float(lat.mean()), "p50": float(np.median(lat)), "p99": float(np.percentile(lat, 99))},
            "us_per_call_kernel": {"mean": float(np.mean(kern)), "p50": float(np.median(kern))},
            "hops_per_query": float(np.mean(Es)), "dist_evals_per_query": float(np.mean(Ds)), "us_per_hop_kernel": float(np.mean(kern) / max(np.mean(Es), 1)),
            "recall_at_10": _recall(np.array(found[:512]), truth, a.k),
            "roofline": dict({"bound": "latency (one dependent hop chain; HBM fraction shown for scale)", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "algorithmic_bytes_per_launch": bytes_q, "avg_launch_ms": float(np.mean(kern)) / 1e3, "kernel": "k_search_spec (walk_spec.hpp), one query",
                              "traffic": traffic, "traffic_source": src, "traffic_over_algorithmic": (traffic / bytes_q) if traffic else None,
                              "frac_is": "frac_algorithmic (SURVEY 8d bytes / HIP-event launch time / 8 TB/s)"}, **fractions(gbs)),
            "cpu_baseline": cpu, **_over_cpu(nq / t_all, cpu), "build_vectors_per_s": n / t_build}


def c3_cosine_1024_batches(a, capi, hip, base, measure_traffic=None, keep=None):
    n, d, nq, B = base.shape[0], base.shape[1], 1024, 8
    ix, t_build = _build(capi, hip, "cos", base, a)
    queries = np.random.default_rng(4).standard_normal((nq * B, d), dtype=np.float32)  # base seed 3, queries seed 4
    out = {"name": "c3_cosine_1024_batches",
           "workload": f"BASELINE config[2]: HNSW search {n}x{d} f32 cos M={a.M} ef_construction={a.efc} ef={a.ef} k={a.k}, {nq}-query batches resident in HBM"}
    row, lst = d * 4, 2 * a.M * 4
    one = _Batches(hip, ix, queries, nq, a, streams=1)
    steps = 24
    elapsed, kms = one.timed(steps)
    bytes_l, Dm, Em = one.algorithmic_bytes(row, lst, steps)
    gbs1 = bytes_l / (float(np.mean(kms)) * 1e-3) / 1e9
    truth, _ = ix.exact_search(queries[:nq], a.k)
    found = one.lanes[0]["slot"].download((nq, a.k), np.uint32)
    two = _Batches(hip, ix, queries, nq, a, streams=2)
    elapsed2, _ = two.timed(steps)
    gbs2 = bytes_l * steps / elapsed2 / 1e9
    cpu, _ = _cpu_port(ix, base, queries[:nq], "cos", a, seconds=5.0)
    traffic, src = _traffic_of(a, measure_traffic, ix, rows=n, dim=d, metric="cos", queries=nq, query_batches=B, base_seed=3, query_seed=4, pmc_steps=8)
    if keep is not None:
        keep["cos_index"] = ix  # the dense leg computes over the same rows
    out.update({"value": nq * steps / elapsed, "unit": "queries/s", "ms_per_step": elapsed / steps * 1e3, "step": "one 1024-query launch, one launch in flight",
                "two_launches_in_flight": {"value": nq * steps / elapsed2, "ms_per_step": elapsed2 / steps * 1e3,
                                           "roofline": dict({"achieved": gbs2, "unit": "GB/s", "basis": "all launches' algorithmic bytes / the timed region"}, **fractions(gbs2))},
                "recall_at_10": _recall(found, truth, a.k), "recall_note": "i.i.d. N(0,1) x 768 under cosine has no neighbourhood structure: the prescribed set, identical on CPU and GPU",
                "dist_evals_per_query": Dm, "expansions_per_query": Em,
                "roofline": dict({"bound": "hbm", "achieved": gbs1, "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_launch": bytes_l,
                                  "avg_launch_ms": float(np.mean(kms)), "kernel": "k_search", "traffic": traffic, "traffic_source": src,
                                  "traffic_over_algorithmic": (traffic / bytes_l) if traffic else None,
                                  "achieved_fabric": (traffic / (float(np.mean(kms)) * 1e-3) / 1e9) if traffic else None,
                                  "frac_fabric": (traffic / (float(np.mean(kms)) * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                                  "frac_is": "frac_algorithmic (SURVEY 8d bytes; D / E counted on the device) / HIP-event launch time / 8 TB/s; frac_fabric = this run's counter bytes"},
                                 **fractions(gbs1)),
                "cpu_baseline": cpu, **_over_cpu(nq * steps / elapsed, cpu), "build_vectors_per_s": n / t_build})
    return out


def headline_host_buffers(a, ix, queries, device_resident_qps, headline_recall):
    """lantern_gpu_search_batch (one call at a time) and lantern_gpu_search_batch_lane on 2 / 4 lanes (one thread per lane, every
    thread handing over its own host batches back to back: the copy-in and padding of one batch run over the search of another)."""
    nq, k = a.queries, a.k
    B = max(1, queries.shape[0] // nq)
    batches = [np.ascontiguousarray(queries[i * nq:(i + 1) * nq]) for i in range(B)]
    ix.search_batch(batches[0], k, a.ef)
    steps = 8
    t0 = time.perf_counter()
    for i in range(steps):
        lab, _, _ = ix.search_batch(batches[i % B], k, a.ef)
    one = time.perf_counter() - t0
    res = {"one_call_at_a_time": {"value": nq * steps / one, "ms_per_step": one / steps * 1e3, "over_device_resident": nq * steps / one / device_resident_qps}}
    for lanes in (2, 4):
        per = 6
        for ln in range(lanes):
            ix.search_batch_lane(ln, batches[ln % B], k, a.ef)
        go = threading.Barrier(lanes + 1)

        def worker(ln):
            go.wait()
            for i in range(per):
                ix.search_batch_lane(ln, batches[(ln + i) % B], k, a.ef)

        ts = [threading.Thread(target=worker, args=(ln,)) for ln in range(lanes)]
        [t.start() for t in ts]
        go.wait()
        t0 = time.perf_counter()
        [t.join() for t in ts]
        dt = time.perf_counter() - t0
        res[f"{lanes}_lanes"] = {"value": nq * per * lanes / dt, "ms_per_step": dt / (per * lanes) * 1e3, "over_device_resident": nq * per * lanes / dt / device_resident_qps}
    best = max(res.values(), key=lambda r: r["value"])
    bytes_each_way = {"queries_in": nq * a.dim * 4, "answers_out": nq * k * 12 + nq * 4}
    return {"name": "headline_host_buffers",
            "workload": f"the headline index ({a.n}x{a.dim} f32 {a.metric}) through lantern_gpu_search_batch[_lane]: {nq}-query batches in HOST memory, answers to HOST memory (PCIe inside the timed region)",
            "value": best["value"], "unit": "queries/s", "ms_per_step": best["ms_per_step"], "step": f"one {nq}-query host batch (pad + H2D + k_search + D2H + copy-out)",
            "modes": res, "device_resident_value": device_resident_qps, "over_device_resident": best["over_device_resident"],
            "pcie_bytes_per_step": bytes_each_way, "recall_at_10": headline_recall, "recall_note": "the headline's queries and graph: the same answers as the device-resident batch",
            "roofline": None, "cpu_baseline": None}


def headline_scan_service(a, capi, ix, device_resident_qps, connections=256, seconds=3.0):
    tool = os.path.join(ROOT, "lantern_amd", "lib", "lantern-scan-load")
    if not os.path.exists(tool):
        return {"name": "headline_scan_service", "error": "lantern-scan-load not built"}
    srv = capi.ScanServer(index=ix, max_batch=1024, max_wait_us=200)
    try:
        runs = {}
        for mode, extra in (("thread_per_connection", []), ("multiplexed_8_client_threads", ["--client-threads", "8"])):
            cmd = [tool, "--port", str(srv.port), "--dim", str(a.dim), "--k", str(a.k), "--connections", str(connections), "--seconds", str(seconds),
                   "--warmup-seconds", "1", "--rows", str(a.n), "--m", str(a.M), "--ef-construction", str(a.efc), "--ef", str(a.ef)] + extra
            before, tb = srv.stats(), srv.timing()
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
            after, ta = srv.stats(), srv.timing()
            line = next((json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")), None)
            if not line:
                runs[mode] = {"error": (p.stderr or p.stdout)[-300:]}
                continue
            req, bat = after["requests"] - before["requests"], after["batches"] - before["batches"]
            dn = max(ta["requests"] - tb["requests"], 1)
            runs[mode] = {"value": line["queries_per_s"], "latency_us": line["latency_us"], "failures": line["failures"], "connected": line["connected"],
                          "mean_batch": req / max(bat, 1), "largest_batch": after["largest_batch"],
                          "server_side_us": {k2: (ta[k2] * ta["requests"] - tb[k2] * tb["requests"]) / dn
                                             for k2 in ("wait_for_batch_us", "batch_closed_to_answer_us", "answer_to_socket_us")}}
    finally:
        srv.stop()
    good = [r for r in runs.values() if "value" in r]
    if not good:
        return {"name": "headline_scan_service", "error": runs}
    best = max(good, key=lambda r: r["value"])
    return {"name": "headline_scan_service",
            "workload": f"the headline index ({a.n}x{a.dim} f32 {a.metric}) behind the scan-side service: {connections} connections, one k={a.k} query per request (scan.c:167-338), lantern-scan-load over loopback TCP",
            "value": best["value"], "unit": "queries/s", "ms_per_step": best["latency_us"]["p50"] / 1e3, "step": "one request on one connection (p50 round trip)",
            "modes": runs, "device_resident_value": device_resident_qps, "over_device_resident": best["value"] / device_resident_qps,
            "recall_at_10": None, "recall_note": "answers == the direct batch (tests/test_scan_server.py); queries here are the load generator's own",
            "roofline": None, "cpu_baseline": None}


def run(a, capi, hip, ix, base, queries, device_resident_qps, headline_recall, measure_traffic):
    """The four (five) entries, each isolated: a failure costs its entry, never the line."""
    out = []

    only = [x for x in os.environ.get("LANTERN_BENCH_SECONDARY", "").split(",") if x]  # debugging: run the named legs only

    def leg(fn, *args, **kw):
        if only and fn.__name__ not in only:
            return
        t0 = time.time()
        print(f"[bench secondary] {fn.__name__} ...", file=sys.stderr, flush=True)
        try:
            e = fn(*args, **kw)
        except Exception as ex:  # noqa: BLE001
            e = {"name": fn.__name__, "error": repr(ex)[:400]}
        e["seconds"] = time.time() - t0
        out.append(e)

    leg(headline_host_buffers, a, ix, queries, device_resident_qps, headline_recall)
    leg(headline_scan_service, a, capi, ix, device_resident_qps)
    leg(c2_one_query_per_call, a, capi, hip, measure_traffic)
    if a.metric == "l2sq" and a.data == "gaussian":
        keep = {}
        leg(c3_cosine_1024_batches, a, capi, hip, base, measure_traffic, keep)
        leg(c3_dense_exact_knn, a, capi, hip, base, keep)
    return out
