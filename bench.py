#!/usr/bin/env python
"""bench.py -- the driver's benchmark contract for the HNSW distance-evaluation hot path.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`): 1M x 768 f32, L2sq, M=16 ef_construction=128, ef=64, k=10.
A "step" is one pass of the hot path (usearch_search_ef semantics, lantern_hnsw/src/hnsw/scan.c:220)
over one batch of synthetic queries that is already resident in HBM.  Queries shard across GPUs
with no collective (the index is replicated in each GPU's HBM), so scaling is weak: every rank
searches its own `--queries` per step and `value` = all ranks' queries / max-over-ranks time.

Besides the contract's fields the JSON line carries
  roofline      algorithmic bytes of the search kernel (D*d*4 + E*2M*4 + d*4 per query,
                D and E counted on the device) / its average launch time (HIP events on the launch
                stream) against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (a port of the usearch algorithm: oracle/hnsw.c) on the SAME graph,
                timed on this host's cores on a bounded sample (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--rows", dest="n", type=int, default=1_000_000, help="indexed vectors (not --n: torchrun's own parser trips over that prefix)")
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--metric", default="l2sq")
    p.add_argument("--M", type=int, default=16)
    p.add_argument("--efc", type=int, default=128)
    p.add_argument("--ef", type=int, default=64)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--queries", type=int, default=8192, help="queries per step per GPU")
    p.add_argument("--waves", type=int, default=4, help="wavefronts per query")
    p.add_argument("--max-wg", type=int, default=0)
    p.add_argument("--add-batch", type=int, default=8192)
    p.add_argument("--truth-queries", type=int, default=1024, help="queries used for recall@k")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget (0 = skip)")
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--quant", default="f32", choices=["f32", "f16", "i8"], help="storage kind (reloption quant_bits 32 / 16 / 8); the headline config is f32")
    p.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) for real multi-GPU runs; gloo only to debug the N>1 path on one GPU")
    p.add_argument("--data", default="gaussian", choices=["gaussian", "lowrank"],
                   help="gaussian = the prescribed i.i.d. N(0,1) set (SURVEY 8d); lowrank = 32 latent dims embedded in --dim (embedding-like)")
    p.add_argument("--data-scale", type=float, default=1.0, help="multiply the synthetic rows and queries (i8 storage quantises [-1, 1]: use 0.3)")
    p.add_argument("--sharded-build", default="auto", choices=["auto", "on", "off"],
                   help="after the search measurement, build the same index ONCE across all ranks (RCCL all-gathers, one child "
                        "process per GPU: lantern_amd/sharded_build.py) and report its rate; auto = when --gpus > 1")
    p.add_argument("--sharded-build-timeout", type=float, default=420.0)
    return p.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        # torch is plumbing here: rendezvous, barrier and the max-over-ranks reduction over RCCL.  It must
        # be imported before the HIP library so both share one HIP runtime (lantern_amd/capi.py note).
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(local_rank % ndev)
        if a.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank % ndev))
        else:
            dist.init_process_group(backend=a.dist_backend)

    from lantern_amd import capi, hip

    assert capi.device_count() > 0, "no HIP device: bench.py measures the HIP path only"
    dev_index = (local_rank % capi.device_count()) if world > 1 else 0
    hip.set_device(dev_index)

    # ---- synthetic data (numpy default_rng, standard normal f32, seeds 3 / 4) -------
    from lantern_amd import synth

    t0 = time.time()
    make_queries = synth.query_maker(a.data, a.dim)
    base = synth.base_rows(a.data, a.n, a.dim)
    if a.data_scale != 1.0:
        raw_queries = make_queries
        make_queries = lambda r, n: raw_queries(r, n) * np.float32(a.data_scale)
        base *= np.float32(a.data_scale)
    labels = np.arange(a.n, dtype=np.uint64) + 1  # 0 is INVALID_ELEMENT_LABEL (hnsw.h:40)
    t_gen = time.time() - t0

    # ---- build the index on this rank's GPU (replica per GPU; deterministic, so all replicas match)
    ix = capi.GpuIndex(a.metric, a.dim, M=a.M, ef_construction=a.efc, ef=a.ef, seed=42, quantization=a.quant)
    ix.reserve(a.n)
    ix.set_add_batch(a.add_batch, 16)
    ix.set_search_shape(a.waves, a.max_wg)
    hip.synchronize()
    t0 = time.time()
    ix.add_many(labels, base)
    ix.flush()
    hip.synchronize()
    t_build = time.time() - t0
    build_counters = ix.counters()

    # ---- this rank's queries, resident in HBM ----------------------------------------------------
    qrng = np.random.default_rng(4 + 1000 * rank)
    nq = a.queries
    queries = make_queries(qrng, nq)
    dq = hip.Buffer.from_numpy(hip.padded_rows(queries, False, a.quant == "f16", a.quant == "i8"))
    d_lab, d_dist, d_slot = hip.Buffer(nq * a.k * 8), hip.Buffer(nq * a.k * 4), hip.Buffer(nq * a.k * 4)
    d_D, d_E = hip.Buffer(nq * 8), hip.Buffer(nq * 8)
    stream = hip.Stream()  # the launch stream; the events below are recorded on it

    def step():
        ix.search_batch_device(dq.ptr, nq, a.k, a.ef, 0, d_lab.ptr, d_dist.ptr, d_slot.ptr, None, d_D.ptr, d_E.ptr, stream.handle)

    def barrier():
        hip.synchronize()
        if world > 1:
            dist.barrier()
        hip.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    ev = [(hip.Event(), hip.Event()) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for s, e in ev:
        s.record(stream.handle)
        step()
        e.record(stream.handle)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [s.elapsed_ms(e) for s, e in ev]  # HIP events on the launch stream: one search launch each
    if world > 1:
        import torch

        from lantern_amd import sharded

        elapsed = sharded.max_over_ranks(elapsed, device=torch.device("cuda", dev_index) if a.dist_backend == "nccl" else None)

    # ---- work-sharded build of the same index across all ranks, in child processes ----------
    sharded_res = None
    if a.sharded_build == "on" or (a.sharded_build == "auto" and world > 1):
        sharded_res = sharded_build_leg(a, rank, world, dev_index, dist, ix if rank == 0 else None)

    # ---- algorithmic bytes of one launch --------------------------------------------
    D = d_D.download(nq, np.uint64).astype(np.float64)
    E = d_E.download(nq, np.uint64).astype(np.float64)
    row_bytes = a.dim * {"f32": 4, "f16": 2, "i8": 1}[a.quant]
    bytes_per_launch = float((D * row_bytes + E * (2 * a.M * 4) + row_bytes).sum())
    avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
    achieved = bytes_per_launch / avg_kernel_s / 1e9

    out = None
    if rank == 0:
        # ---- recall@k against exact f32 k-NN (fp32-MFMA contraction + exact re-rank) --------------
        tq = min(a.truth_queries, nq)
        t0 = time.time()
        truth, _ = ix.exact_search(queries[:tq], a.k)
        t_truth = time.time() - t0
        found = d_slot.download((nq, a.k), np.uint32)[:tq]
        recall = float(np.mean([len(set(f.tolist()) & set(t.tolist())) / a.k for f, t in zip(found, truth)]))

        cpu = None
        if world == 1 and not a.no_cpu and a.cpu_seconds > 0:
            cpu = cpu_baseline(a, ix, base, queries, found)

        traffic = None
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                rec = json.load(open(prof))
                key = f"{a.n}x{a.dim}_{a.metric}_ef{a.ef}_q{nq}_w{a.waves}" + ("" if a.quant == "f32" else "_" + a.quant)
                traffic = rec.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None

        qps = world * nq * a.steps / elapsed
        out = {
            "metric": f"QPS (recall@{a.k} alongside), {a.n}x{a.dim} f32 {a.metric} ef={a.ef} k={a.k}",
            "value": qps,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "f16": "f32 arithmetic on f16 storage", "i8": "int32 arithmetic on i8 storage"}[a.quant],
            "data": ("synthetic" if a.data == "gaussian" else "synthetic (low-rank)") + ("" if a.data_scale == 1.0 else f" x {a.data_scale}"),
            "config": {"workload": f"HNSW search {a.n}x{a.dim} {a.quant} {a.metric} M={a.M} ef_construction={a.efc} ef={a.ef} k={a.k}",
                       "queries_per_step_per_gpu": nq, "global_queries_per_step": nq * world, "waves_per_query": a.waves,
                       "parallelism": f"replicated index, query batch sharded x{world}, no collective"},
            f"recall_at_{a.k}": recall,
            "recall_queries": tq,
            "build_vectors_per_s": a.n / t_build,
            "build_seconds": t_build,
            "build_batches": build_counters["add_batches"],
            "build_counters_per_vector": {k: build_counters[k] / a.n for k in ("add_walk_evals", "add_select_evals", "add_revlink_evals",
                                                                                 "add_reprunes", "add_expansions")},
            "dist_evals_per_query": float(D.mean()),
            "expansions_per_query": float(E.mean()),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_search", "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": avg_kernel_s * 1e3},
            "cpu_baseline": cpu,
            "sharded_build": sharded_res,
            "setup_seconds": {"datagen": t_gen, "build": t_build, "exact_truth": t_truth},
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def sharded_build_leg(a, rank, world, dev_index, dist, ix0):
    """One lantern_amd.sharded_build child per rank (own process: ROCm's HIP + RCCL, no torch), all building ONE index
    together.  Returns rank 0's summary (None on the other ranks); a failure is reported, never raised: this leg runs
    after the search measurement and must not cost the benchmark line."""
    import shutil
    import subprocess
    import tempfile

    rdv = None
    try:
        if rank == 0:
            rdv = tempfile.mkdtemp(prefix="lantern_rdv_")
        if world > 1:
            box = [rdv]
            dist.broadcast_object_list(box, src=0)
            rdv = box[0]
        # whatever happens to this rank's child, the rank still takes part in the gather below (a rank that skipped it
        # would leave its peers waiting in a collective)
        wall = 0.0
        try:
            cmd = [sys.executable, "-m", "lantern_amd.sharded_build", "--rank", str(rank), "--world", str(world), "--rendezvous", rdv,
                   "--device", str(dev_index), "--rows", str(a.n), "--dim", str(a.dim), "--metric", a.metric, "--M", str(a.M),
                   "--efc", str(a.efc), "--ef", str(a.ef), "--add-batch", str(a.add_batch), "--quant", a.quant, "--data", a.data,
                   "--data-scale", str(a.data_scale)]
            env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
            env.setdefault("NCCL_SOCKET_IFNAME", "lo")  # all ranks are on this node; the container's hostname may not resolve
            t0 = time.time()
            try:
                cp = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=a.sharded_build_timeout)
                rc, text = cp.returncode, cp.stdout + cp.stderr
            except subprocess.TimeoutExpired as e:
                rc, text = -9, f"timed out after {a.sharded_build_timeout} s: " + str(e.stdout or "")[-400:]
            wall = time.time() - t0
            res = None
            for line in text.splitlines():
                if line.startswith("SHARDED_BUILD "):
                    res = json.loads(line[len("SHARDED_BUILD "):])
            mine = res if (rc == 0 and res) else {"error": f"rank {rank}: exit {rc}: " + text[-600:]}
        except Exception as e:  # noqa: BLE001
            mine = {"error": f"rank {rank}: {e!r}"}
        if world > 1:
            every = [None] * world
            dist.all_gather_object(every, mine)
        else:
            every = [mine]
        if rank != 0:
            return None
        bad = [r for r in every if "error" in r]
        if bad:
            return {"error": bad[0]["error"], "ranks_failed": len(bad), "world": world}
        secs = max(r["seconds"] for r in every)
        sums = sorted({r["checksum"] for r in every})
        ref = f"{ix0.checksum():016x}"
        return {
            "world": world, "seconds": secs, "vectors_per_s": a.n / secs, "child_wall_seconds": wall,
            "replicas_identical": len(sums) == 1, "identical_to_single_gpu_build": sums == [ref],
            "checksum": sums[0], "single_gpu_checksum": ref,
            "bytes_received_per_rank": [r["exchange"]["bytes_received"] for r in every],
            "collectives": every[0]["exchange"]["collectives"],
            "walk_evals_per_rank": [r["counters"]["add_walk_evals"] for r in every],
            "transport": every[0]["transport"],
        }
    except Exception as e:  # noqa: BLE001 -- reported in the line
        return {"error": repr(e)} if rank == 0 else None
    finally:
        if rank == 0 and rdv:
            shutil.rmtree(rdv, ignore_errors=True)


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


def cpu_baseline(a, ix, base, queries, gpu_found):
    """The oracle (CPU port of the usearch path) on the identical graph, on this host's cores."""
    from oracle import binding as oracle

    native = oracle.build_native() and oracle.use_native(True)  # best CPU code for the baseline: -march=native on this host
    cores = usable_cores()
    g = ix.export_graph()
    mode = oracle.SUM_FAST
    if a.quant == "f16":  # the CPU port works on the rounded values (f32 arithmetic, no conversion cost: favours the CPU)
        base, queries = oracle.round_f16(base), oracle.round_f16(queries)
    if a.quant == "i8":  # the quantised integers held as f32 (int32 accumulation in the port)
        base, queries, mode = oracle.quantize_i8(base), oracle.quantize_i8(queries), oracle.SUM_I8
    ora = oracle.OracleIndex.from_graph(a.metric, base, g, a.M, a.efc, a.ef, 42, mode)
    # size the samples from a short probe so the whole leg stays near the budget (about 30 % of it
    # for the 1-thread leg, 70 % for the all-cores leg).  The all-cores sample cycles through the
    # step's query set: 256 threads need >10^5 queries to reach steady state (each thread first
    # faults in its own visited-set array), far more than one GPU step holds.
    probe = min(64, queries.shape[0])
    t0 = time.perf_counter()
    ora.search_batch(queries[:probe], a.k, a.ef, 1)
    per_q_1t = (time.perf_counter() - t0) / probe
    n1 = int(max(16, min(queries.shape[0], (a.cpu_seconds * 0.3) / per_q_1t)))
    t0 = time.perf_counter()
    ora.search_batch(queries[:n1], a.k, a.ef, 1)
    qps_1t = n1 / (time.perf_counter() - t0)
    # a short all-cores probe sizes the timed all-cores sample (parallel efficiency is host-dependent)
    pq = np.ascontiguousarray(np.tile(queries, (max(1, (cores * 32) // queries.shape[0] + 1), 1))[: cores * 32])
    t0 = time.perf_counter()
    ora.search_batch(pq, a.k, a.ef, cores)
    qps_probe = pq.shape[0] / (time.perf_counter() - t0)
    want = int(a.cpu_seconds * 0.6 * qps_probe)
    reps = int(max(1, min(64, -(-want // queries.shape[0]))))
    tiled = np.ascontiguousarray(np.tile(queries, (reps, 1)))
    nall = tiled.shape[0]
    t0 = time.perf_counter()
    _, _, slots, _, _ = ora.search_batch(tiled, a.k, a.ef, cores)
    qps_all = nall / (time.perf_counter() - t0)
    del tiled
    # index build on the CPU: the port's sequential usearch_add (a PostgreSQL backend builds with one thread,
    # utils.c:66) on a bounded prefix of the same rows.  The rate falls as the graph grows, so this flatters the CPU.
    nb = int(min(base.shape[0], 4096))
    cb = oracle.OracleIndex(a.metric, a.dim, M=a.M, ef_construction=a.efc, ef=a.ef, seed=42, sum_mode=mode)
    t0 = time.perf_counter()
    cb.add_many(np.arange(nb, dtype=np.uint64) + 1, base[:nb])
    cpu_build = nb / (time.perf_counter() - t0)
    del cb
    m = min(nall, gpu_found.shape[0], queries.shape[0])
    agree = float(np.mean([len(set(x.tolist()) & set(y.tolist())) / a.k for x, y in zip(slots[:m], gpu_found[:m])]))
    return {"value": qps_all, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{nall} queries (the step's {queries.shape[0]} cycled x{reps}) on {cores} threads, one query per thread "
                      f"(server.rs:317-359 model); {n1} queries on 1 thread (a PostgreSQL backend, utils.c:66)",
            "value_1_thread": qps_1t, "topk_overlap_with_gpu": agree,
            "build_vectors_per_s_1_thread": cpu_build, "build_sample": f"first {nb} rows, sequential usearch_add",
            "build": "gcc -O3 -march=native + the reference's -fassociative-math flags" if native else "gcc -O3 -march=x86-64-v3 + the reference's -fassociative-math flags",
            "note": "oracle/hnsw.c restates the usearch algorithm; the reference binary itself cannot be built here"}


if __name__ == "__main__":
    main()
