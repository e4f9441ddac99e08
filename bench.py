#!/usr/bin/env python
"""bench.py -- the driver's benchmark contract for the HNSW distance-evaluation hot path.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json `metric`): 1M x 768 f32, L2sq, M=16 ef_construction=128, ef=64, k=10.
A "step" is one pass of the hot path (usearch_search_ef semantics, lantern_hnsw/src/hnsw/scan.c:220)
over one batch of synthetic queries that is already resident in HBM.  Queries shard across GPUs
with no collective (the index is replicated in each GPU's HBM), so scaling is weak: every rank
searches its own `--queries` per step and `value` = all ranks' queries / max-over-ranks time.

Multi-GPU: `python bench.py --gpus N` with no WORLD_SIZE in the environment re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` (what the driver does itself for N > 1); a launch whose world size
differs from --gpus is refused.  torch.distributed.run only LAUNCHES the ranks: the measuring processes never import
PyTorch (it bundles a second HIP runtime).  The ranks rendezvous through a directory (lantern_amd/rendezvous.py), build
ONE index together (lantern_gpu_add_sharded: RCCL all-gathers over xGMI of the top-M neighbour lists; every rank ends
with a bit-identical replica), and then each searches its own queries on its replica.

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
# This process is the launcher of the scan-service leg (bench_secondary.headline_scan_service: four dispatcher lanes, a stream each):
# give the HIP runtime a hardware queue per lane before it initialises (lantern_amd/csrc/index.cpp)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_CEILING_GBS = 6290.0  # same guide: "6.29 TB/s measured (float4 copy, 79 %)" -- what a pure DRAM stream reaches


_T0 = time.time()


def log(msg):
    """Progress on stderr (the JSON line is the only thing on stdout): which leg is running, seconds since start."""
    print(f"[bench {time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


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
    p.add_argument("--streams", type=int, default=1, help="search launches in flight: steps alternate over this many streams, each with its own query batch "
                   "(independent batches of a serving workload; the second fills the machine while the first one's longest walks drain)")
    p.add_argument("--query-batches", type=int, default=4, help="distinct resident query batches the steps rotate through (step i searches batch i mod this): "
                   "no step replays the queries of the step before it, so no launch finds its own rows in L2 / Infinity Cache")
    p.add_argument("--waves", type=int, default=0, help="wavefronts per query (0 = the library's automatic shape)")
    p.add_argument("--max-wg", type=int, default=0)
    p.add_argument("--add-batch", type=int, default=32768, help="largest insertion batch (never more than a sixteenth of the graph): 32768 since round 6 -- 32 768-row "
                   "batches are pinned edge for edge against the oracle (tests/test_gpu_build_parity_production_batch.py: plan32768 case) and the plan (32768, 16) is "
                   "within 0.005 of the sequential reference build's recall at 600k rows (tests/test_gpu_baseline_configs.py, slow: 0.2546 vs 0.2571); "
                   "16384 was the plan of round 5, 8192 of rounds 1 - 4")
    p.add_argument("--truth-queries", type=int, default=1024, help="queries used for recall@k")
    p.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget (0 = skip)")
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--quant", default="f32", choices=["f32", "f16", "i8", "b1"], help="storage kind (reloption quant_bits 32 / 16 / 8 / 1); the headline config is f32")
    p.add_argument("--pq-subvectors", type=int, default=0, help="pq = true with this many subvectors (0 = off): the index is built with the decodings "
                   "resident, then COMPACTED (codes only in HBM) and searched by ADC over the code bytes (lantern_gpu_pq_compact)")
    p.add_argument("--pq-centroids", type=int, default=256)
    p.add_argument("--dist-backend", default="rccl", choices=["rccl", "nccl", "files", "gloo"],
                   help="exchange transport of the collective build at N>1: rccl (alias nccl; xGMI, data stays in HBM) or files (alias "
                        "gloo: the host transport over the rendezvous directory -- debugging, or several ranks on one GPU)")
    p.add_argument("--dry-run", action="store_true", help="launch, rendezvous, barrier and print the line's shape without touching a device (CPU test of the N>1 plumbing)")
    p.add_argument("--build-quality-rows", type=int, default=100_000, help="rows of the build-quality leg (device batched build vs sequential CPU build; 0 = skip)")
    p.add_argument("--data", default="gaussian", choices=["gaussian", "lowrank", "clustered"],
                   help="gaussian = the prescribed i.i.d. N(0,1) set (SURVEY 8d); lowrank = 32 latent dims embedded in --dim; clustered = Gaussian mixture with "
                        "low-dimensional clusters (lantern_amd/synth.py): the set on which HNSW reaches the recall the reference asserts (>= 0.9)")
    p.add_argument("--data-scale", type=float, default=1.0, help="multiply the synthetic rows and queries (i8 storage quantises [-1, 1]: use 0.3)")
    p.add_argument("--collective-timeout", type=float, default=180.0, help="deadline of every exchange of the collective build")
    p.add_argument("--build", choices=["work-sharded", "row-sharded"], default="work-sharded",
                   help="the collective build of a multi-rank job: lantern_gpu_add_sharded (default; one graph, bit-identical replicas, "
                        "DESIGN.md 6) or lantern_gpu_add_row_sharded (the survey's 8e partitioning; recall-equivalent, DESIGN.md 4.6d)")
    p.add_argument("--no-pmc", action="store_true", help="skip the counter passes (roofline.traffic then comes from the committed profiles/pmc_traffic.json, if it has this "
                   "configuration): by default, when rocprofv3 is on PATH, the search leg is re-executed under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` "
                   "(two short passes restricted to k_search) after the timed region and roofline.traffic is THIS run's")
    p.add_argument("--pmc-steps", type=int, default=4, help="search launches per counter pass")
    p.add_argument("--no-two-in-flight", action="store_true", help="skip two_launches_in_flight (the same steps alternating over two streams)")
    p.add_argument("--no-gather-ceiling", action="store_true", help="skip roofline.gather (the random-row fetch rate of this box, three launches of k_gather_walkshape)")
    p.add_argument("--no-dram-model", action="store_true", help="skip roofline.dram_bytes_model (two traced launches + the LRU replay of their traces)")
    p.add_argument("--no-secondary", action="store_true", help="skip the `secondary` array (bench_secondary.py: BASELINE configs [1] and [2], the clustered set, the "
                   "host-buffer and scan-service paths at the headline shape)")
    p.add_argument("--base-seed", type=int, default=3, help="numpy default_rng seed of the indexed rows (SURVEY 8d: C3 = 3, C2 = 1)")
    p.add_argument("--query-seed", type=int, default=4, help="seed of the queries (C3 = 4, C2 = 2)")
    p.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the re-executed search leg: build, launch, print the graph checksum, exit
    p.add_argument("--dense-child", action="store_true", help=argparse.SUPPRESS)  # the re-executed exact k-NN of bench_secondary.c3_dense_exact_knn (counter pass)
    return p.parse_args()


def respawn_under_torchrun(a):
    """`python bench.py --gpus N` typed by hand: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    (the command the driver issues itself), so the plain command measures N GPUs instead of silently measuring one."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.execvpe(cmd[0], cmd, env)


def bring_up_comm(a, rdv, rank, world, capi, device_ids):
    """The exchange transport of the collective build.  RCCL first (data stays in HBM, xGMI); every rank reports whether
    its communicator came up AND passed one small all-gather, and only if ALL did is it used -- otherwise every rank
    switches to the host transport over the rendezvous directory (slower, but a broken fabric never costs the line).
    Returns (comm, transport, note, info): `info` says how many ranks RCCL saw, which device every rank sits on and -- when the
    host transport was taken -- why.  RCCL is not even attempted when two ranks share a device (it refuses duplicate devices
    with ncclInvalidUsage, after a rendezvous that can hang): that launch is a debugging configuration, and says so."""
    import threading

    note = None
    want_rccl = a.dist_backend in ("rccl", "nccl")
    per_device = {}
    for r, d in enumerate(device_ids):
        per_device.setdefault(d, []).append(r)
    shared = {d: rs for d, rs in per_device.items() if len(rs) > 1}
    info = {"requested_backend": a.dist_backend, "devices_by_rank": device_ids, "distinct_devices": len(per_device), "rccl_ranks_seen": 0}
    # (LANTERN_BENCH_RCCL_ON_SHARED_DEVICE=1: only with the test double of tests/fake_rccl/ bound through LANTERN_GPU_RCCL_LIB, which -- unlike
    # RCCL -- takes several ranks on one device: the bring-up below then runs end to end on a one-GPU box, tests/test_gpu_fake_rccl.py)
    if want_rccl and shared and os.environ.get("LANTERN_BENCH_RCCL_ON_SHARED_DEVICE", "0") in ("", "0"):
        note = ("RCCL not attempted: " + "; ".join(f"ranks {rs} share device {d}" for d, rs in shared.items())
                + " -- RCCL refuses two ranks on one device; launch one rank per GPU (or pass --dist-backend files for a one-GPU rehearsal)")
        want_rccl = False
    if want_rccl:
        box = {}

        def attempt():
            try:
                uid = rdv.broadcast(capi.Comm.unique_id() if rank == 0 else None, 0, timeout=120.0)
                c = capi.Comm.rccl(rank, world, uid)
                c.set_timeout(60.0)
                probe = np.zeros(8 * world, dtype=np.uint8)
                probe[8 * rank: 8 * rank + 8] = rank + 1
                c.allgatherv_host(probe, [8 * r for r in range(world)], [8] * world)
                assert all(probe[8 * r] == r + 1 for r in range(world)), "RCCL all-gather returned wrong data"
                box["comm"] = c
            except Exception as e:  # noqa: BLE001
                box["err"] = repr(e)

        t = threading.Thread(target=attempt, daemon=True)  # ncclCommInitRank has no deadline of its own
        t.start()
        t.join(150.0)
        ok = "comm" in box
        verdicts = rdv.allgather((b"1" if ok else b"0") + (box.get("err", "timed out") if not ok else "").encode()[:300], timeout=300.0)
        info["rccl_ranks_seen"] = sum(1 for v in verdicts if v[:1] == b"1")
        if all(v[:1] == b"1" for v in verdicts):
            box["comm"].set_timeout(a.collective_timeout)
            info["transport_used"] = "rccl"
            return box["comm"], "RCCL all-gather-v (grouped ncclBroadcast) on the index stream", None, info
        note = "RCCL unavailable (" + "; ".join(f"rank {r}: {v[1:].decode(errors='replace')}" for r, v in enumerate(verdicts) if v[:1] != b"1") + ")"
    c = capi.Comm.host(rank, world, rdv.allgatherv)
    c.set_timeout(a.collective_timeout)
    info["transport_used"] = "files"
    info["fallback_reason"] = note if a.dist_backend in ("rccl", "nccl") else None
    return c, "host transport over the rendezvous directory (D2H / files / H2D)", note, info


def main():
    a = parse()
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if a.gpus > 1 and env_world == 0:
        respawn_under_torchrun(a)  # does not return
    world = env_world or 1
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {a.gpus}: the line would claim the wrong GPU count; "
                 f"pass --gpus {world} (or launch {a.gpus} ranks)")
    rdv = None
    if world > 1:
        from lantern_amd import rendezvous

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # all ranks are on this node; the container's hostname may not resolve
        rdv = rendezvous.FileRendezvous(rank, world)
        rdv.barrier()

    def finish(line):
        if rank == 0:
            print(json.dumps(line), flush=True)
        if rdv:
            rdv.finalize()

    if a.dry_run:  # the N>1 plumbing without a device: launch, world check, rendezvous, per-rank shard synthesis, reductions, line shape
        from lantern_amd import capi as capi_dry, synth as synth_dry

        lo, hi = capi_dry.shard_range(a.n, world, rank)
        shard = synth_dry.shard_rows(a.data, a.n, a.dim, lo, hi) if world > 1 else synth_dry.base_rows(a.data, a.n, a.dim)
        mine = json.dumps({"rank": rank, "local_rank": local_rank, "rows": [lo, hi], "host_bytes": int(shard.nbytes),
                           "device_bytes_replica": int(a.n) * int(a.dim) * 4 + int(a.n) * (2 * a.M * 4 + 16),
                           "checksum": float(np.float64(shard[:1].sum()))}).encode()
        shards = [json.loads(x) for x in rdv.allgather(mine)] if rdv else [json.loads(mine)]
        elapsed = rdv.max_float(0.001 * (rank + 1)) if rdv else 0.001
        ranks = [int(x) for x in rdv.allgather(str(rank).encode())] if rdv else [0]
        return finish({"dry_run_shards": shards, "rows_covered": sum(s_["rows"][1] - s_["rows"][0] for s_ in shards),"metric": f"QPS (recall@{a.k} alongside), {a.n}x{a.dim} f32 {a.metric} ef={a.ef} k={a.k}", "value": None, "unit": "queries/s",
                       "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed * 1e3, "higher_is_better": True,
                       "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True, "ranks_seen": ranks,
                       "config": {"workload": f"HNSW search {a.n}x{a.dim} {a.quant} {a.metric} M={a.M} ef_construction={a.efc} ef={a.ef} k={a.k}",
                                  "queries_per_step_per_gpu": a.queries, "global_queries_per_step": a.queries * world,
                                  "parallelism": f"replicated index, query batch sharded x{world}, no collective"}})

    from lantern_amd import capi, hip

    assert capi.device_count() > 0, "no HIP device: bench.py measures the HIP path only"
    if a.dense_child:
        import bench_secondary

        return bench_secondary.dense_child(a, capi, hip)
    # one rank per GPU: hipSetDevice(local_rank).  More ranks than devices (a one-GPU rehearsal of the N-rank job) wrap around -- the
    # ranks then share a device, RCCL is not attempted (bring_up_comm) and the line says so.
    dev_index = (local_rank % capi.device_count()) if world > 1 else 0
    hip.set_device(dev_index)
    device_ids = [x.decode() for x in rdv.allgather(hip.device_bus_id(dev_index).encode())] if rdv else [hip.device_bus_id(dev_index)]

    # ---- synthetic data (numpy default_rng, standard normal f32, seeds 3 / 4) -------
    from lantern_amd import synth

    t0 = time.time()
    make_queries = synth.query_maker(a.data, a.dim)
    if world > 1:  # every rank synthesises (and later uploads) only the shard it contributes to the collective build
        shard_lo, shard_hi = capi.shard_range(a.n, world, rank)
        base = synth.shard_rows(a.data, a.n, a.dim, shard_lo, shard_hi)
    else:
        base = synth.base_rows(a.data, a.n, a.dim, a.base_seed)
    if a.data_scale != 1.0:
        raw_queries = make_queries
        make_queries = lambda r, n: raw_queries(r, n) * np.float32(a.data_scale)
        base *= np.float32(a.data_scale)
    labels = np.arange(a.n, dtype=np.uint64) + 1  # 0 is INVALID_ELEMENT_LABEL (hnsw.h:40)
    t_gen = time.time() - t0

    # ---- the index.  One GPU: built here.  N GPUs: ONE collective build -- rank r contributes shard r of the rows, the
    # work of every insertion batch is split over the ranks, the top-M neighbour lists and the re-written adjacency rows
    # are all-gathered -- which leaves a bit-identical replica in every GPU's HBM: exactly what the
    # query-sharded search leg needs.
    pq_kw = {}
    if a.pq_subvectors:
        # a codebook in the layout Lantern hands to usearch_init (pqtable.c:194-240): [C][dim], centroid c of every subvector
        # concatenated; the centroids are sampled rows per subvector (what k-means++ starts from) -- the bench measures the
        # search over code bytes, not codebook quality
        assert world == 1 and a.quant == "f32" and a.dim % a.pq_subvectors == 0
        crng = np.random.default_rng(91)
        sub = a.dim // a.pq_subvectors
        cb = np.zeros((a.pq_centroids, a.dim), dtype=np.float32)
        for sv in range(a.pq_subvectors):
            cb[:, sv * sub:(sv + 1) * sub] = base[crng.choice(a.n, size=a.pq_centroids, replace=False), sv * sub:(sv + 1) * sub]
        pq_kw = {"pq_codebook": cb, "num_subvectors": a.pq_subvectors}
    ix = capi.GpuIndex(a.metric, a.dim, M=a.M, ef_construction=a.efc, ef=a.ef, seed=42, quantization=a.quant, **pq_kw)
    ix.reserve(a.n)
    ix.set_add_batch(a.add_batch, 16)
    ix.set_search_shape(a.waves, a.max_wg)
    ix.set_profiling(True)
    collective = None
    if world > 1:
        comm, transport, note, comm_info = bring_up_comm(a, rdv, rank, world, capi, device_ids)
        lo, hi = capi.shard_range(a.n, world, rank)
        rdv.barrier()
        hip.synchronize()
        t0 = time.time()
        err = None
        try:
            if a.build == "row-sharded":
                ix.add_row_sharded(comm, labels[lo:hi], base)
            else:
                ix.add_sharded(comm, labels[lo:hi], base)  # (this rank's shard: generated above, nothing else is on this host)
            ix.flush()
            hip.synchronize()
        except Exception as e:  # noqa: BLE001 -- every rank fails together: a collective that misses its deadline fails on all
            err = repr(e)
        t_build = time.time() - t0
        errs = [x.decode(errors="replace") for x in rdv.allgather((err or "").encode()[:400], timeout=a.collective_timeout + 60)]
        if any(errs):
            # the collective build failed: fall back to N independent builds of the same (deterministic) index so that the
            # search measurement still stands; the failure is reported in the line
            ix = capi.GpuIndex(a.metric, a.dim, M=a.M, ef_construction=a.efc, ef=a.ef, seed=42, quantization=a.quant)
            ix.reserve(a.n)
            ix.set_add_batch(a.add_batch, 16)
            ix.set_search_shape(a.waves, a.max_wg)
            ix.set_profiling(True)
            hip.synchronize()
            t0 = time.time()
            for r in range(world):  # the same rows the collective build would have seen: every rank's shard, in rank order
                r_lo, r_hi = capi.shard_range(a.n, world, r)
                part = base if r == rank else synth.shard_rows(a.data, a.n, a.dim, r_lo, r_hi) * np.float32(a.data_scale)
                ix.add_many(labels[r_lo:r_hi], part)
            ix.flush()
            hip.synchronize()
            t_build = time.time() - t0
            collective = dict({"error": next(e for e in errs if e), "fallback": "every rank built its own replica", "transport": transport, "transport_note": note},
                              **comm_info)
        else:
            sums = [x.decode() for x in rdv.allgather(f"{ix.checksum():016x}".encode())]
            stats = comm.stats()
            collective = {**comm_info, "world": world, "build": a.build, "seconds": rdv.max_float(t_build), "transport": transport, "transport_note": note,
                          "replicas_identical": len(set(sums)) == 1, "checksum": sums[0],
                          "bytes_received_per_rank": [int(x) for x in rdv.allgather(str(stats["bytes_received"]).encode())],
                          "collectives": stats["collectives"]}
            collective["vectors_per_s"] = a.n / collective["seconds"]
    else:
        hip.synchronize()
        t0 = time.time()
        ix.add_many(labels, base)
        ix.flush()
        hip.synchronize()
        t_build = time.time() - t0
    log(f"index built: {a.n} rows in {t_build:.2f}s")
    build_counters = ix.counters()
    build_profile = ix.build_profile()
    pq_info = None
    pq_truth = None
    if a.pq_subvectors:
        # recall truth = exact k-NN over the DECODED rows (what a PQ index's distances are distances to), taken before the rows go
        mem_before = ix.memory_usage()
        tq0 = make_queries(np.random.default_rng(a.query_seed + 1000 * rank), a.queries * max(a.streams, a.query_batches, 1))[:min(a.truth_queries, a.queries)]  # = queries[:tq] below
        pq_truth, _ = ix.exact_search(tq0, a.k)
        ix.pq_compact()
        mem_after = ix.memory_usage()
        on_the_fly = (a.dim // a.pq_subvectors) % 4 == 0 and os.environ.get("LANTERN_GPU_PQ_ADC", "0") in ("", "0")
        pq_info = {"num_subvectors": a.pq_subvectors, "num_centroids": a.pq_centroids, "row_bytes_decoded_form": mem_before[0], "row_bytes_compact_form": mem_after[0],
                   "other_index_bytes": mem_after[1],
                   "search": ("rows decoded on the fly from the L2-resident centroid tables: the expanded index's arithmetic, bit for bit (device_common.hpp PqdRow)" if on_the_fly else
                              "ADC over the code bytes: per-query table (subvector x centroid) in LDS, lantern_amd/csrc/search_adc_kernel.hip"),
                   "recall_truth": "exact k-NN over the decoded rows"}

    # ---- this rank's queries, resident in HBM ----------------------------------------------------
    # B distinct batches (>= 4 by default), step i searches batch i mod B on stream i mod S: consecutive steps never replay
    # the same queries, so a launch does not find the rows of its own previous run in L2 / the 256 MiB Infinity Cache.
    qrng = np.random.default_rng(a.query_seed + 1000 * rank)
    nq = a.queries
    S = max(1, a.streams)
    B = max(S, a.query_batches, 1)
    all_queries = make_queries(qrng, nq * B)
    queries = all_queries[:nq]  # the batch recall and the CPU baseline are taken on
    streams = [hip.Stream() for _ in range(S)]
    lanes = []  # one per query batch: its rows and outputs
    for i in range(B):
        qi = all_queries[i * nq:(i + 1) * nq]
        rows_i = ix.device_query_rows(qi)  # the index's own storage format and row stride
        q_stride = rows_i.strides[0]
        lanes.append({"dq": hip.Buffer.from_numpy(rows_i),
                      "lab": hip.Buffer(nq * a.k * 8), "dist": hip.Buffer(nq * a.k * 4), "slot": hip.Buffer(nq * a.k * 4),
                      "D": hip.Buffer(nq * 8), "E": hip.Buffer(nq * 8)})
    d_slot = lanes[0]["slot"]

    def step(i=0):
        L = lanes[i % B]
        ix.search_batch_device(L["dq"].ptr, nq, a.k, a.ef, 0, L["lab"].ptr, L["dist"].ptr, L["slot"].ptr, None, L["D"].ptr, L["E"].ptr, streams[i % S].handle,
                               query_stride=q_stride)

    def barrier():
        hip.synchronize()
        if rdv:
            rdv.barrier()
        hip.synchronize()

    for i in range(max(a.warmup, 0)):
        step(i)
    if a.warmup < B:  # every batch's D / E counters are read below: each batch runs at least once (untimed)
        for i in range(a.warmup, B):
            step(i)
    barrier()
    ev = [(hip.Event(), hip.Event()) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for i, (s, e) in enumerate(ev):
        st = streams[i % S].handle
        s.record(st)
        step(i)
        e.record(st)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [s.elapsed_ms(e) for s, e in ev]  # HIP events on the launch stream: one search launch each
    log(f"timed region done: {a.steps} steps in {elapsed:.3f}s")
    if rdv:
        elapsed = rdv.max_float(elapsed)
    # ---- the same steps with TWO launches in flight (independent batches on two streams: the tail of one launch, where the last walks
    # finish on a half-empty device, is filled by the next one).  Reported beside `value`, never as `value`: the contract's roofline is
    # defined per launch, and launches that share the device stretch each other's durations.
    two_in_flight = None
    if world == 1 and S == 1 and B >= 2 and not a.pmc_child and not a.no_two_in_flight:
        st2 = [streams[0], hip.Stream()]

        def step2(i):
            L = lanes[i % B]
            ix.search_batch_device(L["dq"].ptr, nq, a.k, a.ef, 0, L["lab"].ptr, L["dist"].ptr, L["slot"].ptr, None, L["D"].ptr, L["E"].ptr, st2[i % 2].handle,
                                   query_stride=q_stride)
        for i in range(2):  # (the second launch slot's visited bitmaps are allocated on first use)
            step2(i)
        barrier()
        t2 = time.perf_counter()
        for i in range(a.steps):
            step2(i)
        barrier()
        el2 = time.perf_counter() - t2
        two_in_flight = {"value": nq * a.steps / el2, "unit": "queries/s", "ms_per_step": el2 / a.steps * 1e3, "over_one_in_flight": elapsed / el2,
                         "note": "the same K steps alternating over two streams (two launches in flight, independent batches); per-launch durations overlap, so no per-launch roofline"}
    if a.pmc_child:  # a counter pass of measure_traffic(): the launches above are what rocprofv3 counted
        print(json.dumps({"pmc_child": True, "checksum": f"{ix.checksum():016x}", "launches": max(a.warmup, B) + a.steps, "queries_per_launch": nq}), flush=True)
        return

    # ---- algorithmic bytes of one launch --------------------------------------------
    row_bytes = a.dim * {"f32": 4, "f16": 2, "i8": 1, "b1": 0.125}[a.quant]
    if a.pq_subvectors:
        row_bytes = ix.memory_usage()[0] // max(len(ix), 1)  # a compact pq index evaluates a row from its code bytes (their stride in HBM)
    per_lane = []
    for L in lanes:
        Dl = L["D"].download(nq, np.uint64).astype(np.float64)
        El = L["E"].download(nq, np.uint64).astype(np.float64)
        per_lane.append((Dl, El, float((Dl * row_bytes + El * (2 * a.M * 4) + row_bytes).sum())))
    D, E = per_lane[0][0], per_lane[0][1]
    # the timed steps' own launches: step i ran batch i mod B
    bytes_per_launch = float(np.mean([per_lane[i % B][2] for i in range(a.steps)]))
    avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
    # one launch at a time: bytes of a launch / its HIP-event duration.  Launches in flight side by side (--streams > 1) stretch
    # each other's durations, so there the rate is all launches' bytes / the timed region
    achieved = bytes_per_launch / avg_kernel_s / 1e9 if S == 1 else bytes_per_launch * a.steps / elapsed / 1e9

    out = None
    if rank == 0:
        # ---- recall@k against exact f32 k-NN (fp32-MFMA contraction + exact re-rank) --------------
        tq = min(a.truth_queries, nq)
        t0 = time.time()
        truth = pq_truth[:tq] if pq_truth is not None else ix.exact_search(queries[:tq], a.k)[0]
        t_truth = time.time() - t0
        found = d_slot.download((nq, a.k), np.uint32)[:tq]
        recall = float(np.mean([len(set(f.tolist()) & set(t.tolist())) / a.k for f, t in zip(found, truth)]))

        cpu = None
        quality = None
        co_headline = (world == 1 and not a.no_secondary and a.quant == "f32" and not a.pq_subvectors and a.metric == "l2sq" and a.data == "gaussian"
                       and S == 1)  # the clustered set beside the prescribed one (clustered_coheadline)
        qualities = {}
        if world == 1 and not a.no_cpu and a.cpu_seconds > 0 and not a.pq_subvectors:
            log("cpu_baseline ...")
            cpu = cpu_baseline(a, ix, base, queries, found)
            log("build_quality ...")
            if a.build_quality_rows > 0 and a.quant == "f32" and a.metric != "hamming":
                qualities = build_quality(a, [a.data] + (["clustered"] if co_headline else []))
                quality = qualities.get(a.data)

        # ---- HBM-side bytes per launch: counter passes of THIS run (measure_traffic), else the committed passes of this command line
        traffic = traffic_src = None
        measured_here = False
        pmc_detail = None
        key = (f"{a.n}x{a.dim}_{a.metric}_ef{a.ef}_q{nq}_w{a.waves or 4}" + ("" if a.quant == "f32" else "_" + a.quant)
               + ("" if a.data == "gaussian" else "_" + a.data) + (f"_b{B}" if B > 1 else "") + (f"_pq{a.pq_subvectors}" if a.pq_subvectors else ""))
        if world == 1 and not a.no_pmc and S == 1:
            log("counter passes (headline) ...")
            pmc_detail = measure_traffic(a, f"{ix.checksum():016x}")
            log(f"counter passes done: {[p_.get('error', 'ok') for p_ in (pmc_detail or {}).get('passes', [])]}")
            if pmc_detail and pmc_detail.get("hbm_bytes_per_launch"):
                traffic, measured_here = pmc_detail["hbm_bytes_per_launch"], True
                traffic_src = pmc_detail["source"]
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if traffic is None and os.path.exists(prof):
            try:
                hit = json.load(open(prof)).get(key, {})
                traffic = hit.get("hbm_bytes_per_launch")
                traffic_src = hit.get("source")
            except Exception:
                traffic = None
        # ---- distinct rows a launch evaluates (device bitmap, instrumented walk): the cold-miss lower bound of its DRAM bytes
        unique = unique_rows_per_launch(a, ix, step, B, hip) if world == 1 and not a.pq_subvectors else None
        # ---- what of those bytes reaches DRAM: the cache model over the launch's own trace (f32 l2sq / cos, one launch in flight)
        model = None
        if world == 1 and not a.pq_subvectors and a.quant == "f32" and a.metric in ("l2sq", "cos") and S == 1 and not a.no_dram_model:
            log("dram model (traced launches + replay) ...")
            model = dram_model(ix, hip, lanes, nq, a.k, a.ef, q_stride, row_bytes, 2 * a.M * 4, avg_kernel_s, traffic,
                               (pmc_detail or {}).get("read_bytes_per_launch") if measured_here else None)
            log(f"dram model done: {model.get('error') or model.get('seconds')}")
        build_traffic = None
        if world == 1 and not a.no_pmc and not a.pq_subvectors:
            log("counter pass (build kernels) ...")
            build_traffic = measure_build_traffic(a, f"{ix.checksum():016x}")
        gather = None
        if world == 1 and not a.pq_subvectors and not a.no_gather_ceiling:
            log("gather ceiling of this box ...")
            gather = gather_ceiling(ix, a.n, int(row_bytes), int(min(max(float(D.mean()) * nq, 2e6), 2e7)))
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
            "dtype": {"f32": "f32", "f16": "f32 arithmetic on f16 storage", "i8": "int32 arithmetic on i8 storage", "b1": "popcount on 1-bit storage"}[a.quant],
            "data": {"gaussian": "synthetic", "lowrank": "synthetic (low-rank)", "clustered": "synthetic (clustered: " + synth.CLUSTERED_DOC + ")"}[a.data]
                    + ("" if a.data_scale == 1.0 else f" x {a.data_scale}"),
            "config": {"workload": f"HNSW search {a.n}x{a.dim} {a.quant} {a.metric} M={a.M} ef_construction={a.efc} ef={a.ef} k={a.k}",
                       "queries_per_step_per_gpu": nq, "global_queries_per_step": nq * world, "waves_per_query": a.waves,
                       "launches_in_flight": S,
                       "parallelism": f"replicated index, query batch sharded x{world}, no collective"},
            f"recall_at_{a.k}": recall,
            "recall_queries": tq,
            "build_vectors_per_s": a.n / t_build,
            "build_seconds": t_build,
            "build_batches": build_counters["add_batches"],
            "build_counters_per_vector": {k: build_counters[k] / a.n for k in ("add_walk_evals", "add_select_evals", "add_revlink_evals",
                                                                                 "add_reprunes", "add_expansions")},
            "build_roofline": build_roofline(a, build_counters, build_profile, t_build, world, build_traffic),
            "dist_evals_per_query": float(D.mean()),
            "expansions_per_query": float(E.mean()),
            "roofline": roofline(achieved, traffic, traffic_src, avg_kernel_s if S == 1 else elapsed / a.steps, bytes_per_launch, avg_kernel_s, S, B,
                                 adc=bool(a.pq_subvectors), measured_here=measured_here, pmc_detail=pmc_detail, unique=unique, row_bytes=row_bytes,
                                 list_bytes=2 * a.M * 4, expansions_per_launch=float(np.mean([per_lane[i % B][1].sum() for i in range(a.steps)])), model=model,
                                 gather=gather),
            "cpu_baseline": cpu,
            "pq": pq_info,
            "two_launches_in_flight": two_in_flight,
            "build_quality": quality,
            "collective_build": collective,
            "setup_seconds": {"datagen": t_gen, "build": t_build, "exact_truth": t_truth},
        }
        if co_headline:
            # ---- the co-headline: the same shape, index parameters and procedure on the CLUSTERED set -- the set on which recall (>= the
            # reference's 0.7 / 0.9 floors, scripts/integration_tests.py:249-257) and the roofline fractions (< 1) both mean something
            t0 = time.time()
            log("clustered co-headline ...")
            try:
                out["clustered"] = clustered_coheadline(a, capi, hip, qualities.get("clustered"))
            except Exception as ex:  # noqa: BLE001 -- the co-headline never costs the line
                out["clustered"] = {"error": repr(ex)[:400]}
            out["setup_seconds"]["clustered"] = time.time() - t0
            c = out["clustered"]
            out["sets"] = {"gaussian": {"data": "i.i.d. N(0,1) (SURVEY 8d's prescribed set; no neighbourhood structure in 768 dimensions)", "value": qps,
                                        f"recall_at_{a.k}": recall, "frac_algorithmic": out["roofline"]["frac_algorithmic"], "frac_fabric": out["roofline"]["frac_fabric"],
                                        "frac_dram_model": out["roofline"]["frac_dram_model"], "cpu_all_cores": cpu["value"] if cpu else None,
                                        "fabric_over_infinity_cache_gather": out["roofline"].get("fabric_over_infinity_cache_gather"),
                                        "dram_model_over_gather_dram_ceiling": out["roofline"].get("dram_model_over_gather_dram_ceiling")},
                           "clustered": {"data": "clustered: " + synth.CLUSTERED_DOC, "value": c.get("value"), f"recall_at_{a.k}": c.get(f"recall_at_{a.k}"),
                                         "frac_algorithmic": (c.get("roofline") or {}).get("frac_algorithmic"), "frac_fabric": (c.get("roofline") or {}).get("frac_fabric"),
                                         "frac_dram_model": (c.get("roofline") or {}).get("frac_dram_model"),
                                         "cpu_all_cores": (c.get("cpu_baseline") or {}).get("value"),
                                         "fabric_over_infinity_cache_gather": (c.get("roofline") or {}).get("fabric_over_infinity_cache_gather"),
                                         "dram_model_over_gather_dram_ceiling": (c.get("roofline") or {}).get("dram_model_over_gather_dram_ceiling")}}
        if world == 1 and not a.no_secondary and a.quant == "f32" and not a.pq_subvectors and a.metric != "hamming":
            import bench_secondary

            if gather and gather.get("algorithmic_gbs"):
                bench_secondary.GATHER_GBS = gather["algorithmic_gbs"]  # the secondary legs' frac_of_gather_ceiling: this box's, this run's

            t0 = time.time()
            log("secondary legs ...")
            out["secondary"] = bench_secondary.run(a, capi, hip, ix, base, all_queries, qps, recall, None if a.no_pmc else measure_traffic)
            out["setup_seconds"]["secondary"] = time.time() - t0
    finish(out)


# profiles/r06_cache_model_calibration.md: 17.56 M uniformly random 3 KiB rows in the walk's own launch shape (k_gather_walkshape) take 9.56 ms:
# 5.64 TB/s in algorithmic bytes, 5.78 TB/s at the fabric (counters), 5.15 TB/s from DRAM by the cache model.  (Rounds 3 - 5 quoted 6.73 TB/s:
# an average over six launches one of which was the measuring script's 1000-row self-check -- the calibration file has the correction.)
GATHER_CEILING_GBS = 5640.0       # fallback only (--no-gather-ceiling): profiles/r06_cache_model_calibration.md, another box, before the blocked row loads;
GATHER_DRAM_CEILING_GBS = 5150.0  # a run measures its own box's figure (gather_ceiling(), roofline.gather)


def measure_traffic(a, checksum, counters=("FETCH_SIZE", "WRITE_SIZE"), kernel="k_search"):
    """roofline.traffic of THIS run: the search leg re-executed under rocprofv3, one pass per counter (FETCH_SIZE takes three of
    the four TCC slots, WRITE_SIZE two: MI355X_MICROARCH.md "rocprofv3 PMC slots"; gpurun refuses counter passes combined with
    trace domains), restricted to the search kernel (bench_pmc.run_pass).  The child builds the same index from the same seeds
    (checked: graph checksum), runs max(warmup, batches) + --pmc-steps launches over the same rotating query batches and exits; the
    mean counter value per launch is converted as the guide's HBM section prescribes: FETCH_SIZE is KiB and on gfx950 reports
    exactly 1/2 of the bytes of wide coalesced reads -> x 1024 x 2; WRITE_SIZE KiB x 1024 (uncalibrated there; < 0.01 % here, so the
    secondary legs pass counters=("FETCH_SIZE",) and say so).
    Returns None when rocprofv3 is not on PATH (the committed profiles/pmc_traffic.json is used then)."""
    import bench_pmc

    if not bench_pmc.rocprof():
        return None
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--no-pmc", "--no-cpu", "--truth-queries", "0", "--build-quality-rows", "0",
             "--steps", str(max(1, a.pmc_steps)), "--warmup", "1", "--rows", str(a.n), "--dim", str(a.dim), "--metric", a.metric, "--M", str(a.M),
             "--efc", str(a.efc), "--ef", str(a.ef), "--k", str(a.k), "--queries", str(a.queries), "--query-batches", str(a.query_batches),
             "--waves", str(a.waves), "--max-wg", str(a.max_wg), "--add-batch", str(a.add_batch), "--quant", a.quant, "--data", a.data,
             "--data-scale", str(a.data_scale), "--pq-subvectors", str(a.pq_subvectors), "--pq-centroids", str(a.pq_centroids),
             "--base-seed", str(a.base_seed), "--query-seed", str(a.query_seed)]
    out = {"counters": {}, "passes": []}
    t0 = time.time()
    for ctr in counters:
        r = bench_pmc.run_pass(child, kernel, kernel, [ctr], "pmc_child")
        if "error" in r:
            out["passes"].append({"counter": ctr, "error": r["error"]})
            continue
        if r["child"]["checksum"] != checksum:
            out["passes"].append({"counter": ctr, "error": "the child built a different graph"})
            continue
        vals = r["values"][ctr]
        if not vals:
            out["passes"].append({"counter": ctr, "error": f"no counter rows for {kernel} in the rocprofv3 output"})
            continue
        out["counters"][ctr] = {"launches": len(vals), "mean_per_launch": float(np.mean(vals)), "min": float(np.min(vals)), "max": float(np.max(vals))}
        out["passes"].append({"counter": ctr, "launches": len(vals), "command": f"rocprofv3 --kernel-include-regex {kernel} --pmc {ctr} -- python bench.py --pmc-child ..."})
    out["seconds"] = time.time() - t0
    if "FETCH_SIZE" not in out["counters"]:
        out["hbm_bytes_per_launch"] = None
        return out
    rd = out["counters"]["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
    wr = out["counters"].get("WRITE_SIZE", {}).get("mean_per_launch", 0.0) * 1024
    out.update(read_bytes_per_launch=rd, write_bytes_per_launch=wr if "WRITE_SIZE" in out["counters"] else None, hbm_bytes_per_launch=rd + wr,
               source="this run: rocprofv3 " + " / ".join("--pmc " + c for c in counters) + (", separate passes" if len(counters) > 1 else "")
                      + f" restricted to {kernel}, over a re-execution of the search leg "
                      "(same seeds, same graph checksum, same rotating query batches); FETCH_SIZE KiB x 1024 x 2 (gfx950: the counter reports half the "
                      "bytes of wide coalesced reads)" + (", WRITE_SIZE KiB x 1024" if "WRITE_SIZE" in counters else "; writes not counted (< 0.01 % of the reads in the headline's passes)")
                      + " -- /opt/skills/guides/MI355X_MICROARCH.md, HBM")
    return out


def measure_build_traffic(a, checksum):
    """build_roofline.*.traffic of THIS run: one rocprofv3 --pmc FETCH_SIZE pass over a re-execution of the BUILD (the --pmc-child builds
    the same index from the same seeds: graph checksum checked), restricted to the build's kernels; bytes = KiB x 1024 x 2 as for the
    search kernel (measure_traffic).  -> {short kernel name: {"launches", "fetch_bytes"}} or {"error": ...} / None without rocprofv3."""
    import bench_pmc

    if not bench_pmc.rocprof():
        return None
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--no-pmc", "--no-cpu", "--truth-queries", "0", "--build-quality-rows", "0",
             "--steps", "1", "--warmup", "1", "--rows", str(a.n), "--dim", str(a.dim), "--metric", a.metric, "--M", str(a.M),
             "--efc", str(a.efc), "--ef", str(a.ef), "--k", str(a.k), "--queries", "64", "--query-batches", "1",
             "--add-batch", str(a.add_batch), "--quant", a.quant, "--data", a.data, "--data-scale", str(a.data_scale),
             "--base-seed", str(a.base_seed), "--query-seed", str(a.query_seed)]
    regex = "k_insert|k_connect|k_revlink|k_group|k_link|k_merge"
    r = bench_pmc.run_pass(child, "", regex, ["FETCH_SIZE"], "pmc_child", timeout=420, by_kernel=True)
    if "error" in r:
        return {"error": r["error"]}
    if r["child"]["checksum"] != checksum:
        return {"error": "the child built a different graph"}
    out = {k: {"launches": len(v["FETCH_SIZE"]), "fetch_bytes": float(np.sum(v["FETCH_SIZE"])) * 1024 * 2} for k, v in r["values"].items()}
    out["_source"] = (f"this run: rocprofv3 --kernel-include-regex '{regex}' --pmc FETCH_SIZE over a re-execution of the build (same seeds, same graph checksum); "
                      "FETCH_SIZE KiB x 1024 x 2 (gfx950) -- /opt/skills/guides/MI355X_MICROARCH.md, HBM")
    out["_seconds"] = r["seconds"]
    return out


def dram_model(ix, hip, lanes, nq, k, ef, q_stride, row_bytes, list_bytes, launch_s, traffic, read_traffic=None):
    """roofline.dram_bytes_model: what the 256 MiB Infinity Cache MISSES during one launch, by replaying the launch's own memory-object
    trace through an LRU model of the part's caches (lantern_amd/tools/cache_model.c: eight 4 MiB L2s by XCD in front of the shared
    Infinity Cache; objects = rows and adjacency lists; the walks of the launch advance one hop at a time on as many walkers as the
    launch has workgroups).  The part has no counter that separates DRAM from Infinity-Cache service (profiles/r03_counter_notes.md),
    so this is a MODEL and is labelled so; it is checked on every run by its OTHER output -- the bytes the L2s miss -- against the
    fabric-side counter bytes of the same launch shape (roofline.traffic), and offline on traces with known answers
    (tests/test_cache_model.py).  Two launches over different query batches are traced (the instrumented walk records every row a
    query evaluates and every list it reads, in order: lantern_gpu_search_row_trace; same answers, D and E) and replayed back to back:
    the figures are those of the SECOND launch, which finds the caches as the first left them -- the steady state of the timed loop."""
    import bench_cache_model as cm

    cap = 8192
    t0 = time.time()
    traces, counts = [], []
    try:
        for i in range(2):
            L = lanes[i % len(lanes)]
            ix.row_trace_begin(nq, cap)
            ix.search_batch_device(L["dq"].ptr, nq, k, ef, 0, L["lab"].ptr, L["dist"].ptr, None, None, None, None, None, query_stride=q_stride)
            hip.synchronize()
            walkers = ix.last_search_grid()
            t, c = ix.row_trace_end()
            traces.append(t)
            counts.append(c)
    except Exception as e:  # noqa: BLE001 -- a diagnostic: the line stands without it
        try:
            ix.row_trace_end()
        except Exception:  # noqa: BLE001
            pass
        return {"error": repr(e)[:300]}
    listu = list_bytes // 2
    res = cm.replay(traces, counts, walkers, int(row_bytes), int(list_bytes), int(listu))
    cold, steady = res
    out = {"model": "LRU replay of the launch's own row / adjacency-list trace: 8 x 4 MiB L2 (by XCD) -> 256 MiB Infinity Cache, fully associative, object "
                    "granularity, walks advance one hop per turn on `walkers` workgroups (lantern_amd/tools/cache_model.c; tests/test_cache_model.py)",
           "walkers": walkers, "trace_entries_per_launch": steady["accesses"], "dropped_entries": steady["dropped_entries"],
           "trace_bytes_per_launch": steady["access_bytes"],
           "dram_bytes_model": steady["dram_bytes"], "fabric_bytes_model": steady["fabric_bytes"],
           "dram_bytes_model_cold_caches": cold["dram_bytes"], "fabric_bytes_model_cold_caches": cold["fabric_bytes"],
           "distinct_bytes_per_launch": cm.distinct_bytes(traces[1], counts[1], row_bytes, list_bytes, listu),
           "frac_dram_model": steady["dram_bytes"] / launch_s / 1e9 / HBM_PEAK_GBS,
           "frac_fabric_model": steady["fabric_bytes"] / launch_s / 1e9 / HBM_PEAK_GBS,
           "fabric_model_over_counters": (steady["fabric_bytes"] / traffic) if traffic else None,
           # the model replays READS (rows and lists); the counters' writes are the walk's own stores -- answers, and the clears of a
           # workgroup's HBM visited bitmap when the LDS set spills (10M rows at ef = 128: 7 % of the traffic; 1M rows at ef = 64: 0.01 %)
           "fabric_model_over_counter_reads": (steady["fabric_bytes"] / read_traffic) if read_traffic else None,
           "check": "fabric_model_over_counter_reads near 1 says the model's L2 level reproduces what rocprofv3 counted as fabric READS (FETCH_SIZE) for this "
                    "launch shape; the DRAM level is the same replay one cache further out",
           "seconds": time.time() - t0}
    return out


def unique_rows_per_launch(a, ix, step, B, hip):
    """Distinct rows the queries of ONE launch evaluate, averaged over the B resident batches (lantern_gpu_search_unique_rows:
    the instrumented walk sets one bit per evaluated row in a device bitmap).  None where the instrumented kernel does not
    exist (quantised storage, narrow rows)."""
    if a.quant != "f32" or a.metric not in ("l2sq", "cos"):
        return None
    counts = []
    try:
        for i in range(B):
            ix.unique_rows(True)
            step(i)
            hip.synchronize()
            counts.append(ix.unique_rows(False, read=True))
    except Exception:  # noqa: BLE001 -- a diagnostic: the line stands without it
        try:
            ix.unique_rows(False)
        except Exception:  # noqa: BLE001
            pass
        return None
    return float(np.mean(counts)) if counts else None


def gather_ceiling(ix, rows, row_bytes, evaluations, launches=3):
    """The random-row fetch rate of THIS box, measured in this run: the distance phase of a hop alone, in the walk's launch shape
    (k_gather_walkshape: four-wave workgroups, six per CU, two rows per 64-lane group in flight, the same blocked loads), over
    `evaluations` uniformly random rows of the index (no reuse beyond chance, no list, no visited set, no dependent hops).  Boxes of
    this pool differ by up to 10 % on this figure, so the walk is set against the ceiling of the box it ran on, not a constant.
    DRAM share: a uniformly random gather over T bytes through an LRU hierarchy of C bytes hits C / T of the time (the cache model
    reproduces exactly that on this trace: profiles/r06_cache_model_calibration.md), C = the 256 MiB Infinity Cache."""
    from bench_cache_model import MALL_BYTES
    from lantern_amd import capi

    try:
        rng = np.random.default_rng(99)
        q = rng.standard_normal(ix.dims, dtype=np.float32) if ix.metric != capi.METRIC_HAMMING else rng.integers(0, 2**32, ix.dims, dtype=np.uint32)
        old = os.environ.get("LANTERN_GPU_GATHER_WALKSHAPE")
        os.environ["LANTERN_GPU_GATHER_WALKSHAPE"] = "1"
        ms = []
        try:
            for i in range(launches):
                slots = rng.integers(0, rows, size=evaluations, dtype=np.uint32)
                ix.distance_gather(q, slots)
                ms.append(ix.last_gather_ms())
        finally:
            if old is None:
                os.environ.pop("LANTERN_GPU_GATHER_WALKSHAPE", None)
            else:
                os.environ["LANTERN_GPU_GATHER_WALKSHAPE"] = old
        steady = ms[1:] or ms
        alg = evaluations * row_bytes / (float(np.mean(steady)) * 1e-3) / 1e9
        dram_share = 1.0 - min(1.0, MALL_BYTES / float(rows * row_bytes))
        out = {"kernel": "k_gather_walkshape", "evaluations_per_launch": evaluations, "row_bytes": row_bytes, "launch_ms": ms, "algorithmic_gbs": alg,
               "dram_share": dram_share, "dram_gbs": alg * dram_share, "measured_in_this_run": True}
        # the same gather over a part of the table that FITS the Infinity Cache (0.7 of its 256 MiB; far more than the eight 4 MiB L2s hold):
        # what the fabric between the L2s and the Infinity Cache delivers to this access pattern -- the ceiling of a walk whose rows are
        # shared by the queries of a launch (the Gaussian set: nine tenths of its fabric bytes are Infinity-Cache hits)
        cache_rows = int(0.7 * MALL_BYTES / row_bytes)
        if cache_rows < rows:
            os.environ["LANTERN_GPU_GATHER_WALKSHAPE"] = "1"
            cms = []
            try:
                for i in range(launches):
                    ix.distance_gather(q, rng.integers(0, cache_rows, size=evaluations, dtype=np.uint32))
                    cms.append(ix.last_gather_ms())
            finally:
                if old is None:
                    os.environ.pop("LANTERN_GPU_GATHER_WALKSHAPE", None)
                else:
                    os.environ["LANTERN_GPU_GATHER_WALKSHAPE"] = old
            out["infinity_cache"] = {"rows": cache_rows, "table_bytes": cache_rows * row_bytes, "launch_ms": cms,
                                     "algorithmic_gbs": evaluations * row_bytes / (float(np.mean(cms[1:] or cms)) * 1e-3) / 1e9}
        return out
    except Exception as ex:  # noqa: BLE001 -- the ceiling never costs the line
        return {"error": repr(ex)[:300]}


def roofline(achieved_alg, traffic, traffic_src, launch_s, bytes_per_launch, avg_kernel_s, S, B, adc=False, measured_here=False, pmc_detail=None,
             unique=None, row_bytes=0.0, list_bytes=0.0, expansions_per_launch=0.0, model=None, gather=None):
    """The search kernel against the HBM roofline (8 TB/s spec peak).  FOUR fractions, each named for what it is a fraction OF:

    `frac` = `frac_algorithmic` (the contract's definition): one row per distance evaluation, one adjacency row per
        expansion, D and E counted on the device and equal to the oracle's, / the HIP-event launch time / 8 TB/s.  What the walk ASKS
        the memory system for.  Rows that several queries of a launch evaluate (upper levels, hub rows of un-normalised Gaussian data
        under L2sq: each touched row ~54 times per 8192-query launch) are counted once per evaluation but come from L2 / Infinity
        Cache, so on such a set this figure EXCEEDS 1 and is not a fraction of what HBM delivered.
    `frac_fabric`: the bytes the L2s requested from the fabric during a launch (rocprofv3 FETCH_SIZE / WRITE_SIZE, `traffic`,
        measured in this run) / the same time / 8 TB/s.  Physical, but it still INCLUDES what the 256 MiB Infinity Cache served (the
        part has no counter that separates the two: profiles/r03_counter_notes.md) -- an upper bound of the DRAM rate.
    `frac_dram_model`: Infinity-Cache MISSES of an LRU replay of the launch's own trace (dram_model(); lantern_amd/tools/cache_model.c)
        / time / 8 TB/s.  A model, validated by its fabric-side output against the counters (`dram_model.fabric_model_over_counters`).
    `frac_cold_miss_lower_bound`: distinct rows of a launch x row bytes (+ their adjacency rows) / time / 8 TB/s -- what DRAM must
        deliver even with perfect caches.  The true DRAM fraction lies between this and `frac_fabric`; `frac_dram_model` estimates it.
    Beside the fractions of the spec peak, the two ceilings of the access pattern measured in the same run on the same box (`gather`,
    gather_ceiling()): `dram_model_over_gather_dram_ceiling` (the walk's DRAM rate / a uniformly random gather's over the whole table) and
    `fabric_over_infinity_cache_gather` (the walk's fabric rate / the same gather's over a part of the table that fits the Infinity Cache)."""
    alg_frac = achieved_alg / HBM_PEAK_GBS
    fabric = (traffic / launch_s / 1e9) if traffic else None
    if gather and gather.get("algorithmic_gbs"):
        g_alg, g_dram = gather["algorithmic_gbs"], gather["dram_gbs"]
        g_src = ("measured in THIS run on this box (roofline.gather: uniformly random rows of this index through k_gather_walkshape, the distance phase of a hop "
                 "in the walk's launch shape) -- a walk with reuse may exceed the algorithmic figure, not the DRAM one")
    else:
        g_alg, g_dram = GATHER_CEILING_GBS, GATHER_DRAM_CEILING_GBS
        g_src = ("profiles/r06_cache_model_calibration.md (another box, before the blocked row loads: uniformly random 3 KiB rows in the walk's launch shape: "
                 "5.64 TB/s algorithmic, 5.15 TB/s from DRAM by the cache model)")
    g_mall = ((gather or {}).get("infinity_cache") or {}).get("algorithmic_gbs")
    r = {"bound": "hbm",
         "achieved": achieved_alg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_frac,
         "frac_is": f"frac_algorithmic -- SURVEY 8d algorithmic bytes / HIP-event launch time / 8 TB/s (the bench contract's definition).  Above ~{g_alg / HBM_PEAK_GBS:.2f} "
                    "(the random-row gather ceiling, roofline.gather) it is NOT a physical HBM fraction: rows shared by the queries of a launch are counted once per "
                    "evaluation and served by L2 / Infinity Cache.  The physical figures are frac_fabric (counters; includes Infinity-Cache hits) and "
                    "frac_dram_model (cache-model estimate of DRAM bytes), bounded below by frac_cold_miss_lower_bound",
         "traffic": traffic, "traffic_measured_in_this_run": bool(measured_here), "traffic_source": traffic_src,
         "traffic_over_algorithmic": (traffic / bytes_per_launch) if traffic else None,
         "achieved_algorithmic": achieved_alg, "frac_algorithmic": alg_frac,
         "achieved_fabric": fabric, "frac_fabric": (fabric / HBM_PEAK_GBS) if fabric else None,
         "frac_fabric_of_streaming_ceiling": (fabric / HBM_MEASURED_CEILING_GBS) if fabric else None,
         "dram_bytes_model": model.get("dram_bytes_model") if model else None,
         "frac_dram_model": model.get("frac_dram_model") if model else None,
         "frac_dram_model_of_streaming_ceiling": (model["frac_dram_model"] * HBM_PEAK_GBS / HBM_MEASURED_CEILING_GBS) if model and model.get("frac_dram_model") else None,
         "dram_model": model,
         "unique_rows_per_launch": unique,
         "cold_miss_bytes_per_launch": None, "frac_cold_miss_lower_bound": None,
         "dram_bytes_per_launch": None, "frac_dram": None,  # (no DRAM-side counter on this part; see dram_bytes_model)
         "gather_ceiling": g_alg, "gather_ceiling_source": g_src,
         "algorithmic_over_gather_ceiling": achieved_alg / g_alg,
         "gather_dram_ceiling": g_dram, "gather": gather,
         "gather_infinity_cache_ceiling": g_mall,
         "fabric_over_infinity_cache_gather": (fabric / g_mall) if fabric and g_mall else None,
         "dram_model_over_gather_dram_ceiling": (model["dram_bytes_model"] / launch_s / 1e9 / g_dram) if model and model.get("dram_bytes_model") else None,
         "streaming_ceiling": HBM_MEASURED_CEILING_GBS,
         "kernel": "k_search", "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_kernel_s * 1e3,
         "query_batches_rotated": B, "pmc": pmc_detail, "note": None}
    if unique is not None:
        cold = unique * (row_bytes + list_bytes)  # every distinct row once, and (at most) its own adjacency row
        cold = min(cold, unique * row_bytes + expansions_per_launch * list_bytes)
        r["cold_miss_bytes_per_launch"] = cold
        r["frac_cold_miss_lower_bound"] = cold / launch_s / 1e9 / HBM_PEAK_GBS
    notes = []
    if adc:
        r["kernel"] = "k_search over rows decoded on the fly (k_search_adc where subvectors are not whole 16-byte chunks, or with LANTERN_GPU_PQ_ADC=1)"
        notes.append("a compact pq index: HBM holds a row's code bytes only (1/32 of the f32 row at 96 subvectors); every 16-byte chunk of its "
                     "decoding comes from the per-subvector centroid tables, which stay in L2 (786 KB at 96 x 256 x 8 floats) -- the algorithmic HBM "
                     "bytes counted here are the code rows and adjacency rows; the walk is bound by L2 gathers of 32-byte centroid pieces, so the HBM "
                     "fraction says how little of the memory system it needs, not how good it is (lantern_amd/csrc/device_common.hpp PqdRow)")
    if S > 1:
        notes.append(f"{S} launches in flight: rates = all launches' bytes / the timed region; avg_launch_ms is the mean HIP-event "
                     "duration of launches that overlap")
    if alg_frac > 1.0 or achieved_alg > g_alg:
        notes.append("frac (algorithmic) exceeds " + ("1" if alg_frac > 1.0 else "the random-row gather ceiling") + ": rows shared by the queries of a launch "
                     "(upper levels, hub rows) are counted once per evaluation but served by L2 / Infinity Cache; frac_fabric and frac_dram_model are the physical figures")
    r["note"] = "; ".join(notes) or None
    return r


def build_roofline(a, c, prof, t_build, world, traffic=None):
    """Per-phase time of the build (HIP events recorded inside the library around each phase of every batch) against
    the algorithmic traffic of that phase (SURVEY.md 8d "Build unit of work"): the walk reads one row per distance
    evaluation and one adjacency row per expansion; a re-prune needs the cap+1 candidate rows and `close`'s once."""
    if not prof or not prof.get("batches"):
        return None
    row = a.dim * {"f32": 4, "f16": 2, "i8": 1, "b1": 0.125}[a.quant]
    scale = 1.0 / max(world, 1)  # counters and event times are this rank's share of a collective build
    walk_bytes = c["add_walk_evals"] * row + c["add_expansions"] * (2 * a.M * 4)
    out = {"note": "achieved = algorithmic bytes / time of that phase's kernels (HIP events inside the library); peak 8 TB/s HBM",
           "phases_ms": {k: prof[k] for k in ("walk_ms", "connect_ms", "group_ms", "revlink_ms", "exchange_ms")},
           "device_ms_total": sum(prof[k] for k in ("walk_ms", "connect_ms", "group_ms", "revlink_ms", "exchange_ms")),
           "host_and_idle_ms": max(0.0, t_build * 1e3 - sum(prof[k] for k in ("walk_ms", "connect_ms", "group_ms", "revlink_ms", "exchange_ms")))}
    gbs = walk_bytes / max(prof["walk_ms"], 1e-9) / 1e6
    out["walk"] = {"bound": "hbm", "algorithmic_bytes": float(walk_bytes), "ms": prof["walk_ms"], "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": gbs / HBM_PEAK_GBS, "frac_is": "frac_algorithmic (one row per distance evaluation of the insertion walks, one adjacency row per expansion)",
                   "traffic": None, "traffic_over_algorithmic": None, "frac_fabric": None}
    if traffic and "error" not in traffic:
        walk_t = sum(v["fetch_bytes"] for k, v in traffic.items() if k.startswith("k_insert"))
        if walk_t:
            out["walk"].update({"traffic": walk_t, "traffic_over_algorithmic": walk_t / max(walk_bytes, 1.0),
                                "frac_fabric": walk_t / max(prof["walk_ms"], 1e-9) / 1e6 / HBM_PEAK_GBS,
                                "traffic_source": traffic.get("_source")})
        out["traffic_by_kernel"] = {k: v for k, v in traffic.items() if not k.startswith("_")}
    elif traffic:
        out["traffic_error"] = traffic["error"]
    # the re-prune phase is bound by the latency of its dependent chains, not by bytes: the device cuts most requests to a full
    # list from the list's recorded radius without reading a row, so no byte roofline is claimed for it -- counts only
    out["reprune"] = {"bound": "latency (dependent chains per list)", "ms": prof["revlink_ms"], "requests_to_full_lists": c["add_reprunes"],
                      "distance_evaluations": c["add_revlink_evals"],
                      "rows_read_bytes": float(c["add_revlink_evals"]) * row}
    return out


def build_quality(a, kinds):
    """north_star: "recall@10 within +-0.5 % of the reference".  The reference builds with one usearch_add per tuple
    (build.c:83-135); the device builds batch-synchronously (batches of up to --add-batch, never more than size / 16).  On
    --build-quality-rows rows of the bench's OWN shape (--dim, --metric; seeds 1 / 2), for every data kind in `kinds`, both builds
    are made from the same rows -- the sequential one by the CPU port, usearch's own summation flags -- and searched ON THE DEVICE
    with the same queries against exact truth.  The sequential CPU builds (~55 s each at 100k x 768) run side by side on a thread each.
    (tests/test_gpu_fullsize.py asserts the same on the full 1M x 768 headline set, tests/test_gpu_baseline_configs.py on the C2
    set, a low-rank and a clustered set.)  -> {kind: result}"""
    from concurrent.futures import ThreadPoolExecutor

    from lantern_amd import capi, synth
    from oracle import binding as oracle

    n, d, metric = a.build_quality_rows, a.dim, a.metric
    sets = {}
    for kind in dict.fromkeys(kinds):
        make = synth.query_maker(kind, d)
        sets[kind] = (make(np.random.default_rng(1), n), make(np.random.default_rng(2), 1000))
    labels = np.arange(n, dtype=np.uint64) + 1

    def sequential(kind):  # (ctypes releases the GIL: the builds of the sets overlap)
        seq = oracle.OracleIndex(metric, d, M=a.M, ef_construction=a.efc, ef=a.ef, seed=42, sum_mode=oracle.SUM_FAST)
        seq.reserve(n)
        t0 = time.perf_counter()
        seq.add_many(labels, sets[kind][0])
        return seq.export_graph(), time.perf_counter() - t0

    with ThreadPoolExecutor(max_workers=max(1, len(sets))) as ex:
        futures = {kind: ex.submit(sequential, kind) for kind in sets}
        dev_side = {}
        for kind, (base, queries) in sets.items():  # the device builds meanwhile
            dev = capi.GpuIndex(metric, d, M=a.M, ef_construction=a.efc, ef=a.ef, seed=42)
            dev.reserve(n)
            dev.set_add_batch(a.add_batch, 16)
            t0 = time.perf_counter()
            dev.add_many(labels, base)
            dev.flush()
            t_dev = time.perf_counter() - t0
            truth, _ = dev.exact_search(queries, a.k)
            lab, _, _ = dev.search_batch(queries, a.k, a.ef)
            dev_side[kind] = (oracle.recall_at_k(lab.astype(np.int64) - 1, truth), t_dev, truth)
        graphs = {kind: f.result() for kind, f in futures.items()}
    out = {}
    for kind, (base, queries) in sets.items():
        r_dev, t_dev, truth = dev_side[kind]
        graph, t_seq = graphs[kind]
        ref = capi.GpuIndex(metric, d, M=a.M, ef_construction=a.efc, ef=a.ef, seed=42)
        ref.import_graph(base, graph)
        lab2, _, _ = ref.search_batch(queries, a.k, a.ef)
        r_seq = oracle.recall_at_k(lab2.astype(np.int64) - 1, truth)
        out[kind] = {"set": f"{n}x{d} f32 {metric} {kind}, seeds 1 / 2 (the bench's own --dim / --metric), 1000 queries, ef={a.ef}",
                     "recall_device_batched_build": r_dev, "recall_sequential_build": r_seq, "abs_diff": abs(r_dev - r_seq), "device_minus_sequential": r_dev - r_seq,
                     "bar": 0.005, "device_not_worse_than_sequential_by_more_than_bar": bool(r_dev >= r_seq - 0.005),
                     "device_build_seconds": t_dev, "sequential_cpu_build_seconds": t_seq, "sequential_cpu_build_vectors_per_s": n / t_seq,
                     "sequential_builds_side_by_side": len(sets)}
    return out


def clustered_coheadline(a, capi, hip, quality):
    """The headline's shape, index parameters and measuring procedure on the clustered set (lantern_amd/synth.py): value, recall,
    the roofline with counters of THIS run and the DRAM model, the CPU port on all cores and on one (median of three), the build
    quality leg, and BASELINE config[2]'s form (cosine, 1024-query batches) on the same rows."""
    import copy

    from lantern_amd import synth

    b = copy.copy(a)
    b.data, b.query_batches = "clustered", 4
    n, d, nq, B = b.n, b.dim, b.queries, 4
    t0 = time.time()
    base = synth.base_rows("clustered", n, d, b.base_seed)
    make = synth.query_maker("clustered", d)
    all_queries = make(np.random.default_rng(b.query_seed), nq * B)
    t_gen = time.time() - t0

    def build(metric):
        ix = capi.GpuIndex(metric, d, M=b.M, ef_construction=b.efc, ef=b.ef, seed=42)
        ix.reserve(n)
        ix.set_add_batch(b.add_batch, 16)
        ix.set_profiling(True)
        hip.synchronize()
        t0 = time.time()
        ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
        ix.flush()
        hip.synchronize()
        return ix, time.time() - t0

    def resident(ix, queries, per):
        out = []
        for i in range(queries.shape[0] // per):
            rows = ix.device_query_rows(queries[i * per:(i + 1) * per])
            out.append({"dq": hip.Buffer.from_numpy(rows), "stride": rows.strides[0], "lab": hip.Buffer(per * b.k * 8), "dist": hip.Buffer(per * b.k * 4),
                        "slot": hip.Buffer(per * b.k * 4), "D": hip.Buffer(per * 8), "E": hip.Buffer(per * 8)})
        return out

    def timed(ix, lanes, per, steps, stream):
        def step(i):
            L = lanes[i % len(lanes)]
            ix.search_batch_device(L["dq"].ptr, per, b.k, b.ef, 0, L["lab"].ptr, L["dist"].ptr, L["slot"].ptr, None, L["D"].ptr, L["E"].ptr, stream.handle,
                                   query_stride=L["stride"])
        for i in range(max(b.warmup, len(lanes))):
            step(i)
        hip.synchronize()
        ev = [(hip.Event(), hip.Event()) for _ in range(steps)]
        t0 = time.perf_counter()
        for i, (s_, e_) in enumerate(ev):
            s_.record(stream.handle)
            step(i)
            e_.record(stream.handle)
        hip.synchronize()
        elapsed = time.perf_counter() - t0
        per_lane = []
        for L in lanes:
            Dl = L["D"].download(per, np.uint64).astype(np.float64)
            El = L["E"].download(per, np.uint64).astype(np.float64)
            per_lane.append((Dl, El, float((Dl * d * 4 + El * (2 * b.M * 4) + d * 4).sum())))
        bytes_l = float(np.mean([per_lane[i % len(lanes)][2] for i in range(steps)]))
        exp_l = float(np.mean([per_lane[i % len(lanes)][1].sum() for i in range(steps)]))
        return step, elapsed, float(np.mean([s_.elapsed_ms(e_) for s_, e_ in ev])) / 1e3, bytes_l, exp_l, per_lane

    st = hip.Stream()
    ix, t_build = build(b.metric)
    counters, profile = ix.counters(), ix.build_profile()
    lanes = resident(ix, all_queries, nq)
    steps = max(8, min(b.steps, 12))
    step, elapsed, launch_s, bytes_l, exp_l, per_lane = timed(ix, lanes, nq, steps, st)
    two_in_flight = None
    if not b.no_two_in_flight:  # the same steps alternating over two streams (bench main: two_launches_in_flight)
        st2 = [st, hip.Stream()]

        def step2(i):
            L = lanes[i % len(lanes)]
            ix.search_batch_device(L["dq"].ptr, nq, b.k, b.ef, 0, L["lab"].ptr, L["dist"].ptr, L["slot"].ptr, None, L["D"].ptr, L["E"].ptr, st2[i % 2].handle,
                                   query_stride=L["stride"])
        for i in range(2):
            step2(i)
        hip.synchronize()
        t2 = time.perf_counter()
        for i in range(steps):
            step2(i)
        hip.synchronize()
        el2 = time.perf_counter() - t2
        two_in_flight = {"value": nq * steps / el2, "unit": "queries/s", "ms_per_step": el2 / steps * 1e3, "over_one_in_flight": elapsed / el2}
    tq = min(b.truth_queries, nq)
    truth = ix.exact_search(all_queries[:tq], b.k)[0]
    found = lanes[0]["slot"].download((nq, b.k), np.uint32)
    recall = float(np.mean([len(set(f.tolist()) & set(t.tolist())) / b.k for f, t in zip(found[:tq], truth)]))
    cpu = cpu_baseline(b, ix, base, all_queries[:nq], found[:tq]) if not b.no_cpu and b.cpu_seconds > 0 else None
    traffic = src = pmc = None
    if not b.no_pmc:
        pmc = measure_traffic(b, f"{ix.checksum():016x}")
        if pmc and pmc.get("hbm_bytes_per_launch"):
            traffic, src = pmc["hbm_bytes_per_launch"], pmc["source"]
    unique = unique_rows_per_launch(b, ix, step, B, hip)
    model = None if b.no_dram_model else dram_model(ix, hip, lanes, nq, b.k, b.ef, lanes[0]["stride"], d * 4, 2 * b.M * 4, launch_s, traffic,
                                                    (pmc or {}).get("read_bytes_per_launch") if traffic else None)
    qps = nq * steps / elapsed
    out = {"workload": f"HNSW search {n}x{d} f32 {b.metric} M={b.M} ef_construction={b.efc} ef={b.ef} k={b.k}, {nq}-query batches resident in HBM",
           "data": "synthetic (clustered: " + synth.CLUSTERED_DOC + ")", "value": qps, "unit": "queries/s", "steps": steps, "ms_per_step": elapsed / steps * 1e3,
           "two_launches_in_flight": two_in_flight,
           f"recall_at_{b.k}": recall, "recall_queries": tq, "dist_evals_per_query": float(per_lane[0][0].mean()), "expansions_per_query": float(per_lane[0][1].mean()),
           "roofline": roofline(bytes_l / launch_s / 1e9, traffic, src, launch_s, bytes_l, launch_s, 1, B, measured_here=traffic is not None, pmc_detail=pmc, unique=unique,
                                row_bytes=d * 4, list_bytes=2 * b.M * 4, expansions_per_launch=exp_l, model=model,
                                gather=None if b.no_gather_ceiling else gather_ceiling(ix, n, d * 4, int(min(max(float(per_lane[0][0].mean()) * nq, 2e6), 2e7)))),
           "cpu_baseline": cpu, "gpu_over_cpu_all_cores": (qps / cpu["value"]) if cpu else None, "gpu_over_cpu_1_thread": (qps / cpu["value_1_thread"]) if cpu else None,
           "build_vectors_per_s": n / t_build, "build_seconds": t_build, "build_roofline": build_roofline(b, counters, profile, t_build, 1),
           "build_quality": quality, "setup_seconds": {"datagen": t_gen}}
    # ---- BASELINE config[2]'s form on the same rows: cosine, 1024-query batches, one and two launches in flight
    try:
        cix, t_cbuild = build("cos")
        cq = 1024
        cl = resident(cix, all_queries[:cq * 8], cq)
        csteps = 24
        _, c_el, c_launch, c_bytes, _, c_lane = timed(cix, cl, cq, csteps, st)
        ctruth = cix.exact_search(all_queries[:cq], b.k)[0]
        cfound = cl[0]["slot"].download((cq, b.k), np.uint32)
        c_alg = c_bytes / c_launch / 1e9
        out["cosine_1024_query_batches"] = {
            "workload": f"BASELINE config[2]'s form on the clustered rows: HNSW search {n}x{d} f32 cos M={b.M} ef_construction={b.efc} ef={b.ef} k={b.k}, {cq}-query batches resident in HBM",
            "value": cq * csteps / c_el, "unit": "queries/s", "ms_per_step": c_el / csteps * 1e3,
            f"recall_at_{b.k}": float(np.mean([len(set(f.tolist()) & set(t.tolist())) / b.k for f, t in zip(cfound, ctruth)])),
            "dist_evals_per_query": float(c_lane[0][0].mean()), "build_vectors_per_s": n / t_cbuild,
            "roofline": {"bound": "hbm", "achieved": c_alg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": c_alg / HBM_PEAK_GBS, "frac_is": "frac_algorithmic (SURVEY 8d bytes / HIP-event launch time)",
                         "frac_of_gather_ceiling": c_alg / (((out.get("roofline") or {}).get("gather") or {}).get("algorithmic_gbs") or GATHER_CEILING_GBS), "algorithmic_bytes_per_launch": c_bytes, "avg_launch_ms": c_launch * 1e3, "kernel": "k_search", "traffic": None}}
    except Exception as ex:  # noqa: BLE001
        out["cosine_1024_query_batches"] = {"error": repr(ex)[:300]}
    return out


def cpu_baseline(a, ix, base, queries, gpu_found):
    """The oracle (CPU port of the usearch path) on the identical graph, on this host's cores."""
    from oracle import binding as oracle

    import bench_cpu

    native = oracle.build_native() and oracle.use_native(True)  # best CPU code for the baseline: -march=native on this host
    cores = bench_cpu.usable_cores()
    g = ix.export_graph()
    mode = oracle.SUM_FAST
    if a.quant == "f16":  # the CPU port works on the rounded values (f32 arithmetic, no conversion cost: favours the CPU)
        base, queries = oracle.round_f16(base), oracle.round_f16(queries)
    if a.quant == "i8":  # the quantised integers held as f32 (int32 accumulation in the port)
        base, queries, mode = oracle.quantize_i8(base), oracle.quantize_i8(queries), oracle.SUM_I8
    metric, dim = a.metric, a.dim
    if a.quant == "b1":  # quant_bits = 1: the sign bits, Hamming arithmetic (= l2sq over {0, 1} values)
        def pack(x):
            bits = (x > 0).astype(np.uint8)
            pad = (-bits.shape[1]) % 32
            if pad:
                bits = np.concatenate([bits, np.zeros((bits.shape[0], pad), np.uint8)], axis=1)
            return np.ascontiguousarray(np.packbits(bits, axis=1, bitorder="big")).view(np.uint32)
        base, queries, metric = pack(base), pack(queries), "hamming"
        dim = base.shape[1]
    ora = oracle.OracleIndex.from_graph(metric, base, g, a.M, a.efc, a.ef, 42, mode)
    # 1 thread and all cores, three timed repetitions each, median reported (bench_cpu.py)
    rates, slots = bench_cpu.search_rates(ora, queries, a.k, a.ef, a.cpu_seconds, cores)
    # index build on the CPU: the port's sequential usearch_add (a PostgreSQL backend builds with one thread,
    # utils.c:66) on a bounded prefix of the same rows.  The rate falls as the graph grows, so this flatters the CPU.
    nb = int(min(base.shape[0], 4096))
    cb = oracle.OracleIndex(metric, dim, M=a.M, ef_construction=a.efc, ef=a.ef, seed=42, sum_mode=mode)
    t0 = time.perf_counter()
    cb.add_many(np.arange(nb, dtype=np.uint64) + 1, base[:nb])
    cpu_build = nb / (time.perf_counter() - t0)
    del cb
    m = min(slots.shape[0], gpu_found.shape[0], queries.shape[0])
    agree = float(np.mean([len(set(x.tolist()) & set(y.tolist())) / a.k for x, y in zip(slots[:m], gpu_found[:m])]))
    rates.update({"topk_overlap_with_gpu": agree,
                  "build_vectors_per_s_1_thread": cpu_build, "build_sample": f"first {nb} rows, sequential usearch_add",
                  "build": bench_cpu.port_build_note(native),
                  "note": "oracle/hnsw.c restates the usearch algorithm; the reference binary itself cannot be built here"})
    return rates


if __name__ == "__main__":
    main()
