"""BASELINE.json configs [3] and [4] AT THEIR OWN SIZES, device against the oracle on the same graph:

  C4  10 000 000 x 768 f32 L2sq, seed 5; queries 8 192 x 768, seed 6; M=16 ef_construction=128 ef=128 k=10.  The only place
      where the vector block passes 4 GB (30.7 GB), slot ids pass 2^23 and the visited bitmap is 10M bits wide.
  C5  1 000 000 x 1536 f32, seed 7; M=16 ef_construction=128; recall@10 at ef=64 on the 1 000 seed-8 queries; plus the
      edge-for-edge production-plan build at 1536-d and the batched-vs-sequential recall at 100k x 1536.

What is compared (lantern_hnsw/src/hnsw/scan.c:220-228 for the search, build.c:83-135 for the build): on a sample of queries
the oracle (oracle/hnsw.c, LO_SUM_WAVE64 = the device's summation tree) walks the graph EXPORTED from the device -- slots,
labels, distance bits, D and E must be equal; the usearch-order CPU path (LO_SUM_FAST) gives distances within 1e-5 relative
and recall within 0.005 (north_star); over the whole batch the size-independent properties (sortedness, uniqueness,
checksum of the device counters, idempotence across launch shapes, every reported distance = the pair kernel's bits).

C4 needs ~36 GB of host memory for the rows and the exported graph (skipped below 48 GB available); ~90 s, most of it numpy
drawing 7.7 G normals."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M, EFC, K = 16, 128, 10


def host_memory_available() -> int:
    """Bytes this process may still allocate: MemAvailable capped by the cgroup limit."""
    avail = 0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except Exception:
        return 0
    for p, q in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                 ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            lim = open(p).read().strip()
            if lim != "max":
                avail = min(avail, int(lim) - int(open(q).read().strip()))
            break
        except Exception:
            continue
    return avail


def run(hip, ix, queries, ef, waves=0):
    nq = queries.shape[0]
    dq = hip.Buffer.from_numpy(hip.padded_rows(queries, False))
    lab, dist, slot = hip.Buffer(nq * K * 8), hip.Buffer(nq * K * 4), hip.Buffer(nq * K * 4)
    cnt, Dv, Ev = hip.Buffer(nq * 4), hip.Buffer(nq * 8), hip.Buffer(nq * 8)
    ix.set_search_shape(waves)
    ix.search_batch_device(dq.ptr, nq, K, ef, 0, lab.ptr, dist.ptr, slot.ptr, cnt.ptr, Dv.ptr, Ev.ptr)
    hip.synchronize()
    ix.set_search_shape(0)
    return (lab.download((nq, K), np.uint64), dist.download((nq, K), np.float32), slot.download((nq, K), np.uint32),
            cnt.download(nq, np.uint32), Dv.download(nq, np.uint64), Ev.download(nq, np.uint64))


def build(capi, metric, base, ef, plan=None):
    # (8192, 16) unless LANTERN_TEST_ADD_BATCH says otherwise: bench.py builds with 16384-row batches since round 5, and the at-size
    # parity below has been run once with that plan too (profiles/r05_c4_at_size_plan16384.txt)
    if plan is None:
        plan = (int(os.environ.get("LANTERN_TEST_ADD_BATCH", "8192")), 16)
    ix = capi.GpuIndex(metric, base.shape[1], M=M, ef_construction=EFC, ef=ef, seed=42)
    ix.reserve(base.shape[0])
    ix.set_add_batch(*plan)
    t0 = time.time()
    ix.add_many(np.arange(base.shape[0], dtype=np.uint64) + 1, base)
    ix.flush()
    return ix, time.time() - t0


def batch_properties(hip, ix, queries, ef, n, out):
    """The size-independent properties of tests/test_gpu_fullsize.py::test_full_size_properties at this size."""
    lab, dist, slot, cnt, Dv, Ev = out
    assert np.all(cnt == K)
    assert np.all(np.diff(dist, axis=1) >= 0)
    assert slot.max() < n and np.array_equal(lab, slot.astype(np.uint64) + 1)
    s = np.sort(slot, axis=1)
    assert np.all(s[:, 1:] != s[:, :-1])  # no row twice in any answer
    assert Dv.min() > ef and Ev.min() >= 1
    for waves in (4, 1, 8):  # idempotence, independence from the launch shape
        again = run(hip, ix, queries[:512], ef, waves)
        assert np.array_equal(again[2], slot[:512]) and np.array_equal(again[1], dist[:512])
        assert np.array_equal(again[4], Dv[:512]) and np.array_equal(again[5], Ev[:512])
    for qi in range(0, queries.shape[0], max(1, queries.shape[0] // 8)):  # every reported distance is the pair kernel's value
        assert np.array_equal(ix.distance_gather(queries[qi], slot[qi]), dist[qi])


def oracle_sample_parity(oracle, ix, metric, base, queries, ef, out, sample, recall_sample, cores):
    lab, dist, slot, cnt, Dv, Ev = out
    g = ix.export_graph()
    ora = oracle.OracleIndex.from_graph(metric, base, g, M, EFC, ef, 42, oracle.SUM_WAVE64)
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(queries[:sample], K, ef, cores)
    assert np.array_equal(slot[:sample], o_slot) and np.array_equal(lab[:sample], o_lab)
    assert np.array_equal(dist[:sample].view(np.uint32), o_dist.view(np.uint32))  # distance BITS
    assert np.array_equal(Dv[:sample], o_D) and np.array_equal(Ev[:sample], o_E)
    # the reference's summation (usearch loop under -fassociative-math): 1e-5 relative, recall within 0.5 % (north_star)
    fast = oracle.OracleIndex.from_graph(metric, base, g, M, EFC, ef, 42, oracle.SUM_FAST)
    _, f_dist, f_slot, _, _ = fast.search_batch(queries[:recall_sample], K, ef, cores)
    assert np.all(np.abs(f_dist[:sample] - dist[:sample]) <= 1e-5 * np.maximum(1.0, np.abs(f_dist[:sample])))
    truth, _ = ix.exact_search(queries[:recall_sample], K)
    r_gpu, r_cpu = oracle.recall_at_k(slot[:recall_sample], truth), oracle.recall_at_k(f_slot, truth)
    assert abs(r_gpu - r_cpu) <= 0.005, (r_gpu, r_cpu)
    return g, r_gpu, r_cpu


# ------------------------------------------------------------------------------------------------------------------
# C4: 10M x 768, ef = 128
# ------------------------------------------------------------------------------------------------------------------
C4_N, C4_D, C4_EF = 10_000_000, 768, 128


@pytest.mark.skipif(host_memory_available() < 48 << 30, reason="C4 at size holds 30.7 GB of rows + the exported graph on the host: needs >= 48 GB available")
def test_c4_ten_million_rows_ef_128_matches_the_oracle(oracle, cores):
    from lantern_amd import capi, hip

    capi.lib()
    assert capi.device_count() > 0
    t0 = time.time()
    base = np.random.default_rng(5).standard_normal((C4_N, C4_D), dtype=np.float32)
    queries = np.random.default_rng(6).standard_normal((8192, C4_D), dtype=np.float32)
    t_gen = time.time() - t0
    ix, t_build = build(capi, "l2sq", base, C4_EF)
    assert len(ix) == C4_N
    before = ix.counters()
    out = run(hip, ix, queries, C4_EF)
    after = ix.counters()
    lab, dist, slot, cnt, Dv, Ev = out
    # checksum of checksums: the cumulative device counters advanced by exactly the per-query sums
    assert after["search_dist_evals"] - before["search_dist_evals"] == int(Dv.sum())
    assert after["search_expansions"] - before["search_expansions"] == int(Ev.sum())
    batch_properties(hip, ix, queries, C4_EF, C4_N, out)
    # this size's own territory: answers beyond slot 2^23 (a row offset past 4 GB needs more than 32 bits)
    assert (slot >= (1 << 23)).any() and int(slot.max()) * C4_D * 4 > (1 << 32)
    g, r_gpu, r_cpu = oracle_sample_parity(oracle, ix, "l2sq", base, queries, C4_EF, out, 32, 256, cores)
    print(f"C4 10M x 768 ef=128: datagen {t_gen:.0f} s, build {t_build:.1f} s ({C4_N / t_build:.0f} vectors/s), recall@10 device {r_gpu:.4f} "
          f"CPU port {r_cpu:.4f}, D {Dv.mean():.0f}, E {Ev.mean():.1f}")
    # graph invariants at this size
    nbr0 = g["nbr0"]
    valid = nbr0 != 0xFFFFFFFF
    assert np.all(valid[:, :-1] >= valid[:, 1:]) and nbr0[valid].max() < C4_N and valid.sum(axis=1).min() >= 1
    assert abs((g["levels"] >= 1).mean() - 1 / M) < 0.001 and g["levels"][g["entry_slot"]] == g["max_level"]


# ------------------------------------------------------------------------------------------------------------------
# C5: 1M x 1536 build, recall on the 1000 seed-8 queries
# ------------------------------------------------------------------------------------------------------------------
C5_N, C5_D = 1_000_000, 1536


def test_c5_million_rows_1536_dims_build_and_recall(oracle, cores):
    from lantern_amd import capi, hip

    capi.lib()
    assert capi.device_count() > 0
    base = np.random.default_rng(7).standard_normal((C5_N, C5_D), dtype=np.float32)
    queries = np.random.default_rng(8).standard_normal((1000, C5_D), dtype=np.float32)
    ix, t_build = build(capi, "l2sq", base, 64)
    assert len(ix) == C5_N
    out = run(hip, ix, queries, 64)
    lab, dist, slot, cnt, Dv, Ev = out
    batch_properties(hip, ix, queries, 64, C5_N, out)
    g = ix.export_graph()
    # recall@10 (ef=64) of the resulting graph on the 1 000 seed-8 queries -- the device's figure against the
    # CPU port's on the same graph: the device-order port gives the SAME answers (all 1000), the usearch-order port the same
    # recall within the bar
    ora = oracle.OracleIndex.from_graph("l2sq", base, g, M, EFC, 64, 42, oracle.SUM_WAVE64)
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(queries, K, 64, cores)
    assert np.array_equal(slot, o_slot) and np.array_equal(lab, o_lab)
    assert np.array_equal(dist.view(np.uint32), o_dist.view(np.uint32))
    assert np.array_equal(Dv, o_D) and np.array_equal(Ev, o_E)
    truth, _ = ix.exact_search(queries, K)
    r_gpu = oracle.recall_at_k(slot, truth)
    assert r_gpu == oracle.recall_at_k(o_slot, truth)
    fast = oracle.OracleIndex.from_graph("l2sq", base, g, M, EFC, 64, 42, oracle.SUM_FAST)
    _, f_dist, f_slot, _, _ = fast.search_batch(queries, K, 64, cores)
    r_cpu = oracle.recall_at_k(f_slot, truth)
    assert abs(r_gpu - r_cpu) <= 0.005, (r_gpu, r_cpu)
    assert np.all(np.abs(f_dist - dist) <= 1e-5 * np.maximum(1.0, np.abs(f_dist)))
    print(f"C5 1M x 1536: build {t_build:.2f} s ({C5_N / t_build:.0f} vectors/s), recall@10 ef=64 on the 1000 seed-8 queries: device {r_gpu:.4f}, "
          f"CPU port (usearch order) {r_cpu:.4f}; D {Dv.mean():.0f}, E {Ev.mean():.1f}")
    nbr0 = g["nbr0"]
    valid = nbr0 != 0xFFFFFFFF
    assert np.all(valid[:, :-1] >= valid[:, 1:]) and nbr0[valid].max() < C5_N and valid.sum(axis=1).min() >= 1
    assert abs((g["levels"] >= 1).mean() - 1 / M) < 0.002


@pytest.mark.slow  # 150 s: the sequential CPU build of 100k x 1536 rows; in the default run: the same comparison at 100k x 128 / 100k x 768 clustered (test_gpu_baseline_configs.py) and C5 edge for edge at 64k x 1536
@pytest.mark.skipif(os.environ.get("LANTERN_TEST_SLOW", "0") in ("", "0"), reason="slow at-size comparison: LANTERN_TEST_SLOW=1 (bash scripts/gpu.sh tests-slow)")
def test_c5_batched_build_against_the_sequential_reference_build_at_100k_x_1536(oracle):
    """north_star "recall@10 within +-0.5 % of the reference" on C5's own row shape: the reference adds one tuple at a time
    (build.c:83-135); the device in batches of up to 8192 (never more than size / 16).  Same 100k seed-7 rows, same seed-8
    queries, both graphs searched on the device against exact truth.  The sequential build is the CPU port on one thread
    with the reference's summation flags (~100-170 s: the price of this test).

    8 000 queries, not C5's 1 000: on i.i.d. Gaussian rows recall@10 is ~0.36 and its standard error over 1 000 queries is
    ~0.005 -- the bar itself (measured in round 4: the same comparison gave +0.0053 on seeds 1 / 2 and -0.0063 on seeds 7 / 8
    with 1 000 queries each).  The first 1 000 of the 8 000 ARE the seed-8 queries of SURVEY 8d."""
    from lantern_amd import capi, hip

    n = 100_000
    base = np.random.default_rng(7).standard_normal((n, C5_D), dtype=np.float32)
    queries = np.random.default_rng(8).standard_normal((8000, C5_D), dtype=np.float32)
    dev, t_dev = build(capi, "l2sq", base, 64)
    truth, _ = dev.exact_search(queries, K)
    r_dev = oracle.recall_at_k(run(hip, dev, queries, 64)[2], truth)
    seq = oracle.OracleIndex("l2sq", C5_D, M=M, ef_construction=EFC, ef=64, seed=42, sum_mode=oracle.SUM_FAST)
    seq.reserve(n)
    t0 = time.time()
    seq.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    t_seq = time.time() - t0
    gs = seq.export_graph()
    del seq
    ref = capi.GpuIndex("l2sq", C5_D, M=M, ef_construction=EFC, ef=64, seed=42)
    ref.import_graph(base, gs)
    r_seq = oracle.recall_at_k(run(hip, ref, queries, 64)[2], truth)
    print(f"100k x 1536, {queries.shape[0]} queries: recall@10 device-batched build {r_dev:.4f} ({t_dev:.2f} s), sequential reference build {r_seq:.4f} "
          f"(CPU, {t_seq:.0f} s, {n / t_seq:.0f} vectors/s)")
    assert abs(r_dev - r_seq) <= 0.005, (r_dev, r_seq)
    gd = dev.export_graph()
    deg_dev = (gd["nbr0"] != 0xFFFFFFFF).sum(axis=1).mean()
    deg_seq = (gs["nbr0"] != 0xFFFFFFFF).sum(axis=1).mean()
    assert abs(deg_dev - deg_seq) / deg_seq < 0.05, (deg_dev, deg_seq)
    assert np.array_equal(gd["levels"], gs["levels"])
