"""BASELINE.json's full-size configuration (1M x 768 f32, L2sq, M=16, ef_construction=128, ef=64, k=10)
checked through size-independent properties, plus exact parity against the oracle on a sample of queries
(the oracle walks the SAME graph, exported from the device).  Needs an MI355X; ~30 s."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, D, M, EFC, EF, K = 1_000_000, 768, 16, 128, 64, 10


@pytest.fixture(scope="module")
def world():
    from lantern_amd import capi, hip

    capi.lib()
    assert capi.device_count() > 0
    rng = np.random.default_rng(3)
    base = rng.standard_normal((N, D), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", D, M=M, ef_construction=EFC, ef=EF, seed=42)
    ix.reserve(N)
    ix.add_many(np.arange(N, dtype=np.uint64) + 1, base)
    ix.flush()
    queries = np.random.default_rng(4).standard_normal((2048, D), dtype=np.float32)
    return capi, hip, ix, base, queries


def run(hip, ix, queries, waves):
    nq = queries.shape[0]
    dq = hip.Buffer.from_numpy(hip.padded_rows(queries, False))
    lab, dist, slot = hip.Buffer(nq * K * 8), hip.Buffer(nq * K * 4), hip.Buffer(nq * K * 4)
    cnt, Dv, Ev = hip.Buffer(nq * 4), hip.Buffer(nq * 8), hip.Buffer(nq * 8)
    ix.set_search_shape(waves)
    ix.search_batch_device(dq.ptr, nq, K, EF, 0, lab.ptr, dist.ptr, slot.ptr, cnt.ptr, Dv.ptr, Ev.ptr)
    hip.synchronize()
    return (lab.download((nq, K), np.uint64), dist.download((nq, K), np.float32), slot.download((nq, K), np.uint32),
            cnt.download(nq, np.uint32), Dv.download(nq, np.uint64), Ev.download(nq, np.uint64))


def test_full_size_properties(world):
    capi, hip, ix, base, queries = world
    assert len(ix) == N
    before = ix.counters()
    lab, dist, slot, cnt, Dv, Ev = run(hip, ix, queries, 4)
    after = ix.counters()
    # sortedness, range, uniqueness, label = slot + 1
    assert np.all(cnt == K)
    assert np.all(np.diff(dist, axis=1) >= 0)
    assert slot.max() < N and np.array_equal(lab, slot.astype(np.uint64) + 1)
    assert all(len(set(r.tolist())) == K for r in slot)
    # checksum of checksums: the cumulative device counters advanced by exactly the per-query sums
    assert after["search_dist_evals"] - before["search_dist_evals"] == int(Dv.sum())
    assert after["search_expansions"] - before["search_expansions"] == int(Ev.sum())
    assert Dv.min() > EF and Ev.min() >= 1
    # idempotence and independence from the launch shape (1, 4, 8 wavefronts per query)
    for waves in (4, 1, 8):
        again = run(hip, ix, queries[:512], waves)
        assert np.array_equal(again[2], slot[:512]) and np.array_equal(again[1], dist[:512])
        assert np.array_equal(again[4], Dv[:512]) and np.array_equal(again[5], Ev[:512])
    # every reported distance is the distance kernel's value for that (query, row): bit-exact
    for qi in range(0, 2048, 256):
        assert np.array_equal(ix.distance_gather(queries[qi], slot[qi]), dist[qi])


def test_full_size_matches_oracle_on_sample(world, oracle):
    capi, hip, ix, base, queries = world
    g = ix.export_graph()
    ora = oracle.OracleIndex.from_graph("l2sq", base, g, M, EFC, EF, 42, oracle.SUM_WAVE64)
    sample = queries[:48]
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(sample, K, EF, 8)
    lab, dist, slot, cnt, Dv, Ev = run(hip, ix, sample, 4)
    assert np.array_equal(slot, o_slot) and np.array_equal(dist, o_dist) and np.array_equal(lab, o_lab)
    assert np.array_equal(Dv, o_D) and np.array_equal(Ev, o_E)
    # usearch-order CPU path: distances within 1e-5 relative, recall within 0.5 % (north_star)
    fast = oracle.OracleIndex.from_graph("l2sq", base, g, M, EFC, EF, 42, oracle.SUM_FAST)
    _, f_dist, f_slot, _, _ = fast.search_batch(sample, K, EF, 8)
    assert np.all(np.abs(f_dist - dist) <= 1e-5 * np.maximum(1.0, np.abs(f_dist)))
    truth, _ = ix.exact_search(sample, K)
    r_gpu, r_cpu = oracle.recall_at_k(slot, truth), oracle.recall_at_k(f_slot, truth)
    assert abs(r_gpu - r_cpu) <= 0.005


def test_full_size_exact_search_and_graph_invariants(world):
    capi, hip, ix, base, queries = world
    # a stored row queried exactly finds itself at distance 0 (round trip through the MFMA contraction + re-rank)
    rows = np.arange(0, N, N // 64)[:64]
    slots, dists = ix.exact_search(base[rows], 3)
    assert np.array_equal(slots[:, 0], rows.astype(np.uint32)) and np.all(dists[:, 0] == 0)
    g = ix.export_graph()
    nbr0 = g["nbr0"]
    valid = nbr0 != 0xFFFFFFFF
    # lists are EMPTY-terminated (no holes), hold no self loops and no out-of-range slots
    assert np.all(valid[:, :-1] >= valid[:, 1:])
    assert nbr0[valid].max() < N
    assert not np.any(nbr0 == np.arange(N, dtype=np.uint32)[:, None])
    assert valid.sum(axis=1).min() >= 1
    # level distribution: P(level >= 1) = 1/M (insert.c:32-46)
    assert abs((g["levels"] >= 1).mean() - 1 / M) < 0.002
    assert g["levels"][g["entry_slot"]] == g["max_level"]


def test_full_size_build_is_the_same_with_and_without_the_recorded_radii(world, monkeypatch):
    """The headline build (1M x 768, 284 batches of up to 8192 rows) with the re-prune state switched off -- every request to
    a full list through the all-pairs table -- gives the identical graph: the radius cut drops only requests that would be
    cut anyway (checksum over every list of every level)."""
    capi, hip, ix, base, queries = world
    monkeypatch.setenv("LANTERN_GPU_REPRUNE_STATE", "0")
    plain = capi.GpuIndex("l2sq", D, M=M, ef_construction=EFC, ef=EF, seed=42)
    plain.reserve(N)
    plain.add_many(np.arange(N, dtype=np.uint64) + 1, base)
    plain.flush()
    assert plain.checksum() == ix.checksum()
    assert plain.counters()["add_revlink_evals"] > ix.counters()["add_revlink_evals"]  # ... and the state really was off


@pytest.mark.skipif(os.environ.get("LANTERN_TEST_1M_SEQUENTIAL", "0") != "1",
                    reason="a sequential CPU build of 1M x 768 takes ~15 minutes of one core (1.7 k vectors/s at 100k rows, falling); set "
                           "LANTERN_TEST_1M_SEQUENTIAL=1.  The same comparison without a device: scripts/build_quality_cpu.py -> profiles/r03_build_quality_1Mx768.json")
def test_full_size_batched_build_against_the_sequential_reference_build(world, oracle):
    """north_star: "recall@10 within +-0.5 % of the reference".  The reference builds with one usearch_add per tuple
    (build.c:83-135); the device with batches of up to 8192.  Both graphs of the HEADLINE set (1M x 768), searched on the
    device with the same 1000 queries against exact truth.  The sequential build is the CPU port with the reference's own
    summation flags on one thread."""
    import time

    capi, hip, ix, base, queries = world
    q = queries[:1000]
    truth, _ = ix.exact_search(q, K)
    r_dev = oracle.recall_at_k(run(hip, ix, q, 4)[2], truth)
    seq = oracle.OracleIndex("l2sq", D, M=M, ef_construction=EFC, ef=EF, seed=42, sum_mode=oracle.SUM_FAST)
    seq.reserve(N)
    t0 = time.time()
    seq.add_many(np.arange(N, dtype=np.uint64) + 1, base)
    t_seq = time.time() - t0
    g = seq.export_graph()
    del seq
    ref = capi.GpuIndex("l2sq", D, M=M, ef_construction=EFC, ef=EF, seed=42)
    ref.import_graph(base, g)
    r_seq = oracle.recall_at_k(run(hip, ref, q, 4)[2], truth)
    print(f"1M x 768: recall@10 device-batched build {r_dev:.4f}, sequential reference build {r_seq:.4f} (CPU, {t_seq:.0f} s, {N / t_seq:.0f} vectors/s)")
    assert abs(r_dev - r_seq) <= 0.005, (r_dev, r_seq)
    gd = ix.export_graph()
    deg_dev = (gd["nbr0"] != 0xFFFFFFFF).sum(axis=1).mean()
    deg_seq = (g["nbr0"] != 0xFFFFFFFF).sum(axis=1).mean()
    assert abs(deg_dev - deg_seq) / deg_seq < 0.05, (deg_dev, deg_seq)
    assert np.array_equal(gd["levels"], g["levels"])


# ------------------------------------------------------------------------------------------------------------------
# The same size on data with neighbourhood structure (lantern_amd/synth.py "clustered"): the regime in which the reference
# asserts recall (scripts/integration_tests.py:249-264: >= 0.7, warning below 0.9).  On i.i.d. N(0,1) rows recall@10 is 0.18
# for the CPU path and the device alike; here the walk finds the neighbours, and D, E, the hub structure and therefore the
# cache behaviour are those of a real embedding set.
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def clustered():
    from lantern_amd import capi, hip, synth

    capi.lib()
    assert capi.device_count() > 0
    base = synth.base_rows("clustered", N, D)
    queries = synth.query_maker("clustered", D)(np.random.default_rng(4), 2048)
    ix = capi.GpuIndex("l2sq", D, M=M, ef_construction=EFC, ef=EF, seed=42)
    ix.reserve(N)
    ix.add_many(np.arange(N, dtype=np.uint64) + 1, base)
    ix.flush()
    return capi, hip, ix, base, queries


def test_clustered_full_size_recall_and_oracle_parity(clustered, oracle):
    capi, hip, ix, base, queries = clustered
    lab, dist, slot, cnt, Dv, Ev = run(hip, ix, queries, 4)
    truth, _ = ix.exact_search(queries[:1024], K)
    recall = oracle.recall_at_k(slot[:1024], truth)
    print(f"clustered 1M x 768 l2sq: recall@10 {recall:.4f}, D {Dv.mean():.0f}, E {Ev.mean():.1f}")
    assert recall >= 0.9, recall
    assert np.all(cnt == K) and np.all(np.diff(dist, axis=1) >= 0)
    # the oracle on the same graph, in the device's summation order: identical ids, distance bits, D, E
    g = ix.export_graph()
    ora = oracle.OracleIndex.from_graph("l2sq", base, g, M, EFC, EF, 42, oracle.SUM_WAVE64)
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(queries[:64], K, EF, 8)
    assert np.array_equal(slot[:64], o_slot) and np.array_equal(dist[:64], o_dist) and np.array_equal(lab[:64], o_lab)
    assert np.array_equal(Dv[:64], o_D) and np.array_equal(Ev[:64], o_E)
    # the usearch-order CPU path finds the same neighbours: recall within 0.5 % (north_star), distances within 1e-5
    fast = oracle.OracleIndex.from_graph("l2sq", base, g, M, EFC, EF, 42, oracle.SUM_FAST)
    _, f_dist, f_slot, _, _ = fast.search_batch(queries[:1024], K, EF, 8)
    assert abs(oracle.recall_at_k(f_slot, truth) - recall) <= 0.005
    assert np.all(np.abs(f_dist[:64] - dist[:64]) <= 1e-5 * np.maximum(1.0, np.abs(f_dist[:64])))
    # cosine over the same rows (a second index): also in the asserted regime
    cos = capi.GpuIndex("cos", D, M=M, ef_construction=EFC, ef=EF, seed=42)
    cos.reserve(N)
    cos.add_many(np.arange(N, dtype=np.uint64) + 1, base)
    cos.flush()
    c_slot = run(hip, cos, queries[:1024], 4)[2]
    c_truth, _ = cos.exact_search(queries[:1024], K)
    c_recall = oracle.recall_at_k(c_slot, c_truth)
    print(f"clustered 1M x 768 cos: recall@10 {c_recall:.4f}")
    assert c_recall >= 0.9, c_recall
