"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Needs an MI355X.

Bar: bit-exact against the oracle's LO_SUM_WAVE64 order (the device's own reduction tree) for
every metric; within 1e-5 relative of the usearch-order (LO_SUM_SEQ) result for L2sq/cosine;
identical top-k id lists, identical distance-evaluation and expansion counts on the same graph.
"""
import numpy as np
import pytest

from tests.conftest import needs_experimental
from tests.scan_driver import scan as oracle_scan

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star: L2sq / cosine within 1e-5 relative
LABEL0 = 1


@pytest.fixture(scope="module")
def capi():
    from lantern_amd import capi

    capi.lib()
    assert capi.device_count() > 0, "no HIP device: the gpu tests need a real MI355X"
    return capi


def rand_rows(rng, n, d, metric):
    if metric == "hamming":
        return rng.integers(0, 2**32, size=(n, d), dtype=np.uint32)
    return rng.standard_normal((n, d), dtype=np.float32)


# ------------------------------------------------------------------------------------------------
# distances
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("metric", ["l2sq", "cos", "hamming"])
@pytest.mark.parametrize("d", [1, 3, 4, 17, 63, 64, 100, 128, 255, 256, 768, 1536, 2000])
def test_pair_distance_bit_exact_and_within_tolerance(capi, oracle, metric, d):
    rng = np.random.default_rng(d)
    if metric == "hamming" and d > 256:
        pytest.skip("hamming rows are at most 2000*32 bits in Lantern; 256 words covers the group sizes")
    a, b = rand_rows(rng, 1, d, metric)[0], rand_rows(rng, 1, d, metric)[0]
    got = capi.distance(a, b, metric)
    assert got == oracle.distance(a, b, metric, oracle.SUM_WAVE64)
    ref = oracle.distance(a, b, metric, oracle.SUM_SEQ)
    if metric == "hamming":
        assert got == ref
    else:
        assert abs(got - ref) <= TOL * max(1.0, abs(ref))


def test_golden_operator_cases_on_device(capi, golden):
    for c in golden["operators"]["cases"]:
        fn = {"l2sq": capi.l2sq_dist, "cos": capi.cos_dist, "hamming": capi.hamming_dist}[c["op"]]
        d = fn(c["a"], c["b"])
        if "expect" in c:
            assert d == c["expect"], c
        else:
            assert round(d, 2) == c["expect_2dp"], c
    # cosine zero-vector rules (hnsw_vector.out:205-210, hnsw_dist_func.out:58-61)
    assert capi.cos_dist([0, 0, 0], [0, 0, 0]) == 0.0
    assert capi.cos_dist([0, 0, 0], [0, 0, 2]) == 1.0
    assert capi.cos_dist([0, 0, 1], [0, 0, 0]) == 1.0
    with pytest.raises(capi.LanternGpuError, match="expected equally sized arrays but got arrays with dimensions 2 and 3"):
        capi.cos_dist([1, 1], [0, 1, 0])


@pytest.mark.parametrize("metric", ["l2sq", "cos", "hamming"])
def test_distance_matrix_exact_order(capi, oracle, metric):
    rng = np.random.default_rng(5)
    d = 24 if metric == "hamming" else 96
    A, B = rand_rows(rng, 7, d, metric), rand_rows(rng, 33, d, metric)
    got = capi.distance_matrix(A, B, metric, exact_order=True)
    ref = np.array([[oracle.distance(a, b, metric, oracle.SUM_WAVE64) for b in B] for a in A], dtype=np.float32)
    assert np.array_equal(got, ref)


# ------------------------------------------------------------------------------------------------
# search on an identical graph
# ------------------------------------------------------------------------------------------------
CASES = [
    # metric, n, d, M, efc, ef, k
    ("l2sq", 3000, 128, 16, 64, 64, 10),   # BASELINE config[1] shape, reduced n
    ("cos", 2000, 768, 16, 64, 64, 10),    # config[2]/[3] row width
    ("l2sq", 1500, 100, 8, 40, 32, 5),     # ragged width (G=32, padded chunk)
    ("l2sq", 800, 3, 2, 10, 4, 1),         # README example parameters (M=2, efc=10, ef=4)
    ("hamming", 3000, 24, 16, 64, 64, 10), # SURVEY 8(d) hamming check set shape
    ("cos", 600, 1536, 16, 32, 128, 10),   # config[4] width, ef=128
]


@pytest.mark.parametrize("metric,n,d,M,efc,ef,k", CASES)
def test_search_matches_oracle_on_same_graph(capi, oracle, metric, n, d, M, efc, ef, k):
    rng = np.random.default_rng(n + d)
    base, queries = rand_rows(rng, n, d, metric), rand_rows(rng, 64, d, metric)
    ora = oracle.OracleIndex(metric, d, M=M, ef_construction=efc, ef=ef, seed=9, sum_mode=oracle.SUM_WAVE64)
    ora.add_many(np.arange(n, dtype=np.uint64) + LABEL0, base)
    g = ora.export_graph()
    gpu = capi.GpuIndex(metric, d, M=M, ef_construction=efc, ef=ef, seed=9)
    gpu.import_graph(base, g)
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(queries, k)
    from lantern_amd import hip

    rows = gpu.device_query_rows(queries)
    dq = hip.Buffer.from_numpy(rows)
    lab, dist, slot = hip.Buffer(64 * k * 8), hip.Buffer(64 * k * 4), hip.Buffer(64 * k * 4)
    cnt, D, E = hip.Buffer(64 * 4), hip.Buffer(64 * 8), hip.Buffer(64 * 8)
    for waves in (1, 4, 8):
        gpu.set_search_shape(waves)
        gpu.search_batch_device(dq.ptr, 64, k, 0, 0, lab.ptr, dist.ptr, slot.ptr, cnt.ptr, D.ptr, E.ptr, query_stride=rows.strides[0])
        hip.synchronize()
        assert np.array_equal(slot.download((64, k), np.uint32), o_slot), f"top-k slots differ (waves={waves})"
        assert np.array_equal(lab.download((64, k), np.uint64), o_lab)
        assert np.array_equal(dist.download((64, k), np.float32), o_dist)
        assert np.array_equal(D.download(64, np.uint64), o_D), "distance-evaluation counts differ"
        assert np.array_equal(E.download(64, np.uint64), o_E), "expansion counts differ"
    # host-buffer entry points agree with the device-buffer one
    h_lab, h_dist, h_cnt = gpu.search_batch(queries, k)
    assert np.array_equal(h_lab, o_lab) and np.array_equal(h_dist, o_dist)
    l1, d1 = gpu.search(queries[0], k)
    assert np.array_equal(l1, o_lab[0][: len(l1)]) and np.array_equal(d1, o_dist[0][: len(d1)])
    # usearch-order oracle on the same graph: same ids except across near-ties, distances within tolerance
    seq = oracle.OracleIndex.from_graph(metric, base, g, M, efc, ef, 9, oracle.SUM_SEQ)
    _, s_dist, s_slot, _, _ = seq.search_batch(queries, k)
    if metric == "hamming":
        assert np.array_equal(s_slot, o_slot)
    else:
        assert np.all(np.abs(s_dist - o_dist) <= TOL * np.maximum(1.0, np.abs(s_dist)))
        assert oracle.recall_at_k(o_slot, s_slot) >= 0.995


# shapes chosen to hit every reverse-link kernel: LDS-staged rows (short rows), the column-slab sweep
# (>= 64 chunks, M <= 16; all three metrics; 1 and 6 slabs), and the unstaged one (M = 40: 82 rows do not fit in LDS)
@pytest.mark.parametrize("metric,n,d,M,efc", [("l2sq", 1200, 64, 8, 40), ("cos", 700, 256, 16, 64), ("hamming", 900, 8, 6, 32),
                                              ("l2sq", 300, 5, 2, 10), ("hamming", 600, 256, 8, 32), ("l2sq", 700, 1536, 16, 48),
                                              ("cos", 500, 1000, 12, 40), ("l2sq", 400, 512, 40, 48),
                                              # small M on high-dimensional Gaussian rows: lists fill at once and hub nodes collect long CHAINS
                                              # of re-prunes per batch (k_revlink_pairs' chain mode); 2000-d = eight chunks per lane
                                              ("l2sq", 2500, 768, 4, 32), ("cos", 1500, 2000, 8, 40), ("l2sq", 900, 520, 16, 64)])
@pytest.mark.parametrize("plan", [(1, 1), (64, 4), (512, 16)])
def test_build_matches_oracle_edge_for_edge(capi, oracle, metric, n, d, M, efc, plan):
    rng = np.random.default_rng(n * 7 + d)
    base = rand_rows(rng, n, d, metric)
    labels = np.arange(n, dtype=np.uint64) + LABEL0
    ora = oracle.OracleIndex(metric, d, M=M, ef_construction=efc, ef=32, seed=21, sum_mode=oracle.SUM_WAVE64)
    ora.add_planned(labels, base, max_batch=plan[0], min_ratio=plan[1])
    gpu = capi.GpuIndex(metric, d, M=M, ef_construction=efc, ef=32, seed=21)
    gpu.set_add_batch(*plan)
    gpu.add_many(labels, base)
    gpu.flush()
    assert len(gpu) == n
    go, gg = ora.export_graph(), gpu.export_graph(with_vectors=True)
    assert gg["entry_slot"] == go["entry_slot"] and gg["max_level"] == go["max_level"]
    assert np.array_equal(gg["levels"], go["levels"])
    assert np.array_equal(gg["labels"], go["labels"])
    assert np.array_equal(gg["upper_off"], go["upper_off"])
    assert np.array_equal(gg["nbr0"], go["nbr0"]), "level-0 adjacency differs"
    assert np.array_equal(gg["upper_nbr"], go["upper_nbr"]), "upper-level adjacency differs"
    assert np.array_equal(gg["vectors"], base)


def test_sequential_plan_is_usearch_add(capi, oracle):
    # max_batch = 1 must reproduce the strictly sequential usearch_add semantics (oracle lo_add)
    rng = np.random.default_rng(3)
    base = rng.standard_normal((400, 32), dtype=np.float32)
    ora = oracle.OracleIndex("l2sq", 32, M=8, ef_construction=32, seed=4, sum_mode=oracle.SUM_WAVE64)
    ora.add_many(np.arange(400) + 1, base)
    gpu = capi.GpuIndex("l2sq", 32, M=8, ef_construction=32, seed=4)
    gpu.set_add_batch(1, 1)
    for i in range(400):
        gpu.add(i + 1, base[i])
    assert np.array_equal(gpu.export_graph()["nbr0"], ora.export_graph()["nbr0"])


# ------------------------------------------------------------------------------------------------
# the reference's golden cases through the device index and its scan shim
# ------------------------------------------------------------------------------------------------
def gpu_build(capi, metric, rows, M=16, efc=128, ef=64):
    rows = np.asarray(rows)
    ix = capi.GpuIndex(metric, rows.shape[1], M=M, ef_construction=efc, ef=ef, seed=7)
    ix.add_many(np.arange(rows.shape[0], dtype=np.uint64) + LABEL0, rows)
    return ix


def ordered(capi, ix, q, n, init_k=10):
    s = capi.Scan(ix, init_k=init_k)
    s.rescan(q)
    out = s.fetch(n)
    s.end()
    return out


def test_golden_small_world_on_device(capi, golden):
    g, sw = golden["dist_func"], golden["small_world"]
    dist = {"l2sq": capi.l2sq_dist, "cos": capi.cos_dist, "hamming": capi.hamming_dist}
    for metric, key in (("l2sq", "l2sq_sorted"), ("cos", "cos_sorted_2dp"), ("hamming", "hamming_sorted")):
        ix = gpu_build(capi, metric, sw["v"])
        order = ordered(capi, ix, g["query"], 8)
        assert sorted(order) == list(range(LABEL0, LABEL0 + 8))
        assert [round(dist[metric](sw["v"][l - LABEL0], g["query"]), 2) for l in order] == g[key]
    g4 = golden["four_nn_of_each_corner"]
    ix = gpu_build(capi, "l2sq", sw["v"])
    for ident, v in zip(sw["ids"], sw["v"]):
        labels, dists = ix.search(v, 4)
        got = [sw["ids"][int(l) - LABEL0] for l in labels]
        assert got[0] == ident and sorted(got) == sorted(g4["rows"][ident]) and list(dists) == g4["dists"]


def test_golden_streaming_and_pagination_on_device(capi, golden):
    g, sw = golden["streaming"], golden["small_world"]
    ix = gpu_build(capi, "l2sq", sw["v"] + [[99, 99, 2]], M=5, efc=20, ef=20)
    assert len(ordered(capi, ix, g["query"], 3, init_k=g["init_k"])) == g["limit_3_count"]
    got = ordered(capi, ix, g["query"], 15, init_k=g["init_k"])
    assert len(got) == g["limit_15_count"] and len(set(got)) == len(got)
    p = golden["pagination_duplicates"]
    rows = [[np.float32(i) / np.float32(10)] * p["dim"] for i in p["ramp_ids"]] + [[p["dup_value"]] * p["dim"]] * p["dup_count"]
    ix = capi.GpuIndex("l2sq", p["dim"], seed=7)
    ix.set_add_batch(1, 1)  # the reference inserts these rows one by one
    ids = [i + 1 for i in p["ramp_ids"]] + [p["dup_first_id"] + j + 1 for j in range(p["dup_count"])]
    ix.add_many(ids, np.asarray(rows, dtype=np.float32))
    got = ordered(capi, ix, [p["dup_value"]] * p["dim"], p["limit"], init_k=p["init_k"])
    assert len(got) == p["limit"] and len(set(got)) == len(got)


def test_golden_scan_k_trace_expression_index_and_unlogged_insert_on_device(capi, golden):
    """The reference's own trace of usearch_search_ef calls per scan (hnsw_select.out:76-140: 10 | 10 | 4 | 4, 8, 8 and the counts
    3 / 8) asserted on the C++ shim (csrc/scan_shim.cpp vs scan.c:167-338); hnsw_create_expr.out:90-94; hnsw_insert_unlogged.out:62-92."""
    g, sw = golden["scan_k_trace"], golden["small_world"]
    ix = gpu_build(capi, "l2sq", sw["v"], M=g["index"]["M"], efc=g["index"]["ef_construction"], ef=g["index"]["ef"])
    for c in g["cases"]:
        s = capi.Scan(ix, init_k=c["init_k"])
        s.rescan(g["query"])
        got = s.fetch(c["limit"])
        assert len(got) == c["count"] and len(set(got)) == len(got), c
        assert s.trace() == c["k_trace"], c
        s.rescan(g["query"])  # ldb_amrescan starts a new trace
        assert s.trace() == []
        s.end()
    e = golden["create_expr"]
    ix = capi.GpuIndex("l2sq", 3, M=e["M"], seed=7)
    ix.add_many([i + LABEL0 for i in e["ids"]], np.asarray(e["v"], dtype=np.float32))
    assert [l - LABEL0 for l in ordered(capi, ix, e["query"], e["limit"])] == e["expect_ids"]
    u = golden["insert_unlogged"]
    ix = gpu_build(capi, "l2sq", sw["v"])
    ix.add(100, u["inserted"])
    with pytest.raises(capi.LanternGpuError, match=u["wrong_dim_error"]):
        ix.add(101, u["wrong_dim_row"])
    rows = sw["v"] + [u["inserted"]]
    order = ordered(capi, ix, u["query"], 20)
    assert [round(capi.l2sq_dist(rows[8 if l == 100 else l - LABEL0], u["query"]), 2) for l in order] == u["sorted_2dp"]


def test_device_resident_queries_state_their_stride(capi):
    """Bit rows of 96 bytes are stored at a 128-byte stride: device-resident queries laid out at 96 would be read at the wrong
    offsets and past the end of the caller's buffer.  The stride is part of the call and a mismatch is refused; the entry point
    without a stride argument refuses such an index outright (include/lantern_gpu.h)."""
    from lantern_amd import hip

    rng = np.random.default_rng(3)
    n, words, nq, k = 3000, 24, 32, 10
    base = rng.integers(0, 2**32, size=(n, words), dtype=np.uint32)
    queries = rng.integers(0, 2**32, size=(nq, words), dtype=np.uint32)
    ix = capi.GpuIndex("hamming", words, M=8, ef_construction=32, ef=32, seed=2)
    ix.set_add_batch(256, 16)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    assert ix.row_bytes() == 128
    want_lab, want_dist, _ = ix.search_batch(queries, k)
    rows = ix.device_query_rows(queries)
    assert rows.strides[0] == 128 and not rows[:, words:].any()
    dq = hip.Buffer.from_numpy(rows)
    lab, dist = hip.Buffer(nq * k * 8), hip.Buffer(nq * k * 4)
    ix.search_batch_device(dq.ptr, nq, k, 0, 0, lab.ptr, dist.ptr, query_stride=rows.strides[0])
    hip.synchronize()
    assert np.array_equal(lab.download((nq, k), np.uint64), want_lab) and np.array_equal(dist.download((nq, k), np.float32), want_dist)
    with pytest.raises(capi.LanternGpuError, match="query row stride does not match"):
        ix.search_batch_device(dq.ptr, nq, k, 0, 0, lab.ptr, dist.ptr, query_stride=96)
    with pytest.raises(capi.LanternGpuError, match="lantern_gpu_search_batch_device_strided"):
        ix.search_batch_device(dq.ptr, nq, k, 0, 0, lab.ptr, dist.ptr)
    # an index whose rows sit at their own length accepts both forms, and still checks a stated stride
    f = capi.GpuIndex("l2sq", 20, M=8, ef_construction=32, ef=32, seed=2)
    fb = rng.standard_normal((500, 20), dtype=np.float32)
    f.add_many(np.arange(500, dtype=np.uint64) + 1, fb)
    fr = f.device_query_rows(fb[:8])
    assert fr.strides[0] == f.row_bytes() == 80
    fq = hip.Buffer.from_numpy(fr)
    l1, l2 = hip.Buffer(8 * k * 8), hip.Buffer(8 * k * 8)
    f.search_batch_device(fq.ptr, 8, k, 0, 0, l1.ptr)
    f.search_batch_device(fq.ptr, 8, k, 0, 0, l2.ptr, query_stride=80)
    hip.synchronize()
    assert np.array_equal(l1.download((8, k), np.uint64), l2.download((8, k), np.uint64))
    with pytest.raises(capi.LanternGpuError, match="query row stride does not match"):
        f.search_batch_device(fq.ptr, 8, k, 0, 0, l2.ptr, query_stride=96)


def test_golden_exact_order_with_ef_construction_below_m_on_device(capi, golden):
    """hnsw_logged_unlogged.out:35-275 on the device: ef_construction = 2 < M = 14, ids in the pinned order after a build over 8 / 9 / 10
    rows and after every later insert (usearch_add one at a time, as aminsert does)."""
    g = golden["logged_unlogged"]
    ids = g["ids"] + [e["id"] for e in g["inserted"]]
    rows = g["v"] + [e["v"] for e in g["inserted"]]
    for built in (8, 9, 10):
        for total in range(built, 11):
            ix = capi.GpuIndex("l2sq", 4, M=g["index"]["M"], ef_construction=g["index"]["ef_construction"], ef=g["index"]["ef"], seed=7)
            ix.set_add_batch(1, 1)  # CREATE INDEX adds one tuple at a time (build.c:83-135)
            ix.add_many(np.arange(built, dtype=np.uint64) + LABEL0, np.asarray(rows[:built], dtype=np.float32))
            for i in range(built, total):
                ix.add(i + LABEL0, rows[i])
            got = ordered(capi, ix, g["query"], g["limit"])
            assert [[ids[l - LABEL0], int(capi.l2sq_dist(rows[l - LABEL0], g["query"]))] for l in got] == g[f"order_{total}"], (built, total)


def test_golden_insert_dimension_errors_and_misc(capi, golden):
    sw = golden["small_world"]
    ix = gpu_build(capi, "l2sq", sw["v"])
    ix.add(100, golden["insert_then_search"]["inserted"])
    order = ordered(capi, ix, [0, 0, 0], 9)
    rows = sw["v"] + [golden["insert_then_search"]["inserted"]]
    assert [capi.l2sq_dist(rows[8 if l == 100 else l - LABEL0], [0, 0, 0]) for l in order] == golden["insert_then_search"]["sorted"]
    with pytest.raises(capi.LanternGpuError, match="Wrong number of dimensions: 4 instead of 3 expected"):
        ix.add(101, [4, 4, 4, 4])
    with pytest.raises(capi.LanternGpuError, match=golden["dimension_errors"]["sized_array"]):
        ix.search([0, 1, 0, 1], 1)
    # empty index, k larger than the index, label 0 (deleted) is skipped by the scan
    e = capi.GpuIndex("l2sq", 3)
    assert len(e.search([0, 0, 0], 5)[0]) == 0
    d = capi.GpuIndex("l2sq", 3)
    d.add_many([5, 0, 7], [[0, 0, 0], [0, 0, 1], [0, 0, 2]])
    assert ordered(capi, d, [0, 0, 0], 10) == [5, 7]
    m = ix.metadata()
    assert m.neighbors_bytes == 4 + 16 * 6 and m.neighbors_base_bytes == 4 + 32 * 6 and m.connectivity == 16


def test_file_round_trip(capi, oracle, tmp_path, monkeypatch):
    rng = np.random.default_rng(8)
    for metric, d in (("l2sq", 24), ("hamming", 3)):
        base = rand_rows(rng, 500, d, metric)
        ix = capi.GpuIndex(metric, d, M=4, ef_construction=24, seed=2)
        ix.add_many(np.arange(500) + 1, base)
        blob = ix.save_buffer()
        assert ix.save_stream() == blob  # the span stream of the indexing server: the same bytes
        monkeypatch.setenv("LANTERN_GPU_SAVE_CHUNK_BYTES", "5000")  # rows leave the device in ~10 chunks through the two staging buffers
        assert ix.save_stream() == blob and ix.save_buffer() == blob
        monkeypatch.delenv("LANTERN_GPU_SAVE_CHUNK_BYTES")
        g = ix.export_graph()
        # layout: 136-byte header + node tapes: 8+2+(4+2M*6)+level*(4+M*6)+vector (usearch_storage.cpp:19-32)
        vb = d * 4
        assert len(blob) == 136 + sum(10 + (4 + 8 * 6) + int(l) * (4 + 4 * 6) + vb for l in g["levels"])
        assert blob[:7] == b"usearch"
        other = capi.GpuIndex(metric, d, M=4, ef_construction=24, seed=2)
        other.load_buffer(blob)
        g2 = other.export_graph(with_vectors=True)
        for key in ("levels", "nbr0", "upper_off", "upper_nbr", "labels"):
            assert np.array_equal(g[key], g2[key]), key
        assert np.array_equal(g2["vectors"], base) and g2["entry_slot"] == g["entry_slot"]
        path = str(tmp_path / f"{metric}.usearch")
        ix.save(path)
        third = capi.GpuIndex(metric, d, M=4, ef_construction=24, seed=2)
        third.load(path)
        q = rand_rows(rng, 1, d, metric)[0]
        assert np.array_equal(third.search(q, 5)[0], ix.search(q, 5)[0])
        # first node tape: label, level
        assert int.from_bytes(blob[136:144], "little") == 1 and int.from_bytes(blob[144:146], "little") == int(g["levels"][0])
    empty = capi.GpuIndex("l2sq", 8, M=4, ef_construction=8, seed=1)
    assert len(empty.save_buffer()) == 136 and empty.save_stream() == empty.save_buffer()  # an empty index is its header


# ------------------------------------------------------------------------------------------------
# the dense contraction (fp32 MFMA) and the exact k-NN built on it
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("metric,n,d", [("l2sq", 5000, 96), ("cos", 3000, 768), ("l2sq", 70000, 20), ("hamming", 4000, 24),
                                        ("cos", 300, 130)])
def test_exact_search_matches_bruteforce(capi, oracle, metric, n, d):
    rng = np.random.default_rng(n + d)
    base, queries = rand_rows(rng, n, d, metric), rand_rows(rng, 70, d, metric)
    ix = capi.GpuIndex(metric, d, M=4, ef_construction=8, seed=1)
    g = {"levels": np.zeros(n, np.uint8), "nbr0": np.full((n, 8), 0xFFFFFFFF, np.uint32), "upper_off": np.full(n, 0xFFFFFFFF, np.uint32),
         "upper_nbr": np.zeros((0, 4), np.uint32), "labels": None, "entry_slot": 0, "max_level": 0}
    ix.import_graph(base, g)  # vectors only: exact search never touches the graph
    slots, dists = ix.exact_search(queries, 10)
    t_ids, t_d = oracle.bruteforce(base, queries, 10, metric, oracle.SUM_WAVE64, 8)
    assert np.array_equal(slots, t_ids)
    assert np.array_equal(dists, t_d)


@pytest.mark.parametrize("metric,n,d,k", [("l2sq", 30000, 64, 10), ("cos", 70000, 40, 5), ("l2sq", 9000, 200, 100)])
def test_exact_search_fused_epilogue_equals_the_matrix_path(capi, metric, n, d, k, monkeypatch):
    # the contraction's fused top-k epilogue (candidate lists instead of a distance matrix) against the unfused path
    rng = np.random.default_rng(n)
    base, queries = rng.standard_normal((n, d), dtype=np.float32), rng.standard_normal((300, d), dtype=np.float32)
    ix = capi.GpuIndex(metric, d, M=4, ef_construction=8, seed=1)
    g = {"levels": np.zeros(n, np.uint8), "nbr0": np.full((n, 8), 0xFFFFFFFF, np.uint32), "upper_off": np.full(n, 0xFFFFFFFF, np.uint32),
         "upper_nbr": np.zeros((0, 4), np.uint32), "labels": None, "entry_slot": 0, "max_level": 0}
    ix.import_graph(base, g)  # rows only: the exact search does not look at the graph
    monkeypatch.setenv("LANTERN_GPU_DENSE_FUSED", "0")
    want = ix.exact_search(queries, k)
    monkeypatch.setenv("LANTERN_GPU_DENSE_FUSED", "1")
    got = ix.exact_search(queries, k)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_exact_search_survives_adversarially_ordered_rows(capi):
    # rows ordered so that EVERY later row is closer to the queries than all earlier ones: every column passes the fused
    # epilogue's radius test, the candidate lists overflow, and the search must notice and repeat itself the ordinary way
    n, d, k = 20000, 16, 10
    t = np.linspace(100.0, 1.0, n, dtype=np.float32)
    base = np.zeros((n, d), dtype=np.float32)
    base[:, 0] = t
    base[:, 1] = np.random.default_rng(0).standard_normal(n).astype(np.float32) * 1e-3
    queries = np.zeros((5, d), dtype=np.float32)
    queries[:, 0] = np.array([0.0, 0.5, 0.9, -3.0, 0.99], dtype=np.float32)
    ix = capi.GpuIndex("l2sq", d, M=4, ef_construction=8, seed=1)
    g = {"levels": np.zeros(n, np.uint8), "nbr0": np.full((n, 8), 0xFFFFFFFF, np.uint32), "upper_off": np.full(n, 0xFFFFFFFF, np.uint32),
         "upper_nbr": np.zeros((0, 4), np.uint32), "labels": None, "entry_slot": 0, "max_level": 0}
    ix.import_graph(base, g)
    slots, dists = ix.exact_search(queries, k)
    ref = ((base[None, :, :].astype(np.float64) - queries[:, None, :]) ** 2).sum(2)
    for qi in range(5):
        order = np.lexsort((np.arange(n), ref[qi]))[:k]
        assert set(slots[qi].tolist()) == set(order.tolist())


# (na, nb): ragged last tiles; 300 x 40000 = 939 tiles on the kernel's persistent grid of 512 workgroups (two tiles per workgroup, the
# K steps of both in one pipeline); d = 36 / 50: a K tail; d = 7, 20: a single K step per tile; 1 x 1: one workgroup, one row each
@pytest.mark.parametrize("metric,d,na,nb", [("l2sq", 768, 130, 257), ("cos", 768, 130, 257), ("l2sq", 50, 130, 257), ("cos", 1536, 130, 257),
                                            ("cos", 36, 300, 40000), ("l2sq", 20, 129, 70000), ("l2sq", 7, 8, 6), ("cos", 33, 257, 130)])
def test_mfma_distance_matrix_within_tolerance(capi, oracle, metric, d, na, nb):
    rng = np.random.default_rng(d)
    A, B = rng.standard_normal((na, d), dtype=np.float32), rng.standard_normal((nb, d), dtype=np.float32)
    A[3] = 0  # zero-norm rows exercise the cosine rules in the epilogue
    B[5] = 0
    got = capi.distance_matrix(A, B, metric, exact_order=False)
    ref = capi.distance_matrix(A, B, metric, exact_order=True)
    assert np.all(np.abs(got - ref) <= TOL * np.maximum(1.0, np.abs(ref)))
    if metric == "cos":
        assert got[3, 5] == 0.0 and got[3, 0] == 1.0 and got[0, 5] == 1.0


# ------------------------------------------------------------------------------------------------
# the PostgreSQL-page view: node tapes with 6-byte ItemPointer slots behind a retriever callback
# ------------------------------------------------------------------------------------------------
def test_mirror_through_retriever_matches_direct_index(capi):
    import ctypes as C
    import struct

    rng = np.random.default_rng(12)
    n, d, M = 1500, 48, 6
    base = rng.standard_normal((n, d), dtype=np.float32)
    a = capi.GpuIndex("l2sq", d, M=M, ef_construction=40, ef=32, seed=3)
    a.add_many(np.arange(n, dtype=np.uint64) + 1000, base)
    blob = bytearray(a.save_buffer())
    # StoreExternalIndex (external_index.c:298-418): node i goes to some (block, offset); every neighbour slot
    # (u32 seq id in the low 4 of 6 bytes) and the header's entry slot are rewritten to that ItemPointer
    def item_pointer(i):  # ItemPointerData{bi_hi u16, bi_lo u16, posid u16}, as the low 48 bits of a u64
        block, pos = 1 + i // 40, 1 + i % 40
        return struct.pack("<HHH", block >> 16, block & 0xFFFF, pos)

    tapes, off = [], 136
    for i in range(n):
        level = struct.unpack_from("<H", blob, off + 8)[0]
        size = 10 + (4 + 2 * M * 6) + level * (4 + M * 6) + d * 4
        tapes.append(bytearray(blob[off:off + size]))
        off += size
    assert off == len(blob)
    for t in tapes:
        level = struct.unpack_from("<H", t, 8)[0]
        q = 10
        for l in range(level + 1):
            cap = 2 * M if l == 0 else M
            cnt = struct.unpack_from("<I", t, q)[0]
            for j in range(cnt):
                seq = struct.unpack_from("<I", t, q + 4 + j * 6)[0]
                t[q + 4 + j * 6:q + 10 + j * 6] = item_pointer(seq)
            q += 4 + cap * 6
    header = bytearray(blob[:136])
    hbuf = C.create_string_buffer(bytes(header), 136)
    entry_seq = capi.lib().usearch_header_get_entry_slot(hbuf)
    capi.lib().usearch_header_set_entry_slot(hbuf, int.from_bytes(item_pointer(entry_seq), "little"))
    pages = {int.from_bytes(item_pointer(i), "little"): C.create_string_buffer(bytes(t), len(t)) for i, t in enumerate(tapes)}
    calls = []

    def retriever(slot):
        calls.append(slot)
        return C.addressof(pages[slot])

    b = capi.GpuIndex("l2sq", d, M=M, ef_construction=40, ef=32, seed=3, retriever=retriever)
    b.view_mem_lazy(hbuf.raw)
    assert len(b) == n and len(calls) == b.graph_info().size <= n  # usearch_size = the header's count; every reachable node fetched once
    queries = rng.standard_normal((40, d), dtype=np.float32)
    la, da, _ = a.search_batch(queries, 10)
    lb, db, _ = b.search_batch(queries, 10)
    assert np.array_equal(la, lb) and np.array_equal(da, db)
    # the scan shim works on the mirror too
    sa = capi.Scan(a, init_k=4)
    sa.rescan(queries[0])
    sb = capi.Scan(b, init_k=4)
    sb.rescan(queries[0])
    assert sa.fetch(25) == sb.fetch(25)


@pytest.mark.parametrize("metric", ["l2sq", "cos"])
def test_assign_to_clusters_matches_reference_loop(capi, oracle, metric):
    # product_quantization.c:80-124: argmin_j usearch_distance(subvector, center_j), first minimum wins
    rng = np.random.default_rng(21)
    data = rng.standard_normal((5000, 48), dtype=np.float32)
    start, sdim, k = 16, 12, 256
    centers = data[rng.choice(5000, k, replace=False), start:start + sdim].copy()
    centers[7] = centers[3]  # an exact tie: the lower index must win
    idx, dist = capi.assign_to_clusters(data, centers, metric, start, sdim)
    sub = np.ascontiguousarray(data[:, start:start + sdim])
    for i in range(0, 5000, 97):
        d = np.array([oracle.distance(sub[i], c, metric, oracle.SUM_WAVE64) for c in centers], dtype=np.float32)
        assert idx[i] == int(np.argmin(d)) and dist[i] == d.min()
    assert not np.any(idx == 7)


# ------------------------------------------------------------------------------------------------
# f16 storage (reloption quant_bits = 16, options.c:137-158): vectors and queries arrive as f32, are cast to
# f16 (round-to-nearest-even), and every distance is the f32 arithmetic on the rounded values
# (usearch metric_*_gt<f16_t, f32>).  The oracle gets the rounded values and the 8-scalars-per-chunk order.
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("metric,n,d,M,efc", [("l2sq", 1500, 768, 16, 64), ("cos", 900, 200, 8, 40), ("l2sq", 600, 33, 4, 24)])
def test_f16_storage_matches_oracle(capi, oracle, metric, n, d, M, efc):
    rng = np.random.default_rng(n + d)
    base = rng.standard_normal((n, d), dtype=np.float32)
    queries = rng.standard_normal((48, d), dtype=np.float32)
    rb, rq = oracle.round_f16(base), oracle.round_f16(queries)
    labels = np.arange(n, dtype=np.uint64) + 1
    # build: device (f32 in, cast on add) vs oracle on the rounded values, same batch plan
    ora = oracle.OracleIndex(metric, d, M=M, ef_construction=efc, ef=48, seed=5, sum_mode=oracle.SUM_WAVE64_F16)
    ora.add_planned(labels, rb, max_batch=256, min_ratio=8)
    gpu = capi.GpuIndex(metric, d, M=M, ef_construction=efc, ef=48, seed=5, quantization="f16")
    gpu.set_add_batch(256, 8)
    gpu.add_many(labels, base)
    go, gg = ora.export_graph(), gpu.export_graph(with_vectors=True)
    assert np.array_equal(gg["nbr0"], go["nbr0"]) and np.array_equal(gg["upper_nbr"], go["upper_nbr"])
    assert np.array_equal(gg["vectors"], base.astype(np.float16))
    # search: ids, distances and counters identical
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(rq, 10)
    lab, dist, cnt = gpu.search_batch(queries, 10)
    assert np.array_equal(lab, o_lab) and np.array_equal(dist, o_dist)
    # the distance kernel alone, and the exact k-NN (MFMA contraction on the dequantised rows + exact re-rank)
    slots = rng.integers(0, n, 40).astype(np.uint32)
    ref = np.array([oracle.distance(rq[0], rb[s], metric, oracle.SUM_WAVE64_F16) for s in slots], dtype=np.float32)
    assert np.array_equal(gpu.distance_gather(queries[0], slots), ref)
    e_slots, e_dists = gpu.exact_search(queries, 10)
    t_ids, t_d = oracle.bruteforce(rb, rq, 10, metric, oracle.SUM_WAVE64_F16, 8)
    assert np.array_equal(e_slots, t_ids) and np.array_equal(e_dists, t_d)
    # against the f32 index the f16 one is a quantisation, not a different algorithm: distances agree to f16 precision
    full = oracle.OracleIndex.from_graph(metric, base, go, M, efc, 48, 5, oracle.SUM_SEQ)
    _, f_dist, _, _, _ = full.search_batch(queries, 10)
    assert np.median(np.abs(f_dist - o_dist) / np.maximum(np.abs(f_dist), 1e-6)) < 5e-3
    # file: the tape holds d * 2 vector bytes
    blob = gpu.save_buffer()
    assert len(blob) == 136 + sum(10 + (4 + 2 * M * 6) + int(l) * (4 + M * 6) + d * 2 for l in gg["levels"])
    other = capi.GpuIndex(metric, d, M=M, ef_construction=efc, ef=48, seed=5, quantization="f16")
    other.load_buffer(blob)
    assert np.array_equal(other.search_batch(queries, 10)[0], lab)


# i8 storage (reloption quant_bits = 8, options.c:137-158): vectors and queries arrive as f32 and are quantised to
# trunc(clamp(x * 100, -100, 100)) (hnsw_sq.sql:33-34); distances are usearch's l2sq_i8_t / cos_i8_t -- int32
# accumulation, integer-exact.  The oracle gets the quantised integers (SUM_I8); everything must match bit for bit.
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("metric,n,d,M,efc", [("l2sq", 1500, 768, 16, 64), ("cos", 900, 200, 8, 40), ("l2sq", 600, 33, 4, 24),
                                              ("cos", 700, 2000, 16, 48)])
def test_i8_storage_matches_oracle(capi, oracle, metric, n, d, M, efc):
    rng = np.random.default_rng(n + d + 8)
    base = (rng.standard_normal((n, d), dtype=np.float32) * np.float32(0.4)).astype(np.float32)  # some values clamp at +-1
    queries = (rng.standard_normal((48, d), dtype=np.float32) * np.float32(0.4)).astype(np.float32)
    base[0, :3] = [np.nan, 5.0, -7.0]  # NaN -> 0, out-of-range values clamp
    qb, qq = oracle.quantize_i8(base), oracle.quantize_i8(queries)
    assert qb[0, 0] == 0 and qb[0, 1] == 100 and qb[0, 2] == -100
    labels = np.arange(n, dtype=np.uint64) + 1
    ora = oracle.OracleIndex(metric, d, M=M, ef_construction=efc, ef=48, seed=5, sum_mode=oracle.SUM_I8)
    ora.add_planned(labels, qb, max_batch=256, min_ratio=8)
    gpu = capi.GpuIndex(metric, d, M=M, ef_construction=efc, ef=48, seed=5, quantization="i8")
    gpu.set_add_batch(256, 8)
    gpu.add_many(labels, base)
    go, gg = ora.export_graph(), gpu.export_graph(with_vectors=True)
    assert np.array_equal(gg["vectors"], qb.astype(np.int8)), "the stored bytes are the quantisation rule's"
    assert np.array_equal(gg["nbr0"], go["nbr0"]) and np.array_equal(gg["upper_nbr"], go["upper_nbr"])
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(qq, 10)
    lab, dist, cnt = gpu.search_batch(queries, 10)
    assert np.array_equal(lab, o_lab) and np.array_equal(dist, o_dist)
    slots = rng.integers(0, n, 40).astype(np.uint32)
    ref = np.array([oracle.distance(qq[0], qb[s], metric, oracle.SUM_I8) for s in slots], dtype=np.float32)
    assert np.array_equal(gpu.distance_gather(queries[0], slots), ref)
    if metric == "l2sq":  # integer-valued distances
        assert np.array_equal(dist, np.round(dist))
    e_slots, e_dists = gpu.exact_search(queries, 10)
    t_ids, t_d = oracle.bruteforce(qb, qq, 10, metric, oracle.SUM_I8, 8)
    assert np.array_equal(e_slots, t_ids) and np.array_equal(e_dists, t_d)
    # device-resident queries in storage format (what bench.py hands over) give the same answers
    from lantern_amd import hip

    dq = hip.Buffer.from_numpy(hip.padded_rows(queries, False, i8=True))
    d_slot = hip.Buffer(48 * 10 * 4)
    gpu.search_batch_device(dq.ptr, 48, 10, 0, 0, None, None, d_slot.ptr)
    hip.synchronize()
    assert np.array_equal(d_slot.download((48, 10), np.uint32), o_slot)
    # file: the tape holds d vector bytes
    blob = gpu.save_buffer()
    assert len(blob) == 136 + sum(10 + (4 + 2 * M * 6) + int(l) * (4 + M * 6) + d for l in gg["levels"])
    other = capi.GpuIndex(metric, d, M=M, ef_construction=efc, ef=48, seed=5, quantization="i8")
    other.load_buffer(blob)
    assert np.array_equal(other.search_batch(queries, 10)[0], lab)


@pytest.mark.parametrize("metric,d", [("l2sq", 768), ("cos", 768), ("l2sq", 128), ("hamming", 24), ("cos", 100)])
def test_gather_in_the_walks_launch_shape_has_the_same_bits(capi, metric, d, monkeypatch):
    """bench.py gather_ceiling() measures the random-row fetch rate of the box with the distance phase of a hop on its own
    (persistent four-wave workgroups, two rows in flight per group, the blocked row loads): same chains and trees as the plain
    gather, and lantern_gpu_last_gather_ms reports that launch's kernel time."""
    rng = np.random.default_rng(d)
    base = rand_rows(rng, 3000, d, metric)
    ix = capi.GpuIndex(metric, d, M=4, ef_construction=16, ef=16, seed=1)
    ix.add_many(np.arange(3000, dtype=np.uint64) + 1, base)
    q = rand_rows(rng, 1, d, metric)[0]
    slots = rng.integers(0, 3000, 20_001).astype(np.uint32)
    plain = ix.distance_gather(q, slots)
    monkeypatch.setenv("LANTERN_GPU_GATHER_WALKSHAPE", "1")
    assert ix.last_gather_ms() > 0.0
    assert np.array_equal(ix.distance_gather(q, slots), plain)
    assert 0.0 < ix.last_gather_ms() < 50.0


@pytest.mark.parametrize("metric", ["l2sq", "cos"])
def test_i8_rows_wider_than_2032_dims_in_a_small_batch(capi, oracle, metric):
    """Batches of 64 .. 4 x CUs queries take the small-batch launch shape; i8 rows of >= 128 chunks (>= 2033 dims: usearch_init
    does not cap dimensions) have no four-row instantiation and must fall through to the two-row kernel, not fail."""
    rng = np.random.default_rng(2100)
    n, d, M = 700, 2100, 8
    base = (rng.standard_normal((n, d), dtype=np.float32) * np.float32(0.4)).astype(np.float32)
    labels = np.arange(n, dtype=np.uint64) + 1
    gpu = capi.GpuIndex(metric, d, M=M, ef_construction=40, ef=40, seed=5, quantization="i8")
    gpu.set_add_batch(256, 8)
    gpu.add_many(labels, base)
    qb = oracle.quantize_i8(base)
    ora = oracle.OracleIndex.from_graph(metric, qb, gpu.export_graph(), M, 40, 40, 5, oracle.SUM_I8)
    for nq in (63, 64, 200, 1024):
        queries = (rng.standard_normal((nq, d), dtype=np.float32) * np.float32(0.4)).astype(np.float32)
        o_lab, o_dist, _, _, _ = ora.search_batch(oracle.quantize_i8(queries), 10, 40, 4)
        lab, dist, _ = gpu.search_batch(queries, 10)
        assert np.array_equal(lab, o_lab) and np.array_equal(dist, o_dist), nq


# ------------------------------------------------------------------------------------------------
# limits of the reloptions / GUCs (options.c:165-179,324-348; build.c:394-401) and concurrency
# ------------------------------------------------------------------------------------------------
def test_maximum_dimension_connectivity_and_ef(capi, oracle):
    rng = np.random.default_rng(77)
    # dim = 2000 is the largest row Lantern accepts (one node per 8 KB page); M = 128, ef_construction = ef = 400 are the caps
    base = rng.standard_normal((260, 2000), dtype=np.float32)
    labels = np.arange(260, dtype=np.uint64) + 1
    ora = oracle.OracleIndex("l2sq", 2000, M=128, ef_construction=400, ef=400, seed=2, sum_mode=oracle.SUM_WAVE64)
    ora.add_planned(labels, base, max_batch=64, min_ratio=4)
    gpu = capi.GpuIndex("l2sq", 2000, M=128, ef_construction=400, ef=400, seed=2)
    gpu.set_add_batch(64, 4)
    gpu.add_many(labels, base)
    assert np.array_equal(gpu.export_graph()["nbr0"], ora.export_graph()["nbr0"])
    q = rng.standard_normal((8, 2000), dtype=np.float32)
    o_lab, o_dist, _, _, _ = ora.search_batch(q, 200)
    lab, dist, cnt = gpu.search_batch(q, 200)
    assert np.array_equal(lab, o_lab) and np.array_equal(dist, o_dist) and np.all(cnt == 200)
    m = gpu.metadata()
    assert m.neighbors_bytes == 4 + 128 * 6 and m.neighbors_base_bytes == 4 + 256 * 6
    # the node tape of such a row still fits an 8 KB page only at M small; here we only check the size formula
    assert len(gpu.save_buffer()) == 136 + sum(10 + (4 + 256 * 6) + int(l) * (4 + 128 * 6) + 8000 for l in gpu.export_graph()["levels"])


def test_scan_stops_at_1000_rows_and_never_repeats(capi):
    rng = np.random.default_rng(78)
    base = rng.standard_normal((1800, 16), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", 16, M=16, ef_construction=64, ef=64, seed=1)
    ix.add_many(np.arange(1800, dtype=np.uint64) + 1, base)
    q = rng.standard_normal(16, dtype=np.float32)
    for init_k in (1000, 10, 3):
        s = capi.Scan(ix, init_k=init_k)
        s.rescan(q)
        got = s.fetch(5000)
        assert len(set(got)) == len(got)
        first_page = np.array([capi.l2sq_dist(base[l - 1], q) for l in got[:init_k]])
        assert np.all(np.diff(first_page) >= 0)  # within a page the order is the index order
        if init_k == 1000:
            assert len(got) == 1000  # scan.c:249-252: no continuation once 1000 rows were loaded
        else:
            assert 1000 <= len(got) <= 1800  # the doubling continuation passes 1000 and is then cut off
    with pytest.raises(capi.LanternGpuError, match="init_k"):
        capi.Scan(ix, init_k=1001)  # GUC range 1..1000 (options.c:324-336)


def test_concurrent_add_raw_from_many_threads(capi):
    # the external indexer calls add_raw on ONE index from N threads (server.rs:333-356)
    import threading

    rng = np.random.default_rng(79)
    base = rng.standard_normal((4000, 24), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", 24, M=8, ef_construction=32, ef=32, seed=1)
    ix.set_add_batch(512, 16)
    errors = []

    def worker(t):
        try:
            for i in range(t, 4000, 8):
                ix.add(i + 1, base[i])
        except Exception as e:  # noqa
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    assert len(ix) == 4000
    g = ix.export_graph(with_vectors=True)
    assert sorted(g["labels"].tolist()) == list(range(1, 4001))
    assert np.array_equal(g["vectors"], base[g["labels"].astype(np.int64) - 1])  # every row kept its own label
    hits = sum(int(ix.search(base[i], 1)[0][0]) == i + 1 for i in range(0, 4000, 40))
    assert hits >= 90  # M=8, ef=32: an ANN index, not an exact one


@pytest.mark.parametrize("vis_slots", ["0", "256", "1024"])
def test_visited_set_variants_give_identical_walks(capi, oracle, vis_slots, monkeypatch):
    # the LDS visited set spills to the HBM bitmap when it fills up: force "bitmap only" (0), "spills after a few hops"
    # (256) and "spills late" (1024) and require the same ids, distances and counters every time
    rng = np.random.default_rng(5)
    base, queries = rng.standard_normal((4000, 64), dtype=np.float32), rng.standard_normal((64, 64), dtype=np.float32)
    ora = oracle.OracleIndex("l2sq", 64, M=16, ef_construction=64, ef=128, seed=9, sum_mode=oracle.SUM_WAVE64)
    ora.add_many(np.arange(4000, dtype=np.uint64) + 1, base)
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(queries, 10)
    monkeypatch.setenv("LANTERN_GPU_VIS_SLOTS", vis_slots)
    gpu = capi.GpuIndex("l2sq", 64, M=16, ef_construction=64, ef=128, seed=9)
    gpu.import_graph(base, ora.export_graph())
    from lantern_amd import hip

    dq = hip.Buffer.from_numpy(hip.padded_rows(queries, False))
    lab, dist, D, E = hip.Buffer(64 * 10 * 8), hip.Buffer(64 * 10 * 4), hip.Buffer(64 * 8), hip.Buffer(64 * 8)
    gpu.search_batch_device(dq.ptr, 64, 10, 0, 0, lab.ptr, dist.ptr, None, None, D.ptr, E.ptr)
    hip.synchronize()
    assert np.array_equal(lab.download((64, 10), np.uint64), o_lab) and np.array_equal(dist.download((64, 10), np.float32), o_dist)
    assert np.array_equal(D.download(64, np.uint64), o_D) and np.array_equal(E.download(64, np.uint64), o_E)
    assert o_D.max() > 300  # more visits than 3/4 of 256 slots: the spill path really ran


@pytest.mark.parametrize("lds_list", ["0", "1"])
@pytest.mark.parametrize("ef", [1, 7, 63, 64, 65, 127, 128, 129, 200])
def test_candidate_list_in_registers_or_lds_gives_the_oracle_walk(capi, oracle, ef, lds_list, monkeypatch):
    # walk.hpp keeps the candidate list in wave 0's registers for ef <= 128 (one or two keys per lane) and in LDS above;
    # LANTERN_GPU_LDS_LIST=1 forces the LDS form everywhere.  Every (ef, placement) must be the oracle's walk: same ids,
    # distances and evaluation/expansion counts.  The lane/register boundaries (63..65, 127..129) are the point.
    rng = np.random.default_rng(77)
    n, d, k = 3000, 48, 10
    base, queries = rng.standard_normal((n, d), dtype=np.float32), rng.standard_normal((64, d), dtype=np.float32)
    base[1000:1400] = base[:400]  # exact duplicates: equal distances, the slot decides the order
    ora = oracle.OracleIndex("l2sq", d, M=16, ef_construction=64, ef=ef, seed=9, sum_mode=oracle.SUM_WAVE64)
    ora.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    kk = min(k, ef)
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(queries, kk)
    monkeypatch.setenv("LANTERN_GPU_LDS_LIST", lds_list)
    gpu = capi.GpuIndex("l2sq", d, M=16, ef_construction=64, ef=ef, seed=9)
    gpu.import_graph(base, ora.export_graph())
    from lantern_amd import hip

    dq = hip.Buffer.from_numpy(hip.padded_rows(queries, False))
    lab, dist, D, E = hip.Buffer(64 * kk * 8), hip.Buffer(64 * kk * 4), hip.Buffer(64 * 8), hip.Buffer(64 * 8)
    for waves in (1, 2, 4, 8):  # 1: one wave plays both roles; 2: the list wave is also the last wave
        gpu.set_search_shape(waves)
        gpu.search_batch_device(dq.ptr, 64, kk, 0, 0, lab.ptr, dist.ptr, None, None, D.ptr, E.ptr)
        hip.synchronize()
        assert np.array_equal(lab.download((64, kk), np.uint64), o_lab), f"waves={waves}"
        assert np.array_equal(dist.download((64, kk), np.float32), o_dist)
        assert np.array_equal(D.download(64, np.uint64), o_D) and np.array_equal(E.download(64, np.uint64), o_E)


# ------------------------------------------------------------------------------------------------
# the latency-bound walk (walk_spec.hpp): one barrier per hop, speculative row loads behind which the visited filter, the list
# merge and the list-cache fill run, neighbour lists fetched with the rows.  LANTERN_GPU_SPEC=1: the four-wave batch shape,
# =2: the lone-query shape (three role waves + eight row waves), =3: the same shape with two nodes per round, the second one
# speculative (walk_twin.hpp).  Every lanes-per-row regime (8 / 16 / 32 / 64), list widths
# with and without the list prefetch (M0 = 10 has no 16-byte pieces at 8 lanes per row), M0 = 64 (two passes per hop), both
# register-list widths (ef <= 64, <= 128), every storage kind -- against the oracle on the same graph: ids, distance bits, D, E.
# ------------------------------------------------------------------------------------------------
SPEC_SHAPES = [("l2sq", 3000, 128, 16, 64, "f32"), ("cos", 2500, 768, 16, 64, "f32"), ("l2sq", 2000, 40, 16, 100, "f32"), ("l2sq", 1500, 300, 5, 40, "f32"),
               ("cos", 2000, 24, 5, 33, "f32"), ("l2sq", 1200, 64, 32, 128, "f32"), ("hamming", 3000, 24, 16, 64, "b1"), ("l2sq", 900, 1536, 16, 64, "f32"),
               ("l2sq", 1500, 768, 8, 64, "f16"), ("cos", 1500, 256, 16, 48, "i8"), ("l2sq", 700, 2000, 4, 20, "f32"), ("l2sq", 800, 600, 32, 64, "f32")]


@pytest.mark.parametrize("spec", ["1", "2", pytest.param("3", marks=needs_experimental)])
@pytest.mark.parametrize("metric,n,d,M,ef,quant", SPEC_SHAPES)
def test_latency_bound_walk_is_the_oracle_walk(capi, oracle, metric, n, d, M, ef, quant, spec, monkeypatch):
    from lantern_amd import hip

    rng = np.random.default_rng(n + d + M)
    scale = np.float32(0.4 if quant == "i8" else 1.0)
    base = rand_rows(rng, n, d, metric) if metric == "hamming" else rand_rows(rng, n, d, metric) * scale
    nq = 300
    queries = rand_rows(rng, nq, d, metric) if metric == "hamming" else rand_rows(rng, nq, d, metric) * scale
    if metric != "hamming":
        base[n // 2: n // 2 + 100] = base[:100]  # exact duplicates: equal distances, the slot decides
    if quant == "f16":
        obase, oq, mode = oracle.round_f16(base), oracle.round_f16(queries), oracle.SUM_WAVE64_F16
    elif quant == "i8":
        obase, oq, mode = oracle.quantize_i8(base), oracle.quantize_i8(queries), oracle.SUM_I8
    else:
        obase, oq, mode = base, queries, oracle.SUM_WAVE64
    gpu = capi.GpuIndex(metric, d, M=M, ef_construction=48, ef=ef, seed=9, quantization="f32" if quant == "b1" else quant)
    gpu.set_add_batch(256, 8)
    gpu.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    g = gpu.export_graph()
    ora = oracle.OracleIndex.from_graph(metric, obase, g, M, 48, ef, 9, mode)
    k = min(10, ef)
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(oq, k, ef, 4)
    monkeypatch.setenv("LANTERN_GPU_SPEC", spec)
    rows = gpu.device_query_rows(queries)
    dq = hip.Buffer.from_numpy(rows)
    lab, dist, D, E = hip.Buffer(nq * k * 8), hip.Buffer(nq * k * 4), hip.Buffer(nq * 8), hip.Buffer(nq * 8)
    gpu.set_search_shape(0)  # the automatic shape: LANTERN_GPU_SPEC decides
    gpu.search_batch_device(dq.ptr, nq, k, 0, 0, lab.ptr, dist.ptr, None, None, D.ptr, E.ptr, query_stride=rows.strides[0])
    hip.synchronize()
    assert np.array_equal(lab.download((nq, k), np.uint64), o_lab)
    assert np.array_equal(dist.download((nq, k), np.float32), o_dist)
    assert np.array_equal(D.download(nq, np.uint64), o_D), "distance-evaluation counts differ"
    assert np.array_equal(E.download(nq, np.uint64), o_E), "expansion counts differ"
    # the classic kernel on the same index agrees (LANTERN_GPU_SPEC=0)
    monkeypatch.setenv("LANTERN_GPU_SPEC", "0")
    gpu.search_batch_device(dq.ptr, nq, k, 0, 0, lab.ptr, dist.ptr, None, None, D.ptr, E.ptr, query_stride=rows.strides[0])
    hip.synchronize()
    assert np.array_equal(lab.download((nq, k), np.uint64), o_lab) and np.array_equal(D.download(nq, np.uint64), o_D)


# the ONE-WAVE walk (walk_solo.hpp, LANTERN_GPU_SPEC=4; on request only -- measured slower than the 3 + 8 wave shape): f32 l2sq / cos rows of
# fewer than 64 chunks, M <= 16 (a multiple of 4), ef <= 64.  8 and 16 lanes per row, one to four chunks per lane, rows that end inside
# a lane's last chunk and rows that do not, full and short neighbour lists, exact duplicates -- ids, distance bits, D, E of the oracle.
SOLO_SHAPES = [("l2sq", 3000, 128, 16, 64), ("cos", 2500, 128, 16, 64), ("l2sq", 2000, 40, 16, 50), ("cos", 2000, 24, 8, 33), ("l2sq", 1500, 252, 16, 64),
               ("l2sq", 1500, 192, 4, 20), ("cos", 1800, 100, 12, 40), ("l2sq", 1200, 32, 16, 64), ("l2sq", 1000, 4, 16, 10), ("cos", 1500, 200, 16, 64)]


@needs_experimental
@pytest.mark.parametrize("metric,n,d,M,ef", SOLO_SHAPES)
def test_one_wave_walk_is_the_oracle_walk(capi, oracle, metric, n, d, M, ef, monkeypatch):
    from lantern_amd import hip

    rng = np.random.default_rng(n + d + M)
    base = rand_rows(rng, n, d, metric)
    nq = 300
    queries = rand_rows(rng, nq, d, metric)
    base[n // 2: n // 2 + 100] = base[:100]  # exact duplicates: equal distances, the slot decides
    gpu = capi.GpuIndex(metric, d, M=M, ef_construction=48, ef=ef, seed=9)
    gpu.set_add_batch(256, 8)
    gpu.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    g = gpu.export_graph()
    ora = oracle.OracleIndex.from_graph(metric, base, g, M, 48, ef, 9, oracle.SUM_WAVE64)
    k = min(10, ef)
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(queries, k, ef, 4)
    monkeypatch.setenv("LANTERN_GPU_SPEC", "4")
    dq = hip.Buffer.from_numpy(hip.padded_rows(queries, False))
    lab, dist, D, E = hip.Buffer(nq * k * 8), hip.Buffer(nq * k * 4), hip.Buffer(nq * 8), hip.Buffer(nq * 8)
    gpu.set_search_shape(0)  # the automatic shape: LANTERN_GPU_SPEC decides
    before = gpu.counters()["search_solo_launches"]
    for rows in (nq, 7, 1):  # more queries than workgroups can be resident (tickets), a handful, one
        gpu.search_batch_device(dq.ptr, rows, k, 0, 0, lab.ptr, dist.ptr, None, None, D.ptr, E.ptr)
        hip.synchronize()
        assert np.array_equal(lab.download((nq, k), np.uint64)[:rows], o_lab[:rows])
        assert np.array_equal(dist.download((nq, k), np.float32)[:rows], o_dist[:rows])
        assert np.array_equal(D.download(nq, np.uint64)[:rows], o_D[:rows]), "distance-evaluation counts differ"
        assert np.array_equal(E.download(nq, np.uint64)[:rows], o_E[:rows]), "expansion counts differ"
    assert gpu.counters()["search_solo_launches"] == before + 3, "the one-wave kernel did not run"
    # usearch_search_ef, one query per call: the host waits on the kernel's own completion counter
    for i in range(20):
        l1, d1 = gpu.search(queries[i], k)
        assert np.array_equal(l1, o_lab[i]) and np.array_equal(d1, o_dist[i]), i
    assert gpu.counters()["search_solo_launches"] == before + 23
    # ... and without the request a lone query keeps the 3 + 8 wave shape (the faster one: DESIGN.md 4.3c)
    monkeypatch.delenv("LANTERN_GPU_SPEC")
    l1, d1 = gpu.search(queries[0], k)
    assert np.array_equal(l1, o_lab[0]) and gpu.counters()["search_solo_launches"] == before + 23


def test_lone_query_and_small_batches_take_the_latency_bound_walk_by_default(capi, oracle):
    """usearch_search_ef (one query per call) and batches around the switch points of the automatic shapes (the lone-query shape up
    to two queries per CU -- the second after the first on the same workgroup --, the classic four-wave walk beyond) -- same answers
    as the oracle; so has the streaming continuation, which searches for more than k."""
    rng = np.random.default_rng(5)
    n, d, M = 6000, 128, 16
    base, queries = rng.standard_normal((n, d), dtype=np.float32), rng.standard_normal((700, d), dtype=np.float32)
    gpu = capi.GpuIndex("l2sq", d, M=M, ef_construction=64, ef=64, seed=2)
    gpu.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    ora = oracle.OracleIndex.from_graph("l2sq", base, gpu.export_graph(), M, 64, 64, 2, oracle.SUM_WAVE64)
    o_lab, o_dist, _, _, _ = ora.search_batch(queries, 10, 64, 4)
    for i in range(40):
        l1, d1 = gpu.search(queries[i], 10)
        assert np.array_equal(l1, o_lab[i]) and np.array_equal(d1, o_dist[i]), i
    for nq in (1, 2, 100, 256, 257, 512, 513, 700):
        lab, dist, _ = gpu.search_batch(queries[:nq], 10)
        assert np.array_equal(lab, o_lab[:nq]) and np.array_equal(dist, o_dist[:nq]), nq
    # streaming: 10 + 10 + 10 results of one scan = the oracle's top 30 (ef grows with what was handed out)
    first, _ = gpu.search(queries[0], 10)
    second, _ = gpu.search(queries[0], 10, streaming=True)
    third, _ = gpu.search(queries[0], 10, streaming=True)
    got = list(first) + list(second) + list(third)
    assert len(set(got)) == 30


@pytest.mark.parametrize("d", [64, 160])  # 8 lanes per row; 16 lanes per row (where LANTERN_GPU_SPEC=3 has a walk of its own)
def test_latency_bound_walk_with_a_spilling_visited_set(capi, oracle, monkeypatch, d):
    rng = np.random.default_rng(11)
    base, queries = rng.standard_normal((4000, d), dtype=np.float32), rng.standard_normal((64, d), dtype=np.float32)
    ora = oracle.OracleIndex("l2sq", d, M=16, ef_construction=64, ef=128, seed=9, sum_mode=oracle.SUM_WAVE64)
    ora.add_many(np.arange(4000, dtype=np.uint64) + 1, base)
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(queries, 10)
    assert o_D.max() > 300
    for vis_slots in ("256", "0"):  # 256: spills to the HBM bitmap after ~190 visits; 0: the bitmap only
        monkeypatch.setenv("LANTERN_GPU_VIS_SLOTS", vis_slots)
        for spec in ("1", "2", "3"):  # ("3" is "2" in a library without csrc/experimental/)
            monkeypatch.setenv("LANTERN_GPU_SPEC", spec)
            gpu = capi.GpuIndex("l2sq", d, M=16, ef_construction=64, ef=128, seed=9)
            gpu.import_graph(base, ora.export_graph())
            lab, dist, _ = gpu.search_batch(queries, 10)
            assert np.array_equal(lab, o_lab) and np.array_equal(dist, o_dist), (vis_slots, spec)
            c = gpu.counters()
            assert c["search_dist_evals"] == int(o_D.sum()) and c["search_expansions"] == int(o_E.sum())


@pytest.mark.parametrize("lds_list", ["0", "1"])
@pytest.mark.parametrize("efc", [20, 64, 100, 128, 160])
def test_build_is_the_same_graph_for_either_list_placement(capi, oracle, efc, lds_list, monkeypatch):
    rng = np.random.default_rng(efc)
    n, d, M = 1500, 40, 8
    base = rng.standard_normal((n, d), dtype=np.float32)
    labels = np.arange(n, dtype=np.uint64) + 1
    ora = oracle.OracleIndex("l2sq", d, M=M, ef_construction=efc, ef=32, seed=9, sum_mode=oracle.SUM_WAVE64)
    ora.add_planned(labels, base, max_batch=64, min_ratio=4)
    monkeypatch.setenv("LANTERN_GPU_LDS_LIST", lds_list)
    gpu = capi.GpuIndex("l2sq", d, M=M, ef_construction=efc, ef=32, seed=9)
    gpu.set_add_batch(64, 4)
    gpu.add_many(labels, base)
    gpu.flush()
    go, gg = ora.export_graph(), gpu.export_graph()
    assert np.array_equal(gg["levels"], go["levels"])
    assert np.array_equal(gg["nbr0"], go["nbr0"]), "level-0 adjacency differs"
    assert np.array_equal(gg["upper_nbr"], go["upper_nbr"]), "upper-level adjacency differs"


# ------------------------------------------------------------------------------------------------
# the committed fixture (tests/golden/oracle_regression.json): the device must reproduce, bit for bit, every case
# whose summation order is the device's own (WAVE64 / WAVE64_F16 / integer metrics)
# ------------------------------------------------------------------------------------------------
def test_device_reproduces_the_committed_fixture(capi):
    import hashlib
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    import make_oracle_golden as gen

    want = json.load(open(os.path.join(root, "tests", "golden", "oracle_regression.json")))["cases"]
    ran = 0
    for name, metric, n, d, M, efc, ef, k, mode, plan, storage in gen.CASES:
        if mode == "SUM_SEQ" and metric != "hamming":
            continue  # usearch's textbook order: compared within tolerance elsewhere, not bit for bit
        rng = np.random.default_rng(int(hashlib.sha256(name.encode()).hexdigest()[:8], 16))
        base, queries = gen.rows(rng, n, d, metric), gen.rows(rng, 16, d, metric)
        if storage == "i8":  # the fixture quantised base * 0.4; the device quantises what it is given
            base, queries = base * np.float32(0.4), queries * np.float32(0.4)
        gpu = capi.GpuIndex(metric, d, M=M, ef_construction=efc, ef=ef, seed=11, quantization=storage)
        gpu.set_add_batch(*(plan if plan else (1, 1)))
        gpu.add_many(np.arange(n, dtype=np.uint64) + 1, base)
        gpu.flush()
        g = gpu.export_graph()
        h = hashlib.sha256()
        for key in ("levels", "nbr0", "upper_off", "upper_nbr"):
            h.update(np.ascontiguousarray(g[key]).tobytes())
        exp = want[name]["expect"]
        assert h.hexdigest() == exp["graph_sha256"], name
        assert (g["entry_slot"], g["max_level"]) == (exp["entry_slot"], exp["max_level"]), name
        lab, dist, cnt = gpu.search_batch(queries, k)
        assert (lab.astype(np.int64) - 1).tolist() == exp["slots"], name
        assert [[f"{int(x):08x}" for x in row] for row in dist.view(np.uint32)] == exp["dist_bits"], name
        ran += 1
    assert ran == 6


# ------------------------------------------------------------------------------------------------
# The row loads of a distance go out in blocks of steps (device_common.hpp LGPU_ROW_BLOCK: three steps for L2sq, two for cosine,
# four for the single-row form), each step predicated on the lane's own chunk count.  Row widths on both sides of every block
# boundary, for every group width G (64 lanes from 128 chunks, 32 from 64, 16 from 32, else 8): chunks = steps x G - 1, steps x G,
# steps x G + 1 -- builds edge for edge, searches bit for bit (ids, distance bits, D, E), a lone query through the latency-bound walk.
# ------------------------------------------------------------------------------------------------
BLOCK_BOUNDARY_CHUNKS = [1, 7, 8, 9, 15, 16, 17, 23, 24, 25, 31,            # G = 8:  1 .. 4 steps
                         32, 33, 47, 48, 49, 63,                            # G = 16: 2 .. 4 steps
                         64, 65, 95, 96, 97, 127,                           # G = 32: 2 .. 4 steps
                         128, 129, 191, 192, 193, 255, 256, 257, 383, 385]  # G = 64: 2 .. 7 steps


@pytest.mark.parametrize("metric", ["l2sq", "cos", "hamming"])
@pytest.mark.parametrize("chunks", BLOCK_BOUNDARY_CHUNKS)
def test_row_widths_around_every_load_block_boundary(capi, oracle, metric, chunks):
    rng = np.random.default_rng(chunks)
    d = chunks * 4 - int(rng.integers(0, 4))  # the last chunk full or ragged
    n = 500
    base, queries = rand_rows(rng, n, d, metric), rand_rows(rng, 40, d, metric)
    labels = np.arange(n, dtype=np.uint64) + LABEL0
    ora = oracle.OracleIndex(metric, d, M=8, ef_construction=32, ef=24, seed=13, sum_mode=oracle.SUM_WAVE64)
    ora.add_planned(labels, base, max_batch=128, min_ratio=4)
    gpu = capi.GpuIndex(metric, d, M=8, ef_construction=32, ef=24, seed=13)
    gpu.set_add_batch(128, 4)
    gpu.add_many(labels, base)
    go, gg = ora.export_graph(), gpu.export_graph()
    assert np.array_equal(gg["nbr0"], go["nbr0"]) and np.array_equal(gg["upper_nbr"], go["upper_nbr"]) and np.array_equal(gg["levels"], go["levels"])
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(queries, 5)
    from lantern_amd import hip

    rows = gpu.device_query_rows(queries)
    dq = hip.Buffer.from_numpy(rows)
    slot, dist, D, E = hip.Buffer(40 * 5 * 4), hip.Buffer(40 * 5 * 4), hip.Buffer(40 * 8), hip.Buffer(40 * 8)
    for waves in (0, 4):  # the automatic shape (40 queries: the latency-bound walk) and the classic kernel
        gpu.set_search_shape(waves)
        gpu.search_batch_device(dq.ptr, 40, 5, 0, 0, None, dist.ptr, slot.ptr, None, D.ptr, E.ptr, query_stride=rows.strides[0])
        hip.synchronize()
        assert np.array_equal(slot.download((40, 5), np.uint32), o_slot), f"slots differ (waves={waves})"
        assert np.array_equal(dist.download((40, 5), np.float32).view(np.uint32), o_dist.view(np.uint32))
        assert np.array_equal(D.download(40, np.uint64), o_D) and np.array_equal(E.download(40, np.uint64), o_E)
    gpu.set_search_shape(0)
    l1, d1 = gpu.search(queries[0], 5)
    assert np.array_equal(l1, o_lab[0][: len(l1)]) and np.array_equal(d1, o_dist[0][: len(d1)])
    # the gathered distance (single-row form, both launch shapes)
    picks = rng.integers(0, n, 64).astype(np.uint32)
    ref = np.array([oracle.distance(queries[1], base[s], metric, oracle.SUM_WAVE64) for s in picks], dtype=np.float32)
    assert np.array_equal(gpu.distance_gather(queries[1], picks).view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("kind", ["f16", "i8"])
@pytest.mark.parametrize("metric", ["l2sq", "cos"])
@pytest.mark.parametrize("chunks", [15, 16, 17, 47, 48, 49, 95, 96, 97, 125])  # (Lantern caps d at 2000: an i8 row has at most 125 chunks)
def test_quantised_row_widths_around_the_load_block_boundaries(capi, oracle, kind, metric, chunks):
    per_chunk = 8 if kind == "f16" else 16
    rng = np.random.default_rng(chunks + per_chunk)
    d = min(2000, chunks * per_chunk - int(rng.integers(0, per_chunk)))
    n = 400
    base = (rng.standard_normal((n, d), dtype=np.float32) * np.float32(0.4)).astype(np.float32)
    queries = (rng.standard_normal((32, d), dtype=np.float32) * np.float32(0.4)).astype(np.float32)
    stored, mode = (oracle.round_f16, oracle.SUM_WAVE64_F16) if kind == "f16" else (oracle.quantize_i8, oracle.SUM_I8)
    labels = np.arange(n, dtype=np.uint64) + LABEL0
    ora = oracle.OracleIndex(metric, d, M=8, ef_construction=32, ef=24, seed=13, sum_mode=mode)
    ora.add_planned(labels, stored(base), max_batch=128, min_ratio=4)
    gpu = capi.GpuIndex(metric, d, M=8, ef_construction=32, ef=24, seed=13, quantization=kind)
    gpu.set_add_batch(128, 4)
    gpu.add_many(labels, base)
    go, gg = ora.export_graph(), gpu.export_graph()
    assert np.array_equal(gg["nbr0"], go["nbr0"]) and np.array_equal(gg["upper_nbr"], go["upper_nbr"])
    o_lab, o_dist, _, _, _ = ora.search_batch(stored(queries), 5)
    for waves in (0, 4):
        gpu.set_search_shape(waves)
        lab, dist, _ = gpu.search_batch(queries, 5)
        assert np.array_equal(lab, o_lab) and np.array_equal(dist.view(np.uint32), o_dist.view(np.uint32)), f"waves={waves}"
