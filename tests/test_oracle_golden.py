"""Pin the CPU oracle against every result the reference's own tests hold for the HNSW
distance path (fixtures in tests/golden/lantern_expected.json)."""
import math

import numpy as np
import pytest

from tests.scan_driver import scan

SUMS = [0, 1, 2]  # SEQ, WAVE64, FAST
LABEL0 = 1  # labels are heap TIDs in Lantern and never 0 (0 = INVALID_ELEMENT_LABEL)


def build(oracle, metric, rows, M=16, efc=128, ef=64, sum_mode=0, batch=False):
    rows = np.asarray(rows)
    dims = rows.shape[1]
    ix = oracle.OracleIndex(metric, dims, M=M, ef_construction=efc, ef=ef, seed=7, sum_mode=sum_mode)
    labels = np.arange(rows.shape[0], dtype=np.uint64) + LABEL0
    if batch:
        ix.add_planned(labels, rows, max_batch=4, min_ratio=1)
    else:
        ix.add_many(labels, rows)
    return ix


def ordered(oracle, ix, q, n):
    """All rows in index order, driven like the executor does (init_k=10)."""
    return scan(lambda k: ix.search(q, k), len(ix), n)


@pytest.mark.parametrize("sum_mode", SUMS)
def test_dist_func_sorted_distances(oracle, golden, sum_mode):
    g, sw = golden["dist_func"], golden["small_world"]
    q = g["query"]
    for metric, key in (("l2sq", "l2sq_sorted"), ("cos", "cos_sorted_2dp"), ("hamming", "hamming_sorted")):
        ix = build(oracle, metric, sw["v"], sum_mode=sum_mode)
        order = ordered(oracle, ix, q, 8)
        assert sorted(order) == list(range(LABEL0, LABEL0 + 8))
        d = [oracle.distance(sw["v"][l - LABEL0], q, metric, sum_mode) for l in order]
        assert [round(x, 2) for x in d] == g[key]


@pytest.mark.parametrize("sum_mode", SUMS)
def test_dist_func_id_groups(oracle, golden, sum_mode):
    g, sw = golden["dist_func"], golden["small_world"]
    for metric, key in (("l2sq", "l2sq_groups"), ("cos", "cos_groups"), ("hamming", "hamming_groups")):
        groups = {}
        for ident, v in zip(sw["ids"], sw["v"]):
            groups.setdefault(round(oracle.distance(v, g["query"], metric, sum_mode), 2), []).append(ident)
        assert sorted((sorted(ids), d) for d, ids in groups.items()) == sorted((sorted(i), d) for i, d in g[key])


@pytest.mark.parametrize("batch", [False, True])
def test_four_nn_of_each_corner(oracle, golden, batch):
    g, sw = golden["four_nn_of_each_corner"], golden["small_world"]
    ix = build(oracle, "l2sq", sw["v"], batch=batch)
    for ident, v in zip(sw["ids"], sw["v"]):
        labels, dists, _ = ix.search(v, 4)
        got = [sw["ids"][int(l) - LABEL0] for l in labels]
        assert got[0] == ident and sorted(got) == sorted(g["rows"][ident])
        assert list(dists) == g["dists"]


def test_extra_small_world_hamming(oracle, golden):
    g = golden["extra_small_world_ham"]
    ix = build(oracle, "hamming", g["v"])
    labels, dists, _ = ix.search(g["query"], 4)
    assert list(dists) == g["sorted"]


@pytest.mark.parametrize("sum_mode", SUMS)
def test_operator_one_offs(oracle, golden, sum_mode):
    for c in golden["operators"]["cases"]:
        d = oracle.distance(c["a"], c["b"], c["op"], sum_mode)
        if "expect" in c:
            assert d == c["expect"], c
        else:
            assert round(d, 2) == c["expect_2dp"], c
    g = golden["operators"]
    for metric, key in (("cos", "op_test_cos"), ("hamming", "op_test_hamming"), ("l2sq", "op_test_l2sq")):
        d = sorted(oracle.distance(r, g["op_test_query"], metric, sum_mode) for r in g["op_test_rows"])
        assert d == g[key]


def test_cos_zero_vector_rules(oracle, golden):
    g = golden["cos_zero_vector_order"]
    # both zero -> 0, exactly one zero -> 1 (hnsw_vector.out:205-210, hnsw_dist_func.out:58-61)
    assert oracle.distance([0, 0, 0], [0, 0, 0], "cos") == 0.0
    assert oracle.distance([0, 0, 0], [0, 0, 2], "cos") == 1.0
    assert oracle.distance([0, 0, 1], [0, 0, 0], "cos") == 1.0
    ix = build(oracle, "l2sq", g["rows"], M=2)
    assert ordered(oracle, ix, g["query"], 3) == [i - 1 + LABEL0 for i in g["l2_order_ids"]]
    for metric, first, rest, M in (("cos", "cos_first_id", "cos_rest_ids", 2), ("hamming", "ham_first_id", "ham_rest_ids", 3)):
        ix = build(oracle, metric, g["rows"], M=M)
        order = ordered(oracle, ix, g["query"], 3)
        assert order[0] == g[first] - 1 + LABEL0
        assert sorted(order[1:]) == [i - 1 + LABEL0 for i in g[rest]]


def test_index_order_equals_seqscan_order(oracle, golden):
    g = golden["correct"]
    ix = build(oracle, "l2sq", g["rows"], M=g["M"])
    with_index = ordered(oracle, ix, g["query"], 4)
    d = [oracle.distance(r, g["query"], "l2sq") for r in g["rows"]]
    without = [int(i) + LABEL0 for i in np.argsort(d, kind="stable")]
    assert with_index == without


def test_vector_small_world_limit7(oracle, golden):
    g, sw = golden["vector_small_world"], golden["small_world"]
    # hnsw_vector.out:60-102 queries the 9-row table (8 corners + [99,99,2]); the cosine index of
    # :253-289 is built on a re-created 8-row table ("inserted 8 elements")
    for metric, key, rows in (("l2sq", "l2sq", sw["v"] + [g["extra_row"]]), ("cos", "cos_2dp", sw["v"])):
        ix = build(oracle, metric, rows, M=g["M"], efc=g["ef_construction"], ef=g["ef"])
        order = ordered(oracle, ix, g["query"], g["limit"])
        d = [round(oracle.distance(rows[l - LABEL0], g["query"], metric), 2) for l in order]
        assert d == g[key]


def test_streaming_continuation_counts(oracle, golden):
    g, sw = golden["streaming"], golden["small_world"]
    rows = sw["v"] + [[99, 99, 2]]
    assert len(rows) == g["indexed_rows"]
    ix = build(oracle, "l2sq", rows, M=5, efc=20, ef=20)
    search = lambda k: ix.search(g["query"], k)
    assert len(scan(search, len(ix), 3, init_k=g["init_k"])) == g["limit_3_count"]
    got = scan(search, len(ix), 15, init_k=g["init_k"])
    assert len(got) == g["limit_15_count"] and len(set(got)) == len(got)


def test_scan_issues_the_reference_sequence_of_searches(oracle, golden):
    """hnsw_select.out:76-140: the reference logs every usearch_search_ef of a scan ("querying index for %d elements"): one search
    of init_k = 10 for LIMIT 3 and for LIMIT 15 (8 rows: index_size == current ends the scan, scan.c:254-256); with init_k = 4,
    LIMIT 15 takes 4, then 8 (the 4 remaining rows), then 8 again (returns nothing: scan ends) -- and the counts 3 / 8."""
    g, sw = golden["scan_k_trace"], golden["small_world"]
    ix = build(oracle, "l2sq", sw["v"], M=g["index"]["M"], efc=g["index"]["ef_construction"], ef=g["index"]["ef"])
    for c in g["cases"]:
        trace = []
        got = scan(lambda k: ix.search(g["query"], k), len(ix), c["limit"], init_k=c["init_k"], trace=trace)
        assert len(got) == c["count"] and len(set(got)) == len(got), c
        assert trace == c["k_trace"], c


def test_index_on_an_expression(oracle, golden):
    g = golden["create_expr"]  # hnsw_create_expr.out:90-94
    ix = oracle.OracleIndex("l2sq", 3, M=g["M"], seed=7)
    ix.add_many(np.asarray(g["ids"], dtype=np.uint64) + LABEL0, np.asarray(g["v"], dtype=np.float32))
    got = scan(lambda k: ix.search(g["query"], k), len(ix), g["limit"])
    assert [l - LABEL0 for l in got] == g["expect_ids"]


def test_insert_into_unlogged_index(oracle, golden):
    g, sw = golden["insert_unlogged"], golden["small_world"]  # hnsw_insert_unlogged.out:62-92
    ix = build(oracle, "l2sq", sw["v"])
    ix.add(100, g["inserted"])
    rows = sw["v"] + [g["inserted"]]
    order = ordered(oracle, ix, g["query"], 20)
    assert [round(oracle.distance(rows[8 if l == 100 else l - LABEL0], g["query"], "l2sq"), 2) for l in order] == g["sorted_2dp"]


def _logged_unlogged_cases(g):
    """(rows the index is BUILT over, rows inserted one by one afterwards, expected [(id, distance)]) for every index the test creates."""
    ids = g["ids"] + [e["id"] for e in g["inserted"]]
    rows = g["v"] + [e["v"] for e in g["inserted"]]
    for built in (8, 9, 10):          # small_world_idx / _idx2 / _idx3 are created over 8, 9 and 10 rows ...
        for total in range(max(built, 8), 11):   # ... and each is queried after every later insert
            yield ids, rows, built, total, g[f"order_{total}"]


def test_exact_order_with_ef_construction_below_m_and_inserts(oracle, golden):
    """hnsw_logged_unlogged.out: unique distances pin the ORDER of ids; ef_construction = 2 (< M = 14) still links every row; rows that
    arrive through aminsert land where the expected output has them, on indexes built over 8, 9 and 10 rows."""
    g = golden["logged_unlogged"]
    for ids, rows, built, total, want in _logged_unlogged_cases(g):
        ix = oracle.OracleIndex("l2sq", 4, M=g["index"]["M"], ef_construction=g["index"]["ef_construction"], ef=g["index"]["ef"], seed=7)
        ix.add_many(np.arange(built, dtype=np.uint64) + LABEL0, np.asarray(rows[:built], dtype=np.float32))
        for i in range(built, total):
            ix.add(i + LABEL0, rows[i])
        got = scan(lambda k: ix.search(g["query"], k), len(ix), g["limit"])
        assert [[ids[l - LABEL0], int(oracle.distance(rows[l - LABEL0], g["query"], "l2sq"))] for l in got] == want, (built, total)


def test_insert_then_search(oracle, golden):
    g, sw = golden["insert_then_search"], golden["small_world"]
    ix = build(oracle, "l2sq", sw["v"])
    ix.add(100, g["inserted"])
    rows = sw["v"] + [g["inserted"]]
    order = ordered(oracle, ix, g["query"], 9)
    assert len(order) == 9
    d = [oracle.distance(rows[8 if l == 100 else l - LABEL0], g["query"], "l2sq") for l in order]
    assert d == g["sorted"]


def test_partial_index(oracle, golden):
    g, sw = golden["partial_index"], golden["small_world"]
    keep = [(i, v) for i, (v, b) in enumerate(zip(sw["v"], sw["b"])) if not b]
    ix = oracle.OracleIndex("l2sq", 3, M=g["M"], seed=7)
    for i, v in keep:
        ix.add(i + LABEL0, v)
    labels, _, _ = ix.search(g["query"], 3)
    got = [sw["ids"][int(l) - LABEL0] for l in labels]
    assert got[0] == g["first"] and sorted(got[1:]) == sorted(g["then_any_order"])


def test_pagination_with_duplicates(oracle, golden):
    g = golden["pagination_duplicates"]
    rows, ids = [], []
    for i in g["ramp_ids"]:
        rows.append([np.float32(i) / np.float32(10)] * g["dim"])
        ids.append(i + 1)  # +1: label 0 is INVALID_ELEMENT_LABEL
    for j in range(g["dup_count"]):
        rows.append([g["dup_value"]] * g["dim"])
        ids.append(g["dup_first_id"] + j + 1)
    rows = np.asarray(rows, dtype=np.float32)
    ix = oracle.OracleIndex("l2sq", g["dim"], seed=7)
    ix.add_many(ids, rows)
    q = [g["dup_value"]] * g["dim"]
    got = scan(lambda k: ix.search(q, k), len(ix), g["limit"], init_k=g["init_k"])
    assert len(got) == g["limit"]
    assert len(set(got)) == len(got), "an id was returned twice while paginating"


def test_planner_bound(oracle, golden):
    for c in golden["cost_estimate"]["cases"]:
        assert oracle.lib().lo_estimate_visited_tuples(float(c["n"]), c["M"], c["ef"]) == c["tuples"]


def test_level_distribution(oracle):
    # insert.c:32-46: level = floor(-ln(U)/ln(M)); P(level >= 1) = 1/M
    M, n = 16, 200000
    lv = np.array([oracle.level_for(5, i, M) for i in range(n)])
    assert abs((lv >= 1).mean() - 1 / M) < 0.003
    assert abs((lv >= 2).mean() - 1 / M**2) < 0.001
    assert lv.min() == 0 and lv.max() <= 8


def test_sum_orders_agree_within_tolerance(oracle):
    rng = np.random.default_rng(0)
    for d in (3, 16, 100, 128, 768, 1536, 2000):
        a, b = rng.standard_normal(d, dtype=np.float32), rng.standard_normal(d, dtype=np.float32)
        for metric in ("l2sq", "cos"):
            ref = oracle.distance(a, b, metric, 0)
            exact = float(np.sum((a.astype(np.float64) - b) ** 2)) if metric == "l2sq" else float(
                1 - a.astype(np.float64) @ b / math.sqrt(float(a.astype(np.float64) @ a) * float(b.astype(np.float64) @ b)))
            for mode in (1, 2):
                got = oracle.distance(a, b, metric, mode)
                assert abs(got - ref) <= 1e-5 * max(1.0, abs(ref))
            assert abs(ref - exact) <= 1e-5 * max(1.0, abs(exact))


def test_recall_floor_on_random_data(oracle):
    # integration_tests.py:249-257 asserts recall@10 >= 0.7 with M=8; our synthetic analogue
    rng = np.random.default_rng(11)
    base = rng.standard_normal((3000, 32), dtype=np.float32)
    queries = rng.standard_normal((100, 32), dtype=np.float32)
    truth, _ = oracle.bruteforce(base, queries, 10, "l2sq")
    for batch in (False, True):
        ix = oracle.OracleIndex("l2sq", 32, M=8, seed=3)
        labels = np.arange(3000, dtype=np.uint64) + LABEL0
        if batch:
            ix.add_planned(labels, base, max_batch=256, min_ratio=16)
        else:
            ix.add_many(labels, base)
        _, _, slots, D, E = ix.search_batch(queries, 10)
        assert oracle.recall_at_k(slots, truth) >= 0.7
        assert D.min() > 0 and E.min() > 0


def test_i8_quantisation_rule_and_integer_metrics(oracle):
    """quant_bits=8: "i8 uniform [-1-1]=>[-100,100] quantization" (lantern_hnsw/test/sql/hnsw_sq.sql:33-34);
    usearch's l2sq_i8_t / cos_i8_t accumulate in int32.  PARITY UNPINNED by the reference beyond that comment."""
    import numpy as np

    x = np.array([0.0, 1.0, -1.0, 0.505, -0.505, 0.019, 3.0, -3.0, float("nan"), 0.999], dtype=np.float32)
    assert oracle.quantize_i8(x).tolist() == [0, 100, -100, 50, -50, 1, 100, -100, 0, 99]
    a = oracle.quantize_i8(np.array([0.5, -0.25, 1.7, 0.009], dtype=np.float32))
    b = oracle.quantize_i8(np.array([0.1, 0.1, -3.0, 0.5], dtype=np.float32))
    assert oracle.distance(a, b, "l2sq", oracle.SUM_I8) == 40.0**2 + 35.0**2 + 200.0**2 + 50.0**2
    ab, a2, b2 = 50 * 10 - 25 * 10 - 100 * 100, 50**2 + 25**2 + 100**2, 10**2 + 10**2 + 100**2 + 50**2
    want = np.float32(1) - np.float32(ab) / (np.sqrt(np.float32(a2)) * np.sqrt(np.float32(b2)))
    assert oracle.distance(a, b, "cos", oracle.SUM_I8) == want
    z = np.zeros(4, dtype=np.float32)
    assert oracle.distance(z, z, "cos", oracle.SUM_I8) == 0.0 and oracle.distance(z, a, "cos", oracle.SUM_I8) == 1.0
    # an i8 HNSW walk on the oracle is self-consistent with its own brute force
    rng = np.random.default_rng(1)
    base = oracle.quantize_i8(rng.standard_normal((800, 24), dtype=np.float32) * np.float32(0.4))
    ix = oracle.OracleIndex("l2sq", 24, M=8, ef_construction=64, ef=64, seed=3, sum_mode=oracle.SUM_I8)
    ix.add_many(np.arange(800) + 1, base)
    q = oracle.quantize_i8(rng.standard_normal((32, 24), dtype=np.float32) * np.float32(0.4))
    _, _, slots, _, _ = ix.search_batch(q, 10)
    truth, _ = oracle.bruteforce(base, q, 10, "l2sq", oracle.SUM_I8)
    assert oracle.recall_at_k(slots, truth) > 0.9


def test_oracle_regression_fixture(oracle):
    """The oracle itself is pinned: graphs, result lists, distance bits and D/E counters of seven seeded cases
    (tests/golden/oracle_regression.json, made by scripts/make_oracle_golden.py) must not drift."""
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    import make_oracle_golden as gen

    want = json.load(open(os.path.join(root, "tests", "golden", "oracle_regression.json")))["cases"]
    assert sorted(want) == sorted(c[0] for c in gen.CASES)
    for c in gen.CASES:
        assert gen.run_case(*c) == want[c[0]]["expect"], c[0]


def test_wave_order_simd_form_has_the_scalar_restatements_bits(oracle):
    """LO_SUM_WAVE64 over f32 runs eight lanes of the device's reduction tree per AVX2 instruction (the build-parity tests at
    100k rows need it); the scalar loops are the definition.  Every shape class: fewer chunks than lanes, ragged last chunk,
    ragged last round, each lane-count regime (8 / 16 / 32 / 64 lanes)."""
    import numpy as np

    rng = np.random.default_rng(5)
    try:
        for d in list(range(1, 70)) + [96, 100, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 768, 769, 770, 771, 1000, 1536, 2000]:
            for metric in ("l2sq", "cos"):
                a = rng.standard_normal(d, dtype=np.float32)
                b = rng.standard_normal(d, dtype=np.float32)
                oracle.set_wave_simd(True)
                x = oracle.distance(a, b, metric, oracle.SUM_WAVE64)
                oracle.set_wave_simd(False)
                y = oracle.distance(a, b, metric, oracle.SUM_WAVE64)
                assert np.float32(x).tobytes() == np.float32(y).tobytes(), (d, metric)
    finally:
        oracle.set_wave_simd(True)


def test_batch_phases_on_several_threads_build_the_same_graph(oracle):
    """lo_add_batch may run a batch's walks and its (node, level) groups of reverse links on several threads (walks read only
    the pre-batch graph, groups touch one list each): the graph must not depend on the thread count."""
    import numpy as np

    rng = np.random.default_rng(12)
    for metric, n, d, M in (("l2sq", 4000, 48, 6), ("cos", 2500, 130, 4)):
        base = rng.standard_normal((n, d), dtype=np.float32)
        labels = np.arange(n, dtype=np.uint64) + 1
        graphs = []
        for threads in (1, 4):
            ix = oracle.OracleIndex(metric, d, M=M, ef_construction=40, ef=32, seed=6, sum_mode=oracle.SUM_WAVE64)
            ix.set_build_threads(threads)
            ix.add_planned(labels, base, max_batch=1024, min_ratio=8)
            graphs.append(ix.export_graph())
        for key in ("levels", "nbr0", "upper_off", "upper_nbr", "labels"):
            assert np.array_equal(graphs[0][key], graphs[1][key]), (metric, key)
        assert graphs[0]["entry_slot"] == graphs[1]["entry_slot"]


def test_vectorised_level_draw_is_lo_level_for(oracle):
    import numpy as np

    for seed, M, first in ((42, 16, 0), (7, 2, 12345), (1, 5, 999_000), (21, 128, 3)):
        want = np.array([oracle.level_for(seed, first + j, M) for j in range(3000)])
        assert np.array_equal(oracle.levels_for(seed, first, 3000, M), want)
