"""Builds and searches over data made of exact ties: the total order (distance, slot) decides everything.

The reference's own regression data is tie-heavy (small_world: eight corners of a cube, `hnsw_dist_func.out:5-40`; sift rows are small
integers), and PostgreSQL tables hold duplicated and all-zero vectors as a matter of course (`hnsw_vector.out:205-210` pins the cosine
zero rules).  Random Gaussian rows never produce an exact tie, so the cases below are the ones where a heap, a selection or a re-prune
that breaks ties differently from the oracle shows: lattices, every row three times, one row repeated n times, zero rows under cosine,
one-hot rows, bit rows drawn from a handful of patterns.  Bar: graphs edge for edge, ids / distance bits / D / E identical.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LABEL0 = 1


@pytest.fixture(scope="module")
def capi():
    from lantern_amd import capi

    capi.lib()
    assert capi.device_count() > 0, "no HIP device: the gpu tests need a real MI355X"
    return capi


def lattice(rng, n, d):
    return rng.integers(-1, 2, size=(n, d)).astype(np.float32)


def triplicates(rng, n, d):
    distinct = rng.standard_normal(((n + 2) // 3, d), dtype=np.float32)
    return distinct[rng.permutation(np.repeat(np.arange(len(distinct)), 3))[:n]]


def one_row(rng, n, d):
    return np.repeat(rng.standard_normal((1, d), dtype=np.float32), n, axis=0)


def with_zero_rows(rng, n, d):
    x = rng.integers(-2, 3, size=(n, d)).astype(np.float32)
    x[rng.random(n) < 0.2] = 0.0
    return x


def one_hot(rng, n, d):
    x = np.zeros((n, d), np.float32)
    x[np.arange(n), rng.integers(0, d, n)] = rng.integers(1, 4, n).astype(np.float32)
    return x


def few_bit_patterns(rng, n, d):
    patterns = rng.integers(0, 2**32, size=(12, d), dtype=np.uint32)
    return patterns[rng.integers(0, 12, n)]


def sparse_bits(rng, n, d):
    x = np.zeros((n, d), np.uint32)
    x[np.arange(n), rng.integers(0, d, n)] = np.uint32(1) << rng.integers(0, 32, n).astype(np.uint32)
    return x


def device_search(gpu, queries, k, ef):
    from lantern_amd import hip

    nq = len(queries)
    rows = gpu.device_query_rows(queries)
    dq = hip.Buffer.from_numpy(rows)
    lab, dist, slot = hip.Buffer(nq * k * 8), hip.Buffer(nq * k * 4), hip.Buffer(nq * k * 4)
    cnt, D, E = hip.Buffer(nq * 4), hip.Buffer(nq * 8), hip.Buffer(nq * 8)
    gpu.search_batch_device(dq.ptr, nq, k, ef, 0, lab.ptr, dist.ptr, slot.ptr, cnt.ptr, D.ptr, E.ptr, query_stride=rows.strides[0])
    hip.synchronize()
    return (lab.download((nq, k), np.uint64), dist.download((nq, k), np.float32), slot.download((nq, k), np.uint32),
            D.download(nq, np.uint64), E.download(nq, np.uint64))


DATA = {"lattice": lattice, "triplicates": triplicates, "one_row": one_row, "zero_rows": with_zero_rows, "one_hot": one_hot,
        "few_bit_patterns": few_bit_patterns, "sparse_bits": sparse_bits}

CASES = [
    # data, metric, n, d, M, efc
    ("lattice", "l2sq", 1500, 6, 8, 40),
    ("lattice", "cos", 1200, 7, 6, 32),
    ("lattice", "l2sq", 900, 130, 16, 64),     # 32 lanes per row, padded chunk
    ("triplicates", "l2sq", 1500, 64, 8, 40),
    ("triplicates", "cos", 900, 768, 16, 64),
    ("one_row", "l2sq", 700, 32, 4, 24),        # every distance 0
    ("one_row", "cos", 500, 100, 16, 40),
    ("zero_rows", "cos", 1200, 5, 8, 40),       # hnsw_vector.out:205-210
    ("zero_rows", "l2sq", 1000, 9, 5, 30),
    ("one_hot", "l2sq", 1200, 16, 8, 32),
    ("one_hot", "cos", 1200, 24, 8, 32),        # distances 0 or 1 only
    ("few_bit_patterns", "hamming", 1200, 4, 8, 40),
    ("sparse_bits", "hamming", 1000, 3, 6, 32),  # distances 0 or 2 only
]


@pytest.mark.parametrize("plan", [(1, 1), (64, 4), (512, 16)])
@pytest.mark.parametrize("data,metric,n,d,M,efc", CASES)
def test_tie_heavy_build_and_search_match_the_oracle(capi, oracle, data, metric, n, d, M, efc, plan):
    rng = np.random.default_rng(n + 31 * d + M)
    base = DATA[data](rng, n, d)
    queries = np.concatenate([base[rng.integers(0, n, 24)], DATA[data](rng, 24, d)])
    labels = np.arange(n, dtype=np.uint64) + LABEL0
    ora = oracle.OracleIndex(metric, d, M=M, ef_construction=efc, ef=48, seed=5, sum_mode=oracle.SUM_WAVE64)
    ora.add_planned(labels, base, max_batch=plan[0], min_ratio=plan[1])
    gpu = capi.GpuIndex(metric, d, M=M, ef_construction=efc, ef=48, seed=5)
    gpu.set_add_batch(*plan)
    gpu.add_many(labels, base)
    gpu.flush()
    go, gg = ora.export_graph(), gpu.export_graph()
    assert gg["entry_slot"] == go["entry_slot"] and gg["max_level"] == go["max_level"]
    assert np.array_equal(gg["levels"], go["levels"])
    assert np.array_equal(gg["upper_off"], go["upper_off"])
    assert np.array_equal(gg["nbr0"], go["nbr0"]), "level-0 adjacency differs"
    assert np.array_equal(gg["upper_nbr"], go["upper_nbr"]), "upper-level adjacency differs"
    for k, ef in ((10, 48), (1, 1), (30, 30)):
        o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(queries, k, ef=ef)
        lab, dist, slot, D, E = device_search(gpu, queries, k, ef)
        assert np.array_equal(slot, o_slot), f"slots differ at k={k} ef={ef}"
        assert np.array_equal(lab, o_lab)
        assert np.array_equal(dist.view(np.uint32), o_dist.view(np.uint32)), f"distance bits differ at k={k} ef={ef}"
        assert np.array_equal(D, o_D) and np.array_equal(E, o_E)
        h_lab, h_dist, _ = gpu.search_batch(queries, k, ef=ef)
        assert np.array_equal(h_lab, o_lab) and np.array_equal(h_dist.view(np.uint32), o_dist.view(np.uint32))
    # a lone query takes the latency-bound walk
    for q in queries[:4]:
        l1, d1 = gpu.search(q, 10)
        o_lab, o_dist, _, _, _ = ora.search_batch(q[None], 10)
        assert np.array_equal(l1, o_lab[0][: len(l1)]) and np.array_equal(d1, o_dist[0][: len(d1)])


@pytest.mark.parametrize("data,metric,d", [("lattice", "l2sq", 6), ("zero_rows", "cos", 5), ("triplicates", "l2sq", 64), ("few_bit_patterns", "hamming", 4)])
def test_tie_heavy_scan_never_repeats_and_follows_the_oracle(capi, oracle, data, metric, d):
    """A scan cursor over tied rows: continuation must hand out every row once, in the oracle's (distance, slot) order."""
    from tests.scan_driver import scan as oracle_scan

    rng = np.random.default_rng(17 + d)
    n = 400
    base = DATA[data](rng, n, d)
    labels = np.arange(n, dtype=np.uint64) + LABEL0
    ora = oracle.OracleIndex(metric, d, M=8, ef_construction=40, ef=32, seed=5, sum_mode=oracle.SUM_WAVE64)
    ora.add_many(labels, base)
    gpu = capi.GpuIndex(metric, d, M=8, ef_construction=40, ef=32, seed=5)
    gpu.set_add_batch(1, 1)
    gpu.add_many(labels, base)
    for q in (base[3], DATA[data](rng, 1, d)[0]):
        s = capi.Scan(gpu, init_k=10)
        s.rescan(q)
        got = s.fetch(150)
        s.end()
        assert len(set(got)) == len(got)
        want = oracle_scan(lambda k, q=q: ora.search(q, k), n, 150, init_k=10)
        assert list(got) == list(want)


@pytest.mark.parametrize("kind", ["f16", "i8"])
@pytest.mark.parametrize("data,metric,n,d,M,efc", [("lattice", "l2sq", 1200, 6, 8, 40), ("zero_rows", "cos", 1000, 5, 8, 40),
                                                   ("triplicates", "l2sq", 900, 200, 16, 48), ("one_hot", "cos", 900, 40, 6, 32)])
def test_tie_heavy_quantised_storage_matches_the_oracle(capi, oracle, kind, data, metric, n, d, M, efc):
    """The same ties under f16 / i8 storage (options.c:137-158): rows scaled into the i8 range, the oracle fed the stored values."""
    rng = np.random.default_rng(n + d)
    scale = np.float32(0.25)
    base, queries = DATA[data](rng, n, d) * scale, DATA[data](rng, 32, d) * scale
    queries[:16] = base[rng.integers(0, n, 16)]
    stored, mode = (oracle.round_f16, oracle.SUM_WAVE64_F16) if kind == "f16" else (oracle.quantize_i8, oracle.SUM_I8)
    labels = np.arange(n, dtype=np.uint64) + LABEL0
    ora = oracle.OracleIndex(metric, d, M=M, ef_construction=efc, ef=48, seed=5, sum_mode=mode)
    ora.add_planned(labels, stored(base), max_batch=128, min_ratio=4)
    gpu = capi.GpuIndex(metric, d, M=M, ef_construction=efc, ef=48, seed=5, quantization=kind)
    gpu.set_add_batch(128, 4)
    gpu.add_many(labels, base)
    go, gg = ora.export_graph(), gpu.export_graph()
    assert np.array_equal(gg["levels"], go["levels"])
    assert np.array_equal(gg["nbr0"], go["nbr0"]) and np.array_equal(gg["upper_nbr"], go["upper_nbr"])
    o_lab, o_dist, _, _, _ = ora.search_batch(stored(queries), 10)
    lab, dist, _ = gpu.search_batch(queries, 10)
    assert np.array_equal(lab, o_lab) and np.array_equal(dist.view(np.uint32), o_dist.view(np.uint32))
    l1, d1 = gpu.search(queries[0], 10)
    assert np.array_equal(l1, o_lab[0][: len(l1)])
