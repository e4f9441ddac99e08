"""The cache model behind bench.py's roofline.frac_dram_model (lantern_amd/tools/cache_model.c, bench_cache_model.py), on traces whose
answer is known: (1) it IS an LRU (cross-checked against a 20-line reference); (2) a gather with no reuse over a table far larger than
the caches costs its algorithmic bytes at both levels (the k_gather_walkshape calibration: profiles/r06_cache_model_calibration.md --
the counters read 1.024 x the algorithmic bytes there, the model 0.9986 x); (3) an index that fits the Infinity Cache costs ~0 DRAM bytes once warm, and one that fits an L2 ~0
fabric bytes (scripts/profile_small_index.py); (4) hub rows shared by every query are served by the caches while the cold tail is not."""
import collections

import numpy as np
import pytest

ROW, LIST0, LISTU = 3072, 128, 64


@pytest.fixture(scope="module")
def cm():
    from lantern_amd import build

    build.build()
    import bench_cache_model

    bench_cache_model.lib()
    return bench_cache_model


def as_launch(per_query):
    cap = max(len(t) for t in per_query)
    tr = np.zeros((len(per_query), cap), dtype=np.uint32)
    ct = np.zeros(len(per_query), dtype=np.uint32)
    for i, t in enumerate(per_query):
        tr[i, :len(t)] = t
        ct[i] = len(t)
    return tr, ct


def reference_lru(stream, cap_bytes):
    """Byte-capacity LRU over (key, bytes): bytes missed."""
    od, used, missed = collections.OrderedDict(), 0, 0
    for key, b in stream:
        if key in od:
            od.move_to_end(key)
            continue
        missed += b
        while used + b > cap_bytes:
            _, eb = od.popitem(last=False)
            used -= eb
        od[key] = b
        used += b
    return missed


def test_it_is_an_lru(cm):
    """One walker, one XCD: the replay order is the trace order, so both levels must equal the reference LRU on the same stream
    (the Infinity Cache sees the L2's miss stream)."""
    rng = np.random.default_rng(1)
    for _ in range(5):
        nq, hops = 7, 30
        queries = []
        for _q in range(nq):
            t = []
            for _h in range(hops):
                node = int(rng.integers(0, 60))
                t.append(node | (cm.LIST0 if rng.random() < 0.8 else cm.LISTU))
                t += [int(x) for x in rng.integers(0, 60, size=int(rng.integers(0, 6)))]
            queries.append(t)
        tr, ct = as_launch(queries)
        l2_cap, mall_cap = 9 * ROW + 300, 25 * ROW
        got = cm.replay([tr], [ct], walkers=1, row_bytes=ROW, list0_bytes=LIST0, listu_bytes=LISTU, l2_bytes_per_xcd=l2_cap, mall_bytes=mall_cap, xcds=1)[0]
        size = lambda e: LIST0 if e >> 30 == 2 else LISTU if e >> 30 == 3 else ROW
        stream = [(e, size(e)) for t in queries for e in t]
        assert got["accesses"] == len(stream) and got["access_bytes"] == sum(b for _, b in stream)
        # L2 level
        od, used, l2_missed, miss_stream = collections.OrderedDict(), 0, 0, []
        for key, b in stream:
            if key in od:
                od.move_to_end(key)
                continue
            l2_missed += b
            miss_stream.append((key, b))
            while used + b > l2_cap:
                _, eb = od.popitem(last=False)
                used -= eb
            od[key] = b
            used += b
        assert got["fabric_bytes"] == l2_missed
        assert got["dram_bytes"] == reference_lru(miss_stream, mall_cap)


def test_gather_without_reuse_costs_its_algorithmic_bytes(cm):
    """Uniformly random rows of a 3 GB table through 32 MiB of L2 and a 256 MiB Infinity Cache: (nearly) every access misses
    everywhere -- model == algorithmic, as the counters say of the real gather (traffic / algorithmic 1.02: r06_cache_model_calibration.md)."""
    rng = np.random.default_rng(2)
    n_rows, nq, per = 1_000_000, 1024, 300
    tr = rng.integers(0, n_rows, size=(nq, per), dtype=np.uint32)
    ct = np.full(nq, per, dtype=np.uint32)
    got = cm.replay([tr], [ct], walkers=1536, row_bytes=ROW, list0_bytes=LIST0, listu_bytes=LISTU)[0]
    alg = nq * per * ROW
    assert got["access_bytes"] == alg
    assert 0.98 * alg <= got["fabric_bytes"] <= alg
    assert 0.85 * alg <= got["dram_bytes"] <= got["fabric_bytes"]  # the first 256 MiB / 3 GB of the table get a second chance: < 9 %
    assert cm.distinct_bytes(tr, ct, ROW, LIST0, LISTU) <= got["dram_bytes"] + 1e-6


def test_an_index_that_fits_is_free_once_warm(cm):
    """30 000 rows x 3 KiB = 92 MB: larger than the L2s, smaller than the Infinity Cache.  Launch 1 loads it (cold misses = its
    distinct bytes), launch 2 costs no DRAM bytes but still crosses the fabric.  1000 rows (3 MB) fit ONE L2: no fabric bytes either."""
    rng = np.random.default_rng(3)
    nq, per = 2048, 200
    mk = lambda n_rows: (rng.integers(0, n_rows, size=(nq, per), dtype=np.uint32), np.full(nq, per, dtype=np.uint32))
    (t1, c1), (t2, c2) = mk(30_000), mk(30_000)
    r1, r2 = cm.replay([t1, t2], [c1, c2], walkers=1536, row_bytes=ROW, list0_bytes=LIST0, listu_bytes=LISTU)
    assert r1["dram_bytes"] == pytest.approx(cm.distinct_bytes(t1, c1, ROW, LIST0, LISTU), rel=1e-9)
    assert r2["dram_bytes"] <= 0.001 * r2["access_bytes"]
    assert r2["fabric_bytes"] >= 0.5 * r2["access_bytes"]
    (s1, d1), (s2, d2) = mk(1000), mk(1000)
    q1, q2 = cm.replay([s1, s2], [d1, d2], walkers=1536, row_bytes=ROW, list0_bytes=LIST0, listu_bytes=LISTU)
    assert q2["dram_bytes"] == 0 and q2["fabric_bytes"] <= 0.001 * q2["access_bytes"]


def test_hub_rows_are_served_by_the_caches_and_the_tail_is_not(cm):
    """Every query evaluates the same 2000 hub rows (6 MB: more than one L2, far less than the Infinity Cache) and 100 rows of its
    own from a 3 GB table: DRAM delivers about the tail only, the fabric carries part of the hub traffic on top."""
    rng = np.random.default_rng(4)
    nq = 2048
    hubs = np.arange(2000, dtype=np.uint32)
    per_query = []
    for _ in range(nq):
        own = rng.integers(10_000, 1_000_000, size=100, dtype=np.uint32)
        t = np.concatenate([rng.permutation(hubs)[:400], own])
        rng.shuffle(t)
        per_query.append([int(x) for x in t])
    tr, ct = as_launch(per_query)
    warm, got = cm.replay([tr, tr], [ct, ct], walkers=1536, row_bytes=ROW, list0_bytes=LIST0, listu_bytes=LISTU)
    tail = nq * 100 * ROW
    assert 0.9 * tail <= got["dram_bytes"] <= 1.05 * tail
    assert got["fabric_bytes"] > got["dram_bytes"]
    assert got["access_bytes"] == nq * 500 * ROW


def test_dropped_entries_are_reported(cm):
    tr = np.zeros((3, 4), dtype=np.uint32)
    ct = np.array([4, 9, 2], dtype=np.uint32)
    got = cm.replay([tr], [ct], walkers=2, row_bytes=ROW, list0_bytes=LIST0, listu_bytes=LISTU)[0]
    assert got["dropped_entries"] == 5 and got["accesses"] == 4 + 4 + 2


def test_uniformly_random_gather_misses_the_infinity_cache_one_minus_c_over_t_of_the_time(cm):
    """bench.py gather_ceiling() takes the DRAM share of its whole-table gather as 1 - C / T (C: the Infinity Cache, T: the table)
    instead of replaying the 17 M-entry trace in every run: the model gives that figure on a scaled-down case (and gave 0.9126 on the
    full one: profiles/r06_cache_model_calibration.md)."""
    rng = np.random.default_rng(5)
    rows, walkers, per_walker = 20_000, 64, 6_000   # T = 61.4 MB of 3 KiB rows
    mall, l2 = 16 << 20, 256 << 10                   # C = 16 MiB behind eight small L2s
    launches = [as_launch([rng.integers(0, rows, per_walker).astype(np.uint32) for _ in range(walkers)]) for _ in range(2)]
    res = cm.replay([t for t, _ in launches], [c for _, c in launches], walkers, ROW, LIST0, LISTU, l2_bytes_per_xcd=l2, mall_bytes=mall)
    warm = res[1]  # the second launch finds the caches as the first left them
    share = warm["dram_bytes"] / warm["access_bytes"]
    assert abs(share - (1.0 - mall / (rows * ROW))) < 0.02, share
    assert warm["fabric_bytes"] / warm["access_bytes"] > 0.95  # the L2s (2 MiB in all) catch next to nothing of it
