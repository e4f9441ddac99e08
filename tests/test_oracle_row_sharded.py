"""CPU: the oracle's restatement of the row-sharded build (oracle.row_sharded_build / lo_add_batch_cand) -- the checker of
tests/test_gpu_sharded_build.py::test_row_sharded_build_is_the_oracles_restatement_edge_for_edge.  Here: the batch plan and its
apportionment, and that what it builds is a usable HNSW of the same quality as the one-index build of the same rows."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def oracle():
    from oracle import binding

    binding.build()
    return binding


@pytest.mark.parametrize("sizes", [(1000, 1000), (900, 0, 1100), (1, 5, 2000), (7,), (3, 3, 3, 3, 3, 3, 3, 3)])
def test_plan_draws_every_shard_dry_in_proportion(oracle, sizes):
    n = sum(sizes)
    levels = oracle.levels_for(21, 0, n, 8)
    plan = oracle.row_shard_plan(sizes, levels, 128, 8)
    assert plan[0][:2] == (0, 1), "the first node is a batch of its own"
    at, taken, top = 0, [0] * len(sizes), 0
    for first, b, share in plan:
        assert first == at and b >= 1 and sum(share) == b and all(x >= 0 for x in share)
        assert b == 1 or b <= max(1, first // 8) and b <= 128
        if b > 1:
            assert max(levels[first:first + b]) <= top, "a node that raises the top level is inserted alone"
        top = max(top, int(levels[first]) if b == 1 else top)
        for r, x in enumerate(share):
            taken[r] += x
            # never further than one row from the shard's proportional share of the rows handed out so far
            assert abs(taken[r] - sizes[r] * (first + b) / n) <= len(sizes), (r, first, b)
        at += b
    assert at == n and tuple(taken) == tuple(sizes)


def test_restatement_builds_a_graph_as_good_as_the_one_index_build(oracle):
    rng = np.random.default_rng(8)
    n, d = 4000, 24
    base = rng.standard_normal((n, d), dtype=np.float32)
    queries = rng.standard_normal((300, d), dtype=np.float32)
    labels = np.arange(n, dtype=np.uint64) + 1
    cuts = (0, 1700, 1700, 4000)
    shards = [(labels[cuts[r]:cuts[r + 1]], base[cuts[r]:cuts[r + 1]]) for r in range(3)]
    glob, by_slot = oracle.row_sharded_build("l2sq", d, shards, M=8, ef_construction=48, ef=32, seed=21, max_batch=256, min_ratio=8)
    again, by_slot2 = oracle.row_sharded_build("l2sq", d, shards, M=8, ef_construction=48, ef=32, seed=21, max_batch=256, min_ratio=8)
    g, g2 = glob.export_graph(), again.export_graph()
    assert all(np.array_equal(g[k], g2[k]) for k in ("levels", "labels", "nbr0", "upper_off", "upper_nbr")) and np.array_equal(by_slot, by_slot2)
    assert len(glob) == n and np.array_equal(np.sort(by_slot), labels) and np.array_equal(g["labels"], by_slot)
    nbr0 = g["nbr0"]
    live = nbr0 != 0xFFFFFFFF
    assert (nbr0[live] < n).all() and not (nbr0 == np.arange(n, dtype=np.uint32)[:, None]).any()
    assert (live[:, :-1] >= live[:, 1:]).all(), "lists have no holes"
    one = oracle.OracleIndex("l2sq", d, M=8, ef_construction=48, ef=32, seed=21)
    one.add_planned(labels, base, 256, 8)
    truth = oracle.bruteforce(base, queries, 10, "l2sq")[0]
    rec = []
    for ix in (glob, one):
        lab = ix.search_batch(queries, 10, 64)[0]
        rec.append(np.mean([len(set(lab[i].tolist()) & set((truth[i] + 1).tolist())) / 10 for i in range(len(queries))]))
    assert rec[0] >= rec[1] - 0.02, rec


@pytest.mark.parametrize("sizes", [(1000, 1000), (900, 0, 1100), (1, 5, 2000), (7,), (3,) * 8, (0, 0, 5), (12345, 54321, 1)])
@pytest.mark.parametrize("plan", [(128, 8), (8192, 16), (1, 1)])
def test_the_librarys_plan_is_the_restatements(oracle, sizes, plan):
    """lantern_gpu_row_shard_plan (host arithmetic of the collective: no device needed) against oracle.row_shard_plan."""
    from lantern_amd import capi

    mine = capi.row_shard_plan(sizes, 21, 8, *plan)
    theirs = oracle.row_shard_plan(sizes, oracle.levels_for(21, 0, sum(sizes), 8), *plan)
    assert mine == [(f, c, list(s)) for f, c, s in theirs]
