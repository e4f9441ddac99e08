"""Row-partitioned search (SURVEY.md 8e as written: vectors sharded by row, per-shard candidates, all-gather, merge): every
rank owns an index over a disjoint share of the rows; lantern_gpu_search_partitioned searches all shares, all-gathers the
per-rank top-k in HBM and merges on the device.  The test box has one GPU, so the ranks are threads of this process over the
in-process hub (as tests/test_gpu_sharded_build.py); what is checked is the collective's arithmetic: the answer is exactly
the (distance, label)-ordered merge of what each share's own search returns, on every rank, and it is at least as good a
neighbour list as one index over all the rows returns.
"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from lantern_amd import capi

    capi.lib()
    assert capi.device_count() > 0, "no HIP device: the gpu tests need a real MI355X"
    return capi


def f2ord(x):
    """the library's order-preserving map of a float's bits (device_common.hpp f2ord)"""
    b = int(np.float32(x).view(np.uint32))
    return (b ^ 0x80000000) if b < 0x80000000 else (~b & 0xFFFFFFFF)


def merged(parts, k):
    """numpy restatement of the merge: per query the k smallest (distance, label) over all shares' valid entries."""
    nq = parts[0][0].shape[0]
    out_l, out_d, out_c = np.zeros((nq, k), np.uint64), np.full((nq, k), np.inf, np.float32), np.zeros(nq, np.uint32)
    for q in range(nq):
        cand = []
        for lab, dist, cnt in parts:
            cand += [(float(dist[q, i]), int(lab[q, i])) for i in range(int(cnt[q]))]
        cand.sort(key=lambda t: (f2ord(t[0]), t[1]))
        cand = cand[:k]
        out_c[q] = len(cand)
        for i, (d, l) in enumerate(cand):
            out_l[q, i], out_d[q, i] = l, d
    return out_l, out_d, out_c


@pytest.mark.parametrize("metric,world,n,d,k", [("l2sq", 2, 6000, 48, 10), ("cos", 3, 4500, 96, 7), ("l2sq", 3, 50, 16, 20), ("hamming", 2, 3000, 8, 10)])
def test_partitioned_search_is_the_merge_of_the_shares(capi, metric, world, n, d, k):
    rng = np.random.default_rng(n + d)
    if metric == "hamming":
        base = rng.integers(0, 2**32, size=(n, d), dtype=np.uint32)
        queries = rng.integers(0, 2**32, size=(40, d), dtype=np.uint32)
    else:
        base = rng.standard_normal((n, d), dtype=np.float32)
        queries = rng.standard_normal((40, d), dtype=np.float32)
    labels = np.arange(n, dtype=np.uint64) + 1
    # ragged shares; with n = 50 and k = 20 some shares hold fewer than k rows (short per-rank lists), one can be empty
    cuts = [0] + sorted(rng.choice(np.arange(1, n), size=world - 1, replace=False).tolist()) + [n]
    if n == 50:
        cuts = [0, 0, 30, 50]  # rank 0 owns nothing
    shares = []
    for r in range(world):
        ix = capi.GpuIndex(metric, d, M=12, ef_construction=48, ef=64, seed=5)
        if cuts[r + 1] > cuts[r]:
            ix.add_many(labels[cuts[r]:cuts[r + 1]], base[cuts[r]:cuts[r + 1]])
            ix.flush()
        shares.append(ix)
    parts = []
    for ix in shares:
        if len(ix):
            parts.append(ix.search_batch(queries, k))
        else:
            parts.append((np.zeros((40, k), np.uint64), np.full((40, k), np.inf, np.float32), np.zeros(40, np.uint32)))
    want = merged(parts, k)
    comms = capi.Comm.local_world(world)
    got, errs = [None] * world, []

    def run(r):
        try:
            comms[r].set_timeout(60)
            got[r] = shares[r].search_partitioned(comms[r], queries, k)
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errs.append((r, repr(e)))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for r in range(world):
        lab, dist, cnt = got[r]
        assert np.array_equal(cnt, want[2]), f"rank {r}: counts"
        assert np.array_equal(lab, want[0]), f"rank {r}: labels"
        assert np.array_equal(dist, want[1]), f"rank {r}: distances"
    # against ONE index over all the rows: the partitioned answer is no worse a neighbour list (exact truth as the referee)
    if n >= 1000 and metric != "hamming":
        whole = capi.GpuIndex(metric, d, M=12, ef_construction=48, ef=64, seed=5)
        whole.add_many(labels, base)
        whole.flush()
        truth, _ = whole.exact_search(queries, k)
        w_lab, _, _ = whole.search_batch(queries, k)
        slot_of = lambda L: L.astype(np.int64) - 1  # noqa: E731 -- labels are slot + 1 here
        rec = lambda L: float(np.mean([len(set(a.tolist()) & set(t.tolist())) / k for a, t in zip(slot_of(L), truth.astype(np.int64))]))  # noqa: E731
        assert rec(got[0][0]) >= rec(w_lab) - 0.02
