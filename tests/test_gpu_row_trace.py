"""lantern_gpu_search_row_trace: the instrumented walk's memory-object trace (the input of bench.py's DRAM model) is the walk the
oracle walks -- per query, as many row entries as distance evaluations (D), as many level-0 list entries as expansions (E), every
evaluated row exactly once (the visited set), and the answers / D / E of a traced launch are those of an untraced one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from lantern_amd import build, capi

    build.build()
    assert capi.device_count() > 0
    return capi


@pytest.fixture(scope="module")
def oracle():
    from oracle import binding

    binding.build()
    return binding


@pytest.mark.parametrize("metric,n,d", [("l2sq", 6000, 768), ("cos", 4000, 512), ("l2sq", 5000, 128)])
def test_trace_is_the_oracle_walk(capi, oracle, metric, n, d):
    from lantern_amd import hip

    rng = np.random.default_rng(n + d)
    base = rng.standard_normal((n, d), dtype=np.float32)
    nq, k, ef, cap = 300, 10, 64, 4096
    queries = rng.standard_normal((nq, d), dtype=np.float32)
    gpu = capi.GpuIndex(metric, d, M=16, ef_construction=64, ef=ef, seed=5)
    gpu.set_add_batch(512, 16)
    gpu.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    gpu.flush()
    ora = oracle.OracleIndex.from_graph(metric, base, gpu.export_graph(), 16, 64, ef, 5, oracle.SUM_WAVE64)
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(queries, k)
    rows = gpu.device_query_rows(queries)
    dq = hip.Buffer.from_numpy(rows)
    lab, dist, D, E = hip.Buffer(nq * k * 8), hip.Buffer(nq * k * 4), hip.Buffer(nq * 8), hip.Buffer(nq * 8)
    gpu.set_search_shape(4)  # the bandwidth-bound walk (the one the instrumented instantiation exists for)
    gpu.row_trace_begin(nq, cap)
    gpu.search_batch_device(dq.ptr, nq, k, ef, 0, lab.ptr, dist.ptr, None, None, D.ptr, E.ptr, query_stride=rows.strides[0])
    hip.synchronize()
    assert gpu.last_search_grid() >= 1
    trace, counts = gpu.row_trace_end()
    assert np.array_equal(lab.download((nq, k), np.uint64), o_lab) and np.array_equal(dist.download((nq, k), np.float32), o_dist)
    gD, gE = D.download(nq, np.uint64), E.download(nq, np.uint64)
    assert np.array_equal(gD, o_D) and np.array_equal(gE, o_E)
    assert counts.max() <= cap
    for q in range(nq):
        t = trace[q, :counts[q]]
        kind = t >> 30
        rows_q = t[kind < 2]
        assert len(rows_q) == int(o_D[q]), q                       # one entry per distance evaluation
        assert int((kind == 2).sum()) >= int(o_E[q]), q            # level-0 lists: the expansions (+ the last greedy step at level 0: none)
        level0 = rows_q  # every evaluated row at most once per level; the descent may evaluate a row again on a lower level
        assert len(np.unique(level0)) >= len(level0) - int((kind == 3).sum()) * 16
        assert t.max() & 0x3FFFFFFF < n
    # an untraced launch afterwards: the same answers, nothing recorded
    gpu.search_batch_device(dq.ptr, nq, k, ef, 0, lab.ptr, dist.ptr, None, None, D.ptr, E.ptr, query_stride=rows.strides[0])
    hip.synchronize()
    assert np.array_equal(lab.download((nq, k), np.uint64), o_lab) and np.array_equal(D.download(nq, np.uint64), o_D)
