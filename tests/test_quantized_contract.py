"""The reference's contract for quantised indexes (scripts/integration_tests.py:176-265, test/sql/hnsw_sq.sql:17-49,
src/hnsw/options.c:137-158) on a SIFT-shaped synthetic: metric in {l2sq, cos} x quant_bits in {32, 16, 8, 1}, M = 8, queries = the
table's own rows [1, 3, 5, 10, 20, 55, 72, 11].  See tests/quant_contract.py for the clauses.

CPU suite: the oracle (sequential usearch_add, as ambuild does).  `-m gpu`: the device index through `lantern_scan_*`
(the amgettuple shim) with the SQL distance functions `lantern_{l2sq,cos}_dist` recomputing the returned rows' distances.
1000 rows is the reference's size (sift_base1k); 10 000 rows is the same contract one order of magnitude up.
This is the one check the REFERENCE holds for the i8 rule trunc(clamp(100 x)) and for cosine over b1 storage: neither
rule can be read off the tree (the arithmetic is in the un-vendored usearch fork), but an index whose quantisation broke
neighbourhoods would fail these floors against the exact scan over the UNQUANTISED column.
"""
import numpy as np
import pytest

from tests import quant_contract as qc

COMBOS = [(m, b) for m in ("l2sq", "cos") for b in (32, 16, 8, 1)]


def oracle_view(oracle, col, metric, bits):
    """(rows as the oracle stores them, query conversion, oracle metric, summation mode) for one quantisation."""
    if bits == 32:
        return col, (lambda q: q), metric, oracle.SUM_SEQ
    if bits == 16:
        return oracle.round_f16(col), oracle.round_f16, metric, oracle.SUM_WAVE64_F16
    if bits == 8:
        return oracle.quantize_i8(col), oracle.quantize_i8, metric, oracle.SUM_I8
    return qc.pack_bits_msb_first(col), (lambda q: qc.pack_bits_msb_first(q)[0]), ("hamming" if metric == "l2sq" else "cos_b1"), oracle.SUM_SEQ


@pytest.mark.parametrize("n", [1000, 10000])
@pytest.mark.parametrize("metric,bits", COMBOS)
def test_oracle_meets_the_reference_quantised_contract(oracle, metric, bits, n):
    v = qc.sift_like(n)
    col = qc.table_column(v, bits)
    rows, conv, m, sm = oracle_view(oracle, col, metric, bits)
    ix = oracle.OracleIndex(m, rows.shape[1], M=qc.M, ef_construction=qc.EFC, ef=qc.EF, seed=42, sum_mode=sm)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, rows)  # one usearch_add per tuple (build.c:128)
    rec = qc.check_contract(col, metric, bits, lambda q: ix.search(conv(q), qc.LIMIT)[0], lambda a, b: oracle.distance(a, b, metric))
    assert len(rec) == len(qc.QUERY_IDS)


def test_sift_like_has_sift_marginals():
    v = qc.sift_like(1000)
    assert v.shape == (1000, 128) and v.min() >= 0 and v.max() <= 255 and np.array_equal(v, np.floor(v))
    assert 15 < v.mean() < 45 and (v > 150).mean() < 0.01 and 0.05 < (v > 50).mean() < 0.4
    assert len({r.tobytes() for r in v}) == 1000  # no duplicate rows: "first result is the query's row" is well defined
    t = qc.table_column(v, 8)
    assert t.dtype == np.float32 and np.isclose(t.min(), -0.5) and t.max() <= 2.05


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1000, 10000])
@pytest.mark.parametrize("metric,bits", COMBOS)
def test_device_meets_the_reference_quantised_contract(oracle, metric, bits, n):
    from lantern_amd import capi

    capi.lib()
    assert capi.device_count() > 0
    v = qc.sift_like(n)
    col = qc.table_column(v, bits)
    quant = {32: "f32", 16: "f16", 8: "i8", 1: "b1"}[bits]
    ix = capi.GpuIndex(metric, qc.DIM, M=qc.M, ef_construction=qc.EFC, ef=qc.EF, seed=42, quantization=quant)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, col)  # f32 in, quantised on add -- as Lantern hands rows over
    scan = capi.Scan(ix, init_k=qc.LIMIT)

    def index_scan(q):
        scan.rescan(q)
        return scan.fetch(qc.LIMIT)

    dist_fn = capi.l2sq_dist if metric == "l2sq" else capi.cos_dist
    rec = qc.check_contract(col, metric, bits, index_scan, dist_fn)
    # the device index is not merely above the floors: on its own graph it IS the oracle (ids and order), so the floors
    # hold for the arithmetic the oracle restates
    rows, conv, m, sm = oracle_view(oracle, col, metric, bits)
    wave = {32: oracle.SUM_WAVE64, 16: oracle.SUM_WAVE64_F16, 8: oracle.SUM_I8, 1: oracle.SUM_SEQ}[bits]
    g = ix.export_graph()
    same = oracle.OracleIndex.from_graph(m, rows, g, qc.M, qc.EFC, qc.EF, 42, wave)
    for qid in qc.QUERY_IDS:
        o_lab = same.search(conv(col[qid - 1]), qc.LIMIT)[0]
        assert [int(x) for x in o_lab] == index_scan(col[qid - 1])
    scan.end()
    assert min(rec.values()) >= qc.RECALL_FLOOR[bits]
