"""quant_bits = 1 on real[] input (options.c:154-155, test/sql/hnsw_sq.sql) and pq = true indexes (build.c:497-500,
scan.c:75-81, pqtable.c:194-240) on the device.  PARITY UNPINNED BY THE REFERENCE: both arithmetic paths live in the
un-vendored usearch fork and the reference's own expected outputs for them (hnsw_sq.out, hnsw_pq*.out) need sift1k, which
is downloaded at test time.  What is checked is the restated semantics, bit for bit:
  b1   bit i = (x_i > 0); the l2sq distance of {0, 1} vectors IS their Hamming distance -> identical to the oracle's
       Hamming index over the packed bits (graph, ids, distances, D / E); the cosine of {0, 1} vectors is three popcounts;
  pq   every stored vector is replaced by its quantisation (per subvector the nearest centroid, first minimum wins);
       all distances are distances to / between DECODED vectors -> identical to the oracle's f32 index over the decoded
       rows; the file carries num_subvectors code bytes per node."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from lantern_amd import capi

    capi.lib()
    assert capi.device_count() > 0
    return capi


def pack_bits_msb_first(x):
    """usearch's cast to b1x8: bit i = (x_i > 0), most significant bit of each byte first; returned as u32 words (LE)."""
    bits = (np.asarray(x) > 0).astype(np.uint8)
    n, d = bits.shape
    pad = (-d) % 32
    if pad:
        bits = np.concatenate([bits, np.zeros((n, pad), np.uint8)], axis=1)
    return np.packbits(bits, axis=1, bitorder="big").view(np.uint32)


@pytest.mark.parametrize("metric", ["l2sq", "cos"])
@pytest.mark.parametrize("n,d,M,efc", [(3000, 128, 8, 48), (1500, 768, 16, 64), (800, 100, 4, 24)])
def test_quant_bits_1_on_real_input_is_the_hamming_index_of_the_sign_bits(capi, oracle, tmp_path, n, d, M, efc, metric):
    # l2sq over {0, 1} values IS the Hamming distance; cosine over them is 1 - |a & b| / (sqrt |a| sqrt |b|) (the oracle's cos_b1).
    # The reference builds both (scripts/integration_tests.py:664-667: metric x quant_bits); what the fork's cosine computes on bits
    # cannot be read off the tree: PARITY UNPINNED for that half.
    bit_metric = "hamming" if metric == "l2sq" else "cos_b1"
    rng = np.random.default_rng(n + d)
    base = (rng.standard_normal((n, d)) - 0.1).astype(np.float32)  # "v_transformed": roughly centred real values
    base[5, :7] = [0.0, -0.0, np.nan, 1e-30, -1e-30, np.inf, -np.inf]  # only strictly positive values set a bit; NaN does not
    queries = rng.standard_normal((200, d)).astype(np.float32)
    labels = np.arange(n, dtype=np.uint64) + 1
    ix = capi.GpuIndex(metric, d, M=M, ef_construction=efc, ef=40, seed=3, quantization="b1")
    ix.set_add_batch(256, 16)
    ix.add_many(labels, base)
    words = pack_bits_msb_first(base)
    qwords = pack_bits_msb_first(queries)
    # the stored rows are the packed sign bits
    g = ix.export_graph(with_vectors=True)
    assert np.array_equal(g["vectors"], words)
    # the graph is the oracle's Hamming graph over those bits (same batch plan), edge for edge
    ora = oracle.OracleIndex(bit_metric, words.shape[1], M=M, ef_construction=efc, ef=40, seed=3)
    ora.add_planned(labels, words, 256, 16)
    o = ora.export_graph()
    if d % 32 == 0:  # the oracle's hamming metric takes whole words; ragged bit counts are compared through search below
        for key in ("levels", "nbr0", "upper_off", "upper_nbr"):
            assert np.array_equal(g[key], o[key]), key
    on_same = oracle.OracleIndex.from_graph(bit_metric, words, g, M, efc, 40, 3, oracle.SUM_SEQ)
    o_lab, o_dist, o_slot, o_D, o_E = on_same.search_batch(qwords, 10, 40, 4)
    lab, dist, cnt = ix.search_batch(queries, 10)
    assert np.array_equal(lab, o_lab) and np.array_equal(dist, o_dist)
    # one query through usearch_search_ef (f32 in), the exact k-NN, the pair kernel: all on the bits
    l1, d1 = ix.search(queries[0], 10)
    assert np.array_equal(l1, lab[0]) and np.array_equal(d1, dist[0])
    t_slots, t_d = ix.exact_search(queries[:32], 5)
    b_ids, b_d = oracle.bruteforce(words, qwords[:32], 5, bit_metric, oracle.SUM_SEQ, 4)
    assert np.array_equal(t_slots, b_ids) and np.array_equal(t_d, b_d)
    assert np.array_equal(ix.distance_gather(queries[3], t_slots[3]), t_d[3])
    # file: ceil(d / 8) bytes per vector, header kinds l2sq ('e') + b1x8 (1); round trip
    blob = ix.save_buffer()
    assert blob[13:14] == (b"e" if metric == "l2sq" else b"c") and blob[14] == 1
    assert len(blob) == 136 + sum(10 + (4 + 2 * M * 6) + int(l) * (4 + M * 6) + (d + 7) // 8 for l in g["levels"])
    again = capi.GpuIndex(metric, d, M=M, ef_construction=efc, ef=40, seed=3, quantization="b1")
    again.load_buffer(blob)
    assert again.checksum() == ix.checksum() and np.array_equal(again.search_batch(queries, 10)[0], lab)


def code_row_stride(S):
    """Bytes between the code rows of a compact pq index in HBM: num_subvectors rounded up to 16, and 65 .. 127 widened to 128 (one
    cache line per row: index.cpp pq_compact_locked)."""
    s16 = (S + 15) // 16 * 16
    return 128 if 64 < s16 < 128 else s16


def make_codebook(rng, base, S, C):
    """A codebook in the layout Lantern hands to usearch_init (pqtable.c:194-240): [C][d], row c = centroid c of every
    subvector, concatenated.  Centroids are sampled data points per subvector (what k-means++ starts from)."""
    n, d = base.shape
    sub = d // S
    cb = np.zeros((C, d), dtype=np.float32)
    for s in range(S):
        pick = rng.choice(n, size=C, replace=False)
        cb[:, s * sub:(s + 1) * sub] = base[pick, s * sub:(s + 1) * sub]
    return cb


def quantize_reference(oracle, base, cb, S, metric):
    """product_quantization.c:207-240 (quantize_vector): per subvector the centroid with the smallest distance under the
    index metric, strict `<` scan -> first minimum; distances in the pair kernel's reduction order."""
    n, d = base.shape
    sub = d // S
    codes = np.zeros((n, S), dtype=np.uint8)
    for s in range(S):
        ids, _ = oracle.bruteforce(np.ascontiguousarray(cb[:, s * sub:(s + 1) * sub]), np.ascontiguousarray(base[:, s * sub:(s + 1) * sub]), 1, metric,
                                   oracle.SUM_WAVE64, 4)
        codes[:, s] = ids[:, 0]
    dec = np.zeros_like(base)
    for s in range(S):
        dec[:, s * sub:(s + 1) * sub] = cb[codes[:, s], s * sub:(s + 1) * sub]
    return codes, dec


@pytest.mark.parametrize("metric,n,d,S,C,M,efc", [("l2sq", 2500, 128, 32, 256, 8, 48), ("cos", 1500, 768, 96, 64, 16, 64), ("l2sq", 900, 60, 6, 10, 4, 24)])
def test_pq_index_is_the_index_of_the_decoded_vectors(capi, oracle, metric, n, d, S, C, M, efc):
    rng = np.random.default_rng(n + d + S)
    base = rng.standard_normal((n, d), dtype=np.float32)
    queries = rng.standard_normal((150, d), dtype=np.float32)
    labels = np.arange(n, dtype=np.uint64) + 1
    cb = make_codebook(rng, base, S, C)
    cb[1] = cb[0]  # a duplicated centroid: the FIRST minimum must win (strict `<` in the reference loop)
    codes, dec = quantize_reference(oracle, base, cb, S, metric)
    assert not np.any(codes == 1) or C < 2
    ix = capi.GpuIndex(metric, d, M=M, ef_construction=efc, ef=40, seed=5, pq_codebook=cb, num_subvectors=S)
    ix.set_add_batch(256, 16)
    ix.add_many(labels[:n // 2], base[:n // 2])  # a bulk add ...
    for i in range(n // 2, n // 2 + 20):         # ... single adds (usearch_add, one tuple at a time) ...
        ix.add(labels[i], base[i])
    ix.add_many(labels[n // 2 + 20:], base[n // 2 + 20:])
    assert np.array_equal(ix.export_codes(), codes)
    g = ix.export_graph(with_vectors=True)
    assert np.array_equal(g["vectors"], dec)  # HBM holds the decodings
    # the oracle's plain f32 index over the decoded rows, same batch plan: the same graph, edge for edge
    ora = oracle.OracleIndex(metric, d, M=M, ef_construction=efc, ef=40, seed=5, sum_mode=oracle.SUM_WAVE64)
    plan = [n // 2, 20, n - n // 2 - 20]
    at = 0
    for cnt in plan:  # the flush points of the calls above: the bulk add, the 20 buffered single adds, the second bulk add
        ora.add_planned(labels[at:at + cnt], dec[at:at + cnt], 256, 16)
        at += cnt
    o = ora.export_graph()
    for key in ("levels", "nbr0", "upper_off", "upper_nbr"):
        assert np.array_equal(g[key], o[key]), key
    # search: f32 queries against decoded rows
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(queries, 10, 40, 4)
    lab, dist, cnt = ix.search_batch(queries, 10)
    assert np.array_equal(lab, o_lab) and np.array_equal(dist, o_dist)
    # file: num_subvectors code bytes per node (usearch_storage.cpp:29-31); load decodes them again
    blob = ix.save_buffer()
    assert len(blob) == 136 + sum(10 + (4 + 2 * M * 6) + int(l) * (4 + M * 6) + S for l in g["levels"])
    off = 136
    for i in range(5):
        lv = int(g["levels"][i])
        vec_off = off + 10 + (4 + 2 * M * 6) + lv * (4 + M * 6)
        assert blob[vec_off:vec_off + S] == codes[i].tobytes()
        off = vec_off + S
    again = capi.GpuIndex(metric, d, M=M, ef_construction=efc, ef=40, seed=5, pq_codebook=cb, num_subvectors=S)
    again.load_buffer(blob)
    assert again.checksum() == ix.checksum()
    g2 = again.export_graph(with_vectors=True)
    assert np.array_equal(g2["vectors"], dec) and np.array_equal(again.search_batch(queries, 10)[0], lab)
    m = ix.metadata()
    assert m.init_options.pq and m.init_options.num_subvectors == S and m.init_options.num_centroids == C
    with pytest.raises(capi.LanternGpuError, match="must divide"):
        capi.GpuIndex(metric, d, pq_codebook=cb, num_subvectors=7 if d % 7 else 11)


# ------------------------------------------------------------------------------------------------------------------
# The COMPACT form of a pq index: the decodings leave HBM.  Subvectors of whole 16-byte chunks (dimensions / num_subvectors a
# multiple of 4): searches DECODE rows on the fly from the L2-resident centroid tables (device_common.hpp PqdRow) -- the expanded
# form's arithmetic, so every answer, distance bit and counter equals the expanded form's and the oracle's over the decoded rows.
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("metric,n,d,S,C,M,ef", [("l2sq", 3000, 128, 32, 256, 8, 40), ("cos", 2000, 768, 96, 64, 16, 64), ("l2sq", 1500, 96, 4, 200, 8, 100),
                                                  ("cos", 1800, 1536, 96, 256, 16, 48), ("l2sq", 1200, 320, 20, 16, 5, 33)])
def test_compact_pq_index_decoding_rows_on_the_fly_is_the_expanded_index_bit_for_bit(capi, oracle, metric, n, d, S, C, M, ef, monkeypatch):
    from lantern_amd import hip

    rng = np.random.default_rng(n + d + S)
    base = rng.standard_normal((n, d), dtype=np.float32)
    nq = 300
    queries = rng.standard_normal((nq, d), dtype=np.float32)
    cb = make_codebook(rng, base, S, C)
    ix = capi.GpuIndex(metric, d, M=M, ef_construction=48, ef=ef, seed=5, pq_codebook=cb, num_subvectors=S)
    ix.set_add_batch(256, 16)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    g = ix.export_graph(with_vectors=True)

    def device(index, count):
        dq = hip.Buffer.from_numpy(hip.padded_rows(queries[:count], False))
        lab, dist, D, E = hip.Buffer(count * 10 * 8), hip.Buffer(count * 10 * 4), hip.Buffer(count * 8), hip.Buffer(count * 8)
        index.search_batch_device(dq.ptr, count, 10, 0, 0, lab.ptr, dist.ptr, None, None, D.ptr, E.ptr)
        hip.synchronize()
        return lab.download((count, 10), np.uint64), dist.download((count, 10), np.float32), D.download(count, np.uint64), E.download(count, np.uint64)

    want = device(ix, nq)  # the expanded form: the f32 walk over the decodings
    ora = oracle.OracleIndex.from_graph(metric, g["vectors"], g, M, 48, ef, 5, oracle.SUM_WAVE64)
    o_lab, o_dist, _, o_D, o_E = ora.search_batch(queries, 10, ef, 4)
    assert np.array_equal(want[0], o_lab) and np.array_equal(want[1].view(np.uint32), o_dist.view(np.uint32))
    ix.pq_compact()
    assert ix.memory_usage()[0] == n * code_row_stride(S)
    for spec in (None, "0", "2"):  # the automatic shape, the bandwidth-bound walk, the latency-bound one
        if spec is None:
            monkeypatch.delenv("LANTERN_GPU_SPEC", raising=False)
        else:
            monkeypatch.setenv("LANTERN_GPU_SPEC", spec)
        for count in (nq, 1, 37):
            got = device(ix, count)
            assert np.array_equal(got[0], want[0][:count]) and np.array_equal(got[1].view(np.uint32), want[1][:count].view(np.uint32)), (spec, count)
            assert np.array_equal(got[2], o_D[:count]) and np.array_equal(got[3], o_E[:count]), (spec, count)
    monkeypatch.delenv("LANTERN_GPU_SPEC", raising=False)
    l1, d1 = ix.search(queries[0], 10)  # usearch_search_ef, and the host-buffer batch
    assert np.array_equal(l1, want[0][0]) and np.array_equal(d1, want[1][0])
    hl, hd, _ = ix.search_batch(queries[:33], 10)
    assert np.array_equal(hl, want[0][:33]) and np.array_equal(hd, want[1][:33])


# ------------------------------------------------------------------------------------------------------------------
# Subvectors of any other width (or LANTERN_GPU_PQ_ADC=1): ADC over the code bytes (search_adc_kernel.hip).  The summation order
# of ADC is the device's own definition (the fork's PQ metric is not in the reference tree: PARITY UNPINNED);
# the oracle restates it (lo_set_pq_view) and must agree bit for bit; against the decoded-row path distances agree to 1e-5.
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("metric,n,d,S,C,M,ef", [("l2sq", 3000, 128, 32, 256, 8, 40), ("cos", 2000, 768, 96, 64, 16, 64), ("l2sq", 1500, 60, 6, 10, 4, 100),
                                                  ("cos", 1800, 256, 128, 256, 16, 48)])
def test_compact_pq_index_searches_by_adc_over_the_code_bytes(capi, oracle, metric, n, d, S, C, M, ef, monkeypatch):
    from lantern_amd import hip

    monkeypatch.setenv("LANTERN_GPU_PQ_ADC", "1")  # (two of the four shapes would decode on the fly otherwise)
    rng = np.random.default_rng(n + d + S)
    base = rng.standard_normal((n, d), dtype=np.float32)
    nq = 200
    queries = rng.standard_normal((nq, d), dtype=np.float32)
    labels = np.arange(n, dtype=np.uint64) + 1
    cb = make_codebook(rng, base, S, C)
    ix = capi.GpuIndex(metric, d, M=M, ef_construction=48, ef=ef, seed=5, pq_codebook=cb, num_subvectors=S)
    ix.set_add_batch(256, 16)
    ix.add_many(labels[: n - 50], base[: n - 50])
    codes = ix.export_codes()
    g = ix.export_graph(with_vectors=True)
    dec = g["vectors"]
    lab_dec, dist_dec, _ = ix.search_batch(queries, 10)
    blob = ix.save_buffer()
    rows_before, other = ix.memory_usage()
    # ---- compact: num_subvectors bytes per row (padded to 16) instead of 4 * d
    ix.pq_compact()
    rows_after, other_after = ix.memory_usage()
    S16 = code_row_stride(S)
    assert rows_after == (n - 50) * S16 and rows_before >= (n - 50) * d * 4 and other_after == other
    dq = hip.Buffer.from_numpy(hip.padded_rows(queries, False))
    lab, dist, slot = hip.Buffer(nq * 10 * 8), hip.Buffer(nq * 10 * 4), hip.Buffer(nq * 10 * 4)
    D, E = hip.Buffer(nq * 8), hip.Buffer(nq * 8)
    ix.search_batch_device(dq.ptr, nq, 10, 0, 0, lab.ptr, dist.ptr, slot.ptr, None, D.ptr, E.ptr)
    hip.synchronize()
    lab, dist, D, E = lab.download((nq, 10), np.uint64), dist.download((nq, 10), np.float32), D.download(nq, np.uint64), E.download(nq, np.uint64)
    # the oracle's ADC on the same graph: identical ids, distance bits, evaluation and expansion counts
    ora = oracle.OracleIndex.from_graph(metric, dec, g, M, 48, ef, 5, oracle.SUM_WAVE64)
    ora.set_pq_view(cb, codes)
    o_lab, o_dist, o_slot, o_D, o_E = ora.search_batch(queries, 10, ef, 4)
    assert np.array_equal(lab, o_lab) and np.array_equal(dist, o_dist)
    assert np.array_equal(D, o_D) and np.array_equal(E, o_E)
    # both walks of the ADC kernel (the default picks by table size: the lone-query shape of walk_spec.hpp when the table leaves room
    # for one workgroup per CU, the classic 8-wave walk otherwise): the same ids, distance bits, D and E
    for forced in ("0", "1"):
        monkeypatch.setenv("LANTERN_GPU_ADC_SPEC", forced)
        lab2, dist2, D2, E2 = hip.Buffer(nq * 10 * 8), hip.Buffer(nq * 10 * 4), hip.Buffer(nq * 8), hip.Buffer(nq * 8)
        ix.search_batch_device(dq.ptr, nq, 10, 0, 0, lab2.ptr, dist2.ptr, None, None, D2.ptr, E2.ptr)
        hip.synchronize()
        assert np.array_equal(lab2.download((nq, 10), np.uint64), o_lab) and np.array_equal(dist2.download((nq, 10), np.float32), o_dist), forced
        assert np.array_equal(D2.download(nq, np.uint64), o_D) and np.array_equal(E2.download(nq, np.uint64), o_E), forced
    monkeypatch.delenv("LANTERN_GPU_ADC_SPEC")
    # against the decoded-row path: the same neighbours up to near-ties, distances within the north-star's tolerance
    same = lab == lab_dec
    assert same.mean() > 0.97
    assert np.all(np.abs(dist[same] - dist_dec[same]) <= 1e-5 * np.maximum(1.0, np.abs(dist_dec[same])) + 1e-6)
    # one query through usearch_search_ef, and the host-buffer batch entry point
    l1, d1 = ix.search(queries[0], 10)
    assert np.array_equal(l1, lab[0]) and np.array_equal(d1, dist[0])
    hl, hd, _ = ix.search_batch(queries[:33], 10)
    assert np.array_equal(hl, lab[:33]) and np.array_equal(hd, dist[:33])
    # the file is the codes' either way
    assert ix.save_buffer() == blob
    # adding decodes the rows back first (and the graph goes on exactly as if it had never been compact)
    ix.add_many(labels[n - 50:], base[n - 50:])
    twin = capi.GpuIndex(metric, d, M=M, ef_construction=48, ef=ef, seed=5, pq_codebook=cb, num_subvectors=S)
    twin.set_add_batch(256, 16)
    twin.add_many(labels[: n - 50], base[: n - 50])
    twin.add_many(labels[n - 50:], base[n - 50:])
    assert ix.checksum() == twin.checksum() and ix.memory_usage()[0] >= n * d * 4
    assert np.array_equal(ix.search_batch(queries, 10)[0], twin.search_batch(queries, 10)[0])
    # a loaded index compacts itself on request (the scan-side mirror of a built index)
    monkeypatch.setenv("LANTERN_GPU_PQ_COMPACT", "1")
    again = capi.GpuIndex(metric, d, M=M, ef_construction=48, ef=ef, seed=5, pq_codebook=cb, num_subvectors=S)
    again.load_buffer(blob)
    assert again.memory_usage()[0] == (n - 50) * S16
    al, ad, _ = again.search_batch(queries, 10)
    assert np.array_equal(al, lab) and np.array_equal(ad, dist)


def test_pq_build_through_the_indexing_server(capi):
    """CREATE INDEX ... WITH (external=true, pq=true): the codebook travels centroid by centroid before the rows
    (external_index_socket.c:304-320,475-478; server.rs:107-127)."""
    from tests import index_client

    rng = np.random.default_rng(77)
    n, d, S, C = 1200, 64, 16, 32
    base = rng.standard_normal((n, d), dtype=np.float32)
    cb = make_codebook(rng, base, S, C)
    srv = capi.IndexServer()
    try:
        added, blob = index_client.build_index(srv.host, srv.port, 3, d, [r.tobytes() for r in base], np.arange(n) + 1, m=8, efc=40, ef=32,
                                               codebook=cb, num_subvectors=S)
    finally:
        srv.stop()
    assert added == n
    direct = capi.GpuIndex("l2sq", d, M=8, ef_construction=40, ef=32, seed=42, pq_codebook=cb, num_subvectors=S)
    direct.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    assert direct.save_buffer() == blob


@pytest.mark.parametrize("quant", ["f16", "i8", "b1"])
def test_bulk_adds_quantise_on_the_device_exactly_as_single_adds_do_on_the_host(capi, quant):
    """A bulk add uploads f32 rows and converts them on the device; usearch_add (one tuple) converts on the host.  Same rule,
    same bits -- including denormal halves, NaN, infinities, signed zeros, the i8 clamp and values that sit on a rounding tie."""
    rng = np.random.default_rng(5)
    n, d = 300, 70  # 70: a ragged tail in every storage kind (35 words f16, 17.5 words i8, 2.2 words b1)
    base = rng.standard_normal((n, d)).astype(np.float32)
    base[0, :12] = [0.0, -0.0, np.nan, np.inf, -np.inf, 1e-7, -1e-7, 3e-5, 65504.0, 70000.0, 1.0009765625, 0.00048828125]
    base[1, :8] = [0.999, 1.0, 1.001, -1.0, -1.009, 0.0149, 0.015, -0.0199]  # i8: x * 100 truncates toward zero, clamps at +-100
    base[2] = np.float32(2.0) ** rng.integers(-26, 4, size=d)  # halves from deep denormal to normal
    base[3] *= 1e-6
    labels = np.arange(n, dtype=np.uint64) + 1
    bulk = capi.GpuIndex("l2sq", d, M=4, ef_construction=16, seed=1, quantization=quant)
    bulk.set_add_batch(64, 16)
    bulk.add_many(labels, base)  # n >= add_batch_max: the device converts
    single = capi.GpuIndex("l2sq", d, M=4, ef_construction=16, seed=1, quantization=quant)
    single.set_add_batch(64, 16)
    for l, row in zip(labels, base):  # the host converts (pad_row)
        single.add(l, row)
    a, b = bulk.export_graph(with_vectors=True), single.export_graph(with_vectors=True)
    assert np.array_equal(a["vectors"].view(np.uint8), b["vectors"].view(np.uint8))
    if quant == "f16":
        with np.errstate(over="ignore"):
            assert np.array_equal(a["vectors"].view(np.uint16), base.astype(np.float16).view(np.uint16))
