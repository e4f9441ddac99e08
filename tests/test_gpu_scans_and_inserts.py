"""Scans that share one resident index, launches on two streams, the aminsert path (usearch_add_external,
usearch_update_header, lantern_gpu_add_with_level) and the usearch-format header.  Needs an MI355X."""
import ctypes as C
import math
import struct

import threading

import numpy as np
import pytest

from tests.pg_pages import PageStore, graph_by_label

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from lantern_amd import capi

    capi.lib()
    assert capi.device_count() > 0, "no HIP device: the gpu tests need a real MI355X"
    return capi


# ------------------------------------------------------------------------------------------------------------------
# per-scan continuation state (scan.c:99: one usearch handle per scan in the reference)
# ------------------------------------------------------------------------------------------------------------------
def test_interleaved_scans_on_one_index_do_not_disturb_each_other(capi):
    rng = np.random.default_rng(21)
    n, d = 4000, 32
    base = rng.standard_normal((n, d), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", d, M=8, ef_construction=48, ef=32, seed=5)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    qa, qb, qc = rng.standard_normal((3, d), dtype=np.float32)

    def solo(q, init_k, limit):
        s = capi.Scan(ix, init_k=init_k)
        s.rescan(q)
        rows = s.fetch(limit)
        s.end()
        return rows

    want_a, want_b, want_c = solo(qa, 3, 120), solo(qb, 5, 90), solo(qc, 4, 40)
    assert len(set(want_a)) == 120 and len(set(want_b)) == 90
    # two cursors / the two sides of a nested loop: the scans advance in lock step, then one is re-armed half way
    sa, sb = capi.Scan(ix, init_k=3), capi.Scan(ix, init_k=5)
    sa.rescan(qa)
    sb.rescan(qb)
    got_a, got_b = [], []
    for i in range(120):
        got_a.append(sa.gettuple())
        if i < 90:
            got_b.append(sb.gettuple())
        if i == 50:  # a third scan starts and finishes in the middle; plain usearch_search_ef calls happen too
            assert solo(qc, 4, 40) == want_c
            ix.search(qc, 7)
            ix.search(qc, 7, streaming=True)
    assert got_a == want_a and got_b == want_b
    sb.rescan(qa)  # ldb_amrescan on an open scan: it starts over (with ITS init_k), the other scan is untouched
    assert sb.fetch(30) == solo(qa, 5, 30)
    assert sa.gettuple() == solo(qa, 3, 121)[120]
    # cursors expose the same contract without the paging shim
    c1, c2 = ix.cursor(), ix.cursor()
    l1, _ = c1.search(qa, 4)
    l2, _ = c2.search(qb, 4)
    n1, _ = c1.search(qa, 6, streaming=True)
    n2, _ = c2.search(qb, 6, streaming=True)
    ref_a, _ = ix.search(qa, 4)
    more_a, _ = ix.search(qa, 6, streaming=True)
    assert l1.tolist() == ref_a.tolist() and n1.tolist() == more_a.tolist()
    assert not set(l1.tolist()) & set(n1.tolist()) and not set(l2.tolist()) & set(n2.tolist())
    assert c1.seen == 10 and c2.seen == 10


def test_searches_on_two_streams_share_the_index_safely(capi, monkeypatch):
    """lantern_gpu_search_batch_device returns with its kernel still running; a launch on another stream must not share
    the per-workgroup visited bitmaps with it.  LANTERN_GPU_VIS_SLOTS=0 puts EVERY visit in those bitmaps."""
    from lantern_amd import hip

    monkeypatch.setenv("LANTERN_GPU_VIS_SLOTS", "0")
    rng = np.random.default_rng(22)
    n, d, k, nq = 30000, 64, 10, 3000
    base = rng.standard_normal((n, d), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", d, M=16, ef_construction=64, ef=64, seed=6)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    ix.flush()
    qs = [rng.standard_normal((nq, d), dtype=np.float32) for _ in range(2)]
    want = [ix.search_batch(q, k) for q in qs]
    streams = [hip.Stream(), hip.Stream()]
    dq = [hip.Buffer.from_numpy(hip.padded_rows(q, False)) for q in qs]
    out_l = [hip.Buffer(nq * k * 8) for _ in qs]
    out_d = [hip.Buffer(nq * k * 4) for _ in qs]
    for rounds in range(3):
        for i in (0, 1):  # queued back to back on different streams: each gets its own bitmap slab and they overlap
            ix.search_batch_device(dq[i].ptr, nq, k, 0, 0, out_l[i].ptr, out_d[i].ptr, None, None, None, None, streams[i].handle)
        hip.synchronize()
        for i in (0, 1):
            assert np.array_equal(out_l[i].download((nq, k), np.uint64), want[i][0])
            assert np.array_equal(out_d[i].download((nq, k), np.float32), want[i][1])
    # more launches in flight than slabs: the third and fourth queue behind the slab they reuse
    more = [hip.Stream() for _ in range(4)]
    o_l = [hip.Buffer(nq * k * 8) for _ in more]
    o_d = [hip.Buffer(nq * k * 4) for _ in more]
    for rounds in range(2):
        for j, st in enumerate(more):
            ix.search_batch_device(dq[j & 1].ptr, nq, k, 0, 0, o_l[j].ptr, o_d[j].ptr, None, None, None, None, st.handle)
        hip.synchronize()
        for j in range(len(more)):
            assert np.array_equal(o_l[j].download((nq, k), np.uint64), want[j & 1][0])
            assert np.array_equal(o_d[j].download((nq, k), np.float32), want[j & 1][1])


def test_the_lanes_of_the_host_buffer_search_run_side_by_side(capi):
    """lantern_gpu_search_batch_lane: four caller threads, one lane each, many rounds; every answer equals the plain batch search."""
    rng = np.random.default_rng(31)
    n, d, k = 20000, 32, 10
    base = rng.standard_normal((n, d), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", d, M=12, ef_construction=48, ef=48, seed=3)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    ix.flush()
    LANES = (0, 1, 2, 3)
    qs = [rng.standard_normal((700 + 300 * lane, d), dtype=np.float32) for lane in LANES]
    want = [ix.search_batch(q, k) for q in qs]
    errs = []

    def run(lane):
        try:
            for _ in range(20):
                lab, dist, cnt = ix.search_batch_lane(lane, qs[lane], k)
                assert np.array_equal(lab, want[lane][0]) and np.array_equal(dist, want[lane][1]) and np.array_equal(cnt, want[lane][2])
        except Exception as e:  # noqa: BLE001
            errs.append((lane, repr(e)))

    ts = [threading.Thread(target=run, args=(lane,)) for lane in LANES]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for bad in (8, -1):
        with pytest.raises(capi.LanternGpuError, match="lane must be in"):
            ix.search_batch_lane(bad, qs[0], k)


def test_inserts_and_searches_on_other_streams_are_ordered(capi, monkeypatch):
    """An insert batch mutates the graph: it runs behind every search in flight, and a search queued on another stream after
    it sees all of its rows -- without any host synchronisation in between."""
    from lantern_amd import hip

    monkeypatch.setenv("LANTERN_GPU_VIS_SLOTS", "0")
    rng = np.random.default_rng(29)
    n1, n2, d, k, nq = 20000, 6000, 48, 10, 2000
    base = rng.standard_normal((n1 + n2, d), dtype=np.float32)
    labels = np.arange(n1 + n2, dtype=np.uint64) + 1
    q = rng.standard_normal((nq, d), dtype=np.float32)
    whole = capi.GpuIndex("l2sq", d, M=12, ef_construction=48, ef=48, seed=8)
    whole.add_many(labels[:n1], base[:n1])
    whole.flush()
    want_before = whole.search_batch(q, k)
    whole.add_many(labels[n1:], base[n1:])
    whole.flush()
    want_after = whole.search_batch(q, k)
    assert not np.array_equal(want_before[0], want_after[0])
    ix = capi.GpuIndex("l2sq", d, M=12, ef_construction=48, ef=48, seed=8)
    ix.add_many(labels[:n1], base[:n1])
    ix.flush()
    sa, sb = hip.Stream(), hip.Stream()
    dq = hip.Buffer.from_numpy(hip.padded_rows(q, False))
    la, da, lb, db = hip.Buffer(nq * k * 8), hip.Buffer(nq * k * 4), hip.Buffer(nq * k * 8), hip.Buffer(nq * k * 4)
    ix.search_batch_device(dq.ptr, nq, k, 0, 0, la.ptr, da.ptr, None, None, None, None, sa.handle)  # in flight ...
    ix.add_many(labels[n1:], base[n1:])                                                           # ... queued behind it
    ix.search_batch_device(dq.ptr, nq, k, 0, 0, lb.ptr, db.ptr, None, None, None, None, sb.handle)  # ... and this behind the insert
    hip.synchronize()
    assert np.array_equal(la.download((nq, k), np.uint64), want_before[0]) and np.array_equal(da.download((nq, k), np.float32), want_before[1])
    assert np.array_equal(lb.download((nq, k), np.uint64), want_after[0]) and np.array_equal(db.download((nq, k), np.float32), want_after[1])
    assert ix.checksum() == whole.checksum()


# ------------------------------------------------------------------------------------------------------------------
# the insert path: caller-drawn levels (insert.c:32-46), usearch_add_external (insert.c:209), usearch_update_header
# ------------------------------------------------------------------------------------------------------------------
def pg_level(rng, M):
    """hnsw_generate_new_level (insert.c:32-46): floor(-ln(U) * 1/ln(M)) with U from the backend's PRNG."""
    u = 1.0 - rng.random()  # (0, 1]
    return int(-math.log(u) * (1.0 / math.log(M)))


def test_add_with_level_is_usearch_add_at_that_level(capi, oracle):
    rng = np.random.default_rng(23)
    n, extra, d, M = 800, 60, 40, 6
    base = rng.standard_normal((n + extra, d), dtype=np.float32)
    labels = np.arange(n + extra, dtype=np.uint64) + 10
    levels = [pg_level(rng, M) for _ in range(extra)]
    levels[7] = 5  # one insert raises the top level: the entry point moves (insert.c:214 then refreshes the header)
    gpu = capi.GpuIndex("l2sq", d, M=M, ef_construction=32, ef=32, seed=9)
    gpu.set_add_batch(1, 1)  # a backend inserts one tuple at a time
    gpu.add_many(labels[:n], base[:n])
    ora = oracle.OracleIndex("l2sq", d, M=M, ef_construction=32, ef=32, seed=9, sum_mode=oracle.SUM_WAVE64)
    ora.add_many(labels[:n], base[:n])
    for i, lv in enumerate(levels):
        gpu.add(labels[n + i], base[n + i], level=lv)
        ora.add(labels[n + i], base[n + i], level=lv)
    g, o = gpu.export_graph(), ora.export_graph()
    for key in ("levels", "nbr0", "upper_off", "upper_nbr", "labels"):
        assert np.array_equal(g[key], o[key]), key
    assert g["entry_slot"] == o["entry_slot"] == n + 7 and g["max_level"] == o["max_level"] == 5
    assert list(g["levels"][n:]) == levels
    with pytest.raises(capi.LanternGpuError, match="level out of range"):
        gpu.add(1, base[0], level=300)


def test_add_external_links_the_mirror_and_writes_the_pages(capi):
    rng = np.random.default_rng(24)
    n, extra, d, M = 1200, 40, 24, 5
    base = rng.standard_normal((n + extra, d), dtype=np.float32)
    labels = np.arange(n + extra, dtype=np.uint64) + 500
    levels = [pg_level(rng, M) for _ in range(extra)]
    levels[11] = 6  # raises the top level
    a = capi.GpuIndex("l2sq", d, M=M, ef_construction=40, ef=32, seed=4)
    a.set_add_batch(1, 1)
    a.add_many(labels[:n], base[:n])
    store = PageStore(capi, a.save_buffer(), d * 4, M)
    # --- the backend: attach (insert.c:142-151), insert tuple by tuple (insert.c:182-214)
    b = capi.GpuIndex("l2sq", d, M=M, ef_construction=40, ef=32, seed=4, retriever=store.retriever, retriever_mut=store.retriever_mut)
    b.view_mem_lazy(store.header)
    # usearch_size is the header's count; the mirror holds the nodes a walk can reach (a few have lost their last in-link)
    assert len(b) == n and n - 20 <= b.graph_info().size <= n
    header = store.header
    for i, lv in enumerate(levels):
        addr, slot = store.new_tuple(int(labels[n + i]), lv)
        store.mutated.clear()
        b.add_external(labels[n + i], base[n + i], addr, lv, slot)
        header = b.update_header(header)
        # exactly the nodes the new node linked to were re-written (and only through retriever_mut)
        _, _, lists, vec = store.node(slot)
        assert sorted(store.mutated) == sorted(x for lst in lists for x in lst)
        assert vec == base[n + i].tobytes() and len(lists) == lv + 1
    # --- the reference state: the same inserts applied to the direct index
    for i, lv in enumerate(levels):
        a.add(labels[n + i], base[n + i], level=lv)
    want = graph_by_label(a.export_graph(), base)
    got = store.graph_by_label()
    assert got.keys() == want.keys()
    for lab in want:
        assert got[lab] == want[lab], lab
    # header: size, top level, entry slot (in page form)
    assert struct.unpack_from("<Q", header, 80)[0] == n + extra and struct.unpack_from("<Q", header, 104)[0] == 6
    assert capi.header_entry_slot(header) == store.order[n + 11]
    # --- a fresh backend mirrors the updated pages and finds what the direct index finds
    c = capi.GpuIndex("l2sq", d, M=M, ef_construction=40, ef=32, seed=4, retriever=store.retriever)
    c.view_mem_lazy(header)
    assert len(c) == n + extra and c.graph_info().size <= n + extra
    queries = rng.standard_normal((50, d), dtype=np.float32)
    la, da, _ = a.search_batch(queries, 10)
    lc, dc, _ = c.search_batch(queries, 10)
    lb, db, _ = b.search_batch(queries, 10)
    assert np.array_equal(la, lc) and np.array_equal(da, dc) and np.array_equal(la, lb) and np.array_equal(da, db)
    # the same slot twice is refused; an index without pages writes sequential ids into the tape
    with pytest.raises(capi.LanternGpuError, match="already in the index"):
        b.add_external(1, base[0], store.retriever(store.order[0]), 0, store.order[0])
    e = capi.GpuIndex("l2sq", d, M=M, ef_construction=40, ef=32, seed=4)
    e.set_add_batch(1, 1)
    e.add_many(labels[:50], base[:50])
    tape = C.create_string_buffer(store.tape_bytes(1))
    e.add_external(777, base[60], C.addressof(tape), 1, 0)
    g = e.export_graph()
    cnt = struct.unpack_from("<I", tape.raw, 10)[0]
    ids = [struct.unpack_from("<I", tape.raw, 14 + j * 6)[0] for j in range(cnt)]
    assert ids == [int(x) for x in g["nbr0"][50] if x != 0xFFFFFFFF] and struct.unpack_from("<QH", tape.raw, 0) == (777, 1)


def test_add_external_from_an_empty_page_index(capi):
    """CREATE INDEX on an empty table, then INSERTs: the header declares zero nodes (build.c:675-684)."""
    rng = np.random.default_rng(25)
    d, M = 16, 4
    empty = capi.GpuIndex("l2sq", d, M=M, ef_construction=16, ef=16, seed=1)
    store = PageStore(capi, empty.save_buffer(), d * 4, M)
    b = capi.GpuIndex("l2sq", d, M=M, ef_construction=16, ef=16, seed=1, retriever=store.retriever, retriever_mut=store.retriever_mut)
    b.view_mem_lazy(store.header)
    rows = rng.standard_normal((30, d), dtype=np.float32)
    header = store.header
    direct = capi.GpuIndex("l2sq", d, M=M, ef_construction=16, ef=16, seed=1)
    direct.set_add_batch(1, 1)
    for i, row in enumerate(rows):
        lv = 2 if i == 0 else (1 if i % 7 == 0 else 0)
        addr, slot = store.new_tuple(100 + i, lv)
        b.add_external(100 + i, row, addr, lv, slot)
        header = b.update_header(header)
        direct.add(100 + i, row, level=lv)
    assert store.graph_by_label() == graph_by_label(direct.export_graph(), rows)
    assert capi.header_entry_slot(header) == store.order[0] and struct.unpack_from("<Q", header, 80)[0] == 30


# ------------------------------------------------------------------------------------------------------------------
# the 136-byte header (external_index.h:29-66) and untrusted files
# ------------------------------------------------------------------------------------------------------------------
def test_header_carries_upstream_codes_and_both_codings_load(capi):
    rng = np.random.default_rng(26)
    for metric, quant, mcode, scode, d in (("l2sq", "f32", b"e", 11, 12), ("cos", "f16", b"c", 12, 12), ("cos", "i8", b"c", 23, 12),
                                           ("hamming", "f32", b"b", 1, 2)):
        rows = rng.integers(0, 2**32, size=(60, d), dtype=np.uint32) if metric == "hamming" else rng.uniform(-1, 1, (60, d)).astype(np.float32)
        ix = capi.GpuIndex(metric, d, M=4, ef_construction=16, seed=2, quantization=quant)
        ix.add_many(np.arange(60) + 1, rows)
        blob = bytearray(ix.save_buffer())
        # index_dense_head_t: magic, version 2.x, metric_kind_t ASCII code, scalar_kind_t codes, u64 keys
        assert blob[:7] == b"usearch" and struct.unpack_from("<H", blob, 7)[0] == 2
        assert blob[13:14] == mcode and blob[14] == scode and blob[15] == 14 and blob[16] == 2
        assert struct.unpack_from("<QQQ", blob, 17) == (60, 0, d * 32 if metric == "hamming" else d)
        assert struct.unpack_from("<QQQ", blob, 80) == (60, 4, 8)
        again = capi.GpuIndex(metric, d, M=4, ef_construction=16, seed=2, quantization=quant)
        again.load_buffer(bytes(blob))
        assert again.checksum() == ix.checksum()
        # a file written by round 1 of this library (C-API numerals in the kind bytes) still loads
        old = bytearray(blob)
        old[13] = {"l2sq": 3, "cos": 1, "hamming": 8}[metric]
        old[14] = {"f32": 5 if metric == "hamming" else 1, "f16": 3, "i8": 4}[quant]
        old[15], old[16] = 8, 6
        legacy = capi.GpuIndex(metric, d, M=4, ef_construction=16, seed=2, quantization=quant)
        legacy.load_buffer(bytes(old))
        assert legacy.checksum() == ix.checksum()
        other = capi.GpuIndex("cos" if metric == "l2sq" else "l2sq", d * 32 if metric == "hamming" else d, M=4, ef_construction=16)
        with pytest.raises(capi.LanternGpuError, match="does not match the index options"):
            other.load_buffer(bytes(blob))


def test_corrupt_files_are_refused_not_trusted(capi):
    rng = np.random.default_rng(27)
    d, M = 8, 4
    ix = capi.GpuIndex("l2sq", d, M=M, ef_construction=16, seed=3)
    ix.add_many(np.arange(200) + 1, rng.standard_normal((200, d), dtype=np.float32))
    blob = ix.save_buffer()
    g = ix.export_graph()

    def load(b):
        capi.GpuIndex("l2sq", d, M=M, ef_construction=16, seed=3).load_buffer(bytes(b))

    load(blob)
    bad = bytearray(blob)
    struct.pack_into("<Q", bad, 80, 1 << 40)  # a node count no file of this length can hold
    with pytest.raises(capi.LanternGpuError, match="more nodes than it can hold"):
        load(bad)
    bad = bytearray(blob)
    struct.pack_into("<Q", bad, 112, 5000)  # entry slot beyond the nodes
    with pytest.raises(capi.LanternGpuError, match="entry slot or top level"):
        load(bad)
    bad = bytearray(blob)
    struct.pack_into("<Q", bad, 104, int(g["max_level"]) + 1)  # the walk would start above the entry node's lists
    with pytest.raises(capi.LanternGpuError, match="entry node's level"):
        load(bad)
    bad = bytearray(blob)
    struct.pack_into("<Q", bad, 96, 2 * M + 2)
    with pytest.raises(capi.LanternGpuError, match="level-0 connectivity"):
        load(bad)
    with pytest.raises(capi.LanternGpuError, match="truncated|corrupt"):
        load(blob[:len(blob) - 40])
    # an upper-level list that names a level-0 node: a walk would read a list that does not exist
    top = next(i for i in range(200) if g["levels"][i] >= 1 and g["upper_nbr"][g["upper_off"][i]][0] != 0xFFFFFFFF)
    flat = next(i for i in range(200) if g["levels"][i] == 0)
    off = 136
    for i in range(top):
        off += 10 + (4 + 2 * M * 6) + int(g["levels"][i]) * (4 + M * 6) + d * 4
    bad = bytearray(blob)
    struct.pack_into("<I", bad, off + 10 + (4 + 2 * M * 6) + 4, flat)
    with pytest.raises(capi.LanternGpuError, match="does not reach that level"):
        load(bad)


# ------------------------------------------------------------------------------------------------------------------
# mirror lifecycle: (relfilenode, LSN)-keyed cache of HBM mirrors
# ------------------------------------------------------------------------------------------------------------------
def test_mirror_hit_inserts_through_the_second_holders_callbacks(capi):
    """The reference allocates a RetrieverCtx per scan / per insert and frees it at the end (scan.c:34,132, insert.c:130,247).
    A cache hit must therefore call the CURRENT holder's callbacks, never the ones the mirror was built with."""
    import gc

    rng = np.random.default_rng(77)
    n, d, M = 600, 16, 5
    base = rng.standard_normal((n + 3, d), dtype=np.float32)
    a = capi.GpuIndex("l2sq", d, M=M, ef_construction=32, ef=32, seed=8)
    a.set_add_batch(1, 1)
    a.add_many(np.arange(n, dtype=np.uint64) + 1, base[:n])
    store = PageStore(capi, a.save_buffer(), d * 4, M)
    REL = 515151
    calls = {"first": 0, "second": 0}

    def counted(who, fn):
        def f(slot):
            calls[who] += 1
            return fn(slot)
        return f

    kw = dict(metric="l2sq", dims=d, M=M, ef_construction=32, ef=32)
    m1 = capi.Mirror(REL, 1, header=store.header, retriever=counted("first", store.retriever), retriever_mut=counted("first", store.retriever_mut), **kw)
    built = calls["first"]
    assert built > 0
    # the first scan ends: its ctx (here: the ctypes thunks) is freed.  The idle mirror stays resident.
    m1.release()
    del m1
    gc.collect()
    m2 = capi.Mirror(REL, 1, header=store.header, retriever=counted("second", store.retriever), retriever_mut=counted("second", store.retriever_mut), **kw)
    assert calls["second"] == 0 and calls["first"] == built  # a hit: nothing was walked
    addr, slot = store.new_tuple(7001, 0)
    m2.index.add_external(7001, base[n], addr, 0, slot)
    assert calls["second"] > 0 and calls["first"] == built   # the neighbours' tapes were written through the SECOND holder's retriever_mut
    assert 7001 in m2.index.search(base[n], 3)[0].tolist()
    m2.release()
    capi.Mirror.invalidate(REL)


def test_mirror_holders_on_different_threads_keep_their_own_callbacks(capi):
    """A threaded host (the scan service, the test harnesses) runs one holder per thread on a shared mirror: holder B coming and
    going between A's acquire and A's insert neither hands A's writes to B's RetrieverCtx nor leaves A without callbacks."""
    import threading

    rng = np.random.default_rng(78)
    n, d, M = 500, 16, 5
    base = rng.standard_normal((n + 2, d), dtype=np.float32)
    a = capi.GpuIndex("l2sq", d, M=M, ef_construction=32, ef=32, seed=8)
    a.set_add_batch(1, 1)
    a.add_many(np.arange(n, dtype=np.uint64) + 1, base[:n])
    store = PageStore(capi, a.save_buffer(), d * 4, M)
    REL = 616161
    calls = {"A": 0, "B": 0}

    def counted(who, fn):
        def f(slot):
            calls[who] += 1
            return fn(slot)
        return f

    kw = dict(metric="l2sq", dims=d, M=M, ef_construction=32, ef=32)
    mA = capi.Mirror(REL, 1, header=store.header, retriever=counted("A", store.retriever), retriever_mut=counted("A", store.retriever_mut), **kw)
    built = calls["A"]
    errs = []

    def holder_b():
        try:
            mB = capi.Mirror(REL, 1, header=store.header, retriever=counted("B", store.retriever), retriever_mut=counted("B", store.retriever_mut), **kw)
            assert mB.index.h == mA.index.h  # the same device index
            assert len(mB.index.search(base[3], 3)[0]) == 3
            mB.release()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    t = threading.Thread(target=holder_b)
    t.start()
    t.join()
    assert not errs, errs
    addr, slot = store.new_tuple(9001, 0)
    mA.index.add_external(9001, base[n], addr, 0, slot)  # no rebind needed: B's release took B's binding, not A's
    assert calls["A"] > built and calls["B"] == 0
    assert 9001 in mA.index.search(base[n], 3)[0].tolist()
    mA.release()
    capi.Mirror.invalidate(REL)


def test_mirror_cache_hits_rebuilds_invalidation_and_eviction(capi):
    rng = np.random.default_rng(31)
    n, d, M = 900, 16, 5
    base = rng.standard_normal((n + 5, d), dtype=np.float32)
    a = capi.GpuIndex("l2sq", d, M=M, ef_construction=32, ef=32, seed=8)
    a.set_add_batch(1, 1)
    a.add_many(np.arange(n, dtype=np.uint64) + 1, base[:n])
    store = PageStore(capi, a.save_buffer(), d * 4, M)
    kw = dict(metric="l2sq", dims=d, retriever=store.retriever, M=M, ef_construction=32, ef=32, retriever_mut=store.retriever_mut)
    before = capi.Mirror.stats()
    REL = 424242
    # ldb_ambeginscan #1: a miss -> the graph is walked once through the retriever
    m1 = capi.Mirror(REL, 1, header=store.header, **kw)
    walked = len(store.retrieved)
    assert walked == m1.index.graph_info().size > 0
    # ldb_ambeginscan #2 .. #4 at the same version: hits, not a single retriever call, the SAME device index
    m2 = capi.Mirror(REL, 1, header=store.header, **kw)
    m3 = capi.Mirror(REL, 1, header=store.header, **kw)
    assert len(store.retrieved) == walked and m2.index.h == m1.index.h == m3.index.h
    q1, q2 = rng.standard_normal((2, d), dtype=np.float32)
    s1, s2 = capi.Scan(m1.index, init_k=3), capi.Scan(m2.index, init_k=3)
    s1.rescan(q1)
    s2.rescan(q2)
    rows1, rows2 = [], []
    for _ in range(40):
        rows1.append(s1.gettuple())
        rows2.append(s2.gettuple())
    direct1, direct2 = capi.Scan(a, init_k=3), capi.Scan(a, init_k=3)
    direct1.rescan(q1)
    direct2.rescan(q2)
    assert rows1 == direct1.fetch(40) and rows2 == direct2.fetch(40)
    s1.end(); s2.end()
    m3.release()
    # ldb_aminsert by the holder: the change is applied to the mirror and the pages, the stamp advances, no rebuild
    # (m3's release unbound the retriever callbacks -- they were m3's, the latest acquirer's: an insert without binding
    # one's own fails with a message instead of calling through a freed RetrieverCtx)
    addr, slot = store.new_tuple(5001, 0)
    with pytest.raises(capi.LanternGpuError, match="retriever_mut"):
        m1.index.add_external(5001, base[n], addr, 0, slot)
    m1.rebind()
    m1.index.add_external(5001, base[n], addr, 0, slot)
    header2 = m1.index.update_header(store.header)
    m1.advance(2)
    m4 = capi.Mirror(REL, 2, header=header2, **kw)
    assert len(store.retrieved) == walked and m4.index.h == m1.index.h and m4.version == 2
    assert 5001 in m4.index.search(base[n], 3)[0].tolist()
    # somebody else changed the index (version 3): a rebuild; holders of the stale mirror keep a working handle until they let go
    m5 = capi.Mirror(REL, 3, header=header2, **kw)
    assert len(store.retrieved) > walked and m5.index.h != m1.index.h
    assert 5001 in m1.index.search(base[n], 3)[0].tolist()
    for m in (m1, m2, m4):
        m.release()
    st = capi.Mirror.stats()
    assert st["hits"] - before["hits"] == 3 and st["misses"] - before["misses"] == 1 and st["rebuilds"] - before["rebuilds"] == 1
    # a tiny index is declined (policy: the caller stays on the path it has), without an error
    assert capi.Mirror(REL + 1, 1, header=store.header, min_vectors=10_000, **kw).declined
    # DROP INDEX: the mirror goes once nobody holds it; the next acquire walks the pages again
    m5.release()
    capi.Mirror.invalidate(REL)
    calls = len(store.retrieved)
    m6 = capi.Mirror(REL, 3, header=header2, **kw)
    assert len(store.retrieved) > calls
    m6.release()
    # capacity: idle mirrors beyond it are evicted, least recently used first
    capi.Mirror.set_capacity(1)
    m7 = capi.Mirror(REL + 2, 1, header=header2, **kw)
    m7.release()
    assert capi.Mirror.stats()["resident"] == 1
    calls = len(store.retrieved)
    capi.Mirror(REL, 3, header=header2, **kw).release()  # was evicted: walked again
    assert len(store.retrieved) > calls
    capi.Mirror.set_capacity(8)
    capi.Mirror.invalidate(REL)
    capi.Mirror.invalidate(REL + 2)
