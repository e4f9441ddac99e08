"""The scan-side service (lantern_amd/csrc/scan_server.cpp): many clients, one query each at a time, coalesced into
batched search launches.  CPU tests drive the REAL server, sockets, dispatcher and client code over an injected batch
function (the server's pluggable back end), so batching, routing, grouping by (k, ef), error frames and shutdown are
covered without a device; the -m gpu test puts a device index behind it and compares with direct batch search."""
import threading
import time

import numpy as np
import pytest

from tests.conftest import needs_experimental


@pytest.fixture(scope="module")
def capi():
    from lantern_amd import build, capi

    build.build()
    capi.lib()
    return capi


def fake_backend(calls):
    """labels[i][j] = 1000 * (first float of query i) + j ; dists[i][j] = j + ef / 1000 ; records every launch."""
    def fn(queries, k, ef):
        nq = queries.shape[0]
        first = queries.view(np.float32)[:, 0]
        calls.append((nq, k, ef))
        lab = (first[:, None] * 1000 + np.arange(k)[None, :]).astype(np.uint64)
        dst = (np.arange(k, dtype=np.float32)[None, :] + np.float32(ef) / 1000).repeat(nq, 0)
        return lab, dst, np.full(nq, k, dtype=np.uint32)
    return fn


def test_many_clients_are_batched_and_answers_routed(capi):
    calls = []
    srv = capi.ScanServer(batch_fn=fake_backend(calls), vec_bytes=16, max_batch=64, max_wait_us=20000)
    nthreads, per = 24, 12
    got, errs = {}, []
    start = threading.Barrier(nthreads)

    def backend_session(t):
        try:
            c = capi.ScanClient(srv.host, srv.port)
            start.wait()
            for i in range(per):
                ident = t * 100 + i
                q = np.array([ident, 0, 0, 0], dtype=np.float32)
                k = 5 if t % 2 == 0 else 8  # two (k, ef) classes in flight at once
                lab, dst = c.search(q, k, ef=40 if t % 2 == 0 else 0)
                got[ident] = (lab.copy(), dst.copy(), k)
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=backend_session, args=(t,)) for t in range(nthreads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    st = srv.stats()
    legs = srv.timing()
    srv.stop()
    assert not errs, errs
    assert len(got) == nthreads * per
    # the per-leg clock (lantern_scan_server_timing): every answered request was timed once, and every leg took some time
    assert legs["requests"] == nthreads * per
    assert legs["wait_for_batch_us"] >= 0 and legs["batch_closed_to_answer_us"] > 0 and legs["answer_to_socket_us"] > 0
    assert legs["wait_for_batch_us"] < 25_000 * 2  # never longer than the window allows (20 ms here)
    for ident, (lab, dst, k) in got.items():  # every client got ITS answer, in order, for ITS k and ef
        assert lab.tolist() == [ident * 1000 + j for j in range(k)]
        assert np.allclose(dst, np.arange(k) + (0.04 if k == 5 else 0.0))
    assert st["requests"] == nthreads * per
    assert st["batches"] < st["requests"] and st["largest_batch"] > 1, st  # coalescing happened
    assert st["launches"] == len(calls) and sum(c[0] for c in calls) == st["requests"]
    assert {(c[1], c[2]) for c in calls} == {(5, 40), (8, 0)}  # one launch per distinct (k, ef), never mixed


@pytest.mark.parametrize("lanes", [1, 2])
def test_two_dispatchers_keep_two_batches_in_flight(capi, monkeypatch, lanes):
    """One dispatcher forms the next batch while the other's batch is being searched (the default back end runs them on two
    lanes of the index).  A caller-supplied back end is entered by one thread unless LANTERN_SCAN_LANES=2 allows two."""
    import time

    if lanes == 2:
        monkeypatch.setenv("LANTERN_SCAN_LANES", "2")
    else:
        monkeypatch.delenv("LANTERN_SCAN_LANES", raising=False)
    lock, state, calls = threading.Lock(), {"in": 0, "peak": 0}, []
    inner = fake_backend(calls)

    def slow(queries, k, ef):
        with lock:
            state["in"] += 1
            state["peak"] = max(state["peak"], state["in"])
        time.sleep(0.03)  # (releases the GIL: the other dispatcher can enter meanwhile)
        out = inner(queries, k, ef)
        with lock:
            state["in"] -= 1
        return out

    srv = capi.ScanServer(batch_fn=slow, vec_bytes=16, max_batch=4, max_wait_us=500)
    nthreads, per = 12, 5
    got, errs = {}, []

    def session(t):
        try:
            c = capi.ScanClient(srv.host, srv.port)
            for i in range(per):
                ident = t * 100 + i
                lab, _ = c.search(np.array([ident, 0, 0, 0], dtype=np.float32), 3)
                got[ident] = lab.copy()
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=session, args=(t,)) for t in range(nthreads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    st = srv.stats()
    srv.stop()
    assert not errs, errs
    assert len(got) == nthreads * per and all(lab.tolist() == [i * 1000, i * 1000 + 1, i * 1000 + 2] for i, lab in got.items())
    assert st["requests"] == nthreads * per and sum(c[0] for c in calls) == nthreads * per
    assert state["peak"] == lanes, state


def ranked_backend(table):
    """A back end with a real ranking: rows of `table` (n x 4 f32) by squared distance to the query, labels = row + 1,
    except row 3 whose label is 0 (a deleted row, skipped by the scan: scan.c:296-300)."""
    def fn(queries, k, ef):
        q = queries.view(np.float32).reshape(queries.shape[0], -1)
        d = ((q[:, None, :] - table[None, :, :]) ** 2).sum(-1)
        order = np.argsort(d, axis=1, kind="stable")[:, :k]
        lab = (order + 1).astype(np.uint64)
        lab[order == 3] = 0
        cnt = np.full(q.shape[0], min(k, table.shape[0]), dtype=np.uint32)
        out_l = np.zeros((q.shape[0], k), dtype=np.uint64)
        out_d = np.full((q.shape[0], k), np.inf, dtype=np.float32)
        out_l[:, :order.shape[1]] = lab
        out_d[:, :order.shape[1]] = np.take_along_axis(d, order, axis=1)
        return out_l, out_d, cnt
    return fn


def test_concurrent_scans_paginate_independently_through_the_service(capi):
    """The continuation state is the CONNECTION's (the reference: one usearch handle per scan, scan.c:99): backends that
    page through their results at the same time each see exactly what they see alone -- nothing twice, nothing lost."""
    rng = np.random.default_rng(3)
    table = rng.standard_normal((300, 4)).astype(np.float32)
    srv = capi.ScanServer(batch_fn=ranked_backend(table), vec_bytes=16, max_batch=32, max_wait_us=2000)
    queries = rng.standard_normal((12, 4)).astype(np.float32)

    def page_through(c, q):
        rows = []
        lab, _ = c.search(q, 4)
        rows += lab.tolist()
        for k in (8, 16, 32):
            lab, _ = c.search_next(q, k)
            rows += lab.tolist()
        return rows

    solo = []
    c = capi.ScanClient(srv.host, srv.port)
    for q in queries:
        solo.append(page_through(c, q))
    c.close()
    for q, rows in zip(queries, solo):  # the pages are consecutive stretches of the full ranking, label 0 included once
        full = ranked_backend(table)(q.view(np.uint8).reshape(1, -1), 60, 0)[0][0].tolist()
        assert rows == full and len([r for r in rows if r]) == len(set(r for r in rows if r))
    got, errs = {}, []
    start = threading.Barrier(len(queries))

    def session(i):
        try:
            cc = capi.ScanClient(srv.host, srv.port)
            start.wait()
            got[i] = page_through(cc, queries[i])
            cc.close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=session, args=(i,)) for i in range(len(queries))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert [got[i] for i in range(len(queries))] == solo
    # the amgettuple shim over a service connection: init_k = 3, then 6, 12, ... through the continuation; label 0 skipped
    cc = capi.ScanClient(srv.host, srv.port)
    sc = capi.Scan(client=cc, metric="l2sq", dims=4, init_k=3)
    sc.rescan(queries[0])
    rows = sc.fetch(40)
    want = [r for r in ranked_backend(table)(queries[0].view(np.uint8).reshape(1, -1), 60, 0)[0][0].tolist() if r][:40]
    assert rows == want
    sc.rescan(queries[1])  # ldb_amrescan: a fresh scan on the same connection
    assert sc.fetch(5) == [r for r in solo[1] if r][:5]
    sc.end()
    cc.close()
    srv.stop()


def test_a_lone_query_is_not_held_longer_than_the_window(capi):
    srv = capi.ScanServer(batch_fn=fake_backend([]), vec_bytes=8, max_batch=64, max_wait_us=3000)
    c = capi.ScanClient(srv.host, srv.port)
    c.search(np.array([1, 0], dtype=np.float32), 3)
    t0 = time.perf_counter()
    for _ in range(20):
        c.search(np.array([2, 0], dtype=np.float32), 3)
    per = (time.perf_counter() - t0) / 20
    c.close()
    srv.stop()
    assert per < 0.05, per  # the 3 ms window plus the round trip, not a stall


def test_full_batch_leaves_before_the_window_closes(capi):
    calls = []
    srv = capi.ScanServer(batch_fn=fake_backend(calls), vec_bytes=8, max_batch=4, max_wait_us=2_000_000)  # a 2 s window
    out, start = [], threading.Barrier(4)

    def one(t):
        c = capi.ScanClient(srv.host, srv.port)
        start.wait()
        out.append(c.search(np.array([t, 0], dtype=np.float32), 2)[0][0])
        c.close()

    t0 = time.perf_counter()
    ts = [threading.Thread(target=one, args=(t,)) for t in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    took = time.perf_counter() - t0
    srv.stop()
    assert sorted(out) == [0, 1000, 2000, 3000] and took < 1.5, (out, took)  # max_batch reached: no need to wait 2 s


def test_errors_come_back_as_messages_and_the_connection_survives(capi):
    def failing(queries, k, ef):
        if k == 7:
            raise RuntimeError("lantern_gpu: ef/k exceed the LDS budget")
        return fake_backend([])(queries, k, ef)

    srv = capi.ScanServer(batch_fn=failing, vec_bytes=8, max_wait_us=100)
    c = capi.ScanClient(srv.host, srv.port)
    with pytest.raises(capi.LanternGpuError, match="exceed the LDS budget"):
        c.search(np.array([1, 0], dtype=np.float32), 7)
    with pytest.raises(capi.LanternGpuError, match="query of 12 bytes, the index takes 8"):
        c.search(np.array([1, 0, 0], dtype=np.float32), 3)
    lab, _ = c.search(np.array([5, 0], dtype=np.float32), 3)  # same connection, still in step
    assert lab.tolist() == [5000, 5001, 5002]
    with pytest.raises(capi.LanternGpuError, match="bad scan client arguments"):
        c.search(np.array([5, 0], dtype=np.float32), 0)
    c.close()
    srv.stop()


def test_stop_with_connected_clients_does_not_hang(capi):
    srv = capi.ScanServer(batch_fn=fake_backend([]), vec_bytes=8, max_wait_us=100)
    clients = [capi.ScanClient(srv.host, srv.port) for _ in range(5)]
    clients[0].search(np.array([1, 0], dtype=np.float32), 2)
    t0 = time.perf_counter()
    srv.stop()
    assert time.perf_counter() - t0 < 5
    with pytest.raises(capi.LanternGpuError, match="went away|not connected"):
        clients[1].search(np.array([1, 0], dtype=np.float32), 2)
    [c.close() for c in clients]
    with pytest.raises(capi.LanternGpuError, match="cannot connect"):
        capi.ScanClient(srv.host, srv.port)


def test_an_idle_connection_stays_open(capi):
    # a backend connects once per session and may sit idle between queries (the listener's accept timeout must not leak
    # into the connection)
    srv = capi.ScanServer(batch_fn=fake_backend([]), vec_bytes=8, max_wait_us=100)
    c = capi.ScanClient(srv.host, srv.port)
    assert c.search(np.array([3, 0], dtype=np.float32), 2)[0].tolist() == [3000, 3001]
    time.sleep(0.6)
    assert c.search(np.array([4, 0], dtype=np.float32), 2)[0].tolist() == [4000, 4001]
    c.close()
    srv.stop()


def _raw_request(ident, k, nfloats=2, magic=0x5152534C):
    import struct

    vec = np.zeros(nfloats, dtype=np.float32)
    vec[0] = ident
    return struct.pack("<IIII", magic, k, 0, vec.nbytes) + vec.tobytes()


def _raw_reply(sock, k):
    import struct

    buf = b""
    while len(buf) < 12:
        chunk = sock.recv(12 - len(buf))
        assert chunk, "the server closed the connection"
        buf += chunk
    magic, status, count = struct.unpack("<III", buf)
    assert magic == 0x5052534C
    body = b""
    want = count if status else count * 12
    while len(body) < want:
        chunk = sock.recv(want - len(body))
        assert chunk
        body += chunk
    if status:
        return status, body.decode()
    return 0, np.frombuffer(body[: count * 8], dtype=np.uint64).tolist()


def test_requests_in_pieces_back_to_back_and_abandoned(capi):
    """The server's connections are state machines behind epoll, not blocking readers: a request may arrive a few bytes at a time,
    two may arrive in one segment (the second waits in the socket until the first is answered), and a client may vanish in the
    middle of a request -- none of which disturbs the other connections."""
    import socket

    srv = capi.ScanServer(batch_fn=fake_backend([]), vec_bytes=8, max_batch=16, max_wait_us=500)
    slow = socket.create_connection((srv.host, srv.port))
    slow.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
    req = _raw_request(7, 3)
    for i in range(0, len(req), 5):  # 24 bytes in five pieces
        slow.sendall(req[i:i + 5])
        time.sleep(0.02)
    assert _raw_reply(slow, 3) == (0, [7000, 7001, 7002])
    # two requests in one write: answered one after the other, in order
    slow.sendall(_raw_request(8, 2) + _raw_request(9, 4))
    assert _raw_reply(slow, 2) == (0, [8000, 8001])
    assert _raw_reply(slow, 4) == (0, [9000, 9001, 9002, 9003])
    # a client that dies with half a request on the wire
    gone = socket.create_connection((srv.host, srv.port))
    gone.sendall(_raw_request(1, 3)[:10])
    gone.close()
    # a wrong vector size (error frame, connection survives), then a bad magic (error frame, connection closed)
    slow.sendall(_raw_request(5, 3, nfloats=3))
    status, msg = _raw_reply(slow, 3)
    assert status == 1 and "query of 12 bytes, the index takes 8" in msg
    slow.sendall(_raw_request(6, 1))
    assert _raw_reply(slow, 1) == (0, [6000])
    bad = socket.create_connection((srv.host, srv.port))
    bad.sendall(_raw_request(1, 3, magic=0x12345678))
    status, msg = _raw_reply(bad, 3)
    assert status == 1 and "bad request magic" in msg
    assert bad.recv(1) == b""  # closed by the server
    # everybody else is served as if nothing had happened
    c = capi.ScanClient(srv.host, srv.port)
    assert c.search(np.array([4, 0], dtype=np.float32), 2)[0].tolist() == [4000, 4001]
    c.close()
    slow.close()
    bad.close()
    srv.stop()


def test_a_backend_that_stops_reading_its_answers_is_dropped_and_stalls_nobody(capi):
    """A backend that keeps sending requests but never reads an answer fills its socket buffers; the server's answer writes are
    non-blocking (write_answer), so neither a dispatcher nor an I/O thread ever parks in send(): the others are served at their
    usual latency the whole time, and the deaf connection is closed once it cannot take an answer."""
    import socket

    srv = capi.ScanServer(batch_fn=fake_backend([]), vec_bytes=8, max_batch=16, max_wait_us=300)
    deaf = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    deaf.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 4096)  # before connect: a small receive window fills quickly
    deaf.connect((srv.host, srv.port))
    deaf.setblocking(False)
    good = capi.ScanClient(srv.host, srv.port)
    req = _raw_request(3, 100) * 64  # 100 results per answer: 1 212 bytes each
    worst, dropped, sent = 0.0, False, 0
    t_end = time.time() + 20.0
    while time.time() < t_end and not dropped:
        try:
            sent += deaf.send(req)
        except BlockingIOError:
            pass
        except (ConnectionResetError, BrokenPipeError):
            dropped = True
        t0 = time.time()
        assert good.search(np.array([4, 0], dtype=np.float32), 2)[0].tolist() == [4000, 4001]
        worst = max(worst, time.time() - t0)
    assert dropped, f"the deaf connection was still open after {sent} request bytes"
    assert worst < 0.5, f"a healthy backend waited {worst:.3f} s behind a deaf one"
    good.close()
    deaf.close()
    srv.stop()


def test_more_connections_than_io_threads_and_idle_ones_do_not_stall_the_window(capi, monkeypatch):
    """Three I/O threads, forty connections of which thirty never say a word: the active ones are answered within the window
    (nobody waits for the idle ones beyond it), and every connection is served by the thread it was dealt to."""
    monkeypatch.setenv("LANTERN_SCAN_IO_THREADS", "3")
    srv = capi.ScanServer(batch_fn=fake_backend([]), vec_bytes=8, max_batch=64, max_wait_us=2000)
    idle = [capi.ScanClient(srv.host, srv.port) for _ in range(30)]
    active = [capi.ScanClient(srv.host, srv.port) for _ in range(10)]
    out, errs = {}, []

    def session(i):
        try:
            for r in range(15):
                out[(i, r)] = active[i].search(np.array([i * 20 + r, 0], dtype=np.float32), 2)[0].tolist()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    t0 = time.perf_counter()
    ts = [threading.Thread(target=session, args=(i,)) for i in range(10)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    took = time.perf_counter() - t0
    assert not errs, errs
    assert all(out[(i, r)] == [(i * 20 + r) * 1000, (i * 20 + r) * 1000 + 1] for i in range(10) for r in range(15))
    assert took < 15 * 0.2, took  # 15 rounds, each at most the 2 ms window plus the round trip -- not a stall
    assert idle[0].search(np.array([3, 0], dtype=np.float32), 1)[0].tolist() == [3000]  # an idle one wakes up and is served
    [c.close() for c in idle + active]
    srv.stop()


def test_server_on_an_index_needs_a_device(capi):
    if capi.device_count() > 0:
        pytest.skip("a device is present")
    err = __import__("ctypes").c_char_p()
    assert capi.lib().lantern_scan_server_start(None, b"127.0.0.1", 0, 16, 100, __import__("ctypes").byref(err)) is None
    assert b"null index handle" in err.value


@pytest.mark.gpu
@pytest.mark.parametrize("notify", ["1", "0"])  # answers one by one as their walks end (the default) / all of a batch's when its launch ends
def test_concurrent_backends_get_exactly_the_direct_search_results(capi, monkeypatch, notify):
    monkeypatch.setenv("LANTERN_SCAN_NOTIFY", notify)
    n, d, k = 20000, 64, 10
    rng = np.random.default_rng(5)
    base = rng.standard_normal((n, d), dtype=np.float32)
    ix = capi.GpuIndex("l2sq", d, M=16, ef_construction=64, ef=48, seed=3)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    ix.flush()
    queries = rng.standard_normal((16 * 20, d), dtype=np.float32)
    want_lab, want_dst, _ = ix.search_batch(queries, k)
    srv = capi.ScanServer(index=ix, max_batch=64, max_wait_us=300)
    got, errs = {}, []

    def session(t):
        try:
            c = capi.ScanClient(srv.host, srv.port)
            for i in range(20):
                qi = t * 20 + i
                got[qi] = c.search(queries[qi], k)
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=session, args=(t,)) for t in range(16)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    st = srv.stats()
    srv.stop()
    assert not errs, errs
    for qi in range(queries.shape[0]):
        assert np.array_equal(got[qi][0], want_lab[qi]) and np.array_equal(got[qi][1], want_dst[qi])
    assert st["requests"] == 320 and st["batches"] < 320 and st["largest_batch"] > 1, st


@pytest.mark.gpu
@pytest.mark.parametrize("metric,n,d,nq", [("l2sq", 20000, 128, 300), ("cos", 8000, 768, 64), ("l2sq", 5000, 64, 1), ("hamming", 6000, 8, 700)])
def test_lane_notify_hands_on_every_query_once_with_its_final_rows(capi, metric, n, d, nq):
    """lantern_gpu_search_batch_lane_notify: the kernels raise one host-visible word per query as its walk ends and the caller is told
    batch by batch of finished queries.  Every query is handed on exactly once, its rows are the final ones at that moment, and the
    answers are lantern_gpu_search_batch's -- in the latency-bound shapes (<= two queries per CU) and in the classic one."""
    rng = np.random.default_rng(n + nq)
    if metric == "hamming":
        base = rng.integers(0, 2**32, size=(n, d), dtype=np.uint32)
        queries = rng.integers(0, 2**32, size=(nq, d), dtype=np.uint32)
    else:
        base = rng.standard_normal((n, d), dtype=np.float32)
        queries = rng.standard_normal((nq, d), dtype=np.float32)
    ix = capi.GpuIndex(metric, d, M=16, ef_construction=64, ef=48, seed=3)
    ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    ix.flush()
    want_lab, want_dst, want_cnt = ix.search_batch(queries, 10)
    for lane in (0, 3):
        lab, dst, cnt, calls, snap = ix.search_batch_lane_notify(lane, queries, 10)
        assert np.array_equal(lab, want_lab) and np.array_equal(dst, want_dst) and np.array_equal(cnt, want_cnt)
        handed = [j for c in calls for j in c]
        assert sorted(handed) == list(range(nq)), "every query exactly once"
        for j, (l, dd, c) in snap.items():
            assert np.array_equal(l, want_lab[j]) and np.array_equal(dd, want_dst[j]) and c == want_cnt[j], "rows were final when handed on"
    if nq >= 300:
        assert len(calls) > 1, "a batch whose walks differ in length is handed on in more than one piece"


@pytest.mark.gpu
@pytest.mark.parametrize("which", [pytest.param("one_wave_walk", marks=needs_experimental), "adc_table_walk"])
def test_lane_notify_through_the_other_search_kernels(capi, monkeypatch, which):
    """The per-query completion words are raised by every search kernel's tail: here the one-wave walk (k_search_solo,
    LANTERN_GPU_SPEC=4) and the table walk over PQ code bytes (k_search_adc, a compact pq index with LANTERN_GPU_PQ_ADC=1)."""
    rng = np.random.default_rng(77)
    n, d, nq = 6000, 128, 200
    base = rng.standard_normal((n, d), dtype=np.float32)
    queries = rng.standard_normal((nq, d), dtype=np.float32)
    if which == "one_wave_walk":
        monkeypatch.setenv("LANTERN_GPU_SPEC", "4")
        ix = capi.GpuIndex("l2sq", d, M=16, ef_construction=64, ef=48, seed=3)
        ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
        ix.flush()
    else:
        monkeypatch.setenv("LANTERN_GPU_PQ_ADC", "1")
        cb = np.zeros((64, d), dtype=np.float32)
        for s in range(16):
            cb[:, s * 8:(s + 1) * 8] = base[rng.choice(n, size=64, replace=False), s * 8:(s + 1) * 8]
        ix = capi.GpuIndex("l2sq", d, M=16, ef_construction=64, ef=48, seed=3, pq_codebook=cb, num_subvectors=16)
        ix.add_many(np.arange(n, dtype=np.uint64) + 1, base)
        ix.flush()
        ix.pq_compact()
    want_lab, want_dst, want_cnt = ix.search_batch(queries, 10)
    before = ix.counters()["search_solo_launches"]
    lab, dst, cnt, calls, snap = ix.search_batch_lane_notify(1, queries, 10)
    assert np.array_equal(lab, want_lab) and np.array_equal(dst, want_dst) and np.array_equal(cnt, want_cnt)
    assert sorted(j for c in calls for j in c) == list(range(nq))
    assert all(np.array_equal(l, want_lab[j]) for j, (l, _, _) in snap.items())
    if which == "one_wave_walk":
        assert ix.counters()["search_solo_launches"] == before + 1


def test_standalone_binary_fails_loudly_without_a_device_or_arguments(capi):
    import os
    import subprocess

    from lantern_amd import build

    exe = os.path.join(os.path.dirname(build.LIB), "lantern-scan-server")
    assert os.path.exists(exe)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert p.returncode == 2 and "--index and --dim are required" in p.stderr
    if capi.device_count() == 0:
        p = subprocess.run([exe, "--index", "/nonexistent", "--dim", "8"], capture_output=True, text=True, timeout=60)
        assert p.returncode == 1 and "no HIP device" in p.stderr
