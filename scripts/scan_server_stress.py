#!/usr/bin/env python
"""Stress of the scan service on CPU (fake back end): 120 raw-socket clients x 60 requests per cycle, 2 % of the requests abandoned half
sent, 30 % sent in two pieces, the server stopped under load every third cycle, one and two dispatchers alternating.

    python scripts/scan_server_stress.py [cycles]
"""
import os, random, socket, struct, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lantern_amd import capi

def backend(queries, k, ef):
    nq = queries.shape[0]
    first = queries.view(np.float32)[:, 0]
    lab = (first[:, None] * 1000 + np.arange(k)[None, :]).astype(np.uint64)
    dst = np.zeros((nq, k), np.float32)
    return lab, dst, np.full(nq, k, dtype=np.uint32)

def req(ident, k):
    v = np.zeros(2, np.float32); v[0] = ident
    return struct.pack("<IIII", 0x5152534C, k, 0, 8) + v.tobytes()

def recv_all(s, n):
    b = b""
    while len(b) < n:
        c = s.recv(n - len(b))
        if not c: raise EOFError
        b += c
    return b

def client(host, port, cid, rounds, errs, stop_evt):
    rnd = random.Random(cid)
    try:
        s = socket.create_connection((host, port)); s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        for r in range(rounds):
            if stop_evt.is_set(): break
            ident = cid * 1000 + r
            k = rnd.choice((1, 3, 7))
            data = req(ident % 4000, k)
            if rnd.random() < 0.02:
                s.sendall(data[: rnd.randrange(1, len(data))]); s.close()   # vanish mid-request
                s = socket.create_connection((host, port)); continue
            if rnd.random() < 0.3:
                cut = rnd.randrange(1, len(data)); s.sendall(data[:cut]); time.sleep(0.0005); s.sendall(data[cut:])
            else:
                s.sendall(data)
            magic, status, count = struct.unpack("<III", recv_all(s, 12))
            if status == 1:
                msg = recv_all(s, count)
                assert b"stopping" in msg, msg
                break
            assert magic == 0x5052534C and status == 0 and count == k, (magic, status, count, k)
            body = recv_all(s, count * 12)
            labs = np.frombuffer(body[: count * 8], np.uint64)
            assert labs[0] == (ident % 4000) * 1000, (labs[0], ident)
        s.close()
    except (EOFError, ConnectionError, OSError) as e:
        if not stop_evt.is_set(): errs.append(("io", cid, repr(e)))
    except AssertionError as e:
        errs.append(("assert", cid, repr(e)))

for cycle in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    os.environ["LANTERN_SCAN_LANES"] = "2" if cycle % 2 else "1"
    srv = capi.ScanServer(batch_fn=backend, vec_bytes=8, max_batch=64, max_wait_us=300)
    errs, stop_evt = [], threading.Event()
    ts = [threading.Thread(target=client, args=(srv.host, srv.port, c, 60, errs, stop_evt)) for c in range(120)]
    [t.start() for t in ts]
    if cycle % 3 == 2:
        time.sleep(0.3); stop_evt.set(); srv.stop()          # stop under load
        [t.join() for t in ts]
    else:
        [t.join() for t in ts]
        st = srv.stats(); srv.stop()
        print("cycle", cycle, "requests", st["requests"], "batches", st["batches"], "largest", st["largest_batch"], "errors", len(errs))
    bad = [e for e in errs if e[0] == "assert"]
    assert not bad, bad[:3]
    if cycle % 3 != 2: assert not errs, errs[:3]
print("stress ok")
