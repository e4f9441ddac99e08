"""The external indexing server (boundary B3): protocol framing on CPU, a full build round trip on GPU.
Cases follow lantern_cli/tests/external_index_server_test.rs:141-326."""
import json
import struct
import time
import urllib.request

import numpy as np
import pytest

from tests import index_client as ic


@pytest.fixture(scope="module")
def capi():
    from lantern_amd import build, capi

    build.build()
    capi.lib()
    return capi


@pytest.fixture()
def server(capi):
    srv = capi.IndexServer("127.0.0.1", 0, 0, "/tmp")
    yield srv
    srv.stop()


def status_of(srv):
    with urllib.request.urlopen(f"http://127.0.0.1:{srv.status_port}/", timeout=5) as r:
        return json.loads(r.read())


def test_hello_and_invalid_header(server):
    # external_index_server_test.rs:141-167
    s, version, server_type = ic.connect(server.host, server.port)
    assert (version, server_type) == (1, 1)
    s.sendall(bytes([0, 1, 1, 1, 1, 1]))
    assert ic.read_error(s) == "Invalid message header"
    s.close()


def test_short_message(server):
    # external_index_server_test.rs:169-195
    s, _, _ = ic.connect(server.host, server.port)
    s.sendall(bytes([0, 1]))
    assert ic.read_error(s) == "Invalid frame received"
    s.close()


def test_bad_params_and_status_endpoint(server):
    assert status_of(server)["status"] == 0  # idle
    s, _, _ = ic.connect(server.host, server.port)
    s.sendall(ic.init_frame(metric_kind=2, quantization=1, dim=3, m=12, efc=64, ef=32, capacity=4, element_bits=32))
    assert ic.read_error(s) == "Invalid metric 2"  # cli.rs:56-69
    s.close()
    s, _, _ = ic.connect(server.host, server.port)
    s.sendall(ic.init_frame(metric_kind=1, quantization=9, dim=3, m=12, efc=64, ef=32, capacity=4, element_bits=32))
    assert ic.read_error(s) == "Invalid scalar quantization"  # server.rs:94-101
    s.close()
    s, _, _ = ic.connect(server.host, server.port)
    s.sendall(struct.pack("<I", ic.END_MSG) + bytes(44))
    assert ic.read_error(s) == "send init message first"  # server.rs:209
    s.close()
    time.sleep(0.1)
    st = status_of(server)
    assert st["status"] == 2 and st["status_updated_at"] > 0  # failed
    assert server.served == 3


def test_build_without_device_is_refused_loudly(capi, server):
    if capi.device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(ic.IndexServerError, match="no HIP device"):
        ic.build_index(server.host, server.port, 3, 3, [np.zeros(3, np.float32).tobytes()], [1])


@pytest.mark.gpu
def test_full_build_round_trip(capi, server):
    # external_index_server_test.rs:197-326 (f32 cosine, 14 rows, capacity = len/2 to force a resize)
    tuples = [(i, v) for i, v in enumerate([[0, 0, 0], [0, 0, 1], [0, 0, 2], [0, 0, 3], [0, 1, 0], [0, 1, 1], [0, 1, 2], [0, 1, 3],
                                            [1, 0, 0], [1, 0, 1], [1, 0, 2], [1, 0, 3], [1, 1, 0], [1, 1, 1]])]
    rows = [np.asarray(v, np.float32).tobytes() for _, v in tuples]
    n, data = ic.build_index(server.host, server.port, 1, 3, rows, [t[0] for t in tuples], m=12, efc=64, ef=32, capacity=len(tuples) // 2)
    assert n == len(tuples) and len(data) > 136 and data[:7] == b"usearch"
    received = capi.GpuIndex("cos", 3, M=12, ef_construction=64, ef=32)
    received.load_buffer(data)
    local = capi.GpuIndex("cos", 3, M=12, ef_construction=64, ef=32)
    local.add_many([t[0] for t in tuples], np.asarray([t[1] for t in tuples], np.float32))
    assert len(received) == len(local) == len(tuples)  # what the Rust test asserts (:316)
    assert received.save_buffer() == local.save_buffer()  # and byte-for-byte the same index file
    time.sleep(0.1)
    assert status_of(server)["status"] == 3


@pytest.mark.gpu
def test_hamming_and_larger_build(capi, server):
    rng = np.random.default_rng(2)
    # hamming: element_bits = 1, dim = bits, payload ceil(dim/8) bytes (server.rs:226-230)
    words = rng.integers(0, 2**32, size=(700, 3), dtype=np.uint32)
    n, data = ic.build_index(server.host, server.port, 8, 96, [w.tobytes() for w in words], np.arange(700) + 1, m=8, efc=32, ef=16,
                             element_bits=1, quantization=5)
    assert n == 700
    ix = capi.GpuIndex("hamming", 3, M=8, ef_construction=32, ef=16)
    ix.load_buffer(data)
    labels, dists = ix.search(words[5], 1)
    assert labels[0] == 6 and dists[0] == 0
    # f32 l2sq, 3000 rows
    base = rng.standard_normal((3000, 32), dtype=np.float32)
    n, data = ic.build_index(server.host, server.port, 3, 32, [r.tobytes() for r in base], np.arange(3000) + 1, m=16, efc=64, ef=64)
    ix = capi.GpuIndex("l2sq", 32, M=16, ef_construction=64, ef=64)
    ix.load_buffer(data)
    hits = sum(int(ix.search(base[i], 1)[0][0]) == i + 1 for i in range(0, 3000, 30))
    assert n == 3000 and hits >= 95
    # an unsupported storage kind (f64) is refused with an error frame, not a hang
    with pytest.raises(ic.IndexServerError, match="f32, f16, i8 or b1 storage"):
        ic.build_index(server.host, server.port, 3, 32, [], [], element_bits=32, quantization=2)
    # quant_bits = 8: f32 rows in (element_bits = 32), i8 storage; raw i8 rows (element_bits = 8) give the same file
    small = (base[:600] * np.float32(0.3)).astype(np.float32)
    q8 = np.trunc(np.clip(small * np.float32(100.0), -100, 100)).astype(np.int8)
    n3, f3 = ic.build_index(server.host, server.port, 3, 32, [r.tobytes() for r in small], np.arange(600) + 1, m=8, efc=32, ef=16,
                            element_bits=32, quantization=4)
    n4, f4 = ic.build_index(server.host, server.port, 3, 32, [r.tobytes() for r in q8], np.arange(600) + 1, m=8, efc=32, ef=16,
                            element_bits=8, quantization=4)
    assert n3 == n4 == 600 and f3 == f4
    assert len(f3) == 136 + sum(10 + (4 + 16 * 6) + lv * (4 + 8 * 6) + 32 for lv in _levels_of(f3, 600, 8, 32))
    # quant_bits = 16: PostgreSQL still streams f32 rows (element_bits = 32); the index stores halves.
    # The Rust tests also stream raw halves (element_bits = 16): both give the same index file.
    n1, f1 = ic.build_index(server.host, server.port, 3, 32, [r.tobytes() for r in base[:600]], np.arange(600) + 1, m=8, efc=32, ef=16,
                            element_bits=32, quantization=3)
    n2, f2 = ic.build_index(server.host, server.port, 3, 32, [r.astype(np.float16).tobytes() for r in base[:600]], np.arange(600) + 1, m=8,
                            efc=32, ef=16, element_bits=16, quantization=3)
    assert n1 == n2 == 600 and f1 == f2
    assert len(f1) == 136 + sum(10 + (4 + 16 * 6) + lv * (4 + 8 * 6) + 32 * 2 for lv in _levels_of(f1, 600, 8, 64))


def _levels_of(blob, n, M, vec_bytes):
    out, off = [], 136
    for _ in range(n):
        lv = struct.unpack_from("<H", blob, off + 8)[0]
        out.append(lv)
        off += 10 + (4 + 2 * M * 6) + lv * (4 + M * 6) + vec_bytes
    assert off == len(blob)
    return out


# ------------------------------------------------------------------------------------------------------------------
# TLS: `start-indexing-server --cert C --key K` (lantern_cli/src/external_index/cli.rs:146, server.rs:437-470,548); the
# PostgreSQL side connects with OpenSSL, TLS >= 1.2, and does not verify the certificate (external_index_socket_ssl.c:6-62).
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tls_files(tmp_path_factory):
    import shutil
    import subprocess

    if not shutil.which("openssl"):
        pytest.skip("no openssl binary to make a test certificate with")
    d = tmp_path_factory.mktemp("tls")
    cert, key = str(d / "cert.pem"), str(d / "key.pem")
    subprocess.check_call(["openssl", "req", "-x509", "-newkey", "rsa:2048", "-nodes", "-keyout", key, "-out", cert, "-days", "2", "-subj", "/CN=lantern-index-server"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return cert, key


def test_tls_server_speaks_the_protocol_after_the_handshake(capi, tls_files):
    cert, key = tls_files
    srv = capi.IndexServer("127.0.0.1", 0, 0, "/tmp", cert=cert, key=key)
    try:
        s, version, server_type = ic.connect(srv.host, srv.port, tls=True)
        assert (version, server_type) == (1, 1) and s.version() in ("TLSv1.2", "TLSv1.3")
        s.sendall(bytes([0, 1, 1, 1, 1, 1]))
        assert ic.read_error(s) == "Invalid message header"  # the error frames travel inside the session too
        s.close()
        # a client that does not speak TLS gets no protocol bytes and does not take the server down
        import socket

        plain = socket.create_connection((srv.host, srv.port), timeout=5)
        plain.sendall(b"\x00" * 64)
        plain.settimeout(5)
        try:
            assert plain.recv(8) != struct.pack("<II", 1, 1)
        except (ConnectionResetError, socket.timeout):
            pass
        plain.close()
        s, version, server_type = ic.connect(srv.host, srv.port, tls=True)
        assert (version, server_type) == (1, 1)
        s.sendall(ic.init_frame(metric_kind=2, quantization=1, dim=3, m=12, efc=64, ef=32, capacity=4, element_bits=32))
        assert ic.read_error(s) == "Invalid metric 2"
        s.close()
    finally:
        srv.stop()
    # a certificate without its key, or files that are not PEM, are refused with a message
    with pytest.raises(capi.LanternGpuError, match="both a certificate and a private key"):
        capi.IndexServer("127.0.0.1", 0, 0, "/tmp", cert=cert)
    with pytest.raises(capi.LanternGpuError, match="cannot load the certificate"):
        capi.IndexServer("127.0.0.1", 0, 0, "/tmp", cert=key, key=cert)


@pytest.mark.gpu
def test_full_build_round_trip_over_tls(capi, tls_files):
    cert, key = tls_files
    rng = np.random.default_rng(4)
    n, d = 3000, 64
    base = rng.standard_normal((n, d), dtype=np.float32)
    srv = capi.IndexServer("127.0.0.1", 0, 0, "/tmp", cert=cert, key=key)
    try:
        added, blob = ic.build_index(srv.host, srv.port, 3, d, [r.tobytes() for r in base], np.arange(n) + 1, m=8, efc=40, ef=32, tls=True)
    finally:
        srv.stop()
    assert added == n
    direct = capi.GpuIndex("l2sq", d, M=8, ef_construction=40, ef=32, seed=42)
    direct.add_many(np.arange(n, dtype=np.uint64) + 1, base)
    assert direct.save_buffer() == blob  # the same index file as the plain server's and as a local build
