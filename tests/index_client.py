"""Test-side client of the external indexing protocol: what PostgreSQL does in
lantern_hnsw/src/hnsw/external_index_socket.c (create_external_index_session :322-486,
external_index_send_tuple :517-536, external_index_receive_metadata :488-515) and what the Rust
tests do in lantern_cli/tests/external_index_server_test.rs:141-326."""
import socket
import struct

PROTOCOL_VERSION, SERVER_TYPE_INDEXER = 1, 1
INIT_MSG, END_MSG, ERR_MSG = 0x13333337, 0x31333337, 0x37333337


class IndexServerError(RuntimeError):
    pass


def recv_exact(s, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = s.recv(n - len(buf))
        if not chunk:
            raise IndexServerError("connection closed")
        buf += chunk
    return bytes(buf)


def connect(host, port, tls=False):
    s = socket.create_connection((host, port), timeout=30)
    if tls:  # what init_ssl does on the PostgreSQL side (external_index_socket_ssl.c:39-62): TLS >= 1.2, no certificate verification
        import ssl

        ctx = ssl.SSLContext(ssl.PROTOCOL_TLS_CLIENT)
        ctx.check_hostname = False
        ctx.verify_mode = ssl.CERT_NONE
        ctx.minimum_version = ssl.TLSVersion.TLSv1_2
        s = ctx.wrap_socket(s)
    version, server_type = struct.unpack("<II", recv_exact(s, 8))
    return s, version, server_type


def read_error(s, first4=None):
    hdr = first4 if first4 is not None else recv_exact(s, 4)
    assert struct.unpack("<I", hdr)[0] == ERR_MSG, hdr
    (n,) = struct.unpack("<I", recv_exact(s, 4))
    return recv_exact(s, n).decode()


def init_frame(metric_kind, quantization, dim, m, efc, ef, capacity, element_bits, pq=0, num_centroids=0, num_subvectors=0):
    # external_index_params_t field order: external_index_socket.h:24-38
    return struct.pack("<12I", INIT_MSG, pq, metric_kind, quantization, dim, m, efc, ef, num_centroids, num_subvectors, capacity, element_bits)


def build_index(host, port, metric_kind, dim, rows, labels, m=16, efc=128, ef=64, element_bits=32, quantization=1, capacity=None,
                codebook=None, num_subvectors=0, tls=False):
    """Returns (num_added, index_file_bytes).  rows: bytes-like per row.  codebook: [num_centroids][dim] f32 rows (pq = true):
    sent centroid by centroid, then END_MSG (external_index_send_codebook, external_index_socket.c:304-320)."""
    s, version, server_type = connect(host, port, tls)
    assert (version, server_type) == (PROTOCOL_VERSION, SERVER_TYPE_INDEXER)
    if codebook is None:
        s.sendall(init_frame(metric_kind, quantization, dim, m, efc, ef, capacity if capacity is not None else len(labels), element_bits))
    else:
        s.sendall(init_frame(metric_kind, quantization, dim, m, efc, ef, capacity if capacity is not None else len(labels), element_bits,
                             pq=1, num_centroids=len(codebook), num_subvectors=num_subvectors))
        for row in codebook:
            s.sendall(bytes(row))
        s.sendall(struct.pack("<I", END_MSG))
    status = recv_exact(s, 1)
    if status != b"\x00":
        raise IndexServerError(read_error(s, status + recv_exact(s, 3)))
    for label, row in zip(labels, rows):
        s.sendall(struct.pack("<Q", int(label)) + bytes(row))
    s.sendall(struct.pack("<I", END_MSG))
    s.settimeout(600)  # external_index_socket.c:502 disables the read timeout while the index is built
    head = recv_exact(s, 4)
    if struct.unpack("<I", head)[0] == ERR_MSG:
        raise IndexServerError(read_error(s, head))
    (num_added,) = struct.unpack("<Q", head + recv_exact(s, 4))
    (size,) = struct.unpack("<Q", recv_exact(s, 8))
    data = recv_exact(s, size)
    s.close()
    return num_added, data
