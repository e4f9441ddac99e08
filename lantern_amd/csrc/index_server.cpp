// index_server.cpp -- the external indexing server (boundary B3): a drop-in for
// `lantern-cli start-indexing-server` (lantern_cli/src/external_index/server.rs) whose worker pool is
// replaced by the device builder.  PostgreSQL connects to it from `CREATE INDEX ... WITH (external=true)`
// (lantern_hnsw/src/hnsw/external_index_socket.c:322-536) and is not changed at all.
//
// Wire protocol, little-endian only (external_index_socket.c:337-339):
//   server -> u32 PROTOCOL_VERSION (1), u32 SERVER_TYPE (1 = indexing server)          server.rs:183-184
//   client -> u32 INIT_MSG 0x13333337 + external_index_params_t (11 x u32)             external_index_socket.h:24-38
//   [pq = 1: num_centroids codebook frames of dim f32 each, then END_MSG                server.rs:107-127, external_index_socket.c:304-320]
//   server -> u8 0                                                                      server.rs:206
//   client -> rows [u64 label][dim * element_bits/8 bytes | ceil(dim/8) bytes if bits<8] server.rs:226-230, :169-174
//   client -> u32 END_MSG 0x31333337
//   server -> u64 rows added, u64 index file size, usearch-format file bytes            server.rs:388-422
//   on any error: u32 ERR_MSG 0x37333337, u32 length, message                           server.rs:561-573
// One connection is served at a time, as in the reference (server.rs:537-583).
#include <arpa/inet.h>
#include <dlfcn.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lantern_gpu.h"
#include "abi_guard.hpp"

namespace {

constexpr uint32_t PROTOCOL_VERSION = 1, SERVER_TYPE = 1;
constexpr uint32_t INIT_MSG = 0x13333337u, END_MSG = 0x31333337u, ERR_MSG = 0x37333337u;
constexpr size_t   INDEX_HEADER_LENGTH = 4 * 12;  // magic + 11 params (server.rs:33-35)
constexpr int      SOCKET_TIMEOUT_S = 10;         // server.rs:26, external_index_socket.h:17
constexpr size_t   ADD_CHUNK = 8192;              // rows handed to the device builder at a time

struct Fail { std::string msg; };

enum Status { IDLE = 0, IN_PROGRESS = 1, FAILED = 2, SUCCEEDED = 3 };  // server.rs:44-49

}  // namespace

struct lantern_index_server
{
    int                 listen_fd = -1, status_fd = -1;
    int                 port = 0, status_port = 0;
    std::atomic<bool>   stop{ false };
    std::atomic<int>    status{ IDLE };
    std::atomic<long long> status_updated_at{ 0 };
    std::atomic<uint64_t>  served{ 0 };
    std::thread         accept_thread, status_thread;
    std::string         tmp_dir;
    void               *tls_ctx = nullptr;  // SSL_CTX of a server started with a certificate (server.rs:454-470), else NULL
};

namespace {

long long now_ms() { return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count(); }
void set_status(lantern_index_server *s, int st) { s->status = st; s->status_updated_at = now_ms(); }

// ---- TLS (lantern_cli start-indexing-server --cert C --key K: server.rs:437-470,548; the PostgreSQL side connects with OpenSSL and
// does not verify the certificate: external_index_socket_ssl.c:6-62).  libssl is bound at run time (as RCCL is in comm.cpp): a
// host without it can still run the plain server.
struct TlsApi
{
    void *lib = nullptr;
    const void *(*TLS_server_method)() = nullptr;
    void *(*SSL_CTX_new)(const void *) = nullptr;
    void (*SSL_CTX_free)(void *) = nullptr;
    int (*SSL_CTX_use_certificate_chain_file)(void *, const char *) = nullptr;
    int (*SSL_CTX_use_PrivateKey_file)(void *, const char *, int) = nullptr;
    int (*SSL_CTX_check_private_key)(const void *) = nullptr;
    long (*SSL_CTX_ctrl)(void *, int, long, void *) = nullptr;
    void *(*SSL_new)(void *) = nullptr;
    void (*SSL_free)(void *) = nullptr;
    int (*SSL_set_fd)(void *, int) = nullptr;
    int (*SSL_accept)(void *) = nullptr;
    int (*SSL_read)(void *, void *, int) = nullptr;
    int (*SSL_write)(void *, const void *, int) = nullptr;
    int (*SSL_shutdown)(void *) = nullptr;
    bool ok = false;
};
TlsApi *tls_api()
{
    static TlsApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for(const char *name : { "libssl.so.3", "libssl.so", "libssl.so.1.1" }) {
            api.lib = ::dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if(api.lib) break;
        }
        if(!api.lib) return;
        bool all = true;
#define TLS_SYM(f) all = ((*(void **)&api.f = ::dlsym(api.lib, #f)) != nullptr) && all
        TLS_SYM(TLS_server_method); TLS_SYM(SSL_CTX_new); TLS_SYM(SSL_CTX_free); TLS_SYM(SSL_CTX_use_certificate_chain_file);
        TLS_SYM(SSL_CTX_use_PrivateKey_file); TLS_SYM(SSL_CTX_check_private_key); TLS_SYM(SSL_CTX_ctrl); TLS_SYM(SSL_new); TLS_SYM(SSL_free);
        TLS_SYM(SSL_set_fd); TLS_SYM(SSL_accept); TLS_SYM(SSL_read); TLS_SYM(SSL_write); TLS_SYM(SSL_shutdown);
#undef TLS_SYM
        api.ok = all;
    });
    return &api;
}

// one accepted connection: the socket, and the TLS session over it if the server has a certificate
struct Wire
{
    int   fd = -1;
    void *ssl = nullptr;
    ssize_t recv_some(void *buf, size_t n)
    {
        if(ssl) return (ssize_t)tls_api()->SSL_read(ssl, buf, (int)std::min<size_t>(n, 1u << 30));
        return ::recv(fd, buf, n, 0);
    }
    ssize_t send_some(const void *buf, size_t n)
    {
        if(ssl) return (ssize_t)tls_api()->SSL_write(ssl, buf, (int)std::min<size_t>(n, 1u << 30));
        return ::send(fd, buf, n, MSG_NOSIGNAL);
    }
};

void write_all(Wire &w, const void *buf, size_t n)
{
    const char *p = (const char *)buf;
    while(n) {
        ssize_t sent = w.send_some(p, n);
        if(sent <= 0) throw Fail{ "socket write failed" };
        p += sent;
        n -= (size_t)sent;
    }
}
// gather-write of lantern_gpu_save_stream's spans: sendmsg (writev with MSG_NOSIGNAL), resumed after partial writes; over TLS one
// record stream, span by span
static_assert(sizeof(lantern_gpu_span) == sizeof(struct iovec) && offsetof(lantern_gpu_span, data) == offsetof(struct iovec, iov_base) &&
                  offsetof(lantern_gpu_span, size) == offsetof(struct iovec, iov_len),
              "lantern_gpu_span must be layout-compatible with struct iovec");
int send_spans(void *ctx, const lantern_gpu_span *spans, size_t count)
{
    Wire &wire = *(Wire *)ctx;
    if(wire.ssl) {
        for(size_t i = 0; i < count; ++i) {
            const char *p = (const char *)spans[ i ].data;
            size_t      n = spans[ i ].size;
            while(n) {
                const ssize_t sent = wire.send_some(p, n);
                if(sent <= 0) return -1;
                p += sent;
                n -= (size_t)sent;
            }
        }
        return 0;
    }
    const int    fd = wire.fd;
    struct iovec iov[ 1024 ];
    while(count) {
        const size_t m = std::min<size_t>(count, 1024);
        std::memcpy(iov, spans, m * sizeof(struct iovec));
        size_t first = 0;
        while(first < m) {
            struct msghdr mh;
            std::memset(&mh, 0, sizeof(mh));
            mh.msg_iov = iov + first;
            mh.msg_iovlen = m - first;
            ssize_t w = ::sendmsg(fd, &mh, MSG_NOSIGNAL);
            if(w <= 0) return -1;
            while(first < m && (size_t)w >= iov[ first ].iov_len) w -= (ssize_t)iov[ first++ ].iov_len;
            if(first < m && w > 0) {
                iov[ first ].iov_base = (char *)iov[ first ].iov_base + w;
                iov[ first ].iov_len -= (size_t)w;
            }
        }
        spans += m;
        count -= m;
    }
    return 0;
}
void read_exact(Wire &w, void *buf, size_t n)
{
    char *p = (char *)buf;
    while(n) {
        ssize_t r = w.recv_some(p, n);
        if(r <= 0) throw Fail{ "failed to fill whole buffer" };
        p += r;
        n -= (size_t)r;
    }
}

enum Frame { FRAME_INIT, FRAME_DATA, FRAME_EXIT };

// read_frame (server.rs:275-309): one read, then fill up to expected_size
Frame read_frame(Wire &fd, std::vector<uint8_t> &buf, size_t expected_size, bool want_init)
{
    buf.assign(expected_size, 0);
    ssize_t got = fd.recv_some(buf.data(), expected_size);
    if(got < 4) throw Fail{ "Invalid frame received" };
    uint32_t hdr;
    std::memcpy(&hdr, buf.data(), 4);
    if(hdr == END_MSG) return FRAME_EXIT;
    if(want_init && hdr != INIT_MSG) throw Fail{ "Invalid message header" };
    if(expected_size > (size_t)got) read_exact(fd, buf.data() + got, expected_size - (size_t)got);
    return hdr == INIT_MSG ? FRAME_INIT : FRAME_DATA;
}

void serve(lantern_index_server *srv, Wire &fd)
{
    usearch_index_t index = nullptr;
    usearch_error_t err = nullptr;
    std::string     failure;
    bool            failed = false, length_sent = false;
    try {
        uint32_t hello[ 2 ] = { PROTOCOL_VERSION, SERVER_TYPE };
        write_all(fd, hello, 8);
        std::vector<uint8_t> buf;
        if(read_frame(fd, buf, INDEX_HEADER_LENGTH, true) != FRAME_INIT) throw Fail{ "send init message first" };
        uint32_t p[ 11 ];
        std::memcpy(p, buf.data() + 4, sizeof(p));
        const uint32_t pq = p[ 0 ], metric_kind = p[ 1 ], quant = p[ 2 ], dim = p[ 3 ], m = p[ 4 ], efc = p[ 5 ], ef = p[ 6 ],
                       num_centroids = p[ 7 ], num_subvectors = p[ 8 ], estimated_capacity = p[ 9 ], element_bits = p[ 10 ];
        if(quant > 5) throw Fail{ "Invalid scalar quantization" };                                        // server.rs:94-101
        if(metric_kind != 1 && metric_kind != 3 && metric_kind != 8) throw Fail{ "Invalid metric " + std::to_string(metric_kind) };  // cli.rs:56-69
        // every size below comes from the socket: bound it before anything is allocated from it (Lantern itself caps
        // vectors at HNSW_MAX_DIM = 2000 scalars, options.h:14; bit vectors arrive as 32 x that many dimensions)
        if(dim == 0 || dim > 2000u * 32u) throw Fail{ "Invalid dimensions " + std::to_string(dim) };
        if(element_bits != 1 && element_bits != 8 && element_bits != 16 && element_bits != 32) throw Fail{ "Invalid element bits " + std::to_string(element_bits) };
        usearch_init_options_t o;
        std::memset(&o, 0, sizeof(o));
        o.metric_kind = (usearch_metric_kind_t)metric_kind;
        o.quantization = quant <= 1 ? usearch_scalar_f32_k : (usearch_scalar_kind_t)quant;
        o.dimensions = dim;
        o.connectivity = m;
        o.expansion_add = efc;
        o.expansion_search = ef;
        o.pq = pq == 1;
        o.num_centroids = num_centroids;
        o.num_subvectors = num_subvectors;
        std::vector<float> codebook;
        if(o.pq) {  // server.rs:107-127: frames of `dim` floats until END_MSG; row c = centroid c of every subvector, concatenated
            if(num_centroids == 0 || num_centroids > 256) throw Fail{ "Invalid number of centroids" };
            for(;;) {
                Frame f = read_frame(fd, buf, (size_t)dim * 4, false);
                if(f == FRAME_EXIT) break;
                if(f != FRAME_DATA) throw Fail{ "Invalid message received" };
                if(codebook.size() >= (size_t)256 * dim) throw Fail{ "Codebook larger than 256 centroids" };
                const float *row = (const float *)buf.data();
                codebook.insert(codebook.end(), row, row + dim);
            }
            if(codebook.size() != (size_t)num_centroids * dim) throw Fail{ "Codebook size does not match num_centroids" };
        }
        index = usearch_init(&o, o.pq ? codebook.data() : nullptr, &err);
        if(err) throw Fail{ err };
        usearch_reserve(index, estimated_capacity, &err);
        if(err) throw Fail{ err };
        const uint8_t okb = 0;
        write_all(fd, &okb, 1);

        // receive_rows (server.rs:214-267)
        const size_t dims = usearch_dimensions(index, &err);
        const size_t vec_bytes = element_bits < 8 ? (dims + 7) / 8 : dims * (element_bits / 8);
        const size_t payload = 8 + vec_bytes;
        // the rows' scalar kind follows element_bits (server.rs:226-230, add_raw(label, bytes, element_bits) :349)
        const usearch_scalar_kind_t kind = element_bits < 8 ? usearch_scalar_b1_k : element_bits == 8 ? usearch_scalar_i8_k : element_bits == 16 ? usearch_scalar_f16_k : usearch_scalar_f32_k;
        // The socket is drained by THIS thread while a builder thread hands finished chunks to the device: the reference
        // overlaps the two the same way (a reader feeding a channel that a pool of add_raw workers drains: server.rs:214-267,
        // :317-359).  At most four chunks wait between the two; a failed add stops the reader at its next chunk.
        // row buffers are page-locked and recycled (six of them): a chunk's 50 MB go up at the host link's rate, not through the
        // runtime's pageable staging (which cost as much as the device batches of the chunk: the builder was the stream's limit)
        struct Chunk { std::vector<uint64_t> labels; uint8_t *rows = nullptr; size_t bytes = 0; };
        const size_t chunk_rows = std::max<size_t>(64, std::min<size_t>(ADD_CHUNK, (64u << 20) / std::max<size_t>(vec_bytes, 1)));
        // where the stream's time goes (LANTERN_INDEX_SERVER_TRACE=1 prints it to stderr when the rows have been received)
        double t_recv = 0, t_parse = 0, t_blocked = 0, t_add = 0, t_idle = 0;
        auto   now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        std::mutex              qmu;
        std::condition_variable qcv;
        std::deque<Chunk>       ready;
        const size_t            buf_bytes = chunk_rows * vec_bytes;
        std::vector<uint8_t *>  pool, all_bufs;
        std::vector<std::vector<uint8_t>> fallback;  // ordinary memory if page-locking fails
        for(int i = 0; i < 6; ++i) {
            uint8_t *b = (uint8_t *)lantern_gpu_host_alloc(buf_bytes);
            if(!b) {
                fallback.emplace_back(buf_bytes);
                b = fallback.back().data();
            } else {
                all_bufs.push_back(b);
            }
            pool.push_back(b);
        }
        struct PoolGuard { std::vector<uint8_t *> &v; ~PoolGuard() { for(uint8_t *b : v) lantern_gpu_host_free(b); } } pool_guard{ all_bufs };
        bool                    closed = false;
        std::string             add_error;
        std::thread builder([&] {
            for(;;) {
                Chunk c;
                {
                    const double                 w0 = now_s();
                    std::unique_lock<std::mutex> lk(qmu);
                    qcv.wait(lk, [&] { return closed || !ready.empty(); });
                    t_idle += now_s() - w0;
                    if(ready.empty()) return;
                    c = std::move(ready.front());
                    ready.pop_front();
                }
                qcv.notify_all();
                usearch_error_t e2 = nullptr;
                const double    a0 = now_s();
                lantern_gpu_add_many(index, c.labels.data(), c.rows, c.labels.size(), kind, &e2);
                t_add += now_s() - a0;
                {
                    std::lock_guard<std::mutex> g(qmu);
                    pool.push_back(c.rows);  // (add_many has copied the rows to the device: the buffer is free again)
                }
                qcv.notify_all();
                if(e2) {
                    std::lock_guard<std::mutex> g(qmu);
                    add_error = e2;
                    closed = true;
                    ready.clear();
                    qcv.notify_all();
                    return;
                }
            }
        });
        // hand a filled chunk to the builder and take an empty buffer for the next one (waiting for the builder if all six are full)
        auto take_buffer = [&](Chunk &c) {
            const double                 b0 = now_s();
            std::unique_lock<std::mutex> lk(qmu);
            qcv.wait(lk, [&] { return closed || !pool.empty(); });
            t_blocked += now_s() - b0;
            if(closed || pool.empty()) return false;
            c.rows = pool.back();
            pool.pop_back();
            c.bytes = 0;
            c.labels.clear();
            c.labels.reserve(chunk_rows);
            return true;
        };
        auto hand_over = [&](Chunk &c) {
            if(c.labels.empty()) return true;
            {
                std::lock_guard<std::mutex> g(qmu);
                if(closed) return false;
                ready.push_back(std::move(c));
            }
            c = Chunk();
            qcv.notify_all();
            return take_buffer(c);
        };
        std::string read_failure;
        try {
            // The tuples arrive one write each (external_index_socket.c:517-536) but the kernel coalesces them: the socket is
            // read a megabyte at a time and the frames are cut out of the buffer ([u64 label][vector] each; a 4-byte END_MSG
            // closes the stream -- the reference's read_frame looks at the first four bytes of what one read returned in the same
            // way, server.rs:275-309).
            Chunk cur;
            if(!take_buffer(cur)) throw Fail{ "indexing server: no row buffer" };
            std::vector<uint8_t> inbuf(std::max<size_t>(1u << 20, payload * 4));
            size_t have = 0;
            bool   ended = false;
            while(!ended) {
                const double  r0 = now_s();
                const ssize_t got = fd.recv_some(inbuf.data() + have, inbuf.size() - have);
                const double  r1 = now_s();
                t_recv += r1 - r0;
                if(got <= 0) throw Fail{ have ? "failed to fill whole buffer" : "Invalid frame received" };
                have += (size_t)got;
                size_t pos = 0;
                for(;;) {
                    if(have - pos < 4) break;
                    uint32_t hdr;
                    std::memcpy(&hdr, inbuf.data() + pos, 4);
                    // a frame that starts with END_MSG ends the stream (the reference decides on the first four bytes of a frame in
                    // the same way; a tuple whose label happens to start with these bytes is misread there too)
                    if(hdr == END_MSG) { ended = true; break; }
                    if(have - pos < payload) break;
                    uint64_t label;
                    std::memcpy(&label, inbuf.data() + pos, 8);
                    cur.labels.push_back(label);
                    std::memcpy(cur.rows + cur.bytes, inbuf.data() + pos + 8, vec_bytes);
                    cur.bytes += vec_bytes;
                    pos += payload;
                    if(cur.labels.size() == chunk_rows && !hand_over(cur)) { ended = true; break; }
                }
                if(pos) {
                    std::memmove(inbuf.data(), inbuf.data() + pos, have - pos);
                    have -= pos;
                }
                t_parse += now_s() - r1;
            }
            if(!cur.labels.empty()) {  // the last, partial chunk (no further buffer is needed behind it)
                std::lock_guard<std::mutex> g(qmu);
                if(!closed) ready.push_back(std::move(cur));
            }
            qcv.notify_all();
        } catch(const Fail &f) {
            read_failure = f.msg;
        } catch(const std::exception &ex) {
            read_failure = std::string("indexing server: ") + ex.what();
        }
        {
            std::lock_guard<std::mutex> g(qmu);
            closed = true;
        }
        qcv.notify_all();
        builder.join();
        if(std::getenv("LANTERN_INDEX_SERVER_TRACE"))
            std::fprintf(stderr, "index server: reader recv %.2f s, parse+copy %.2f s (of which blocked on the builder %.2f s); builder add_many %.2f s, idle %.2f s\n",
                         t_recv, t_parse, t_blocked, t_add, t_idle);
        if(!read_failure.empty()) throw Fail{ read_failure };
        if(!add_error.empty()) throw Fail{ add_error };
        lantern_gpu_flush(index, &err);
        if(err) throw Fail{ err };

        // the build may take longer than the socket timeout on the client side; the client disables its
        // read timeout while it waits (external_index_socket.c:502)
        const uint64_t count = usearch_size(index, &err);
        write_all(fd, &count, 8);
        const size_t len = usearch_serialized_length(index, &err);
        if(err) throw Fail{ err };
        const uint64_t len64 = len;
        write_all(fd, &len64, 8);
        length_sent = true;  // from here on the client reads `len64` bytes of index file: a failure may only cut the stream short
        // the file goes out as it is formatted: node prefixes + vector bytes from page-locked row chunks, gathered by sendmsg
        // (r2 built the whole 6.4 GB file in memory first: 5 of the 9 s of a 1M x 1536 build)
        lantern_gpu_save_stream(index, send_spans, &fd, &err);
        if(err) throw Fail{ err };
        set_status(srv, SUCCEEDED);
    } catch(const Fail &f) {
        failure = f.msg;
        failed = true;
    } catch(const std::exception &ex) {  // bad_alloc / length_error from a network-supplied size: an error frame, not an abort
        failure = std::string("indexing server: ") + ex.what();
        failed = true;
    } catch(...) {
        failure = "indexing server: unexpected failure";
        failed = true;
    }
    if(failed) set_status(srv, FAILED);
    // An error frame is only a frame BEFORE the file length has gone out (the reference builds the whole file first, so its
    // failures always arrive as clean frames: server.rs:382-435).  Once the client is reading file bytes, anything appended
    // would be parsed as node tapes: the stream is simply closed and the client sees a short read.
    if(failed && !length_sent) {
        uint8_t        out[ 8 ];
        const uint32_t hdr = ERR_MSG, n = (uint32_t)failure.size();
        std::memcpy(out, &hdr, 4);
        std::memcpy(out + 4, &n, 4);
        (void)fd.send_some(out, 8);
        (void)fd.send_some(failure.data(), failure.size());
    }
    if(index) usearch_free(index, &err);
    srv->served++;
}

int listen_on(const char *host, int port, int *bound_port)
{
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if(fd < 0) return -1;
    int one = 1;
    ::setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in a;
    std::memset(&a, 0, sizeof(a));
    a.sin_family = AF_INET;
    a.sin_port = htons((uint16_t)port);
    if(::inet_pton(AF_INET, host && *host ? host : "0.0.0.0", &a.sin_addr) != 1 || ::bind(fd, (sockaddr *)&a, sizeof(a)) != 0 ||
       ::listen(fd, 16) != 0) {
        ::close(fd);
        return -1;
    }
    socklen_t len = sizeof(a);
    ::getsockname(fd, (sockaddr *)&a, &len);
    *bound_port = ntohs(a.sin_port);
    timeval tv{ 0, 200000 };  // so the accept loop can notice `stop`
    ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    return fd;
}

void accept_loop(lantern_index_server *srv)
{
    while(!srv->stop) {
        int fd = ::accept(srv->listen_fd, nullptr, nullptr);
        if(fd < 0) continue;
        timeval tv{ SOCKET_TIMEOUT_S, 0 };
        ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
        ::setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
        int one = 1;
        ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        Wire wire;
        wire.fd = fd;
        if(srv->tls_ctx) {  // the handshake first (server.rs:548: a rustls ServerConnection per accepted stream)
            TlsApi *t = tls_api();
            wire.ssl = t->SSL_new(srv->tls_ctx);
            if(!wire.ssl || t->SSL_set_fd(wire.ssl, fd) != 1 || t->SSL_accept(wire.ssl) != 1) {
                if(wire.ssl) t->SSL_free(wire.ssl);
                ::close(fd);
                continue;  // not a TLS client (or it gave up): nothing was served
            }
        }
        set_status(srv, IN_PROGRESS);
        serve(srv, wire);
        if(wire.ssl) {
            (void)tls_api()->SSL_shutdown(wire.ssl);
            tls_api()->SSL_free(wire.ssl);
        }
        ::shutdown(fd, SHUT_RDWR);
        ::close(fd);
    }
}

// the optional HTTP status endpoint: {"status":0..3,"status_updated_at":ms} (server.rs:586-597)
void status_loop(lantern_index_server *srv)
{
    while(!srv->stop) {
        int fd = ::accept(srv->status_fd, nullptr, nullptr);
        if(fd < 0) continue;
        char   req[ 1024 ];
        timeval tv{ 1, 0 };
        ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
        (void)::recv(fd, req, sizeof(req), 0);
        std::string body = "{\"status\":" + std::to_string(srv->status.load()) + ",\"status_updated_at\":" +
                           std::to_string(srv->status_updated_at.load()) + "}";
        std::string resp = "HTTP/1.1 200 OK\r\nContent-Type: application/json\r\nContent-Length: " + std::to_string(body.size()) +
                           "\r\nConnection: close\r\n\r\n" + body;
        (void)::send(fd, resp.data(), resp.size(), MSG_NOSIGNAL);
        ::close(fd);
    }
}

}  // namespace

extern "C" {

lantern_index_server_t *lantern_index_server_start_tls(const char *host, int port, int status_port, const char *tmp_dir, const char *cert_pem,
                                                       const char *key_pem, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    void *ctx = nullptr;
    if((cert_pem != nullptr) != (key_pem != nullptr)) { if(e) *e = "lantern_gpu: TLS needs both a certificate and a private key"; return nullptr; }
    if(cert_pem) {
        TlsApi *t = tls_api();
        if(!t->ok) { if(e) *e = "lantern_gpu: libssl could not be loaded: TLS is not available on this host"; return nullptr; }
        ctx = t->SSL_CTX_new(t->TLS_server_method());
        if(!ctx) { if(e) *e = "lantern_gpu: could not create the TLS context"; return nullptr; }
        (void)t->SSL_CTX_ctrl(ctx, 123 /* SSL_CTRL_SET_MIN_PROTO_VERSION */, 0x0303 /* TLS 1.2 */, nullptr);  // external_index_socket_ssl.c:13-21
        (void)t->SSL_CTX_ctrl(ctx, 33 /* SSL_CTRL_MODE */, 4 /* SSL_MODE_AUTO_RETRY */, nullptr);             // :31
        if(t->SSL_CTX_use_certificate_chain_file(ctx, cert_pem) != 1 || t->SSL_CTX_use_PrivateKey_file(ctx, key_pem, 1 /* PEM */) != 1 ||
           t->SSL_CTX_check_private_key(ctx) != 1) {
            t->SSL_CTX_free(ctx);
            if(e) *e = "lantern_gpu: cannot load the certificate / private key (PEM files expected)";
            return nullptr;
        }
    }
    lantern_index_server *s = new lantern_index_server();
    s->tls_ctx = ctx;
    s->tmp_dir = tmp_dir ? tmp_dir : "/tmp";
    s->listen_fd = listen_on(host, port, &s->port);
    if(s->listen_fd < 0) {
        if(e) *e = "lantern_gpu: cannot bind the indexing server socket";
        if(ctx) tls_api()->SSL_CTX_free(ctx);
        delete s;
        return nullptr;
    }
    set_status(s, IDLE);
    if(status_port >= 0) {
        s->status_fd = listen_on(host, status_port, &s->status_port);
        if(s->status_fd >= 0) s->status_thread = std::thread(status_loop, s);
    }
    s->accept_thread = std::thread(accept_loop, s);
    return s;
}
LANTERN_ABI_CATCH(e)

lantern_index_server_t *lantern_index_server_start(const char *host, int port, int status_port, const char *tmp_dir, usearch_error_t *e)
{
    return lantern_index_server_start_tls(host, port, status_port, tmp_dir, nullptr, nullptr, e);
}

int lantern_index_server_port(lantern_index_server_t *s) { return s ? s->port : -1; }
int lantern_index_server_status_port(lantern_index_server_t *s) { return s ? s->status_port : -1; }
int lantern_index_server_status(lantern_index_server_t *s) { return s ? s->status.load() : -1; }
uint64_t lantern_index_server_served(lantern_index_server_t *s) { return s ? s->served.load() : 0; }

void lantern_index_server_stop(lantern_index_server_t *s)
try {
    if(!s) return;
    s->stop = true;
    if(s->accept_thread.joinable()) s->accept_thread.join();
    if(s->status_thread.joinable()) s->status_thread.join();
    if(s->listen_fd >= 0) ::close(s->listen_fd);
    if(s->status_fd >= 0) ::close(s->status_fd);
    if(s->tls_ctx) tls_api()->SSL_CTX_free(s->tls_ctx);
    delete s;
}
LANTERN_ABI_CATCH_VOID(nullptr)

}  // extern "C"
