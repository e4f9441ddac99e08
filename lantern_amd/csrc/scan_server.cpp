// scan_server.cpp -- the scan-side service: ONE HBM-resident index serving the k-NN queries of many PostgreSQL
// backends, batched ("keep an HBM mirror of the index ..., serve amgettuple by batching
// concurrent backends' queries to the GPU").
//
// Why a service: PostgreSQL is process-per-backend and ldb_amgettuple (lantern_hnsw/src/hnsw/scan.c:167-338) asks for
// one query at a time.  A lone walk on the device is latency-bound (DESIGN.md section 5, config[1]: ~0.4 ms), while a
// batch of walks runs at the HBM roofline -- and 64 backends each holding their own 3 GB mirror is not an option.  So
// the mirror lives in this one process; a backend's scan sends its query here (lantern_scan_client_*: what
// ldb_amgettuple calls instead of usearch_search_ef), the server coalesces whatever arrived within a short window
// into one lantern_gpu_search_batch launch and routes the answers back.
//
// Wire protocol (ours; the reference has no such component), little-endian:
//   client -> u32 0x5152534C ("LSRQ"), u32 k, u32 ef (0 = index default), u32 vector bytes, the vector
//             or u32 0x4352534C ("LSRC"), same fields: the CONTINUATION of this connection's scan -- the next k rows of
//             the same query (usearch_search_ef(streaming = true), scan.c:273-281)
//   server -> u32 0x5052534C ("LSRP"), u32 status (0 = ok), u32 count, count x u64 labels, count x f32 distances
//             status != 0: u32 length, message
// A connection carries one request at a time (a backend runs one scan step at a time) and stays open across requests.
// The continuation state -- which rows this scan has been handed since its last "LSRQ" -- belongs to the CONNECTION
// (one backend, one scan at a time), never to the shared index: any number of backends paginate concurrently.  A
// continuation is served as a search for |handed out| + k rows from which the rows already handed out are dropped
// (by label: a heap TID is indexed once; rows with label 0 -- deleted, skipped by the scan -- are dropped by count).
//
// Threads: an acceptor; one reader per connection (blocking read -> enqueue -> wait for the answer -> write); one
// dispatcher that takes up to `max_batch` queued requests -- waiting at most `max_wait_us` after the first one for
// company -- groups them by (k, ef) and calls the batch search function once per group.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../include/lantern_gpu.h"

namespace {

constexpr uint32_t REQ_MAGIC = 0x5152534Cu, CONT_MAGIC = 0x4352534Cu, REP_MAGIC = 0x5052534Cu;
constexpr uint32_t MAX_K = 4096, MAX_VEC_BYTES = 1u << 20;

bool read_exact(int fd, void *buf, size_t n)
{
    char *p = (char *)buf;
    while(n) {
        ssize_t r = ::recv(fd, p, n, 0);
        if(r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}
bool write_all(int fd, const void *buf, size_t n)
{
    const char *p = (const char *)buf;
    while(n) {
        ssize_t r = ::send(fd, p, n, MSG_NOSIGNAL);
        if(r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}

struct Pending
{
    uint32_t             k = 0, ef = 0;
    std::vector<uint8_t> vec;
    // filled by the dispatcher
    bool                     done = false;
    std::string              error;
    std::vector<uint64_t>    labels;
    std::vector<float>       dists;
    std::mutex               mu;
    std::condition_variable  cv;
};

}  // namespace

struct lantern_scan_server
{
    lantern_batch_search_fn fn = nullptr;
    void                   *fn_ctx = nullptr;
    usearch_index_t         index = nullptr;  // the default backend: lantern_gpu_search_batch on this index
    usearch_scalar_kind_t   kind = usearch_scalar_f32_k;
    size_t                  vec_bytes = 0, max_batch = 256;
    unsigned                max_wait_us = 200;
    int                     listen_fd = -1, port = 0;
    std::atomic<bool>       stop{ false };
    std::thread             accept_thread, dispatch_thread[ 2 ];
    int                     lanes = 1;   // dispatchers: one collects the next batch while the other's batch is on the device
    std::mutex              collect_mu;  // held by the dispatcher that is collecting (one batch is formed at a time)
    std::mutex              mu;  // queue + connection list
    std::condition_variable cv;
    std::deque<std::shared_ptr<Pending>> queue;
    struct Conn
    {
        std::thread       t;
        int               fd = -1;
        std::atomic<bool> done{ false };
    };
    std::vector<std::unique_ptr<Conn>> conns;  // guarded by mu
    size_t                  open_conns = 0, in_flight = 0;  // guarded by mu: connections being served; requests taken into a batch and not answered yet
    std::atomic<uint64_t>   n_requests{ 0 }, n_batches{ 0 }, n_launches{ 0 }, max_batch_seen{ 0 };
    std::atomic<uint64_t>   batch_hist[ 16 ] = {};  // batches by size: bin b counts sizes in [2^b, 2^(b+1))
};

namespace {

thread_local int tl_lane = 0;  // which dispatcher this thread is (the default backend's lane)

int default_backend(void *ctx, const void *queries, size_t nq, size_t, size_t k, size_t ef, uint64_t *labels, float *dists, uint32_t *counts,
                    const char **err)
{
    lantern_scan_server *s = (lantern_scan_server *)ctx;
    usearch_error_t      e = nullptr;
    lantern_gpu_search_batch_lane(s->index, tl_lane, queries, nq, s->kind, k, ef, labels, dists, counts, &e);
    if(e) { *err = e; return 1; }
    return 0;
}

void fulfil(const std::shared_ptr<Pending> &p)
{
    {
        std::lock_guard<std::mutex> g(p->mu);
        p->done = true;
    }
    p->cv.notify_all();
}

// Two dispatchers take turns: whoever holds collect_mu forms the next batch (first request, the window, up to max_batch) while
// the other one's batch is being searched, so the device always has the next launch queued behind the current one -- and, the
// two lanes' launches running in separate slots of the index, overlapping it.  Batching is as with one dispatcher: batches
// are formed one at a time, from everything that arrived meanwhile.
void dispatch_loop(lantern_scan_server *s, int lane)
{
    tl_lane = lane;
    std::vector<std::shared_ptr<Pending>> batch;
    std::vector<uint8_t>  qbuf;
    std::vector<uint64_t> labels;
    std::vector<float>    dists;
    std::vector<uint32_t> counts;
    for(;;) {
        batch.clear();
        {
            std::lock_guard<std::mutex>  turn(s->collect_mu);
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [&] { return s->stop.load() || !s->queue.empty(); });
            if(s->stop && s->queue.empty()) return;
            // the first request is here: give the others `max_wait_us` to join, unless the batch is already full -- or nobody is
            // left who could join: a backend has one request outstanding at a time (amgettuple is synchronous), so once every open
            // connection has a request queued here or in the other dispatcher's batch, the window would only add latency
            // (64 backends in a closed loop: 488 -> ~290 us per scan)
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(s->max_wait_us);
            s->cv.wait_until(lk, deadline, [&] { return s->stop.load() || s->queue.size() >= s->max_batch || s->queue.size() + s->in_flight >= s->open_conns; });
            while(!s->queue.empty() && batch.size() < s->max_batch) {
                batch.push_back(s->queue.front());
                s->queue.pop_front();
            }
            s->in_flight += batch.size();
        }
        if(batch.empty()) continue;
        s->n_batches += 1;
        {
            int bin = 0;
            while(bin < 15 && ((size_t)2 << bin) <= batch.size()) ++bin;
            s->batch_hist[ bin ] += 1;
        }
        uint64_t seen = s->max_batch_seen.load();
        while(batch.size() > seen && !s->max_batch_seen.compare_exchange_weak(seen, batch.size())) {}
        // one launch per distinct (k, ef): scans of one workload share them (init_k, the ef GUC)
        std::map<std::pair<uint32_t, uint32_t>, std::vector<size_t>> groups;
        for(size_t i = 0; i < batch.size(); ++i) groups[ { batch[ i ]->k, batch[ i ]->ef } ].push_back(i);
        for(auto &kv : groups) {
            const size_t k = kv.first.first, ef = kv.first.second, nq = kv.second.size();
            qbuf.resize(nq * s->vec_bytes);
            for(size_t j = 0; j < nq; ++j) std::memcpy(&qbuf[ j * s->vec_bytes ], batch[ kv.second[ j ] ]->vec.data(), s->vec_bytes);
            labels.assign(nq * k, 0);
            dists.assign(nq * k, 0.f);
            counts.assign(nq, 0);
            const char *err = nullptr;
            const int   rc = s->fn(s->fn_ctx, qbuf.data(), nq, s->vec_bytes, k, ef, labels.data(), dists.data(), counts.data(), &err);
            s->n_launches += 1;
            for(size_t j = 0; j < nq; ++j) {
                Pending &p = *batch[ kv.second[ j ] ];
                if(rc != 0) {
                    p.error = err ? err : "lantern_scan_server: the batch search failed";
                } else {
                    const size_t c = std::min<size_t>(counts[ j ], k);
                    p.labels.assign(labels.begin() + (ptrdiff_t)(j * k), labels.begin() + (ptrdiff_t)(j * k + c));
                    p.dists.assign(dists.begin() + (ptrdiff_t)(j * k), dists.begin() + (ptrdiff_t)(j * k + c));
                }
                fulfil(batch[ kv.second[ j ] ]);
            }
        }
        {
            std::lock_guard<std::mutex> g(s->mu);
            s->in_flight -= batch.size();
        }
    }
}

bool reply_error(int fd, const std::string &msg)
{
    uint32_t head[ 3 ] = { REP_MAGIC, 1u, (uint32_t)msg.size() };
    return write_all(fd, head, sizeof(head)) && write_all(fd, msg.data(), msg.size());
}

void reader_loop(lantern_scan_server *s, lantern_scan_server::Conn *conn)
{
    const int fd = conn->fd;
    // this connection's scan: labels handed out since its last fresh request, and how many label-0 rows among them
    std::unordered_set<uint64_t> seen;
    size_t                       seen_zero = 0;
    for(;;) {
        uint32_t head[ 4 ];
        if(!read_exact(fd, head, sizeof(head))) break;  // peer closed, or the server is stopping (shutdown on the fd)
        if(head[ 0 ] != REQ_MAGIC && head[ 0 ] != CONT_MAGIC) { reply_error(fd, "lantern_scan_server: bad request magic"); break; }
        const bool     cont = head[ 0 ] == CONT_MAGIC;
        const uint32_t k = head[ 1 ], ef = head[ 2 ], nbytes = head[ 3 ];
        if(nbytes > MAX_VEC_BYTES) { reply_error(fd, "lantern_scan_server: vector too large"); break; }
        auto p = std::make_shared<Pending>();
        p->k = k;
        p->ef = ef;
        p->vec.resize(nbytes);
        if(nbytes && !read_exact(fd, p->vec.data(), nbytes)) break;
        if(nbytes != s->vec_bytes) {  // the reference's text for a wrong dimension: hnsw.c:474-476
            if(!reply_error(fd, "lantern_scan_server: query of " + std::to_string(nbytes) + " bytes, the index takes " + std::to_string(s->vec_bytes))) break;
            continue;
        }
        if(k == 0 || k > MAX_K) {
            if(!reply_error(fd, "lantern_scan_server: k out of range")) break;
            continue;
        }
        if(!cont) { seen.clear(); seen_zero = 0; }
        const size_t handed = seen.size() + seen_zero;
        if(handed + k > MAX_K) {
            if(!reply_error(fd, "lantern_scan_server: the scan has paged past the service's row limit")) break;
            continue;
        }
        p->k = (uint32_t)(handed + k);  // enough rows to find k that were not handed out yet
        s->n_requests += 1;
        {
            std::lock_guard<std::mutex> g(s->mu);
            if(s->stop) { reply_error(fd, "lantern_scan_server: stopping"); break; }
            s->queue.push_back(p);
        }
        s->cv.notify_all();
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->done; });
        }
        if(!p->error.empty()) {
            if(!reply_error(fd, p->error)) break;
            continue;
        }
        // drop what this scan already has; keep the first k of the rest
        std::vector<uint64_t> out_l;
        std::vector<float>    out_d;
        size_t                zeros_to_skip = seen_zero;
        for(size_t i = 0; i < p->labels.size() && out_l.size() < k; ++i) {
            const uint64_t l = p->labels[ i ];
            if(l == 0) {
                if(zeros_to_skip) { --zeros_to_skip; continue; }
                ++seen_zero;
            } else if(!seen.insert(l).second) {
                continue;
            }
            out_l.push_back(l);
            out_d.push_back(p->dists[ i ]);
        }
        const uint32_t c = (uint32_t)out_l.size();
        uint32_t       rep[ 3 ] = { REP_MAGIC, 0u, c };
        if(!write_all(fd, rep, sizeof(rep)) || (c && (!write_all(fd, out_l.data(), c * 8) || !write_all(fd, out_d.data(), c * 4)))) break;
    }
    {
        // the descriptor leaves the server's books BEFORE it is closed: stop() must never shut down a recycled number
        std::lock_guard<std::mutex> g(s->mu);
        conn->fd = -1;
        s->open_conns -= 1;
    }
    s->cv.notify_all();  // (a collecting dispatcher may have been waiting for this connection's next request)
    ::shutdown(fd, SHUT_RDWR);
    ::close(fd);
    conn->done = true;
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
    if(::inet_pton(AF_INET, host && *host ? host : "127.0.0.1", &a.sin_addr) != 1 || ::bind(fd, (sockaddr *)&a, sizeof(a)) != 0 ||
       ::listen(fd, 256) != 0) {
        ::close(fd);
        return -1;
    }
    socklen_t len = sizeof(a);
    ::getsockname(fd, (sockaddr *)&a, &len);
    *bound_port = ntohs(a.sin_port);
    timeval tv{ 0, 100000 };  // so the accept loop can notice `stop`
    ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    return fd;
}

void accept_loop(lantern_scan_server *s)
{
    while(!s->stop) {
        int fd = ::accept(s->listen_fd, nullptr, nullptr);
        if(fd < 0) continue;
        int one = 1;
        ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        timeval never{ 0, 0 };  // Linux hands the listener's receive timeout down to accepted sockets: a backend may idle for hours
        ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &never, sizeof(never));
        std::lock_guard<std::mutex> g(s->mu);
        if(s->stop) { ::close(fd); break; }
        // reap the readers of connections that have ended (a backend connects once per session, but sessions come and go)
        for(size_t i = 0; i < s->conns.size();) {
            if(s->conns[ i ]->done) {
                s->conns[ i ]->t.join();
                s->conns.erase(s->conns.begin() + (ptrdiff_t)i);
            } else {
                ++i;
            }
        }
        auto conn = std::make_unique<lantern_scan_server::Conn>();
        conn->fd = fd;
        lantern_scan_server::Conn *raw = conn.get();
        s->open_conns += 1;
        conn->t = std::thread(reader_loop, s, raw);
        s->conns.push_back(std::move(conn));
    }
}

lantern_scan_server *start_common(lantern_scan_server *s, const char *host, int port, size_t max_batch, unsigned max_wait_us, usearch_error_t *e)
{
    s->max_batch = max_batch ? max_batch : 256;
    s->max_wait_us = max_wait_us;
    s->listen_fd = listen_on(host, port, &s->port);
    if(s->listen_fd < 0) {
        if(e) *e = "lantern_gpu: cannot bind the scan server socket";
        delete s;
        return nullptr;
    }
    for(int l = 0; l < s->lanes; ++l) s->dispatch_thread[ l ] = std::thread(dispatch_loop, s, l);
    s->accept_thread = std::thread(accept_loop, s);
    return s;
}

}  // namespace

struct lantern_scan_client
{
    int         fd = -1;
    std::string err;
};

extern "C" {

lantern_scan_server_t *lantern_scan_server_start(usearch_index_t index, const char *host, int port, size_t max_batch, unsigned max_wait_us,
                                                 usearch_error_t *e)
{
    if(e) *e = nullptr;
    if(!index) { if(e) *e = "lantern_gpu: null index handle"; return nullptr; }
    usearch_error_t err = nullptr;
    const metadata_t m = usearch_index_metadata(index, &err);
    if(err) { if(e) *e = err; return nullptr; }
    lantern_scan_server *s = new lantern_scan_server();
    s->index = index;
    // queries arrive the way Lantern hands them to usearch_search_ef: f32 scalars, or bits for hamming (scan.c:84-88)
    const bool ham = m.init_options.metric_kind == usearch_metric_hamming_k;
    s->kind = ham ? usearch_scalar_b1_k : usearch_scalar_f32_k;
    s->vec_bytes = ham ? (m.dimensions + 7) / 8 : m.dimensions * 4;
    s->fn = default_backend;
    s->fn_ctx = s;
    s->lanes = 2;  // lantern_gpu_search_batch_lane
    return start_common(s, host, port, max_batch, max_wait_us, e);
}

lantern_scan_server_t *lantern_scan_server_start_fn(lantern_batch_search_fn fn, void *ctx, size_t vec_bytes, const char *host, int port,
                                                    size_t max_batch, unsigned max_wait_us, usearch_error_t *e)
{
    if(e) *e = nullptr;
    if(!fn || vec_bytes == 0 || vec_bytes > MAX_VEC_BYTES) { if(e) *e = "lantern_gpu: bad scan server arguments"; return nullptr; }
    lantern_scan_server *s = new lantern_scan_server();
    s->fn = fn;
    s->fn_ctx = ctx;
    s->vec_bytes = vec_bytes;
    // a caller-supplied backend is called from ONE thread unless LANTERN_SCAN_LANES=2 says it may be entered by two at a time
    if(const char *ln = std::getenv("LANTERN_SCAN_LANES")) s->lanes = std::atoi(ln) >= 2 ? 2 : 1;
    return start_common(s, host, port, max_batch, max_wait_us, e);
}

int lantern_scan_server_port(lantern_scan_server_t *s) { return s ? s->port : -1; }

void lantern_scan_server_stats(lantern_scan_server_t *s, uint64_t *requests, uint64_t *batches, uint64_t *launches, uint64_t *largest_batch)
{
    if(requests) *requests = s ? s->n_requests.load() : 0;
    if(batches) *batches = s ? s->n_batches.load() : 0;
    if(launches) *launches = s ? s->n_launches.load() : 0;
    if(largest_batch) *largest_batch = s ? s->max_batch_seen.load() : 0;
}

size_t lantern_scan_server_batch_histogram(lantern_scan_server_t *s, uint64_t *bins, size_t nbins)
{
    const size_t n = nbins < 16 ? nbins : 16;
    for(size_t i = 0; i < n; ++i) bins[ i ] = s ? s->batch_hist[ i ].load() : 0;
    return n;
}

void lantern_scan_server_stop(lantern_scan_server_t *s)
{
    if(!s) return;
    {
        std::lock_guard<std::mutex> g(s->mu);
        s->stop = true;
        for(auto &c : s->conns)
            if(c->fd >= 0) ::shutdown(c->fd, SHUT_RDWR);  // wakes readers blocked in recv
    }
    s->cv.notify_all();
    if(s->accept_thread.joinable()) s->accept_thread.join();
    for(auto &t : s->dispatch_thread)
        if(t.joinable()) t.join();
    // requests that were still queued when the dispatcher left: answer them so that their readers can finish
    std::deque<std::shared_ptr<Pending>> rest;
    {
        std::lock_guard<std::mutex> g(s->mu);
        rest.swap(s->queue);
    }
    for(auto &p : rest) {
        p->error = "lantern_scan_server: stopping";
        fulfil(p);
    }
    // the acceptor has ended, so `conns` no longer changes; readers may still be locking mu to retire their descriptor
    for(auto &c : s->conns)
        if(c->t.joinable()) c->t.join();
    if(s->listen_fd >= 0) ::close(s->listen_fd);
    delete s;
}

// ---- client side: what a backend's ldb_amgettuple calls in place of usearch_search_ef ---------------------------------

lantern_scan_client_t *lantern_scan_client_connect(const char *host, int port, usearch_error_t *e)
{
    if(e) *e = nullptr;
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in a;
    std::memset(&a, 0, sizeof(a));
    a.sin_family = AF_INET;
    a.sin_port = htons((uint16_t)port);
    if(fd < 0 || ::inet_pton(AF_INET, host && *host ? host : "127.0.0.1", &a.sin_addr) != 1 || ::connect(fd, (sockaddr *)&a, sizeof(a)) != 0) {
        if(fd >= 0) ::close(fd);
        if(e) *e = "lantern_gpu: cannot connect to the scan server";
        return nullptr;
    }
    int one = 1;
    ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    lantern_scan_client *c = new lantern_scan_client();
    c->fd = fd;
    return c;
}

static size_t client_request(lantern_scan_client_t *c, uint32_t magic, const void *query, size_t query_bytes, size_t k, size_t ef,
                             usearch_label_t *labels, float *distances, usearch_error_t *e)
{
    if(e) *e = nullptr;
    if(!c || c->fd < 0) { if(e) *e = "lantern_gpu: the scan client is not connected"; return 0; }
    if(!query || !labels || !distances || k == 0 || k > MAX_K || query_bytes > MAX_VEC_BYTES) { if(e) *e = "lantern_gpu: bad scan client arguments"; return 0; }
    auto fail = [&](const char *msg) {
        c->err = msg;
        ::close(c->fd);
        c->fd = -1;  // the stream is out of step: this connection is done
        if(e) *e = c->err.c_str();
        return (size_t)0;
    };
    uint32_t head[ 4 ] = { magic, (uint32_t)k, (uint32_t)ef, (uint32_t)query_bytes };
    if(!write_all(c->fd, head, sizeof(head)) || !write_all(c->fd, query, query_bytes)) return fail("lantern_gpu: the scan server went away");
    uint32_t rep[ 3 ];
    if(!read_exact(c->fd, rep, sizeof(rep)) || rep[ 0 ] != REP_MAGIC) return fail("lantern_gpu: the scan server went away");
    if(rep[ 1 ] != 0) {  // an error frame: the connection stays usable
        std::string msg(rep[ 2 ] <= 4096 ? rep[ 2 ] : 0, '\0');
        if(rep[ 2 ] > 4096 || (rep[ 2 ] && !read_exact(c->fd, &msg[ 0 ], rep[ 2 ]))) return fail("lantern_gpu: the scan server went away");
        c->err = msg;
        if(e) *e = c->err.c_str();
        return 0;
    }
    const uint32_t count = rep[ 2 ];
    if(count > k) return fail("lantern_gpu: the scan server answered with more rows than asked for");
    if(count && (!read_exact(c->fd, labels, (size_t)count * 8) || !read_exact(c->fd, distances, (size_t)count * 4)))
        return fail("lantern_gpu: the scan server went away");
    return count;
}

size_t lantern_scan_client_search(lantern_scan_client_t *c, const void *query, size_t query_bytes, size_t k, size_t ef, usearch_label_t *labels,
                                  float *distances, usearch_error_t *e)
{
    return client_request(c, REQ_MAGIC, query, query_bytes, k, ef, labels, distances, e);
}

// the next k rows of the scan this connection started with lantern_scan_client_search (same query)
size_t lantern_scan_client_search_next(lantern_scan_client_t *c, const void *query, size_t query_bytes, size_t k, size_t ef,
                                       usearch_label_t *labels, float *distances, usearch_error_t *e)
{
    return client_request(c, CONT_MAGIC, query, query_bytes, k, ef, labels, distances, e);
}

void lantern_scan_client_close(lantern_scan_client_t *c)
{
    if(!c) return;
    if(c->fd >= 0) {
        ::shutdown(c->fd, SHUT_RDWR);
        ::close(c->fd);
    }
    delete c;
}

}  // extern "C"
