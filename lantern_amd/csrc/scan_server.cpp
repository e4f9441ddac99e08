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
// Threads: an acceptor, a few I/O threads and the dispatchers (two on a device index, one in front of a caller-supplied back
// end unless it says it may be entered twice) -- no thread per connection.  Every connection belongs to one I/O thread's epoll
// set, armed one-shot: it is disarmed from the moment its request has been read until its answer has been written (a backend has
// one request outstanding; whatever it sends meanwhile waits in the socket).  An I/O thread reads the requests of its
// connections that are ready and queues them; the dispatcher whose turn it is closes the batch when it is full, when the window
// (`max_wait_us` after the first request) ends, or when nobody is left who could join -- every open connection already has a
// request queued or in a batch being searched.  It hands the turn over, groups the batch by (k, ef), calls the batch search
// once per group, gives each answer back to the I/O thread of its connection (one wake-up per I/O thread and batch), which
// filters it against the scan's history, writes it and re-arms the connection.  A request costs three system calls spread
// over the I/O threads and no context switch of its own: wake-ups are shared by whatever is ready.
// (r2-r3 had a blocking reader thread per connection: 512 wake-ups per round of 256 backends on the server alone, 316-396 k
// queries/s on 16 cores.  One thread doing all socket work -- tried on the way -- is worse: a send on a loopback socket wakes
// its receiver, ~10 us per request in series.)
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/epoll.h>
#include <sys/eventfd.h>
#include <sys/socket.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cerrno>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/lantern_gpu.h"
#include "abi_guard.hpp"

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
bool write_all(int fd, const void *buf, size_t n)  // the CLIENT's side: a blocking send of its one outstanding request
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
// The SERVER's side never blocks on a backend.  Answers are written by a dispatcher (batches of up to eight) or by one of a few
// shared I/O threads: a backend that sends requests but stops reading its answers (stuck, SIGSTOPped, pipelining without
// reading) would otherwise park that thread in send() once its socket buffers are full -- and with it every connection the
// thread serves, or, in the dispatcher's case, the whole service.  An answer is at most a few hundred bytes and a backend has
// one request outstanding, so a healthy connection always has room: the send is non-blocking, a short one waits for POLLOUT
// for at most kSendGraceMs in total, and a connection that still cannot take its answer is reported as gone (it is closed by
// its I/O thread, the backend sees a reset instead of a silent stall).
constexpr int kSendGraceMs = 20;
bool write_answer(int fd, const void *buf, size_t n)
{
    const char *p = (const char *)buf;
    int         grace = kSendGraceMs;
    while(n) {
        ssize_t r = ::send(fd, p, n, MSG_NOSIGNAL | MSG_DONTWAIT);
        if(r > 0) {
            p += r;
            n -= (size_t)r;
            continue;
        }
        if(r < 0 && errno == EINTR) continue;
        if(r < 0 && (errno == EAGAIN || errno == EWOULDBLOCK) && grace > 0) {
            pollfd    pf{ fd, POLLOUT, 0 };
            const int step = std::min(grace, 5);
            (void)::poll(&pf, 1, step);
            grace -= step;
            continue;
        }
        return false;
    }
    return true;
}

// one connection = one backend: its read buffer, the request it has outstanding, and the scan it is paging through
struct Conn
{
    int                  fd = -1, io = 0;  // its socket; the I/O thread it belongs to
    std::vector<uint8_t> in;  // bytes read so far of the next request
    // the outstanding request
    bool                 cont = false;
    uint32_t             want = 0, k = 0, ef = 0;  // rows the client asked for; rows the search is asked for (handed out + want)
    std::vector<uint8_t> vec;
    uint64_t             t_read = 0, t_closed = 0;  // ns: its request was read / its batch closed (lantern_scan_server_timing)
    // this connection's scan: labels handed out since its last fresh request, and how many label-0 rows among them
    std::unordered_set<uint64_t> seen;
    size_t                       seen_zero = 0;
};

// a request's way back to its connection
struct Done
{
    Conn                 *c = nullptr;
    bool                  gone = false;  // the dispatcher answered it itself and the write failed: only the connection's end is left to do
    uint64_t              t_known = 0;  // ns: when the dispatcher learnt the answer
    std::string           error;  // non-empty: an error frame
    std::vector<uint64_t> labels;
    std::vector<float>    dists;
};

struct IoThread
{
    int                  epfd = -1, evfd = -1;  // its connections, one-shot; completions / new connections / stop
    std::thread          t;
    std::mutex           mu;  // the two lists below
    std::vector<Done>    done;
    std::vector<std::unique_ptr<Conn>> incoming;
    std::unordered_map<int, std::unique_ptr<Conn>> conns;  // touched by the thread itself only
};

}  // namespace

constexpr int kMaxLanes = 8;  // = Index::kLanes of the device library (lantern_gpu_search_batch_lane)

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
    std::thread             accept_thread, dispatch_thread[ kMaxLanes ];
    int                     lanes = 1;   // dispatchers: one collects the next batch while the other's batch is on the device
    bool                    window = true;   // notify mode: LANTERN_SCAN_WINDOW=0 drops the batching window (measured slower)
    bool                    notify = false;  // the device index: answers go back one by one as their walks end (lantern_gpu_search_batch_lane_notify)
    std::vector<std::unique_ptr<IoThread>> io;
    std::mutex              collect_mu;  // held by the dispatcher that is collecting (one batch is formed at a time)
    std::mutex              mu;          // the queue and the two counts below
    std::condition_variable cv;
    std::deque<Conn *>      queue;       // connections whose request has been read (disarmed until answered)
    size_t                  open_conns = 0, in_flight = 0;  // connections being served; requests in a closed batch, not answered yet
    std::atomic<uint64_t>   n_requests{ 0 }, n_batches{ 0 }, n_launches{ 0 }, max_batch_seen{ 0 };
    std::atomic<uint64_t>   batch_hist[ 16 ] = {};  // batches by size: bin b counts sizes in [2^b, 2^(b+1))
    // where a request's time on the server goes, summed in ns over all answered requests: read -> its batch closes (the window and
    // the wait for a free dispatcher) | batch closed -> its answer is known (padding, copy, launch, ITS walk -- or, without notify,
    // the batch's longest) | answer known -> written to the socket
    std::atomic<uint64_t>   t_wait_ns{ 0 }, t_search_ns{ 0 }, t_reply_ns{ 0 }, t_count{ 0 };
};

namespace {

inline uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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

bool reply_error(int fd, const std::string &msg)
{
    uint32_t head[ 3 ] = { REP_MAGIC, 1u, (uint32_t)msg.size() };
    return write_answer(fd, head, sizeof(head)) && write_answer(fd, msg.data(), msg.size());
}

bool arm(IoThread *t, Conn *c, int op)
{
    epoll_event ev;
    std::memset(&ev, 0, sizeof(ev));
    ev.events = EPOLLIN | EPOLLRDHUP | EPOLLONESHOT;
    ev.data.ptr = c;
    return ::epoll_ctl(t->epfd, op, c->fd, &ev) == 0;
}

// the connection is over: out of its epoll set and table, descriptor closed (by its I/O thread, its only holder while armed
// or while its answer is being written)
void drop(lantern_scan_server *s, IoThread *t, Conn *c)
{
    const int fd = c->fd;
    ::epoll_ctl(t->epfd, EPOLL_CTL_DEL, fd, nullptr);
    t->conns.erase(fd);  // (frees c)
    {
        std::lock_guard<std::mutex> g(s->mu);
        s->open_conns -= 1;
    }
    s->cv.notify_all();  // (a collecting dispatcher may have been waiting for this connection's next request)
    ::shutdown(fd, SHUT_RDWR);
    ::close(fd);
}

// Reads what the socket holds of the connection's next request -- never past its end: a valid request is 16 + vec_bytes long, and
// a connection's following request stays in the socket until the connection is armed again.  1 = a complete, valid request is
// in c (vec, k, ef, want); 0 = not yet, or an error frame went out and the connection waits for its next request: re-arm;
// -1 = the connection is over.
int read_request(lantern_scan_server *s, Conn *c)
{
    for(;;) {
        size_t goal = 16 + s->vec_bytes;
        if(c->in.size() >= 16) {
            uint32_t nbytes;
            std::memcpy(&nbytes, c->in.data() + 12, 4);
            if(nbytes > MAX_VEC_BYTES) { reply_error(c->fd, "lantern_scan_server: vector too large"); return -1; }
            goal = 16 + (size_t)nbytes;
            if(c->in.size() >= goal) break;
        }
        uint8_t       tmp[ 16384 ];
        const ssize_t r = ::recv(c->fd, tmp, std::min(sizeof(tmp), goal - c->in.size()), MSG_DONTWAIT);
        if(r > 0) c->in.insert(c->in.end(), tmp, tmp + r);
        else if(r == 0) return -1;  // peer closed
        else if(errno == EAGAIN || errno == EWOULDBLOCK) return 0;
        else if(errno != EINTR) return -1;
    }
    uint32_t head[ 4 ];
    std::memcpy(head, c->in.data(), 16);
    const uint32_t k = head[ 1 ], ef = head[ 2 ], nbytes = head[ 3 ];
    if(head[ 0 ] != REQ_MAGIC && head[ 0 ] != CONT_MAGIC) { reply_error(c->fd, "lantern_scan_server: bad request magic"); return -1; }
    c->cont = head[ 0 ] == CONT_MAGIC;
    c->vec.assign(c->in.begin() + 16, c->in.begin() + 16 + (ptrdiff_t)nbytes);
    c->in.erase(c->in.begin(), c->in.begin() + 16 + (ptrdiff_t)nbytes);
    if(nbytes != s->vec_bytes)  // the reference's text for a wrong dimension: hnsw.c:474-476
        return reply_error(c->fd, "lantern_scan_server: query of " + std::to_string(nbytes) + " bytes, the index takes " + std::to_string(s->vec_bytes)) ? 0 : -1;
    if(k == 0 || k > MAX_K) return reply_error(c->fd, "lantern_scan_server: k out of range") ? 0 : -1;
    if(!c->cont) { c->seen.clear(); c->seen_zero = 0; }
    const size_t handed = c->seen.size() + c->seen_zero;
    if(handed + k > MAX_K) return reply_error(c->fd, "lantern_scan_server: the scan has paged past the service's row limit") ? 0 : -1;
    c->want = k;
    c->k = (uint32_t)(handed + k);  // enough rows to find k that were not handed out yet
    c->ef = ef;
    return 1;
}

// the answer of one request: drop what this scan already has, keep the first `want` of the rest
bool reply_rows(Conn *c, const uint64_t *labels, const float *dists, size_t count)
{
    std::vector<uint8_t>  out(12 + (size_t)c->want * 12);
    std::vector<uint64_t> ls;
    std::vector<float>    ds;
    size_t                zeros_to_skip = c->seen_zero;
    for(size_t i = 0; i < count && ls.size() < c->want; ++i) {
        const uint64_t l = labels[ i ];
        if(l == 0) {
            if(zeros_to_skip) { --zeros_to_skip; continue; }
            ++c->seen_zero;
        } else if(!c->seen.insert(l).second) {
            continue;
        }
        ls.push_back(l);
        ds.push_back(dists[ i ]);
    }
    const uint32_t n = (uint32_t)ls.size();
    const uint32_t rep[ 3 ] = { REP_MAGIC, 0u, n };
    std::memcpy(out.data(), rep, 12);
    if(n) {
        std::memcpy(out.data() + 12, ls.data(), (size_t)n * 8);
        std::memcpy(out.data() + 12 + (size_t)n * 8, ds.data(), (size_t)n * 4);
    }
    return write_answer(c->fd, out.data(), 12 + (size_t)n * 12);  // one send per answer
}

void io_loop(lantern_scan_server *s, IoThread *t)
{
    epoll_event       events[ 256 ];
    std::vector<Done> done;
    std::vector<std::unique_ptr<Conn>> fresh;
    for(;;) {
        const int n = ::epoll_wait(t->epfd, events, 256, -1);
        if(n < 0 && errno != EINTR) return;
        size_t queued = 0;
        for(int i = 0; i < n; ++i) {
            Conn *c = (Conn *)events[ i ].data.ptr;
            if(c) {  // a connection with something to read
                const int got = read_request(s, c);
                if(got < 0) drop(s, t, c);
                else if(got == 0) { if(!arm(t, c, EPOLL_CTL_MOD)) drop(s, t, c); }
                else {
                    s->n_requests += 1;
                    c->t_read = now_ns();
                    std::lock_guard<std::mutex> g(s->mu);
                    s->queue.push_back(c);
                    ++queued;
                }
                continue;
            }
            // the event descriptor: answers to write, connections to adopt, or the end
            uint64_t v;
            (void)!::read(t->evfd, &v, 8);
            if(s->stop) return;
            {
                std::lock_guard<std::mutex> g(t->mu);
                done.swap(t->done);
                fresh.swap(t->incoming);
            }
            for(auto &up : fresh) {
                Conn *nc = up.get();
                t->conns[ nc->fd ] = std::move(up);
                if(!arm(t, nc, EPOLL_CTL_ADD)) drop(s, t, nc);
            }
            fresh.clear();
            for(Done &d : done) {
                const bool sent = !d.gone && (d.error.empty() ? reply_rows(d.c, d.labels.data(), d.dists.data(), d.labels.size()) : reply_error(d.c->fd, d.error));
                if(d.t_known) s->t_reply_ns += now_ns() - d.t_known;
                if(!sent || !arm(t, d.c, EPOLL_CTL_MOD)) drop(s, t, d.c);  // answered: the connection may speak again
            }
            done.clear();
        }
        if(queued) s->cv.notify_all();  // one wake-up for everything this round read
    }
}

// Two dispatchers take turns: whoever holds collect_mu forms the next batch while the other one's batch is being searched, so
// the device always has the next launch queued behind the current one -- and, the two lanes' launches running in separate
// slots of the index, overlapping it.
void dispatch_loop(lantern_scan_server *s, int lane)
{
    tl_lane = lane;
    std::vector<Conn *>   batch;
    std::vector<uint8_t>  qbuf;
    std::vector<uint64_t> labels;
    std::vector<float>    dists;
    std::vector<uint32_t> counts;
    std::vector<char>     touched;
    for(;;) {
        batch.clear();
        {
            std::lock_guard<std::mutex>  turn(s->collect_mu);
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [&] { return s->stop.load() || !s->queue.empty(); });
            if(s->stop) return;
            // the first request is here: give the others `max_wait_us` to join, unless the batch is already full -- or nobody is
            // left who could join: a backend has one request outstanding at a time (amgettuple is synchronous), so once every open
            // connection has a request queued here or in the other dispatcher's batch, the window would only add latency
            // With two dispatchers a batch also closes at its SHARE of the backends (open connections / dispatchers): the two halves
            // of a population that moves in step then take turns -- one half on the device while the other half's answers and next
            // requests are on the wire -- instead of everybody waiting for the device and then the device waiting for everybody.
            const auto   deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(s->max_wait_us);
            const size_t lanes = (size_t)s->lanes;
            auto         share = [&] { return std::min(s->max_batch, std::max<size_t>(1, (s->open_conns + lanes - 1) / lanes)); };
            // (Answers that go back one by one -- notify -- still gain from the window: measured in round 5 at 1M x 768, a dispatcher
            // that takes whatever is queued the moment it is free forms batches of 44 instead of 62 at 256 backends and of 75 instead of
            // 150 - 180 at 1024, and serves 452 k against 501 k and 583 k against 715 k scans/s: every launch costs the host a padding
            // pass, a copy and a launch under the runtime's lock.  LANTERN_SCAN_WINDOW=0 selects that policy.)
            // ... and with FEW backends the window only costs: at 16 backends a dispatcher that takes what is there serves 55 k against
            // 47 k scans/s (p50 276 against 280 us, the answer 233 us after its batch closed against 306); at 64 the two policies are
            // even (201 - 204 k against 208 k).  So: no window below eight backends per lane.
            const bool no_window = s->notify && (!s->window || s->open_conns < 8 * lanes);
            if(!no_window)
                s->cv.wait_until(lk, deadline, [&] { return s->stop.load() || s->queue.size() >= share() || s->queue.size() + s->in_flight >= s->open_conns; });
            if(s->stop) return;
            const size_t take = no_window ? s->max_batch : share();
            while(!s->queue.empty() && batch.size() < take) {
                batch.push_back(s->queue.front());
                s->queue.pop_front();
            }
            s->in_flight += batch.size();
        }
        if(batch.empty()) continue;
        {
            const uint64_t t = now_ns();
            for(Conn *c : batch) c->t_closed = t;
        }
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
        touched.assign(s->io.size(), 0);
        // A handful of answers the dispatcher writes itself (a disarmed connection has one holder at a time, and this is it):
        // one thread hand-off less on the path of a lone backend.  Larger batches go back to the I/O threads, whose sends
        // run side by side.
        const bool direct = batch.size() <= 8;
        for(auto &kv : groups) {
            const size_t k = kv.first.first, ef = kv.first.second, nq = kv.second.size();
            qbuf.resize(nq * s->vec_bytes);
            for(size_t j = 0; j < nq; ++j) std::memcpy(&qbuf[ j * s->vec_bytes ], batch[ kv.second[ j ] ]->vec.data(), s->vec_bytes);
            labels.assign(nq * k, 0);
            dists.assign(nq * k, 0.f);
            counts.assign(nq, 0);
            // answers `which[0 .. count)` of this group back to their connections (rc != 0: the error frame `msg`), then one wake-up per
            // I/O thread that got any, and the requests leave the in-flight count
            std::string       msg;
            int               rc = 0;
            std::vector<char> delivered(nq, 0);
            auto deliver = [&](const uint32_t *which, size_t count) {
                const uint64_t t_known = now_ns();
                for(size_t w = 0; w < count; ++w) {
                    const size_t j = which[ w ];
                    delivered[ j ] = 1;
                    Done         d;
                    d.c = batch[ kv.second[ j ] ];
                    d.t_known = t_known;
                    s->t_wait_ns += d.c->t_closed - d.c->t_read;
                    s->t_search_ns += t_known - d.c->t_closed;
                    s->t_count += 1;
                    if(direct) {
                        const bool sent = rc != 0 ? reply_error(d.c->fd, msg)
                                                  : reply_rows(d.c, &labels[ j * k ], &dists[ j * k ], std::min<size_t>(counts[ j ], k));
                        s->t_reply_ns += now_ns() - t_known;
                        if(sent && arm(s->io[ (size_t)d.c->io ].get(), d.c, EPOLL_CTL_MOD)) continue;
                        d.gone = true;  // its I/O thread takes it down
                    }
                    if(d.gone) {
                    } else if(rc != 0) {
                        d.error = msg;
                    } else {
                        const size_t cn = std::min<size_t>(counts[ j ], k);
                        d.labels.assign(labels.begin() + (ptrdiff_t)(j * k), labels.begin() + (ptrdiff_t)(j * k + cn));
                        d.dists.assign(dists.begin() + (ptrdiff_t)(j * k), dists.begin() + (ptrdiff_t)(j * k + cn));
                    }
                    IoThread *t = s->io[ (size_t)d.c->io ].get();
                    touched[ (size_t)d.c->io ] = 1;
                    std::lock_guard<std::mutex> g(t->mu);
                    t->done.push_back(std::move(d));
                }
                const uint64_t one = 1;
                for(size_t i = 0; i < s->io.size(); ++i)
                    if(touched[ i ]) {
                        (void)!::write(s->io[ i ]->evfd, &one, 8);
                        touched[ i ] = 0;
                    }
                std::lock_guard<std::mutex> g(s->mu);
                s->in_flight -= count;
            };
            const char *err = nullptr;
            if(s->notify) {
                // the device index: every answer goes back when ITS walk ends, not when the batch's longest one does
                // (lantern_gpu_search_batch_lane_notify; the callback runs on this thread)
                struct Ctx { decltype(deliver) *fn; } cx{ &deliver };
                usearch_error_t ue = nullptr;
                lantern_gpu_search_batch_lane_notify(s->index, tl_lane, qbuf.data(), nq, s->kind, k, ef, labels.data(), dists.data(), counts.data(),
                                                     [](void *c, const uint32_t *which, size_t count) {
                                                         Ctx *x = (Ctx *)c;
                                                         (*x->fn)(which, count);
                                                     },
                                                     &cx, &ue);
                if(ue) { rc = 1; err = ue; }
            } else {
                rc = s->fn(s->fn_ctx, qbuf.data(), nq, s->vec_bytes, k, ef, labels.data(), dists.data(), counts.data(), &err);
            }
            s->n_launches += 1;
            if(rc != 0) msg = std::string(err ? err : "lantern_scan_server: the batch search failed");
            // everything (the plain back end) -- or what a failed launch never handed on: those get the error frame.  (A connection
            // whose answer went out has been re-armed and may already carry its next request: it is never answered twice.)
            std::vector<uint32_t> rest;
            for(size_t j = 0; j < nq; ++j)
                if(!delivered[ j ]) rest.push_back((uint32_t)j);
            if(!rest.empty()) deliver(rest.data(), rest.size());
        }
    }
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
       ::listen(fd, 1024) != 0) {
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
    size_t next = 0;
    while(!s->stop) {
        int fd = ::accept(s->listen_fd, nullptr, nullptr);
        if(fd < 0) continue;
        int one = 1;
        ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        timeval never{ 0, 0 };  // Linux hands the listener's receive timeout down to accepted sockets: a backend may idle for hours
        ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &never, sizeof(never));
        auto conn = std::make_unique<Conn>();
        conn->fd = fd;
        conn->io = (int)(next++ % s->io.size());
        IoThread *t = s->io[ (size_t)conn->io ].get();
        {
            std::lock_guard<std::mutex> g(s->mu);
            if(s->stop) { ::close(fd); break; }
            s->open_conns += 1;
        }
        {
            std::lock_guard<std::mutex> g(t->mu);
            t->incoming.push_back(std::move(conn));
        }
        const uint64_t one64 = 1;
        (void)!::write(t->evfd, &one64, 8);
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
    // socket work is spread over a few threads (a send on a loopback socket wakes its receiver: ~10 us apiece in series)
    size_t nio = std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 2));
    if(const char *ev = std::getenv("LANTERN_SCAN_IO_THREADS")) nio = (size_t)std::max(1, std::min(64, std::atoi(ev)));
    bool ok = true;
    for(size_t i = 0; i < nio && ok; ++i) {
        auto t = std::make_unique<IoThread>();
        t->epfd = ::epoll_create1(EPOLL_CLOEXEC);
        t->evfd = ::eventfd(0, EFD_CLOEXEC | EFD_NONBLOCK);
        epoll_event wev;
        std::memset(&wev, 0, sizeof(wev));
        wev.events = EPOLLIN;
        wev.data.ptr = nullptr;
        ok = t->epfd >= 0 && t->evfd >= 0 && ::epoll_ctl(t->epfd, EPOLL_CTL_ADD, t->evfd, &wev) == 0;
        s->io.push_back(std::move(t));
    }
    if(!ok) {
        if(e) *e = "lantern_gpu: cannot set up the scan server's epoll sets";
        for(auto &t : s->io) {
            if(t->epfd >= 0) ::close(t->epfd);
            if(t->evfd >= 0) ::close(t->evfd);
        }
        ::close(s->listen_fd);
        delete s;
        return nullptr;
    }
    for(auto &t : s->io) t->t = std::thread(io_loop, s, t.get());
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
try {
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
    // dispatchers = lanes of lantern_gpu_search_batch_lane (up to four batches in flight on the device, each in its own slab of
    // visited bitmaps).  Default four; LANTERN_SCAN_LANES = 1 .. 8.  Measured with lantern-scan-load on 100k x 128 (round 4,
    // profiles/r04_scan_load_lanes.jsonl; lanes 1 / 2 / 3 / 4): 16 backends p50 201 / 214 / 193 / 173 us, 64: 301 / 243 / 235 / 227 us,
    // 256: 389 k / 560 k / 563 k / 628 k scans/s; 8 backends 157 us whatever the number.
    s->lanes = 4;
    if(const char *ln = std::getenv("LANTERN_SCAN_LANES")) s->lanes = std::min(kMaxLanes, std::max(1, std::atoi(ln)));
    // answers one by one as their walks end (LANTERN_SCAN_NOTIFY=0: the whole batch's answers when its launch ends, as before round 5)
    s->notify = !(std::getenv("LANTERN_SCAN_NOTIFY") && std::atoi(std::getenv("LANTERN_SCAN_NOTIFY")) == 0);
    s->window = !(std::getenv("LANTERN_SCAN_WINDOW") && std::atoi(std::getenv("LANTERN_SCAN_WINDOW")) == 0);
    // every lane launches on a stream of its own; the HIP runtime gives a process GPU_MAX_HW_QUEUES hardware queues (4 unless the
    // process was started with more) and streams that share one run their kernels one after the other.  The setting is read when
    // the runtime initialises, i.e. it belongs to whoever starts the process (lantern-scan-server's main() sets 16): say so, once,
    // rather than lose a fifth of the service's throughput silently.
    {
        const char *hq = std::getenv("GPU_MAX_HW_QUEUES");
        const int   queues = hq ? std::atoi(hq) : 4;
        static bool warned = false;
        if(s->lanes + 1 > queues && !warned) {
            warned = true;
            std::fprintf(stderr,
                         "lantern_gpu: the scan service runs %d lanes but GPU_MAX_HW_QUEUES=%s gives the process %d hardware queues: lanes that share a queue "
                         "run one behind the other.  Start the process with GPU_MAX_HW_QUEUES=16 in its environment (INTEGRATION.md section 7).\n",
                         s->lanes, hq ? hq : "(unset)", queues);
        }
    }
    return start_common(s, host, port, max_batch, max_wait_us, e);
}
LANTERN_ABI_CATCH(e)

lantern_scan_server_t *lantern_scan_server_start_fn(lantern_batch_search_fn fn, void *ctx, size_t vec_bytes, const char *host, int port,
                                                    size_t max_batch, unsigned max_wait_us, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    if(!fn || vec_bytes == 0 || vec_bytes > MAX_VEC_BYTES) { if(e) *e = "lantern_gpu: bad scan server arguments"; return nullptr; }
    lantern_scan_server *s = new lantern_scan_server();
    s->fn = fn;
    s->fn_ctx = ctx;
    s->vec_bytes = vec_bytes;
    // a caller-supplied backend is called from ONE thread unless LANTERN_SCAN_LANES = 2 .. 8 says it may be entered by that many at a time
    if(const char *ln = std::getenv("LANTERN_SCAN_LANES")) s->lanes = std::min(kMaxLanes, std::max(1, std::atoi(ln)));
    return start_common(s, host, port, max_batch, max_wait_us, e);
}
LANTERN_ABI_CATCH(e)

int lantern_scan_server_port(lantern_scan_server_t *s) { return s ? s->port : -1; }

void lantern_scan_server_stats(lantern_scan_server_t *s, uint64_t *requests, uint64_t *batches, uint64_t *launches, uint64_t *largest_batch)
try {
    if(requests) *requests = s ? s->n_requests.load() : 0;
    if(batches) *batches = s ? s->n_batches.load() : 0;
    if(launches) *launches = s ? s->n_launches.load() : 0;
    if(largest_batch) *largest_batch = s ? s->max_batch_seen.load() : 0;
}
LANTERN_ABI_CATCH_VOID(nullptr)

// mean microseconds a request spends on the server, by leg: read -> batch closed | batch closed -> answer known | answer known ->
// written; out[3] = requests the means are over.  Counters are cumulative since start.
void lantern_scan_server_timing(lantern_scan_server_t *s, double *out4)
{
    if(!s || !out4) return;
    const double n = (double)std::max<uint64_t>(s->t_count.load(), 1);
    out4[ 0 ] = (double)s->t_wait_ns.load() / n / 1e3;
    out4[ 1 ] = (double)s->t_search_ns.load() / n / 1e3;
    out4[ 2 ] = (double)s->t_reply_ns.load() / n / 1e3;
    out4[ 3 ] = (double)s->t_count.load();
}

size_t lantern_scan_server_batch_histogram(lantern_scan_server_t *s, uint64_t *bins, size_t nbins)
try {
    const size_t n = nbins < 16 ? nbins : 16;
    for(size_t i = 0; i < n; ++i) bins[ i ] = s ? s->batch_hist[ i ].load() : 0;
    return n;
}
LANTERN_ABI_CATCH(nullptr)

void lantern_scan_server_stop(lantern_scan_server_t *s)
try {
    if(!s) return;
    {
        std::lock_guard<std::mutex> g(s->mu);
        s->stop = true;
    }
    s->cv.notify_all();
    const uint64_t one = 1;
    for(auto &t : s->io) (void)!::write(t->evfd, &one, 8);
    if(s->accept_thread.joinable()) s->accept_thread.join();
    for(auto &t : s->dispatch_thread)
        if(t.joinable()) t.join();
    for(auto &t : s->io)
        if(t->t.joinable()) t->t.join();
    // nobody else is left.  Requests that were read but never searched, answers that were never written: an error frame each;
    // then everything still connected is closed (clients see "went away")
    for(Conn *c : s->queue) reply_error(c->fd, "lantern_scan_server: stopping");
    for(auto &t : s->io) {
        for(Done &d : t->done) reply_error(d.c->fd, "lantern_scan_server: stopping");
        for(auto &kv : t->conns) {
            ::shutdown(kv.first, SHUT_RDWR);
            ::close(kv.first);
        }
        for(auto &c : t->incoming) ::close(c->fd);
        ::close(t->epfd);
        ::close(t->evfd);
    }
    if(s->listen_fd >= 0) ::close(s->listen_fd);
    delete s;
}
LANTERN_ABI_CATCH_VOID(nullptr)

// ---- client side: what a backend's ldb_amgettuple calls in place of usearch_search_ef ---------------------------------

lantern_scan_client_t *lantern_scan_client_connect(const char *host, int port, usearch_error_t *e)
try {
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
LANTERN_ABI_CATCH(e)

static size_t client_request(lantern_scan_client_t *c, uint32_t magic, const void *query, size_t query_bytes, size_t k, size_t ef,
                             usearch_label_t *labels, float *distances, usearch_error_t *e)
try {
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
LANTERN_ABI_CATCH(e)

size_t lantern_scan_client_search(lantern_scan_client_t *c, const void *query, size_t query_bytes, size_t k, size_t ef, usearch_label_t *labels,
                                  float *distances, usearch_error_t *e)
try {
    return client_request(c, REQ_MAGIC, query, query_bytes, k, ef, labels, distances, e);
}
LANTERN_ABI_CATCH(e)

// the next k rows of the scan this connection started with lantern_scan_client_search (same query)
size_t lantern_scan_client_search_next(lantern_scan_client_t *c, const void *query, size_t query_bytes, size_t k, size_t ef,
                                       usearch_label_t *labels, float *distances, usearch_error_t *e)
try {
    return client_request(c, CONT_MAGIC, query, query_bytes, k, ef, labels, distances, e);
}
LANTERN_ABI_CATCH(e)

void lantern_scan_client_close(lantern_scan_client_t *c)
try {
    if(!c) return;
    if(c->fd >= 0) {
        ::shutdown(c->fd, SHUT_RDWR);
        ::close(c->fd);
    }
    delete c;
}
LANTERN_ABI_CATCH_VOID(nullptr)

}  // extern "C"
