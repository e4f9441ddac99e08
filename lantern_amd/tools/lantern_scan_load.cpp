// lantern-scan-load -- closed-loop load generator for the scan-side service (lantern_amd/csrc/scan_server.cpp): what N
// PostgreSQL backends running ORDER BY ... LIMIT k scans against ONE HBM-resident index look like to the device.
//
// Builds a synthetic index (i.i.d. N(0,1) rows; --rows x --dim), starts the service on a loopback port, opens --connections
// client connections -- one thread each, the way a backend holds one -- and lets every one issue single k-NN queries back to
// back for --seconds (lantern_scan_client_search: what ldb_amgettuple calls in place of usearch_search_ef,
// lantern_hnsw/src/hnsw/scan.c:220-228).  Prints one JSON line: queries/s, latency percentiles, the service's batch-size
// histogram.  Everything goes through the C ABI of liblantern_gpu.so; the clients speak TCP to the server like separate processes.
// --client-threads T (T > 0) drives the same number of connections from T threads instead, each multiplexing its share with
// epoll and speaking the wire protocol of scan_server.cpp itself: the generator's own scheduling cost (two context switches per
// query and connection thread, on the cores the server shares with it) leaves the picture, and what remains is the service.
// --port P [--host H] drives a service that is ALREADY running (another process's index: bench.py's headline index) instead of
// building one here; the service's own statistics are then the other process's to report.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/epoll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lantern_gpu.h"

using Clock = std::chrono::steady_clock;

int main(int argc, char **argv)
{
    size_t   rows = 100000, dim = 128, m = 16, efc = 128, ef = 64, k = 10, connections = 256, max_batch = 1024, pool = 8192, client_threads = 0;
    unsigned wait_us = 200;
    double   seconds = 5.0, warm = 1.0;
    int      ext_port = 0;
    std::string ext_host = "127.0.0.1";
    for(int i = 1; i < argc; ++i) {
        auto val = [&](const char *name) -> const char * { return std::strcmp(argv[ i ], name) == 0 && i + 1 < argc ? argv[ ++i ] : nullptr; };
        if(const char *v = val("--rows")) rows = (size_t)std::atoll(v);
        else if(const char *v = val("--dim")) dim = (size_t)std::atoll(v);
        else if(const char *v = val("--m")) m = (size_t)std::atoll(v);
        else if(const char *v = val("--ef-construction")) efc = (size_t)std::atoll(v);
        else if(const char *v = val("--ef")) ef = (size_t)std::atoll(v);
        else if(const char *v = val("--k")) k = (size_t)std::atoll(v);
        else if(const char *v = val("--connections")) connections = (size_t)std::atoll(v);
        else if(const char *v = val("--client-threads")) client_threads = (size_t)std::atoll(v);
        else if(const char *v = val("--max-batch")) max_batch = (size_t)std::atoll(v);
        else if(const char *v = val("--max-wait-us")) wait_us = (unsigned)std::atoi(v);
        else if(const char *v = val("--seconds")) seconds = std::atof(v);
        else if(const char *v = val("--warmup-seconds")) warm = std::atof(v);
        else if(const char *v = val("--port")) ext_port = std::atoi(v);
        else if(const char *v = val("--host")) ext_host = v;
        else {
            std::fprintf(stderr, "usage: %s [--rows N --dim D --m M --ef-construction E --ef E --k K] [--connections C] [--max-batch B] "
                                 "[--max-wait-us U] [--seconds S] [--warmup-seconds W] [--client-threads T] [--port P [--host H]]\n", argv[ 0 ]);
            return 2;
        }
    }
    usearch_error_t err = nullptr;
    usearch_init_options_t o;
    std::memset(&o, 0, sizeof(o));
    o.metric_kind = usearch_metric_l2sq_k;
    o.quantization = usearch_scalar_f32_k;
    o.dimensions = dim;
    o.connectivity = m;
    o.expansion_add = efc;
    o.expansion_search = ef;
    const bool external = ext_port > 0;
    usearch_index_t ix = external ? nullptr : usearch_init(&o, nullptr, &err);
    if(err) { std::fprintf(stderr, "%s\n", err); return 1; }
    std::vector<float> base(external ? 0 : rows * dim), queries(pool * dim);
    {
        std::mt19937_64                 rng(1);
        std::normal_distribution<float> nd(0.f, 1.f);
        for(float &x : base) x = nd(rng);
        std::mt19937_64 qr(2);
        for(float &x : queries) x = nd(qr);
    }
    double                 build_s = 0;
    lantern_scan_server_t *srv = nullptr;
    if(!external) {
        std::vector<usearch_label_t> labels(rows);
        for(size_t i = 0; i < rows; ++i) labels[ i ] = i + 1;
        const auto tb0 = Clock::now();
        usearch_reserve(ix, rows, &err);
        lantern_gpu_add_many(ix, labels.data(), base.data(), rows, usearch_scalar_f32_k, &err);
        if(!err) lantern_gpu_flush(ix, &err);
        if(err) { std::fprintf(stderr, "%s\n", err); return 1; }
        build_s = std::chrono::duration<double>(Clock::now() - tb0).count();
        std::vector<float>().swap(base);
        srv = lantern_scan_server_start(ix, "127.0.0.1", 0, max_batch, wait_us, &err);
        if(!srv) { std::fprintf(stderr, "%s\n", err ? err : "cannot start the scan server"); return 1; }
    }
    const int   port = external ? ext_port : lantern_scan_server_port(srv);
    const char *host = ext_host.c_str();

    std::atomic<int>      phase{ 0 };  // 0 warm-up, 1 timed, 2 stop
    std::atomic<size_t>   failures{ 0 }, connected{ 0 };
    std::vector<std::vector<uint32_t>> lat(connections);  // microseconds, timed phase only
    std::vector<std::thread>           threads;
    // ---- T threads, each with its share of the connections behind one epoll set
    for(size_t w = 0; w < client_threads; ++w) {
        threads.emplace_back([&, w] {
            struct Cl
            {
                int                  fd = -1;
                Clock::time_point    t0;
                std::vector<uint8_t> in;
                std::mt19937         pick;
            };
            const size_t lo = connections * w / client_threads, hi = connections * (w + 1) / client_threads;
            std::vector<Cl> cls(hi - lo);
            const int       ep = ::epoll_create1(0);
            std::vector<uint8_t> req(16 + dim * 4);
            auto send_next = [&](Cl &c) {
                const float   *q = &queries[ (size_t)(c.pick() % pool) * dim ];
                const uint32_t head[ 4 ] = { 0x5152534Cu, (uint32_t)k, 0u, (uint32_t)(dim * 4) };
                std::memcpy(req.data(), head, 16);
                std::memcpy(req.data() + 16, q, dim * 4);
                c.t0 = Clock::now();
                c.in.clear();
                size_t off = 0;
                while(off < req.size()) {
                    const ssize_t r = ::send(c.fd, req.data() + off, req.size() - off, MSG_NOSIGNAL);
                    if(r <= 0) return false;
                    off += (size_t)r;
                }
                return true;
            };
            for(size_t i = 0; i < cls.size(); ++i) {
                Cl &c = cls[ i ];
                c.pick.seed((unsigned)(lo + i) * 7919u + 13u);
                c.fd = ::socket(AF_INET, SOCK_STREAM, 0);
                sockaddr_in a;
                std::memset(&a, 0, sizeof(a));
                a.sin_family = AF_INET;
                a.sin_port = htons((uint16_t)port);
                ::inet_pton(AF_INET, host, &a.sin_addr);
                if(c.fd < 0 || ::connect(c.fd, (sockaddr *)&a, sizeof(a)) != 0) { failures++; if(c.fd >= 0) ::close(c.fd); c.fd = -1; continue; }
                int one = 1;
                ::setsockopt(c.fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
                connected++;
                epoll_event ev;
                std::memset(&ev, 0, sizeof(ev));
                ev.events = EPOLLIN;
                ev.data.u64 = i;
                ::epoll_ctl(ep, EPOLL_CTL_ADD, c.fd, &ev);
                if(!send_next(c)) failures++;
            }
            auto &mylat = lat[ lo < lat.size() ? lo : 0 ];
            mylat.reserve(1 << 20);
            epoll_event evs[ 256 ];
            const size_t full = 12 + k * 12;  // an answer with k rows
            while(phase.load(std::memory_order_relaxed) < 2) {
                const int n = ::epoll_wait(ep, evs, 256, 50);
                for(int e = 0; e < n; ++e) {
                    Cl     &c = cls[ evs[ e ].data.u64 ];
                    uint8_t tmp[ 4096 ];
                    const ssize_t r = ::recv(c.fd, tmp, std::min(sizeof(tmp), full - c.in.size()), MSG_DONTWAIT);
                    if(r <= 0) { if(r == 0 || (errno != EAGAIN && errno != EINTR)) { failures++; ::epoll_ctl(ep, EPOLL_CTL_DEL, c.fd, nullptr); } continue; }
                    c.in.insert(c.in.end(), tmp, tmp + r);
                    if(c.in.size() < 12) continue;
                    uint32_t rep[ 3 ];
                    std::memcpy(rep, c.in.data(), 12);
                    if(rep[ 0 ] != 0x5052534Cu || rep[ 1 ] != 0 || rep[ 2 ] != k) { failures++; ::epoll_ctl(ep, EPOLL_CTL_DEL, c.fd, nullptr); continue; }
                    if(c.in.size() < full) continue;
                    if(phase.load(std::memory_order_relaxed) == 1)
                        mylat.push_back((uint32_t)std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - c.t0).count());
                    if(!send_next(c)) { failures++; ::epoll_ctl(ep, EPOLL_CTL_DEL, c.fd, nullptr); }
                }
            }
            for(Cl &c : cls)
                if(c.fd >= 0) ::close(c.fd);
            ::close(ep);
        });
    }
    for(size_t c = 0; client_threads == 0 && c < connections; ++c) {
        threads.emplace_back([&, c] {
            usearch_error_t          e = nullptr;
            lantern_scan_client_t   *cl = lantern_scan_client_connect(host, port, &e);
            if(!cl) { failures++; return; }
            connected++;
            std::vector<usearch_label_t> lab(k);
            std::vector<float>           dist(k);
            std::mt19937                 pick((unsigned)c * 7919u + 13u);
            lat[ c ].reserve(1 << 16);
            while(phase.load(std::memory_order_relaxed) < 2) {
                const float *q = &queries[ (size_t)(pick() % pool) * dim ];
                const auto   t0 = Clock::now();
                const size_t got = lantern_scan_client_search(cl, q, dim * 4, k, 0, lab.data(), dist.data(), &e);
                const auto   t1 = Clock::now();
                if(e || got != k) { failures++; if(e) break; }
                if(phase.load(std::memory_order_relaxed) == 1) lat[ c ].push_back((uint32_t)std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count());
            }
            lantern_scan_client_close(cl);
        });
    }
    std::this_thread::sleep_for(std::chrono::duration<double>(warm));
    uint64_t r0 = 0, b0 = 0, l0 = 0, big = 0;
    uint64_t h0[ 16 ] = {}, h1[ 16 ] = {};
    if(srv) {
        lantern_scan_server_stats(srv, &r0, &b0, &l0, &big);
        lantern_scan_server_batch_histogram(srv, h0, 16);
    }
    const auto t0 = Clock::now();
    phase = 1;
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    phase = 2;
    const double elapsed = std::chrono::duration<double>(Clock::now() - t0).count();
    uint64_t r1 = 0, b1 = 0, l1 = 0;
    if(srv) {
        lantern_scan_server_stats(srv, &r1, &b1, &l1, &big);
        lantern_scan_server_batch_histogram(srv, h1, 16);
    }
    for(auto &t : threads) t.join();
    std::vector<uint32_t> all;
    for(auto &v : lat) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end());
    auto pct = [&](double p) { return all.empty() ? 0u : all[ std::min(all.size() - 1, (size_t)(p * (double)all.size())) ]; };
    double mean = 0;
    for(uint32_t x : all) mean += x;
    mean = all.empty() ? 0 : mean / (double)all.size();
    std::printf("{\"tool\": \"lantern-scan-load\", \"index\": \"%zux%zu f32 l2sq M=%zu ef_construction=%zu ef=%zu\", \"k\": %zu, \"connections\": %zu, "
                "\"connected\": %zu, \"client_threads\": %zu, \"max_batch\": %zu, \"max_wait_us\": %u, \"seconds\": %.3f, \"queries\": %zu, \"queries_per_s\": %.1f, "
                "\"latency_us\": {\"mean\": %.1f, \"p50\": %u, \"p90\": %u, \"p99\": %u, \"max\": %u}, \"failures\": %zu, "
                "\"service\": {\"requests\": %llu, \"batches\": %llu, \"launches\": %llu, \"mean_batch\": %.1f, \"largest_batch\": %llu, \"batch_size_histogram\": {",
                rows, dim, m, efc, ef, k, connections, connected.load(), client_threads, max_batch, wait_us, elapsed, all.size(), (double)all.size() / elapsed, mean, pct(0.5), pct(0.9),
                pct(0.99), all.empty() ? 0u : all.back(), failures.load(), (unsigned long long)(r1 - r0), (unsigned long long)(b1 - b0),
                (unsigned long long)(l1 - l0), b1 > b0 ? (double)(r1 - r0) / (double)(b1 - b0) : 0.0, (unsigned long long)big);
    bool first = true;
    for(int b = 0; b < 16; ++b) {
        if(h1[ b ] == h0[ b ]) continue;
        std::printf("%s\"%d-%d\": %llu", first ? "" : ", ", 1 << b, (2 << b) - 1, (unsigned long long)(h1[ b ] - h0[ b ]));
        first = false;
    }
    std::printf("}}, \"index_build_seconds\": %.2f}\n", build_s);
    if(srv) lantern_scan_server_stop(srv);
    if(ix) usearch_free(ix, &err);
    return failures.load() ? 1 : 0;
}
