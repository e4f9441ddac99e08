// lantern-scan-load -- closed-loop load generator for the scan-side service (lantern_amd/csrc/scan_server.cpp): what N
// PostgreSQL backends running ORDER BY ... LIMIT k scans against ONE HBM-resident index look like to the device.
//
// Builds a synthetic index (i.i.d. N(0,1) rows; --rows x --dim), starts the service on a loopback port, opens --connections
// client connections -- one thread each, the way a backend holds one -- and lets every one issue single k-NN queries back to
// back for --seconds (lantern_scan_client_search: what ldb_amgettuple calls in place of usearch_search_ef,
// lantern_hnsw/src/hnsw/scan.c:220-228).  Prints one JSON line: queries/s, latency percentiles, the service's batch-size
// histogram.  Everything goes through the C ABI of liblantern_gpu.so; the clients speak TCP to the server like separate processes.
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
    size_t   rows = 100000, dim = 128, m = 16, efc = 128, ef = 64, k = 10, connections = 256, max_batch = 1024, pool = 8192;
    unsigned wait_us = 200;
    double   seconds = 5.0, warm = 1.0;
    for(int i = 1; i < argc; ++i) {
        auto val = [&](const char *name) -> const char * { return std::strcmp(argv[ i ], name) == 0 && i + 1 < argc ? argv[ ++i ] : nullptr; };
        if(const char *v = val("--rows")) rows = (size_t)std::atoll(v);
        else if(const char *v = val("--dim")) dim = (size_t)std::atoll(v);
        else if(const char *v = val("--m")) m = (size_t)std::atoll(v);
        else if(const char *v = val("--ef-construction")) efc = (size_t)std::atoll(v);
        else if(const char *v = val("--ef")) ef = (size_t)std::atoll(v);
        else if(const char *v = val("--k")) k = (size_t)std::atoll(v);
        else if(const char *v = val("--connections")) connections = (size_t)std::atoll(v);
        else if(const char *v = val("--max-batch")) max_batch = (size_t)std::atoll(v);
        else if(const char *v = val("--max-wait-us")) wait_us = (unsigned)std::atoi(v);
        else if(const char *v = val("--seconds")) seconds = std::atof(v);
        else if(const char *v = val("--warmup-seconds")) warm = std::atof(v);
        else {
            std::fprintf(stderr, "usage: %s [--rows N --dim D --m M --ef-construction E --ef E --k K] [--connections C] [--max-batch B] "
                                 "[--max-wait-us U] [--seconds S] [--warmup-seconds W]\n", argv[ 0 ]);
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
    usearch_index_t ix = usearch_init(&o, nullptr, &err);
    if(err) { std::fprintf(stderr, "%s\n", err); return 1; }
    std::vector<float> base(rows * dim), queries(pool * dim);
    {
        std::mt19937_64                 rng(1);
        std::normal_distribution<float> nd(0.f, 1.f);
        for(float &x : base) x = nd(rng);
        std::mt19937_64 qr(2);
        for(float &x : queries) x = nd(qr);
    }
    std::vector<usearch_label_t> labels(rows);
    for(size_t i = 0; i < rows; ++i) labels[ i ] = i + 1;
    const auto tb0 = Clock::now();
    usearch_reserve(ix, rows, &err);
    lantern_gpu_add_many(ix, labels.data(), base.data(), rows, usearch_scalar_f32_k, &err);
    if(!err) lantern_gpu_flush(ix, &err);
    if(err) { std::fprintf(stderr, "%s\n", err); return 1; }
    const double build_s = std::chrono::duration<double>(Clock::now() - tb0).count();
    lantern_scan_server_t *srv = lantern_scan_server_start(ix, "127.0.0.1", 0, max_batch, wait_us, &err);
    if(!srv) { std::fprintf(stderr, "%s\n", err ? err : "cannot start the scan server"); return 1; }
    const int port = lantern_scan_server_port(srv);

    std::atomic<int>      phase{ 0 };  // 0 warm-up, 1 timed, 2 stop
    std::atomic<size_t>   failures{ 0 }, connected{ 0 };
    std::vector<std::vector<uint32_t>> lat(connections);  // microseconds, timed phase only
    std::vector<std::thread>           threads;
    for(size_t c = 0; c < connections; ++c) {
        threads.emplace_back([&, c] {
            usearch_error_t          e = nullptr;
            lantern_scan_client_t   *cl = lantern_scan_client_connect("127.0.0.1", port, &e);
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
    uint64_t r0, b0, l0, big;
    uint64_t h0[ 16 ], h1[ 16 ];
    lantern_scan_server_stats(srv, &r0, &b0, &l0, &big);
    lantern_scan_server_batch_histogram(srv, h0, 16);
    const auto t0 = Clock::now();
    phase = 1;
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    phase = 2;
    const double elapsed = std::chrono::duration<double>(Clock::now() - t0).count();
    uint64_t r1, b1, l1;
    lantern_scan_server_stats(srv, &r1, &b1, &l1, &big);
    lantern_scan_server_batch_histogram(srv, h1, 16);
    for(auto &t : threads) t.join();
    std::vector<uint32_t> all;
    for(auto &v : lat) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end());
    auto pct = [&](double p) { return all.empty() ? 0u : all[ std::min(all.size() - 1, (size_t)(p * (double)all.size())) ]; };
    double mean = 0;
    for(uint32_t x : all) mean += x;
    mean = all.empty() ? 0 : mean / (double)all.size();
    std::printf("{\"tool\": \"lantern-scan-load\", \"index\": \"%zux%zu f32 l2sq M=%zu ef_construction=%zu ef=%zu\", \"k\": %zu, \"connections\": %zu, "
                "\"connected\": %zu, \"max_batch\": %zu, \"max_wait_us\": %u, \"seconds\": %.3f, \"queries\": %zu, \"queries_per_s\": %.1f, "
                "\"latency_us\": {\"mean\": %.1f, \"p50\": %u, \"p90\": %u, \"p99\": %u, \"max\": %u}, \"failures\": %zu, "
                "\"service\": {\"requests\": %llu, \"batches\": %llu, \"launches\": %llu, \"mean_batch\": %.1f, \"largest_batch\": %llu, \"batch_size_histogram\": {",
                rows, dim, m, efc, ef, k, connections, connected.load(), max_batch, wait_us, elapsed, all.size(), (double)all.size() / elapsed, mean, pct(0.5), pct(0.9),
                pct(0.99), all.empty() ? 0u : all.back(), failures.load(), (unsigned long long)(r1 - r0), (unsigned long long)(b1 - b0),
                (unsigned long long)(l1 - l0), b1 > b0 ? (double)(r1 - r0) / (double)(b1 - b0) : 0.0, (unsigned long long)big);
    bool first = true;
    for(int b = 0; b < 16; ++b) {
        if(h1[ b ] == h0[ b ]) continue;
        std::printf("%s\"%d-%d\": %llu", first ? "" : ", ", 1 << b, (2 << b) - 1, (unsigned long long)(h1[ b ] - h0[ b ]));
        first = false;
    }
    std::printf("}}, \"index_build_seconds\": %.2f}\n", build_s);
    lantern_scan_server_stop(srv);
    usearch_free(ix, &err);
    return failures.load() ? 1 : 0;
}
