// lantern-index-load -- the PostgreSQL side of `CREATE INDEX ... WITH (external = true)` as a load generator: streams N
// synthetic rows to the external indexing server over its socket exactly as lantern_hnsw does (one write of
// [u64 label][dim x f32] per tuple: external_index_socket.c:517-536; handshake and init frame :322-486; END_MSG, then
// [u64 rows added][u64 file size][usearch-format file]: :488-515) and reports vectors/s END TO END -- first tuple sent to
// last byte of the index file received -- beside the time the stream itself took.
//   --host H --port P            an indexing server that is already running (lantern-index-server, or the reference's)
//   (default)                    starts liblantern_gpu.so's server in this process on a loopback port
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lantern_gpu.h"

using Clock = std::chrono::steady_clock;
static const uint32_t INIT_MSG = 0x13333337u, END_MSG = 0x31333337u, ERR_MSG = 0x37333337u;

static bool write_all(int fd, const void *buf, size_t n)
{
    const char *p = (const char *)buf;
    while(n) {
        ssize_t w = ::send(fd, p, n, MSG_NOSIGNAL);
        if(w <= 0) return false;
        p += w;
        n -= (size_t)w;
    }
    return true;
}
static bool read_exact(int fd, void *buf, size_t n)
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

int main(int argc, char **argv)
{
    std::string host;
    int         port = 0;
    size_t      rows = 1000000, dim = 1536, m = 16, efc = 128, ef = 64, per_write = 1;
    uint32_t    metric = 3;  // l2sq (cli.rs:56-69)
    for(int i = 1; i < argc; ++i) {
        auto val = [&](const char *name) -> const char * { return std::strcmp(argv[ i ], name) == 0 && i + 1 < argc ? argv[ ++i ] : nullptr; };
        if(const char *v = val("--host")) host = v;
        else if(const char *v = val("--port")) port = std::atoi(v);
        else if(const char *v = val("--rows")) rows = (size_t)std::atoll(v);
        else if(const char *v = val("--dim")) dim = (size_t)std::atoll(v);
        else if(const char *v = val("--m")) m = (size_t)std::atoll(v);
        else if(const char *v = val("--ef-construction")) efc = (size_t)std::atoll(v);
        else if(const char *v = val("--ef")) ef = (size_t)std::atoll(v);
        else if(const char *v = val("--metric")) metric = std::strcmp(v, "cos") == 0 ? 1u : 3u;
        else if(const char *v = val("--tuples-per-write")) per_write = std::max<size_t>(1, (size_t)std::atoll(v));  // 1 = what PostgreSQL does
        else {
            std::fprintf(stderr, "usage: %s [--host H --port P] [--rows N --dim D --m M --ef-construction E --ef E --metric l2sq|cos]\n", argv[ 0 ]);
            return 2;
        }
    }
    usearch_error_t           err = nullptr;
    lantern_index_server_t   *srv = nullptr;
    if(host.empty()) {
        srv = lantern_index_server_start("127.0.0.1", 0, -1, "/tmp", &err);
        if(!srv) { std::fprintf(stderr, "%s\n", err ? err : "cannot start the indexing server"); return 1; }
        host = "127.0.0.1";
        port = lantern_index_server_port(srv);
    }
    // the tuples, generated before the clock starts (uniform [-1, 1): a cheap generator, eight threads)
    const size_t       tuple = 8 + dim * 4;
    std::vector<char>  data(rows * tuple);
    {
        std::vector<std::thread> gen;
        const size_t             T = 8;
        for(size_t t = 0; t < T; ++t)
            gen.emplace_back([&, t] {
                uint64_t s = 0x9E3779B97F4A7C15ull * (t + 1);
                for(size_t r = rows * t / T; r < rows * (t + 1) / T; ++r) {
                    char          *p = &data[ r * tuple ];
                    const uint64_t label = r + 1;
                    std::memcpy(p, &label, 8);
                    float *v = (float *)(p + 8);
                    for(size_t j = 0; j < dim; ++j) {
                        s ^= s << 13; s ^= s >> 7; s ^= s << 17;  // xorshift64
                        v[ j ] = (float)(int32_t)(s >> 32) * (1.0f / 2147483648.0f);
                    }
                }
            });
        for(auto &t : gen) t.join();
    }
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in a;
    std::memset(&a, 0, sizeof(a));
    a.sin_family = AF_INET;
    a.sin_port = htons((uint16_t)port);
    if(fd < 0 || ::inet_pton(AF_INET, host.c_str(), &a.sin_addr) != 1 || ::connect(fd, (sockaddr *)&a, sizeof(a)) != 0) { std::perror("connect"); return 1; }
    uint32_t hello[ 2 ];
    if(!read_exact(fd, hello, 8) || hello[ 0 ] != 1 || hello[ 1 ] != 1) { std::fprintf(stderr, "unexpected server greeting\n"); return 1; }
    // external_index_params_t: pq, metric_kind, quantization, dim, m, efc, ef, num_centroids, num_subvectors, capacity, element_bits
    const uint32_t init[ 12 ] = { INIT_MSG, 0, metric, 1, (uint32_t)dim, (uint32_t)m, (uint32_t)efc, (uint32_t)ef, 0, 0, (uint32_t)rows, 32 };
    uint8_t        ok = 1;
    if(!write_all(fd, init, sizeof(init)) || !read_exact(fd, &ok, 1) || ok != 0) { std::fprintf(stderr, "the server refused the init frame\n"); return 1; }
    const auto t0 = Clock::now();
    for(size_t r = 0; r < rows; r += per_write) {  // one write per tuple, as PostgreSQL (more per write only to find the server's own limit)
        const size_t cnt = std::min(per_write, rows - r);
        if(!write_all(fd, &data[ r * tuple ], tuple * cnt)) { std::fprintf(stderr, "the stream broke at tuple %zu\n", r); return 1; }
    }
    if(!write_all(fd, &END_MSG, 4)) return 1;
    const auto t_sent = Clock::now();
    uint64_t   added = 0, size = 0;
    uint32_t   head = 0;
    if(!read_exact(fd, &head, 4)) return 1;
    if(head == ERR_MSG) {
        uint32_t n = 0;
        read_exact(fd, &n, 4);
        std::string msg(n, '\0');
        read_exact(fd, &msg[ 0 ], n);
        std::fprintf(stderr, "server error: %s\n", msg.c_str());
        return 1;
    }
    uint32_t hi = 0;
    if(!read_exact(fd, &hi, 4) || !read_exact(fd, &size, 8)) return 1;
    added = ((uint64_t)hi << 32) | head;
    const auto        t_built = Clock::now();
    std::vector<char> file(size);
    if(!read_exact(fd, file.data(), size)) return 1;
    const auto t_end = Clock::now();
    ::close(fd);
    auto secs = [](Clock::time_point x, Clock::time_point y) { return std::chrono::duration<double>(y - x).count(); };
    std::printf("{\"tool\": \"lantern-index-load\", \"rows\": %zu, \"dim\": %zu, \"m\": %zu, \"ef_construction\": %zu, \"metric\": %u, \"rows_added\": %llu, "
                "\"stream_seconds\": %.3f, \"stream_vectors_per_s\": %.0f, \"stream_GB_per_s\": %.2f, \"until_index_ready_seconds\": %.3f, "
                "\"index_file_bytes\": %llu, \"file_download_seconds\": %.3f, \"end_to_end_seconds\": %.3f, \"end_to_end_vectors_per_s\": %.0f, "
                "\"server\": \"%s\", \"tuples_per_write\": %zu}\n",
                rows, dim, m, efc, metric, (unsigned long long)added, secs(t0, t_sent), (double)rows / secs(t0, t_sent), (double)(rows * tuple) / secs(t0, t_sent) / 1e9,
                secs(t0, t_built), (unsigned long long)size, secs(t_built, t_end), secs(t0, t_end), (double)rows / secs(t0, t_end), srv ? "in-process liblantern_gpu.so" : "remote", per_write);
    if(srv) lantern_index_server_stop(srv);
    return added == rows ? 0 : 1;
}
