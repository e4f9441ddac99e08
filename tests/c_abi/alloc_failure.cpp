// An allocation failure inside the library must come back as an error string, never as an exception through the C boundary
// (std::terminate in a PostgreSQL backend; the reference warns about this class of failure: lantern_hnsw/src/hnsw/utils.h:22-25).
//
// This program replaces the global operator new with one that fails the N-th allocation OF THE CALLING THREAD once armed, and
// sweeps N over the first allocations of host-only entry points (no device needed): either the call succeeds, or it reports an
// error string and returns its "nothing" value -- the process must survive every N.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "lantern_gpu.h"

static thread_local long g_countdown = -1;  // -1 = disarmed
static thread_local long g_failed = 0;

void *operator new(std::size_t n)
{
    if(g_countdown >= 0 && g_countdown-- == 0) {
        g_countdown = -1;
        ++g_failed;
        throw std::bad_alloc();
    }
    void *p = std::malloc(n ? n : 1);
    if(!p) throw std::bad_alloc();
    return p;
}
void *operator new[](std::size_t n) { return operator new(n); }
void  operator delete(void *p) noexcept { std::free(p); }
void  operator delete[](void *p) noexcept { std::free(p); }
void  operator delete(void *p, std::size_t) noexcept { std::free(p); }
void  operator delete[](void *p, std::size_t) noexcept { std::free(p); }

static int batch_fn(void *, const void *, size_t nq, size_t, size_t, size_t, usearch_label_t *, float *, uint32_t *counts, const char **)
{
    for(size_t i = 0; i < nq; ++i) counts[ i ] = 0;
    return 0;
}

int main()
{
    int errors_seen = 0, successes = 0;
    // (1) the in-process communicator group: one shared hub + `world` handles
    for(long n = 0; n < 12; ++n) {
        lantern_gpu_comm_t *out[ 4 ] = { nullptr, nullptr, nullptr, nullptr };
        usearch_error_t     e = nullptr;
        g_failed = 0;
        g_countdown = n;
        lantern_gpu_comm_init_local(4, out, &e);
        g_countdown = -1;
        if(g_failed) {
            if(!e || !std::strstr(e, "out of host memory")) { std::printf("comm_init_local: allocation %ld failed but error is %s\n", n, e ? e : "(null)"); return 1; }
            ++errors_seen;
        } else {
            if(e) { std::printf("comm_init_local: unexpected error %s\n", e); return 1; }
            ++successes;
            for(auto *c : out) lantern_gpu_comm_free(c);
        }
    }
    // (2) the scan service over a caller-supplied batch function (threads, sockets, queues)
    for(long n = 0; n < 6; ++n) {
        usearch_error_t e = nullptr;
        g_failed = 0;
        g_countdown = n;
        lantern_scan_server_t *s = lantern_scan_server_start_fn(batch_fn, nullptr, 16, "127.0.0.1", 0, 8, 100, &e);
        g_countdown = -1;
        if(g_failed) {
            if(s || !e) { std::printf("scan_server_start_fn: allocation %ld failed, server %p error %s\n", n, (void *)s, e ? e : "(null)"); return 1; }
            ++errors_seen;
        } else if(s) {
            ++successes;
            lantern_scan_server_stop(s);
        }
    }
    std::printf("ok: %d failures reported as error strings, %d calls succeeded\n", errors_seen, successes);
    return errors_seen > 0 ? 0 : 1;
}
