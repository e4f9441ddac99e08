// scan_shim.cpp -- the index-AM side of a scan, restated over the device index:
// ldb_ambeginscan / ldb_amrescan / ldb_amgettuple / ldb_amendscan of
// lantern_hnsw/src/hnsw/scan.c:24-338, including the HnswScanState fields of scan.h:12-34.
//
// The PostgreSQL executor calls amgettuple once per returned row; the first call runs one search
// with k = lantern_hnsw.init_k, later calls pop buffered labels, and when the buffer is exhausted
// the search is continued for 2*count more results through the streaming form of
// usearch_search_ef.  Deleted rows carry label 0 and are skipped.  The continuation state (what has been handed
// out so far) belongs to the SCAN: any number of scans may be open on one index at a time (two cursors, the two
// sides of a nested loop) and each pages through its own result.
//
// Two back ends: a cursor on a local index (lantern_scan_begin), or a connection to the scan-side service
// (lantern_scan_begin_client, scan_server.cpp) when the HBM mirror lives in another process.
#include <cstring>
#include <vector>

#include "index.hpp"
#include "abi_guard.hpp"

struct lantern_scan
{
    usearch_index_t        index = nullptr;   // local back end
    lantern_gpu_cursor_t  *cursor = nullptr;  // what THIS scan has been handed so far: in the reference every scan owns its own
                                              // usearch handle (scan.c:99); here many scans share one resident index
    lantern_scan_client_t *client = nullptr;  // service back end (the connection holds the continuation state)
    size_t                 client_query_bytes = 0;
    int                    init_k = 10;  // GUC lantern_hnsw.init_k  (options.c:324-348, options.h:44)
    int                    ef = 0;       // GUC lantern_hnsw.ef, 0 = use the index's ef (scan.c:179)
    bool                   first = true;
    bool                   armed = false;
    usearch_scalar_kind_t  scalar = usearch_scalar_unknown_k;
    std::vector<char>      query;
    std::vector<float>     distances;         // HnswScanState.distances
    std::vector<usearch_label_t> labels;      // HnswScanState.labels
    int                    count = 0, current = 0;  // HnswScanState.count / .current
    std::vector<int>       k_trace;           // the k of every usearch_search_ef this scan issued since its last rescan: what the
                                              // reference logs as "querying index for %d elements" (scan.c:219, :272)
};

static const usearch_label_t INVALID_ELEMENT_LABEL = 0;  // lantern_hnsw/src/hnsw.h:40

// one round of usearch_search_ef on behalf of the scan
static size_t scan_search(lantern_scan *s, size_t k, bool streaming, usearch_error_t *err)
{
    s->distances.resize(k);
    s->labels.resize(k);
    s->k_trace.push_back((int)k);
    if(s->client)
        return streaming ? lantern_scan_client_search_next(s->client, s->query.data(), s->query.size(), k, (size_t)s->ef, s->labels.data(),
                                                           s->distances.data(), err)
                         : lantern_scan_client_search(s->client, s->query.data(), s->query.size(), k, (size_t)s->ef, s->labels.data(),
                                                      s->distances.data(), err);
    return lantern_gpu_cursor_search(s->cursor, s->query.data(), s->scalar, k, (size_t)s->ef, streaming, s->labels.data(), s->distances.data(), err);
}

extern "C" {

lantern_scan_t *lantern_scan_begin(usearch_index_t index, int init_k, int ef, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    if(!index) { if(e) *e = "lantern_gpu: null index handle"; return nullptr; }
    if(init_k < 1 || init_k > 1000) { if(e) *e = "lantern_hnsw.init_k must be in [1, 1000]"; return nullptr; }  // options.c:324-336
    lantern_gpu_cursor_t *cur = lantern_gpu_cursor_open(index, e);
    if(!cur) return nullptr;
    lantern_scan *s = new lantern_scan();
    s->index = index;
    s->cursor = cur;
    s->init_k = init_k;
    s->ef = ef;
    return s;
}
LANTERN_ABI_CATCH(e)

lantern_scan_t *lantern_scan_begin_client(lantern_scan_client_t *client, size_t query_bytes, int init_k, int ef, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    if(!client || query_bytes == 0) { if(e) *e = "lantern_gpu: null scan-service connection or empty query size"; return nullptr; }
    if(init_k < 1 || init_k > 1000) { if(e) *e = "lantern_hnsw.init_k must be in [1, 1000]"; return nullptr; }
    lantern_scan *s = new lantern_scan();
    s->client = client;
    s->client_query_bytes = query_bytes;
    s->init_k = init_k;
    s->ef = ef;
    return s;
}
LANTERN_ABI_CATCH(e)

void lantern_scan_rescan(lantern_scan_t *s, const void *query, usearch_scalar_kind_t kind, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    if(!s || !query) { if(e) *e = "cannot scan hnsw index without order"; return; }  // scan.c:192
    size_t bytes = s->client_query_bytes;
    if(!s->client) {
        lgpu::Index *ix = (lgpu::Index *)s->index;
        if(!lgpu::kind_accepted(ix, (int)kind)) { if(e) *e = "lantern_gpu: scalar kind of the query does not match the index"; return; }
        bytes = lgpu::input_bytes(ix, (int)kind);
    }
    s->query.assign((const char *)query, (const char *)query + bytes);
    s->scalar = kind;
    s->first = true;  // ldb_amrescan: scanstate->first = true (scan.c:150)
    s->armed = true;
    s->count = s->current = 0;
    s->k_trace.clear();
}
LANTERN_ABI_CATCH_VOID(e)

bool lantern_scan_gettuple(lantern_scan_t *s, usearch_label_t *label, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    if(!s || !s->armed) { if(e) *e = "cannot scan hnsw index without order"; return false; }
    usearch_error_t err = nullptr;
    if(s->first) {
        const size_t got = scan_search(s, (size_t)s->init_k /* scan.c:186 */, false /* the first round is never streaming */, &err);
        if(err) { if(e) *e = err; return false; }
        s->count = (int)got;
        s->current = 0;
        s->first = false;
    }
    if(s->current == s->count) {  // scan.c:240-292
        const int k = s->count * 2;
        if(s->count >= 1000) return false;  // "skipping streaming after loading 1000 elements" (scan.c:249-252)
        if(!s->client) {
            const size_t index_size = usearch_size(s->index, &err);
            if((int)index_size == s->current) return false;  // scan.c:254-256
        }
        if(k == 0) return false;
        const size_t got = scan_search(s, (size_t)k, true /* streaming */, &err);
        if(err) { if(e) *e = err; return false; }
        s->count = (int)got;
        s->current = 0;  // the index returned the NEXT batch, so restart at its head (scan.c:283-286)
    }
    while(s->current < s->count) {  // scan.c:294-335
        const usearch_label_t l = s->labels[ (size_t)s->current ];
        s->current++;
        if(l == INVALID_ELEMENT_LABEL) continue;  // deleted element
        if(label) *label = l;
        return true;
    }
    return false;
}
LANTERN_ABI_CATCH(e)

size_t lantern_scan_trace(lantern_scan_t *s, int *ks, size_t cap)
try {
    if(!s) return 0;
    for(size_t i = 0; i < s->k_trace.size() && i < cap && ks; i++) ks[ i ] = s->k_trace[ i ];
    return s->k_trace.size();
}
LANTERN_ABI_CATCH(nullptr)

void lantern_scan_end(lantern_scan_t *s)
try {
    if(!s) return;
    lantern_gpu_cursor_close(s->cursor);
    delete s;
}
LANTERN_ABI_CATCH_VOID(nullptr)

}  // extern "C"
