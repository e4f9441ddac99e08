// scan_shim.cpp -- the index-AM side of a scan, restated over the device index:
// ldb_ambeginscan / ldb_amrescan / ldb_amgettuple / ldb_amendscan of
// lantern_hnsw/src/hnsw/scan.c:24-338, including the HnswScanState fields of scan.h:12-34.
//
// The PostgreSQL executor calls amgettuple once per returned row; the first call runs one search
// with k = lantern_hnsw.init_k, later calls pop buffered labels, and when the buffer is exhausted
// the search is continued for 2*count more results through the streaming form of
// usearch_search_ef.  Deleted rows carry label 0 and are skipped.
#include <cstring>
#include <vector>

#include "index.hpp"

struct lantern_scan
{
    usearch_index_t       index;
    int                   init_k;  // GUC lantern_hnsw.init_k  (options.c:324-348, options.h:44)
    int                   ef;      // GUC lantern_hnsw.ef, 0 = use the index's ef (scan.c:179)
    bool                  first;
    bool                  armed;
    usearch_scalar_kind_t scalar;
    std::vector<char>     query;
    std::vector<float>    distances;         // HnswScanState.distances
    std::vector<usearch_label_t> labels;     // HnswScanState.labels
    int                   count, current;    // HnswScanState.count / .current
};

static const usearch_label_t INVALID_ELEMENT_LABEL = 0;  // lantern_hnsw/src/hnsw.h:40

extern "C" {

lantern_scan_t *lantern_scan_begin(usearch_index_t index, int init_k, int ef, usearch_error_t *e)
{
    if(e) *e = nullptr;
    if(!index) { if(e) *e = "lantern_gpu: null index handle"; return nullptr; }
    if(init_k < 1 || init_k > 1000) { if(e) *e = "lantern_hnsw.init_k must be in [1, 1000]"; return nullptr; }  // options.c:324-336
    lantern_scan *s = new lantern_scan();
    s->index = index;
    s->init_k = init_k;
    s->ef = ef;
    s->first = true;
    s->armed = false;
    s->scalar = usearch_scalar_unknown_k;
    s->count = s->current = 0;
    return s;
}

void lantern_scan_rescan(lantern_scan_t *s, const void *query, usearch_scalar_kind_t kind, usearch_error_t *e)
{
    if(e) *e = nullptr;
    if(!s || !query) { if(e) *e = "cannot scan hnsw index without order"; return; }  // scan.c:192
    lgpu::Index *ix = (lgpu::Index *)s->index;
    if(!lgpu::kind_accepted(ix, (int)kind)) { if(e) *e = "lantern_gpu: scalar kind of the query does not match the index"; return; }
    const size_t bytes = lgpu::input_bytes(ix, (int)kind);
    s->query.assign((const char *)query, (const char *)query + bytes);
    s->scalar = kind;
    s->first = true;  // ldb_amrescan: scanstate->first = true (scan.c:150)
    s->armed = true;
    s->count = s->current = 0;
}

bool lantern_scan_gettuple(lantern_scan_t *s, usearch_label_t *label, usearch_error_t *e)
{
    if(e) *e = nullptr;
    if(!s || !s->armed) { if(e) *e = "cannot scan hnsw index without order"; return false; }
    usearch_error_t err = nullptr;
    if(s->first) {
        const int k = s->init_k;  // scan.c:186
        s->distances.resize((size_t)k);
        s->labels.resize((size_t)k);
        const size_t got = usearch_search_ef(s->index, s->query.data(), s->scalar, (size_t)k, (size_t)s->ef,
                                             false /* the first round is never streaming */, s->labels.data(),
                                             s->distances.data(), &err);
        if(err) { if(e) *e = err; return false; }
        s->count = (int)got;
        s->current = 0;
        s->first = false;
    }
    if(s->current == s->count) {  // scan.c:240-292
        const int    k = s->count * 2;
        const size_t index_size = usearch_size(s->index, &err);
        if(s->count >= 1000) return false;  // "skipping streaming after loading 1000 elements" (scan.c:249-252)
        if((int)index_size == s->current) return false;  // scan.c:254-256
        if(k == 0) return false;
        s->distances.resize((size_t)k);
        s->labels.resize((size_t)k);
        const size_t got = usearch_search_ef(s->index, s->query.data(), s->scalar, (size_t)k, (size_t)s->ef, true /* streaming */,
                                             s->labels.data(), s->distances.data(), &err);
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

void lantern_scan_end(lantern_scan_t *s) { delete s; }

}  // extern "C"
