// mirror_cache.cpp -- lifecycle of the HBM mirrors of page-resident indexes ("keep an HBM
// mirror of the index keyed by (relfilenode, LSN)").
//
// The reference attaches to an index anew for every scan and every insert: usearch_init + usearch_view_mem_lazy per
// ldb_ambeginscan (lantern_hnsw/src/hnsw/scan.c:99-110) and per ldb_aminsert (insert.c:142-151; "todo:: do usearch init in
// indexInfo->ii_AmCache").  That is free there -- the view is lazy, nodes are fetched per hop through the retriever --
// but a device mirror is a WALK of the whole graph through the retriever plus an upload, so it has to outlive the call
// that made it.  This cache owns the mirrors of a process (a backend, or the scan-side service):
//
//   key       (relation, version): the index's relfilenode and whatever the caller uses as its change stamp -- the LSN
//             of the header page, or HnswIndexHeaderPage.num_vectors (external_index.h:38-56) for an append-only index
//   acquire   same relation + same version -> the resident mirror, reference counted (a HIT: no retriever call at all);
//             same relation, another version -> a fresh mirror replaces it (the stale one is freed when its last user
//             releases it); unknown relation -> a mirror is built (usearch_init + usearch_view_mem_lazy)
//   callbacks the retriever / retriever_mut / retriever_ctx of the caller's init options are the reference's per-scan, per-insert
//             RetrieverCtx (scan.c:34,132, insert.c:130,247): the mirror carries the LATEST acquirer's (or rebind's) and drops
//             them at any release -- no ctx pointer outlives the acquire / release pair that brought it
//   advance   the holder of a mirror that applied a change itself (usearch_add_external + usearch_update_header in
//             ldb_aminsert) re-stamps it instead of forcing a rebuild
//   invalidate  DROP INDEX / REINDEX / VACUUM: the relation's mirror goes as soon as nobody holds it
//   capacity  at most `max_resident` idle mirrors stay in HBM (least recently used goes first)
//
// Scans on a shared mirror keep their continuation state in their own cursor (lantern_scan_begin / lantern_gpu_cursor_*).
// Policy, not fallback: below `min_vectors` acquire declines (returns NULL with no error) and the caller stays on the
// path it has -- the reference's in-process usearch; this library itself never computes on the CPU.
#include <cstring>
#include <list>
#include <mutex>
#include <string>

#include "index.hpp"
#include "abi_guard.hpp"

struct lantern_mirror
{
    uint64_t        relation = 0, version = 0;
    usearch_index_t index = nullptr;
    int             refs = 0;
    bool            stale = false;  // replaced or invalidated: free at the last release
    uint64_t        last_use = 0;
};

namespace {
struct Cache
{
    std::mutex                 mu;
    std::list<lantern_mirror>  entries;
    size_t                     max_resident = 8;
    uint64_t                   clock = 0, hits = 0, misses = 0, rebuilds = 0;
};
Cache &cache()
{
    static Cache c;
    return c;
}

// the mirror's callbacks become this holder's (NULL opts: nobody's -- no ctx pointer outlives an acquire / release pair)
void rebind_retriever(lgpu::Index *ix, const usearch_init_options_t *opts)
{
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    ix->opts.retriever = opts ? opts->retriever : nullptr;
    ix->opts.retriever_mut = opts ? opts->retriever_mut : nullptr;
    ix->opts.retriever_ctx = opts ? opts->retriever_ctx : nullptr;
}

void destroy(lantern_mirror &m)
{
    usearch_error_t ignore = nullptr;
    if(m.index) usearch_free(m.index, &ignore);
    m.index = nullptr;
}

// under c.mu: free stale idle entries, then idle entries beyond the capacity, least recently used first
void trim(Cache &c)
{
    for(auto it = c.entries.begin(); it != c.entries.end();) {
        if(it->stale && it->refs == 0) {
            destroy(*it);
            it = c.entries.erase(it);
        } else {
            ++it;
        }
    }
    for(;;) {
        size_t idle = 0;
        auto   victim = c.entries.end();
        for(auto it = c.entries.begin(); it != c.entries.end(); ++it) {
            if(it->refs != 0) continue;
            ++idle;
            if(victim == c.entries.end() || it->last_use < victim->last_use) victim = it;
        }
        if(idle <= c.max_resident || victim == c.entries.end()) break;
        destroy(*victim);
        c.entries.erase(victim);
    }
}
}  // namespace

extern "C" {

lantern_mirror_t *lantern_mirror_acquire(uint64_t relation, uint64_t version, usearch_init_options_t *opts, float *pq_codebook, char *header136,
                                         size_t min_vectors, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    if(!opts || !header136) { if(e) *e = "lantern_gpu: null init options or header"; return nullptr; }
    Cache &c = cache();
    {
        std::lock_guard<std::mutex> g(c.mu);
        for(auto &m : c.entries) {
            if(m.relation == relation && !m.stale && m.version == version) {
                // The reference allocates its RetrieverCtx per scan and per insert and frees it at the end (scan.c:34,132,
                // insert.c:130,247): the callbacks and the ctx a mirror was BUILT with are gone by now.  The mirror takes
                // this holder's for as long as it holds it (usearch_add_external writes through retriever_mut).
                rebind_retriever((lgpu::Index *)m.index, opts);
                m.refs++;
                m.last_use = ++c.clock;
                c.hits++;
                return &m;
            }
        }
    }
    // policy: a tiny index is not worth a mirror (the header carries the node count: external_index.h:59-66)
    uint64_t declared = 0;
    std::memcpy(&declared, header136 + 80, 8);
    if(declared < min_vectors) return nullptr;
    // The build -- a walk of the whole page graph through the retriever plus an upload -- runs OUTSIDE the cache lock:
    // scans of other relations keep acquiring and releasing meanwhile.
    usearch_error_t err = nullptr;
    usearch_index_t ix = usearch_init(opts, pq_codebook, &err);
    if(!ix) { if(e) *e = err; return nullptr; }
    usearch_view_mem_lazy(ix, header136, &err);
    if(err) {
        // the message belongs to the index: keep it alive in a static buffer before the index goes
        static thread_local std::string kept;
        kept = err;
        usearch_error_t ignore = nullptr;
        usearch_free(ix, &ignore);
        if(e) *e = kept.c_str();
        return nullptr;
    }
    std::lock_guard<std::mutex> g(c.mu);
    // somebody else may have built the same (relation, version) in the meantime: theirs stays, ours goes
    for(auto &m : c.entries) {
        if(m.relation == relation && !m.stale && m.version == version) {
            usearch_error_t ignore = nullptr;
            usearch_free(ix, &ignore);
            rebind_retriever((lgpu::Index *)m.index, opts);
            m.refs++;
            m.last_use = ++c.clock;
            c.hits++;
            return &m;
        }
    }
    bool replaced = false;
    for(auto &m : c.entries)
        if(m.relation == relation && !m.stale) { m.stale = true; replaced = true; }
    (replaced ? c.rebuilds : c.misses)++;
    c.entries.emplace_back();
    lantern_mirror &m = c.entries.back();
    m.relation = relation;
    m.version = version;
    m.index = ix;
    m.refs = 1;
    m.last_use = ++c.clock;
    trim(c);
    return &m;
}
LANTERN_ABI_CATCH(e)

usearch_index_t lantern_mirror_index(lantern_mirror_t *m) { return m ? m->index : nullptr; }
uint64_t        lantern_mirror_version(lantern_mirror_t *m) { return m ? m->version : 0; }

void lantern_mirror_rebind(lantern_mirror_t *m, const usearch_init_options_t *opts)
try {
    if(m) rebind_retriever((lgpu::Index *)m->index, opts);
}
LANTERN_ABI_CATCH_VOID(nullptr)

void lantern_mirror_advance(lantern_mirror_t *m, uint64_t new_version)
try {
    if(!m) return;
    Cache &c = cache();
    std::lock_guard<std::mutex> g(c.mu);
    m->version = new_version;
}
LANTERN_ABI_CATCH_VOID(nullptr)

void lantern_mirror_release(lantern_mirror_t *m)
try {
    if(!m) return;
    Cache &c = cache();
    std::lock_guard<std::mutex> g(c.mu);
    if(m->refs > 0) m->refs--;
    // The releasing holder's ctx is about to be freed (scan.c:132, insert.c:247) and the entry cannot tell whose callbacks it
    // carries: ANY release unbinds them.  Scans never call them on a mirror; an inserter that still holds the mirror binds
    // its own again (lantern_mirror_rebind) -- without that usearch_add_external fails with a message, never through a
    // dangling pointer.
    rebind_retriever((lgpu::Index *)m->index, nullptr);
    m->last_use = ++c.clock;
    trim(c);
}
LANTERN_ABI_CATCH_VOID(nullptr)

void lantern_mirror_invalidate(uint64_t relation)
try {
    Cache &c = cache();
    std::lock_guard<std::mutex> g(c.mu);
    for(auto &m : c.entries)
        if(m.relation == relation) m.stale = true;
    trim(c);
}
LANTERN_ABI_CATCH_VOID(nullptr)

void lantern_mirror_set_capacity(size_t max_resident)
try {
    Cache &c = cache();
    std::lock_guard<std::mutex> g(c.mu);
    c.max_resident = max_resident;
    trim(c);
}
LANTERN_ABI_CATCH_VOID(nullptr)

void lantern_mirror_stats(uint64_t *hits, uint64_t *misses, uint64_t *rebuilds, uint64_t *resident)
try {
    Cache &c = cache();
    std::lock_guard<std::mutex> g(c.mu);
    if(hits) *hits = c.hits;
    if(misses) *misses = c.misses;
    if(rebuilds) *rebuilds = c.rebuilds;
    if(resident) *resident = c.entries.size();
}
LANTERN_ABI_CATCH_VOID(nullptr)

}  // extern "C"
