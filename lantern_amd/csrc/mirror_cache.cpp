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
//             RetrieverCtx (scan.c:34,132, insert.c:130,247).  The mirror keeps them PER HOLDER -- a holder is a host thread: a
//             PostgreSQL backend is one, a threaded service runs one holder per thread -- bound by that thread's acquire /
//             rebind and dropped by that thread's release: no ctx pointer outlives the acquire / release pair that brought
//             it, and one holder's release never takes away another's (usearch_add_external looks up the CALLING thread's)
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

// the calling thread's callbacks on this mirror (NULL opts: it has none any more).  Takes the INDEX's lock, so it is never
// called under the cache lock: a long operation on one mirror must not hold up acquire / release of every other relation.
void rebind_retriever(lgpu::Index *ix, const usearch_init_options_t *opts)
{
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    ix->holder_bound = true;
    ix->opts.retriever = ix->opts.retriever_mut = nullptr;  // the RetrieverCtx the mirror was built with is not ours to keep
    ix->opts.retriever_ctx = nullptr;
    if(opts && (opts->retriever || opts->retriever_mut))
        ix->holders[ std::this_thread::get_id() ] = lgpu::Index::HolderBinding{ opts->retriever, opts->retriever_mut, opts->retriever_ctx };
    else
        ix->holders.erase(std::this_thread::get_id());
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
        lantern_mirror *hit = nullptr;
        {
            std::lock_guard<std::mutex> g(c.mu);
            for(auto &m : c.entries) {
                if(m.relation == relation && !m.stale && m.version == version) {
                    m.refs++;  // ours from here on: the entry cannot go away while the cache lock is dropped
                    m.last_use = ++c.clock;
                    c.hits++;
                    hit = &m;
                    break;
                }
            }
        }
        if(hit) {
            // The reference allocates its RetrieverCtx per scan and per insert and frees it at the end (scan.c:34,132,
            // insert.c:130,247): the callbacks and the ctx a mirror was BUILT with are gone by now.  The mirror takes
            // this holder's for as long as it holds it (usearch_add_external writes through retriever_mut).
            rebind_retriever((lgpu::Index *)hit->index, opts);
            return hit;
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
    // the mirror was built through this holder's callbacks: from now on they are looked up per holder (ours is the first)
    rebind_retriever((lgpu::Index *)ix, opts);
    lantern_mirror *theirs = nullptr;
    {
        std::lock_guard<std::mutex> g(c.mu);
        // somebody else may have built the same (relation, version) in the meantime: theirs stays, ours goes
        for(auto &m : c.entries) {
            if(m.relation == relation && !m.stale && m.version == version) {
                m.refs++;
                m.last_use = ++c.clock;
                c.hits++;
                theirs = &m;
                break;
            }
        }
        if(!theirs) {
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
    }
    usearch_error_t ignore = nullptr;
    usearch_free(ix, &ignore);
    rebind_retriever((lgpu::Index *)theirs->index, opts);
    return theirs;
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
    // The releasing holder's ctx is about to be freed (scan.c:132, insert.c:247): its binding goes first, while its reference
    // still keeps the index alive and before the cache lock is taken.  Only THIS holder's: another thread that holds the same
    // mirror keeps its own.  (Several handles held by one thread share that thread's binding: after releasing one of them an
    // insert through another needs lantern_mirror_rebind, and fails with a message -- never through a dangling pointer --
    // without it.)
    rebind_retriever((lgpu::Index *)m->index, nullptr);
    Cache &c = cache();
    std::lock_guard<std::mutex> g(c.mu);
    if(m->refs > 0) m->refs--;
    m->last_use = ++c.clock;
    if(m->refs == 0 && m->index) {
        // nobody holds the mirror: whatever bindings are left belong to holders that went away without a release of THIS handle
        // (a thread that exited while another handle of the mirror was still out) -- their RetrieverCtx is gone and their thread id
        // may be reused.  The index's lock is only TRIED: the invariant of rebind_retriever stands -- the cache lock never WAITS
        // for an index lock, so a long operation on one mirror (a service lane or an indexing thread working on the handle it
        // was given earlier) cannot stall acquire / release of every other relation.  A busy index keeps its stale bindings until
        // the next release that finds it idle; they are never dereferenced meanwhile (a binding is looked up by the CALLING
        // thread's id, and a caller without a live one gets an error message: index.hpp binding()).
        lgpu::Index *ix = (lgpu::Index *)m->index;
        std::unique_lock<std::mutex> gi(ix->mu, std::try_to_lock);
        if(gi.owns_lock()) ix->holders.clear();
    }
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
