// cache_model.c -- LRU replay of a search launch's memory-object trace through the MI355X cache hierarchy: eight per-XCD L2s
// (4 MiB each) in front of the memory-side Infinity Cache (256 MiB, shared).  A measurement tool of bench.py, not part of the
// search path: it turns the trace the instrumented walk records (lantern_gpu_search_row_trace: every row a query evaluates and every
// adjacency list it reads, in order) into
//     fabric bytes  = what the L2s miss            -- comparable with the rocprofv3 FETCH_SIZE counter of the same launch (the check)
//     DRAM bytes    = what the Infinity Cache misses -- the figure the part exposes no counter for (profiles/r03_counter_notes.md)
// so that bench.py can print roofline.frac_dram_model beside frac_fabric (counters) and frac_algorithmic.
//
// Model.  Objects, not lines: a row (row_bytes, e.g. 3072) or an adjacency list (list0_bytes / listu_bytes) is fetched whole and is
// present or absent as a whole; both caches are fully associative LRU by bytes; a miss fills both levels.  The queries of a launch
// run on `walkers` resident workgroups, workgroup w on XCD w % xcds (round-robin dispatch); the first `walkers` queries start at
// once, a workgroup that finishes takes the next query (the kernel's ticket); the walkers advance in lock step, one HOP per turn
// (a hop = one adjacency list + the rows of its unvisited neighbours: the walk's dependent unit).  Caches persist across the
// launches handed to one replay, so launch 2 sees what launch 1 left behind (steady state: bench.py rotates query batches).
// What the model leaves out: set conflicts, the L2s' and the Infinity Cache's real replacement and allocation policies, partial
// lines, timing differences between hops.  It is validated on traces whose answer is known (tests/test_cache_model.py: no reuse,
// a table that fits, an exact-LRU cross-check) and, in every bench run, by its fabric figure against the counters.
//
//   gcc -O2 -shared -fPIC -o libcache_model.so cache_model.c        (lantern_amd/build.py)
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct
{
    uint64_t l2_bytes_per_xcd;  // 4 MiB
    uint64_t mall_bytes;        // 256 MiB
    uint32_t xcds;              // 8
    uint32_t walkers;           // resident workgroups of the launch (grid size)
    uint32_t row_bytes, list0_bytes, listu_bytes;
    uint32_t reserved;
} cm_config;

typedef struct
{
    double accesses, access_bytes;    // every object asked for (== the algorithmic bytes of SURVEY 8d, without the query rows)
    double row_accesses, list_accesses;
    double l2_miss_bytes;             // fabric-side bytes
    double mall_miss_bytes;           // DRAM-side bytes
    double l2_hits, mall_hits;        // object counts
    double dropped_entries;           // trace entries beyond the per-query capacity (not replayed)
} cm_result;

// ---- a byte-capacity LRU over u32 keys: chained hash table + doubly linked recency list, all in index arrays ----
typedef struct
{
    uint64_t  cap_bytes, used_bytes;
    uint32_t  nbuckets, nnodes, free_head, lru_head, lru_tail;  // head = most recent
    uint32_t *bucket;                                           // [nbuckets] -> node or NIL
    uint32_t *key, *size, *chain, *prev, *next;                 // [nnodes]
} lru_t;
#define NIL 0xFFFFFFFFu

static uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

static int lru_init(lru_t *c, uint64_t cap_bytes, uint32_t min_object_bytes)
{
    memset(c, 0, sizeof(*c));
    c->cap_bytes = cap_bytes;
    uint64_t n = cap_bytes / (min_object_bytes ? min_object_bytes : 1) + 16;
    if(n > 0x7FFFFFF0ull) return -1;
    c->nnodes = (uint32_t)n;
    uint32_t nb = 1024;
    while(nb < c->nnodes * 2u && nb < 0x40000000u) nb <<= 1;
    c->nbuckets = nb;
    c->bucket = (uint32_t *)malloc((size_t)nb * 4);
    c->key = (uint32_t *)malloc((size_t)c->nnodes * 4);
    c->size = (uint32_t *)malloc((size_t)c->nnodes * 4);
    c->chain = (uint32_t *)malloc((size_t)c->nnodes * 4);
    c->prev = (uint32_t *)malloc((size_t)c->nnodes * 4);
    c->next = (uint32_t *)malloc((size_t)c->nnodes * 4);
    if(!c->bucket || !c->key || !c->size || !c->chain || !c->prev || !c->next) return -1;
    memset(c->bucket, 0xFF, (size_t)nb * 4);
    for(uint32_t i = 0; i < c->nnodes; ++i) c->chain[ i ] = i + 1 < c->nnodes ? i + 1 : NIL;  // the free list runs through `chain`
    c->free_head = 0;
    c->lru_head = c->lru_tail = NIL;
    return 0;
}

static void lru_free(lru_t *c)
{
    free(c->bucket); free(c->key); free(c->size); free(c->chain); free(c->prev); free(c->next);
    memset(c, 0, sizeof(*c));
}

static void lru_unlink(lru_t *c, uint32_t n)
{
    const uint32_t p = c->prev[ n ], x = c->next[ n ];
    if(p != NIL) c->next[ p ] = x; else c->lru_head = x;
    if(x != NIL) c->prev[ x ] = p; else c->lru_tail = p;
}

static void lru_push_front(lru_t *c, uint32_t n)
{
    c->prev[ n ] = NIL;
    c->next[ n ] = c->lru_head;
    if(c->lru_head != NIL) c->prev[ c->lru_head ] = n; else c->lru_tail = n;
    c->lru_head = n;
}

static void lru_evict_tail(lru_t *c)
{
    const uint32_t n = c->lru_tail;
    lru_unlink(c, n);
    uint32_t *link = &c->bucket[ hash32(c->key[ n ]) & (c->nbuckets - 1) ];
    while(*link != n) link = &c->chain[ *link ];
    *link = c->chain[ n ];
    c->used_bytes -= c->size[ n ];
    c->chain[ n ] = c->free_head;
    c->free_head = n;
}

// 1 = hit (refreshed), 0 = miss (inserted, evicting least recently used objects as needed)
static int lru_access(lru_t *c, uint32_t key, uint32_t bytes)
{
    const uint32_t b = hash32(key) & (c->nbuckets - 1);
    for(uint32_t n = c->bucket[ b ]; n != NIL; n = c->chain[ n ])
        if(c->key[ n ] == key) {
            if(c->lru_head != n) { lru_unlink(c, n); lru_push_front(c, n); }
            return 1;
        }
    if(bytes > c->cap_bytes) return 0;  // larger than the cache: streams through
    while(c->used_bytes + bytes > c->cap_bytes || c->free_head == NIL) lru_evict_tail(c);
    const uint32_t n = c->free_head;
    c->free_head = c->chain[ n ];
    c->key[ n ] = key;
    c->size[ n ] = bytes;
    c->chain[ n ] = c->bucket[ b ];
    c->bucket[ b ] = n;
    c->used_bytes += bytes;
    lru_push_front(c, n);
    return 0;
}

static inline uint32_t object_bytes(const cm_config *cfg, uint32_t entry)
{
    const uint32_t kind = entry >> 30;
    return kind == 2 ? cfg->list0_bytes : kind == 3 ? cfg->listu_bytes : cfg->row_bytes;
}

// Replays `launches` launches back to back through one set of caches.  traces[l]: [nq[l]][cap] entries, counts[l]: [nq[l]] (a count
// above cap: only cap entries exist).  out[l]: the figures of launch l.  Returns 0, or -1 on allocation failure / bad arguments.
int cache_model_replay(const cm_config *cfg, int launches, const uint32_t *const *traces, const uint32_t *const *counts, const uint32_t *nq,
                       uint32_t cap, cm_result *out)
{
    if(!cfg || launches <= 0 || !traces || !counts || !nq || !out || cfg->xcds == 0 || cfg->walkers == 0 || cfg->xcds > 64) return -1;
    uint32_t min_obj = cfg->row_bytes;
    if(cfg->list0_bytes && cfg->list0_bytes < min_obj) min_obj = cfg->list0_bytes;
    if(cfg->listu_bytes && cfg->listu_bytes < min_obj) min_obj = cfg->listu_bytes;
    if(min_obj == 0) return -1;
    lru_t l2[ 64 ], mall;
    int   rc = 0;
    memset(l2, 0, sizeof(l2));
    memset(&mall, 0, sizeof(mall));
    for(uint32_t x = 0; x < cfg->xcds; ++x) rc |= lru_init(&l2[ x ], cfg->l2_bytes_per_xcd, min_obj);
    rc |= lru_init(&mall, cfg->mall_bytes, min_obj);
    uint32_t *wq = (uint32_t *)malloc((size_t)cfg->walkers * 4), *wpos = (uint32_t *)malloc((size_t)cfg->walkers * 4);
    if(rc || !wq || !wpos) { rc = -1; goto done; }
    for(int l = 0; l < launches; ++l) {
        cm_result r;
        memset(&r, 0, sizeof(r));
        for(uint32_t q = 0; q < nq[ l ]; ++q)
            if(counts[ l ][ q ] > cap) r.dropped_entries += (double)(counts[ l ][ q ] - cap);
        uint32_t next_q = 0, active = 0;
        for(uint32_t w = 0; w < cfg->walkers; ++w) {
            if(next_q < nq[ l ]) { wq[ w ] = next_q++; wpos[ w ] = 0; ++active; }
            else wq[ w ] = NIL;
        }
        while(active) {
            for(uint32_t w = 0; w < cfg->walkers; ++w) {
                uint32_t q = wq[ w ];
                if(q == NIL) continue;
                const uint32_t *t = traces[ l ] + (size_t)q * cap;
                const uint32_t  n = counts[ l ][ q ] < cap ? counts[ l ][ q ] : cap;
                uint32_t        p = wpos[ w ], taken = 0;
                lru_t          *mine = &l2[ w % cfg->xcds ];
                // one hop: entries up to (not including) the next list marker, at least one entry
                while(p < n && !(taken && (t[ p ] >> 31))) {
                    const uint32_t e = t[ p++ ], bytes = object_bytes(cfg, e);
                    ++taken;
                    r.accesses += 1.0;
                    r.access_bytes += (double)bytes;
                    if(e >> 31) r.list_accesses += 1.0; else r.row_accesses += 1.0;
                    if(lru_access(mine, e, bytes)) { r.l2_hits += 1.0; continue; }
                    r.l2_miss_bytes += (double)bytes;
                    if(lru_access(&mall, e, bytes)) { r.mall_hits += 1.0; continue; }
                    r.mall_miss_bytes += (double)bytes;
                }
                wpos[ w ] = p;
                if(p >= n) {  // this walk is over: the workgroup takes the next ticket
                    if(next_q < nq[ l ]) { wq[ w ] = next_q++; wpos[ w ] = 0; }
                    else { wq[ w ] = NIL; --active; }
                }
            }
        }
        out[ l ] = r;
    }
done:
    for(uint32_t x = 0; x < cfg->xcds; ++x) lru_free(&l2[ x ]);
    lru_free(&mall);
    free(wq);
    free(wpos);
    return rc;
}
