// walk.hpp -- the HNSW graph walk as one workgroup's cooperative device functions.
//
// Follows usearch's search_for_one_ / search_to_find_in_base_ / search_to_insert_ / refine_ as
// Lantern reaches them through usearch_search_ef (lantern_hnsw/src/hnsw/scan.c:220-228) and
// usearch_add (lantern_hnsw/src/hnsw/build.c:128), re-shaped for a 64-lane machine:
//
//  * usearch keeps two structures per search: `next` (a heap of candidates) and `top` (the ef best
//    so far).  Under the total order (distance, slot) an element evicted from `top` can never be
//    expanded later, so a single sorted list with an "expanded" bit per entry is equivalent:
//    "pop the best candidate" = "first unexpanded entry"; the walk ends when there is none.
//  * all unvisited neighbours of the popped node are evaluated at once (one G-lane group per row,
//    several rows in flight per wave), then merged into the list.  The result equals inserting them one by
//    one: the list ends up holding the ef smallest of (old list U new), whatever the order of insertion.
//    Two forms of the list: in one wave's REGISTERS (search_level_reg: ef <= 128, the usual case; a hop's
//    bookkeeping split over two waves) and in LDS with a parallel rank-merge (search_level: any ef).
//  * the visited set is an open-addressing hash set in LDS (four-slot buckets) that spills to a bitmap in HBM
//    owned by the workgroup when three quarters full (visit_test_and_set / hop_is_new below).
//
// All functions must be called by every thread of the workgroup (they contain barriers).
#pragma once
#include <type_traits>

#include "device_common.hpp"

namespace lgpu {

// scalar slots in LDS
enum { S_POS = 0, S_NNEW, S_CNT, S_ANY, S_BAD, S_CUR, S_CURD, S_CHANGED, S_VISCNT, S_SPILL, S_QN2, S_NNEW0, S_NNEW1, S_ANY0, S_ANY1,
       S_FRONT = 16, S_WORST = 20,  // two u64 each (by hop parity): search_level_reg's hand-off from the list wave to the visit wave
       S_MASK = 24,                 // two u64 (by hop parity): walk_spec.hpp's "which neighbours of the hop were new"
       S_SCALARS = 28 };
// S_QN2: ||query||^2 as float bits (cosine metrics; set by the kernel before a walk: device_common.hpp "cached row norms")

struct WalkLds
{
    uint4    *q;        // the query row (chunks uint4)
    uint64_t *keys;     // current list   (ef_cap)
    uint64_t *keys2;    // merge target   (ef_cap)
    uint64_t *newkeys;  // keys of this hop's new neighbours (cap_max)
    uint64_t *sorted;   // the same, sorted                  (cap_max)
    uint32_t *newids;   // unvisited neighbour slots         (cap_max)
    int      *scal;     // S_* scalars
    uint32_t *vis;      // visited hash set (vis_slots entries, any multiple of 4; 0 = the HBM bitmap only)
    uint32_t  vis_slots;
    uint32_t *touched;  // diagnostic instantiations only (PROF): one bit per row of the index, set when ANY query of the launch
                        // evaluates the row (lantern_gpu_search_unique_rows); never read, and never set, elsewhere
    // diagnostic instantiations only (PROF; lantern_gpu_search_row_trace): THIS query's memory-object trace, in the order the walk
    // asks for them -- a row evaluation is the row's slot, an adjacency list read is the node's slot | TRACE_LIST0 (level 0) or
    // | TRACE_LISTU (an upper level).  trace_count is bumped atomically (the lanes of a hop append side by side); entries beyond
    // trace_cap are counted, not stored.  The input of the cache model behind bench.py's frac_dram_model.
    uint32_t *trace, *trace_count;
    uint32_t  trace_cap;
    // the UNDO LOG of the workgroup's HBM visited bitmap (VisUndo below): undo_cap entries behind the bitmap's words
    uint32_t *undo;
    uint32_t  undo_cap;
};
constexpr uint32_t TRACE_LIST0 = 0x80000000u, TRACE_LISTU = 0xC0000000u;

template <bool PROF> __device__ __forceinline__ void trace_append(const WalkLds &s, uint32_t entry)
{
    if constexpr(PROF) {
        if(s.trace) {
            const uint32_t p = atomicAdd(s.trace_count, 1u);
            if(p < s.trace_cap) s.trace[ p ] = entry;
        }
    } else {
        (void)s;
        (void)entry;
    }
}

// PROF: record that row `id` was evaluated by this launch (the union over the launch's queries = the rows the launch needs at
// least once from HBM: the cold-miss lower bound of its DRAM traffic)
template <bool PROF> __device__ __forceinline__ void mark_touched(const WalkLds &s, uint32_t id)
{
    if constexpr(PROF) {
        if(s.touched) atomicOr(&s.touched[ id >> 5 ], 1u << (id & 31));
        trace_append<PROF>(s, id);
    } else {
        (void)s;
        (void)id;
    }
}

// the first operand of a (query, row) evaluation: the query row in LDS -- or, for ADC over PQ codes, just the chunk index (the
// table sits at the start of LDS: device_common.hpp AdcQuery)
template <int METRIC> __device__ __forceinline__ auto walk_query(const WalkLds &s)
{
    if constexpr(METRIC >= M_ADC && METRIC < M_PQD) return AdcQuery{};
    else return (const uint4 *)s.q;
}

// Carve the workgroup's dynamic LDS.  Every offset stays 16-byte aligned.
__device__ __forceinline__ unsigned char *carve_walk(unsigned char *p, WalkLds &s, uint32_t chunks, uint32_t ef_cap, uint32_t cap_max,
                                                     uint32_t vis_slots = 0)
{
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    s.q = (uint4 *)p;            p += (size_t)chunks * 16;
    s.keys = (uint64_t *)p;      p += up16((size_t)ef_cap * 8);
    s.keys2 = (uint64_t *)p;     p += up16((size_t)ef_cap * 8);
    s.newkeys = (uint64_t *)p;   p += up16((size_t)cap_max * 8);
    s.sorted = (uint64_t *)p;    p += up16((size_t)cap_max * 8);
    s.newids = (uint32_t *)p;    p += up16((size_t)cap_max * 4);
    s.scal = (int *)p;           p += S_SCALARS * 4;
    s.vis = (uint32_t *)p;       p += (size_t)vis_slots * 4;
    s.vis_slots = vis_slots;
    return p;
}
__host__ inline size_t walk_lds_bytes(uint32_t chunks, uint32_t ef_cap, uint32_t cap_max, uint32_t vis_slots = 0)
{
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    return (size_t)chunks * 16 + 2 * up16((size_t)ef_cap * 8) + 2 * up16((size_t)cap_max * 8) + up16((size_t)cap_max * 4) + S_SCALARS * 4 +
           (size_t)vis_slots * 4;
}

// ---- visited set ------------------------------------------------------------------------------------------
// usearch keeps a growing hash set of visited slots per search.  Here: an open-addressing hash set in LDS
// (no HBM round trip per hop, nothing to clear in HBM per query) that SPILLS to the workgroup's HBM bitmap once it
// is three quarters full: from then on new slots are recorded in the bitmap (all-zero whenever a walk starts: VisUndo below) and a
// lookup consults both.  With vis_slots == 0 only the bitmap is used.
// multiply-shift onto [0, slots): any table size, so the set can be sized to the LDS a given occupancy leaves
__device__ __forceinline__ uint32_t vis_hash(uint32_t x, uint32_t buckets) { return __umulhi(x * 0x9E3779B1u, buckets); }

// The LDS visited set: open addressing over BUCKETS of four slots (one ds_read_b128 looks at a whole bucket).  A slot id lives
// in the first bucket, counting from its home, that had a free slot when it arrived; buckets never lose entries, so a lookup
// ends at the first bucket that holds the id or still has a free slot.  Insert-only, exact; used by the lanes of wave 0.
struct VisProbe
{
    bool found;  // the id is in the bucket
    int  e;      // a free slot of the bucket (-1: none), searched from slot `rot` on so that lanes spread over the free slots
};
__device__ __forceinline__ VisProbe vis_look(const WalkLds &s, uint32_t bucket, uint32_t x, int rot)
{
    asm volatile("" ::: "memory");  // the bucket is re-read after every CAS
    const uint4    w = ((const uint4 *)s.vis)[ bucket ];
    const uint32_t v[ 4 ] = { w.x, w.y, w.z, w.w };
    VisProbe       r;
    r.found = w.x == x || w.y == x || w.z == x || w.w == x;
    r.e = -1;
#pragma unroll
    for(int i = 3; i >= 0; --i) {
        const int j = (rot + i) & 3;
        if(v[ j ] == EMPTY) r.e = j;
    }
    return r;
}

// ---- the HBM bitmap is kept ALL-ZERO between walks, by undoing instead of clearing [r6] ------------------------------------
// Until round 5 a walk whose LDS set filled up CLEARED its workgroup's whole bitmap before the first bit went in: cap / 8 bytes of
// writes per spilling walk -- 1.25 MB at 10M rows, where at ef = 128 most walks spill: 7.8 GB of writes per 8192-query launch, 7 % of
// the launch's fabric traffic (profiles/r06_bench_line_10Mx768_ef128.json), and 10 % of a 10M-row build's.  Now the bitmap is zero when
// a walk starts (zeroed when it is allocated; every walk leaves it as it found it): the ids a walk records in it are also appended
// to an undo log behind the bitmap, and when the walk ends the visit wave zeroes the words the log names (a word's other bits are
// the same walk's).  A walk that overflows the log clears the whole bitmap at its end, as a spilling walk used to at its spill.
struct VisUndo
{
    uint32_t cnt = 0;      // entries in the log (wave-uniform in the visit wave)
    bool     over = false; // more ids were recorded than the log holds: clear everything at the end
};
// after a filter pass that recorded ids in the BITMAP: `mine` = this lane recorded `id`; newmask = __ballot(mine).  Visit wave only.
__device__ __forceinline__ void undo_record(const WalkLds &s, VisUndo &u, bool mine, uint32_t id, unsigned long long newmask, int lane)
{
    const uint32_t n = (uint32_t)__popcll(newmask);
    if(n == 0 || u.over) return;
    if(u.cnt + n <= s.undo_cap) {
        if(mine) __hip_atomic_store(&s.undo[ u.cnt + (uint32_t)__popcll(newmask & ((1ull << lane) - 1ull)) ], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u.cnt += n;
    } else {
        u.over = true;
    }
}
// at the end of a walk, by the visit wave: the bitmap goes back to all-zero
__device__ __forceinline__ void undo_apply(const WalkLds &s, uint32_t *bitmap, uint32_t bm_words, const VisUndo &u, int lane)
{
    if(u.over) {
        uint4 *b4 = (uint4 *)bitmap;
        for(uint32_t i = (uint32_t)lane; i < bm_words / 4; i += 64) b4[ i ] = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the clears reach L2 before the next walk's atomics on these words
    } else if(u.cnt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the log's own stores have left
        for(uint32_t i = (uint32_t)lane; i < u.cnt; i += 64) {
            const uint32_t id = __hip_atomic_load(&s.undo[ i ], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)atomicAnd(&bitmap[ id >> 5 ], 0u);  // (an atomic, like the atomicOr's that set the bits: same path to L2, in order)
        }
    }
}

// true if `x` was already visited; otherwise records it.  Called by the lanes of wave 0 only.
__device__ __forceinline__ bool visit_test_and_set(WalkLds &s, uint32_t *bitmap, uint32_t x, bool spilled)
{
    if(s.vis_slots) {
        const uint32_t nb4 = s.vis_slots >> 2;
        uint32_t       b = vis_hash(x, nb4);
        for(;;) {
            const VisProbe p = vis_look(s, b, x, (int)(threadIdx.x & 3));
            if(p.found) return true;
            if(p.e < 0) {  // a full bucket: the id may live further on
                b = b + 1 == nb4 ? 0u : b + 1;
                continue;
            }
            if(spilled) break;  // not in the LDS set (read-only once spilled): the bitmap decides
            const uint32_t cur = atomicCAS(&s.vis[ 4 * b + (uint32_t)p.e ], EMPTY, x);
            if(cur == EMPTY) return false;  // recorded; the caller counts the slots it added (one ballot per pass)
            if(cur == x) return true;
            // another lane took that slot in the meantime: look at the bucket again
        }
    }
    const uint32_t bit = 1u << (x & 31);
    return (atomicOr(&bitmap[ x >> 5 ], bit) & bit) != 0;
}

// Was `x` visited?  Nothing is recorded (the two-nodes-per-round walk tests the neighbours of the node it expands speculatively
// and records them only once the speculation has come true: walk_twin.hpp).
__device__ __forceinline__ bool visit_test(const WalkLds &s, uint32_t *bitmap, uint32_t x, bool spilled)
{
    if(s.vis_slots) {
        const uint32_t nb4 = s.vis_slots >> 2;
        uint32_t       b = vis_hash(x, nb4);
        for(;;) {
            const VisProbe p = vis_look(s, b, x, 0);
            if(p.found) return true;
            if(p.e >= 0) break;  // a bucket with a free slot ends the chain: not in the LDS set
            b = b + 1 == nb4 ? 0u : b + 1;
        }
        if(!spilled) return false;
    }
    return ((atomicOr(&bitmap[ x >> 5 ], 0u) >> (x & 31)) & 1u) != 0;  // (an atomic read: the bits are set by atomics, past the CU's L1)
}

// One lane's share of a hop's visited filter: is neighbour `nb` (EMPTY = no neighbour) new?  The common case is straight-line
// code for the whole wave -- one bucket read, one CAS -- and only lanes that met a full bucket or lost a slot to another lane
// take the probing loop (its divergent control flow costs more scalar instructions than the LDS accesses themselves).
__device__ __forceinline__ bool hop_is_new(WalkLds &s, uint32_t *bitmap, uint32_t nb, bool spilled)
{
    bool isnew = false;
    if(s.vis_slots && !spilled) {
        const bool     valid = nb != EMPTY;
        const uint32_t b = vis_hash(valid ? nb : 0u, s.vis_slots >> 2);
        const VisProbe p = vis_look(s, b, nb, (int)(threadIdx.x & 3));
        const bool     attempt = valid && !p.found && p.e >= 0;
        uint32_t       cur = nb;
        if(attempt) cur = atomicCAS(&s.vis[ 4 * b + (uint32_t)(p.e & 3) ], EMPTY, nb);
        isnew = attempt && cur == EMPTY;
        const bool unsettled = valid && !p.found && !isnew && !(attempt && cur == nb);
        if(__ballot(unsettled) != 0ull) {
            if(unsettled) isnew = !visit_test_and_set(s, bitmap, nb, false);
        }
    } else if(nb != EMPTY) {
        isnew = !visit_test_and_set(s, bitmap, nb, spilled);
    }
    return isnew;
}

__device__ __forceinline__ int lower_bound_keys(const uint64_t *a, int n, uint64_t k)
{
    int lo = 0, hi = n;
    while(lo < hi) {
        int mid = (lo + hi) >> 1;
        if(a[ mid ] < k) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- merge helpers: four adjacent lanes (a quad) share the LDS reads of one key's position --------------------------
// sum over the quad (every lane of the quad gets it): two DPP adds
template <int L> __device__ __forceinline__ int quad_sum(int x)  // L = 2 or 4 cooperating lanes
{
    uint32_t v = (uint32_t)x;
    v = v + dpp_take<0xB1, 0xF>(v);
    if(L == 4) v = v + dpp_take<0x4E, 0xF>(v);
    return (int)v;
}
// #{j < n : a[j] < k} with the reads split over the quad (sub = lane & 3): 16-byte reads of two keys, all of a lane's reads
// independent of one another (one LDS round trip per eight keys of the lane's share)
template <int L> __device__ __forceinline__ int quad_count_below(const uint64_t *a, int n, uint64_t k, int sub)
{
    const uint4 *a4 = (const uint4 *)a;  // a is 16-byte aligned (carve_walk)
    const int    pairs = n >> 1;
    int          c = 0;
    for(int p = sub; p < pairs; p += L) {
        const uint4    w = a4[ p ];
        const uint64_t k0 = ((uint64_t)w.y << 32) | w.x, k1 = ((uint64_t)w.w << 32) | w.z;
        c += (k0 < k) + (k1 < k);
    }
    if((n & 1) && sub == 0) c += a[ n - 1 ] < k;
    return quad_sum<L>(c);
}
// lower_bound over the sorted a[0..n) as two rounds of independent reads instead of log2(n) dependent ones: first the last
// key of every block of eight (how many whole blocks lie below k), then the eight keys of the block k falls into
template <int L> __device__ __forceinline__ int quad_lower_bound(const uint64_t *a, int n, uint64_t k, int sub)
{
    const int nb = (n + 7) >> 3;
    int       c = 0;
    for(int b = sub; b < nb; b += L) {
        const int last = 8 * b + 7 < n ? 8 * b + 7 : n - 1;
        c += a[ last ] < k;
    }
    const int blk = quad_sum<L>(c);
    if(blk >= nb) return n;
    const int lo = 8 * blk, len = n - lo < 8 ? n - lo : 8;
    int       d = 0;
    for(int j = sub; j < len; j += L) d += a[ lo + j ] < k;
    return lo + quad_sum<L>(d);
}

// ---- search_for_one_: greedy descent over levels (begin, end] ---------------------------------------
// Returns the closest slot (same value in every thread).  D counts distance evaluations.
template <int METRIC, int G, bool PROF = false>
__device__ uint32_t greedy_descent(const View &v, WalkLds &s, uint32_t start, int begin_level, int end_level, uint32_t &D)
{
    const int tid = threadIdx.x, T = blockDim.x, g = tid / G, gl = tid % G, NG = T / G;
    float *newd = (float *)s.newkeys;  // reuse: one f32 per neighbour
    const float qn2 = __int_as_float(s.scal[ S_QN2 ]);
    if(g == 0) {
        float d = group_dist_n<METRIC, G>(walk_query<METRIC>(s), row_of_m<METRIC>(v, start), (int)v.chunks, gl, qn2, row_norm<METRIC>(v, start));
        if(gl == G - 1) { s.scal[ S_CUR ] = (int)start; s.scal[ S_CURD ] = __float_as_int(d); mark_touched<PROF>(s, start); }
    }
    D += 1;
    __syncthreads();
    for(int level = begin_level; level > end_level; --level) {
        for(;;) {
            uint32_t        cur = (uint32_t)s.scal[ S_CUR ];
            uint32_t        cap;
            const uint32_t *list = neighbors_of(v, cur, level, cap);
            // gather the (EMPTY-terminated, hole-free) list: wave 0, one ballot per 64 slots (an LDS atomicMax here is
            // serialised by the compiler into a scalar loop over the lanes)
            if(tid < 64) {
                if(tid == 0) trace_append<PROF>(s, cur | (level > 0 ? TRACE_LISTU : TRACE_LIST0));
                int count = 0;
                for(uint32_t off = 0; off < cap; off += 64) {
                    const uint32_t i = off + (uint32_t)(tid & 63);
                    const uint32_t nb = i < cap ? list[ i ] : EMPTY;
                    if(i < cap) s.newids[ i ] = nb;
                    count += (int)__popcll(__ballot(nb != EMPTY));
                }
                if(tid == 0) s.scal[ S_NNEW ] = count;
            }
            __syncthreads();
            const int nn = s.scal[ S_NNEW ];
            for(int i = g; i < nn; i += NG) {
                const uint32_t id = s.newids[ i ];
                float d = group_dist_n<METRIC, G>(walk_query<METRIC>(s), row_of_m<METRIC>(v, id), (int)v.chunks, gl, qn2, row_norm<METRIC>(v, id));
                if(gl == G - 1) { newd[ i ] = d; mark_touched<PROF>(s, id); }
            }
            D += (uint32_t)nn;
            __syncthreads();
            if(tid == 0) {
                // the sequential scan of search_for_one_: first strictly-closer wins, in list order
                float    best = __int_as_float(s.scal[ S_CURD ]);
                uint32_t bslot = cur;
                int      changed = 0;
                for(int i = 0; i < nn; ++i)
                    if(newd[ i ] < best) { best = newd[ i ]; bslot = s.newids[ i ]; changed = 1; }
                s.scal[ S_CUR ] = (int)bslot;
                s.scal[ S_CURD ] = __float_as_int(best);
                s.scal[ S_CHANGED ] = changed;
            }
            __syncthreads();
            if(!s.scal[ S_CHANGED ]) break;
        }
    }
    uint32_t r = (uint32_t)s.scal[ S_CUR ];
    __syncthreads();
    return r;
}

// ---- search_to_find_in_base_ / search_to_insert_ -----------------------------------------------------
// On return s.keys[0..cnt) holds the result ascending by (distance, slot); returns cnt.
//
// One hop = three phases, three barriers:
//   (1) wave 0 alone, no barrier inside: pop (first unexpanded entry: one ballot per 64 entries), the popped node's
//       neighbour list (ONE 128-byte request), the visited test-and-set (LDS hash set, spilling to the HBM bitmap), the
//       ballot compaction of the new ids into LDS.  The other waves wait at the barrier.
//   (2) all waves: one G-lane group per row, two rows in flight per group -> keys.
//   (3) all waves: rank-merge into the ef-bounded list -- a thread per key, every position computed from the UNSORTED new
//       keys (old key i goes to i + #{new < it}, new key t to #{new < it} + lower_bound(old, it)), so there is no
//       intermediate sort and no barrier inside the phase.
// Round 1 took six barriers per hop, found the pop position with an LDS atomicMin (which the compiler serialises into a
// scalar loop over the active lanes: ~1.4 us per hop on its own) and sorted the new keys before merging; measured with
// the instrumented kernel (scripts/profile_hop_phases.py) the three serial phases cost 3 000 - 4 000 cycles EACH, more
// than the row evaluation of a 128-d hop.  The scalars wave 0 publishes are double-buffered by hop parity, so a wave that
// is slow to read them is never overtaken by the next hop's values.
// PROF (diagnostic instantiations only): thread 0 accumulates shader-clock cycles per phase into prof[0..6): [4] pop, [5] the
// neighbour list's arrival, [0] visited filter + compaction (together: wave 0's section) | [1] wait at the first barrier |
// [2] distances | [3] merge.
// One hop's distance phase: keys of the nnew unvisited neighbours in s.newids -> s.newkeys.  One G-lane group per row, ROWS rows in
// flight per group.  ANY: also raise *any_slot when a key beats `worst` (the LDS-list walk skips its merge otherwise).
template <int METRIC, int G, int ROWS, bool ANY>
__device__ __forceinline__ void hop_distances(const View &v, WalkLds &s, int nnew, float qn2, uint64_t worst, int *any_slot)
{
    const int tid = threadIdx.x, T = blockDim.x, g = tid / G, gl = tid % G, NG = T / G;
    if constexpr(ROWS > 2) {
        for(int i = g; i < nnew; i += ROWS * NG) {
            decltype(row_of_m<METRIC>(v, 0u)) rows[ ROWS ];
            uint32_t     ids[ ROWS ];
            float        n2[ ROWS ], d[ ROWS ];
#pragma unroll
            for(int r = 0; r < ROWS; ++r) {
                const int j = i + r * NG;
                ids[ r ] = s.newids[ j < nnew ? j : i ];
                rows[ r ] = row_of_m<METRIC>(v, ids[ r ]);
                n2[ r ] = row_norm<METRIC>(v, ids[ r ]);
            }
            group_distR_n<METRIC, G, ROWS>(walk_query<METRIC>(s), rows, (int)v.chunks, gl, qn2, n2, d);
            if(gl == G - 1) {
                bool any = false;
#pragma unroll
                for(int r = 0; r < ROWS; ++r) {
                    const int j = i + r * NG;
                    if(j < nnew) {
                        const uint64_t k = make_key(d[ r ], ids[ r ]);
                        s.newkeys[ j ] = k;
                        any |= k < worst;
                    }
                }
                if(ANY && any) *any_slot = 1;
            }
        }
    } else
    for(int i = g; i < nnew; i += 2 * NG) {
        const int      j = i + NG;
        const uint32_t id0 = s.newids[ i ];
        const uint32_t id1 = j < nnew ? s.newids[ j ] : id0;
        float          d0, d1;
        group_dist2_n<METRIC, G>(walk_query<METRIC>(s), row_of_m<METRIC>(v, id0), row_of_m<METRIC>(v, id1), (int)v.chunks, gl, qn2, row_norm<METRIC>(v, id0),
                                 row_norm<METRIC>(v, id1), d0, d1);
        if(gl == G - 1) {
            uint64_t k0 = make_key(d0, id0);
            s.newkeys[ i ] = k0;
            bool any = k0 < worst;
            if(j < nnew) {
                uint64_t k1 = make_key(d1, id1);
                s.newkeys[ j ] = k1;
                any |= k1 < worst;
            }
            if(ANY && any) *any_slot = 1;
        }
    }
}

template <int METRIC, int G, bool PROF = false, int ROWS = 2>
__device__ int search_level(const View &v, WalkLds &s, uint32_t *bitmap, uint32_t bm_words, uint32_t start, int level, int ef,
                            uint32_t &D, uint32_t &E, unsigned long long *prof = nullptr)
{
    const int tid = threadIdx.x, T = blockDim.x, g = tid / G, gl = tid % G;
    const int lane = tid & 63;
    // "this is wave 0" as a value the compiler KNOWS is wave-uniform: the serial section below then runs under uniform control
    // flow (no exec save/restore around it, its counters and loop state in scalar registers)
    const bool wave0 = __builtin_amdgcn_readfirstlane(tid) < 64;
    unsigned long long tl = 0;
    if constexpr(PROF) tl = (unsigned long long)clock64();
#define LGPU_MARK(i)                                                  \
    if constexpr(PROF) {                                              \
        if(tid == 0) {                                                \
            const unsigned long long t_ = (unsigned long long)clock64(); \
            prof[ i ] += t_ - tl;                                     \
            tl = t_;                                                  \
        }                                                             \
    }
    // visits.clear(): the LDS set; the HBM bitmap is all-zero between walks (VisUndo)
    for(uint32_t i = tid; i < s.vis_slots; i += T) s.vis[ i ] = EMPTY;
    const float qn2 = __int_as_float(s.scal[ S_QN2 ]);
    if(g == 0) {
        float d = group_dist_n<METRIC, G>(walk_query<METRIC>(s), row_of_m<METRIC>(v, start), (int)v.chunks, gl, qn2, row_norm<METRIC>(v, start));
        if(gl == G - 1) { s.keys[ 0 ] = make_key(d, start); mark_touched<PROF>(s, start); }
    }
    D += 1;
    __syncthreads();
    // wave 0's private walk state: how many slots the LDS set holds, whether it has spilled to the bitmap, the bitmap's undo log
    uint32_t viscnt = 0;
    bool     spilled = false;
    VisUndo  undo;
    if(tid == 0) {
        (void)visit_test_and_set(s, bitmap, start, false);
        viscnt = s.vis_slots ? 1u : 0u;
    }
    viscnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)viscnt);  // uniform in wave 0 (tid 0 is its first lane); unused elsewhere
    if(wave0 && !s.vis_slots) undo_record(s, undo, lane == 0, start, 1ull, lane);  // bitmap-only mode: the start node's bit
    int      cnt = 1;
    for(int hop = 0;; ++hop) {
        int *const nnew_slot = &s.scal[ (hop & 1) ? S_NNEW1 : S_NNEW0 ];
        int *const any_slot = &s.scal[ (hop & 1) ? S_ANY1 : S_ANY0 ];
        // ---- (1) wave 0: pop + neighbour list + visited filter
        if(wave0) {
            int      pos = -1;
            uint32_t node = EMPTY;
            bool     mine = false;  // this lane holds the popped key
            uint64_t popped = 0;
            for(int base = 0; base < cnt; base += 64) {  // first unexpanded entry = pop of usearch's `next` heap
                const int                i = base + lane;
                const uint64_t           key = i < cnt ? s.keys[ i ] : 1ull;
                const unsigned long long m = __ballot(!key_expanded(key));
                if(m) {
                    const int first = (int)__builtin_ctzll(m);
                    pos = base + first;
                    const uint32_t lo = (uint32_t)key;  // slot : flag, the low word of the key
                    node = (uint32_t)__builtin_amdgcn_readlane((int)lo, first) >> 1;
                    mine = lane == first;
                    popped = key;
                    break;
                }
            }
            if(pos < 0) {
                if(lane == 0) *nnew_slot = -1;  // the walk is over
            } else {
                E += 1;
                LGPU_MARK(4)
                // the LDS set must keep room for one full neighbour list; otherwise spill to the HBM bitmap (the set holds
                // 3/4 * vis_slots slots, a search visits D of them).  The bitmap is all-zero (VisUndo): nothing to clear.
                if(s.vis_slots && !spilled && viscnt + v.M0 > s.vis_slots / 4 * 3) spilled = true;
                uint32_t        cap;
                const uint32_t *list = neighbors_of(v, node, level, cap);
                if(lane == 0) trace_append<PROF>(s, node | (level > 0 ? TRACE_LISTU : TRACE_LIST0));
                int nb_new = 0;
                for(uint32_t off = 0; off < cap; off += 64) {
                    const uint32_t i = off + (uint32_t)lane;
                    const uint32_t nb = i < cap ? list[ i ] : EMPTY;
                    if constexpr(PROF) {  // make the list's arrival visible to the phase clock
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        LGPU_MARK(5)
                    }
                    const bool isnew = hop_is_new(s, bitmap, nb, spilled);
                    const unsigned long long m = __ballot(isnew);
                    if(isnew) {
                        s.newids[ nb_new + __popcll(m & ((1ull << lane) - 1ull)) ] = nb;
                        mark_touched<PROF>(s, nb);
                    }
                    if(spilled || !s.vis_slots) undo_record(s, undo, isnew, nb, m, lane);  // these ids went into the HBM bitmap
                    nb_new += __popcll(m);
                }
                if(s.vis_slots && !spilled) viscnt += (uint32_t)nb_new;
                if(lane == 0) { *nnew_slot = nb_new; *any_slot = 0; }
                if(mine) s.keys[ pos ] = popped | 1ull;  // mark it expanded: the lane that read the key writes it back (no LDS read-modify-write)
            }
        }
        LGPU_MARK(0)
        __syncthreads();
        const int nnew = *nnew_slot;
        LGPU_MARK(1)
        if(nnew < 0) break;
        if(nnew == 0) continue;
        // ---- (2) distances: one G-lane group per row, two rows in flight per group (ROWS = 4: the small-batch shape)
        const uint64_t worst = cnt == ef ? s.keys[ cnt - 1 ] : ~0ull;
        hop_distances<METRIC, G, ROWS, true>(v, s, nnew, qn2, worst, any_slot);
        D += (uint32_t)nnew;
        __syncthreads();
        LGPU_MARK(2)
        if(!*any_slot) continue;  // nothing beats the current radius: list unchanged
        // ---- (3) merge: every key's position in the merged list straight from the unsorted new keys.  Two or four
        // adjacent lanes per key (four when the workgroup has the threads for one pass over ef + 2M keys): they split the
        // LDS reads -- all independent: no binary-search chain -- and sum with DPP adds.
        {
            const int total = cnt + nnew;
            auto merge_pass = [&](auto lanes) {
                constexpr int L = decltype(lanes)::value;
                const int     sub = tid & (L - 1);
                for(int t = tid / L; t < ((total + 63) & ~63); t += T / L) {
                    const bool     live = t < total, is_new = t >= cnt;
                    const uint64_t k = !live ? 0ull : is_new ? s.newkeys[ t - cnt ] : s.keys[ t ];
                    const int      below = quad_count_below<L>(s.newkeys, nnew, k, sub);  // new keys smaller than k (keys are distinct)
                    int            p = t + below;
                    if(is_new) p = below + quad_lower_bound<L>(s.keys, cnt, k, sub);
                    if(live && sub == 0 && p < ef) s.keys2[ p ] = k;
                }
            };
            if(T >= 4 * total) merge_pass(std::integral_constant<int, 4>{});
            else merge_pass(std::integral_constant<int, 2>{});
        }
        cnt = cnt + nnew < ef ? cnt + nnew : ef;
        uint64_t *tmp = s.keys; s.keys = s.keys2; s.keys2 = tmp;
        __syncthreads();
        LGPU_MARK(3)
    }
#undef LGPU_MARK
    if(wave0) undo_apply(s, bitmap, bm_words, undo, lane);  // the workgroup's HBM bitmap goes back to all-zero
    return cnt;
}

// ---------------------------------------------------------------------------------------------------
// The same walk with the candidate list in WAVE 0's REGISTERS (ef <= 64 * KPL): lane l of register r holds the (64 r + l)-th
// smallest key, ~0 past the end.  A hop's serial bookkeeping -- merge the previous hop's keys, pop, neighbour list, visited
// filter -- is then one barrier-free section of wave 0:
//   * pop     = one ballot over the expanded flags + one readlane,
//   * merge   = for every new key that beats the radius (few, once the list has settled): its rank is one ballot + popcount,
//               the insertion one wave-wide DPP shift (wave_shr:1) -- no LDS traffic, no second pass over the list,
// and a hop costs two barriers (after the section, after the distances) instead of three.  Inserting the keys one at a time into
// an ef-bounded sorted list leaves the ef smallest of (list U new keys), flags intact: the same list the merge of search_level
// produces, so results are identical (tests/test_gpu_parity.py runs both).
__device__ __forceinline__ uint64_t wave_shr1(uint64_t x)  // lane l <- lane l-1 (lane 0 keeps its own value)
{
    const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)x, (int)(uint32_t)x, 0x138, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(x >> 32), (int)(uint32_t)(x >> 32), 0x138, 0xF, 0xF, false);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t readlane64(uint64_t x, int l)  // l uniform
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

#define LGPU_MARK(i)                                                  \
    if constexpr(PROF) {                                              \
        if(tid == 0) {                                                \
            const unsigned long long t_ = (unsigned long long)clock64(); \
            prof[ i ] += t_ - tl;                                     \
            tl = t_;                                                  \
        }                                                             \
    }
template <int METRIC, int G, int KPL, bool PROF = false, int ROWS = 2>
__device__ int search_level_reg(const View &v, WalkLds &s, uint32_t *bitmap, uint32_t bm_words, uint32_t start, int level, int ef,
                                uint32_t &D, uint32_t &E, unsigned long long *prof = nullptr)
{
    const int tid = threadIdx.x, T = blockDim.x, g = tid / G, gl = tid % G;
    const int lane = tid & 63;
    // Wave roles, as values the compiler KNOWS are wave-uniform (the serial sections then run under uniform control flow: no
    // exec save/restore, counters and loop state in scalar registers).  With two or more waves the hop's bookkeeping is SPLIT:
    //   visit wave (wave 0): which node to expand -> its neighbour list -> visited filter -> compaction of the new ids;
    //   list wave  (wave 1): merge of the previous hop's keys into the register list, pop + "expanded" mark, and the two
    //                        values the visit wave's NEXT decision needs: the list's first unexpanded key (`front`) and its
    //                        radius (`worst`), published in LDS slots double-buffered by hop parity.
    // The visit wave does not wait for the merge: the node to expand is min(front, smallest new key inside the radius) -- the
    // first unexpanded entry of the merged list (a key the merge truncates away is never that minimum) -- which it finds with a
    // few ballots over the unsorted new keys.  Merge and list fetch + visited filter, the two long serial chains of a hop, run
    // side by side.  A one-wave workgroup does the list role, then the visit role.
    const int  wv = __builtin_amdgcn_readfirstlane(tid) >> 6;
    const bool split = T >= 128;
    const bool visit_wave = wv == 0, list_wave = split ? wv == 1 : wv == 0;  // (which second wave makes no difference: measured)
    // [r6] Speculative fetch of the FRONT's neighbour list.  The node a hop expands is min(front, best new key of the previous hop); the
    // front is published before the previous hop's distances are even requested, and it IS the next node in 88 % of the hops on
    // clustered data (49 % on i.i.d. Gaussian: scripts/experiments/front_hit_rate.py).  So the visit wave requests the front's list
    // (one 128-byte line at M = 16) right before the distance phase and finds it in a register one hop later; a wrong guess costs the
    // line.  Level 0 only (an upper list's address needs a load of its own), lists of at most 64 entries, never in the instrumented
    // walk (its trace is the logical access sequence).  LGPU_LIST_PREFETCH: 0 off, 1 rows of fewer than 128 chunks (G < 64), 2 all.
    constexpr bool PF = !PROF && (LGPU_LIST_PREFETCH >= 2 || (LGPU_LIST_PREFETCH == 1 && G < 64));
    const bool     pf_on = PF && split && level == 0 && v.M0 <= 64;
    uint32_t       pf_node = EMPTY, pf_nb = EMPTY;
    unsigned long long tl = 0;
    if constexpr(PROF) tl = (unsigned long long)clock64();
    for(uint32_t i = tid; i < s.vis_slots; i += T) s.vis[ i ] = EMPTY;  // (the HBM bitmap is all-zero between walks: VisUndo)
    const float qn2 = __int_as_float(s.scal[ S_QN2 ]);
    if(g == 0) {
        float d = group_dist_n<METRIC, G>(walk_query<METRIC>(s), row_of_m<METRIC>(v, start), (int)v.chunks, gl, qn2, row_norm<METRIC>(v, start));
        if(gl == G - 1) { s.newkeys[ 0 ] = make_key(d, start); mark_touched<PROF>(s, start); }
    }
    D += 1;
    __syncthreads();
    uint64_t *const front_pub = (uint64_t *)&s.scal[ S_FRONT ];  // [2] by hop parity
    uint64_t *const worst_pub = (uint64_t *)&s.scal[ S_WORST ];  // [2]
    // visit wave's private state: how many slots the LDS set holds, whether it has spilled to the bitmap, the bitmap's undo log
    uint32_t viscnt = 0;
    bool     spilled = false;
    VisUndo  undo;
    if(tid == 0) {
        (void)visit_test_and_set(s, bitmap, start, false);
        viscnt = s.vis_slots ? 1u : 0u;
    }
    viscnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)viscnt);
    if(visit_wave && !s.vis_slots) undo_record(s, undo, lane == 0, start, 1ull, lane);  // bitmap-only mode: the start node's bit
    // list wave's private state
    uint64_t           K[ KPL ];     // the list; lanes past position ef - 1 hold leftovers of the shifts and are masked out by
    unsigned long long live[ KPL ];  // `live`: the lanes of register r whose position 64 r + lane is below ef
#pragma unroll
    for(int r = 0; r < KPL; ++r) {
        K[ r ] = ~0ull;
        const int m = ef - 64 * r;
        live[ r ] = m >= 64 ? ~0ull : m <= 0 ? 0ull : (1ull << m) - 1ull;
    }
    int cnt = 1, pend = 0;
    if(list_wave) {
        const uint64_t k0 = s.newkeys[ 0 ];
        if(lane == 0) {
            K[ 0 ] = k0;
            front_pub[ 0 ] = k0;
            worst_pub[ 0 ] = ef == 1 ? k0 : ~0ull;
        }
    }
    if(split) __syncthreads();
    for(int hop = 0;; ++hop) {
        const int  par = hop & 1;
        int *const nnew_slot = &s.scal[ par ? S_NNEW1 : S_NNEW0 ];
        uint32_t   node = EMPTY;  // one-wave form: the list role hands the popped node to the visit role
        unsigned long long tw = 0;
        if constexpr(PROF) tw = (unsigned long long)clock64();
        if(list_wave) {
            // ---- merge the previous hop's keys: one at a time into the sorted registers (rank = one ballot, insertion = one
            // wave-wide DPP shift); only keys inside the radius, ~4 of a hop's ~30 once the list has settled
            for(int base = 0; base < pend; base += 64) {
                const uint64_t N = base + lane < pend ? s.newkeys[ base + lane ] : ~0ull;
                uint64_t       worst = ~0ull;
                if(cnt == ef) {
                    const int wl = (ef - 1) & 63;
#pragma unroll
                    for(int r = 0; r < KPL; ++r)
                        if(r == (ef - 1) >> 6) worst = readlane64(K[ r ], wl);
                }
                unsigned long long todo = __ballot(N < worst);
                while(todo) {
                    const int t = (int)__builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    const uint64_t k = readlane64(N, t);
                    int            p = 0;
#pragma unroll
                    for(int r = 0; r < KPL; ++r) p += (int)__popcll(__ballot(K[ r ] < k) & live[ r ]);
                    if(p >= ef) continue;  // the radius moved in since `todo` was taken
                    const int r0 = p >> 6, l0 = p & 63;
#pragma unroll
                    for(int r = KPL - 1; r >= 0; --r) {
                        if(r < r0) continue;  // uniform
                        const uint64_t sh = wave_shr1(K[ r ]);
                        if(r > r0) {
                            const uint64_t carry = readlane64(K[ r - 1 > 0 ? r - 1 : 0 ], 63);
                            K[ r ] = lane == 0 ? carry : sh;
                        } else {
                            if(lane > l0) K[ r ] = sh;
                            if(lane == l0) K[ r ] = k;
                        }
                    }
                    cnt = cnt < ef ? cnt + 1 : ef;
                }
            }
            // ---- pop: the first unexpanded key (~0 carries the flag); mark it; publish the next hop's front and radius
            int first = -1, fr = 0;
#pragma unroll
            for(int r = 0; r < KPL; ++r) {
                const unsigned long long m = __ballot(!key_expanded(K[ r ])) & live[ r ];
                if(first < 0 && m) {
                    first = (int)__builtin_ctzll(m);
                    fr = r;
                    node = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)K[ r ], first) >> 1;
                }
            }
            if(first >= 0) {
#pragma unroll
                for(int r = 0; r < KPL; ++r)
                    if(r == fr && lane == first) K[ r ] |= 1ull;  // expanded
                if(split) {
                    uint64_t nf = ~0ull, nw = ~0ull;
                    bool     have = false;
#pragma unroll
                    for(int r = 0; r < KPL; ++r) {
                        const unsigned long long m = __ballot(!key_expanded(K[ r ])) & live[ r ];
                        if(!have && m) {
                            nf = readlane64(K[ r ], (int)__builtin_ctzll(m));
                            have = true;
                        }
                        if(cnt == ef && r == (ef - 1) >> 6) nw = readlane64(K[ r ], (ef - 1) & 63);
                    }
                    if(lane == 0) {
                        front_pub[ par ^ 1 ] = nf;
                        worst_pub[ par ^ 1 ] = nw;
                    }
                }
            }
            LGPU_MARK(3)
            if constexpr(PROF) {  // split form: the list wave's own clock (slot 4, which the visit wave leaves alone then)
                if(split && lane == 0) prof[ 4 ] += (unsigned long long)clock64() - tw;
            }
        }
        if(visit_wave) {
            if(split) {
                // ---- which node the list wave is popping right now: min(front, smallest new key inside the radius)
                const uint64_t f = front_pub[ par ], w = worst_pub[ par ];
                uint64_t       t = f != ~0ull ? f : w;
                bool           got = f != ~0ull;
                for(int base = 0; base < pend; base += 64) {
                    const uint64_t     N = base + lane < pend ? s.newkeys[ base + lane ] : ~0ull;
                    unsigned long long m = __ballot(N < t);
                    while(m) {  // each round at least halves the expected number of smaller keys
                        t = readlane64(N, (int)__builtin_ctzll(m));
                        got = true;
                        m = __ballot(N < t);
                    }
                }
                node = got ? (uint32_t)t >> 1 : EMPTY;
                LGPU_MARK(3)
            }
            if(node == EMPTY) {
                if(lane == 0) *nnew_slot = -1;  // the walk is over
            } else {
                E += 1;
                if(!split) LGPU_MARK(4)
                // the LDS set must keep room for one full neighbour list; otherwise spill to the HBM bitmap (the set holds
                // 3/4 * vis_slots slots, a search visits D of them).  The bitmap is all-zero (VisUndo): nothing to clear.
                if(s.vis_slots && !spilled && viscnt + v.M0 > s.vis_slots / 4 * 3) spilled = true;
                uint32_t        cap;
                const uint32_t *list = neighbors_of(v, node, level, cap);
                if(lane == 0) trace_append<PROF>(s, node | (level > 0 ? TRACE_LISTU : TRACE_LIST0));
                const bool have_list = pf_on && node == pf_node;  // (uniform) the speculative fetch of the previous hop was this node's list
                int nb_new = 0;
                for(uint32_t off = 0; off < cap; off += 64) {
                    const uint32_t i = off + (uint32_t)lane;
                    uint32_t       nb;
                    if(have_list) nb = pf_nb;  // (cap <= 64: one pass)
                    else nb = i < cap ? list[ i ] : EMPTY;
                    if constexpr(PROF) {  // make the list's arrival visible to the phase clock
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        LGPU_MARK(5)
                    }
                    const bool isnew = hop_is_new(s, bitmap, nb, spilled);
                    const unsigned long long m = __ballot(isnew);
                    if(isnew) {
                        s.newids[ nb_new + __popcll(m & ((1ull << lane) - 1ull)) ] = nb;
                        mark_touched<PROF>(s, nb);
                    }
                    if(spilled || !s.vis_slots) undo_record(s, undo, isnew, nb, m, lane);  // these ids went into the HBM bitmap
                    nb_new += __popcll(m);
                }
                if(s.vis_slots && !spilled) viscnt += (uint32_t)nb_new;
                if(lane == 0) *nnew_slot = nb_new;
            }
        }
        LGPU_MARK(0)
        __syncthreads();
        const int nnew = *nnew_slot;
        LGPU_MARK(1)
        if(nnew < 0) break;
        pend = nnew;
        if(pf_on && visit_wave) {  // the next hop's front (the list wave published it before the barrier): request its list now
            const uint64_t nf = front_pub[ par ^ 1 ];
            pf_node = nf != ~0ull ? (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)nf >> 1)) : EMPTY;
            if(pf_node != EMPTY) pf_nb = (uint32_t)lane < v.M0 ? v.nbr0[ (size_t)pf_node * v.M0 + (uint32_t)lane ] : EMPTY;
        }
        if(nnew == 0) continue;
        hop_distances<METRIC, G, ROWS, false>(v, s, nnew, qn2, ~0ull, nullptr);
        D += (uint32_t)nnew;
        __syncthreads();
        LGPU_MARK(2)
    }
    if(visit_wave) undo_apply(s, bitmap, bm_words, undo, lane);  // the workgroup's HBM bitmap goes back to all-zero
    // the result goes where the callers read it: s.keys, ascending
    if(list_wave) {
#pragma unroll
        for(int r = 0; r < KPL; ++r)
            if(r * 64 + lane < cnt) s.keys[ r * 64 + lane ] = K[ r ];
        if(lane == 0) s.scal[ S_CNT ] = cnt;
    }
    __syncthreads();
    return s.scal[ S_CNT ];
}
#undef LGPU_MARK

// ---- refine_: the neighbour-selection heuristic --------------------------------------------------------
struct RefineLds
{
    float    *cd;   // candidates in: distance to the centre
    uint32_t *cid;  //                slot
    float    *sd;   // sorted / selected out
    uint32_t *sid;
};
__device__ __forceinline__ unsigned char *carve_refine(unsigned char *p, RefineLds &r, uint32_t n_max)
{
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    r.cd = (float *)p;      p += up16((size_t)n_max * 4);
    r.cid = (uint32_t *)p;  p += up16((size_t)n_max * 4);
    r.sd = (float *)p;      p += up16((size_t)n_max * 4);
    r.sid = (uint32_t *)p;  p += up16((size_t)n_max * 4);
    return p;
}
__host__ inline size_t refine_lds_bytes(uint32_t n_max)
{
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    return 4 * up16((size_t)n_max * 4);
}

// in: r.cd/r.cid[0..n) in any order.  Sorts by (distance, tie_mix(slot, centre)) into r.sd/r.sid,
// then keeps an entry iff no already-kept entry is strictly closer to it than the centre is.
// Returns the number kept (<= needed), left in r.sd/r.sid[0..keep).  `scal` = WalkLds-style scalars.
template <int METRIC, int G>
__device__ int refine(const View &v, RefineLds &r, int *scal, int n, int needed, uint32_t centre, uint32_t &Dr)
{
    const int tid = threadIdx.x, T = blockDim.x, g = tid / G, gl = tid % G, NG = T / G;
    for(int t = tid; t < n; t += T) {
        const float    d = r.cd[ t ];
        const uint32_t id = r.cid[ t ];
        const uint64_t k = ((uint64_t)f2ord(d) << 32) | tie_mix(id, centre);
        int            rank = 0;
        for(int j = 0; j < n; ++j) {
            const uint64_t kj = ((uint64_t)f2ord(r.cd[ j ]) << 32) | tie_mix(r.cid[ j ], centre);
            rank += kj < k;
        }
        r.sd[ rank ] = d;
        r.sid[ rank ] = id;
    }
    if(tid == 0) scal[ S_BAD ] = 0;
    __syncthreads();
    if(n < needed) return n;
    int submitted = 1, consumed = 1;
    while(submitted < needed && consumed < n) {
        const uint32_t cid = r.sid[ consumed ];
        const float    cdist = r.sd[ consumed ];
        const float    cn2 = row_norm<METRIC>(v, cid);
        for(int i = g; i < submitted; i += NG) {
            const uint32_t kid = r.sid[ i ];
            float inter = group_dist_n<METRIC, G>(row_of_m<METRIC>(v, cid), row_of_m<METRIC>(v, kid), (int)v.chunks, gl, cn2, row_norm<METRIC>(v, kid));
            if(gl == G - 1 && inter < cdist) scal[ S_BAD ] = 1;
        }
        Dr += (uint32_t)submitted;
        __syncthreads();
        const bool good = scal[ S_BAD ] == 0;
        __syncthreads();
        if(tid == 0) {
            scal[ S_BAD ] = 0;
            if(good) { r.sid[ submitted ] = cid; r.sd[ submitted ] = cdist; }
        }
        if(good) submitted++;
        consumed++;
        __syncthreads();
    }
    return submitted;
}

}  // namespace lgpu
