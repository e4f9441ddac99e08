// kernels.hip -- gfx950 kernels of the HNSW distance-evaluation path.
//
//   (k_search and k_insert, the two walk kernels, live in search_kernel.hip / insert_kernel.hip; the walk itself in walk.hpp)
//   k_connect       connect_new_node_: neighbour selection with the kept rows in registers; writes the new
//                   node's lists and emits the reverse-link requests.
//   k_revlink_*     the reverse-link half (usearch reconnect_neighbor_nodes_): k_revlink_append (one wave per
//                   (node, level) group: append while there is room; drop the requests that sort behind a full
//                   list's recorded radius), then the re-prune of full lists by row shape: k_revlink_pairs
//                   (all-pairs table, rows in registers, chain form), k_revlink_staged (rows staged whole in LDS),
//                   k_revlink (rows read from L2: lists longer than 32 entries, rows beyond 2048 f32 dims).
//   k_fill_norms    sqrt(||row||^2) of newly stored rows for the cosine metrics (device_common.hpp).
//   k_apply_own_links / k_pack_lists / k_apply_lists
//                   scatter kernels either side of the all-gathers of the work-sharded build (comm.cpp).
//   k_gather        metric(query, row[slots[i]]) -- the distance kernel on its own (tests, profiling).
//   k_pairs         na x nb pairwise distances in the walk's exact reduction order (usearch_distance,
//                   hnsw.c:296-345; PQ k-means assign product_quantization.c:80-124).
//
// HBM-bound by design: every row is read with 16-byte-per-lane coalesced loads (1 KiB per wave
// instruction at G = 64), neighbour ids are staged through LDS, reductions are wavefront shuffles (DPP).
#include <algorithm>
#include <cstdlib>
#include <string>

#include "kernels.hpp"
#include "walk.hpp"
#include "dispatch.hpp"

namespace lgpu {

// ---------------------------------------------------------------------------------------------------
// k_connect: connect_new_node_ -- the neighbour-selection heuristic over one walk result, one workgroup of four
// waves per (new node, level).  It is its own kernel because its best shape differs from the walk's: the kept
// rows live in REGISTERS (G = 64: wave w owns kept entries w, w+4, w+8, w+12; shorter rows: the 256 / G groups of G lanes own
// them round robin -- [r4] all row lengths, not only G = 64: a lone insertion at 128 dimensions spent 62 us here), every group
// loads the candidate row once (the next candidate's row is already in flight) and tests it against its own kept rows with no
// memory access, so a candidate costs one barrier instead of a round of L2 reads.  Same lane/chunk ownership and reduction
// tree as group_dist<METRIC, G>, hence the same bits.  Rows that do not fit this shape (more than G x CPLC chunks, M > 16)
// take the generic refine().
template <int METRIC, int G, int CPLC = 4>  // CPLC: 16-byte chunks per lane of the register path (rows of up to 64 * CPLC chunks)
__global__ void __launch_bounds__(256) k_connect(ConnectArgs a)
{
    const int tid = threadIdx.x, T = blockDim.x;
    RefineLds r;
    unsigned char *p = carve_refine(lgpu_smem, r, a.efc);
    int      *scal = (int *)p;                  p += S_SCALARS * 4;
    uint32_t *kid = (uint32_t *)p;              p += (size_t)((a.view.M + 3) & ~3u) * 4;  // selected slots (<= M)
    float    *kd = (float *)p;
    const uint32_t item = a.item_begin + blockIdx.x;
    const uint32_t b = a.item_node[ item ];
    const uint32_t M = a.view.M;
    const uint32_t me = a.first_slot + b;
    const int      level = (int)(item - a.link_off[ b ] / M);
    LinkReq       *out = a.links + a.link_off[ b ] + (uint32_t)level * M;
    const int      n = (int)a.top_count[ item ];
    const uint64_t *top = a.tops + (size_t)item * a.efc;
    for(int i = tid; i < n; i += T) {
        const uint64_t key = top[ i ];
        r.cd[ i ] = key_dist(key);
        r.cid[ i ] = key_slot(key);
    }
    __syncthreads();
    uint32_t Dr = 0;
    int      keep;
    const int chunks = (int)a.view.chunks;
    // NG groups of G lanes; group g owns kept entries g, g + NG, ... (KPG of them: four per wave at G = 64, one per group at G <= 16)
    constexpr int NG = 256 / G, KPG = (16 + NG - 1) / NG;
    const int     grp = tid / G, gl = tid % G;
    if(chunks <= G * CPLC && M <= 16 && T == 256) {
        // ---- sort by (distance, tie_mix(slot, me)) into sd / sid
        for(int t = tid; t < n; t += T) {
            const uint64_t k = ((uint64_t)f2ord(r.cd[ t ]) << 32) | tie_mix(r.cid[ t ], me);
            int            rank = 0;
            for(int j = 0; j < n; ++j) rank += (((uint64_t)f2ord(r.cd[ j ]) << 32) | tie_mix(r.cid[ j ], me)) < k;
            r.sd[ rank ] = r.cd[ t ];
            r.sid[ rank ] = r.cid[ t ];
        }
        if(tid < 3) scal[ tid ] = 0;  // three rotating "rejected" flags
        __syncthreads();
        if(n < (int)M) {  // refine_: fewer candidates than needed -> keep them all
            keep = n;
            for(int i = tid; i < n; i += T) { kid[ i ] = r.sid[ i ]; kd[ i ] = r.sd[ i ]; }
        } else {
            auto load_row = [&](uint32_t slot, uint4 (&v)[ CPLC ]) {
                const uint4 *row = row_of(a.view, slot);
#pragma unroll
                for(int c = 0; c < CPLC; ++c) {
                    const int ch = gl + G * c;
                    v[ c ] = ch < chunks ? row[ ch ] : make_uint4(0, 0, 0, 0);
                }
            };
            uint4 kept[ KPG ][ CPLC ], cur[ CPLC ], nxt[ CPLC ];
            float keptn[ KPG ];  // cached norms of this group's kept rows (cosine metrics)
#pragma unroll
            for(int j = 0; j < KPG; ++j) {
                keptn[ j ] = 0.f;
#pragma unroll
                for(int c = 0; c < CPLC; ++c) kept[ j ][ c ] = make_uint4(0, 0, 0, 0);
            }
            load_row(r.sid[ 0 ], cur);
            if(grp == 0) {
#pragma unroll
                for(int c = 0; c < CPLC; ++c) kept[ 0 ][ c ] = cur[ c ];
                keptn[ 0 ] = row_norm<METRIC>(a.view, r.sid[ 0 ]);
            }
            if(tid == 0) { kid[ 0 ] = r.sid[ 0 ]; kd[ 0 ] = r.sd[ 0 ]; }
            int submitted = 1, consumed = 1;
            if(n > 1) load_row(r.sid[ 1 ], nxt);
            // (measured and dropped, r4: FOUR candidates per barrier -- every group tests all four against its kept rows, the groups
            // share out the six pairs among the four, every thread replays the four decisions from flags in LDS.  Same picks, but a
            // lone insertion's selection at 128 dimensions went from 48.7 to 59.6 us: 2.5x the distance evaluations per group, each a
            // dependent fma + DPP chain on a SIMD with one wave, cost more than the three barriers saved.)
            while(submitted < (int)M && consumed < n) {
#pragma unroll
                for(int c = 0; c < CPLC; ++c) cur[ c ] = nxt[ c ];
                const float    cdist = r.sd[ consumed ];
                const uint32_t cslot = r.sid[ consumed ];
                const float    cn2 = row_norm<METRIC>(a.view, cslot);
                if(consumed + 1 < n) load_row(r.sid[ consumed + 1 ], nxt);  // in flight while this one is tested
                bool bad = false;
#pragma unroll
                for(int j = 0; j < KPG; ++j) {
                    if(grp + NG * j < submitted) {  // group-uniform
                        RowAcc<METRIC> acc;
#pragma unroll
                        for(int c = 0; c < CPLC; ++c) acc.add(cur[ c ], kept[ j ][ c ]);
                        const float d = acc.template finish_n<G>(cn2, keptn[ j ]);
                        bad |= d < cdist;  // meaningful in the group's last lane
                    }
                }
                const int slot = consumed % 3;
                if(gl == G - 1 && bad) scal[ slot ] = 1;
                if(tid == 0) scal[ (consumed + 1) % 3 ] = 0;
                Dr += (uint32_t)submitted;
                __syncthreads();
                if(scal[ slot ] == 0) {
                    const int owner = submitted % NG, j = submitted / NG;
                    if(grp == owner) {
#pragma unroll
                        for(int jj = 0; jj < KPG; ++jj)
                            if(jj == j) {
#pragma unroll
                                for(int c = 0; c < CPLC; ++c) kept[ jj ][ c ] = cur[ c ];
                                keptn[ jj ] = cn2;
                            }
                    }
                    if(tid == 0) { kid[ submitted ] = cslot; kd[ submitted ] = cdist; }
                    submitted++;
                }
                consumed++;
            }
            keep = submitted;
        }
        __syncthreads();
    } else {
        keep = refine<METRIC, G>(a.view, r, scal, n, (int)M, me, Dr);
        for(int i = tid; i < keep; i += T) { kid[ i ] = r.sid[ i ]; kd[ i ] = r.sd[ i ]; }
        __syncthreads();
    }
    // ---- the node's own list and one reverse-link request per pick
    uint32_t  cap;
    uint32_t *list = neighbors_of(a.view, me, level, cap);
    for(uint32_t i = tid; i < M; i += T) {
        LinkReq req;
        req.close = EMPTY;
        req.level = (uint32_t)level;
        req.new_slot = me;
        req.d = 0.f;
        if((int)i < keep) {
            list[ i ] = kid[ i ];
            req.close = kid[ i ];
            req.d = kd[ i ];
        }
        out[ i ] = req;
    }
    if(tid == 0 && a.totals) atomicAdd(&a.totals[ 0 ], (unsigned long long)Dr);
}

// ---------------------------------------------------------------------------------------------------
template <int METRIC, int G>
__global__ void __launch_bounds__(256) k_revlink(RevlinkArgs a)
{
    const int tid = threadIdx.x, T = blockDim.x, g = tid / G, gl = tid % G, NG = T / G;
    RefineLds r;
    unsigned char *p = carve_refine(lgpu_smem, r, a.view.M0 + 1);
    int      *scal = (int *)p;
    const uint32_t ng = *a.ngroups;
    uint32_t       pairs = 0;
    for(uint32_t gi = blockIdx.x; gi < ng; gi += gridDim.x) {
    const uint32_t begin = a.groups[ gi ].x, end = a.groups[ gi ].y;
    const uint32_t close = a.reqs[ begin ].close;
    const int      level = (int)a.reqs[ begin ].level;
    uint32_t       cap;
    uint32_t      *list = neighbors_of(a.view, close, level, cap);
    __syncthreads();  // the previous group is done with the LDS
    if(tid == 0) scal[ S_CNT ] = 0;
    __syncthreads();
    for(uint32_t i = tid; i < cap; i += T) {
        const uint32_t nb = list[ i ];
        r.cid[ i ] = nb;
        if(nb != EMPTY) atomicMax(&scal[ S_CNT ], (int)i + 1);
    }
    __syncthreads();
    int      c = scal[ S_CNT ];
    const int c0 = c;
    bool     have_d = false;
    for(uint32_t t = begin; t < end; ++t) {
        const uint32_t vnew = a.reqs[ t ].new_slot;
        const float    dv = a.reqs[ t ].d;
        if(c < (int)cap) {  // room: close_header.push_back(new_slot)
            if(tid == 0) { r.cid[ c ] = vnew; r.cd[ c ] = dv; list[ c ] = vnew; }
            c++;
            __syncthreads();
            continue;
        }
        if(!have_d) {  // distances of the original entries to `close`, once
            for(int i = g; i < c0; i += NG) {
                float d = group_dist_n<METRIC, G>(row_of(a.view, close), row_of(a.view, r.cid[ i ]), (int)a.view.chunks, gl,
                                                  row_norm<METRIC>(a.view, close), row_norm<METRIC>(a.view, r.cid[ i ]));
                if(gl == G - 1) r.cd[ i ] = d;
            }
            pairs += (uint32_t)c0;
            have_d = true;
        }
        if(tid == 0) { r.cid[ c ] = vnew; r.cd[ c ] = dv; }
        __syncthreads();
        const int keep = refine<METRIC, G>(a.view, r, scal, c + 1, (int)cap, close, pairs);
        for(uint32_t i = tid; i < cap; i += T) {
            if((int)i < keep) {
                r.cid[ i ] = r.sid[ i ];
                r.cd[ i ] = r.sd[ i ];
                list[ i ] = r.sid[ i ];
            } else {
                list[ i ] = EMPTY;
            }
        }
        c = keep;
        __syncthreads();
    }
    }  // groups
    if(tid == 0 && a.totals) atomicAdd(&a.totals[ 0 ], (unsigned long long)pairs);
}

// ---------------------------------------------------------------------------------------------------
// k_revlink_staged: the same reverse-link step with the candidate rows staged in LDS.
//
// Re-pruning a full list evaluates up to cap*(cap+1)/2 candidate-candidate distances over only cap+2
// distinct rows (33 list entries + the new node + `close` at M=16).  k_revlink re-reads both rows of
// every pair from L2 (measured: the build's dominant cost, L2-bandwidth bound).  Here the rows are
// loaded ONCE with coalesced 16-byte loads into LDS (34 x 3 KiB = 102 KiB at d=768), all pair
// distances are evaluated from LDS in one parallel phase, and the heuristic itself runs on the small
// distance matrix.  The metrics are bitwise symmetric and the pair set is a superset of what the
// sequential heuristic evaluates, so the result is identical to k_revlink / the oracle.
struct StagedLds
{
    uint4    *rows;   // [(cap + 2)][chunks]
    float    *cd;     // [cap + 1] distance to `close`, candidate order
    uint32_t *cid;    // [cap + 1]
    float    *sd;     // sorted
    uint32_t *sid;
    uint16_t *sidx;   // sorted position -> candidate (= staged row) index
    float    *pair;   // [(cap + 1)][(cap + 1)] by sorted positions, i > j
    int      *scal;
};
__host__ __device__ inline size_t staged_lds_bytes(uint32_t chunks, uint32_t cap)
{
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t n = cap + 1;
    return (size_t)(cap + 2) * chunks * 16 + 4 * up16(n * 4) + up16(n * 2) + up16(n * n * 4) + S_SCALARS * 4 + up16((n + 1) * 4) + 2 * up16(n + 1);
}

// k_revlink_append: the cheap half of the reverse-link step, one WAVE per (close, level) group: append
// requests while the list has room.  A group whose list fills up is handed to k_revlink_staged through a
// worklist entry {group, first unprocessed request}.  Splitting the step keeps the trivial groups (the
// majority) out of the 110 KiB-LDS kernel, which can only hold one workgroup per CU.
struct RevWork
{
    uint32_t group;
    uint32_t t_start;
};

__global__ void __launch_bounds__(256) k_revlink_append(RevlinkArgs a, RevWork *work, uint32_t *work_count)
{
    const int      lane = threadIdx.x & 63;
    const uint32_t ng = *a.ngroups, nwaves = gridDim.x * (blockDim.x >> 6);
    for(uint32_t gi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); gi < ng; gi += nwaves) {
    const uint32_t begin = a.groups[ gi ].x, end = a.groups[ gi ].y;
    const uint32_t close = a.reqs[ begin ].close;
    const int      level = (int)a.reqs[ begin ].level;
    uint32_t       cap;
    uint32_t      *list = neighbors_of(a.view, close, level, cap);
    // count = position of the first EMPTY slot (lists have no holes)
    uint32_t c = 0;
    for(uint32_t off = 0; off < cap; off += 64) {
        const uint32_t i = off + (uint32_t)lane;
        const bool     used = i < cap && list[ i ] != EMPTY;
        c += (uint32_t)__popcll(__ballot(used));
    }
    const uint32_t room = cap - c, nreq = end - begin;
    const uint32_t take = room < nreq ? room : nreq;
    for(uint32_t t = (uint32_t)lane; t < take; t += 64) list[ c + t ] = a.reqs[ begin + t ].new_slot;
    if(take < nreq) {
        uint32_t t0 = begin + take;
        // PERSISTENT STATE (RevlinkArgs::radius0): a list some earlier re-prune left full is still greedy-consistent and sorted,
        // and its radius -- d(close, last entry) -- is on record.  A request that sorts behind the radius would be cut by the
        // re-prune without changing anything: the leading run of such requests is dropped here, without reading a row -- the
        // common case at a hub, across batches -- and a group made of nothing else gets no work item at all.
        if(take == 0 && a.radius0) {
            const float rad = level == 0 ? a.radius0[ close ] : a.radius_upper[ a.view.upper_off[ close ] + (uint32_t)(level - 1) ];
            if(rad == rad) {  // not NaN
                const uint64_t kr = ((uint64_t)f2ord(rad) << 32) | tie_mix(list[ cap - 1 ], close);
                uint32_t       skipped = 0;
                for(uint32_t base = t0; base < end; base += 64) {
                    const uint32_t t = base + (uint32_t)lane;
                    const bool     enters = t < end && !(kr < (((uint64_t)f2ord(a.reqs[ t < end ? t : begin ].d) << 32) | tie_mix(a.reqs[ t < end ? t : begin ].new_slot, close)));
                    const unsigned long long m = __ballot(enters);
                    if(m) { skipped += (uint32_t)__builtin_ctzll(m); break; }
                    skipped += end - base < 64u ? end - base : 64u;
                }
                t0 += skipped;
                if(lane == 0 && skipped && a.totals) atomicAdd(&a.totals[ 1 ], (unsigned long long)skipped);  // they count as re-prunes
            }
        }
        if(t0 < end && lane == 0) {
            const uint32_t w = atomicAdd(work_count, 1u);
            work[ w ].group = gi;
            work[ w ].t_start = t0;
        }
    }
    }  // groups
}

template <int METRIC, int G>
__global__ void __launch_bounds__(512) k_revlink_staged(RevlinkArgs a, const RevWork *work, const uint32_t *work_count)
{
    const int tid = threadIdx.x, T = blockDim.x, g = tid / G, gl = tid % G, NG = T / G, lane = tid & 63;
    const uint32_t nwork = *work_count;
    uint32_t pairs = 0, reprunes = 0;
    auto order_key = [](float d, uint32_t id, uint32_t centre) { return ((uint64_t)f2ord(d) << 32) | tie_mix(id, centre); };
    for(uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    const uint32_t gi = work[ wi ].group;
    const uint32_t begin = a.groups[ gi ].x, end = a.groups[ gi ].y;
    const uint32_t t_first = work[ wi ].t_start;
    const uint32_t close = a.reqs[ begin ].close;
    const int      level = (int)a.reqs[ begin ].level;
    uint32_t       cap;
    uint32_t      *list = neighbors_of(a.view, close, level, cap);
    const int      chunks = (int)a.view.chunks;
    StagedLds s;
    uint8_t  *lrs, *socc;  // chain state: LDS row of list entry i; row occupied by a list entry
    float    *dxy;         // chain step: d(x, row)
    {
        auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
        unsigned char *p = lgpu_smem;
        const size_t   n = cap + 1;
        s.rows = (uint4 *)p;     p += (size_t)(cap + 2) * chunks * 16;
        s.cd = (float *)p;       p += up16(n * 4);
        s.cid = (uint32_t *)p;   p += up16(n * 4);
        s.sd = (float *)p;       p += up16(n * 4);
        s.sid = (uint32_t *)p;   p += up16(n * 4);
        s.sidx = (uint16_t *)p;  p += up16(n * 2);
        s.pair = (float *)p;     p += up16(n * n * 4);
        s.scal = (int *)p;       p += S_SCALARS * 4;
        dxy = (float *)p;        p += up16((n + 1) * 4);
        lrs = (uint8_t *)p;      p += up16(n + 1);
        socc = (uint8_t *)p;
    }
    __syncthreads();  // the previous work item is done with the LDS
    // the list is full at hand-off: entries [c0, cap) were appended by k_revlink_append from the requests
    // [begin, t_first), whose distances to `close` are known; the older entries are evaluated on first use
    for(uint32_t i = tid; i < cap; i += T) s.cid[ i ] = list[ i ];
    int       c = (int)cap;
    const int c0 = (int)cap - (int)(t_first - begin);
    for(int i = c0 + tid; i < (int)cap; i += T) s.cd[ i ] = a.reqs[ begin + (uint32_t)(i - c0) ].d;
    __syncthreads();
    bool have_d = false;
    // CHAIN (see k_revlink_pairs): after a full re-prune that leaves the list full, s.cid / s.cd hold the greedy-consistent
    // list in order, the entries' rows stay in LDS (row lrs[i]), and one more candidate costs its own row and <= cap
    // distances -- or nothing, when it sorts behind the last entry.  cap <= 64 (one lane of wave 0 per entry).
    bool chain = false;
    for(uint32_t t = t_first; t < end; ++t) {
        const uint32_t vnew = a.reqs[ t ].new_slot;
        const float    dv = a.reqs[ t ].d;
        if(chain) {
            reprunes++;
            const uint64_t kx = order_key(dv, vnew, close);
            const bool     before = lane < c && order_key(s.cd[ lane < c ? lane : 0 ], s.cid[ lane < c ? lane : 0 ], close) < kx;
            const int      pos = (int)__popcll(__ballot(before));  // every wave for itself
            if(pos == c) continue;                                 // cut unseen: behind the last entry of a full list
            // x's row into a free LDS row (wave 0 finds it: there are cap + 2 rows for <= cap entries)
            if(tid < 64) {
                const int                nrows = (int)cap + 2;
                const unsigned long long free_rows = __ballot(lane < nrows && socc[ lane < nrows ? lane : 0 ] == 0);
                if(lane == 0) s.scal[ 1 ] = (int)__builtin_ctzll(free_rows);
            }
            __syncthreads();
            const int xrow = s.scal[ 1 ];
            for(int ch = tid; ch < chunks; ch += T) s.rows[ (size_t)xrow * chunks + ch ] = row_of(a.view, vnew)[ ch ];
            __syncthreads();
            const float xn = row_norm<METRIC>(a.view, vnew);
            for(int i = g; i < c; i += NG) {  // d(x, entry i), one G-lane group per entry
                const int   r = lrs[ i ];
                const float d = group_dist_n<METRIC, G>(s.rows + (size_t)xrow * chunks, s.rows + (size_t)r * chunks, chunks, gl, xn,
                                                        row_norm<METRIC>(a.view, s.cid[ i ]));
                if(gl == G - 1) dxy[ i ] = d;
            }
            pairs += (uint32_t)c;
            __syncthreads();
            if(tid < 64) {
                const bool     live = lane < c;
                const int      myrow = live ? (int)lrs[ lane ] : 0;
                const float    di = live ? dxy[ lane ] : 0.f, mysd = live ? s.cd[ lane ] : 0.f;
                const uint32_t myid = live ? s.cid[ lane ] : 0u;
                const bool     x_bad = __ballot(before && di < dv) != 0ull;
                int            cn = c;
                if(!x_bad) {
                    const bool               keep = live && (before || !(di < mysd));
                    const unsigned long long after_kept = __ballot(keep && !before);
                    int np = lane;
                    if(live && !before) np = pos + 1 + (int)__popcll(after_kept & ((1ull << lane) - 1ull));
                    const bool stays = keep && np < (int)cap;
                    cn = pos + 1 + (int)__popcll(after_kept);
                    if(cn > (int)cap) cn = (int)cap;
                    if(live && !stays) socc[ myrow ] = 0;
                    if(stays) { s.cid[ np ] = myid; s.cd[ np ] = mysd; lrs[ np ] = (uint8_t)myrow; }
                    if(lane == 0) { s.cid[ pos ] = vnew; s.cd[ pos ] = dv; lrs[ pos ] = (uint8_t)xrow; socc[ xrow ] = 1; }
                }
                if(lane == 0) s.scal[ S_CNT ] = cn;
            }
            __syncthreads();
            c = s.scal[ S_CNT ];
            if(c < (int)cap) { chain = false; have_d = true; }  // short of cap: appends follow; every entry's distance to `close` is in s.cd
            continue;
        }
        if(c < (int)cap) {
            if(tid == 0) { s.cid[ c ] = vnew; s.cd[ c ] = dv; list[ c ] = vnew; }
            c++;
            __syncthreads();
            continue;
        }
        reprunes++;
        // ---- stage the c current entries, the new node (row c) and `close` (row c + 1)
        if(tid == 0) { s.cid[ c ] = vnew; s.cd[ c ] = dv; }
        __syncthreads();
        const int n = c + 1;
        {
            // eight independent 16-byte loads in flight per thread before the first LDS store (a plain
            // load->store loop serialises one HBM round trip per iteration)
            const int total = (n + 1) * chunks;
            for(int base = tid; base < total; base += T * 8) {
                uint4 v[ 8 ];
#pragma unroll
                for(int u = 0; u < 8; ++u) {
                    const int idx = base + u * T;
                    if(idx < total) {
                        const int      r = idx / chunks, ch = idx - r * chunks;
                        const uint32_t slot = r < n ? s.cid[ r ] : close;
                        v[ u ] = row_of(a.view, slot)[ ch ];
                    }
                }
#pragma unroll
                for(int u = 0; u < 8; ++u) {
                    const int idx = base + u * T;
                    if(idx < total) s.rows[ idx ] = v[ u ];
                }
            }
        }
        __syncthreads();
        if(!have_d) {  // distances of the original entries to `close`, once (a = close, b = entry)
            for(int i = g; i < c0; i += NG) {
                float d = group_dist_n<METRIC, G>(s.rows + (size_t)n * chunks, s.rows + (size_t)i * chunks, chunks, gl,
                                                  row_norm<METRIC>(a.view, close), row_norm<METRIC>(a.view, s.cid[ i ]));
                if(gl == G - 1) s.cd[ i ] = d;
            }
            pairs += (uint32_t)c0;
            have_d = true;
            __syncthreads();
        }
        // ---- sort by (distance to close, tie_mix(slot, close))
        for(int x = tid; x < n; x += T) {
            const uint64_t k = order_key(s.cd[ x ], s.cid[ x ], close);
            int            rank = 0;
            for(int j = 0; j < n; ++j) rank += order_key(s.cd[ j ], s.cid[ j ], close) < k;
            s.sd[ rank ] = s.cd[ x ];
            s.sid[ rank ] = s.cid[ x ];
            s.sidx[ rank ] = (uint16_t)x;
        }
        __syncthreads();
        // ---- all candidate-candidate distances (sorted positions i > j), from LDS
        {
            const int total = n * (n - 1) / 2;
            int       i = 1, j = g;
            while(j >= i) { j -= i; ++i; }
            for(int p = g; p < total; p += NG) {
                float d = group_dist_n<METRIC, G>(s.rows + (size_t)s.sidx[ i ] * chunks, s.rows + (size_t)s.sidx[ j ] * chunks, chunks, gl,
                                                  row_norm<METRIC>(a.view, s.sid[ i ]), row_norm<METRIC>(a.view, s.sid[ j ]));
                if(gl == G - 1) s.pair[ i * n + j ] = d;
                j += NG;
                while(j >= i) { j -= i; ++i; }
            }
            pairs += (uint32_t)total;
        }
        __syncthreads();
        // ---- the heuristic on the distance matrix: wave 0, one lane per already-kept entry
        // (lane x holds the sorted positions of kept entries x, x+64, x+128, x+192; cap <= 256)
        if(tid < 64) {
            int       kpos[ 4 ] = { 0, 0, 0, 0 };  // kept[0] = sorted position 0
            int       submitted = 1, consumed = 1;
            while(submitted < (int)cap && consumed < n) {
                const float cdist = s.sd[ consumed ];
                bool        bad = false;
#pragma unroll
                for(int j = 0; j < 4; ++j) {
                    const int x = lane + 64 * j;
                    if(x < submitted) bad |= s.pair[ consumed * n + kpos[ j ] ] < cdist;
                }
                if(!__any(bad)) {
                    if((submitted & 63) == lane) {
#pragma unroll
                        for(int j = 0; j < 4; ++j)
                            if((submitted >> 6) == j) kpos[ j ] = consumed;
                    }
                    submitted++;
                }
                consumed++;
            }
            if(cap <= 62) {  // chain bookkeeping: which LDS rows the kept entries sit in (row = candidate index before the sort)
                if(lane < (int)cap + 2) socc[ lane ] = 0;
                if(lane < submitted) { lrs[ lane ] = (uint8_t)s.sidx[ kpos[ 0 ] ]; socc[ s.sidx[ kpos[ 0 ] ] ] = 1; }
            }
#pragma unroll
            for(int j = 0; j < 4; ++j) {
                const int x = lane + 64 * j;
                if(x < submitted) { s.cd[ x ] = s.sd[ kpos[ j ] ]; s.cid[ x ] = s.sid[ kpos[ j ] ]; }
            }
            if(lane == 0) s.scal[ S_CNT ] = submitted;
        }
        __syncthreads();
        c = s.scal[ S_CNT ];
        chain = c == (int)cap && cap <= 62;
    }
    for(uint32_t i = tid; i < cap; i += T) list[ i ] = (int)i < c ? s.cid[ i ] : EMPTY;
    }  // work items
    if(tid == 0 && a.totals) { atomicAdd(&a.totals[ 0 ], (unsigned long long)pairs); atomicAdd(&a.totals[ 1 ], (unsigned long long)reprunes); }
}

// ---------------------------------------------------------------------------------------------------
// k_revlink_pairs: the re-prune of a full list (rows of 128..256 chunks, cap <= 32) with NO sequential dependency in
// its distance phase.  Round 1 walked the sorted candidates one by one -- one barrier and one LDS hand-off per
// candidate, a ~0.85 us chain x 32 -- because whether candidate c is tested against candidate p depends on p having been
// kept.  Here ALL pairs among the <= 33 candidates and `close` are evaluated up front ((n+1) n / 2 <= 561 independent
// distances, ~18 % more than the sequential walk evaluates), then the keep/drop scan runs over that table with 64-bit
// "blocker" masks (one wave, ~n steps of readlane).  Rows never leave registers for the arithmetic's second operand:
//   * wave w (of 8) loads rows w, w+8, w+16, ... of the (n+1)-row set straight from HBM into registers (<= 5 rows,
//     all loads in flight at once) -- every row is read from HBM exactly once per re-prune;
//   * in round t each wave publishes its row 8t+w to one half of a double-buffered LDS ring; after ONE barrier every wave
//     reads the round's <= 8 rows from LDS and evaluates them against its own register rows with a higher index.
// Five rounds, five barriers, for what was thirty-two.  Lane/chunk ownership and the reduction tree are those of
// group_dist<METRIC, 64> (lane l owns chunks l, l+64, ...; one fma chain per accumulator in memory order; DPP butterfly),
// and the metrics are bitwise symmetric, so every table entry has the bits the sequential kernels / the oracle compute.
//
// CHAINS.  Requests to one (node, level) are applied in new-slot order, each against the list the previous one left: a
// hub node that 300 new nodes of a batch picked is a chain of 300 dependent re-prunes, and with one workgroup per group
// that chain IS the kernel's run time on hubby data (i.i.d. Gaussian rows: largest group 200-400 requests per 8192-node
// batch; measured, the waves of this kernel were resident 28 % of its duration -- the rest was the tail of such chains).
// A re-prune's output is greedy-consistent: every kept entry passed against all kept entries before it, in sorted order.
// Adding ONE candidate x to such a list L needs none of L's pairs again:
//     * entries before x in the order are kept as they were (their tests involve only entries before them);
//     * x is kept iff no entry before it is closer to it than `close` is: d(x, y) >= d(x, close) for all y before x;
//     * an entry c after x is kept iff x does not block it: x dropped, or d(c, x) >= d(c, close)  (entries x removes
//       cannot have blocked anything: everything after them was kept WITH them present);
//     * the result is cut at cap -- and is greedy-consistent again.
// That is <= cap distances d(x, .) instead of (cap+1) cap / 2 pairs -- and none at all when the list is full and x sorts
// behind its last entry (x is cut: the common case at a hub, decided from the request's own d(x, close)).  So after the
// first full re-prune of a group the kept rows STAY in the waves' registers and every further request costs one row
// (x's), <= 5 distances per wave and three barriers; the chain's critical path drops from ~5 us to <1 us per link.
// The chain ends when a step leaves the list short of cap (later requests append unchecked; the next overflow takes the
// full path again).  Same decisions, same order, same bits as the sequential algorithm: graphs stay edge-for-edge equal
// to the oracle's.
constexpr int PAIRS_NMAX = 34;   // candidates (<= 33) + close
constexpr int PAIRS_SLOTS = 40;  // register row slots: 8 waves x 5
__host__ __device__ inline size_t pairs_lds_bytes(int cpl)
{
    return (size_t)2 * 8 * 64 * cpl * 16 + (size_t)PAIRS_NMAX * PAIRS_NMAX * 4 + 10 * 160 + 320 + 64 + 64;
}

template <int METRIC, int CPL>
__global__ void __launch_bounds__(512, CPL <= 3 ? 4 : 2) k_revlink_pairs(RevlinkArgs a, const RevWork *work, const uint32_t *work_count)
{
    constexpr int NM = PAIRS_NMAX;
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *p = lgpu_smem;
    uint4    *ring = (uint4 *)p;                   p += (size_t)2 * 8 * 64 * CPL * 16;  // [half][wave][64 * CPL]
    float    *pair = (float *)p;                   p += (size_t)NM * NM * 4;            // [i][j], i > j; row n = close
    float    *cd = (float *)p;                     p += 160;
    float    *sd = (float *)p;                     p += 160;
    float    *nrm = (float *)p;                    p += 160;
    uint32_t *cid = (uint32_t *)p;                 p += 160;
    uint32_t *sid = (uint32_t *)p;                 p += 160;
    int      *rank = (int *)p;                     p += 160;
    uint16_t *sidx = (uint16_t *)p;                p += 160;
    float    *lsd = (float *)p;                    p += 160;   // chain state: d(close, list entry), list order
    float    *dxy = (float *)p;                    p += 160;   // chain step: d(x, row slot)
    uint8_t  *lrs = (uint8_t *)p;                  p += 64;    // chain state: row slot of list entry i
    uint8_t  *socc = (uint8_t *)p;                 p += 64;    // chain state: row slot occupied by a list entry
    unsigned long long *blockers = (unsigned long long *)p;  p += 320;
    int      *scal = (int *)p;                     // [0] count, [1] x's row slot (-1: x dropped)
    const uint32_t nwork = *work_count;
    const int      chunks = (int)a.view.chunks;
    uint32_t       pairs = 0, reprunes = 0;
    auto load_row = [&](uint32_t slot, uint4 (&v)[ CPL ]) {
        const uint4 *row = row_of(a.view, slot);
#pragma unroll
        for(int c = 0; c < CPL; ++c) {
            const int ch = lane + 64 * c;
            v[ c ] = ch < chunks ? row[ ch ] : make_uint4(0, 0, 0, 0);
        }
    };
    auto order_key = [](float d, uint32_t id, uint32_t centre) { return ((uint64_t)f2ord(d) << 32) | tie_mix(id, centre); };
    for(uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        const uint32_t gi = work[ wi ].group;
        const uint32_t begin = a.groups[ gi ].x, end = a.groups[ gi ].y;
        const uint32_t t_first = work[ wi ].t_start;
        const uint32_t close = a.reqs[ begin ].close;
        const int      level = (int)a.reqs[ begin ].level;
        uint32_t       cap;
        uint32_t      *list = neighbors_of(a.view, close, level, cap);
        __syncthreads();  // the previous work item is done with the LDS
        for(uint32_t i = tid; i < cap; i += T) cid[ i ] = list[ i ];
        int c = (int)cap;  // the list is full at hand-off (k_revlink_append filled it from the requests [begin, t_first))
        __syncthreads();
        // the rows of the current list entries, in this wave's registers while a chain runs: row slot r = wave + 8 j
        uint4 own[ 5 ][ CPL ];
        float ownn[ 5 ];
        bool  chain = false;  // cid / lsd / lrs / socc + `own` describe a full, greedy-consistent list
        for(uint32_t t = t_first; t < end; ++t) {
            const uint32_t vnew = a.reqs[ t ].new_slot;
            const float    dv = a.reqs[ t ].d;
            if(chain) {
                // ---- one more candidate for a consistent full list (c == cap)
                reprunes++;
                const uint64_t kx = order_key(dv, vnew, close);
                const bool     before = lane < c && order_key(lsd[ lane < c ? lane : 0 ], cid[ lane < c ? lane : 0 ], close) < kx;
                const int      pos = (int)__popcll(__ballot(before));  // every wave computes it for itself: no barrier
                if(pos == c) continue;                                 // x sorts behind the last entry of a full list: cut, unseen
                uint4 cur[ CPL ];
                if(wave == (int)(t & 7)) {
                    load_row(vnew, cur);
#pragma unroll
                    for(int cc = 0; cc < CPL; ++cc) ring[ lane + 64 * cc ] = cur[ cc ];
                }
                __syncthreads();
                if(wave != (int)(t & 7)) {
#pragma unroll
                    for(int cc = 0; cc < CPL; ++cc) cur[ cc ] = ring[ lane + 64 * cc ];
                }
                const float xn = row_norm<METRIC>(a.view, vnew);
#pragma unroll
                for(int j = 0; j < 5; ++j) {
                    const int r = wave + 8 * j;
                    if(socc[ r ]) {  // wave-uniform
                        RowAcc<METRIC> acc;
#pragma unroll
                        for(int cc = 0; cc < CPL; ++cc) acc.add(own[ j ][ cc ], cur[ cc ]);
                        const float d = acc.template finish_n<64>(ownn[ j ], xn);
                        if(lane == 63) dxy[ r ] = d;
                    }
                }
                pairs += (uint32_t)c;
                __syncthreads();
                if(tid < 64) {
                    const bool  live = lane < c;
                    const int   myslot = live ? (int)lrs[ lane ] : 0;
                    const float di = live ? dxy[ myslot ] : 0.f, mysd = live ? lsd[ lane ] : 0.f;
                    const uint32_t myid = live ? cid[ lane ] : 0u;
                    const bool  x_bad = __ballot(before && di < dv) != 0ull;  // an entry before x is closer to x than `close` is
                    int         xslot = -1, cn = c;
                    if(!x_bad) {
                        const bool               keep = live && (before || !(di < mysd));  // after x: dropped iff x is closer to it than `close`
                        const unsigned long long after_kept = __ballot(keep && !before);
                        int np = lane;  // new position
                        if(live && !before) np = pos + 1 + (int)__popcll(after_kept & ((1ull << lane) - 1ull));
                        const bool stays = keep && np < (int)cap;
                        cn = pos + 1 + (int)__popcll(after_kept);
                        if(cn > (int)cap) cn = (int)cap;
                        // a row slot for x: free before this step (there are 40 slots for <= 33 rows)
                        const unsigned long long free_slots = __ballot(lane < PAIRS_SLOTS && socc[ lane < PAIRS_SLOTS ? lane : 0 ] == 0);
                        xslot = (int)__builtin_ctzll(free_slots);
                        if(live && !stays) socc[ myslot ] = 0;
                        if(stays) { cid[ np ] = myid; lsd[ np ] = mysd; lrs[ np ] = (uint8_t)myslot; }
                        if(lane == 0) { cid[ pos ] = vnew; lsd[ pos ] = dv; lrs[ pos ] = (uint8_t)xslot; socc[ xslot ] = 1; }
                    }
                    if(lane == 0) { scal[ 0 ] = cn; scal[ 1 ] = xslot; }
                }
                __syncthreads();
                c = scal[ 0 ];
                const int xslot = scal[ 1 ];
                if(xslot >= 0 && wave == (xslot & 7)) {  // x's row moves into its slot's registers
#pragma unroll
                    for(int j = 0; j < 5; ++j)
                        if(j == (xslot >> 3)) {
#pragma unroll
                            for(int cc = 0; cc < CPL; ++cc) own[ j ][ cc ] = cur[ cc ];
                            ownn[ j ] = xn;
                        }
                }
                if(c < (int)cap) chain = false;  // short of cap: the following requests append unchecked
                continue;
            }
            if(c < (int)cap) {
                if(tid == 0) { cid[ c ] = vnew; list[ c ] = vnew; }
                c++;
                __syncthreads();
                continue;
            }
            reprunes++;
            const int n = c + 1;  // candidates 0..n-1; `close` is row n
            if(tid == 0) cid[ c ] = vnew;
            __syncthreads();
            // ---- this wave's rows, all in flight at once
#pragma unroll
            for(int j = 0; j < 5; ++j) {
                const int r = wave + 8 * j;
                ownn[ j ] = 0.f;
                if(r <= n) {
                    const uint32_t slot = r < n ? cid[ r ] : close;
                    load_row(slot, own[ j ]);
                    ownn[ j ] = row_norm<METRIC>(a.view, slot);
                } else {
#pragma unroll
                    for(int cc = 0; cc < CPL; ++cc) own[ j ][ cc ] = make_uint4(0, 0, 0, 0);
                }
            }
            if(kCachedNorms<METRIC>) {
                if(tid <= n) nrm[ tid ] = row_norm<METRIC>(a.view, tid < n ? cid[ tid ] : close);
            }
            // ---- rounds: publish row 8t + wave, one barrier, evaluate the round's rows against the own rows above them
#pragma unroll
            for(int t8 = 0; t8 < 5; ++t8) {
                if(8 * t8 <= n) {  // block-uniform
                    uint4 *half = ring + (size_t)(t8 & 1) * 8 * 64 * CPL;
                    if(8 * t8 + wave <= n) {
#pragma unroll
                        for(int cc = 0; cc < CPL; ++cc) half[ (size_t)wave * 64 * CPL + lane + 64 * cc ] = own[ t8 ][ cc ];
                    }
                    __syncthreads();  // also: everybody is done reading this half from round t8 - 2
#pragma unroll 1  // one ring row in registers at a time: unrolled, the scheduler hoists all eight rows' LDS reads and spills
                    for(int u = 0; u < 8; ++u) {
                        const int jr = 8 * t8 + u;
                        if(jr > n) break;
                        // does any own row sit above jr?  (rows 8 j + wave with j > t8, or j == t8 and wave > u)
                        if(!(8 * (t8 + 1) + wave <= n || (wave > u && 8 * t8 + wave <= n))) continue;
                        uint4 cur[ CPL ];
#pragma unroll
                        for(int cc = 0; cc < CPL; ++cc) cur[ cc ] = half[ (size_t)u * 64 * CPL + lane + 64 * cc ];
                        const float jn = kCachedNorms<METRIC> ? nrm[ jr ] : 0.f;
#pragma unroll
                        for(int j = 0; j < 5; ++j) {
                            const int i = wave + 8 * j;
                            if(j >= t8 && i <= n && i > jr) {  // wave-uniform
                                RowAcc<METRIC> acc;
#pragma unroll
                                for(int cc = 0; cc < CPL; ++cc) acc.add(own[ j ][ cc ], cur[ cc ]);
                                const float d = acc.template finish_n<64>(ownn[ j ], jn);
                                if(lane == 63) pair[ i * NM + jr ] = d;
                            }
                        }
                    }
                }
            }
            pairs += (uint32_t)((n + 1) * n / 2);
            __syncthreads();
            // ---- distance to `close` = row n of the table; sort by (distance, tie_mix(slot, close))
            for(int x = tid; x < NM; x += T) {
                rank[ x ] = 0;
                blockers[ x ] = 0ull;
                if(x < n) cd[ x ] = pair[ n * NM + x ];
            }
            __syncthreads();
            for(int cell = tid; cell < n * n; cell += T) {
                const int x = cell / n, j = cell - x * n;
                if(order_key(cd[ j ], cid[ j ], close) < order_key(cd[ x ], cid[ x ], close)) atomicAdd(&rank[ x ], 1);
            }
            __syncthreads();
            for(int x = tid; x < n; x += T) {
                const int r = rank[ x ];
                sd[ r ] = cd[ x ];
                sid[ r ] = cid[ x ];
                sidx[ r ] = (uint16_t)x;
            }
            __syncthreads();
            // ---- the heuristic.  blockers[c] = sorted positions p < c that, if kept, reject c (dist(c, p) < dist(c, close));
            // built in parallel, then one wave resolves the sequential dependency: c is kept iff none of its blockers is
            for(int cell = tid; cell < n * n; cell += T) {
                const int cpos = cell / n, ppos = cell - cpos * n;
                if(ppos < cpos) {
                    const int ci = sidx[ cpos ], pi = sidx[ ppos ];
                    const int hi = ci > pi ? ci : pi, lo = ci > pi ? pi : ci;
                    if(pair[ hi * NM + lo ] < sd[ cpos ]) atomicOr(&blockers[ cpos ], 1ull << ppos);
                }
            }
            __syncthreads();
            if(tid < 64) {
                const unsigned long long mine = lane < n ? blockers[ lane ] : 0ull;
                unsigned long long       kept = 1ull;  // sorted position 0 is always kept
                int                      submitted = 1;
                for(int cpos = 1; cpos < n && submitted < (int)cap; ++cpos) {
                    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(mine & 0xFFFFFFFFull), cpos);
                    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(mine >> 32), cpos);
                    const unsigned long long b = ((unsigned long long)hi << 32) | lo;
                    if((b & kept) == 0ull) { kept |= 1ull << cpos; submitted++; }
                }
                if(lane < PAIRS_SLOTS) socc[ lane ] = 0;
                if(lane < submitted) {  // lane x takes the x-th kept position: the list, its distances, its rows' slots
                    unsigned long long m = kept;
                    for(int s2 = 0; s2 < lane; ++s2) m &= m - 1ull;
                    const int kpos = __builtin_ctzll(m);
                    cid[ lane ] = sid[ kpos ];
                    lsd[ lane ] = sd[ kpos ];
                    lrs[ lane ] = (uint8_t)sidx[ kpos ];
                    socc[ sidx[ kpos ] ] = 1;
                }
                if(lane == 0) scal[ 0 ] = submitted;
            }
            __syncthreads();
            c = scal[ 0 ];
            chain = c == (int)cap;  // the rows of the kept entries sit in `own` at their candidate indices
        }
        for(uint32_t i = tid; i < cap; i += T) list[ i ] = (int)i < c ? cid[ i ] : EMPTY;
        // what later batches may rely on (k_revlink_append): a full list out of a re-prune -> its radius; else nothing
        if(a.radius0 && tid == 0) {
            float *const radp = level == 0 ? a.radius0 + close : a.radius_upper + (a.view.upper_off[ close ] + (uint32_t)(level - 1));
            *radp = chain ? lsd[ c - 1 ] : __builtin_nanf("");
        }
    }
    if(tid == 0 && a.totals) { atomicAdd(&a.totals[ 0 ], (unsigned long long)pairs); atomicAdd(&a.totals[ 1 ], (unsigned long long)reprunes); }
}

// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// Work-sharded build: scatter kernels either side of the all-gathers (pure index traffic, no arithmetic).
__global__ void __launch_bounds__(256) k_apply_own_links(View v, uint32_t first_slot, const uint32_t *link_off, const LinkReq *links,
                                                         uint32_t total_links)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= total_links) return;
    const LinkReq req = links[ j ];
    if(req.close == EMPTY) return;  // lists are EMPTY-initialised beyond the kept entries
    const uint32_t b = req.new_slot - first_slot;
    const uint32_t i = j - link_off[ b ] - req.level * v.M;
    uint32_t       cap;
    uint32_t      *list = neighbors_of(v, req.new_slot, (int)req.level, cap);
    list[ i ] = req.close;
}

__global__ void __launch_bounds__(256) k_pack_lists(RevlinkArgs a, uint32_t *records)
{
    const uint32_t rw = a.view.M0 + 2;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = t / rw, w = t - g * rw;
    if(g >= *a.ngroups) return;
    const LinkReq  first = a.reqs[ a.groups[ g ].x ];
    uint32_t       cap;
    const uint32_t *list = neighbors_of(a.view, first.close, (int)first.level, cap);
    uint32_t       val;
    if(w == 0) val = first.close;
    else if(w == 1) val = first.level;
    else val = (w - 2) < cap ? list[ w - 2 ] : EMPTY;
    records[ (size_t)g * rw + w ] = val;
}

__global__ void __launch_bounds__(256) k_apply_lists(View v, const uint32_t *records, uint32_t nrecords)
{
    const uint32_t rw = v.M0 + 2;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = t / rw, w = t - g * rw;
    if(g >= nrecords || w < 2) return;
    const uint32_t close = records[ (size_t)g * rw ];
    if(close == EMPTY) return;  // padding of a rank's segment
    const int level = (int)records[ (size_t)g * rw + 1 ];
    uint32_t  cap;
    uint32_t *list = neighbors_of(v, close, level, cap);
    if(w - 2 < cap) list[ w - 2 ] = records[ (size_t)g * rw + w ];
}

// ---------------------------------------------------------------------------------------------------
// k_fill_norms: ||row||^2 of the rows [first, first + count) into View::norm2 -- once, when the rows enter the index
// (cosine metrics; device_common.hpp "cached row norms").  One G-lane group per row, the a2 chain of Acc<M_COS>.
template <int METRIC, int G>
__global__ void __launch_bounds__(256) k_fill_norms(View v, uint32_t first, uint32_t count, float *norm2)
{
    const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) / G, gl = threadIdx.x % G;
    const uint32_t ngroups = gridDim.x * blockDim.x / G;
    for(uint32_t i = gid; i < count; i += ngroups) {
        const float n2 = group_norm<METRIC, G>(row_of(v, first + i), (int)v.chunks, (int)gl);
        if(gl == G - 1) norm2[ first + i ] = n2;
    }
}

template <int METRIC, int G>
__global__ void __launch_bounds__(256) k_gather(View v, const uint4 *query, const uint32_t *slots, uint32_t n, float *out)
{
    const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) / G, gl = threadIdx.x % G;
    const uint32_t ngroups = gridDim.x * blockDim.x / G;
    for(uint32_t i = gid; i < n; i += ngroups) {
        float d = group_dist<METRIC, G>(query, row_of(v, slots[ i ]), (int)v.chunks, (int)gl);
        if(gl == G - 1) out[ i ] = d;
    }
}

// The same evaluations in the WALK's shape (calibration of the random-row fetch rate a hop's distance phase can reach:
// scripts/bench_gather_ceiling.py): persistent workgroups of four waves, six per CU, the query row in LDS, every G-lane group
// keeping TWO rows in flight (group_dist2_n) -- hop_distances without the hop.  Same chains, same trees: same bits as k_gather.
template <int METRIC, int G>
__global__ void __launch_bounds__(256, 6) k_gather_walkshape(View v, const uint4 *query, const uint32_t *slots, uint32_t n, float *out)
{
    uint4 *q = (uint4 *)lgpu_smem;
    __shared__ float qn_s;
    const uint32_t tid = threadIdx.x, gl = tid % G, NG = blockDim.x / G;
    for(uint32_t i = tid; i < v.chunks; i += blockDim.x) q[ i ] = query[ i ];
    __syncthreads();
    if(kCachedNorms<METRIC>) {
        if(tid < G) {
            const float qn = group_norm<METRIC, G>(q, (int)v.chunks, (int)tid);
            if(tid == G - 1) qn_s = qn;
        }
        __syncthreads();
    }
    const float    qn2 = kCachedNorms<METRIC> ? qn_s : 0.f;
    const uint32_t gid = blockIdx.x * NG + tid / G, ngroups = gridDim.x * NG;
    for(uint32_t i = gid; i < n; i += 2 * ngroups) {
        const uint32_t j = i + ngroups;
        const uint32_t id0 = slots[ i ], id1 = j < n ? slots[ j ] : id0;
        float          d0, d1;
        group_dist2_n<METRIC, G>(q, row_of(v, id0), row_of(v, id1), (int)v.chunks, (int)gl, qn2, row_norm<METRIC>(v, id0), row_norm<METRIC>(v, id1), d0, d1);
        if(gl == G - 1) {
            out[ i ] = d0;
            if(j < n) out[ j ] = d1;
        }
    }
}

template <int METRIC, int G>
__global__ void __launch_bounds__(256) k_pairs(const uint4 *a, uint32_t na, const uint4 *b, uint32_t nb, uint32_t chunks, float *out)
{
    const uint64_t gid = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const uint32_t gl = threadIdx.x % G;
    const uint64_t ngroups = (uint64_t)gridDim.x * blockDim.x / G, total = (uint64_t)na * nb;
    for(uint64_t p = gid; p < total; p += ngroups) {
        const uint32_t i = (uint32_t)(p / nb), j = (uint32_t)(p % nb);
        float d = group_dist<METRIC, G>(a + (size_t)i * chunks, b + (size_t)j * chunks, (int)chunks, (int)gl);
        if(gl == G - 1) out[ p ] = d;
    }
}

size_t connect_lds_bytes(uint32_t efc, uint32_t M) { return refine_lds_bytes(efc) + S_SCALARS * 4 + (size_t)((M + 3) & ~3u) * 8; }

hipError_t launch_connect(int metric, const ConnectArgs &a, hipStream_t stream)
{
    if(a.items == 0) return hipSuccess;
    const size_t lds = connect_lds_bytes(a.efc, a.view.M);
    // rows of more than 256 chunks (d > 1024 f32) take the register path too, with six or eight chunks per lane
    // (one wave per SIMD: the 512-entry register file is the workgroup's)
#define CALL(MM, GG)                                                                                                          \
    {                                                                                                                         \
        if(GG == 64 && a.view.chunks <= 192) hipLaunchKernelGGL((k_connect<MM, GG == 64 ? 64 : GG, GG == 64 ? 3 : 4>), dim3(a.items), dim3(256), lds, stream, a); \
        else if(GG != 64 || a.view.chunks <= 256) hipLaunchKernelGGL((k_connect<MM, GG, 4>), dim3(a.items), dim3(256), lds, stream, a); \
        else if(a.view.chunks <= 384) hipLaunchKernelGGL((k_connect<MM, GG == 64 ? 64 : GG, GG == 64 ? 6 : 4>), dim3(a.items), dim3(256), lds, stream, a); \
        else hipLaunchKernelGGL((k_connect<MM, GG == 64 ? 64 : GG, GG == 64 ? 8 : 4>), dim3(a.items), dim3(256), lds, stream, a); \
    }
    LGPU_DISPATCH(metric, a.view.chunks, CALL);
#undef CALL
    return hipGetLastError();
}

hipError_t launch_revlink(int metric, const RevlinkArgs &a, void *work, uint32_t *work_count, int num_cus, hipStream_t stream, bool work_count_is_zero)
{
    if(a.max_groups == 0) return hipSuccess;
    // the number of groups is known to the device only (*a.ngroups): every kernel below loops over it with a grid sized
    // for the chip, so nothing here waits for the grouping pass
    const int append_grid = (int)std::min<uint32_t>((a.max_groups + 3) / 4, (uint32_t)num_cus * 8);
    // staged variant whenever the 2M+2 rows of a level-0 re-prune fit in LDS (d <= 1024 at M = 16)
    const size_t staged = staged_lds_bytes(a.view.chunks, a.view.M0);
    const bool i8 = mcode_is_i8(metric);  // i8 rows take the LDS-staged / generic kernels (Lantern caps d at 2000: <= 125 chunks)
    if(!i8 && metric != M_COS_B1 && a.view.chunks >= 128 && a.view.chunks <= 512 && a.view.M0 <= 32 && work && work_count) {
        // d = 512..2048 f32 rows, M <= 16: all-pairs re-prune with the rows in registers (k_revlink_pairs)
        hipError_t e = work_count_is_zero ? hipSuccess : hipMemsetAsync(work_count, 0, 4, stream);
        if(e != hipSuccess) return e;
        hipLaunchKernelGGL(k_revlink_append, dim3(append_grid), dim3(256), 0, stream, a, (RevWork *)work, work_count);
        static const bool force4 = std::getenv("LANTERN_GPU_REGS_CPL4") != nullptr;  // tuning: always the four-chunks-per-lane variant
        const bool cpl3 = a.view.chunks <= 192 && !force4;
        // chunks per lane: 3 covers d <= 768 f32 at two workgroups per CU; 4 / 6 / 8 (d <= 1024 / 1536 / 2048) run one per CU
        const int    cpl = cpl3 ? 3 : a.view.chunks <= 256 ? 4 : a.view.chunks <= 384 ? 6 : 8;
        const size_t lds = pairs_lds_bytes(cpl);
        // (a work item is a group whose list overflowed: never more than there are groups -- a lone insertion has at most 2M of them,
        // and a grid of hundreds of idle 512-thread workgroups costs more to dispatch than its re-prunes take)
        const int    grid = (int)std::min<uint32_t>((uint32_t)(num_cus * (cpl3 ? 2 : 1)), a.max_groups);
#define PAIRS_ONE(MM, CC)                                                                                                               \
    {                                                                                                                                   \
        static LdsAttrCache attr_;        \
        ensure_dynamic_lds((const void *)k_revlink_pairs<MM, CC>, lds, attr_);    \
        hipLaunchKernelGGL((k_revlink_pairs<MM, CC>), dim3(grid), dim3(512), lds, stream, a, (const RevWork *)work, work_count);        \
    }
#define PAIRS(MM)                                                                                                                       \
    {                                                                                                                                   \
        if(cpl == 3) PAIRS_ONE(MM, 3) else if(cpl == 4) PAIRS_ONE(MM, 4) else if(cpl == 6) PAIRS_ONE(MM, 6) else PAIRS_ONE(MM, 8)       \
    }
        switch(metric) {
            case M_L2SQ: PAIRS(M_L2SQ); break;
            case M_COS: PAIRS(M_COS); break;
            case M_HAMMING: PAIRS(M_HAMMING); break;
            case M_L2SQ_F16: PAIRS(M_L2SQ_F16); break;
            case M_COS_F16: PAIRS(M_COS_F16); break;
            default: return hipErrorInvalidValue;
        }
#undef PAIRS
#undef PAIRS_ONE
        return hipGetLastError();
    }
    if(staged <= 150 * 1024 && a.view.M0 <= 256 && work && work_count) {
        // as many 8-wave workgroups per CU as the staged rows leave LDS for (short rows: up to four)
        const int staged_per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (150 * 1024) / staged));
        hipError_t e = work_count_is_zero ? hipSuccess : hipMemsetAsync(work_count, 0, 4, stream);
        if(e != hipSuccess) return e;
        hipLaunchKernelGGL(k_revlink_append, dim3(append_grid), dim3(256), 0, stream, a, (RevWork *)work, work_count);
#define CALL(MM, GG)                                                                                                    \
    {                                                                                                                   \
        static LdsAttrCache attr_;        \
        ensure_dynamic_lds((const void *)k_revlink_staged<MM, GG>, staged, attr_);    \
        hipLaunchKernelGGL((k_revlink_staged<MM, GG>), dim3(std::min<uint32_t>((uint32_t)(num_cus * staged_per_cu), a.max_groups)), dim3(512), staged, stream, a, (const RevWork *)work, work_count); \
    }
        LGPU_DISPATCH(metric, a.view.chunks, CALL);
#undef CALL
        return hipGetLastError();
    }
    const size_t lds = refine_lds_bytes(a.view.M0 + 1) + S_SCALARS * 4;
    const int    ggrid = (int)std::min<uint32_t>(a.max_groups, (uint32_t)num_cus * 8);
#define CALL(MM, GG) hipLaunchKernelGGL((k_revlink<MM, GG>), dim3(ggrid), dim3(256), lds, stream, a)
    LGPU_DISPATCH(metric, a.view.chunks, CALL);
#undef CALL
    return hipGetLastError();
}

hipError_t launch_apply_own_links(const View &v, uint32_t first_slot, const uint32_t *link_off, const LinkReq *links,
                                  uint32_t total_links, hipStream_t stream)
{
    if(total_links == 0) return hipSuccess;
    hipLaunchKernelGGL(k_apply_own_links, dim3((total_links + 255) / 256), dim3(256), 0, stream, v, first_slot, link_off, links, total_links);
    return hipGetLastError();
}

hipError_t launch_pack_lists(const RevlinkArgs &a, uint32_t *records, hipStream_t stream)
{
    if(a.max_groups == 0) return hipSuccess;
    const uint64_t threads = (uint64_t)a.max_groups * (a.view.M0 + 2);
    hipLaunchKernelGGL(k_pack_lists, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream, a, records);
    return hipGetLastError();
}

hipError_t launch_apply_lists(const View &v, const uint32_t *records, uint32_t nrecords, hipStream_t stream)
{
    if(nrecords == 0) return hipSuccess;
    const uint64_t threads = (uint64_t)nrecords * (v.M0 + 2);
    hipLaunchKernelGGL(k_apply_lists, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream, v, records, nrecords);
    return hipGetLastError();
}

hipError_t launch_fill_norms(int metric, const View &v, uint32_t first, uint32_t count, float *norm2, hipStream_t stream)
{
    if(count == 0 || (metric != M_COS && metric != M_COS_F16)) return hipSuccess;
    const int G_ = group_lanes_for(v.chunks);
    uint32_t  blocks = (uint32_t)(((uint64_t)count * G_ + 255) / 256);
    if(blocks > 16384) blocks = 16384;
#define FN(MM, GG) hipLaunchKernelGGL((k_fill_norms<MM, GG>), dim3(blocks), dim3(256), 0, stream, v, first, count, norm2)
#define FNG(MM) switch(G_) { case 64: FN(MM, 64); break; case 32: FN(MM, 32); break; case 16: FN(MM, 16); break; default: FN(MM, 8); }
    if(metric == M_COS) FNG(M_COS) else FNG(M_COS_F16)
#undef FNG
#undef FN
    return hipGetLastError();
}

hipError_t launch_gather(int metric, const View &v, const uint4 *query, const uint32_t *slots, uint32_t n, float *out,
                         hipStream_t stream)
{
    if(n == 0) return hipSuccess;
    const int G_ = group_lanes_for(v.chunks);
    uint32_t  blocks = (uint32_t)(((uint64_t)n * G_ + 255) / 256);
    if(blocks > 8192) blocks = 8192;
    if(const char *ws = std::getenv("LANTERN_GPU_GATHER_WALKSHAPE")) {  // calibration runs: the walk's launch shape (k_gather_walkshape)
        if(std::atoi(ws) != 0) {
            int dev = 0, cus = 256;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)cus * 6, ((uint64_t)n * G_ + 255) / 256);
            const size_t   lds = (size_t)v.chunks * 16;
#define CALLW(MM, GG) hipLaunchKernelGGL((k_gather_walkshape<MM, GG>), dim3(grid), dim3(256), lds, stream, v, query, slots, n, out)
            LGPU_DISPATCH(metric, v.chunks, CALLW);
#undef CALLW
            return hipGetLastError();
        }
    }
#define CALL(MM, GG) hipLaunchKernelGGL((k_gather<MM, GG>), dim3(blocks), dim3(256), 0, stream, v, query, slots, n, out)
    LGPU_DISPATCH(metric, v.chunks, CALL);
#undef CALL
    return hipGetLastError();
}

hipError_t launch_pairs(int metric, const uint4 *a, uint32_t na, const uint4 *b, uint32_t nb, uint32_t chunks, float *out,
                        hipStream_t stream)
{
    if(na == 0 || nb == 0) return hipSuccess;
    const int G_ = group_lanes_for(chunks);
    uint64_t  want = ((uint64_t)na * nb * G_ + 255) / 256;
    uint32_t  blocks = want > 8192 ? 8192u : (uint32_t)want;
#define CALL(MM, GG) hipLaunchKernelGGL((k_pairs<MM, GG>), dim3(blocks), dim3(256), 0, stream, a, na, b, nb, chunks, out)
    LGPU_DISPATCH(metric, chunks, CALL);
#undef CALL
    return hipGetLastError();
}

}  // namespace lgpu
