// walk_spec.hpp -- the base-layer walk for LATENCY-BOUND launches (a lone query, batches that cannot fill the chip).
//
// search_level_reg (walk.hpp) spends a hop's time in a chain of dependent steps: decide which node to expand -> fetch its
// neighbour list (HBM round trip) -> visited filter -> barrier -> fetch the new neighbours' rows (HBM round trip) -> reduce ->
// barrier.  When the memory system is not saturated that chain IS the walk's speed (BASELINE config[1]: 2.75 us per hop for a
// lone query against ~1 us on a CPU core; config[2]: a 1024-query batch ends with its longest walk).  This form takes every
// step that is not a true dependency off the chain:
//
//   * ONE barrier per hop.  Every wave decides the next node itself (the same few ballots over the previous hop's keys the
//     visit wave ran alone), so nothing has to be handed out before the row loads start.
//   * SPECULATIVE rows.  The rows of ALL neighbours are requested at once, before the visited filter: ~28 of 32 neighbours
//     are new anyway.  The filter runs on one wave WHILE the loads are in flight and publishes a 64-bit "new" mask; keys of
//     old neighbours are ignored by everybody from the next hop on.  D counts the new ones only (the oracle's count).
//   * The neighbour list comes with the row.  Each row request also asks for the row's own level-0 list (128 B at M = 16);
//     the lists of a hop land in an LDS staging area, and those of keys that can still be expanded (inside the radius) move on
//     to a small direct-mapped LDS cache.  The node a hop expands is either one of the previous hop's keys (half the hops:
//     staging area, no lookup) or an older list entry (cache, tag checked; HBM on a miss) -- the dependent list fetch is gone
//     from ~9 hops in 10.  Lists are immutable while a search kernel runs (inserts wait for searches).
//   * The merge of the previous hop's keys into the register list (list wave), the visited filter (visit wave) and the cache
//     fill (third wave) all run in the shadow of the row loads.  With dedicated role waves (DED: the lone-query shape, 3 + 8
//     waves) they do nothing else; in the 4-wave batch shape each also takes its share of rows, whose loads it issues first.
//
// Same walk, same results: the decision rule, the list, the visited set and the arithmetic are those of search_level_reg;
// ids, distance bits, D and E equal the oracle's (tests/test_gpu_parity.py runs this form beside the others).
// Requirements (checked by the launcher): level 0, 2 <= M0 <= 64, at least two waves, ef <= 64 * KPL.
#pragma once
#include "walk.hpp"

namespace lgpu {

struct SpecLds
{
    uint32_t *stage;   // [2][M0][M0] the level-0 lists of a hop's neighbours, by hop parity and neighbour index (NULL: no list prefetch)
    uint32_t *ctag;    // [cache_entries] slot whose list the entry holds (EMPTY: none)
    uint32_t *clist;   // [cache_entries][M0]
    uint32_t  cache_entries;  // power of two, or 0
    // the two-nodes-per-round walk (walk_twin.hpp) on top: the same for the SPECULATIVE node of a round
    uint32_t *stage2;  // [2][M0][M0]
    uint64_t *keys2;   // [2][M0] its neighbours' keys, by round parity
    uint64_t *tw;      // [2][8] by round parity: the list's first three unexpanded keys | its radius | the two "new" masks
};
constexpr int TW_F0 = 0, TW_W = 3, TW_MASK1 = 4, TW_MASK2 = 5, TW_STRIDE = 8;
__device__ __forceinline__ unsigned char *carve_spec(unsigned char *p, SpecLds &c, uint32_t M0, uint32_t prefetch, uint32_t cache_entries, uint32_t twin = 0)
{
    c.stage = c.stage2 = nullptr;
    c.ctag = c.clist = nullptr;
    c.keys2 = c.tw = nullptr;
    c.cache_entries = 0;
    if(prefetch) {
        c.stage = (uint32_t *)p;   p += (size_t)2 * M0 * M0 * 4;
        c.cache_entries = cache_entries;
        c.clist = (uint32_t *)p;   p += (size_t)cache_entries * M0 * 4;
        c.ctag = (uint32_t *)p;    p += (((size_t)cache_entries * 4) + 15) & ~(size_t)15;
    }
    if(twin) {
        if(prefetch) { c.stage2 = (uint32_t *)p; p += (size_t)2 * M0 * M0 * 4; }
        c.keys2 = (uint64_t *)p;   p += (((size_t)2 * M0 * 8) + 15) & ~(size_t)15;
        c.tw = (uint64_t *)p;      p += (size_t)2 * TW_STRIDE * 8;
    }
    return p;
}
__host__ inline size_t spec_lds_bytes(uint32_t M0, uint32_t prefetch, uint32_t cache_entries, uint32_t twin = 0)
{
    return (prefetch ? (size_t)2 * M0 * M0 * 4 + (size_t)cache_entries * M0 * 4 + ((((size_t)cache_entries * 4) + 15) & ~(size_t)15) : 0) +
           (twin ? (prefetch ? (size_t)2 * M0 * M0 * 4 : 0) + ((((size_t)2 * M0 * 8) + 15) & ~(size_t)15) + (size_t)2 * TW_STRIDE * 8 : 0);
}

__device__ __forceinline__ uint64_t uniform64(uint64_t x)  // a value every lane read from the same address, as scalar registers
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// minimum of a u32 over the wave's 64 lanes, complete in lane 63 (the butterfly of group_sum with min for +, ~0 as the value of
// lanes a step does not write): six DPP ops instead of a ballot / readlane round per candidate
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_take_or_max(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x)
{
    x = min(x, dpp_take_or_max<0xB1, 0xF>(x));   // quad_perm [1,0,3,2]
    x = min(x, dpp_take_or_max<0x4E, 0xF>(x));   // quad_perm [2,3,0,1]
    x = min(x, dpp_take_or_max<0x141, 0xF>(x));  // row_half_mirror
    x = min(x, dpp_take_or_max<0x140, 0xF>(x));  // row_mirror
    x = min(x, dpp_take_or_max<0x142, 0xA>(x));  // row_bcast15 -> rows 1, 3
    x = min(x, dpp_take_or_max<0x143, 0xC>(x));  // row_bcast31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// words of a row's own level-0 list each lane of the group fetches: as few as the group's width allows (one per lane from 32
// lanes on: a 32-entry list costs one register per row in flight)
template <int G> constexpr int spec_list_words() { return G >= 32 ? 1 : G == 16 ? 2 : 4; }
template <int LW> __device__ __forceinline__ void spec_list_load(const uint32_t *src, uint32_t (&dst)[ 4 ])
{
    if constexpr(LW == 1) dst[ 0 ] = src[ 0 ];
    else if constexpr(LW == 2) { const uint2 t = *(const uint2 *)src; dst[ 0 ] = t.x; dst[ 1 ] = t.y; }
    else { const uint4 t = *(const uint4 *)src; dst[ 0 ] = t.x; dst[ 1 ] = t.y; dst[ 2 ] = t.z; dst[ 3 ] = t.w; }
}
template <int LW> __device__ __forceinline__ void spec_list_store(uint32_t *dst, const uint32_t (&src)[ 4 ])
{
    if constexpr(LW == 1) dst[ 0 ] = src[ 0 ];
    else if constexpr(LW == 2) *(uint2 *)dst = make_uint2(src[ 0 ], src[ 1 ]);
    else *(uint4 *)dst = make_uint4(src[ 0 ], src[ 1 ], src[ 2 ], src[ 3 ]);
}

// the rows one G-lane group has in flight during one pass of a hop's distance phase
template <int ROWS, int U> struct SpecPass
{
    int      j[ ROWS ];       // neighbour index of each row (>= count: a duplicate of a valid row, result discarded)
    uint32_t id[ ROWS ];
    float    n2[ ROWS ];
    uint4    y[ ROWS ][ U ];  // the first U chunks per lane of each row
    uint32_t L[ ROWS ][ 4 ];  // this lane's LW words of each row's own level-0 list (lanes < M0 / LW; only [..][0..LW) are used)
    bool     have;            // the group has at least one row in this pass
};

template <int METRIC, int G, int ROWS, int U>
__device__ __forceinline__ void spec_issue(const View &v, const SpecLds &c, SpecPass<ROWS, U> &p, uint32_t nb, int count, int base, int group, int ngroups,
                                           int gl)
{
    const int j0 = base + group;
    p.have = j0 < count;
#pragma unroll
    for(int r = 0; r < ROWS; ++r) {
        p.j[ r ] = j0 + r * ngroups;
        const int jj = p.j[ r ] < count ? p.j[ r ] : (p.have ? j0 : 0);
        if constexpr(G == 64) p.id[ r ] = (uint32_t)__builtin_amdgcn_readlane((int)nb, jj);  // one group per wave: jj is wave-uniform
        else p.id[ r ] = (uint32_t)__builtin_amdgcn_ds_bpermute(jj << 2, (int)nb);
    }
    if(p.have) {
#pragma unroll
        for(int u = 0; u < U; ++u) {
            const int ch = gl + u * G;
            if(ch < (int)v.chunks) {
#pragma unroll
                for(int r = 0; r < ROWS; ++r) p.y[ r ][ u ] = row_of_m<METRIC>(v, p.id[ r ])[ ch ];
            }
        }
#pragma unroll
        for(int r = 0; r < ROWS; ++r) p.n2[ r ] = row_norm<METRIC>(v, p.id[ r ]);
        constexpr int LW = spec_list_words<G>();
        if(c.stage && gl * LW < (int)v.M0) {
#pragma unroll
            for(int r = 0; r < ROWS; ++r) spec_list_load<LW>(v.nbr0 + (size_t)p.id[ r ] * v.M0 + (size_t)(gl * LW), p.L[ r ]);
        }
    }
}

template <int METRIC, int G, int ROWS, int U>
__device__ __forceinline__ void spec_consume(const View &v, const WalkLds &s, const SpecLds &c, SpecPass<ROWS, U> &p, int count, int gl, float qn2,
                                             uint64_t *kout, uint32_t *stage_out)
{
    if(!p.have) return;
    RowAcc<METRIC> acc[ ROWS ];
#pragma unroll
    for(int u = 0; u < U; ++u) {
        const int ch = gl + u * G;
        if(ch < (int)v.chunks) {
            const uint4 x = walk_query<METRIC>(s)[ ch ];
#pragma unroll
            for(int r = 0; r < ROWS; ++r) acc[ r ].add(x, p.y[ r ][ u ]);
        }
    }
    for(int ch = gl + U * G; ch < (int)v.chunks; ch += G) {  // rows longer than U chunks per lane
        const uint4 x = walk_query<METRIC>(s)[ ch ];
        uint4       yy[ ROWS ];
#pragma unroll
        for(int r = 0; r < ROWS; ++r) yy[ r ] = row_of_m<METRIC>(v, p.id[ r ])[ ch ];
#pragma unroll
        for(int r = 0; r < ROWS; ++r) acc[ r ].add(x, yy[ r ]);
    }
#pragma unroll
    for(int r = 0; r < ROWS; ++r) {
        const float d = acc[ r ].template finish_n<G>(qn2, p.n2[ r ]);
        if(gl == G - 1 && p.j[ r ] < count) kout[ p.j[ r ] ] = make_key(d, p.id[ r ]);
    }
    constexpr int LW = spec_list_words<G>();
    if(c.stage && gl * LW < (int)v.M0) {
#pragma unroll
        for(int r = 0; r < ROWS; ++r)
            if(p.j[ r ] < count) spec_list_store<LW>(stage_out + (size_t)p.j[ r ] * v.M0 + (size_t)(gl * LW), p.L[ r ]);
    }
}

// ---- search_for_one_ for the latency-bound launches: greedy descent over levels (begin, end] with ONE barrier per step.
// Every wave reads the current node's list itself (64 bytes at M = 16), the rows are spread over all groups of the workgroup,
// their distances meet in LDS (double-buffered by step parity), and every wave finds the step's winner itself: the sequential
// scan of search_for_one_ -- "first strictly closer wins, in list order" -- ends at the smallest distance, the lowest list index
// among equals, and only if that is strictly closer than the current node: a 64-bit minimum over (distance, index).
// Same node, same D as greedy_descent (walk.hpp), a third of its barriers and no single-thread scan.
template <int METRIC, int G>
__device__ uint32_t greedy_descent_spec(const View &v, WalkLds &s, uint32_t start, int begin_level, int end_level, uint32_t &D)
{
    constexpr int GPW = 64 / G;
    const int     tid = threadIdx.x, lane = tid & 63;
    const int     wv = __builtin_amdgcn_readfirstlane(tid) >> 6, NW = blockDim.x >> 6;
    const int     g = lane / G, gl = lane % G, NG = NW * GPW, group = wv * GPW + g;
    float *const  newd = (float *)s.newkeys;  // [2][M] by step parity (cap_max >= 2 M keys of 8 bytes: room for 4 M floats)
    const float   qn2 = __int_as_float(s.scal[ S_QN2 ]);
    if(wv == 0 && g == 0) {
        const float d = group_dist_n<METRIC, G>(walk_query<METRIC>(s), row_of_m<METRIC>(v, start), (int)v.chunks, gl, qn2, row_norm<METRIC>(v, start));
        if(gl == G - 1) s.scal[ S_CURD ] = __float_as_int(d);
    }
    D += 1;
    __syncthreads();
    uint32_t cur = start;
    float    curd = __int_as_float(__builtin_amdgcn_readfirstlane(s.scal[ S_CURD ]));
    int      step = 0;
    for(int level = begin_level; level > end_level; --level) {
        for(;;) {
            const uint32_t *list = v.upper_nbr + ((size_t)v.upper_off[ cur ] + (size_t)(level - 1)) * v.M;  // level >= 1 here
            const uint32_t  nb = lane < (int)v.M ? list[ lane ] : EMPTY;
            const int       nn = (int)__popcll(__ballot(nb != EMPTY));
            float *const    out = newd + (size_t)(step & 1) * v.M;
            for(int base = 0; base < nn; base += NG) {
                const int      i = base + group;
                const uint32_t id = (uint32_t)__builtin_amdgcn_ds_bpermute((i < nn ? i : 0) << 2, (int)nb);
                if(i < nn) {
                    const float d = group_dist_n<METRIC, G>(walk_query<METRIC>(s), row_of_m<METRIC>(v, id), (int)v.chunks, gl, qn2, row_norm<METRIC>(v, id));
                    if(gl == G - 1) out[ i ] = d;
                }
            }
            D += (uint32_t)nn;
            __syncthreads();
            ++step;
            const uint64_t key = lane < nn ? (((uint64_t)f2ord(out[ lane ]) << 32) | (uint32_t)lane) : ~0ull;
            uint64_t           t = ~0ull;
            unsigned long long m = __ballot(key < t);
            while(m) {
                t = readlane64(key, (int)__builtin_ctzll(m));
                m = __ballot(key < t);
            }
            if(t == ~0ull) break;  // an empty list
            const float dmin = ord2f((uint32_t)(t >> 32));
            if(!(dmin < curd)) break;
            curd = dmin;
            cur = (uint32_t)__builtin_amdgcn_readlane((int)nb, (int)(t & 63ull));
        }
    }
    return cur;
}

// DED: waves 0..2 are role waves only (visit filter | list | cache fill), waves 3.. evaluate rows.  Otherwise every wave
// evaluates rows and waves 0, 1 and (if there are three) 2 take the roles on top.
// PROF (diagnostic instantiations): waves 0..3 (visit | list | fill | a row wave; in the four-wave shape all four evaluate rows)
// add up shader-clock cycles per section of a hop -- [0] decision, [1] neighbour list, [2] issuing the row loads, [3] the role
// section, [4] loads landing + distances, [5] the wait at the barrier -- and [6] hops into prof[8 * wave + i]; [7] of waves
// 0 / 3 / 2 counts where the lists came from: staging area | cache | HBM.
// ROLE >= 0: the instantiation for ONE wave's set of roles -- bits 1 visit | 2 list | 4 fill | 8 rows -- with the other roles'
// sections compiled out.  The instantiations run side by side in one workgroup (search_level_spec below sends each wave to its
// own), meet at the same barriers, and each keeps only its own roles' scalars live: the all-roles-in-one-loop form of the
// dedicated-role shape reloads 43 spilled scalar registers per pass of the hop loop, 41 of them inside the list wave's merge loop.
template <int METRIC, int G, int KPL, int ROWS, int U, bool DED, bool PROF, int ROLE>
__device__ int search_level_spec_impl(const View &v, WalkLds &s, const SpecLds &c, uint32_t *bitmap, uint32_t bm_words, uint32_t start, int ef, uint32_t &D,
                                      uint32_t &E, unsigned long long *prof)
{
    unsigned long long pacc[ 8 ] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tl = 0;
    unsigned           src_stage = 0, src_cache = 0, src_hbm = 0;
#define LGPU_SMARK(i)                                                      \
    if constexpr(PROF) {                                                   \
        const unsigned long long t_ = (unsigned long long)clock64();       \
        pacc[ i ] += t_ - tl;                                              \
        tl = t_;                                                           \
    }
    constexpr int GPW = 64 / G;  // groups per wave
    const int     tid = threadIdx.x, T = blockDim.x, lane = tid & 63;
    const int     wv = __builtin_amdgcn_readfirstlane(tid) >> 6, NW = T >> 6;
    const int     g = lane / G, gl = lane % G;
    const bool    visit_wave = ROLE >= 0 ? (ROLE & 1) != 0 : wv == 0, list_wave = ROLE >= 0 ? (ROLE & 2) != 0 : wv == 1;
    const bool    fill_wave = ROLE >= 0 ? (ROLE & 4) != 0 : NW >= 3 ? wv == 2 : wv == 0;  // the cache fill: a third wave if there is one
    const bool    row_wave = ROLE >= 0 ? (ROLE & 8) != 0 : DED ? wv >= 3 : true;
    const int     ngroups = (DED ? NW - 3 : NW) * GPW;                  // G-lane groups that evaluate rows
    const int     group = ((DED ? wv - 3 : wv) * GPW) + g;              // this lane's group among them
    const uint32_t M0 = v.M0;
    for(uint32_t i = tid; i < s.vis_slots; i += T) s.vis[ i ] = EMPTY;  // (the HBM bitmap is all-zero between walks: walk.hpp VisUndo)
    for(uint32_t i = tid; i < c.cache_entries; i += T) c.ctag[ i ] = EMPTY;
    const float     qn2 = __int_as_float(s.scal[ S_QN2 ]);
    uint64_t *const front_pub = (uint64_t *)&s.scal[ S_FRONT ];  // [2] by hop parity: first unexpanded key of the list the hop starts from
    uint64_t *const worst_pub = (uint64_t *)&s.scal[ S_WORST ];  // [2] its radius (~0: not full)
    uint64_t *const mask_pub = (uint64_t *)&s.scal[ S_MASK ];    // [2] which neighbours of the hop were new
    // a hop's keys, by hop parity: s.newkeys | s.sorted (both hold cap_max >= M0 keys), addressed as ONE base + parity * stride -- a
    // select between two pointers loses their LDS address space and the accesses become flat_load / flat_store (seen in the ISA:
    // every wave's first load of a hop, on the critical path, and the row waves' key stores)
    uint64_t *const keys0 = s.newkeys;
    const ptrdiff_t kstride = s.sorted - s.newkeys;
    // "hop -1" (parity 1) evaluated one row: the start node
    if(wv == 0 && g == 0) {
        const float d = group_dist_n<METRIC, G>(walk_query<METRIC>(s), row_of_m<METRIC>(v, start), (int)v.chunks, gl, qn2, row_norm<METRIC>(v, start));
        if(gl == G - 1) keys0[ kstride ] = make_key(d, start);
        constexpr int LW = spec_list_words<G>();
        if(c.stage && gl * LW < (int)M0) {
            uint32_t piece[ 4 ];
            spec_list_load<LW>(v.nbr0 + (size_t)start * M0 + (size_t)(gl * LW), piece);
            spec_list_store<LW>(c.stage + (size_t)M0 * M0 + (size_t)(gl * LW), piece);
        }
    }
    if(tid == 0) {
        mask_pub[ 1 ] = 1ull;
        front_pub[ 0 ] = ~0ull;  // the list is still empty:
        worst_pub[ 0 ] = ~0ull;  // no front, no radius
    }
    __syncthreads();
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
    // list wave's private state (walk.hpp search_level_reg): lane l of register r holds the (64 r + l)-th smallest key
    uint64_t           K[ KPL ];
    unsigned long long live[ KPL ];
#pragma unroll
    for(int r = 0; r < KPL; ++r) {
        K[ r ] = ~0ull;
        const int m = ef - 64 * r;
        live[ r ] = m >= 64 ? ~0ull : m <= 0 ? 0ull : (1ull << m) - 1ull;
    }
    int cnt = 0;
    if constexpr(PROF) tl = (unsigned long long)clock64();
    for(int hop = 0;; ++hop) {
        const int             par = hop & 1, prv = par ^ 1;
        const uint64_t *const kin = keys0 + (ptrdiff_t)prv * kstride;
        // ---- what the previous hop left: its keys, which of them were new, and the list they are still to be merged into
        const uint64_t           f = uniform64(front_pub[ par ]), w = uniform64(worst_pub[ par ]);
        const unsigned long long pm = uniform64(mask_pub[ prv ]);
        const uint64_t           N = ((pm >> lane) & 1ull) ? kin[ lane ] : ~0ull;
        D += (uint32_t)__popcll(pm);
        // ---- the node this hop expands: the first unexpanded entry of (list merged with the new keys) = min(front, smallest new
        // key inside the radius); a key the merge truncates away is never that minimum (walk.hpp)
        uint64_t t = f != ~0ull ? f : w;
        bool     got = f != ~0ull;
        int      jt = -1;  // the previous hop's neighbour index of the chosen key (-1: it is the list's front)
        if(list_wave && DED) {
            got = got || __ballot(N < t) != 0ull;  // the list wave finds the node in its own registers; it only needs "is there one"
        } else {
            // the smallest new key below t: the wave-wide minimum of the distance words (six DPP ops), then the lanes that hold
            // it -- one, unless two new rows are at exactly the same distance, which the slot words then decide
            const uint32_t hi = N < t ? (uint32_t)(N >> 32) : 0xFFFFFFFFu;
            const uint32_t mh = wave_min_u32(hi);
            unsigned long long m = __ballot(N < t && (uint32_t)(N >> 32) == mh);
            if(m) {
                jt = (int)__builtin_ctzll(m);
                t = readlane64(N, jt);
                got = true;
                m &= m - 1ull;
                while(m) {  // exact ties in distance: the smaller slot wins (keys are (distance, slot))
                    const int      j2 = (int)__builtin_ctzll(m);
                    const uint64_t t2 = readlane64(N, j2);
                    if(t2 < t) { t = t2; jt = j2; }
                    m &= m - 1ull;
                }
            }
        }
        if(!got) break;  // every wave sees the same keys, mask, front and radius: all leave together
        E += 1;
        LGPU_SMARK(0)
        uint32_t nb = EMPTY;
        int      count = 0;
        uint32_t node = EMPTY;
        if(!(list_wave && DED)) {
            node = (uint32_t)t >> 1;
            // ---- its neighbour list: staged with its row by the previous hop | cached since an earlier one | HBM
            // (the cache's tag and its entry are read together -- one LDS round trip -- and the entry is dropped on a mismatch)
            bool hit = false;
            if(c.stage && jt >= 0) {
                if(lane < (int)M0) nb = c.stage[ ((size_t)prv * M0 + (size_t)jt) * M0 + (uint32_t)lane ];
                hit = true;
                src_stage += 1;
            } else if(c.cache_entries) {
                const uint32_t e = node & (c.cache_entries - 1);
                const uint32_t tag = c.ctag[ e ];
                const uint32_t ent = lane < (int)M0 ? c.clist[ (size_t)e * M0 + (uint32_t)lane ] : EMPTY;
                hit = (uint32_t)__builtin_amdgcn_readfirstlane((int)tag) == node;
                if(hit) nb = ent;
                src_cache += hit;
            }
            if(!hit) {
                if(lane < (int)M0) nb = v.nbr0[ (size_t)node * M0 + (uint32_t)lane ];
                src_hbm += 1;
            }
            count = (int)__popcll(__ballot(nb != EMPTY));  // lists are EMPTY-terminated and hole-free
        }
        LGPU_SMARK(1)
        // ---- the rows of ALL its neighbours (old ones too: most are new, and the filter runs behind the loads)
        SpecPass<ROWS, U> p;
        p.have = false;
        if(row_wave) spec_issue<METRIC, G, ROWS, U>(v, c, p, nb, count, 0, group, ngroups, gl);
        LGPU_SMARK(2)
        // ---- in the shadow of the loads: the three role sections
        if(visit_wave) {
            // the LDS set must keep room for one full neighbour list; otherwise spill to the HBM bitmap (walk.hpp)
            if(s.vis_slots && !spilled && viscnt + M0 > s.vis_slots / 4 * 3) spilled = true;  // (the bitmap is all-zero: VisUndo)
            const bool               isnew = hop_is_new(s, bitmap, nb, spilled);
            const unsigned long long nm = __ballot(isnew);
            if(s.vis_slots && !spilled) viscnt += (uint32_t)__popcll(nm);
            if(spilled || !s.vis_slots) undo_record(s, undo, isnew, nb, nm, lane);  // these ids went into the HBM bitmap
            if(lane == 0) mask_pub[ par ] = nm;
        }
        if(list_wave) {
            // merge the previous hop's new keys: one at a time into the sorted registers (rank = one ballot, insertion = one
            // wave-wide DPP shift); only keys inside the radius
            uint64_t worst = ~0ull;
            if(cnt == ef) {
                const int wl = (ef - 1) & 63;
#pragma unroll
                for(int r = 0; r < KPL; ++r)
                    if(r == (ef - 1) >> 6) worst = readlane64(K[ r ], wl);
            }
            unsigned long long todo = __ballot(N < worst);
            while(todo) {
                const int tt = (int)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                const uint64_t k = readlane64(N, tt);
                int            pos = 0;
#pragma unroll
                for(int r = 0; r < KPL; ++r) pos += (int)__popcll(__ballot(K[ r ] < k) & live[ r ]);
                if(pos >= ef) continue;  // the radius moved in since `todo` was taken
                const int r0 = pos >> 6, l0 = pos & 63;
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
            // pop: the first unexpanded key (the node every other wave chose); mark it; publish the next hop's front and radius
            int first = -1, fr = 0;
#pragma unroll
            for(int r = 0; r < KPL; ++r) {
                const unsigned long long m = __ballot(!key_expanded(K[ r ])) & live[ r ];
                if(first < 0 && m) {
                    first = (int)__builtin_ctzll(m);
                    fr = r;
                }
            }
#pragma unroll
            for(int r = 0; r < KPL; ++r)
                if(r == fr && lane == first) K[ r ] |= 1ull;  // expanded
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
                front_pub[ prv ] = nf;  // (hop + 1) & 1
                worst_pub[ prv ] = nw;
            }
        }
        if(fill_wave && c.cache_entries) {
            // the lists of the previous hop's keys that can still be expanded (inside the radius the hop started with: a superset
            // of what the merge keeps) move from the staging area to the cache.  One wave writes the cache; the entry of the node
            // being expanded right now is left alone (slower waves may still be reading it).
            unsigned long long todo = __ballot(N < w);
            const uint32_t     busy = node & (c.cache_entries - 1);
            while(todo) {
                const int tt = (int)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)N, tt) >> 1;
                const uint32_t e = slot & (c.cache_entries - 1);
                if(e == busy || slot == node) continue;
                if(lane < (int)M0) c.clist[ (size_t)e * M0 + (uint32_t)lane ] = c.stage[ ((size_t)prv * M0 + (size_t)tt) * M0 + (uint32_t)lane ];
                if(lane == 0) c.ctag[ e ] = slot;
            }
        }
        LGPU_SMARK(3)
        // ---- distances -> this hop's keys (all neighbours; the mask sorts out the old ones)
        if(row_wave) {
            uint64_t *const kout = keys0 + (ptrdiff_t)par * kstride;
            uint32_t *const sout = c.stage ? c.stage + (size_t)par * M0 * M0 : nullptr;
            spec_consume<METRIC, G, ROWS, U>(v, s, c, p, count, gl, qn2, kout, sout);
            for(int base = ngroups * ROWS; base < count; base += ngroups * ROWS) {  // lists longer than one pass covers
                spec_issue<METRIC, G, ROWS, U>(v, c, p, nb, count, base, group, ngroups, gl);
                spec_consume<METRIC, G, ROWS, U>(v, s, c, p, count, gl, qn2, kout, sout);
            }
        }
        if constexpr(PROF) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        LGPU_SMARK(4)
        __syncthreads();
        LGPU_SMARK(5)
        pacc[ 6 ] += 1;
    }
#undef LGPU_SMARK
    if constexpr(PROF) {
        if(prof && lane == 0 && wv < 4) {
            pacc[ 7 ] = wv == 0 ? src_stage : wv == 3 ? src_cache : wv == 2 ? src_hbm : 0;  // (the dedicated list wave looks nothing up)
            for(int i = 0; i < 8; ++i) atomicAdd(&prof[ 8 * wv + i ], pacc[ i ]);
        }
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

template <int METRIC, int G, int KPL, int ROWS, int U, bool DED, bool PROF = false>
__device__ int search_level_spec(const View &v, WalkLds &s, const SpecLds &c, uint32_t *bitmap, uint32_t bm_words, uint32_t start, int ef, uint32_t &D,
                                 uint32_t &E, unsigned long long *prof = nullptr)
{
#define LGPU_SPEC_ROLE(R) return search_level_spec_impl<METRIC, G, KPL, ROWS, U, DED, PROF, R>(v, s, c, bitmap, bm_words, start, ef, D, E, prof)
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
    if constexpr(PROF) {
        LGPU_SPEC_ROLE(-1);
    } else if constexpr(DED) {  // (at least four waves: three role waves, then row waves)
        if(wv == 0) LGPU_SPEC_ROLE(1);
        if(wv == 1) LGPU_SPEC_ROLE(2);
        if(wv == 2) LGPU_SPEC_ROLE(4);
        LGPU_SPEC_ROLE(8);
    } else {
        // (four waves that all evaluate rows: per-role instantiations were measured too -- 107 -> 30..77 reloads per pass, no gain on
        // the 1024-query batch it is meant for, twice the compile time -- and left out)
        (void)wv;
        LGPU_SPEC_ROLE(-1);
    }
#undef LGPU_SPEC_ROLE
}

}  // namespace lgpu
