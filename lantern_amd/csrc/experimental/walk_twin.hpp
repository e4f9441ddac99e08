// walk_twin.hpp -- the latency-bound base-layer walk with TWO nodes per round: the one the walk expands next, and the one it will
// expand after that if the first one's neighbours bring nothing closer.
//
// search_level_spec (walk_spec.hpp) took everything off a hop's critical path that is not a true dependency; what is left is one
// memory round trip per hop -- the rows of the expanded node's neighbours -- plus the wait for the slowest of those ~28 random
// rows.  A walk that is alone on its CU has bandwidth to burn, so this form spends it on that dependency:
//
//   * A round knows the list's first THREE unexpanded keys (published by the list wave) and the keys of the node(s) the previous
//     round evaluated.  It picks x, the node to expand -- min(front, smallest pending key inside the radius), the rule of
//     search_level_reg -- and y, the runner-up among the same candidates: the node that WILL be the next one expanded unless one
//     of x's own neighbours turns out closer than y (measured on the oracle: the list's front is the next pop in 51 % of hops).
//   * The rows of x's AND of y's neighbours are requested together.  x's neighbours go through the visited filter as usual
//     (test and set); y's are only TESTED against the set as it stands after x's -- nothing about y is recorded anywhere yet.
//   * The next round decides again, from scratch, which node comes next.  If it is y (a "hit"), y's expansion is already done:
//     its neighbours' keys are in LDS, its tested-new neighbours are recorded in the visited set now, D and E advance as the
//     oracle's do, and the round goes straight on to the node after y -- two expansions for one memory round trip.  If it is
//     not y, y's keys are dropped (its rows are warm in L2 for when its turn comes) and nothing else happened.
//
// Same walk, same results: which node is expanded when, what the visited set holds when a node's neighbours are filtered, and
// the order in which keys enter the list are exactly search_level_reg's -- speculation only changes WHEN rows are read.  ids,
// distance bits, D and E equal the oracle's (tests/test_gpu_parity.py runs this form beside the others).
// Dedicated role waves only (the lone-query shape: visit | list | fill | row waves); requires what search_level_spec requires.
#pragma once
#include "../walk_spec.hpp"

namespace lgpu {

struct KeyPick
{
    uint64_t key;   // ~0: none
    int      lane;  // the lane that held it
};
// the smallest key over the wave's lanes (~0 in lanes without a candidate); the same answer in every lane
__device__ __forceinline__ KeyPick wave_min_key(uint64_t k)
{
    const uint32_t     hi = (uint32_t)(k >> 32);
    const uint32_t     mh = wave_min_u32(hi);
    unsigned long long m = __ballot(k != ~0ull && hi == mh);
    KeyPick            r{ ~0ull, -1 };
    if(m) {
        r.lane = (int)__builtin_ctzll(m);
        r.key = readlane64(k, r.lane);
        m &= m - 1ull;
        while(m) {  // exact ties in distance: the smaller slot wins (keys are (distance, slot))
            const int      j2 = (int)__builtin_ctzll(m);
            const uint64_t t2 = readlane64(k, j2);
            if(t2 < r.key) { r.key = t2; r.lane = j2; }
            m &= m - 1ull;
        }
    }
    return r;
}

// The level-0 list of `node` without a dependent HBM load if it can be had: staged last round with the row of the key it was
// (set 1 / 2, neighbour index j), or cached since an earlier round.  `fetch`: read it from HBM otherwise; without it the
// function reports false and the caller does without the node.
__device__ __forceinline__ bool twin_list(const View &v, const SpecLds &c, uint32_t node, int set, int j, int prv, int lane, bool fetch, uint32_t &nb,
                                          unsigned &from_stage, unsigned &from_cache, unsigned &from_hbm)
{
    const uint32_t M0 = v.M0;
    nb = EMPTY;
    if(c.stage && set == 1) {
        if(lane < (int)M0) nb = c.stage[ ((size_t)prv * M0 + (size_t)j) * M0 + (uint32_t)lane ];
        from_stage += 1;
        return true;
    }
    if(c.stage2 && set == 2) {
        if(lane < (int)M0) nb = c.stage2[ ((size_t)prv * M0 + (size_t)j) * M0 + (uint32_t)lane ];
        from_stage += 1;
        return true;
    }
    if(c.cache_entries) {
        const uint32_t e = node & (c.cache_entries - 1);
        const uint32_t tag = c.ctag[ e ];
        const uint32_t ent = lane < (int)M0 ? c.clist[ (size_t)e * M0 + (uint32_t)lane ] : EMPTY;
        if((uint32_t)__builtin_amdgcn_readfirstlane((int)tag) == node) {
            nb = ent;
            from_cache += 1;
            return true;
        }
    }
    if(!fetch) return false;
    if(lane < (int)M0) nb = v.nbr0[ (size_t)node * M0 + (uint32_t)lane ];
    from_hbm += 1;
    return true;
}

// ROLE >= 0: the instantiation for one wave's role -- bits 1 visit | 2 list | 4 fill | 8 rows -- as in search_level_spec_impl;
// ROLE < 0 (the diagnostic instantiation): all of them, by wave index.  PROF as there.
template <int METRIC, int G, int KPL, int ROWS, int U, bool PROF, int ROLE>
__device__ int search_level_twin_impl(const View &v, WalkLds &s, const SpecLds &c, uint32_t *bitmap, uint32_t bm_words, uint32_t start, int ef, uint32_t &D,
                                      uint32_t &E, unsigned long long *prof)
{
    unsigned long long pacc[ 8 ] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tl = 0;
    unsigned           src_stage = 0, src_cache = 0, src_hbm = 0, hits = 0;
#define LGPU_TMARK(i)                                                      \
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
    const bool    fill_wave = ROLE >= 0 ? (ROLE & 4) != 0 : wv == 2;
    const bool    row_wave = ROLE >= 0 ? (ROLE & 8) != 0 : wv >= 3;
    const int     ngroups = (NW - 3) * GPW;    // G-lane groups that evaluate rows
    const int     group = (wv - 3) * GPW + g;  // this lane's group among them
    const uint32_t M0 = v.M0;
    for(uint32_t i = tid; i < s.vis_slots; i += T) s.vis[ i ] = EMPTY;  // (the HBM bitmap is all-zero between walks: walk.hpp VisUndo)
    for(uint32_t i = tid; i < c.cache_entries; i += T) c.ctag[ i ] = EMPTY;
    const float     qn2 = __int_as_float(s.scal[ S_QN2 ]);
    uint64_t *const k1b[ 2 ] = { s.newkeys, s.sorted };        // the keys of a round's certain node, by round parity
    uint64_t *const k2b[ 2 ] = { c.keys2, c.keys2 + M0 };      // ... of its speculative node
    // "round -1" (parity 1) evaluated one row: the start node
    if(wv == 0 && g == 0) {
        const float d = group_dist_n<METRIC, G>(walk_query<METRIC>(s), row_of_m<METRIC>(v, start), (int)v.chunks, gl, qn2, row_norm<METRIC>(v, start));
        if(gl == G - 1) k1b[ 1 ][ 0 ] = make_key(d, start);
        constexpr int LW = spec_list_words<G>();
        if(c.stage && gl * LW < (int)M0) {
            uint32_t piece[ 4 ];
            spec_list_load<LW>(v.nbr0 + (size_t)start * M0 + (size_t)(gl * LW), piece);
            spec_list_store<LW>(c.stage + (size_t)M0 * M0 + (size_t)(gl * LW), piece);
        }
    }
    if(tid == 0) {
        c.tw[ TW_STRIDE + TW_MASK1 ] = 1ull;
        c.tw[ TW_STRIDE + TW_MASK2 ] = 0ull;
        c.tw[ TW_F0 ] = c.tw[ TW_F0 + 1 ] = c.tw[ TW_F0 + 2 ] = ~0ull;  // the list is still empty: no front,
        c.tw[ TW_W ] = ~0ull;                                              // no radius
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
    int      cnt = 0;
    // every wave's: last round's speculative node (EMPTY: none) and its neighbour list, one slot per lane
    uint32_t ynode = EMPTY, nby_prev = EMPTY;
    if constexpr(PROF) tl = (unsigned long long)clock64();
    for(int round = 0;; ++round) {
        const int             par = round & 1, prv = par ^ 1;
        const uint64_t *const tin = c.tw + (size_t)par * TW_STRIDE, *const tprev = c.tw + (size_t)prv * TW_STRIDE;
        // ---- what the previous round left
        const uint64_t           f0 = uniform64(tin[ TW_F0 ]), f1 = uniform64(tin[ TW_F0 + 1 ]), f2 = uniform64(tin[ TW_F0 + 2 ]), w = uniform64(tin[ TW_W ]);
        const unsigned long long pm1 = uniform64(tprev[ TW_MASK1 ]);
        const unsigned long long pm2 = ynode != EMPTY ? uniform64(tprev[ TW_MASK2 ]) : 0ull;
        const uint64_t           N1 = ((pm1 >> lane) & 1ull) ? k1b[ prv ][ lane ] : ~0ull;
        uint64_t                 N2 = ((pm2 >> lane) & 1ull) ? k2b[ prv ][ lane ] : ~0ull;
        D += (uint32_t)__popcll(pm1);
        // ---- the node whose turn it is: min(front, smallest pending key below it / inside the radius)
        const uint64_t t0 = f0 != ~0ull ? f0 : w;
        const bool     closer = __ballot(N1 < t0) != 0ull;
        if(f0 == ~0ull && !closer) break;  // every wave sees the same values: all leave together
        E += 1;
        // Was it last round's speculative node?  Then its expansion is done: its neighbours' keys are N2, and the candidates for
        // the next node are what is left of the list's front entries, N1 and N2.
        const bool hit = ynode != EMPTY && !closer && f0 != ~0ull && key_slot(f0) == ynode;
        uint64_t   P0 = f0, P1 = f1;
        if(hit) {
            D += (uint32_t)__popcll(pm2);
            P0 = f1;
            P1 = f2;
            hits += 1;
        } else {
            N2 = ~0ull;
        }
        // ---- x: the first of the candidates (on a hit: the node AFTER the one just completed)
        const uint64_t tp = P0 != ~0ull ? P0 : w;
        const uint64_t c1 = N1 < tp ? N1 : ~0ull, c2 = N2 < tp ? N2 : ~0ull;
        const bool     two = c2 < c1;
        const KeyPick  a = wave_min_key(two ? c2 : c1);
        if(hit) {
            if(P0 == ~0ull && a.key == ~0ull) break;  // the walk ends with the node just completed
            E += 1;
        }
        const bool     x_front = a.key == ~0ull;
        const uint64_t xkey = x_front ? P0 : a.key;
        const uint32_t xnode = key_slot(xkey);
        const int      xset = x_front ? 0 : (__builtin_amdgcn_readlane((int)two, a.lane) ? 2 : 1);
        // ---- y: the runner-up -- what the list's front will be once x is popped and the pending keys are merged
        uint64_t ykey;
        int      yset = 0, yj = -1;
        {
            const uint64_t d1 = N1 < w ? N1 : ~0ull, d2 = N2 < w ? N2 : ~0ull;  // every pending key inside the radius
            uint64_t       m2;
            if(x_front) m2 = d2 < d1 ? d2 : d1;
            else m2 = lane == a.lane ? (xset == 2 ? d1 : d2) : (d2 < d1 ? d2 : d1);  // x's own lane: the key of the other set
            const KeyPick  b = wave_min_key(m2);
            const uint64_t fr = x_front ? P1 : P0;  // the best of the list's entries still standing
            if(b.key < fr) {
                ykey = b.key;
                yj = b.lane;
                const uint64_t mine1 = d1;
                yset = (readlane64(mine1, b.lane) == b.key) ? 1 : 2;
            } else {
                ykey = fr;
            }
        }
        LGPU_TMARK(0)
        // ---- their neighbour lists.  x's comes from HBM if it is nowhere in LDS; y is only worth its rows if its list is at hand
        uint32_t nbx = EMPTY, nby = EMPTY;
        (void)twin_list(v, c, xnode, xset, a.lane, prv, lane, true, nbx, src_stage, src_cache, src_hbm);
        uint32_t ynew = EMPTY;
        if(ykey != ~0ull) {
            unsigned dummy = 0;
            ynew = key_slot(ykey);
            if(!twin_list(v, c, ynew, yset, yj, prv, lane, false, nby, dummy, dummy, dummy)) ynew = EMPTY;
        }
        const int countx = (int)__popcll(__ballot(nbx != EMPTY));  // lists are EMPTY-terminated and hole-free
        const int county = ynew != EMPTY ? (int)__popcll(__ballot(nby != EMPTY)) : 0;
        LGPU_TMARK(1)
        // ---- the rows of ALL neighbours of both (old ones too: the filters run behind the loads)
        SpecPass<ROWS, U> p1, p2;
        p1.have = p2.have = false;
        if(row_wave) {
            spec_issue<METRIC, G, ROWS, U>(v, c, p1, nbx, countx, 0, group, ngroups, gl);
            if(county) spec_issue<METRIC, G, ROWS, U>(v, c, p2, nby, county, 0, group, ngroups, gl);
        }
        LGPU_TMARK(2)
        // ---- in the shadow of the loads: the three role sections
        if(visit_wave) {
            // the LDS set must keep room for the lists of this round (a completed y's and x's); otherwise spill to the bitmap
            if(s.vis_slots && !spilled && viscnt + 2 * M0 > s.vis_slots / 4 * 3) spilled = true;  // (the bitmap is all-zero: VisUndo)
            const bool to_bitmap = spilled || !s.vis_slots;
            if(hit) {  // the node just completed: its neighbours that tested new last round are visited from now on
                bool rec = false;
                if((pm2 >> lane) & 1ull) rec = !visit_test_and_set(s, bitmap, nby_prev, spilled);
                if(to_bitmap) undo_record(s, undo, rec, nby_prev, __ballot(rec), lane);
                if(s.vis_slots && !spilled) viscnt += (uint32_t)__popcll(pm2);
            }
            const bool               isnew = hop_is_new(s, bitmap, nbx, spilled);
            const unsigned long long nm = __ballot(isnew);
            if(s.vis_slots && !spilled) viscnt += (uint32_t)__popcll(nm);
            if(to_bitmap) undo_record(s, undo, isnew, nbx, nm, lane);
            unsigned long long nm2 = 0ull;
            if(county) nm2 = __ballot(nby != EMPTY && !visit_test(s, bitmap, nby, spilled));
            if(lane == 0) {
                c.tw[ (size_t)par * TW_STRIDE + TW_MASK1 ] = nm;
                c.tw[ (size_t)par * TW_STRIDE + TW_MASK2 ] = nm2;
            }
        }
        if(list_wave) {
            // merge the pending keys inside the radius, one at a time into the sorted registers (rank = one ballot, insertion =
            // one wave-wide DPP shift): N1, then -- the node they belong to having been completed -- N2
#pragma unroll
            for(int pass = 0; pass < 2; ++pass) {
                const uint64_t Np = pass == 0 ? N1 : N2;
                if(pass == 1 && !hit) break;
                uint64_t worst = ~0ull;
                if(cnt == ef) {
                    const int wl = (ef - 1) & 63;
#pragma unroll
                    for(int r = 0; r < KPL; ++r)
                        if(r == (ef - 1) >> 6) worst = readlane64(K[ r ], wl);
                }
                unsigned long long todo = __ballot(Np < worst);
                while(todo) {
                    const int tt = (int)__builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    const uint64_t k = readlane64(Np, tt);
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
                // pop: the first unexpanded key -- the node completed this round (pass 0 of a hit: last round's y, whose keys
                // pass 1 merges next) or the node every other wave is loading the rows of (x)
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
            }
            // publish the next round's first three unexpanded keys and its radius
            uint64_t nf[ 3 ] = { ~0ull, ~0ull, ~0ull }, nw = ~0ull;
            int      have = 0;
#pragma unroll
            for(int r = 0; r < KPL; ++r) {
                unsigned long long m = __ballot(!key_expanded(K[ r ])) & live[ r ];
                while(m && have < 3) {
                    nf[ have++ ] = readlane64(K[ r ], (int)__builtin_ctzll(m));
                    m &= m - 1ull;
                }
                if(cnt == ef && r == (ef - 1) >> 6) nw = readlane64(K[ r ], (ef - 1) & 63);
            }
            if(lane == 0) {
                uint64_t *const tout = c.tw + (size_t)prv * TW_STRIDE;  // (round + 1) & 1
                tout[ TW_F0 ] = nf[ 0 ];
                tout[ TW_F0 + 1 ] = nf[ 1 ];
                tout[ TW_F0 + 2 ] = nf[ 2 ];
                tout[ TW_W ] = nw;
            }
        }
        if(fill_wave && c.cache_entries) {
            // the lists of the pending keys that can still be expanded (inside the radius the round started with: a superset of
            // what the merge keeps) move from the staging areas to the cache.  One wave writes the cache; the entries of the two
            // nodes being looked up right now are left alone (slower waves may still be reading them).
            const uint32_t busy1 = xnode & (c.cache_entries - 1), busy2 = ynew != EMPTY ? (ynew & (c.cache_entries - 1)) : busy1;
#pragma unroll
            for(int pass = 0; pass < 2; ++pass) {
                const uint64_t        Np = pass == 0 ? N1 : N2;
                const uint32_t *const st = pass == 0 ? c.stage : c.stage2;
                if(!st) continue;
                unsigned long long todo = __ballot(Np < w);
                while(todo) {
                    const int tt = (int)__builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)Np, tt) >> 1;
                    const uint32_t e = slot & (c.cache_entries - 1);
                    if(e == busy1 || e == busy2) continue;
                    if(lane < (int)M0) c.clist[ (size_t)e * M0 + (uint32_t)lane ] = st[ ((size_t)prv * M0 + (size_t)tt) * M0 + (uint32_t)lane ];
                    if(lane == 0) c.ctag[ e ] = slot;
                }
            }
        }
        LGPU_TMARK(3)
        // ---- distances -> this round's keys (all neighbours; the masks sort out the old ones)
        if(row_wave) {
            uint64_t *const kout1 = k1b[ par ], *const kout2 = k2b[ par ];
            uint32_t *const sout1 = c.stage ? c.stage + (size_t)par * M0 * M0 : nullptr;
            uint32_t *const sout2 = c.stage2 ? c.stage2 + (size_t)par * M0 * M0 : nullptr;
            spec_consume<METRIC, G, ROWS, U>(v, s, c, p1, countx, gl, qn2, kout1, sout1);
            if(county) spec_consume<METRIC, G, ROWS, U>(v, s, c, p2, county, gl, qn2, kout2, sout2);
            for(int base = ngroups * ROWS; base < countx; base += ngroups * ROWS) {  // lists longer than one pass covers
                spec_issue<METRIC, G, ROWS, U>(v, c, p1, nbx, countx, base, group, ngroups, gl);
                spec_consume<METRIC, G, ROWS, U>(v, s, c, p1, countx, gl, qn2, kout1, sout1);
            }
            for(int base = ngroups * ROWS; base < county; base += ngroups * ROWS) {
                spec_issue<METRIC, G, ROWS, U>(v, c, p2, nby, county, base, group, ngroups, gl);
                spec_consume<METRIC, G, ROWS, U>(v, s, c, p2, county, gl, qn2, kout2, sout2);
            }
        }
        ynode = ynew;
        nby_prev = nby;
        if constexpr(PROF) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        LGPU_TMARK(4)
        __syncthreads();
        LGPU_TMARK(5)
        pacc[ 6 ] += 1;
    }
#undef LGPU_TMARK
    if constexpr(PROF) {
        if(prof && lane == 0 && wv < 4) {
            pacc[ 7 ] = wv == 0 ? src_stage : wv == 3 ? src_cache : wv == 2 ? src_hbm : hits;  // (wave 1: rounds that completed a speculative node)
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

template <int METRIC, int G, int KPL, int ROWS, int U, bool PROF = false>
__device__ int search_level_twin(const View &v, WalkLds &s, const SpecLds &c, uint32_t *bitmap, uint32_t bm_words, uint32_t start, int ef, uint32_t &D,
                                 uint32_t &E, unsigned long long *prof = nullptr)
{
#define LGPU_TWIN_ROLE(R) return search_level_twin_impl<METRIC, G, KPL, ROWS, U, PROF, R>(v, s, c, bitmap, bm_words, start, ef, D, E, prof)
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >> 6;
    if constexpr(PROF) {
        LGPU_TWIN_ROLE(-1);
    } else {  // (at least four waves: three role waves, then row waves)
        if(wv == 0) LGPU_TWIN_ROLE(1);
        if(wv == 1) LGPU_TWIN_ROLE(2);
        if(wv == 2) LGPU_TWIN_ROLE(4);
        LGPU_TWIN_ROLE(8);
    }
#undef LGPU_TWIN_ROLE
}

}  // namespace lgpu
