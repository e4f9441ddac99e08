// walk_solo.hpp -- the ONE-WAVE walk: a lone query (usearch_search_ef, one call per scan: lantern_hnsw/src/hnsw/scan.c:220-228)
// walked by a single wavefront, with no workgroup barrier and no hand-over between waves anywhere.
//
// walk_spec.hpp (three role waves + eight row waves, one barrier per hop) left a lone hop at ~3.3 k cycles of which ~1.0-1.3 k
// were every wave WAITING at the barrier and most of the rest dependent LDS round trips between the roles.  Here
// one wave owns the whole hop, for rows of fewer than 64 chunks (d <= 252 f32) and lists of at most 32 neighbours:
//   * the candidate list is the sorted register list of walk.hpp (one key per lane, ef <= 64);
//   * ALL neighbours' rows of the expanded node are requested at once -- G lanes per row (the index's own lanes-per-row, so every
//     distance has the bits of every other kernel), 64 / G groups, 32 / (64 / G) rows per group -- before anything else is decided;
//   * for G <= 16 the DPP butterfly leaves the complete sum in EVERY lane of the group, so the lane that owns neighbour j picks its
//     distance out of its own registers: no LDS transpose, no bpermute;
//   * the visited set is a bitmap in LDS private to the wave: one ds_or_rtn per neighbour, no hash probe, nothing in HBM;
//   * every row request also asks for that row's own level-0 list; the lists of keys inside the radius go to a direct-mapped LDS
//     cache (tag + 128 bytes), so the expanded node's list is an LDS read in most hops instead of a dependent HBM round trip;
//     which of two neighbours that hash to one entry may write it is settled by an LDS exchange on a per-hop stamp;
//   * the merge of a hop's keys into the list happens in the shadow of the NEXT hop's row loads: the next node is
//     min(first unexpanded list entry, smallest new key inside the radius), which needs no merged list.
// Which node is expanded when, what the visited set holds at every filter and the order keys enter the list are the oracle's:
// ids, distance bits, D and E are those of every other launch shape (tests/test_gpu_parity.py::test_one_wave_walk_is_the_oracle_walk).
// MEASURED SLOWER than the 3 + 8 wave shape it was meant to beat (117.7 against 102.0 us per lone 100k x 128 query): one wave has to
// issue a hop's ~540 instructions itself.  On request only (LANTERN_GPU_SPEC=4); DESIGN.md 4.3c has the section profile.
#pragma once
#include "../device_common.hpp"
#include "../walk.hpp"

namespace lgpu {

typedef __attribute__((address_space(3))) uint32_t *LdsU32;  // LDS words: ds_read / ds_write / ds atomics, never flat
typedef uint32_t lgpu_u32x4 __attribute__((ext_vector_type(4)));  // (HIP's uint4 is a class: no assignment through an address-space pointer)
typedef __attribute__((address_space(3))) lgpu_u32x4 *LdsU128;
__device__ __forceinline__ uint4 lds_ld16(LdsU128 p) { const lgpu_u32x4 w = *p; return make_uint4(w.x, w.y, w.z, w.w); }
__device__ __forceinline__ void  lds_st16(LdsU128 p, const uint4 &w) { *p = lgpu_u32x4{ w.x, w.y, w.z, w.w }; }
typedef __attribute__((address_space(3))) uint64_t *LdsU64;

struct SoloLds
{
    LdsU64   keys;    // [64] the result, for the kernel's output section
    LdsU32   tags;    // [ne] slot whose list the cache entry holds (EMPTY = none)
    LdsU32   stamps;  // [ne] the hop that last claimed the entry
    LdsU32   data;    // [ne][32] neighbour lists, EMPTY padded
    LdsU32   bitmap;  // [bm_words] visited bits of the running query
    uint32_t ne_log2;
    uint32_t bm_words;  // multiple of 4
};
__host__ __device__ inline size_t solo_lds_bytes(uint32_t ne, uint32_t bm_words) { return 64 * 8 + (size_t)ne * (4 + 4 + 128) + (size_t)bm_words * 4; }
__device__ __forceinline__ void   carve_solo(unsigned char *p, SoloLds &s, uint32_t ne_log2, uint32_t bm_words)
{
    const uint32_t ne = 1u << ne_log2;
    s.keys = (LdsU64)p;    p += 64 * 8;
    s.data = (LdsU32)p;    p += (size_t)ne * 128;
    s.tags = (LdsU32)p;    p += (size_t)ne * 4;
    s.stamps = (LdsU32)p;  p += (size_t)ne * 4;
    s.bitmap = (LdsU32)p;
    s.ne_log2 = ne_log2;
    s.bm_words = bm_words;
}

__device__ __forceinline__ uint32_t solo_hash(uint32_t id, uint32_t ne_log2) { return (id * 0x9E3779B1u) >> (32u - ne_log2); }

// x[idx] for a lane-varying idx out of REGISTERS: is[i] = all ones where idx == i, made once per kernel and hidden from the optimiser
// behind an empty asm; the pick is r |= bits(x[i]) & is[i] -- one v_and_or_b32 per element.  (A select chain LLVM turns back into an
// indexed array -- a scratch-memory round trip per hop -- and selects on hoisted compares cost a 64-bit scalar mask each: 32 SGPRs.)
template <int N> struct LanePick
{
    uint32_t is[ N ];
    __device__ __forceinline__ void set(int idx)
    {
#pragma unroll
        for(int i = 0; i < N; ++i) {
            is[ i ] = idx == i ? 0xFFFFFFFFu : 0u;
            asm volatile("" : "+v"(is[ i ]));
        }
    }
};

// a 16-byte load at (uniform base) + (32-bit byte offset): the scalar-base form of global_load, one 32-bit multiply-add per address
// instead of 64-bit arithmetic per lane (the host admits an index here only if every table is below 4 GB)
__device__ __forceinline__ uint4    ld16(const void *base, uint32_t off) { return *(const uint4 *)((const char *)base + off); }
__device__ __forceinline__ uint32_t ld4(const void *base, uint32_t off) { return *(const uint32_t *)((const char *)base + off); }
__device__ __forceinline__ uint4    empty4() { return make_uint4(EMPTY, EMPTY, EMPTY, EMPTY); }
__device__ __forceinline__ uint4    sel4(bool c, const uint4 &a, const uint4 &b) { return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }

// smallest 64-bit key of the wave below `bound` (`bound` itself if none); every lane gets it
__device__ __forceinline__ uint64_t wave_min_below(uint64_t key, uint64_t bound, bool &got)
{
    uint64_t           t = bound;
    unsigned long long m = __ballot(key < t);
    while(m) {  // each round at least halves the expected number of smaller keys
        t = readlane64(key, (int)__builtin_ctzll(m));
        got = true;
        m = __ballot(key < t);
    }
    return t;
}

// RAGGED: chunks is not a multiple of G (the last chunk of a lane's share may lie past the row: read a valid chunk, use zeros)
template <int METRIC, int G, int CPL, bool RAGGED> struct SoloWalk
{
    static_assert(G == 8 || G == 16, "every lane of a group must end up with the complete sum: G <= 16");
    static constexpr int NG = 64 / G;    // groups per wave
    static constexpr int RPG = 32 / NG;  // rows per group per hop (= G / 2): 32 neighbours in one pass
    static constexpr int IV = RPG / 4;   // uint4 loads that cover a group's ids

    const View &v;
    SoloLds    &s;
    int         lane, g, gl;
    uint32_t    row_bytes, lane_off, last_off;  // bytes of a row; this lane's first chunk; its last chunk (clamped into the row)
    bool        last_ok;
    uint4       Q[ CPL ];
    float       qn;  // sqrt(||q||^2), cosine only
    LanePick<RPG> pick_mine, pick_pair;  // this lane's own neighbour (gl & (RPG - 1)) / the neighbour whose list it helps to cache (gl >> 1)

    __device__ __forceinline__ SoloWalk(const View &v_, SoloLds &s_) : v(v_), s(s_)
    {
        lane = (int)(threadIdx.x & 63);
        g = lane / G;
        gl = lane % G;
        pick_mine.set(gl & (RPG - 1));
        pick_pair.set(gl >> 1);
        row_bytes = v.chunks * 16u;
        lane_off = (uint32_t)gl * 16u;
        const int last = gl + (CPL - 1) * G;
        last_ok = !RAGGED || last < (int)v.chunks;
        last_off = (uint32_t)(last_ok ? last : gl) * 16u;  // (CPL = 1: every lane below `chunks`... or lane 0's chunk, masked)
        if(RAGGED && CPL == 1 && !last_ok) last_off = 0u;
    }

    __device__ __forceinline__ void load_query(const uint4 *q)
    {
#pragma unroll
        for(int c = 0; c < CPL; ++c) {
            const int ch = gl + c * G;
            Q[ c ] = ch < (int)v.chunks ? q[ ch ] : make_uint4(0, 0, 0, 0);
        }
        qn = 0.f;
        if constexpr(kCachedNorms<METRIC>) {
            NormAcc<METRIC> a;
#pragma unroll
            for(int c = 0; c < CPL; ++c)
                if(gl + c * G < (int)v.chunks) a.add(Q[ c ]);
            qn = __builtin_sqrtf(group_sum<G>(a.s));
        }
    }

    // the ids this lane needs of a list: its group's RPG rows, the neighbour whose key it will hold (gl < RPG), the neighbour whose
    // list it helps to cache (pair gl >> 1).  Entries at or past `cap` (a multiple of 4, <= 32) are EMPTY.
    struct Ids
    {
        uint32_t row[ RPG ], mine, pair;
    };
    __device__ __forceinline__ Ids ids_lds(uint32_t e) const  // cache entry e: 32 words, EMPTY padded
    {
        Ids           r;
        const LdsU32  list = s.data + e * 32u + (uint32_t)(g * RPG);
#pragma unroll
        for(int i = 0; i < IV; ++i) {
            const uint4 w = lds_ld16((LdsU128)(list + 4 * i));
            r.row[ 4 * i ] = w.x; r.row[ 4 * i + 1 ] = w.y; r.row[ 4 * i + 2 ] = w.z; r.row[ 4 * i + 3 ] = w.w;
        }
        r.mine = list[ gl & (RPG - 1) ];
        r.pair = list[ gl >> 1 ];
        return r;
    }
    __device__ __forceinline__ Ids ids_hbm(const void *base, uint32_t off, uint32_t cap) const  // list of `cap` words at base + off
    {
        Ids            r;
        const uint32_t j0 = (uint32_t)(g * RPG);
#pragma unroll
        for(int i = 0; i < IV; ++i) {
            const uint32_t j = j0 + 4 * i;
            const bool     in = j < cap;
            const uint4    w = sel4(in, ld16(base, off + (in ? j : 0u) * 4u), empty4());
            r.row[ 4 * i ] = w.x; r.row[ 4 * i + 1 ] = w.y; r.row[ 4 * i + 2 ] = w.z; r.row[ 4 * i + 3 ] = w.w;
        }
        const uint32_t jm = j0 + (uint32_t)(gl & (RPG - 1)), jp = j0 + (uint32_t)(gl >> 1);
        const uint32_t wm = ld4(base, off + (jm < cap ? jm : 0u) * 4u), wp = ld4(base, off + (jp < cap ? jp : 0u) * 4u);
        r.mine = jm < cap ? wm : EMPTY;
        r.pair = jp < cap ? wp : EMPTY;
        return r;
    }
    __device__ __forceinline__ void hide_keys(Ids &r) const  // lanes gl >= RPG hold no key
    {
        if(gl >= RPG) r.mine = EMPTY;
    }

    // the RPG rows of this lane's group against the query, in two steps so that the caller can put work between the requests and
    // their use: rows_issue asks for every chunk; rows_reduce forms the raw G-lane sums (l2sq: the distance; cos: the ab chain),
    // complete in every lane of the group -- each lane keeps the sum of ITS neighbour (gl & (RPG - 1)) and of the neighbour whose
    // list it helps to cache (gl >> 1), picked as the sums are formed (values, never an array: an indexed array would live in
    // scratch memory).  EMPTY ids read row 0 and are ignored by the caller.
    struct Rows
    {
        uint4 r[ RPG ][ CPL ];
    };
    __device__ __forceinline__ void rows_issue(const Ids &id, Rows &R) const
    {
#pragma unroll
        for(int p = 0; p < RPG; ++p) {
            const uint32_t at = (id.row[ p ] == EMPTY ? 0u : id.row[ p ]) * row_bytes;
#pragma unroll
            for(int c = 0; c + 1 < CPL; ++c) R.r[ p ][ c ] = ld16(v.vec, at + lane_off + (uint32_t)(c * G * 16));
            R.r[ p ][ CPL - 1 ] = ld16(v.vec, at + last_off);
        }
    }
    __device__ __forceinline__ void rows_reduce(const Rows &R, float &sum_mine, float &sum_pair) const
    {
        uint32_t bm = 0u, bp = 0u;
#pragma unroll
        for(int p = 0; p < RPG; ++p) {
            RowAcc<METRIC> a;
#pragma unroll
            for(int c = 0; c + 1 < CPL; ++c) a.add(Q[ c ], R.r[ p ][ c ]);
            if constexpr(RAGGED) a.add(Q[ CPL - 1 ], sel4(last_ok, R.r[ p ][ CPL - 1 ], make_uint4(0, 0, 0, 0)));  // (Q is zero there too)
            else a.add(Q[ CPL - 1 ], R.r[ p ][ CPL - 1 ]);
            const uint32_t sp = __float_as_uint(group_sum<G>(a.s));
            bm |= sp & pick_mine.is[ p ];
            bp |= sp & pick_pair.is[ p ];
        }
        sum_mine = __uint_as_float(bm);
        sum_pair = __uint_as_float(bp);
    }
    __device__ __forceinline__ void eval(const Ids &id, float &sum_mine, float &sum_pair) const
    {
        Rows R;
        rows_issue(id, R);
        rows_reduce(R, sum_mine, sum_pair);
    }
    __device__ __forceinline__ float finish(float sum, float row_norm_rooted) const
    {
        if constexpr(kCachedNorms<METRIC>) return cos_finish_rooted(sum, qn, row_norm_rooted);
        else return sum;
    }
    __device__ __forceinline__ float norm_of(uint32_t id) const
    {
        if constexpr(kCachedNorms<METRIC>) return __uint_as_float(ld4(v.norm2, (id == EMPTY ? 0u : id) * 4u));
        else return 0.f;
    }

    // ---- search_for_one_: greedy descent over levels (max_level .. 1); returns the closest slot and its distance.
    __device__ uint32_t descend(uint32_t &D, float &d_out)
    {
        uint32_t cur = v.entry;
        Ids      one;
#pragma unroll
        for(int p = 0; p < RPG; ++p) one.row[ p ] = p == 0 ? cur : EMPTY;
        one.mine = one.pair = EMPTY;
        float       sm, sp;
        uint32_t    cur_uo = v.max_level > 0 ? v.upper_off[ cur ] : 0u;
        const float ncur = norm_of(cur);
        eval(one, sm, sp);
        float curd = finish(sp, ncur);  // (pair index 0 = row 0 in lanes 0 and 1 of every group)
        curd = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(curd)));
        D += 1;
        for(int level = v.max_level; level > 0; --level) {
            for(;;) {
                Ids id = ids_hbm(v.upper_nbr, (cur_uo + (uint32_t)(level - 1)) * v.M * 4u, v.M);
                hide_keys(id);
                const bool     have = id.mine != EMPTY;
                const uint32_t uo = ld4(v.upper_off, (have ? id.mine : 0u) * 4u);
                const float    nrm = norm_of(id.mine);
                eval(id, sm, sp);
                const float d = finish(sm, nrm);
                D += (uint32_t)__popcll(__ballot(have));
                const uint32_t j = (uint32_t)(g * RPG + (gl & (RPG - 1)));  // position in the list: first strictly closer wins = min (distance, position)
                const uint64_t key = have ? (((uint64_t)f2ord(d) << 32) | j) : ~0ull;
                bool           got = false;
                const uint64_t t = wave_min_below(key, ~0ull, got);
                if(!got) break;  // an empty list
                const float dmin = ord2f((uint32_t)(t >> 32));
                if(!(dmin < curd)) break;
                curd = dmin;
                const int jb = (int)(t & 31ull), holder = (jb / RPG) * G + jb % RPG;
                cur = (uint32_t)__builtin_amdgcn_readlane((int)id.mine, holder);
                cur_uo = (uint32_t)__builtin_amdgcn_readlane((int)uo, holder);
            }
        }
        d_out = curd;
        return cur;
    }

    // ---- search_to_find_in_base_: the ef-bounded walk over level 0 from `start` (whose distance the descent just took: the
    // oracle evaluates it once more here -- counted in D, not re-read).  Leaves the list in s.keys, returns its length.
    // PROF (the diagnostic instantiation): shader-clock cycles per section of a hop, summed over the query, into prof[0..8):
    // [0] list look-up (LDS cache / HBM)  [1] issuing the row, list and norm loads  [2] merge of the previous hop's keys (in the
    // loads' shadow)  [3] visited filter + cache claims  [4] waiting for the rows + the distances  [5] cache writes
    // [6] choosing the next node  [7] hops  [8] hops whose list was not in the LDS cache.  Every stamp costs an s_memtime + wait (~40 cycles).
    template <bool PROF = false>
    __device__ int level0(uint32_t start, float start_d, int ef, uint32_t &D, uint32_t &E, uint32_t &hop_ctr, unsigned long long *prof = nullptr)
    {
        unsigned long long tl = 0, pacc[ 8 ] = { 0, 0, 0, 0, 0, 0, 0, 0 }, misses = 0;
#define LGPU_SOLO_MARK(i)                                                   \
    if constexpr(PROF) {                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  \
        const unsigned long long t_ = (unsigned long long)clock64();        \
        pacc[ i ] += t_ - tl;                                               \
        tl = t_;                                                            \
    }
        const unsigned long long live = ef >= 64 ? ~0ull : (1ull << ef) - 1ull;
        const uint32_t           M0 = v.M0, list_bytes = M0 * 4u;
        const uint32_t           half = (uint32_t)(gl & 1);
        // visits.clear(); visits.set(start)
        for(uint32_t i = (uint32_t)lane; i < s.bm_words / 4; i += 64) lds_st16((LdsU128)s.bitmap + i, make_uint4(0, 0, 0, 0));
        uint64_t K = lane == 0 ? make_key(start_d, start) : ~0ull;  // lane l: the l-th smallest key, ~0 past the end
        uint64_t N = ~0ull;                                          // the previous hop's new keys, not merged yet
        int      cnt = 1;
        uint32_t node = start;
        D += 1;
        if(lane == 0) s.bitmap[ start >> 5 ] = 1u << (start & 31);
        if constexpr(PROF) tl = (unsigned long long)clock64();
        while(node != EMPTY) {
            E += 1;
            ++hop_ctr;
            // ---- [1] the node's list: the LDS cache (tag and ids read together), else HBM
            const uint32_t e = solo_hash(node, s.ne_log2);
            const uint32_t tag = s.tags[ e ];
            Ids            id = ids_lds(e);
            if((uint32_t)__builtin_amdgcn_readfirstlane((int)tag) != node) {
                id = ids_hbm(v.nbr0, node * list_bytes, M0);
                if constexpr(PROF) misses += 1;
            }
            hide_keys(id);
            if constexpr(PROF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a list that came from HBM is charged to the look-up)
            LGPU_SOLO_MARK(0)
            // ---- [2] every neighbour's row, its own list (two lanes per list, 64 bytes each) and norm: all requested now
            uint4 L[ 4 ];
            {
                const uint32_t at = (id.pair == EMPTY ? 0u : id.pair) * list_bytes;
#pragma unroll
                for(int i = 0; i < 4; ++i) {
                    const uint32_t w = half * 16u + 4u * (uint32_t)i;
                    const bool     in = w < M0;
                    L[ i ] = sel4(in, ld16(v.nbr0, at + (in ? w : 0u) * 4u), empty4());
                }
            }
            const float nrm_mine = norm_of(id.mine), nrm_pair = norm_of(id.pair);
            Rows        R;
            rows_issue(id, R);
            LGPU_SOLO_MARK(1)
            // ---- [3] in the shadow of the loads: merge the previous hop's keys, mark `node` expanded
            {
                const uint64_t     worst = cnt == ef ? readlane64(K, ef - 1) : ~0ull;
                unsigned long long todo = __ballot(N < worst);
                while(todo) {
                    const int t = (int)__builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    const uint64_t k = readlane64(N, t);
                    const int      p = (int)__popcll(__ballot(K < k) & live);
                    if(p >= ef) continue;  // the radius moved in since `todo` was taken
                    const uint64_t sh = wave_shr1(K);
                    if(lane > p) K = sh;
                    if(lane == p) K = k;
                    cnt = cnt < ef ? cnt + 1 : ef;
                }
                if(K != ~0ull && key_slot(K) == node) K |= 1ull;
            }
            LGPU_SOLO_MARK(2)
            // ---- [4] visited filter (one LDS atomic per neighbour) and the cache claims
            bool isnew = false;
            if(id.mine != EMPTY) {
                const uint32_t bit = 1u << (id.mine & 31);
                isnew = (__hip_atomic_fetch_or(s.bitmap + (id.mine >> 5), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & bit) == 0;
            }
            const uint32_t ep = solo_hash(id.pair == EMPTY ? 0u : id.pair, s.ne_log2);
            uint32_t       won = 0;
            if(half == 0 && id.pair != EMPTY && s.tags[ ep ] != id.pair)
                won = __hip_atomic_exchange(s.stamps + ep, hop_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != hop_ctr;
            won = dpp_take<0xA0, 0xF>(won);  // quad_perm [0,0,2,2]: the pair's even lane decides for both
            D += (uint32_t)__popcll(__ballot(isnew));
            LGPU_SOLO_MARK(3)
            // ---- [5] distances -> this hop's new keys
            float sum_mine, sum_pair;
            rows_reduce(R, sum_mine, sum_pair);
            const float d_mine = finish(sum_mine, nrm_mine);
            N = isnew ? make_key(d_mine, id.mine) : ~0ull;
            if constexpr(PROF) asm volatile("" ::"v"(d_mine));
            LGPU_SOLO_MARK(4)
            // ---- [6] lists of keys inside the radius -> cache
            {
                const float    d_pair = finish(sum_pair, nrm_pair);
                const uint64_t worst = cnt == ef ? readlane64(K, ef - 1) : ~0ull;
                if(won && make_key(d_pair, id.pair) < worst) {
                    const LdsU128 dst = (LdsU128)(s.data + ep * 32u) + half * 4u;
#pragma unroll
                    for(int i = 0; i < 4; ++i) lds_st16(dst + i, L[ i ]);
                    if(half == 0) s.tags[ ep ] = id.pair;
                }
            }
            LGPU_SOLO_MARK(5)
            // ---- [7] the next node: min(first unexpanded entry of the list, smallest new key inside the radius)
            {
                const unsigned long long m = __ballot(!key_expanded(K)) & live;
                const uint64_t           f = m ? readlane64(K, (int)__builtin_ctzll(m)) : ~0ull;
                const uint64_t           w = cnt == ef ? readlane64(K, ef - 1) : ~0ull;
                bool                     got = f != ~0ull;
                const uint64_t           t = wave_min_below(N, got ? f : w, got);
                node = got ? key_slot(t) : EMPTY;
            }
            LGPU_SOLO_MARK(6)
        }
#undef LGPU_SOLO_MARK
        if constexpr(PROF) {
            if(lane == 0 && prof) {
                for(int i = 0; i < 7; ++i) atomicAdd(&prof[ i ], pacc[ i ]);
                atomicAdd(&prof[ 7 ], (unsigned long long)E);
                atomicAdd(&prof[ 8 ], misses);  // hops whose list came from HBM (not in the LDS cache)
            }
        }
        if(lane < cnt) s.keys[ lane ] = K;
        return cnt;
    }
};

}  // namespace lgpu
