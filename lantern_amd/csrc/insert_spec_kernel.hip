// insert_spec_kernel.hip -- k_insert in its latency-bound form: the walk of usearch_add when the batch is a handful of vectors --
// ldb_aminsert's one row at a time (lantern_hnsw/src/hnsw/insert.c:32-46, usearch_add_external) and the first batches of every
// build.  Level 0 -- where a new node almost always lives, and where the ef_construction-wide walk is -- runs walk_spec.hpp's
// lone-query shape (three role waves + eight row waves, one barrier per hop, neighbour lists fetched with the rows); the levels
// above it (one node in sixteen has any) keep walk.hpp's walk.  Same candidates in the same order as k_insert: the graph does
// not depend on which of the two ran (tests/: the build parity tests start from an empty index and pass through this kernel).
// f32 rows under l2sq / cos only; other storage kinds keep k_insert.
#ifndef LGPU_LIST_PREFETCH  // as insert_kernel.hip
#define LGPU_LIST_PREFETCH 0
#endif
#include "kernels.hpp"
#include "walk.hpp"
#include "walk_spec.hpp"

namespace lgpu {

template <int METRIC, int G, int KPL>
__global__ void __launch_bounds__(704, 1) k_insert_spec(InsertArgs a)
{
    constexpr int ROWS = G == 64 ? 4 : G == 32 ? 2 : 1, U = G == 64 ? 3 : 2;  // as the search kernels of this shape (search_spec_kernel.hip)
    const int     tid = threadIdx.x, T = blockDim.x;
    WalkLds       s;
    SpecLds       sc;
    {
        unsigned char *end = carve_walk(lgpu_smem, s, a.view.chunks, a.efc, a.view.M0, a.vis_slots);
        carve_spec(end, sc, a.view.M0, a.spec_prefetch, a.spec_cache);
    }
    uint32_t      *bitmap = a.bitmaps + (size_t)blockIdx.x * (a.bm_words + kVisUndoWords);
    s.undo = bitmap + a.bm_words;
    s.undo_cap = a.undo_cap;
    const uint32_t chunks = a.view.chunks, M = a.view.M;
    for(uint32_t b = a.b_begin + blockIdx.x; b < a.count;) {
        const uint32_t me = a.first_slot + b;
        const int      target = a.view.levels[ me ];
        const uint32_t item0 = a.link_off[ b ] / M;  // one item per (node, level)
        {
            const uint4 *own = row_of(a.view, me);
            for(uint32_t i = tid; i < chunks; i += T) s.q[ i ] = own[ i ];
            for(uint32_t i = tid; i <= (uint32_t)target; i += T) a.top_count[ item0 + i ] = 0;  // levels above max_level stay empty
            if(tid == 0) s.scal[ S_QN2 ] = __float_as_int(row_norm<METRIC>(a.view, me));     // the "query" is a stored row
        }
        __syncthreads();
        uint32_t D = 0, E = 0;
        uint32_t cur = greedy_descent_spec<METRIC, G>(a.view, s, a.view.entry, a.view.max_level, target, D);
        for(int level = target < a.view.max_level ? target : a.view.max_level; level >= 0; --level) {
            int cnt;
            if(level == 0) cnt = search_level_spec<METRIC, G, KPL, ROWS, U, true, false>(a.view, s, sc, bitmap, a.bm_words, cur, (int)a.efc, D, E, nullptr);
            else cnt = search_level_reg<METRIC, G, KPL>(a.view, s, bitmap, a.bm_words, cur, level, (int)a.efc, D, E);
            uint64_t *top = a.tops + (size_t)(item0 + (uint32_t)level) * a.efc;
            for(int i = tid; i < cnt; i += T) top[ i ] = s.keys[ i ] & ~1ull;  // drop the "expanded" bit
            if(tid == 0) {
                a.top_count[ item0 + (uint32_t)level ] = (uint32_t)cnt;
                // sel[0] of the heuristic = minimum by (distance, tie_mix(slot, me)): only an exact tie at the smallest distance
                // can differ from keys[0]
                const uint32_t d0 = (uint32_t)(s.keys[ 0 ] >> 32);
                uint32_t       best = key_slot(s.keys[ 0 ]);
                for(int i = 1; i < cnt && (uint32_t)(s.keys[ i ] >> 32) == d0; ++i) {
                    const uint32_t id = key_slot(s.keys[ i ]);
                    if(tie_mix(id, me) < tie_mix(best, me)) best = id;
                }
                s.scal[ S_CUR ] = (int)best;
            }
            __syncthreads();
            cur = (uint32_t)s.scal[ S_CUR ];
            __syncthreads();
        }
        if(tid == 0) {
            if(a.totals) {
                atomicAdd(&a.totals[ 0 ], (unsigned long long)D);
                atomicAdd(&a.totals[ 1 ], (unsigned long long)E);
            }
            s.scal[ S_POS ] = a.ticket ? (int)(a.b_begin + gridDim.x + atomicAdd(a.ticket, 1u)) : (int)(b + gridDim.x);
        }
        __syncthreads();
        b = (uint32_t)s.scal[ S_POS ];
        __syncthreads();
    }
}

bool insert_spec_supported(int metric, uint32_t efc, uint32_t M0) { return (metric == M_L2SQ || metric == M_COS) && efc <= 128 && M0 >= 2 && M0 <= 64; }

size_t insert_spec_lds_bytes(uint32_t chunks, uint32_t efc, uint32_t M0, uint32_t vis_slots, uint32_t prefetch, uint32_t cache_entries)
{
    return walk_lds_bytes(chunks, efc, M0, vis_slots) + spec_lds_bytes(M0, prefetch, cache_entries);
}

hipError_t launch_insert_spec(int metric, const InsertArgs &a, int waves, int grid, hipStream_t stream)
{
    if(!insert_spec_supported(metric, a.efc, a.view.M0) || waves < 4 || waves > 11) return hipErrorInvalidValue;
    const size_t lds = insert_spec_lds_bytes(a.view.chunks, a.efc, a.view.M0, a.vis_slots, a.spec_prefetch, a.spec_cache);
    const int    kpl = a.efc <= 64 ? 1 : 2;
    const int    G_ = group_lanes_for(a.view.chunks);
#define LGPU_INS1(MM, GG, KK)                                                                                                   \
    {                                                                                                                           \
        static LdsAttrCache attr_;        \
        ensure_dynamic_lds((const void *)k_insert_spec<MM, GG, KK>, lds, attr_);    \
        hipLaunchKernelGGL((k_insert_spec<MM, GG, KK>), dim3(grid), dim3(64 * waves), lds, stream, a);                          \
    }
#define LGPU_INS(MM)                                                                  \
    switch(G_) {                                                                      \
        case 64: if(kpl == 1) LGPU_INS1(MM, 64, 1) else LGPU_INS1(MM, 64, 2) break;   \
        case 32: if(kpl == 1) LGPU_INS1(MM, 32, 1) else LGPU_INS1(MM, 32, 2) break;   \
        case 16: if(kpl == 1) LGPU_INS1(MM, 16, 1) else LGPU_INS1(MM, 16, 2) break;   \
        default: if(kpl == 1) LGPU_INS1(MM, 8, 1) else LGPU_INS1(MM, 8, 2)            \
    }
    if(metric == M_L2SQ) LGPU_INS(M_L2SQ)
    else LGPU_INS(M_COS)
#undef LGPU_INS
#undef LGPU_INS1
    return hipGetLastError();
}

}  // namespace lgpu
