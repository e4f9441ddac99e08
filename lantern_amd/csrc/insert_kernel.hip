// insert_kernel.hip -- k_insert: the walk of usearch_add (build.c:128; server.rs:349 add_raw).  Per new vector: descent + a
// per-level ef_construction-wide walk (walk.hpp); the sorted results go to k_connect (kernels.hip).
#include <algorithm>
#include <cstdlib>

#ifndef LGPU_LIST_PREFETCH  // the insertion walks do not fetch the front's list ahead: measured -2 % on builds (profiles/r06_list_prefetch_ab.md)
#define LGPU_LIST_PREFETCH 0
#endif
#ifdef LGPU_INSERT_ROW_BLOCK_COS  // this translation unit's own block for the cosine row pairs (device_common.hpp LGPU_ROW_BLOCK_COS)
#define LGPU_ROW_BLOCK_COS LGPU_INSERT_ROW_BLOCK_COS
#endif
#include "kernels.hpp"
#include "walk.hpp"
#include "dispatch.hpp"

namespace lgpu {

// ---------------------------------------------------------------------------------------------------
// k_insert: the WALK half of an insertion.  Per new vector: descent to its level, then per level an
// ef_construction-wide search_level whose sorted result (<= efc keys) goes to HBM for k_connect.  The start of
// the next lower level is connect_new_node_'s first pick: the closest result under (distance, tie_mix).
template <int METRIC, int G, int KPL = 2>  // KPL: as k_search (keys per lane of wave 0's register list; 0 = LDS list)
#ifndef LGPU_INSERT_MIN_BLOCKS
#define LGPU_INSERT_MIN_BLOCKS 6  // (measured: 5 and 4 -- 81 / 90 registers, no spills -- build at the same speed; DESIGN_HISTORY.md H.2 item 1)
#endif
// [r6] the cosine walks are compiled for the FIVE workgroups per CU that k_insert's LDS lets run anyway (<= 96 VGPRs): with the blocked row
// loads they spill at 80 (48 bytes of scratch, reloads inside a hop) -- same-box A/B at 1M x 768: walk 742 -> 691 ms, 949 -> 993 k vectors/s;
// the other metrics measure the same or slower at five (profiles/r06_row_block_ab.jsonl, second table)
__global__ void __launch_bounds__(512, (METRIC % 100 == M_COS) ? 5 : LGPU_INSERT_MIN_BLOCKS) k_insert(InsertArgs a)
{
    const int tid = threadIdx.x, T = blockDim.x;
    WalkLds   s;
    carve_walk(lgpu_smem, s, a.view.chunks, a.efc, a.view.M0, a.vis_slots);
    uint32_t *bitmap = a.bitmaps + (size_t)blockIdx.x * (a.bm_words + kVisUndoWords);
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
        const int lowest = a.only_upper ? 1 : 0;  // (block-uniform: a node of level 0 has nothing to walk then)
        uint32_t cur = target >= lowest ? greedy_descent<METRIC, G>(a.view, s, a.view.entry, a.view.max_level, target, D) : 0u;
        for(int level = target < a.view.max_level ? target : a.view.max_level; level >= lowest; --level) {
            int cnt;
            if constexpr(KPL > 0) cnt = search_level_reg<METRIC, G, KPL>(a.view, s, bitmap, a.bm_words, cur, level, (int)a.efc, D, E);
            else cnt = search_level<METRIC, G>(a.view, s, bitmap, a.bm_words, cur, level, (int)a.efc, D, E);
            uint64_t *top = a.tops + (size_t)(item0 + (uint32_t)level) * a.efc;
            for(int i = tid; i < cnt; i += T) top[ i ] = s.keys[ i ] & ~1ull;  // drop the "expanded" bit
            if(tid == 0) {
                a.top_count[ item0 + (uint32_t)level ] = (uint32_t)cnt;
                // sel[0] of the heuristic = minimum by (distance, tie_mix(slot, me)): only an exact tie at the
                // smallest distance can differ from keys[0]
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

size_t insert_lds_bytes(uint32_t chunks, uint32_t efc, uint32_t M0, uint32_t vis_slots) { return walk_lds_bytes(chunks, efc, M0, vis_slots); }

hipError_t launch_insert(int metric, const InsertArgs &a, int waves, int grid, hipStream_t stream)
{
    const size_t lds = insert_lds_bytes(a.view.chunks, a.efc, a.view.M0, a.vis_slots);
    const int    kpl = a.lds_list ? 0 : a.efc <= 64 ? 1 : a.efc <= 128 ? 2 : 0;
#define LGPU_LAUNCH_INSERT(...)                                                                                        \
    {                                                                                                                  \
        static LdsAttrCache attr_;        \
        ensure_dynamic_lds((const void *)k_insert<__VA_ARGS__>, lds, attr_);    \
        hipLaunchKernelGGL((k_insert<__VA_ARGS__>), dim3(grid), dim3(64 * waves), lds, stream, a);                     \
    }
#define CALL(MM, GG)                               \
    {                                              \
        if(kpl == 1) LGPU_LAUNCH_INSERT(MM, GG, 1) \
        else if(kpl == 2) LGPU_LAUNCH_INSERT(MM, GG, 2) \
        else LGPU_LAUNCH_INSERT(MM, GG, 0)         \
    }
    LGPU_DISPATCH(metric, a.view.chunks, CALL);
#undef CALL
#undef LGPU_LAUNCH_INSERT
    return hipGetLastError();
}

}  // namespace lgpu
