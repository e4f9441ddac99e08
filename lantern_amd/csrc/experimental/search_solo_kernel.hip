// search_solo_kernel.hip -- k_search_solo: usearch_search_ef (lantern_hnsw/src/hnsw/scan.c:220-228) for a lone query or a handful,
// ONE WAVE per query, no workgroup barrier (walk_solo.hpp).  f32 l2sq / cos rows of fewer than 64 chunks, M <= 16, ef <= 64.
#include "../kernels.hpp"
#include "walk_solo.hpp"

namespace lgpu {

template <int METRIC, int G, int CPL, bool RAGGED, bool PROF = false>
__global__ void __launch_bounds__(64, 1) k_search_solo(SearchArgs a)
{
    const int lane = (int)(threadIdx.x & 63);
    SoloLds   s;
    carve_solo(lgpu_smem, s, a.spec_cache, a.vis_slots);
    for(uint32_t i = (uint32_t)lane; i < (1u << a.spec_cache); i += 64) {
        s.tags[ i ] = EMPTY;
        s.stamps[ i ] = 0u;
    }
    uint32_t hop_ctr = 0;
    for(uint32_t q = blockIdx.x; q < a.nq;) {
        uint32_t D = 0, E = 0;
        int      cnt = 0;
        if(a.view.n != 0) {
            SoloWalk<METRIC, G, CPL, RAGGED> w(a.view, s);
            w.load_query(a.queries + (size_t)q * a.view.chunks);
            float          d0;
            const uint32_t start = w.descend(D, d0);
            cnt = w.template level0<PROF>(start, d0, (int)a.ef, D, E, hop_ctr, PROF ? a.phase_cycles : nullptr);
        }
        __builtin_amdgcn_wave_barrier();
        int got = cnt - (int)a.skip;
        got = got < 0 ? 0 : (got > (int)a.k ? (int)a.k : got);
        for(uint32_t i = (uint32_t)lane; i < a.k; i += 64) {
            const size_t o = (size_t)q * a.k + i;
            if((int)i < got) {
                const uint64_t key = s.keys[ a.skip + i ];
                const uint32_t slot = key_slot(key);
                if(a.out_labels) a.out_labels[ o ] = a.labels[ slot ];
                if(a.out_dists) a.out_dists[ o ] = key_dist(key);
                if(a.out_slots) a.out_slots[ o ] = slot;
            } else {
                if(a.out_labels) a.out_labels[ o ] = 0;  // INVALID_ELEMENT_LABEL (hnsw.h:40)
                if(a.out_dists) a.out_dists[ o ] = __builtin_inff();
                if(a.out_slots) a.out_slots[ o ] = EMPTY;
            }
        }
        uint32_t next = q + gridDim.x;
        if(lane == 0) {
            if(a.out_counts) a.out_counts[ q ] = (uint32_t)got;
            if(a.out_D) a.out_D[ q ] = D;
            if(a.out_E) a.out_E[ q ] = E;
            if(a.totals) { atomicAdd(&a.totals[ 0 ], (unsigned long long)D); atomicAdd(&a.totals[ 1 ], (unsigned long long)E); }
            if(a.ticket) next = gridDim.x + atomicAdd(a.ticket, 1u);
        }
        next = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
        if(a.done || a.done_flags) {
            // a host that waits on this counter (or on the query's flag) instead of on the stream sees this query's answers first:
            // every lane's stores are complete (vmcnt) before lane 0 releases at system scope
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_wave_barrier();
            if(lane == 0) {
                __threadfence_system();
                if(a.done) __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                if(a.done_flags) __hip_atomic_store(&a.done_flags[ q ], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        q = next;
    }
}

size_t search_solo_lds_bytes(uint32_t ne_log2, uint32_t bm_words) { return solo_lds_bytes(1u << ne_log2, bm_words); }

bool search_solo_supported(int metric, uint32_t chunks, uint32_t M, uint32_t M0, uint32_t ef)
{
    return (metric == M_L2SQ || metric == M_COS) && chunks >= 1 && chunks < 64 && M0 <= 32 && M0 == 2 * M && M % 4 == 0 && M >= 4 && ef <= 64 && ef >= 1;
}

#define LGPU_LAUNCH_SOLO_R(MM, GG, CC, RR)                                                            \
    {                                                                                                 \
        static LdsAttrCache attr_;                                                                    \
        ensure_dynamic_lds((const void *)k_search_solo<MM, GG, CC, RR>, lds, attr_);                  \
        hipLaunchKernelGGL((k_search_solo<MM, GG, CC, RR>), dim3(grid), dim3(64), lds, stream, a);    \
    }
#define LGPU_LAUNCH_SOLO(MM, GG, CC)                                                              \
    {                                                                                             \
        if(ragged) LGPU_LAUNCH_SOLO_R(MM, GG, CC, true)                                           \
        else LGPU_LAUNCH_SOLO_R(MM, GG, CC, false)                                                \
    }
#define LGPU_SOLO_CPL(MM, GG)                                                                     \
    switch(cpl) {                                                                                 \
        case 1: LGPU_LAUNCH_SOLO(MM, GG, 1) break;                                                \
        case 2: LGPU_LAUNCH_SOLO(MM, GG, 2) break;                                                \
        case 3: LGPU_LAUNCH_SOLO(MM, GG, 3) break;                                                \
        default: LGPU_LAUNCH_SOLO(MM, GG, 4) break;                                               \
    }

// a.spec_cache = log2 of the list-cache entries, a.vis_slots = words of the LDS visited bitmap (a multiple of 4, >= ceil(n / 32))
hipError_t launch_search_solo(int metric, const SearchArgs &a, int grid, hipStream_t stream)
{
    if(!search_solo_supported(metric, a.view.chunks, a.view.M, a.view.M0, a.ef)) return hipErrorInvalidValue;
    const size_t lds = search_solo_lds_bytes(a.spec_cache, a.vis_slots);
    if(lds > 160 * 1024 || (size_t)a.vis_slots * 32 < a.view.n) return hipErrorInvalidValue;
    const int  G_ = group_lanes_for(a.view.chunks), cpl = ((int)a.view.chunks + G_ - 1) / G_;
    const bool ragged = (int)a.view.chunks % G_ != 0;
    if(a.phase_cycles) {  // the diagnostic instantiation (lantern_gpu_spec_profile): f32 l2sq, 32-chunk rows (128-d)
        if(metric != M_L2SQ || G_ != 16 || cpl != 2 || ragged) return hipErrorInvalidValue;
        static LdsAttrCache attr_;
        ensure_dynamic_lds((const void *)k_search_solo<M_L2SQ, 16, 2, false, true>, lds, attr_);
        hipLaunchKernelGGL((k_search_solo<M_L2SQ, 16, 2, false, true>), dim3(grid), dim3(64), lds, stream, a);
        return hipGetLastError();
    }
    if(metric == M_L2SQ) {
        if(G_ == 16) LGPU_SOLO_CPL(M_L2SQ, 16)
        else LGPU_SOLO_CPL(M_L2SQ, 8)
    } else {
        if(G_ == 16) LGPU_SOLO_CPL(M_COS, 16)
        else LGPU_SOLO_CPL(M_COS, 8)
    }
    return hipGetLastError();
}

}  // namespace lgpu
