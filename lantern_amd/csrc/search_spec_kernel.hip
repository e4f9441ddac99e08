// search_spec_kernel.hip -- k_search in its latency-bound forms (walk_spec.hpp): a lone query (three role waves + eight row
// waves) and batches that cannot fill the chip (four waves).  Same template as search_kernel.hip, its own translation unit.
#include "search_kernel.hpp"

namespace lgpu {

size_t search_spec_lds_bytes(uint32_t M0, uint32_t prefetch, uint32_t cache_entries, uint32_t twin) { return spec_lds_bytes(M0, prefetch, cache_entries, twin); }

// rows in flight per G-lane group, so that eight row waves cover a 32-entry list in one pass (the four-wave shape takes two)
#define LGPU_SPEC_ROWS(GG) ((GG) == 64 ? 4 : (GG) == 32 ? 2 : 1)
#define LGPU_LAUNCH_SPEC(MM, GG)                                                        \
    {                                                                                   \
        if(a.spec == 2) {                                                               \
            if(kpl == 1) LGPU_LAUNCH_SEARCH(MM, GG, false, LGPU_SPEC_ROWS(GG), 1, 2)    \
            else LGPU_LAUNCH_SEARCH(MM, GG, false, LGPU_SPEC_ROWS(GG), 2, 2)            \
        } else {                                                                        \
            if(kpl == 1) LGPU_LAUNCH_SEARCH(MM, GG, false, LGPU_SPEC_ROWS(GG), 1, 1)    \
            else LGPU_LAUNCH_SEARCH(MM, GG, false, LGPU_SPEC_ROWS(GG), 2, 1)            \
        }                                                                               \
    }

hipError_t launch_search_spec(int metric, const SearchArgs &a, int waves, int grid, hipStream_t stream)
{
    if(a.ef > 128 || a.view.M0 > 64 || a.view.M0 < 2 || waves < 2 || (a.spec >= 2 && waves < 4)) return hipErrorInvalidValue;
    const size_t lds = search_lds_bytes(a.view.chunks, a.ef, a.view.M0, a.vis_slots) + spec_lds_bytes(a.view.M0, a.spec_prefetch, a.spec_cache, a.spec == 3);
    const int    kpl = a.ef <= 64 ? 1 : 2;
    if(a.phase_cycles) {  // diagnostic instantiations (lantern_gpu_spec_profile): f32 l2sq / cos rows of 32..63 and of >= 128 chunks, ef <= 64
        const int G_ = group_lanes_for(a.view.chunks);
#if LGPU_EXPERIMENTAL
#define LGPU_PROF_SPEC(MM, GG, RR) { if(a.spec == 3) LGPU_LAUNCH_SEARCH(MM, GG, true, RR, 1, 3) else if(a.spec == 2) LGPU_LAUNCH_SEARCH(MM, GG, true, RR, 1, 2) else LGPU_LAUNCH_SEARCH(MM, GG, true, RR, 1, 1) }
#else
#define LGPU_PROF_SPEC(MM, GG, RR) { if(a.spec == 2) LGPU_LAUNCH_SEARCH(MM, GG, true, RR, 1, 2) else LGPU_LAUNCH_SEARCH(MM, GG, true, RR, 1, 1) }
#endif
        if(kpl == 1 && metric == M_L2SQ && G_ == 16) LGPU_PROF_SPEC(M_L2SQ, 16, 1)
        else if(kpl == 1 && metric == M_L2SQ && G_ == 64) LGPU_PROF_SPEC(M_L2SQ, 64, 4)
        else if(kpl == 1 && metric == M_COS && G_ == 64) LGPU_PROF_SPEC(M_COS, 64, 4)
#undef LGPU_PROF_SPEC
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
#if !LGPU_EXPERIMENTAL
    if(a.spec >= 3) return hipErrorInvalidValue;  // (index.cpp never asks: without the experimental unit LANTERN_GPU_SPEC=3 / 4 mean 2)
#else
    if(a.spec == 3) {
        // Two nodes per round, the second one speculative (walk_twin.hpp).  Measured and NOT adopted: parity-green,
        // 51 rounds instead of 69 hops for the lone 100k x 128 query, but a round costs 1.9 hops -- the walk is bound by the
        // dependent instructions of its bookkeeping, not by the memory round trip the speculation hides.  Kept for the f32
        // metrics at the two common row shapes behind LANTERN_GPU_SPEC=3, with its parity tests.
        const int G_ = group_lanes_for(a.view.chunks);
        if(kpl != 1) return hipErrorInvalidValue;
        if(metric == M_L2SQ && G_ == 16) LGPU_LAUNCH_SEARCH(M_L2SQ, 16, false, 1, 1, 3)
        else if(metric == M_L2SQ && G_ == 64) LGPU_LAUNCH_SEARCH(M_L2SQ, 64, false, 4, 1, 3)
        else if(metric == M_COS && G_ == 64) LGPU_LAUNCH_SEARCH(M_COS, 64, false, 4, 1, 3)
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
#endif
    if(mcode_is_pqd(metric)) {  // a compact pq index, rows decoded on the fly
        const int G_ = group_lanes_for(a.view.chunks);
#define PQD_G(MM)                                                                                                                   \
    switch(G_) { case 64: LGPU_LAUNCH_SPEC(MM, 64); break; case 32: LGPU_LAUNCH_SPEC(MM, 32); break; case 16: LGPU_LAUNCH_SPEC(MM, 16); break; \
                 default: LGPU_LAUNCH_SPEC(MM, 8); }
        if(metric == M_L2SQ_PQD) PQD_G(M_L2SQ_PQD)
        else if(metric == M_COS_PQD) PQD_G(M_COS_PQD)
        else return hipErrorInvalidValue;
#undef PQD_G
        return hipGetLastError();
    }
    LGPU_DISPATCH(metric, a.view.chunks, LGPU_LAUNCH_SPEC);
    return hipGetLastError();
}

}  // namespace lgpu
