// search_spec_kernel.hip -- k_search in its latency-bound forms (walk_spec.hpp): a lone query (three role waves + eight row
// waves) and batches that cannot fill the chip (four waves).  Same template as search_kernel.hip, its own translation unit.
#include "search_kernel.hpp"

namespace lgpu {

size_t search_spec_lds_bytes(uint32_t M0, uint32_t prefetch, uint32_t cache_entries) { return spec_lds_bytes(M0, prefetch, cache_entries); }

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
    if(a.ef > 128 || a.view.M0 > 64 || a.view.M0 < 2 || waves < 2 || (a.spec == 2 && waves < 4)) return hipErrorInvalidValue;
    const size_t lds = search_lds_bytes(a.view.chunks, a.ef, a.view.M0, a.vis_slots) + spec_lds_bytes(a.view.M0, a.spec_prefetch, a.spec_cache);
    const int    kpl = a.ef <= 64 ? 1 : 2;
    if(a.phase_cycles) {  // diagnostic instantiations (lantern_gpu_spec_profile): f32 l2sq / cos rows of 32..63 and of >= 128 chunks, ef <= 64
        const int G_ = group_lanes_for(a.view.chunks);
        if(kpl == 1 && metric == M_L2SQ && G_ == 16) { if(a.spec == 2) LGPU_LAUNCH_SEARCH(M_L2SQ, 16, true, 1, 1, 2) else LGPU_LAUNCH_SEARCH(M_L2SQ, 16, true, 1, 1, 1) }
        else if(kpl == 1 && metric == M_L2SQ && G_ == 64) { if(a.spec == 2) LGPU_LAUNCH_SEARCH(M_L2SQ, 64, true, 4, 1, 2) else LGPU_LAUNCH_SEARCH(M_L2SQ, 64, true, 4, 1, 1) }
        else if(kpl == 1 && metric == M_COS && G_ == 64) { if(a.spec == 2) LGPU_LAUNCH_SEARCH(M_COS, 64, true, 4, 1, 2) else LGPU_LAUNCH_SEARCH(M_COS, 64, true, 4, 1, 1) }
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    LGPU_DISPATCH(metric, a.view.chunks, LGPU_LAUNCH_SPEC);
    return hipGetLastError();
}

}  // namespace lgpu
