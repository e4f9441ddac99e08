// search_kernel.hip -- k_search launcher for the bandwidth-bound shapes (search_kernel.hpp has the kernel; the latency-bound
// instantiations live in search_spec_kernel.hip).  Its own translation unit: the instantiations (metric x lanes per row x list
// placement x rows in flight) compile in parallel with the build-side kernels.
#include "search_kernel.hpp"

namespace lgpu {

size_t search_lds_bytes(uint32_t chunks, uint32_t ef_cap, uint32_t M0, uint32_t vis_slots) { return walk_lds_bytes(chunks, ef_cap, M0, vis_slots); }
// ... for the list placement of this launch (walk.hpp search_level_reg): one key per lane of wave 0 up to ef = 64, two up to
// 128, the LDS list beyond (or when LANTERN_GPU_LDS_LIST asks for it)
#define LGPU_LAUNCH_SEARCH_KPL(MM, GG, PP, RR)                  \
    {                                                           \
        if(kpl == 1) LGPU_LAUNCH_SEARCH(MM, GG, PP, RR, 1)      \
        else if(kpl == 2) LGPU_LAUNCH_SEARCH(MM, GG, PP, RR, 2) \
        else LGPU_LAUNCH_SEARCH(MM, GG, PP, RR, 0)              \
    }

hipError_t launch_search(int metric, const SearchArgs &a, int waves, int grid, hipStream_t stream)
{
    if(a.spec) return launch_search_spec(metric, a, waves, grid, stream);
    const size_t lds = search_lds_bytes(a.view.chunks, a.ef, a.view.M0, a.vis_slots);
    const int    kpl = a.lds_list ? 0 : a.ef <= 64 ? 1 : a.ef <= 128 ? 2 : 0;
    const int    G_ = group_lanes_for(a.view.chunks);
    if(a.wide_rows && !a.phase_cycles && G_ == 64) {  // the small-batch shape (rows of >= 128 chunks)
        bool launched = true;
        switch(metric) {
            case M_L2SQ: LGPU_LAUNCH_SEARCH_KPL(M_L2SQ, 64, false, 4); break;
            case M_COS: LGPU_LAUNCH_SEARCH_KPL(M_COS, 64, false, 4); break;
            case M_HAMMING: LGPU_LAUNCH_SEARCH_KPL(M_HAMMING, 64, false, 4); break;
            case M_L2SQ_F16: LGPU_LAUNCH_SEARCH_KPL(M_L2SQ_F16, 64, false, 4); break;
            case M_COS_F16: LGPU_LAUNCH_SEARCH_KPL(M_COS_F16, 64, false, 4); break;
            case M_L2SQ_PQD: LGPU_LAUNCH_SEARCH_KPL(M_L2SQ_PQD, 64, false, 4); break;
            case M_COS_PQD: LGPU_LAUNCH_SEARCH_KPL(M_COS_PQD, 64, false, 4); break;
            default: launched = false;  // i8 storage (rows of >= 2033 dims) has no four-row instantiation: the two-row shape below
        }
        if(launched) return hipGetLastError();
    }
    if(a.phase_cycles) {  // diagnostic instantiations: the f32 metrics at the two common row shapes
        if(metric == M_L2SQ && G_ == 64) LGPU_LAUNCH_SEARCH_KPL(M_L2SQ, 64, true, 2)
        else if(metric == M_L2SQ && G_ == 16) LGPU_LAUNCH_SEARCH_KPL(M_L2SQ, 16, true, 2)
        else if(metric == M_COS && G_ == 64) LGPU_LAUNCH_SEARCH_KPL(M_COS, 64, true, 2)
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
#define CALL(MM, GG) LGPU_LAUNCH_SEARCH_KPL(MM, GG, false, 2)
    if(mcode_is_pqd(metric)) {  // a compact pq index, rows decoded on the fly (device_common.hpp PqdRow); G by the DECODED row
#define PQD_G(MM)                                                                    \
    switch(G_) { case 64: CALL(MM, 64); break; case 32: CALL(MM, 32); break; case 16: CALL(MM, 16); break; default: CALL(MM, 8); }
        if(metric == M_L2SQ_PQD) PQD_G(M_L2SQ_PQD)
        else if(metric == M_COS_PQD) PQD_G(M_COS_PQD)
        else return hipErrorInvalidValue;
#undef PQD_G
        return hipGetLastError();
    }
    LGPU_DISPATCH(metric, a.view.chunks, CALL);
#undef CALL
    return hipGetLastError();
}

}  // namespace lgpu
