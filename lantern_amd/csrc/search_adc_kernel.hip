// search_adc_kernel.hip -- k_search_adc: usearch_search_ef (scan.c:220-228) over a pq = true index that keeps only its CODE
// BYTES in HBM (usearch_storage.cpp:29-31: a node carries num_subvectors bytes; scan.c:75-81, build.c:497-500,
// product_quantization.c:207-240).  The walk is walk.hpp's; what changes is the evaluation of a row:
//
//   per query   the table lut[s][c] (s < num_subvectors, c < num_centroids) = the metric's partial sum between subvector s of
//               the query and centroid c of that subvector: l2sq: one fma chain of (q - c)^2 over the subvector's dimensions in
//               memory order; cos: the chain of q * c.  98 KB of LDS at 96 subvectors x 256 centroids (one workgroup per CU).
//   per row     num_subvectors table entries added up: lane l of the row's 8-lane group owns the 16 codes of chunk l and adds
//               their entries in code order; the lanes' sums meet in the 8-lane tree (device_common.hpp RowAcc<M_*_ADC>).
//               cos: 1 - sum / (|q| * |row|), |row| = the cached norm of the row's decoding, |q| in the f32 kernels' own order.
//
// A distance to a decoded vector, as the reference defines a PQ index's distances -- in ADC's summation order, which is ours
// to define (the fork's is not in the tree): the oracle restates exactly this (oracle/hnsw.c lo_set_pq_view), device and
// oracle agree bit for bit, and both agree with the decoded-row distances to 1e-5 relative.  PARITY UNPINNED BY THE REFERENCE.
// Memory: 10M x 768 at 96 subvectors = 0.96 GB of rows instead of 30.7 GB.
#include "search_kernel.hpp"

namespace lgpu {

// The per-query table: one thread per (subvector, centroid) entry, one fma chain each.  B entries per thread at a time, their
// centroid loads issued together: the table of centroids comes from L2 (786 KB per query at 96 x 256 x 8), and one chain per
// thread at a time left the loads' latency bare -- 35 round trips per query, a third of a query's time at ef = 64.
template <int METRIC, int B>
__device__ __forceinline__ void adc_build_table(float *lut, const float *centers, const float *rawq, uint32_t S, uint32_t C, uint32_t subdim, uint32_t sub_floats,
                                                uint32_t tid, uint32_t T)
{
    for(uint32_t e0 = tid; e0 < S * C; e0 += B * T) {
        const float *cent[ B ], *qs[ B ];
        float        acc[ B ];
        bool         on[ B ];
#pragma unroll
        for(int u = 0; u < B; ++u) {
            const uint32_t e = e0 + (uint32_t)u * T;
            on[ u ] = e < S * C;
            const uint32_t ee = on[ u ] ? e : 0u, sv = ee / C, c = ee % C;
            cent[ u ] = centers + ((size_t)sv * C + c) * sub_floats;
            qs[ u ] = rawq + (size_t)sv * subdim;
            acc[ u ] = 0.f;
        }
        for(uint32_t j4 = 0; j4 < sub_floats; j4 += 4) {
            float4 cv[ B ];
#pragma unroll
            for(int u = 0; u < B; ++u) cv[ u ] = *(const float4 *)(cent[ u ] + j4);  // (rows of the centroid table are padded to whole float4s)
#pragma unroll
            for(int u = 0; u < B; ++u) {
                const float cj[ 4 ] = { cv[ u ].x, cv[ u ].y, cv[ u ].z, cv[ u ].w };
#pragma unroll
                for(int jj = 0; jj < 4; ++jj) {
                    if(j4 + (uint32_t)jj < subdim) {
                        const float qv = qs[ u ][ j4 + (uint32_t)jj ];
                        if constexpr(METRIC == M_L2SQ_ADC) {
                            const float t = qv - cj[ jj ];
                            acc[ u ] = __builtin_fmaf(t, t, acc[ u ]);
                        } else {
                            acc[ u ] = __builtin_fmaf(qv, cj[ jj ], acc[ u ]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for(int u = 0; u < B; ++u) {
            const uint32_t e = e0 + (uint32_t)u * T;
            if(on[ u ]) lut[ (size_t)(e / C) * ADC_LUT_STRIDE + e % C ] = acc[ u ];
        }
    }
}

// SPEC: the latency-bound walk of walk_spec.hpp in its lone-query shape (three role waves + eight row waves, one barrier per hop,
// neighbour lists fetched with the rows).  A table of 96 x 256 entries leaves room for ONE workgroup per CU, so every query of
// a batch walks alone on its CU whatever the batch size: the walk that is fastest alone is the one to run.
// (SPEC 2 = the lone-query shape.  The two-nodes-per-round form of walk_twin.hpp was measured here too -- 1.98 -> 1.65 M queries/s at
// 96 subvectors -- and is not instantiated.)
template <int METRIC, int KPL, int SPEC = 0>
__global__ void __launch_bounds__(SPEC ? 704 : 512, SPEC ? 1 : 2) k_search_adc(SearchArgs)
{
    // (arguments are re-read from the kernarg segment where a query needs them, as in k_search -- search_kernel.hpp: kept live
    // across the persistent loop they cost the hop loop ~120 scalar-register spill reloads)
    constexpr int G = 8;  // rows are at most 8 chunks (128 codes)
    const int     tid = threadIdx.x, T = blockDim.x;
    WalkLds       s;
    SpecLds       sc;
    uint32_t      lut_chunks;
    {
        const KernargBytes ka = kernarg_opaque();
        lut_chunks = LGPU_VIEW_ARG(ka, SearchArgs, chunks) * 16 * ADC_LUT_STRIDE / 4;
        unsigned char *end = carve_walk(lgpu_smem, s, lut_chunks + LGPU_SEARCH_ARG(ka, adc_qchunks), LGPU_SEARCH_ARG(ka, ef), LGPU_VIEW_ARG(ka, SearchArgs, M0),
                                        LGPU_SEARCH_ARG(ka, vis_slots));  // s.q = the table, then the raw query row
        if constexpr(SPEC != 0) carve_spec(end, sc, LGPU_VIEW_ARG(ka, SearchArgs, M0), LGPU_SEARCH_ARG(ka, spec_prefetch), LGPU_SEARCH_ARG(ka, spec_cache));
        else (void)end;
    }
    float *const       lut = (float *)s.q;
    const uint4 *const rawq4 = s.q + lut_chunks;
    const float *const rawq = (const float *)rawq4;
    for(uint32_t q = blockIdx.x; q < LGPU_SEARCH_ARG(kernarg_opaque(), nq);) {
        uint32_t D = 0, E = 0;
        int      cnt = 0;
        {
            const KernargBytes ka = kernarg_opaque();
            const uint32_t     S = LGPU_SEARCH_ARG(ka, adc_S), C = LGPU_SEARCH_ARG(ka, adc_C), subdim = LGPU_SEARCH_ARG(ka, adc_subdim),
                           sub_floats = ((subdim + 3) / 4) * 4, qchunks = LGPU_SEARCH_ARG(ka, adc_qchunks), S16 = LGPU_VIEW_ARG(ka, SearchArgs, chunks) * 16;
            const uint4 *queries = LGPU_SEARCH_ARG(ka, queries);
            for(uint32_t i = tid; i < qchunks; i += T) ((uint4 *)rawq4)[ i ] = queries[ (size_t)q * qchunks + i ];
            __syncthreads();
            // entry 0 of the padding rows is +0.0
            adc_build_table<METRIC, 4>(lut, LGPU_SEARCH_ARG(ka, adc_centers), rawq, S, C, subdim, sub_floats, (uint32_t)tid, (uint32_t)T);  // (8 and 16 at a time: no faster)
            for(uint32_t sv = S + tid; sv < S16; sv += T) lut[ (size_t)sv * ADC_LUT_STRIDE ] = 0.f;
            if constexpr(METRIC == M_COS_ADC) {  // |query|: the chain and tree the f32 cosine kernels use for a row of this many chunks
                const int Gq = group_lanes_for(qchunks);
                if(tid < Gq) {
                    float qn;
                    switch(Gq) {
                        case 64: qn = group_norm<M_COS, 64>(rawq4, (int)qchunks, tid); break;
                        case 32: qn = group_norm<M_COS, 32>(rawq4, (int)qchunks, tid); break;
                        case 16: qn = group_norm<M_COS, 16>(rawq4, (int)qchunks, tid); break;
                        default: qn = group_norm<M_COS, 8>(rawq4, (int)qchunks, tid); break;
                    }
                    if(tid == Gq - 1) s.scal[ S_QN2 ] = __float_as_int(qn);
                }
            }
            __syncthreads();
        }
        {
            const KernargBytes ka = kernarg_opaque();
            View               v;
            LGPU_LOAD_VIEW(v, ka, SearchArgs)
            const uint32_t bm_words = LGPU_SEARCH_ARG(ka, bm_words);
            uint32_t      *bitmap = LGPU_SEARCH_ARG(ka, bitmaps) + (size_t)blockIdx.x * (bm_words + kVisUndoWords);
            s.undo = bitmap + bm_words;
            s.undo_cap = LGPU_SEARCH_ARG(ka, undo_cap);
            const int      ef = (int)LGPU_SEARCH_ARG(ka, ef);
            if(v.n != 0) {
                if constexpr(SPEC != 0) {
                    static_assert(SPEC == 0 || KPL > 0, "the latency-bound walk keeps its list in registers");
                    const uint32_t start = greedy_descent_spec<METRIC, G>(v, s, v.entry, v.max_level, 0, D);
                    cnt = search_level_spec<METRIC, G, (KPL > 0 ? KPL : 1), 1, 2, true, false>(v, s, sc, bitmap, bm_words, start, ef, D, E, nullptr);
                } else {
                    const uint32_t start = greedy_descent<METRIC, G>(v, s, v.entry, v.max_level, 0, D);
                    if constexpr(KPL > 0) cnt = search_level_reg<METRIC, G, KPL>(v, s, bitmap, bm_words, start, 0, ef, D, E);
                    else cnt = search_level<METRIC, G>(v, s, bitmap, bm_words, start, 0, ef, D, E);
                }
            }
        }
        const KernargBytes kb = kernarg_opaque();
        const uint32_t     k = LGPU_SEARCH_ARG(kb, k), skip = LGPU_SEARCH_ARG(kb, skip);
        const uint64_t    *labels = LGPU_SEARCH_ARG(kb, labels);
        uint64_t          *out_labels = LGPU_SEARCH_ARG(kb, out_labels);
        float             *out_dists = LGPU_SEARCH_ARG(kb, out_dists);
        uint32_t          *out_slots = LGPU_SEARCH_ARG(kb, out_slots);
        int                got = cnt - (int)skip;
        got = got < 0 ? 0 : (got > (int)k ? (int)k : got);
        for(uint32_t i = tid; i < k; i += T) {
            const size_t o = (size_t)q * k + i;
            if((int)i < got) {
                const uint64_t key = s.keys[ skip + i ];
                const uint32_t slot = key_slot(key);
                if(out_labels) out_labels[ o ] = labels[ slot ];
                if(out_dists) out_dists[ o ] = key_dist(key);
                if(out_slots) out_slots[ o ] = slot;
            } else {
                if(out_labels) out_labels[ o ] = 0;  // INVALID_ELEMENT_LABEL (hnsw.h:40)
                if(out_dists) out_dists[ o ] = __builtin_inff();
                if(out_slots) out_slots[ o ] = EMPTY;
            }
        }
        uint32_t *const done = LGPU_SEARCH_ARG(kb, done);
        if(tid == 0) {
            uint32_t *const           out_counts = LGPU_SEARCH_ARG(kb, out_counts);
            uint64_t *const           out_D = LGPU_SEARCH_ARG(kb, out_D), *const out_E = LGPU_SEARCH_ARG(kb, out_E);
            unsigned long long *const totals = LGPU_SEARCH_ARG(kb, totals);
            uint32_t *const           ticket = LGPU_SEARCH_ARG(kb, ticket);
            if(out_counts) out_counts[ q ] = (uint32_t)got;
            if(out_D) out_D[ q ] = D;
            if(out_E) out_E[ q ] = E;
            if(totals) { atomicAdd(&totals[ 0 ], (unsigned long long)D); atomicAdd(&totals[ 1 ], (unsigned long long)E); }
            s.scal[ S_POS ] = ticket ? (int)(gridDim.x + atomicAdd(ticket, 1u)) : (int)(q + gridDim.x);
        }
        __syncthreads();
        uint32_t *const done_flags = LGPU_SEARCH_ARG(kb, done_flags);
        if(tid == 0 && (done || done_flags)) {
            __threadfence_system();
            if(done) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if(done_flags) __hip_atomic_store(&done_flags[ q ], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        q = (uint32_t)s.scal[ S_POS ];
        __syncthreads();
    }
}

size_t search_adc_lds_bytes(uint32_t code_chunks, uint32_t qchunks, uint32_t ef_cap, uint32_t M0, uint32_t vis_slots)
{
    return walk_lds_bytes(code_chunks * 16 * ADC_LUT_STRIDE / 4 + qchunks, ef_cap, M0, vis_slots);
}

hipError_t launch_search_adc(int metric, const SearchArgs &a, int waves, int grid, hipStream_t stream)
{
    if(a.view.chunks == 0 || a.view.chunks > 8 || a.adc_C == 0 || a.adc_C > (uint32_t)ADC_LUT_STRIDE) return hipErrorInvalidValue;
    const int kpl = a.lds_list ? 0 : a.ef <= 64 ? 1 : a.ef <= 128 ? 2 : 0;
    if(a.spec && (kpl == 0 || waves < 4 || waves > 11 || a.view.M0 > 64 || a.view.M0 < 2)) return hipErrorInvalidValue;
    const size_t lds = search_adc_lds_bytes(a.view.chunks, a.adc_qchunks, a.ef, a.view.M0, a.vis_slots) + (a.spec ? spec_lds_bytes(a.view.M0, a.spec_prefetch, a.spec_cache) : 0);
#define LGPU_ADC1(MM, KK, SS)                                                                                                   \
    {                                                                                                                           \
        static LdsAttrCache attr_;        \
        ensure_dynamic_lds((const void *)k_search_adc<MM, KK, SS>, lds, attr_);    \
        hipLaunchKernelGGL((k_search_adc<MM, KK, SS>), dim3(grid), dim3(64 * waves), lds, stream, a);                           \
    }
#define LGPU_ADC(MM)                                  \
    {                                                 \
        if(a.spec) {                                  \
            if(kpl == 1) LGPU_ADC1(MM, 1, 2)          \
            else LGPU_ADC1(MM, 2, 2)                  \
        } else if(kpl == 1) LGPU_ADC1(MM, 1, 0)       \
        else if(kpl == 2) LGPU_ADC1(MM, 2, 0)         \
        else LGPU_ADC1(MM, 0, 0)                      \
    }
    if(metric == M_L2SQ_ADC) LGPU_ADC(M_L2SQ_ADC)
    else if(metric == M_COS_ADC) LGPU_ADC(M_COS_ADC)
    else return hipErrorInvalidValue;
#undef LGPU_ADC
#undef LGPU_ADC1
    return hipGetLastError();
}

}  // namespace lgpu
