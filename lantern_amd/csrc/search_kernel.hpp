// search_kernel.hpp -- the k_search template (usearch_search_ef, lantern_hnsw/src/hnsw/scan.c:220-228, 273-281): one workgroup per
// query, persistent over the batch (work handed out by ticket); greedy descent + ef-bounded base-layer walk (walk.hpp,
// walk_spec.hpp).  Instantiated in two translation units that compile side by side: search_kernel.hip (the bandwidth-bound
// shapes) and search_spec_kernel.hip (the latency-bound ones).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdlib>

#include "kernels.hpp"
#include "walk.hpp"
#include "walk_spec.hpp"
// LGPU_EXPERIMENTAL (build.py, LANTERN_BUILD_EXPERIMENTAL=1): the walk variants that lost their A/B -- two nodes per round
// (experimental/walk_twin.hpp, SPEC 3) and the one-wave walk (experimental/walk_solo.hpp) -- are kept with their records and
// parity tests but are NOT in the default library.
#ifndef LGPU_EXPERIMENTAL
#define LGPU_EXPERIMENTAL 0
#endif
#if LGPU_EXPERIMENTAL
#include "experimental/walk_twin.hpp"
#endif
#include "dispatch.hpp"

namespace lgpu {

// ---------------------------------------------------------------------------------------------------
// ROWS = 4 is the SMALL-BATCH shape: when the batch cannot fill six workgroups per CU anyway (<= four 4-wave workgroups per
// CU), every workgroup keeps four rows per group in flight instead of two and may use 128 VGPRs (four waves per SIMD): a
// CU's fetch rate is set by the bytes it has in flight, and at 1024 queries x 768-d the two-row shape left it at ~60 %.
// Kernel arguments are RE-READ from the kernarg segment at the two points of a query that need them (before the walk: the
// view, the query pointer, ef; after it: the output pointers) through a pointer the compiler cannot see through.  Left to
// itself it loads all ~45 argument dwords once and keeps them live across the persistent loop -- over the hop loop, which
// already needs ~60 scalars -- and pays with ~60 scalar-register spill reloads per hop; a dozen scalar loads per QUERY are free.
typedef const __attribute__((address_space(4))) unsigned char *KernargBytes;
__device__ __forceinline__ KernargBytes kernarg_opaque()
{
    KernargBytes p = (KernargBytes)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
#define LGPU_KARG(base, T, ...) (*(const __attribute__((address_space(4))) T *)((base) + (__VA_ARGS__)))
#define LGPU_SEARCH_ARG(base, field) LGPU_KARG(base, decltype(SearchArgs::field), offsetof(SearchArgs, field))
#define LGPU_VIEW_ARG(base, STRUCT, field) LGPU_KARG(base, decltype(View::field), offsetof(STRUCT, view) + offsetof(View, field))
#define LGPU_LOAD_VIEW(v, base, STRUCT)             \
    {                                               \
        v.vec = LGPU_VIEW_ARG(base, STRUCT, vec);   \
        v.chunks = LGPU_VIEW_ARG(base, STRUCT, chunks); \
        v.M = LGPU_VIEW_ARG(base, STRUCT, M);       \
        v.M0 = LGPU_VIEW_ARG(base, STRUCT, M0);     \
        v.nbr0 = LGPU_VIEW_ARG(base, STRUCT, nbr0); \
        v.upper_off = LGPU_VIEW_ARG(base, STRUCT, upper_off); \
        v.upper_nbr = LGPU_VIEW_ARG(base, STRUCT, upper_nbr); \
        v.levels = LGPU_VIEW_ARG(base, STRUCT, levels); \
        v.norm2 = LGPU_VIEW_ARG(base, STRUCT, norm2); \
        v.n = LGPU_VIEW_ARG(base, STRUCT, n);       \
        v.entry = LGPU_VIEW_ARG(base, STRUCT, entry); \
        v.max_level = LGPU_VIEW_ARG(base, STRUCT, max_level); \
    }
// ... and the fields only the decode-on-the-fly metrics read
#define LGPU_LOAD_VIEW_PQD(v, base, STRUCT)                   \
    {                                                         \
        v.pq_centers = LGPU_VIEW_ARG(base, STRUCT, pq_centers); \
        v.pq_cps = LGPU_VIEW_ARG(base, STRUCT, pq_cps);       \
        v.pq_C = LGPU_VIEW_ARG(base, STRUCT, pq_C);           \
        v.pq_inv = LGPU_VIEW_ARG(base, STRUCT, pq_inv);       \
        v.pq_row_bytes = LGPU_VIEW_ARG(base, STRUCT, pq_row_bytes); \
    }

// SPEC: the latency-bound walk of walk_spec.hpp -- 1: every wave evaluates rows and waves 0..2 carry the roles on top (the
// small-batch shape, four waves); 2: three dedicated role waves + row waves (the lone-query shape, 3 + 8 waves); 3: the same
// shape with two nodes per round, the second one speculative (walk_twin.hpp).
template <int METRIC, int G, bool PROF = false, int ROWS = 2, int KPL = 1, int SPEC = 0>
#ifndef LGPU_SEARCH_MIN_BLOCKS_COS
#define LGPU_SEARCH_MIN_BLOCKS_COS 6
#endif
__global__ void __launch_bounds__(SPEC >= 2 ? 704 : 512, SPEC >= 2 ? 3 : (SPEC == 1 || ROWS != 2) ? 4 : (METRIC % 100 == M_COS) ? LGPU_SEARCH_MIN_BLOCKS_COS : 6)  // SPEC 0, ROWS 2: <= 80 VGPRs, six 4-wave workgroups per CU
k_search(SearchArgs)
{
    const int tid = threadIdx.x, T = blockDim.x;
    WalkLds   s;
    SpecLds   sc;
    {
        const KernargBytes ka = kernarg_opaque();
        unsigned char     *end = carve_walk(lgpu_smem, s, LGPU_VIEW_ARG(ka, SearchArgs, chunks), LGPU_SEARCH_ARG(ka, ef), LGPU_VIEW_ARG(ka, SearchArgs, M0),
                                            LGPU_SEARCH_ARG(ka, vis_slots));
        if constexpr(SPEC != 0) carve_spec(end, sc, LGPU_VIEW_ARG(ka, SearchArgs, M0), LGPU_SEARCH_ARG(ka, spec_prefetch), LGPU_SEARCH_ARG(ka, spec_cache), SPEC == 3 ? 1u : 0u);
        else (void)end;
    }
    for(uint32_t q = blockIdx.x; q < LGPU_SEARCH_ARG(kernarg_opaque(), nq);) {
        uint32_t D = 0, E = 0;
        int      cnt = 0;
        unsigned long long pc[ 8 ] = { 0, 0, 0, 0, 0, 0, 0, 0 }, t_q = 0;
        {
            const KernargBytes ka = kernarg_opaque();
            View               v;
            LGPU_LOAD_VIEW(v, ka, SearchArgs)
            if constexpr(METRIC >= M_PQD) LGPU_LOAD_VIEW_PQD(v, ka, SearchArgs)
            const uint32_t chunks = v.chunks, bm_words = LGPU_SEARCH_ARG(ka, bm_words);
            uint32_t      *bitmap = LGPU_SEARCH_ARG(ka, bitmaps) + (size_t)blockIdx.x * (bm_words + kVisUndoWords);
            s.undo = bitmap + bm_words;
            s.undo_cap = LGPU_SEARCH_ARG(ka, undo_cap);
            const uint4   *queries = LGPU_SEARCH_ARG(ka, queries);
            const int      ef = (int)LGPU_SEARCH_ARG(ka, ef);
            for(uint32_t i = tid; i < chunks; i += T) s.q[ i ] = queries[ (size_t)q * chunks + i ];
            __syncthreads();
            if(kCachedNorms<METRIC>) {  // ||query||^2 once per query, by the chain Acc<M_COS> would run for every row
                if(tid < G) {
                    const float qn = group_norm<METRIC, G>(s.q, (int)chunks, tid);
                    if(tid == G - 1) s.scal[ S_QN2 ] = __float_as_int(qn);
                }
                __syncthreads();
            }
            if constexpr(PROF) {
                t_q = (unsigned long long)clock64();
                s.touched = LGPU_SEARCH_ARG(ka, touched);
                s.trace_cap = LGPU_SEARCH_ARG(ka, trace_cap);
                uint32_t *const tr = LGPU_SEARCH_ARG(ka, trace);
                s.trace = tr ? tr + (size_t)q * s.trace_cap : nullptr;
                s.trace_count = tr ? LGPU_SEARCH_ARG(ka, trace_count) + q : nullptr;
            }
            if(v.n != 0) {
                uint32_t start;
                if constexpr(SPEC != 0) start = greedy_descent_spec<METRIC, G>(v, s, v.entry, v.max_level, 0, D);
                else start = greedy_descent<METRIC, G, PROF>(v, s, v.entry, v.max_level, 0, D);
                if constexpr(PROF) pc[ 6 ] = (unsigned long long)clock64() - t_q;
                // KPL keys per lane of wave 0 hold the candidate list (ef <= 64 KPL); KPL = 0: the list lives in LDS
#if LGPU_EXPERIMENTAL
                if constexpr(SPEC == 3)
                    cnt = search_level_twin<METRIC, G, KPL, ROWS, (G == 64 ? 3 : 2), PROF>(v, s, sc, bitmap, bm_words, start, ef, D, E,
                                                                                           PROF ? LGPU_SEARCH_ARG(ka, phase_cycles) : nullptr);
                else
#endif
                if constexpr(SPEC != 0)
                    cnt = search_level_spec<METRIC, G, KPL, ROWS, (G == 64 && SPEC == 2 ? 3 : 2), SPEC == 2, PROF>(v, s, sc, bitmap, bm_words, start, ef, D, E,
                                                                                                                     PROF ? LGPU_SEARCH_ARG(ka, phase_cycles) : nullptr);
                else if constexpr(KPL > 0) cnt = search_level_reg<METRIC, G, KPL, PROF, ROWS>(v, s, bitmap, bm_words, start, 0, ef, D, E, pc);
                else cnt = search_level<METRIC, G, PROF, ROWS>(v, s, bitmap, bm_words, start, 0, ef, D, E, pc);
            }
        }
        const KernargBytes kb = kernarg_opaque();
        if constexpr(PROF && SPEC == 0) {
            unsigned long long *const phase_cycles = LGPU_SEARCH_ARG(kb, phase_cycles);
            if(tid == 0) pc[ 7 ] = (unsigned long long)clock64() - t_q;
            if((tid & 63) == 0 && phase_cycles) {  // thread 0, and the list wave's first lane (slot 4 of the split walk)
                for(int i = 0; i < 8; ++i)
                    if(tid == 0 ? (i != 4 || pc[ 4 ] != 0) : (i == 4 && pc[ 4 ] != 0)) atomicAdd(&phase_cycles[ i ], pc[ i ]);
            }
        }
        const uint32_t  k = LGPU_SEARCH_ARG(kb, k), skip = LGPU_SEARCH_ARG(kb, skip);
        const uint64_t *labels = LGPU_SEARCH_ARG(kb, labels);
        uint64_t       *out_labels = LGPU_SEARCH_ARG(kb, out_labels);
        float          *out_dists = LGPU_SEARCH_ARG(kb, out_dists);
        uint32_t       *out_slots = LGPU_SEARCH_ARG(kb, out_slots);
        int             got = cnt - (int)skip;
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
        if(tid == 0) {
            uint32_t *const           out_counts = LGPU_SEARCH_ARG(kb, out_counts);
            uint64_t *const           out_D = LGPU_SEARCH_ARG(kb, out_D), *const out_E = LGPU_SEARCH_ARG(kb, out_E);
            unsigned long long *const totals = LGPU_SEARCH_ARG(kb, totals);
            uint32_t *const           ticket = LGPU_SEARCH_ARG(kb, ticket);
            if(out_counts) out_counts[ q ] = (uint32_t)got;
            if(out_D) out_D[ q ] = D;
            if(out_E) out_E[ q ] = E;
            if(totals) { atomicAdd(&totals[ 0 ], (unsigned long long)D); atomicAdd(&totals[ 1 ], (unsigned long long)E); }
            // next query: a ticket (walks differ in length by 2x; static striding leaves workgroups idle at the end)
            s.scal[ S_POS ] = ticket ? (int)(gridDim.x + atomicAdd(ticket, 1u)) : (int)(q + gridDim.x);
        }
        __syncthreads();
        if(tid == 0) {
            // a host that waits on this counter instead of on the stream (the lone-query path: index.cpp search_one_locked)
            // sees this query's answers first: they were written before the barrier above, and the fence orders them
            uint32_t *const done = LGPU_SEARCH_ARG(kb, done), *const done_flags = LGPU_SEARCH_ARG(kb, done_flags);
            if(done || done_flags) __threadfence_system();
            if(done) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if(done_flags) __hip_atomic_store(&done_flags[ q ], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        q = (uint32_t)s.scal[ S_POS ];
        __syncthreads();
    }
}

// one instantiation: opt the kernel in to its dynamic LDS size, then launch
#define LGPU_LAUNCH_SEARCH(...)                                                                                        \
    {                                                                                                                  \
        static LdsAttrCache attr_;        \
        ensure_dynamic_lds((const void *)k_search<__VA_ARGS__>, lds, attr_);    \
        hipLaunchKernelGGL((k_search<__VA_ARGS__>), dim3(grid), dim3(64 * waves), lds, stream, a);                     \
    }

}  // namespace lgpu
