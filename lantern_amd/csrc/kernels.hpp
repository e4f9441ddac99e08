// kernels.hpp -- host-callable launchers of the gfx950 kernels (definitions in kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "device_common.hpp"
#include <atomic>

// LANTERN_BUILD_EXPERIMENTAL=1 (lantern_amd/build.py -> -DLGPU_EXPERIMENTAL=1) adds the walk variants that lost their A/B to the
// library: two nodes per round (experimental/walk_twin.hpp, LANTERN_GPU_SPEC=3) and the one-wave walk (experimental/walk_solo.hpp +
// experimental/search_solo_kernel.hip, LANTERN_GPU_SPEC=4 / LANTERN_GPU_SOLO=1).  The default library does not contain them: the
// switches then select the default latency-bound walk.  lantern_gpu_version() names the build.
#ifndef LGPU_EXPERIMENTAL
#define LGPU_EXPERIMENTAL 0
#endif

namespace lgpu {

struct SearchArgs
{
    View            view;
    const uint4    *queries;   // [nq][chunks], zero padded
    uint32_t        nq, k, ef, skip;
    const uint64_t *labels;    // [n]
    uint64_t       *out_labels;  // [nq][k] or NULL
    float          *out_dists;   // [nq][k] or NULL
    uint32_t       *out_slots;   // [nq][k] or NULL
    uint32_t       *out_counts;  // [nq] or NULL
    uint64_t       *out_D;       // [nq] or NULL
    uint64_t       *out_E;       // [nq] or NULL
    uint32_t       *bitmaps;     // [grid][bm_words + kVisUndoWords]: a workgroup's visited bitmap (all-zero between walks) + its undo log
    uint32_t        bm_words;    // multiple of 4
    uint32_t        undo_cap;    // entries of the undo log a walk may use (<= kVisUndoWords; LANTERN_GPU_VIS_UNDO for tests of the overflow path)
    uint32_t        vis_slots;   // LDS visited-set slots (a multiple of 4; 0 = HBM bitmap only)
    unsigned long long *totals;  // [2] cumulative D, E (atomicAdd) or NULL
    uint32_t       *ticket;      // zeroed before the launch: queries beyond the first gridDim.x are handed out dynamically
                                 // (NULL = static striding); results do not depend on who runs a query
    int             lds_list;    // keep the candidate list in LDS even when it fits wave 0's registers (LANTERN_GPU_LDS_LIST=1: the
                                 // round-1 walk, kept for A/B parity runs and for ef > 128)
    int             wide_rows;   // small batch: the four-rows-in-flight instantiation (k_search<.., ROWS = 4>)
    int             spec;        // latency-bound launches (walk_spec.hpp): 1 = four-wave shape, 2 = dedicated role waves (3 + 8 waves), 3 = the same with two
                                 // nodes per round, the second speculative (walk_twin.hpp); 0 = off
    uint32_t        spec_prefetch;  // fetch every evaluated row's own level-0 list with the row (M0 % 4 == 0, M0 / 4 <= lanes per row)
    uint32_t        spec_cache;     // entries of the LDS list cache (power of two; 0 = none)
    // ADC over PQ codes (search_adc_kernel.hip; view.vec = the code rows, view.chunks = 16-byte chunks per code row)
    const float    *adc_centers;    // [S][C][sub_floats] per-subvector centroid tables, rows zero padded to whole chunks
    uint32_t        adc_S, adc_C, adc_subdim;
    uint32_t        adc_qchunks;    // 16-byte chunks of a (raw f32) query row
    unsigned long long *phase_cycles;  // diagnostics (lantern_gpu_search_phase_profile): [8] shader-clock cycles summed over the
    uint32_t       *done;        // NULL, or a counter in host-visible memory: +1 (system scope) per finished query, after its answers
    uint32_t       *done_flags;  // NULL, or [nq] words in host-visible memory: done_flags[q] = 1 (system scope, release) once query q's answers
                                 // are written -- a host that keeps the answers in device-mapped memory hands each one on as ITS walk ends
                                 // instead of when the launch's longest walk does (lantern_gpu_search_batch_lane_notify)
    uint32_t       *touched;     // diagnostics (lantern_gpu_search_unique_rows; the instrumented instantiations only): [ceil(n / 32)] one
                                 // bit per row, set when any query of the launch evaluates the row; NULL = off
    uint32_t       *trace;       // diagnostics (lantern_gpu_search_row_trace; the instrumented instantiations only): [nq][trace_cap] the
    uint32_t       *trace_count; // memory objects every query asks for, in order (walk.hpp trace_append), and [nq] how many; NULL = off
    uint32_t        trace_cap;
};                               // launch's queries by phase: pop | list + visited | distances | merge | descent | whole query

// one reverse-link request produced by the insert pass: add `new_slot` to `close`'s list at `level`
struct LinkReq
{
    uint32_t close;  // EMPTY = unused entry
    uint32_t level;
    uint32_t new_slot;
    float    d;  // metric(new_slot, close)
};

struct InsertArgs
{
    View            view;       // n = size BEFORE the batch; entry/max_level frozen
    uint32_t        first_slot; // the batch occupies slots [first_slot, first_slot + count)
    uint32_t        b_begin;    // this launch walks the batch members [b_begin, count) -- a rank of a work-sharded
    uint32_t        count;      // build (shard.cpp) walks only its own sub-range; 0 / batch size otherwise
    uint32_t        efc;
    const uint32_t *link_off;   // [batch] first LinkReq of each new node (M*(level+1) entries each); /M = first item
    uint64_t       *tops;       // [items][efc] walk results, ascending (distance, slot) keys
    uint32_t       *top_count;  // [items]
    uint32_t       *bitmaps;     // as SearchArgs
    uint32_t        bm_words;
    uint32_t        undo_cap;
    uint32_t        vis_slots;   // LDS visited-set slots (0 = HBM bitmap only)
    unsigned long long *totals;  // [2] cumulative D, E
    uint32_t       *ticket;      // as SearchArgs::ticket, over the batch members
    int             lds_list;    // as SearchArgs::lds_list
    uint32_t        spec_prefetch, spec_cache;  // the latency-bound form (insert_spec_kernel.hip): as SearchArgs'
    uint32_t        only_upper = 0;  // k_insert only: 1 = walk levels >= 1 only (the row-sharded build takes a node's level-0 candidates from the shards)
};

// neighbour selection of the new nodes: one item per (new node, level)
struct ConnectArgs
{
    View            view;
    uint32_t        first_slot;
    uint32_t        item_begin;  // first item of this launch (a rank's sub-range of a work-sharded build; 0 otherwise)
    uint32_t        items;       // items of this launch
    uint32_t        efc;
    const uint32_t *link_off;    // [count]
    const uint32_t *item_node;   // [items] index of the new node within the batch
    const uint64_t *tops;
    const uint32_t *top_count;
    LinkReq        *links;       // [sum M*(level+1)]
    unsigned long long *totals;  // [1] cumulative pair evaluations
};

struct RevlinkArgs
{
    View            view;
    const uint32_t *ngroups;      // device: number of (close, level) groups of this batch (written by the grouping pass)
    const uint2    *groups;       // device: [*ngroups] {begin, end} into reqs; any order (groups are independent of one another)
    uint32_t        max_groups;   // host-side upper bound on *ngroups (sizes grids and the worklist)
    const LinkReq  *reqs;         // sorted by (close, level), stable: within a group in new-slot order
    unsigned long long *totals;   // [2] cumulative pair evaluations, re-prunes
    // Persistent re-prune state (k_revlink_pairs), or NULL: radius[node] / radius_upper[upper block] = d(node, LAST list entry)
    // when that list is FULL, greedy-consistent and stored in ascending (distance, tie) order -- i.e. it is exactly what a
    // re-prune left -- and NaN otherwise.  A request whose own distance sorts behind the radius is cut without reading a row;
    // any other request to such a list starts the chain form (<= cap distances) instead of the all-pairs table.
    float          *radius0, *radius_upper;
};

// The grouping pass (grouping.hip): the reverse-link requests of a batch, as k_connect left them (new-slot-major, EMPTY
// entries in between), become `sorted` -- stable radix sort by (close, level) -- and the list of groups.  Everything stays
// on the device and on `stream`: the batches of a build queue up without a host round trip.
// world > 1 (work-sharded build): only the requests whose `close` this rank owns (close % world == rank) are kept;
// owner_counts[world] (device, zeroed here) receives the number of requests per owner.
struct GroupScratch
{
    uint64_t *keys_a, *keys_b;   // [n]
    uint32_t *idx_a, *idx_b;     // [n]
    void     *temp;              // rocPRIM temporary storage
    size_t    temp_bytes;
};
size_t     group_temp_bytes(size_t n);
hipError_t launch_group_requests(const LinkReq *links, uint32_t n, const GroupScratch &gs, LinkReq *sorted, uint2 *groups, uint32_t *ngroups,
                                 int world, int rank, uint32_t *owner_counts, hipStream_t stream, uint32_t *zero_me = nullptr);
hipError_t launch_group_keys_owned(const LinkReq *links, uint32_t n, const GroupScratch &gs, uint32_t *ngroups, int world, int rank, uint32_t *owner_counts,
                                   hipStream_t stream, uint32_t *zero_me = nullptr);
hipError_t launch_group_sort_owned(const LinkReq *links, uint32_t m, const GroupScratch &gs, LinkReq *sorted, uint2 *groups, uint32_t *ngroups,
                                   hipStream_t stream);
// (zero_me: a device word the pass sets to 0 on the way -- the reverse-link kernels' work counter; launch_revlink is then told so)
// link_off[i] = M * sum_{j<i} (level_j + 1), item_node[item] = i for the (level_i + 1) items of node i -- the layout of
// one batch, from the levels already in HBM (no per-batch host upload)
hipError_t launch_batch_layout(const uint8_t *levels, uint32_t b, uint32_t M, uint32_t *link_off, uint32_t *item_node, hipStream_t stream);
// a few new rows and their metadata from a device-mapped host block to their places (one kernel instead of four copies)
hipError_t launch_stage_small(const void *rows, const uint64_t *labels, const uint32_t *upper_off, const uint8_t *levels, uint32_t count, uint32_t chunks,
                              void *d_rows, uint64_t *d_labels, uint32_t *d_upper_off, uint8_t *d_levels, hipStream_t stream);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device, size reached) instead of on every launch:
// a driver call of microseconds that a lone query or a one-row insertion would repeat every time.  `cache` is a static of the
// launcher's expansion for ONE instantiation: zero-initialised, [device ordinal] = the largest size set so far.
struct LdsAttrCache { std::atomic<size_t> set[ 64 ]; };
inline void ensure_dynamic_lds(const void *fn, size_t lds, LdsAttrCache &cache)
{
    int dev = 0;
    if(hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return;
    }
    if(cache.set[ dev ].load(std::memory_order_acquire) >= lds && lds > 0) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    size_t seen = cache.set[ dev ].load(std::memory_order_relaxed);
    while(seen < lds && !cache.set[ dev ].compare_exchange_weak(seen, lds, std::memory_order_release)) {}
}

// All launchers return hipSuccess or the launch error.  `metric` is a usearch_metric_kind_t value.
hipError_t launch_search(int metric, const SearchArgs &a, int waves, int grid, hipStream_t stream);
hipError_t launch_search_spec(int metric, const SearchArgs &a, int waves, int grid, hipStream_t stream);  // a.spec != 0 (search_spec_kernel.hip)
size_t     search_spec_lds_bytes(uint32_t M0, uint32_t prefetch, uint32_t cache_entries, uint32_t twin = 0);
#if LGPU_EXPERIMENTAL
// the one-wave walk (experimental/search_solo_kernel.hip, experimental/walk_solo.hpp): a.spec_cache = log2 of the list-cache entries,
// a.vis_slots = words of the LDS bitmap
bool       search_solo_supported(int metric, uint32_t chunks, uint32_t M, uint32_t M0, uint32_t ef);
size_t     search_solo_lds_bytes(uint32_t ne_log2, uint32_t bm_words);
hipError_t launch_search_solo(int metric, const SearchArgs &a, int grid, hipStream_t stream);
#endif
hipError_t launch_search_adc(int metric, const SearchArgs &a, int waves, int grid, hipStream_t stream);  // metric = M_L2SQ_ADC / M_COS_ADC
size_t     search_adc_lds_bytes(uint32_t code_chunks, uint32_t qchunks, uint32_t ef_cap, uint32_t M0, uint32_t vis_slots);
hipError_t launch_insert(int metric, const InsertArgs &a, int waves, int grid, hipStream_t stream);
// a handful of insertions (ldb_aminsert, the first batches of a build): level 0 by the lone-query walk (f32 l2sq / cos, efc <= 128)
bool       insert_spec_supported(int metric, uint32_t efc, uint32_t M0);
size_t     insert_spec_lds_bytes(uint32_t chunks, uint32_t efc, uint32_t M0, uint32_t vis_slots, uint32_t prefetch, uint32_t cache_entries);
hipError_t launch_insert_spec(int metric, const InsertArgs &a, int waves, int grid, hipStream_t stream);
hipError_t launch_connect(int metric, const ConnectArgs &a, hipStream_t stream);
// work: scratch of max_groups x 8 bytes; work_count: one u32 (both device memory)
hipError_t launch_revlink(int metric, const RevlinkArgs &a, void *work, uint32_t *work_count, int num_cus, hipStream_t stream,
                          bool work_count_is_zero = false);
// ---- work-sharded build (shard.cpp): the exchange steps either side of the RCCL all-gathers -------------
// Every rank holds the complete request array after the first all-gather; the own lists of the nodes another
// rank connected are rebuilt from their requests: list(new_slot, level)[i] = links[link_off[b] + level*M + i].close
hipError_t launch_apply_own_links(const View &v, uint32_t first_slot, const uint32_t *link_off, const LinkReq *links,
                                  uint32_t total_links, hipStream_t stream);
// One record per reverse-link group a rank owned: [close, level, list[0..M0)] (rec_words = M0 + 2 u32 words).
// pack: records[g] for g in [0, ngroups); apply: every record with close != EMPTY is written into the local replica.
hipError_t launch_pack_lists(const RevlinkArgs &a, uint32_t *records, hipStream_t stream);
hipError_t launch_apply_lists(const View &v, const uint32_t *records, uint32_t nrecords, hipStream_t stream);

// ---- product quantisation (grouping.hip): a PQ index holds every row's DECODING in the vector block (all distances are
// distances to / between decoded vectors) and the codes -- what the file and the pages carry -- beside it
hipError_t launch_pq_take(const float *rows, uint32_t row_floats, uint32_t first, uint32_t count, uint32_t s, uint32_t subdim, uint32_t sub_floats,
                          float *sub, hipStream_t stream);
hipError_t launch_pq_put(float *rows, uint32_t row_floats, uint32_t first, uint32_t count, uint32_t s, uint32_t subdim, uint32_t S, const uint32_t *nearest,
                         const float *codebook, uint32_t dims, uint8_t *codes, hipStream_t stream);
hipError_t launch_pq_decode(float *rows, uint32_t row_floats, uint32_t first, uint32_t count, uint32_t subdim, uint32_t S, const float *codebook, uint32_t dims,
                            const uint8_t *codes, hipStream_t stream);

// f32 rows -> the stored rows of an f16 / i8 / b1 index, on the device (the rules of pad_row, element for element)
hipError_t launch_merge_parts(const uint64_t *labels, const float *dists, uint32_t world, uint32_t nq, uint32_t k, uint64_t *out_labels,
                              float *out_dists, uint32_t *out_counts, hipStream_t stream);
// row-sharded build: the shards' candidate lists, merged, as the selection kernel's level-0 input
hipError_t launch_merge_candidates(const uint64_t *labels, const float *dists, uint32_t world, uint32_t nq, uint32_t k, uint32_t first_slot,
                                   const uint32_t *link_off, uint32_t M, uint32_t stride, uint64_t *tops, uint32_t *top_count, hipStream_t stream);
hipError_t launch_store_quantised(const float *src, uint32_t dims, uint32_t count, int kind, uint32_t *rows, uint32_t row_words, hipStream_t stream);

// ||row||^2 of rows [first, first + count) into norm2 (cosine metrics only; no-op otherwise)
hipError_t launch_fill_norms(int metric, const View &v, uint32_t first, uint32_t count, float *norm2, hipStream_t stream);
// out[i] = metric(query, row(slots[i]))
hipError_t launch_gather(int metric, const View &v, const uint4 *query, const uint32_t *slots, uint32_t n, float *out,
                         hipStream_t stream);
// out[i*nb + j] = metric(a[i], b[j]) in the graph walk's exact reduction order
hipError_t launch_pairs(int metric, const uint4 *a, uint32_t na, const uint4 *b, uint32_t nb, uint32_t chunks, float *out,
                        hipStream_t stream);

// dense contraction + exact k-NN building blocks (bruteforce.hip)
hipError_t launch_row_norms(const uint4 *rows, uint32_t n, uint32_t chunks, float *out, hipStream_t stream);
hipError_t launch_dequant_f16(const uint4 *src, size_t nchunks, uint4 *dst, hipStream_t stream);
hipError_t launch_dequant_i8(const uint4 *src, size_t nchunks, uint4 *dst, hipStream_t stream);
hipError_t launch_dense(int metric, const uint4 *Q, uint32_t nq, const uint4 *B, uint32_t nb, uint32_t chunks, const float *qn,
                        const float *bn, float *out, uint32_t ldo, hipStream_t stream);
hipError_t launch_dense_topk(int metric, const uint4 *Q, uint32_t nq, const uint4 *B, uint32_t nb, uint32_t chunks, const float *qn,
                             const float *bn, const uint64_t *best, uint32_t kk, uint64_t *cand, uint32_t *cnt, uint32_t cap, uint32_t c_base,
                             hipStream_t stream);
hipError_t launch_select_candidates(uint32_t nq, uint64_t *best, uint32_t kk, const uint64_t *cand, uint32_t *cnt, uint32_t cap, uint32_t *overflow,
                                    hipStream_t stream);
hipError_t launch_select(const float *dist, uint32_t ldo, uint32_t nq, uint32_t ncols, uint32_t c_base, uint64_t *best, uint32_t kk,
                         hipStream_t stream);
hipError_t launch_rerank(int metric, const uint4 *Q, uint32_t nq, const uint4 *B, uint32_t chunks, const uint64_t *best, uint32_t kk,
                         uint32_t k, uint32_t *out_slots, float *out_dists, hipStream_t stream);

size_t search_lds_bytes(uint32_t chunks, uint32_t ef_cap, uint32_t M0, uint32_t vis_slots);
size_t insert_lds_bytes(uint32_t chunks, uint32_t efc, uint32_t M0, uint32_t vis_slots);

}  // namespace lgpu
