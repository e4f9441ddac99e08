// index.hpp -- the device-resident HNSW index behind the usearch-shaped C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <functional>
#include <vector>

#include "../../include/lantern_gpu.h"
#include "kernels.hpp"

namespace lgpu {

// what one scan has been handed so far (device slots): a continuation searches for |seen| + k results and returns the
// first k that were not returned before, so a scan never sees a row twice even though a wider search may rank the
// earlier rows differently
struct Cursor
{
    std::unordered_set<uint32_t> seen;
};

constexpr uint64_t kIndexMagic = 0x4C414E5445524E31ull;  // "LANTERN1": the first word of every live index handle

struct Index
{
    uint64_t magic = kIndexMagic;  // checked by every entry point (index.cpp H()): a stale, freed or foreign pointer handed over as a
                                   // usearch_index_t is refused with an error string instead of being dereferenced as an index
    // ---- configuration (usearch_init_options_t as Lantern fills it) -------------------------------
    usearch_init_options_t opts{};
    int      metric = 0;         // usearch_metric_kind_t
    int      mcode = 0;          // kernel metric code: metric, +100 for f16 storage (device_common.hpp)
    int      scalar = 0;         // STORAGE kind: usearch_scalar_f32_k, _f16_k (quant_bits=16) or _b1_k
    uint32_t words = 0;          // 4-byte words per vector as the caller supplies it
    uint32_t chunks = 0;         // 16-byte chunks per stored row (zero padded)
    uint32_t natural_chunks = 0; // the vector's own length in 16-byte chunks; < chunks where rows are stored at a widened stride (bit rows of 65 .. 127 bytes)
    uint32_t M = 16, M0 = 32, efc = 128, ef = 64;
    uint64_t seed = 42;
    size_t   add_batch_max = 8192, add_min_ratio = 16;
    int      search_waves = 0 /* automatic */, search_max_wg = 0, insert_waves = 4;
    int      search_vis_slots = -1;  // -1 = automatic size of the LDS visited set, 0 = HBM bitmap only

    // ---- quantised views of f32 input ----------------------------------------------------------------------------
    // quant_bits = 1 on real[] (options.c:154-155, test/sql/hnsw_sq.sql "binary > 0 quantization"): an l2sq index whose
    // rows are one bit per dimension, bit = (x > 0); sum (a - b)^2 over {0, 1} values IS the Hamming distance, so the
    // Hamming kernels run it exactly.  Callers still hand f32 arrays to usearch_add / usearch_search_ef.
    bool     b1_from_f32 = false;
    // pq = true (build.c:497-500, scan.c:75-81): the vector block holds every row's DECODING (concatenated centroids),
    // d_codes its num_subvectors code bytes (what the file / the pages carry: usearch_storage.cpp:29-31)
    bool     pq = false;
    uint32_t pq_S = 0, pq_C = 0, pq_subdim = 0;
    std::vector<float> h_codebook;       // [pq_C][dimensions]: row c = centroid c of every subvector, concatenated (pqtable.c:194-240)
    float   *d_codebook = nullptr;
    float   *d_centers = nullptr;        // [pq_S][pq_C][sub_floats]: the per-subvector centroid tables, rows zero padded to chunks
    uint8_t *d_codes = nullptr;          // [cap][pq_S]
    // COMPACT form of a pq index (lantern_gpu_pq_compact): the decodings are gone from HBM (d_vec == NULL) and searches run
    // ADC over the code rows (search_adc_kernel.hip); whatever needs rows again -- an insert, the exact search -- decodes
    // them back first (pq_expand_locked)
    bool     pq_compact = false;
    uint8_t *d_codes16 = nullptr;        // [n][pq_S16]: the code rows zero padded to whole 16-byte chunks
    uint32_t pq_S16 = 0;
    uint32_t pqd_inv = 0;                // compact form: the checked multiply-shift inverse of chunks-per-subvector (0: decode-on-the-fly not possible -> the ADC table walk)

    // ---- graph state ------------------------------------------------------------------------------
    size_t   n = 0, cap = 0;
    uint32_t entry = EMPTY;
    int      max_level = -1;
    size_t   upper_blocks = 0, upper_cap = 0;

    // ---- HBM ---------------------------------------------------------------------------------------
    uint4    *d_vec = nullptr;
    float    *d_norm2 = nullptr;  // ||row||^2 per stored row, cosine metrics only (device_common.hpp "cached row norms")
    uint64_t *d_labels = nullptr;
    uint8_t  *d_levels = nullptr;
    uint32_t *d_nbr0 = nullptr;
    uint32_t *d_upper_off = nullptr;
    uint32_t *d_upper_nbr = nullptr;
    float    *d_radius0 = nullptr, *d_radius_upper = nullptr;  // re-prune state per list (kernels.hpp RevlinkArgs::radius0)
    bool      radius_stale = false;  // lists changed without the state being maintained: reset it before the next use
    uint32_t *d_bitmaps = nullptr;
    size_t    bitmap_slots = 0, bm_words = 0;
    uint32_t *d_tickets = nullptr;  // ring of work tickets, one per launch in flight (kernels.hpp SearchArgs::ticket)
    uint32_t  ticket_next = 0;
    bool      use_tickets = true;   // LANTERN_GPU_TICKETS=0: static striding (tuning / debugging)
    unsigned long long *d_totals = nullptr;  // [0..1] search D,E  [2..4] insert D,E,refine  [5] revlink pairs

    // scratch (grown on demand)
    static const int kLanes = 8;  // lantern_gpu_search_batch_lane: batches one caller each may keep in flight side by side
    void  *d_scratch[ 12 + 2 * kLanes ] = {};  // [12 ..]: queries / answers of the lanes of lantern_gpu_search_batch_lane
    size_t scratch_bytes[ 12 + 2 * kLanes ] = {};
    hipStream_t lane_stream[ kLanes ] = {};  // created on first use
    char       *lane_host[ kLanes + 1 ] = {};  // page-locked staging of queries and answers: the lanes (one caller each), [kLanes] lantern_gpu_search_batch (under mu)
    size_t      lane_host_bytes[ kLanes + 1 ] = {};

    // ---- host mirrors ------------------------------------------------------------------------------
    std::vector<uint64_t> labels;
    std::vector<uint8_t>  levels;
    std::vector<uint32_t> upper_off;

    // ---- buffered inserts --------------------------------------------------------------------------
    std::mutex            mu;  // add_raw is called from N threads on one index (server.rs:333-356)
    std::vector<uint64_t> pend_labels;
    std::vector<uint32_t> pend_rows;    // chunks*4 words per pending vector, zero padded
    std::vector<int>      pend_levels;  // -1 = draw with level_for()

    // host copy of a batch's layout (sizes the launches and the exchanges; the device derives its own from the levels)
    std::vector<uint32_t> h_link_off;

    // ---- build profile (lantern_gpu_set_profiling): HIP events around the phases of every batch
    struct ProfBatch { hipEvent_t ev[ 6 ] = {}; };
    bool                    profiling = false;
    bool                    phase_profile = false;  // diagnostics: instrumented walk kernel (lantern_gpu_search_phase_profile)
    bool                    spec_profile = false;   // diagnostics: the instrumented latency-bound walk (lantern_gpu_spec_profile)
    uint32_t               *d_touched = nullptr;    // diagnostics: one bit per row evaluated by the instrumented searches (lantern_gpu_search_unique_rows)
    size_t                  touched_words = 0;
    bool                    unique_rows_on = false; // the bitmap is handed to a launch only in this mode, and only while it covers `cap`
    uint32_t               *d_trace = nullptr, *d_trace_count = nullptr;  // diagnostics: [trace_nq][trace_cap] + [trace_nq] (lantern_gpu_search_row_trace)
    size_t                  trace_nq = 0, trace_cap = 0;
    bool                    trace_on = false;
    float                   last_gather_ms = 0.f;   // kernel time of the last lantern_gpu_distance_gather launch (HIP events on the index stream)
    int                     last_search_grid = 0;   // workgroups of the last bandwidth-bound search launch (lantern_gpu_last_search_grid)
    std::deque<ProfBatch>   prof_pending;
    std::vector<hipEvent_t> prof_free;
    lantern_gpu_build_profile prof{};

    // ---- streaming continuation of usearch_search_ef (scan.c:273-281) ----------------------------
    // In the reference every scan owns its own usearch handle (scan.c:99), so "what this scan has been handed so far"
    // is per handle there.  Here ONE resident index serves many scans, so that state is a Cursor owned by the scan
    // (lantern_scan, a scan-service connection, lantern_gpu_cursor_*); usearch_search_ef itself -- one handle, one
    // scan, as in the reference -- uses the index's default cursor.
    Cursor default_cursor;

    // ---- page slots of a mirrored index (usearch_view_mem_lazy): device id -> the node's 48-bit slot in the PostgreSQL
    // pages (an ItemPointer, external_index.c:380-409), and back.  usearch_add_external writes the lists it changes
    // through retriever_mut in that form.
    bool                                   page_mode = false;  // attached through usearch_view_mem_lazy
    // the header's node count at attach time and the mirror's: they differ by the nodes no walk can reach (a re-prune
    // may drop a node's last in-link; such nodes stay in the pages and in the header's count)
    size_t                                 page_declared = 0, page_attach_n = 0;
    std::vector<uint64_t>                  page_slots;
    std::unordered_map<uint64_t, uint32_t> page_ids;
    // A mirror kept by the cache (mirror_cache.cpp) outlives the RetrieverCtx it was built with and is shared by holders that
    // each bring their own (scan.c:34,132, insert.c:130,247): its callbacks are looked up PER HOLDER -- a holder is a host
    // thread (a PostgreSQL backend is one; a threaded service runs one holder per thread) -- instead of in `opts`.
    struct HolderBinding { usearch_node_retriever_t retriever = nullptr, retriever_mut = nullptr; void *ctx = nullptr; };
    bool                                               holder_bound = false;  // true: `holders` decides, `opts.retriever*` are unused
    std::unordered_map<std::thread::id, HolderBinding> holders;
    // the calling thread's callbacks (ix->mu held)
    HolderBinding current_holder() const
    {
        if(!holder_bound) return HolderBinding{ opts.retriever, opts.retriever_mut, opts.retriever_ctx };
        auto it = holders.find(std::this_thread::get_id());
        return it == holders.end() ? HolderBinding{} : it->second;
    }

    // ---- single-query path (usearch_search_ef): one pinned, device-mapped block [query row | labels | distances |
    // slots | count] -- the kernel reads the query from it and writes the answer into it, so a lone query costs one
    // launch and one stream synchronisation, no copy commands
    char  *h_single = nullptr, *h_single_dev = nullptr;  // host / device address of the block
    size_t h_single_bytes = 0;
    // page-locked staging of a SMALL insertion (ldb_aminsert's one row): rows | labels | upper offsets | levels are read
    // from here by ONE kernel over the bus (k_stage_small): no copies, no wait before the batch's kernels
    char  *h_stage = nullptr, *h_stage_dev = nullptr;  // host / device address of the block
    size_t h_stage_bytes = 0;

    // ---- launches that share per-index scratch are ordered across streams.  The walk kernels use per-workgroup visited
    // bitmaps indexed by blockIdx only, so two launches may overlap only if they use different bitmap slabs.  SEARCH launches
    // have two slabs ("launch slots"): two batches on two streams run side by side (the second fills the machine while the
    // first one's longest walks drain); a third waits for the slot it reuses.  INSERT batches mutate the graph: they wait for
    // every search in flight, and searches on other streams wait for them.
    static const int kSearchSlots = kLanes;  // one slab of visited bitmaps per search launch in flight (allocated on first use)
    uint32_t   *slot_bitmaps[ kSearchSlots ] = {};  // [0] aliases d_bitmaps (the slab inserts use too)
    size_t      slot_rows[ kSearchSlots ] = {}, slot_words[ kSearchSlots ] = {};
    hipEvent_t  slot_done[ kSearchSlots ] = {};
    hipStream_t slot_stream[ kSearchSlots ] = {};
    bool        slot_pending[ kSearchSlots ] = {};
    unsigned    slot_next = 0;
    hipEvent_t  insert_done = nullptr;
    bool        insert_pending = false;

    // ---- counters ----------------------------------------------------------------------------------
    uint64_t c_search_queries = 0, c_add_vectors = 0, c_add_batches = 0, c_solo_launches = 0;

    hipStream_t stream = nullptr;
    int         device = 0;
    int         num_cus = 256;
    std::string err;

    View view() const;
};

// implemented in index.cpp
const char *set_err(Index *ix, const std::string &msg);
bool        flush_locked(Index *ix);            // false -> ix->err set
bool        ensure_bitmaps(Index *ix, size_t slots);
bool        pq_encode_rows(Index *ix, size_t first, size_t count);  // raw f32 rows [first, first+count) in d_vec -> codes + decodings
bool        pq_decode_rows(Index *ix, size_t first, size_t count);  // d_codes -> d_vec
bool        pq_compact_locked(Index *ix);  // drop the decodings, keep the codes (searches: ADC)
bool        pq_expand_locked(Index *ix);   // decode them back (no-op unless compact)
bool        fill_norms(Index *ix, size_t first, size_t count);  // after rows [first, first + count) are in d_vec
void       *scratch(Index *ix, int which, size_t bytes);
bool        pad_row(const Index *ix, const void *vec, int kind_in, uint32_t *dst);
size_t      input_bytes(const Index *ix, int kind_in);
bool        kind_accepted(const Index *ix, int kind_in);
int         search_grid(const Index *ix, size_t nq, int waves, int waves_per_cu);
// `done`: NULL, or a device-visible counter the kernel bumps per finished query; the caller then WAITS ON IT (not on the
// stream) and no completion event is queued behind the launch
bool        run_search_device(Index *ix, const uint4 *d_queries, size_t nq, size_t k, size_t ef, size_t skip,
                              uint64_t *d_labels, float *d_dists, uint32_t *d_slots, uint32_t *d_counts, uint64_t *d_D,
                              uint64_t *d_E, hipStream_t stream, int waves, uint32_t *done = nullptr,
                       uint32_t *done_flags = nullptr);

// one usearch_search_ef on behalf of `cur` (the caller holds ix->mu); returns the number of results
size_t      search_one_locked(Index *ix, Cursor *cur, const void *query, int kind, size_t k, size_t ef, bool streaming,
                              uint64_t *labels, float *distances);
// usearch_size of the index: for a mirror, the header's count plus what was inserted since
inline size_t logical_size(const Index *ix) { return ix->page_mode ? ix->page_declared + (ix->n - ix->page_attach_n) : ix->n; }
void        prof_resolve(Index *ix, size_t keep);          // fold finished batches' event times into ix->prof
bool        order_launch(Index *ix, hipStream_t stream);   // before an insert batch (exclusive use of the index)
bool        record_launch(Index *ix, hipStream_t stream);  // behind it
int         acquire_search_slot(Index *ix, hipStream_t stream, size_t grid);  // before a search launch: its bitmap slab (< 0: error)
bool        release_search_slot(Index *ix, int slot, hipStream_t stream);     // behind it
bool        import_graph_locked(Index *ix, size_t size, const void *vectors, const uint64_t *labels, const uint8_t *levels,
                                const uint32_t *nbr0, const uint32_t *upper_off, const uint32_t *upper_nbr, uint32_t entry_slot,
                                int32_t max_level, bool vectors_are_codes = false);  // pq: `vectors` = num_subvectors code bytes per row

// usearch-format serialisation (usearch_file.cpp)
size_t serialized_length(Index *ix);
bool   serialize(Index *ix, char *buf, size_t len);
using SpanSink = std::function<bool(const lantern_gpu_span *, size_t)>;
bool   serialize_stream(Index *ix, const SpanSink &sink);  // the same bytes as a sequence of spans (rows staged in chunks)
bool   deserialize(Index *ix, const char *buf, size_t len);

}  // namespace lgpu
