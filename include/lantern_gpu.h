/*
 * lantern_gpu.h -- C ABI of the MI355X-native HNSW distance-evaluation path for Lantern.
 *
 * Two groups of entry points:
 *
 *  (1) The usearch C API exactly as Lantern calls it (boundary B1 of SURVEY.md section 8b).
 *      Lantern static-links usearch's c/lib.cpp into lantern.so (lantern_hnsw/CMakeLists.txt:89,
 *      115-120); usearch.h itself is NOT in the reference tree (un-vendored submodule), so the
 *      prototypes below are reconstructed from the call sites cited next to each one.  A
 *      maintainer links liblantern_gpu.so instead of c/lib.cpp and recompiles.
 *
 *  (2) lantern_gpu_* / lantern_scan_* / lantern_*_dist: batched and device-resident forms the
 *      reference lacks, the amgettuple paging shim (scan.c:167-338) and the SQL-callable
 *      distance functions' semantics (hnsw.c:296-405).
 *
 * Conventions (same as the reference's): errors are a `const char*` out-parameter, NULL on
 * success, pointing at a static or index-owned string on failure (hnsw.c:341-343, scan.c:100,
 * build.c:545-551); result arrays are caller-allocated (scan.c:207-212); vectors are borrowed
 * for the duration of the call.  There is NO CPU fallback: without a HIP device every compute
 * entry point fails with an error string.
 */
#ifndef LANTERN_GPU_H
#define LANTERN_GPU_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LANTERN_GPU_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* (1) usearch C API as used by Lantern                                                        */
/* ------------------------------------------------------------------------------------------ */

typedef void       *usearch_index_t;
typedef uint64_t    usearch_key_t;
typedef uint64_t    usearch_label_t; /* 6-byte heap TID packed in u64: utils.c:69-75 */
typedef float       usearch_distance_t;
typedef const char *usearch_error_t;

/* enum values pinned by lantern_cli/src/external_index/cli.rs:56-69 and server.rs:94-101 */
typedef enum usearch_metric_kind_t {
    usearch_metric_unknown_k = 0,
    usearch_metric_cos_k = 1,
    usearch_metric_l2sq_k = 3,
    usearch_metric_hamming_k = 8
} usearch_metric_kind_t;

typedef enum usearch_scalar_kind_t {
    usearch_scalar_unknown_k = 0,
    usearch_scalar_f32_k = 1,
    usearch_scalar_f64_k = 2,
    usearch_scalar_f16_k = 3,
    usearch_scalar_i8_k = 4,
    usearch_scalar_b1_k = 5
} usearch_scalar_kind_t;

/* external_index.c:613-697: slot -> pointer to the node tape */
typedef void *(*usearch_node_retriever_t)(void *ctx, uint64_t slot);

/* Fields are the ones Lantern sets: utils.c:57-67, scan.c:60-96, build.c:495-515, insert.c:116-132 */
typedef struct usearch_init_options_t
{
    usearch_metric_kind_t    metric_kind;
    void                    *metric; /* custom metric; Lantern always passes NULL (utils.c:63) */
    usearch_scalar_kind_t    quantization;
    size_t                   dimensions; /* f32 scalars, or BITS for hamming (scan.c:84-88) */
    size_t                   connectivity;
    size_t                   expansion_add;
    size_t                   expansion_search;
    size_t                   num_threads;
    bool                     pq;
    size_t                   num_centroids;
    size_t                   num_subvectors;
    void                    *retriever_ctx;
    usearch_node_retriever_t retriever;
    usearch_node_retriever_t retriever_mut;
} usearch_init_options_t;

/* usearch_storage.cpp:19-32,63-81 read these */
typedef struct metadata_t
{
    size_t                 neighbors_bytes;      /* 4 + M*6      (upper levels) */
    size_t                 neighbors_base_bytes; /* 4 + 2M*6     (level 0)      */
    double                 inverse_log_connectivity;
    size_t                 connectivity;
    size_t                 dimensions;
    usearch_init_options_t init_options;
} metadata_t;

#define USEARCH_SEARCH_EF_INVALID_VALUE 0 /* options.c:341: 0 = "use the index's ef" */
#define LANTERN_SLOT_SIZE 6               /* validate_index.c:39, hnsw.h:42-49 */
#define USEARCH_HEADER_SIZE 136           /* external_index.h:29-30 */
#define USEARCH_EMPTY_INDEX_SIZE USEARCH_HEADER_SIZE /* build.c:678 */

/* scan.c:99, build.c:517,675, insert.c:142.
 * quantization: f32 / f16 / i8 storage of real[] input (quant_bits 32 / 16 / 8), or b1 -- for hamming (integer[] input:
 *   bits) and for l2sq over real[] (quant_bits = 1, options.c:154-155: bit = (x > 0); the l2sq distance of {0,1} vectors
 *   is their Hamming distance).
 * pq = true (build.c:497-500, scan.c:75-81): pq_codebook = num_centroids rows of `dimensions` floats, row c = centroid c
 *   of every subvector, concatenated (pqtable.c:194-240); copied.  Every stored vector is replaced by its quantisation
 *   (per subvector the nearest centroid under the index metric, first minimum wins: product_quantization.c:80-124);
 *   all distances are distances to / between the DECODED vectors; the file and the pages carry num_subvectors code bytes
 *   per node (usearch_storage.cpp:29-31).  The device keeps the decodings resident in HBM next to the codes. */
LANTERN_GPU_EXPORT usearch_index_t usearch_init(usearch_init_options_t *, float *pq_codebook, usearch_error_t *);
/* scan.c:131, build.c:450,549,597,684, insert.c:237 */
LANTERN_GPU_EXPORT void usearch_free(usearch_index_t, usearch_error_t *);
/* build.c:124,543, insert.c:182 */
LANTERN_GPU_EXPORT void usearch_reserve(usearch_index_t, size_t capacity, usearch_error_t *);
/* scan.c:246, build.c:117,121,558,590, insert.c:148 */
LANTERN_GPU_EXPORT size_t usearch_size(usearch_index_t, usearch_error_t *);
/* build.c:116 */
LANTERN_GPU_EXPORT size_t usearch_capacity(usearch_index_t, usearch_error_t *);
/* server.rs:222-229 (Index::dimensions) */
LANTERN_GPU_EXPORT size_t usearch_dimensions(usearch_index_t, usearch_error_t *);
/* build.c:128; Rust add_raw server.rs:349.  Inserts are buffered and applied in batches on the
 * device (see lantern_gpu_set_add_batch); any reader (search/size/save) flushes first. */
LANTERN_GPU_EXPORT void usearch_add(usearch_index_t, usearch_label_t, const void *vector, usearch_scalar_kind_t,
                                    usearch_error_t *);
/* insert.c:209 (ldb_aminsert): one sequential insertion at the caller-drawn level (insert.c:32-46) of a node whose tape
 * the caller allocated inside a PostgreSQL page (`node_tape`, headed by usearch_init_node, usearch_storage.cpp:34-44)
 * at 48-bit page slot `slot`.  The node is linked into the HBM mirror; its own lists and stored vector are written into
 * `node_tape`, the re-written lists of the nodes it linked to into their tapes through init_options.retriever_mut
 * (external_index.c:673-697).  On an index that was not attached with usearch_view_mem_lazy the slots written are the
 * sequential ids of the file format. */
LANTERN_GPU_EXPORT void usearch_add_external(usearch_index_t, usearch_label_t, const void *vector, void *node_tape,
                                             usearch_scalar_kind_t, int16_t level, uint64_t slot, usearch_error_t *);
/* scan.c:220-228,273-281.  ef == 0 -> index default.  streaming == true returns the NEXT k results of
 * the same query: the index remembers what it handed out since the last non-streaming call, searches for
 * that many + k, and returns the first k that were not returned before (never a row twice). */
LANTERN_GPU_EXPORT size_t usearch_search_ef(usearch_index_t, const void *query, usearch_scalar_kind_t, size_t k,
                                            size_t ef, bool streaming, usearch_label_t *labels, float *distances,
                                            usearch_error_t *);
/* The per-scan half of the streaming contract.  In the reference every scan owns its own usearch handle (scan.c:99),
 * so usearch_search_ef's "what was handed out since the last non-streaming call" is per scan there.  When ONE resident
 * index serves many scans, each scan opens a cursor and searches through it; usearch_search_ef(h, ...) is the same call
 * on the index's built-in cursor.  Cursors must be closed before usearch_free. */
typedef struct lantern_gpu_cursor lantern_gpu_cursor_t;
LANTERN_GPU_EXPORT lantern_gpu_cursor_t *lantern_gpu_cursor_open(usearch_index_t, usearch_error_t *);
LANTERN_GPU_EXPORT size_t lantern_gpu_cursor_search(lantern_gpu_cursor_t *, const void *query, usearch_scalar_kind_t, size_t k,
                                                    size_t ef, bool streaming, usearch_label_t *labels, float *distances,
                                                    usearch_error_t *);
LANTERN_GPU_EXPORT size_t lantern_gpu_cursor_seen(lantern_gpu_cursor_t *); /* rows handed out since the last non-streaming call */
LANTERN_GPU_EXPORT void   lantern_gpu_cursor_close(lantern_gpu_cursor_t *);
/* hnsw.c:317,326,340; product_quantization.c:102,185.  One pair, evaluated on the device. */
LANTERN_GPU_EXPORT float usearch_distance(const void *a, const void *b, usearch_scalar_kind_t, size_t dims,
                                          usearch_metric_kind_t, usearch_error_t *);
/* build.c:561, insert.c:160, utils.c:91 */
LANTERN_GPU_EXPORT metadata_t usearch_index_metadata(usearch_index_t, usearch_error_t *);
/* build.c:583: usearch-format file = 136-byte header + node tapes in slot order (App. B) */
LANTERN_GPU_EXPORT void usearch_save(usearch_index_t, const char *path, usearch_error_t *);
/* build.c:679 */
LANTERN_GPU_EXPORT void usearch_save_buffer(usearch_index_t, char *buffer, size_t length, usearch_error_t *);
/* Rust Index::load_from_buffer (external_index_server_test.rs:271-314) / usearch_load */
LANTERN_GPU_EXPORT void usearch_load(usearch_index_t, const char *path, usearch_error_t *);
LANTERN_GPU_EXPORT void usearch_load_buffer(usearch_index_t, const char *buffer, size_t length, usearch_error_t *);
LANTERN_GPU_EXPORT size_t usearch_serialized_length(usearch_index_t, usearch_error_t *);
/* scan.c:110, insert.c:151: attach to an index that lives in PostgreSQL pages.  The reachable graph is
 * walked ONCE through init_options.retriever (slot -> node tape, external_index.c:613-671) from the
 * header's entry slot and mirrored into HBM; neighbour slots are the 6-byte ItemPointers of
 * external_index.c:380-409.  Searches then run on the mirror and return the nodes' labels. */
LANTERN_GPU_EXPORT void usearch_view_mem_lazy(usearch_index_t, char *header136, usearch_error_t *);
/* insert.c:214: refresh size / max_level / entry slot in the header copy (the entry slot in the form the index was
 * attached in: a 48-bit page slot after usearch_view_mem_lazy, a sequential id otherwise) */
LANTERN_GPU_EXPORT void usearch_update_header(usearch_index_t, char *header136, usearch_error_t *);
/* external_index.c:411,417 */
LANTERN_GPU_EXPORT uint64_t usearch_header_get_entry_slot(char *header136);
LANTERN_GPU_EXPORT void     usearch_header_set_entry_slot(char *header136, uint64_t slot);

/* Lantern's node-tape helpers (lantern_hnsw/src/hnsw/usearch_storage.hpp:9-23).  They are Lantern's own functions, but the
 * reference compiles them against usearch's C++ templates (usearch_storage.cpp:2-16: node_at<>), so whatever replaces usearch
 * brings them: same names, same signatures (`ldb_unaligned_slot_union_t *` = the 6-byte slots of hnsw.h:42-49, returned as
 * void * here; Lantern keeps its own declaration).  Host-only byte arithmetic over
 *   [key u64][level u16] { [count u32][slot 6 B x cap] } x (level + 1) [vector bytes]        (validate_index.c:105-226)
 * Callers: external_index.c:96-97,394-398,488, insert.c:207, delete.c:54-58, utils.c:93. */
#ifndef HNSW_USEARCH_STORAGE_H
LANTERN_GPU_EXPORT uint32_t UsearchNodeBytes(const metadata_t *metadata, int vector_bytes, int level);
LANTERN_GPU_EXPORT void usearch_init_node(metadata_t *meta, char *tape, usearch_key_t key, uint32_t level, uint64_t slot_id,
                                          void *vector, size_t vector_len);
LANTERN_GPU_EXPORT uint32_t node_tuple_size(char *node, uint32_t vector_dim, const metadata_t *meta);
LANTERN_GPU_EXPORT usearch_label_t label_from_node(char *node);
LANTERN_GPU_EXPORT unsigned long level_from_node(char *node);
LANTERN_GPU_EXPORT void reset_node_label(char *node);
LANTERN_GPU_EXPORT void *get_node_neighbors_mut(const metadata_t *meta, char *node, uint32_t level, uint32_t *neighbors_count);
#endif
/* reloption quant_bits -> scalar kind, as ldb_HnswGetScalarKind (options.c:137-158); *err carries the reference's text for a value the
 * enum reloption rejects (hnsw_sq.out:30-35) or one it has not implemented (4, 2: options.c:150-153).  unset = no reloption given. */
LANTERN_GPU_EXPORT usearch_scalar_kind_t lantern_quant_bits_scalar_kind(int quant_bits, bool unset, usearch_error_t *err);

/* ------------------------------------------------------------------------------------------ */
/* (2) batched / device-resident forms                                                         */
/* ------------------------------------------------------------------------------------------ */

LANTERN_GPU_EXPORT const char *lantern_gpu_version(void);
LANTERN_GPU_EXPORT int         lantern_gpu_device_count(void);

/* seed of the level draw (insert.c:32-46 uses PG's global PRNG; here levels are a stateless
 * hash of (seed, slot) so that builds are reproducible).  Call before the first add. */
LANTERN_GPU_EXPORT void lantern_gpu_set_seed(usearch_index_t, uint64_t seed, usearch_error_t *);
/* Insert batching: at most `max_batch` pending vectors are inserted per device pass and never
 * more than size/min_ratio (so early inserts are near-sequential).  max_batch = 1 reproduces
 * usearch_add's strictly sequential semantics.  Defaults: 8192, 16. */
LANTERN_GPU_EXPORT void lantern_gpu_set_add_batch(usearch_index_t, size_t max_batch, size_t min_ratio,
                                                  usearch_error_t *);
/* many inserts in one call (the external indexer's row stream, server.rs:214-267) */
LANTERN_GPU_EXPORT void lantern_gpu_add_many(usearch_index_t, const usearch_label_t *labels, const void *vectors,
                                             size_t n, usearch_scalar_kind_t, usearch_error_t *);
/* The two host-side rules that make a build reproducible by any builder (pure functions, no device):
 * the level of the node at `slot` -- floor(-ln(U) / ln(M)) as insert.c:32-46, U a hash of (seed, slot) -- and the
 * prefix of the pending vectors that forms the next device batch (<= max_batch, <= size / min_ratio, a vector that
 * raises the top level goes alone). */
LANTERN_GPU_EXPORT int    lantern_gpu_level_for(uint64_t seed, uint64_t slot, uint32_t connectivity);
LANTERN_GPU_EXPORT size_t lantern_gpu_plan_batch(size_t current_size, int max_level, const int *pending_levels, size_t pending,
                                                 size_t max_batch, size_t min_ratio);
/* apply all buffered inserts now */
LANTERN_GPU_EXPORT void lantern_gpu_flush(usearch_index_t, usearch_error_t *);
/* page-locked host memory for rows that are about to be handed to lantern_gpu_add_many / usearch_add: the upload runs at the
 * host link's rate instead of through the runtime's pageable staging (the indexing server's row chunks: 50 MB each).  NULL on
 * failure (callers fall back to ordinary memory). */
LANTERN_GPU_EXPORT void *lantern_gpu_host_alloc(size_t bytes);
LANTERN_GPU_EXPORT void  lantern_gpu_host_free(void *);
/* usearch_save_buffer without the buffer: the index file (the bytes usearch_save_buffer would produce) handed to `write` as a
 * sequence of spans, up to 1024 per call, in file order -- header, then per node its formatted prefix and its vector bytes
 * straight out of a page-locked staging buffer that the rows reach in ~64 MB chunks, the next chunk's copy overlapping this
 * one's consumption.  A span is layout-compatible with struct iovec: the indexing server passes them to writev(2)
 * (server.rs:388-422 sends the file it has just written to disk).  `write` returns 0 to go on, anything else aborts.  The index
 * stays locked until the last span has been consumed: other calls on it wait for a slow consumer. */
typedef struct lantern_gpu_span { const void *data; size_t size; } lantern_gpu_span;
typedef int (*lantern_gpu_write_fn)(void *ctx, const lantern_gpu_span *spans, size_t count);
LANTERN_GPU_EXPORT void lantern_gpu_save_stream(usearch_index_t, lantern_gpu_write_fn write, void *ctx, usearch_error_t *);
/* usearch_add with a caller-drawn level (insert.c:32-46); usearch_add_external is this plus the write-back to the pages */
LANTERN_GPU_EXPORT void lantern_gpu_add_with_level(usearch_index_t, usearch_label_t, const void *vector,
                                                   usearch_scalar_kind_t, int level, usearch_error_t *);

/* nq independent usearch_search_ef calls in one launch; host buffers.
 * labels/distances: nq x k (unused tail: label 0, +inf); counts: nq (may be NULL). */
LANTERN_GPU_EXPORT void lantern_gpu_search_batch(usearch_index_t, const void *queries, size_t nq,
                                                 usearch_scalar_kind_t, size_t k, size_t ef, usearch_label_t *labels,
                                                 float *distances, uint32_t *counts, usearch_error_t *);
/* The same for a caller that keeps up to EIGHT batches in flight (`lane` 0 .. 7, one caller thread per lane): each lane has its
 * own stream and staging buffers inside the index, and the wait for the answers does not hold the index's lock, so the lanes'
 * launches overlap on the device (the scan-side service below runs one dispatcher per lane over this). */
LANTERN_GPU_EXPORT void lantern_gpu_search_batch_lane(usearch_index_t, int lane, const void *queries, size_t nq,
                                                      usearch_scalar_kind_t, size_t k, size_t ef, usearch_label_t *labels,
                                                      float *distances, uint32_t *counts, usearch_error_t *);
/* The same with every answer handed on AS ITS OWN WALK ENDS: `done(ctx, which, count)` is called from the calling thread for the
 * `count` queries (indices in `which`) that finished since the last call -- their rows of labels / distances / counts are filled in by
 * then -- until every query has been handed on; the function returns after the last call.  The walks of a launch differ in length by
 * 2x and more; a service that answers each client when ITS walk is over does not make it wait for the longest one of its batch. */
typedef void (*lantern_gpu_queries_done_fn)(void *ctx, const uint32_t *which, size_t count);
LANTERN_GPU_EXPORT void lantern_gpu_search_batch_lane_notify(usearch_index_t, int lane, const void *queries, size_t nq,
                                                             usearch_scalar_kind_t, size_t k, size_t ef, usearch_label_t *labels,
                                                             float *distances, uint32_t *counts, lantern_gpu_queries_done_fn done, void *done_ctx,
                                                             usearch_error_t *);
/* Same, every buffer already in device memory, asynchronous on `stream` (a hipStream_t; NULL =
 * the default stream).  slots (u32 internal ids), counts, dist_evals (D) and expansions (E) may
 * be NULL.  `skip` drops that many leading results per query (streaming continuation).
 * Launches on DIFFERENT streams overlap, two at a time (each gets its own slab of visited bitmaps; a third queues behind
 * the slab it reuses): with independent batches -- a serving workload -- the second fills the machine while the first one's
 * longest walks drain (1M x 768 cosine, 1024-query batches: 0.68 -> 0.87 of the HBM peak).  Insert batches run behind every
 * search in flight, and searches queued on other streams behind them; no host synchronisation is involved. */
LANTERN_GPU_EXPORT void lantern_gpu_search_batch_device(usearch_index_t, const void *d_queries, size_t nq, size_t k,
                                                        size_t ef, size_t skip, uint64_t *d_labels, float *d_distances,
                                                        uint32_t *d_slots, uint32_t *d_counts, uint64_t *d_dist_evals,
                                                        uint64_t *d_expansions, void *stream, usearch_error_t *);
/* The same with the caller's query row stride stated: query i is read at d_queries + i * query_stride_bytes, which must equal
 * lantern_gpu_row_bytes() -- anything else is refused ("the query row stride does not match ...") instead of being read at
 * the wrong offsets and past the end of the caller's buffer.  lantern_gpu_search_batch_device (no stride argument) is accepted
 * only for indexes whose stored stride is the vector's own length rounded up to 16 bytes; for an index that widens its rows
 * (bit rows of 65 .. 127 bytes are stored at 128) it fails and names this entry point. */
LANTERN_GPU_EXPORT void lantern_gpu_search_batch_device_strided(usearch_index_t, const void *d_queries, size_t query_stride_bytes,
                                                                size_t nq, size_t k, size_t ef, size_t skip, uint64_t *d_labels,
                                                                float *d_distances, uint32_t *d_slots, uint32_t *d_counts,
                                                                uint64_t *d_dist_evals, uint64_t *d_expansions, void *stream,
                                                                usearch_error_t *);
/* kernel shape of the search launch: waves per query (1..8; 0 = automatic: 4 when the batch fills the chip, up to 8 for
 * smaller batches) and resident workgroups (0 = auto) */
LANTERN_GPU_EXPORT void lantern_gpu_set_search_shape(usearch_index_t, int waves_per_query, int max_workgroups,
                                                     usearch_error_t *);

/* Exact k-NN over the index's vectors (the seq-scan `ORDER BY v <op> q LIMIT k`; ground truth
 * for recall, index_autotune/mod.rs:196-203).  Host buffers; slots: nq x k ascending by
 * (distance, slot). */
LANTERN_GPU_EXPORT void lantern_gpu_exact_search(usearch_index_t, const void *queries, size_t nq, size_t k,
                                                 uint32_t *slots, float *distances, usearch_error_t *);

/* Diagnostic for the measurement harness: HIP events around every launch of the fp32-MFMA contraction (k_dense_f32) inside
 * lantern_gpu_exact_search / _assign_to_clusters / _distance_matrix.  on != 0 starts recording; on == 0 stops and writes up to
 * `cap` records -- ms[i], rows[i] x cols[i] (queries x base rows of the launch), fused[i] (1: the launch with the fused top-k
 * epilogue) -- and returns how many launches were recorded.  Process-wide; any pointer may be NULL. */
LANTERN_GPU_EXPORT size_t lantern_gpu_dense_profile(int on, float *ms, uint32_t *rows, uint32_t *cols, uint32_t *fused, size_t cap);

/* Diagnostic for the measurement harness: the memory objects every query of a launch asks for, in order (rows evaluated, adjacency
 * lists read) -- the input of the cache model behind bench.py's roofline.frac_dram_model (lantern_amd/tools/cache_model.c).
 * on = 1: allocate nq x per_query_cap entries and trace the following launches of at most nq queries (the instrumented walk:
 * f32 l2sq / cos, as lantern_gpu_search_unique_rows; same answers, D and E).  on = 0: copy the LAST traced launch's trace
 * (nq x per_query_cap u32) and counts (nq u32; above per_query_cap: the tail was dropped) out, free the buffers, switch it off.
 * Entry = slot (a row evaluation) | 0x80000000 (level-0 list of that node read) | 0xC0000000 (an upper-level list). */
LANTERN_GPU_EXPORT void lantern_gpu_search_row_trace(usearch_index_t, int on, size_t nq, size_t per_query_cap, uint32_t *trace,
                                                     uint32_t *counts, usearch_error_t *);

/* workgroups of the last search launch = walks resident on the device at a time (the cache model's `walkers`) */
LANTERN_GPU_EXPORT int lantern_gpu_last_search_grid(usearch_index_t, usearch_error_t *);

/* Gathered distances: out[i] = metric(query, row(slots[i])) -- the kernel the graph walk is
 * made of, exposed for tests and profiling.  Host buffers. */
LANTERN_GPU_EXPORT void lantern_gpu_distance_gather(usearch_index_t, const void *query, const uint32_t *slots,
                                                    size_t n, float *out, usearch_error_t *);
/* kernel time (ms, HIP events on the index stream) of the last lantern_gpu_distance_gather launch of this index */
LANTERN_GPU_EXPORT float lantern_gpu_last_gather_ms(usearch_index_t, usearch_error_t *);
/* Dense na x nb distance matrix between two host matrices (f32 rows of `dims` scalars, or u32
 * words for hamming with dims = bits).  `exact_order` != 0 uses the per-pair reduction order of
 * the graph walk (bit-identical to usearch_distance); 0 uses the fp32-MFMA contraction. */
LANTERN_GPU_EXPORT void lantern_gpu_distance_matrix(const void *a, size_t na, const void *b, size_t nb,
                                                    usearch_scalar_kind_t, size_t dims, usearch_metric_kind_t,
                                                    int exact_order, float *out, usearch_error_t *);

/* PQ k-means assignment, product_quantization.c:80-124 (assign_to_clusters): for every row i of `dataset`
 * (n rows of row_dims f32) the nearest of k centroids under `metric` over the subvector
 * [subvector_start, subvector_start + subvector_dim); the first minimum wins, as in the reference's
 * strict-< loop.  centers: k x subvector_dim f32.  out_distance may be NULL.  cos / l2sq only. */
LANTERN_GPU_EXPORT void lantern_gpu_assign_to_clusters(const float *dataset, size_t n, size_t row_dims, size_t subvector_start,
                                                       size_t subvector_dim, const float *centers, size_t k,
                                                       usearch_metric_kind_t metric, uint32_t *out_cluster,
                                                       float *out_distance, usearch_error_t *);

/* flat graph exchange (tests, CPU-baseline timing on the identical graph, sharded serving) */
typedef struct lantern_gpu_graph_info
{
    size_t   size;
    size_t   upper_blocks; /* sum of node levels */
    uint32_t connectivity;
    uint32_t entry_slot;
    int32_t  max_level;
    uint32_t vector_words; /* 4-byte words per stored vector */
} lantern_gpu_graph_info;
LANTERN_GPU_EXPORT lantern_gpu_graph_info lantern_gpu_graph_info_get(usearch_index_t, usearch_error_t *);
/* any output pointer may be NULL.  levels[size] u8; nbr0[size][2M] u32 (0xFFFFFFFF = empty);
 * upper_off[size] u32; upper_nbr[upper_blocks][M] u32; labels[size] u64; vectors[size][words] */
LANTERN_GPU_EXPORT void lantern_gpu_export_graph(usearch_index_t, uint8_t *levels, uint32_t *nbr0, uint32_t *upper_off,
                                                 uint32_t *upper_nbr, uint64_t *labels, void *vectors,
                                                 usearch_error_t *);
/* pq indexes: the code bytes, codes[size][num_subvectors] (export_graph's `vectors` are the decoded f32 rows) */
LANTERN_GPU_EXPORT void lantern_gpu_export_codes(usearch_index_t, uint8_t *codes, usearch_error_t *);
/* pq = true indexes, COMPACT form.  A node of a PQ index carries num_subvectors code bytes (usearch_storage.cpp:29-31); the
 * device keeps every row's DECODING next to the codes because adding evaluates stored row against stored row.  A read-mostly
 * index -- the scan-side mirror of a built index (scan.c:75-110) -- gives the decodings up: lantern_gpu_pq_compact frees them
 * (10M x 768 at 96 subvectors: 0.96 GB of rows instead of 30.7 GB) and every search from then on evaluates rows by asymmetric
 * distance computation over the code bytes: a per-query table of partial sums (subvector x centroid) in LDS, num_subvectors
 * table entries added per row (lantern_amd/csrc/search_adc_kernel.hip; summation order defined there and restated by the
 * oracle -- distances agree with the decoded-row path to 1e-5 relative, not bit for bit: PARITY UNPINNED BY THE REFERENCE).
 * Anything that needs rows again (usearch_add, the exact search, an export with vectors) decodes them back first, as does
 * lantern_gpu_pq_expand.  LANTERN_GPU_PQ_COMPACT=1 compacts a pq index right after it is loaded or mirrored. */
LANTERN_GPU_EXPORT void lantern_gpu_pq_compact(usearch_index_t, usearch_error_t *);
LANTERN_GPU_EXPORT void lantern_gpu_pq_expand(usearch_index_t, usearch_error_t *);
/* HBM held by the index: the vector block (the code rows of a compact pq index) | adjacency, labels, levels, norms, codes */
/* bytes of one stored row in device memory = the row stride of `d_queries` in lantern_gpu_search_batch_device: the vector zero padded
 * to whole 16-byte chunks, and bit rows of 65 .. 127 bytes (768 bits: quant_bits = 1 at 768 dimensions, hamming over integer[24]) to 128 --
 * one cache line per row instead of a line and a piece of the next (usearch_storage.cpp:29-31 sizes the ON-TAPE vector; this is HBM only) */
LANTERN_GPU_EXPORT size_t lantern_gpu_row_bytes(usearch_index_t, usearch_error_t *);
LANTERN_GPU_EXPORT void lantern_gpu_memory_usage(usearch_index_t, size_t *row_bytes, size_t *other_bytes, usearch_error_t *);
LANTERN_GPU_EXPORT void lantern_gpu_import_graph(usearch_index_t, size_t size, const void *vectors,
                                                 const uint64_t *labels, const uint8_t *levels, const uint32_t *nbr0,
                                                 const uint32_t *upper_off, const uint32_t *upper_nbr,
                                                 uint32_t entry_slot, int32_t max_level, usearch_error_t *);
/* cumulative counters since init: distance evaluations and expanded nodes of searches, and the
 * same for the insert path */
typedef struct lantern_gpu_counters
{
    uint64_t search_dist_evals, search_expansions, search_queries;
    uint64_t add_dist_evals, add_expansions, add_vectors, add_batches;
    /* breakdown of add_dist_evals: walk / neighbour selection of the new node / reverse-link re-pruning.  The walk counts are
     * the reference algorithm's own (asserted equal to the oracle's).  The other two count what the DEVICE evaluates: a
     * candidate is tested against all kept rows at once (k_connect) and a full list's pair table is filled up front
     * (k_revlink_pairs), where sequential usearch stops at the first blocker -- they are >= the CPU path's counts and are the
     * right numerators for the device's rooflines, not for a CPU comparison. */
    uint64_t add_walk_evals, add_select_evals, add_revlink_evals, add_reprunes;
    /* search launches that took the one-wave walk (csrc/walk_solo.hpp): what a lone usearch_search_ef call (scan.c:220-228) runs
     * on an index it applies to -- tests assert the kernel that was meant is the kernel that ran */
    uint64_t search_solo_launches;
} lantern_gpu_counters;
LANTERN_GPU_EXPORT lantern_gpu_counters lantern_gpu_counters_get(usearch_index_t, usearch_error_t *);

/* Build profile: with profiling on, every insertion batch is bracketed by HIP events on the index stream (phase by phase);
 * the sums are milliseconds of device time per phase since init.  walk = k_insert (+ the batch layout), connect =
 * k_connect, group = the request grouping (+ exchange 1 of a sharded build), revlink = append + re-prune kernels,
 * exchange = exchange 2 of a sharded build (0 on one GPU). */
typedef struct lantern_gpu_build_profile
{
    double   walk_ms, connect_ms, group_ms, revlink_ms, exchange_ms;
    uint64_t batches;
} lantern_gpu_build_profile;
LANTERN_GPU_EXPORT void lantern_gpu_set_profiling(usearch_index_t, int on, usearch_error_t *);
LANTERN_GPU_EXPORT lantern_gpu_build_profile lantern_gpu_build_profile_get(usearch_index_t, usearch_error_t *);

/* Diagnostics of the walk kernel: while on, searches run an instrumented instantiation (f32 l2sq / cos rows of >= 128 or
 * 32..63 chunks only) whose thread 0 sums shader-clock cycles per phase of every hop.  out8 (may be NULL) receives and
 * clears the sums: visited filter + compaction | wait at the hop's first barrier | distances | merge | pop | arrival of the
 * neighbour list | upper-level descent | whole query.  When the walk splits a hop's bookkeeping over two waves (register
 * list, two or more waves per query: walk.hpp search_level_reg) thread 0 never merges: slot 3 ("merge") is then its pop
 * DECISION and slot 4 ("pop") the list wave's whole section (merge + pop + hand-off), which runs beside slots 3, 5 and 0. */
LANTERN_GPU_EXPORT void lantern_gpu_search_phase_profile(usearch_index_t, int on, unsigned long long *out8, usearch_error_t *);
/* Diagnostics: how many DISTINCT rows the searches between (on = 1) and the read (on = 0, rows != NULL) evaluated, over all their
 * queries: unique rows x row bytes is the cold-miss lower bound of a launch's DRAM traffic (every such row has to come from
 * HBM at least once), to put beside the algorithmic bytes (one row per evaluation) and the counters' fabric-side bytes.  Runs
 * the instrumented instantiation of the walk kernel (f32 l2sq / cos; same walk, same D and E; slower). */
LANTERN_GPU_EXPORT void lantern_gpu_search_unique_rows(usearch_index_t, int on, uint64_t *rows, usearch_error_t *);
/* the same for the latency-bound walk (lantern_amd/csrc/walk_spec.hpp): out32[8 * wave + i], waves 0..3 = visit | list | cache
 * fill | a row wave; i = 0 decision, 1 neighbour list, 2 issuing the row loads, 3 the wave's role section, 4 loads landing +
 * distances, 5 wait at the hop's barrier, 6 hops, 7 where lists came from (wave 0: staging area, wave 3: cache, wave 2: HBM) */
LANTERN_GPU_EXPORT void lantern_gpu_spec_profile(usearch_index_t, int on, unsigned long long *out32, usearch_error_t *);

/* order-independent-of-builder fingerprint of the graph (levels, labels, both adjacency arrays, entry point):
 * equal on two indexes iff they hold the same graph; used to check that replicas agree without moving them */
LANTERN_GPU_EXPORT uint64_t lantern_gpu_graph_checksum(usearch_index_t, usearch_error_t *);

/* ------------------------------------------------------------------------------------------ */
/* Work-sharded index build over the GPUs of one node (the reference's     */
/* own parallel build is a thread pool on one shared index, server.rs:328-359).  One process or  */
/* thread per GPU, each with its own usearch_index_t replica.  lantern_gpu_add_sharded is a       */
/* COLLECTIVE: rank r passes shard r of the rows (global slot order = rank order); the shards are */
/* all-gathered into every replica's HBM, then every insertion batch is split across the ranks:   */
/* each walks and connects its share of the new nodes, the ranks all-gather the resulting top-M   */
/* neighbour lists, each applies the reverse links of the nodes it owns and the re-written        */
/* adjacency rows are all-gathered.  All replicas end up bit-identical to the graph one GPU       */
/* builds from the same rows with the same batch parameters.                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct lantern_gpu_comm lantern_gpu_comm_t;
#define LANTERN_GPU_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */
/* RCCL transport (xGMI, data stays in HBM): rank 0 draws the id and ships it to its peers by any means,
 * then every rank calls init_rccl with the HIP device it will build on current (hipSetDevice). */
LANTERN_GPU_EXPORT void lantern_gpu_comm_unique_id(char *id128, usearch_error_t *);
LANTERN_GPU_EXPORT lantern_gpu_comm_t *lantern_gpu_comm_init_rccl(int rank, int world, const char *id128, usearch_error_t *);
/* Host transport: the caller supplies an in-place all-gather over a HOST buffer (MPI, gloo, sockets):
 * on entry bytes [offsets[rank], +counts[rank]) are valid, on return (0 = success) every rank's segment is. */
typedef int (*lantern_gpu_allgatherv_fn)(void *ctx, void *host_buf, const size_t *offsets, const size_t *counts, int world,
                                         int rank);
LANTERN_GPU_EXPORT lantern_gpu_comm_t *lantern_gpu_comm_init_host(int rank, int world, lantern_gpu_allgatherv_fn, void *ctx,
                                                                  usearch_error_t *);
/* In-process world: `world` communicators for `world` threads of this process (one thread per GPU of a node,
 * or several ranks on one GPU in tests); out[world]. */
LANTERN_GPU_EXPORT void lantern_gpu_comm_init_local(int world, lantern_gpu_comm_t **out, usearch_error_t *);
LANTERN_GPU_EXPORT void lantern_gpu_comm_free(lantern_gpu_comm_t *);
LANTERN_GPU_EXPORT int  lantern_gpu_comm_rank(lantern_gpu_comm_t *);
LANTERN_GPU_EXPORT int  lantern_gpu_comm_world(lantern_gpu_comm_t *);
/* a collective that does not complete within `seconds` (default 180) fails instead of hanging */
LANTERN_GPU_EXPORT void lantern_gpu_comm_set_timeout(lantern_gpu_comm_t *, double seconds);
/* bytes this rank received and number of exchanges so far */
LANTERN_GPU_EXPORT void lantern_gpu_comm_stats(lantern_gpu_comm_t *, uint64_t *bytes_received, uint64_t *collectives);
/* the exchange primitive itself (in-place all-gather with per-rank sizes), host and device buffers */
LANTERN_GPU_EXPORT void lantern_gpu_comm_allgatherv_host(lantern_gpu_comm_t *, void *host_buf, const size_t *offsets,
                                                         const size_t *counts, usearch_error_t *);
LANTERN_GPU_EXPORT void lantern_gpu_comm_allgatherv_device(lantern_gpu_comm_t *, void *device_buf, const size_t *offsets,
                                                           const size_t *counts, void *stream, usearch_error_t *);
/* the balanced split used for batches and shards: rank r of `world` takes [n*r/world, n*(r+1)/world) */
LANTERN_GPU_EXPORT void lantern_gpu_shard_range(size_t n, int world, int rank, size_t *begin, size_t *end);
/* the collective insert described above; n_shard may be 0 */
LANTERN_GPU_EXPORT void lantern_gpu_add_sharded(usearch_index_t, lantern_gpu_comm_t *, const usearch_label_t *labels_shard,
                                                const void *vectors_shard, size_t n_shard, usearch_scalar_kind_t,
                                                usearch_error_t *);

/* Row-SHARDED build (SURVEY.md section 8e as written: "vectors sharded by row; each GPU generates candidates within its shard
 * for a tile of the rows; all-gather of the candidate lists; merge"): a COLLECTIVE on an EMPTY index (f32 / f16 / i8 / b1 rows,
 * not pq).  Rank r passes shard r of the rows.  Every rank keeps a graph over ITS rows only, grown in lock step with the global
 * one.  A batch of the global build (the usual plan, its members drawn from all shards in proportion): every rank inserts its
 * share into its own graph; the batch's rows are all-gathered (the only time a row crosses the fabric); every rank searches its
 * own graph for each row of the batch (k_search; K = max(2M + 1, 2 expansion_add / world) answers), the per-rank lists are
 * all-gathered in HBM (12 bytes per candidate) and merged on the device by (distance, slot) into the input of the neighbour
 * selection; selection and reverse links then run on every rank as in a one-GPU batch (the upper levels, ~1 / M of the rows, are
 * walked in the global graph as usual).  Every rank ends with the same graph and the same rows (lantern_gpu_graph_checksum).
 * The slots follow the batches, not the ranks: labels identify rows.  The graph is NOT the one usearch_add builds edge for edge
 * (a row's candidates are the union of `world` approximate searches instead of one): it equals, edge for edge, the CPU
 * restatement of THIS collective (oracle.row_sharded_build; tests/test_gpu_sharded_build.py), and is compared with the
 * one-GPU build by recall (profiles/r04_row_sharded_build.md: on par or better), where lantern_gpu_add_sharded above equals
 * the one-GPU build itself.  Cost: every rank searches ALL rows and links ALL rows; only the shard-graph insertions fall with
 * the world size (measured bound: 1.3x at 4-8 ranks) -- lantern_gpu_add_sharded is the build that scales.
 * LANTERN_GPU_ROW_SHARD_K / LANTERN_GPU_ROW_SHARD_EF override the candidates per shard / the expansion they are found with. */
LANTERN_GPU_EXPORT void lantern_gpu_add_row_sharded(usearch_index_t, lantern_gpu_comm_t *, const usearch_label_t *labels_shard,
                                                    const void *vectors_shard, size_t n_shard, usearch_scalar_kind_t,
                                                    usearch_error_t *);
/* the batches lantern_gpu_add_row_sharded runs for these shard sizes and where their rows come from (host arithmetic only, no
 * device): first[t] / count[t] = the batch's slots, share[t * world + r] = how many of its rows are shard r's (rank 0's take the
 * first slots of the batch).  Returns the number of batches; at most `capacity` are written. */
LANTERN_GPU_EXPORT size_t lantern_gpu_row_shard_plan(const uint64_t *shard_sizes, int world, uint64_t seed, uint32_t connectivity,
                                                     size_t max_batch, size_t min_ratio, size_t *first, size_t *count, size_t *share,
                                                     size_t capacity);

/* Row-PARTITIONED search (SURVEY.md section 8e as written: "vectors sharded by row; each GPU produces its best candidates
 * within its shard; all-gather; merge"), for indexes whose vectors do not fit one GPU's HBM: a COLLECTIVE over `comm`.  Every
 * rank holds its OWN index over a disjoint share of the rows -- built independently with usearch_add / lantern_gpu_add_many, no
 * exchange at build time, vector memory scales with the number of GPUs -- and passes the SAME queries.  Each searches its
 * share (usearch_search_ef semantics, ef per share), the per-rank top-k (label, distance) lists are all-gathered in place in
 * HBM (RCCL: xGMI; nq x k x 12 bytes per rank) and merged on the device by (distance, label).  Every rank returns the same
 * global top-k.  Labels must be unique across the shares.  (The work-sharded build above keeps ONE graph, bit-identical to
 * the single-GPU build, but needs every row in every GPU's HBM; this form trades that identity for capacity.) */
LANTERN_GPU_EXPORT void lantern_gpu_search_partitioned(usearch_index_t, lantern_gpu_comm_t *, const void *queries, size_t nq,
                                                       usearch_scalar_kind_t, size_t k, size_t ef, usearch_label_t *labels,
                                                       float *distances, uint32_t *counts, usearch_error_t *);

/* ------------------------------------------------------------------------------------------ */
/* amgettuple paging shim: ldb_ambeginscan / ldb_amgettuple / ldb_amendscan (scan.c:24-338)     */
/* ------------------------------------------------------------------------------------------ */
typedef struct lantern_scan lantern_scan_t;
/* init_k = GUC lantern_hnsw.init_k (options.h:44, default 10); ef = GUC lantern_hnsw.ef or 0 */
LANTERN_GPU_EXPORT lantern_scan_t *lantern_scan_begin(usearch_index_t, int init_k, int ef, usearch_error_t *);
/* the same scan driven through a scan-service connection (declared below) instead of a local index: what a PostgreSQL
 * backend runs when the mirror lives in the service process.  query_bytes = the size of one query vector. */
struct lantern_scan_client;
LANTERN_GPU_EXPORT lantern_scan_t *lantern_scan_begin_client(struct lantern_scan_client *, size_t query_bytes, int init_k, int ef,
                                                             usearch_error_t *);
/* ldb_amrescan: (re)arm the scan with an ORDER BY key; the vector is copied */
LANTERN_GPU_EXPORT void lantern_scan_rescan(lantern_scan_t *, const void *query, usearch_scalar_kind_t,
                                            usearch_error_t *);
/* ldb_amgettuple: true and *label set while tuples remain.  Skips label 0 (deleted, scan.c:296-300),
 * doubles k through the streaming continuation (scan.c:240-292), stops at 1000 rows (:249-252). */
LANTERN_GPU_EXPORT bool lantern_scan_gettuple(lantern_scan_t *, usearch_label_t *label, usearch_error_t *);
/* The k of every usearch_search_ef the scan has issued since its last rescan, in order -- the sequence the reference logs as
 * "LANTERN querying index for %d elements" (scan.c:219, :272; pinned by test/expected/hnsw_select.out:76-140: 10 | 4, 8, 8).
 * Writes min(n, cap) values to ks (may be NULL), returns n. */
LANTERN_GPU_EXPORT size_t lantern_scan_trace(lantern_scan_t *, int *ks, size_t cap);
LANTERN_GPU_EXPORT void lantern_scan_end(lantern_scan_t *);

/* ------------------------------------------------------------------------------------------ */
/* Lifecycle of the HBM mirrors of page-resident indexes.  The reference      */
/* attaches anew per scan and per insert (scan.c:99-110, insert.c:142-151: free for a lazy view);  */
/* a device mirror is a walk of the graph through the retriever, so a process keeps its mirrors in  */
/* this cache, keyed by (relation = relfilenode, version = LSN of the header page or num_vectors).   */
/* ------------------------------------------------------------------------------------------ */
typedef struct lantern_mirror lantern_mirror_t;
/* The resident mirror of (relation, version), or a fresh one: usearch_init(opts, pq_codebook) + usearch_view_mem_lazy(header).
 * A newer version replaces the relation's mirror (the stale one is freed when its last holder releases it).  Returns NULL
 * with *err == NULL when the header declares fewer than min_vectors nodes: the caller stays on the path it has. */
LANTERN_GPU_EXPORT lantern_mirror_t *lantern_mirror_acquire(uint64_t relation, uint64_t version, usearch_init_options_t *opts,
                                                            float *pq_codebook, char *header136, size_t min_vectors, usearch_error_t *);
LANTERN_GPU_EXPORT usearch_index_t lantern_mirror_index(lantern_mirror_t *);
LANTERN_GPU_EXPORT uint64_t        lantern_mirror_version(lantern_mirror_t *);
/* The mirror's retriever callbacks (init_options.retriever / retriever_mut / retriever_ctx -- the reference's per-scan,
 * per-insert RetrieverCtx: scan.c:34,132, insert.c:130,247) are kept PER HOLDER, a holder being a host thread (a PostgreSQL
 * backend is one; a threaded service runs one holder per thread): bound by that thread's acquire or rebind, dropped by that
 * thread's release -- no ctx pointer outlives the acquire / release pair that brought it, and one holder's release never takes
 * away another's.  usearch_add_external uses the calling thread's and fails with a message when it has none (a thread that
 * holds several handles of one mirror and released one of them binds its own again first). */
LANTERN_GPU_EXPORT void lantern_mirror_rebind(lantern_mirror_t *, const usearch_init_options_t *opts);
/* the holder applied a change itself (usearch_add_external + usearch_update_header): re-stamp instead of rebuilding */
LANTERN_GPU_EXPORT void lantern_mirror_advance(lantern_mirror_t *, uint64_t new_version);
LANTERN_GPU_EXPORT void lantern_mirror_release(lantern_mirror_t *);
/* DROP INDEX / REINDEX / VACUUM */
LANTERN_GPU_EXPORT void lantern_mirror_invalidate(uint64_t relation);
/* idle mirrors kept resident (default 8; least recently used goes first) */
LANTERN_GPU_EXPORT void lantern_mirror_set_capacity(size_t max_resident);
LANTERN_GPU_EXPORT void lantern_mirror_stats(uint64_t *hits, uint64_t *misses, uint64_t *rebuilds, uint64_t *resident);

/* ------------------------------------------------------------------------------------------ */
/* External indexing server (boundary B3): drop-in for `lantern-cli start-indexing-server`        */
/* (lantern_cli/src/external_index/server.rs:176-435,526-584); PostgreSQL side untouched           */
/* (lantern_hnsw/src/hnsw/external_index_socket.c:322-536).                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct lantern_index_server lantern_index_server_t;
/* Bind host:port (port 0 = ephemeral; default of the reference is 0.0.0.0:8998, cli.rs:126-151), start the
 * accept thread.  status_port: HTTP status endpoint {"status":0..3} (server.rs:586-597), -1 = none.
 * One connection is served at a time, as in the reference. */
LANTERN_GPU_EXPORT lantern_index_server_t *lantern_index_server_start(const char *host, int port, int status_port,
                                                                      const char *tmp_dir, usearch_error_t *);
/* The same over TLS (`start-indexing-server --cert C --key K`: cli.rs:146, server.rs:437-470,548): PEM files; every accepted
 * connection starts with the handshake (the PostgreSQL side: external_index_socket_ssl.c:6-62, TLS >= 1.2, no certificate
 * verification).  libssl is bound at run time; NULL, NULL = the plain server.  The router redirect of
 * external_index_socket.c:411-447 is the ROUTER's half of the protocol (server type 0x2), not this server's. */
LANTERN_GPU_EXPORT lantern_index_server_t *lantern_index_server_start_tls(const char *host, int port, int status_port, const char *tmp_dir,
                                                                          const char *cert_pem, const char *key_pem, usearch_error_t *);
LANTERN_GPU_EXPORT int      lantern_index_server_port(lantern_index_server_t *);
LANTERN_GPU_EXPORT int      lantern_index_server_status_port(lantern_index_server_t *);
/* 0 idle, 1 in progress, 2 failed, 3 succeeded (server.rs:44-49) */
LANTERN_GPU_EXPORT int      lantern_index_server_status(lantern_index_server_t *);
LANTERN_GPU_EXPORT uint64_t lantern_index_server_served(lantern_index_server_t *);
LANTERN_GPU_EXPORT void     lantern_index_server_stop(lantern_index_server_t *);

/* ------------------------------------------------------------------------------------------ */
/* Scan-side service: ONE HBM-resident index serves the k-NN queries of    */
/* many PostgreSQL backends, coalesced into batched launches.  A backend's ldb_amgettuple          */
/* (scan.c:167-338) calls lantern_scan_client_search where it calls usearch_search_ef today; the   */
/* server waits at most `max_wait_us` after the first queued query for company (at most            */
/* `max_batch` queries per launch; less if every connected backend is already accounted for or    */
/* a dispatcher's share of them is), runs one lantern_gpu_search_batch per distinct (k, ef) and    */
/* routes the answers back.  No thread per connection: a few epoll I/O threads                    */
/* (LANTERN_SCAN_IO_THREADS) carry all sockets.  Wire format: lantern_amd/csrc/scan_server.cpp.    */
/* ------------------------------------------------------------------------------------------ */
typedef struct lantern_scan_server lantern_scan_server_t;
typedef struct lantern_scan_client lantern_scan_client_t;
/* the batch search a server runs: nq queries of vec_bytes each -> labels/dists [nq][k], counts [nq]; 0 = ok,
 * otherwise *err points at a message that outlives the call */
typedef int (*lantern_batch_search_fn)(void *ctx, const void *queries, size_t nq, size_t vec_bytes, size_t k, size_t ef,
                                       uint64_t *labels, float *distances, uint32_t *counts, const char **err);
/* serve `index` (built, loaded or mirrored; it must outlive the server) on host:port (port 0 = ephemeral).  Two dispatcher
 * threads take turns -- one forms the next batch while the other's is being searched, each on its own lane of the index
 * (lantern_gpu_search_batch_lane) -- so consecutive launches overlap on the device.  A caller-supplied back end
 * (lantern_scan_server_start_fn) is entered by one thread only, unless LANTERN_SCAN_LANES=2 says it may be entered by two. */
LANTERN_GPU_EXPORT lantern_scan_server_t *lantern_scan_server_start(usearch_index_t index, const char *host, int port,
                                                                    size_t max_batch, unsigned max_wait_us, usearch_error_t *);
/* the same front end over any batch search function (tests; a host that shards queries over several GPUs) */
LANTERN_GPU_EXPORT lantern_scan_server_t *lantern_scan_server_start_fn(lantern_batch_search_fn, void *ctx, size_t vec_bytes,
                                                                       const char *host, int port, size_t max_batch,
                                                                       unsigned max_wait_us, usearch_error_t *);
LANTERN_GPU_EXPORT int  lantern_scan_server_port(lantern_scan_server_t *);
/* queries received, batches formed, search launches (one per distinct (k, ef) of a batch), largest batch so far */
LANTERN_GPU_EXPORT void lantern_scan_server_stats(lantern_scan_server_t *, uint64_t *requests, uint64_t *batches,
                                                  uint64_t *launches, uint64_t *largest_batch);
/* batches formed so far by size: bins[b] counts batches of 2^b .. 2^(b+1) - 1 requests (b < 16); returns the bins written */
/* mean microseconds of a request on the server by leg -- out4[0] read -> its batch closes, [1] batch closed -> its answer is known,
 * [2] answer known -> written to the socket -- and out4[3] the number of requests the means are over (cumulative since start) */
LANTERN_GPU_EXPORT void lantern_scan_server_timing(lantern_scan_server_t *, double *out4);
LANTERN_GPU_EXPORT size_t lantern_scan_server_batch_histogram(lantern_scan_server_t *, uint64_t *bins, size_t nbins);
LANTERN_GPU_EXPORT void lantern_scan_server_stop(lantern_scan_server_t *);
/* client: one connection per backend, one query at a time; returns the number of results (<= k), ascending */
LANTERN_GPU_EXPORT lantern_scan_client_t *lantern_scan_client_connect(const char *host, int port, usearch_error_t *);
LANTERN_GPU_EXPORT size_t lantern_scan_client_search(lantern_scan_client_t *, const void *query, size_t query_bytes, size_t k,
                                                     size_t ef, usearch_label_t *labels, float *distances, usearch_error_t *);
/* the streaming continuation (scan.c:273-281) through the service: the NEXT k rows of the scan this connection began with
 * its last lantern_scan_client_search; the state is the connection's, so concurrent backends never disturb each other */
LANTERN_GPU_EXPORT size_t lantern_scan_client_search_next(lantern_scan_client_t *, const void *query, size_t query_bytes, size_t k,
                                                          size_t ef, usearch_label_t *labels, float *distances, usearch_error_t *);
LANTERN_GPU_EXPORT void   lantern_scan_client_close(lantern_scan_client_t *);

/* ------------------------------------------------------------------------------------------ */
/* SQL-callable distance functions' semantics (hnsw.c:296-405): dimension checks + messages     */
/* ------------------------------------------------------------------------------------------ */
/* l2sq_dist(real[], real[]) -> float4; error text hnsw.c:301-303 */
LANTERN_GPU_EXPORT float lantern_l2sq_dist(const float *a, int a_dim, const float *b, int b_dim, usearch_error_t *);
LANTERN_GPU_EXPORT float lantern_cos_dist(const float *a, int a_dim, const float *b, int b_dim, usearch_error_t *);
/* hamming_dist(integer[], integer[]) -> int32 (hnsw.c:308-319,370-376): bits = dim*32 */
LANTERN_GPU_EXPORT int32_t lantern_hamming_dist(const int32_t *a, int a_dim, const int32_t *b, int b_dim,
                                                usearch_error_t *);

#ifdef __cplusplus
}
#endif
#endif /* LANTERN_GPU_H */
