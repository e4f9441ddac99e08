/*
 * lantern_oracle.h -- CPU oracle for the Lantern HNSW distance-evaluation hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (lantern_amd/csrc, the
 * C-ABI in include/) never links, imports or calls anything in this directory.
 *
 * What it restates
 * ----------------
 * Lantern (reference at /root/reference, read-only) delegates every distance and every
 * graph hop to usearch (un-vendored fork github.com/Ngalstyan4/usearch, branch pg-rebase,
 * Rust pin rev aa4f91d21230fd611b6c7741fa06be8c20acc9a9 -- lantern_cli/Cargo.toml:40,
 * .gitmodules:1-4; lantern_hnsw/third_party/usearch/ is an EMPTY directory).  The
 * algorithm below is therefore the published usearch 2.x algorithm (index_gt::add /
 * search, metric_*_gt), anchored on Lantern's own call sites:
 *   usearch_distance   lantern_hnsw/src/hnsw.c:296-345
 *   usearch_search_ef  lantern_hnsw/src/hnsw/scan.c:220-228, :273-281
 *   usearch_add        lantern_hnsw/src/hnsw/build.c:128
 *   level draw         lantern_hnsw/src/hnsw/insert.c:32-46
 *   level capacities   lantern_hnsw/src/hnsw/validate_index.c:140-151 (2M at level 0, M above)
 *
 * Parity pinning: the reference's own tests pin this path only on small exact cases
 * (expected/hnsw_dist_func.out, hnsw_operators.out, hnsw_correct.out,
 * hnsw_vector.out, hnsw_insert.out, hnsw_delete.out, hnsw_cost_estimate.out).  All of
 * those are checked in tests/test_oracle_golden.py.  At 100k..10M points the reference
 * pins nothing that is available offline (datasets are downloaded at test time), so for
 * those sizes: PARITY UNPINNED BY THE REFERENCE -- this oracle is the arbiter.
 *
 * Deviations from usearch that are deliberate and documented:
 *   - candidate ordering is the TOTAL order (distance, slot) instead of distance only,
 *     so results are independent of heap/sorted-buffer tie behaviour;
 *   - levels come from a stateless hash RNG keyed by (seed, slot), not from
 *     std::default_random_engine, so builds are reproducible;
 *   - LO_SUM_WAVE64 is a second summation order that models the gfx950 kernel's
 *     reduction tree bit-for-bit (used to assert exact traversal parity).
 */
#ifndef LANTERN_ORACLE_H
#define LANTERN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* metric / scalar enum values follow usearch as seen from Lantern:
 * lantern_cli/src/external_index/cli.rs:56-69, server.rs:94-101 */
/* LO_METRIC_COS_B1: the cosine of the {0, 1} vectors that bit rows stand for (quant_bits = 1 on a cosine index; rows are bits like
 * hamming's): 1 - |a & b| / (sqrt |a| sqrt |b|), the f32 metric's zero-norm rules.  PARITY UNPINNED BY THE REFERENCE. */
enum { LO_METRIC_COS = 1, LO_METRIC_L2SQ = 3, LO_METRIC_HAMMING = 8, LO_METRIC_COS_B1 = 9 };
#define LO_METRIC_IS_BITS(m) ((m) == LO_METRIC_HAMMING || (m) == LO_METRIC_COS_B1)

/* summation order for f32 metrics */
enum {
    LO_SUM_SEQ = 0,    /* usearch metric_l2sq_gt / metric_cos_gt: one running f32 sum, i = 0..d-1 */
    LO_SUM_WAVE64 = 1, /* device order: G-lane fmaf chains + butterfly with offsets 1..G/2 */
    LO_SUM_FAST = 2,   /* same maths as SEQ, compiled with the reference's -fassociative-math flags
                          (lantern_hnsw/CMakeLists.txt:122-136): what the CPU baseline times */
    LO_SUM_WAVE64_F16 = 3, /* device order for f16 STORAGE (quant_bits=16, options.c:137-158): 8 scalars per
                          16-byte chunk.  The oracle always computes on f32 arrays; for an f16 index the caller
                          passes values already rounded to f16 (usearch casts at add/search and its
                          metric_*_gt<f16_t, f32> converts each element back to f32 before the arithmetic) */
    LO_SUM_I8 = 4      /* i8 STORAGE (quant_bits=8): the caller passes f32 arrays that already hold the quantised
                          integers trunc(clamp(x*100, -100, 100)) (lantern_hnsw/test/sql/hnsw_sq.sql:33-34); the
                          arithmetic is usearch's l2sq_i8_t / cos_i8_t: int32 accumulation, integer-exact, so there
                          is no summation order to model.  PARITY UNPINNED: the reference pins no i8 result offline */
};

#define LO_EMPTY_SLOT 0xFFFFFFFFu

typedef struct lo_index lo_index;

/* -------- pairwise metrics (usearch_distance; hnsw.c:296-345) ---------------------------- */
/* dims = number of f32 scalars (cos / l2sq) or number of BITS (hamming; hnsw.c:317-319). */
float lo_distance(const void *a, const void *b, size_t dims, int metric, int sum_mode);
/* number of cooperating lanes the device uses for a row of `dims` f32 scalars */
int lo_wave_group_lanes(size_t dims);

/* -------- level draw (insert.c:32-46; usearch choose_random_level_) ---------------------- */
int lo_level_for(uint64_t seed, uint64_t slot, uint32_t connectivity);

/* -------- exact k-NN (ground truth for recall; index_autotune/mod.rs:196-203) ------------ */
/* rows: n x dims f32 (or n x dims/32 u32 words for hamming).  out_ids/out_dists: nq x k,
 * ascending by (distance, id); short rows are padded with LO_EMPTY_SLOT / INFINITY. */
void lo_bruteforce(const void *rows, size_t n, size_t dims, int metric, int sum_mode,
                   const void *queries, size_t nq, size_t k, uint32_t *out_ids, float *out_dists,
                   int nthreads);

/* -------- HNSW index ---------------------------------------------------------------------- */
lo_index *lo_create(int metric, size_t dims, uint32_t connectivity, uint32_t expansion_add,
                    uint32_t expansion_search, uint64_t seed, int sum_mode);
void lo_free(lo_index *);
int lo_reserve(lo_index *, size_t capacity);
size_t lo_size(const lo_index *);
size_t lo_capacity(const lo_index *);
int lo_max_level(const lo_index *);
uint32_t lo_entry_slot(const lo_index *);

/* usearch_add (build.c:128): one sequential insertion; level from lo_level_for(seed, slot). */
int lo_add(lo_index *, uint64_t label, const void *vec);
/* usearch_add_external-style insertion with a caller-chosen level (insert.c:209). */
int lo_add_with_level(lo_index *, uint64_t label, const void *vec, int level);
/* Batch-synchronous insertion: every vector of the batch searches the graph as it was when
 * the batch started; links are then applied in slot order.  n == 1 is exactly lo_add.  This
 * is the semantics of the device builder (and approximates usearch's concurrent add_raw,
 * lantern_cli/src/external_index/server.rs:333-356). */
int lo_add_batch(lo_index *, const uint64_t *labels, const void *vecs, size_t n);
/* the row-sharded build's batch (lantern_amd/csrc/index.cpp add_row_sharded_locked): as lo_add_batch, but node i's level-0
 * candidates are cand_slot / cand_d [i * stride .. + cand_n[i]) instead of a walk of this graph; n == 1 may raise the top level */
int lo_add_batch_cand(lo_index *, const uint64_t *labels, const void *vecs, size_t n, const uint32_t *cand_slot, const float *cand_d,
                      const uint32_t *cand_n, size_t stride);
/* threads for the two phases of lo_add_batch (walks of a batch are independent; so are its (node, level) groups of
 * reverse links): the graph does not depend on the number -- tests/test_oracle_golden.py builds with 1 and with 4 */
void lo_set_build_threads(lo_index *, int nthreads);
/* ADC view of a pq = true index (lantern_amd/csrc/search_adc_kernel.hip restated): the index's vectors are the DECODED rows,
 * `codes` their num_subvectors code bytes, `codebook` [num_centroids][dims]; searches from now on evaluate a row as the sum of
 * per-subvector table entries in the device's order.  PARITY UNPINNED BY THE REFERENCE (the fork's PQ metric is not in the tree). */
int lo_set_pq_view(lo_index *, uint32_t num_subvectors, uint32_t num_centroids, const float *codebook, const uint8_t *codes);
/* LO_SUM_WAVE64 over f32 runs eight lanes of the tree per AVX2 instruction where the host has them; 0 selects the scalar
 * restatement (identical bits: tests compare the two) */
void lo_set_wave_simd(int on);
/* The device builder's batch plan: which prefix of the pending vectors forms the next batch.
 * Returned value >= 1. */
size_t lo_plan_batch(size_t current_size, int max_level, const int *pending_levels, size_t pending,
                     size_t max_batch, size_t min_ratio);

/* usearch_search_ef (scan.c:220-228): ef == 0 -> index default.  Returns count <= k,
 * ascending by (distance, slot).  `skip` = number of leading results to drop (the
 * streaming continuation of scan.c:273-281 asks for the NEXT k after the ones it has). */
size_t lo_search(lo_index *, const void *query, size_t k, size_t ef, size_t skip, uint64_t *out_labels,
                 float *out_dists, uint32_t *out_slots);
/* many queries, `nthreads` host threads, one query per thread at a time */
void lo_search_batch(lo_index *, const void *queries, size_t nq, size_t k, size_t ef, uint64_t *out_labels,
                     float *out_dists, uint32_t *out_slots, uint64_t *out_D, uint64_t *out_E, int nthreads);

/* counters of the last lo_search on this index (single-thread use): distance evaluations
 * and expanded (popped) base-layer nodes -- SURVEY.md section 8(d) */
uint64_t lo_last_distance_evals(const lo_index *);
uint64_t lo_last_expansions(const lo_index *);

/* -------- flat graph exchange with the device library ------------------------------------- */
/* levels[n] (u8); nbr0[n][2M] u32 LO_EMPTY_SLOT-terminated; upper_off[n] u32 (index of the
 * node's first upper block, LO_EMPTY_SLOT if level 0); upper_nbr[n_upper_blocks][M]. */
size_t lo_upper_blocks(const lo_index *);
void lo_export_graph(const lo_index *, uint8_t *levels, uint32_t *nbr0, uint32_t *upper_off, uint32_t *upper_nbr,
                     uint64_t *labels);
lo_index *lo_import_graph(int metric, size_t dims, uint32_t connectivity, uint32_t expansion_add,
                          uint32_t expansion_search, uint64_t seed, int sum_mode, size_t n, const void *vectors,
                          const uint64_t *labels, const uint8_t *levels, const uint32_t *nbr0,
                          const uint32_t *upper_off, const uint32_t *upper_nbr, uint32_t entry_slot, int max_level,
                          int borrow_vectors);
const void *lo_vector(const lo_index *, uint32_t slot);

/* planner bound on visited tuples (lantern_hnsw/src/hnsw.c:89-132), used by a golden test */
uint64_t lo_estimate_visited_tuples(double num_tuples, uint32_t M, uint32_t ef);

#ifdef __cplusplus
}
#endif
#endif
