// index.cpp -- host side of the C ABI: index lifecycle, buffered/batched inserts, search launches.
//
// Mirrors the usearch C API as Lantern calls it (prototypes and reference
// call sites are listed in include/lantern_gpu.h).  The graph and the vector block live in HBM for
// the whole life of the index; the host keeps only labels/levels (for serialisation) and the
// buffer of not-yet-inserted vectors.
#include "index.hpp"
#include "abi_guard.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <time.h>

#include "comm.hpp"
#include "host_util.hpp"

// the exact k-NN building block (fp32-MFMA contraction + exact re-rank), defined with the C ABI below
static bool exact_knn_device(int mcode, uint32_t chunks, const uint4 *d_base, size_t nb, const uint4 *d_q, size_t nq, size_t k,
                             uint32_t *d_slots, float *d_dists, hipStream_t st);

// The HIP runtime multiplexes a process's streams onto FOUR hardware queues by default (GPU_MAX_HW_QUEUES); streams that share a queue
// run their kernels one after the other.  This library keeps up to eight search launches in flight on streams of their own (the lanes of
// lantern_gpu_search_batch_lane: the scan-side service's dispatchers) beside the index's stream and whatever the caller made: with
// four queues two lanes end up behind one another -- measured in round 5 at 1M x 768, 256 backends: a batch's answers 100 us later,
// 443 k instead of 551 - 589 k scans/s, and which lanes collide depends on how many streams the process happened to create first.
// The flag is read when the runtime initialises, so it belongs to whoever STARTS the process: lantern-scan-server's main() sets it,
// bench.py and the tests set it before the library is loaded, INTEGRATION.md section 7 tells a host that embeds the service to.
// The library itself never touches the environment of the process it is loaded into (a PostgreSQL backend, a Python process with
// other HIP users in it); lantern_scan_server_start warns on stderr when it runs more lanes than the setting gives queues.

namespace lgpu {

// LANTERN_GPU_LDS_LIST=1: walks keep their candidate list in LDS even when it fits wave 0's registers (walk.hpp search_level vs
// search_level_reg; identical results -- the switch exists for A/B timing and for the parity test that runs both)
static int lds_list_env()
{
    const char *e = std::getenv("LANTERN_GPU_LDS_LIST");
    return e && std::atoi(e) != 0;
}
static const char *kNoDevice = "lantern_gpu: no HIP device available (this library has no CPU fallback)";

const char *set_err(Index *ix, const std::string &msg)
{
    ix->err = msg;
    // HIP keeps the last error of the thread until somebody reads it, and every launch wrapper ends in hipGetLastError(): a
    // failed allocation (an absurd usearch_reserve) would otherwise fail the NEXT kernel launch of an index that is perfectly
    // usable.  Every failure path ends here, so here it is taken off.
    (void)hipGetLastError();
    return ix->err.c_str();
}

#define HIPCHK(ix, expr)                                                                  \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if(e_ != hipSuccess) {                                                            \
            set_err(ix, std::string("lantern_gpu: HIP error in " #expr ": ") + hipGetErrorString(e_)); \
            return false;                                                                 \
        }                                                                                 \
    } while(0)

static bool dev_grow(Index *ix, void **p, size_t old_bytes, size_t new_bytes, int fill_new_with)
{
    void *q = nullptr;
    HIPCHK(ix, hipMalloc(&q, new_bytes ? new_bytes : 16));
    bool ok = true;
    if(*p && old_bytes) ok = hipMemcpyAsync(q, *p, old_bytes, hipMemcpyDeviceToDevice, ix->stream) == hipSuccess;
    if(ok && fill_new_with >= 0 && new_bytes > old_bytes)
        ok = hipMemsetAsync((char *)q + old_bytes, fill_new_with, new_bytes - old_bytes, ix->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(ix->stream) == hipSuccess;
    if(!ok) {  // the old block stays the index's
        (void)hipFree(q);
        set_err(ix, "lantern_gpu: HIP failure while growing a device array");
        return false;
    }
    if(*p) HIPCHK(ix, hipFree(*p));
    *p = q;
    return true;
}

void *scratch(Index *ix, int which, size_t bytes)
{
    if(ix->scratch_bytes[ which ] >= bytes && ix->d_scratch[ which ]) return ix->d_scratch[ which ];
    if(ix->d_scratch[ which ]) (void)hipFree(ix->d_scratch[ which ]);
    ix->d_scratch[ which ] = nullptr;
    ix->scratch_bytes[ which ] = 0;
    size_t want = bytes + bytes / 2 + 256;
    if(hipMalloc(&ix->d_scratch[ which ], want) != hipSuccess) {
        set_err(ix, "lantern_gpu: out of device memory (scratch)");
        return nullptr;
    }
    ix->scratch_bytes[ which ] = want;
    return ix->d_scratch[ which ];
}

View Index::view() const
{
    View v{};
    v.vec = d_vec;
    v.chunks = chunks;
    v.M = M;
    v.M0 = M0;
    v.nbr0 = d_nbr0;
    v.upper_off = d_upper_off;
    v.upper_nbr = d_upper_nbr;
    v.levels = d_levels;
    v.norm2 = d_norm2;
    v.n = (uint32_t)n;
    v.entry = entry;
    v.max_level = max_level;
    return v;
}

static bool reserve_locked(Index *ix, size_t newcap)
{
    if(newcap <= ix->cap) return true;
    if(newcap >= 0x7FFFFFFFull) { set_err(ix, "lantern_gpu: capacity above 2^31-1 slots is not supported"); return false; }
    if(!pq_expand_locked(ix)) return false;  // growing means adding: a compact pq index gets its rows back first
    const size_t oc = ix->cap, row = (size_t)ix->chunks * 16;
    if(!dev_grow(ix, (void **)&ix->d_vec, oc * row, newcap * row, -1)) return false;
    if(mcode_base(ix->mcode) == M_COS && !mcode_is_i8(ix->mcode) && !dev_grow(ix, (void **)&ix->d_norm2, oc * 4, newcap * 4, -1)) return false;
    if(ix->pq && !dev_grow(ix, (void **)&ix->d_codes, oc * ix->pq_S, newcap * ix->pq_S, 0)) return false;
    if(!dev_grow(ix, (void **)&ix->d_labels, oc * 8, newcap * 8, -1)) return false;
    if(!dev_grow(ix, (void **)&ix->d_levels, oc, newcap, 0)) return false;
    if(!dev_grow(ix, (void **)&ix->d_nbr0, oc * ix->M0 * 4, newcap * ix->M0 * 4, 0xFF)) return false;
    if(!dev_grow(ix, (void **)&ix->d_radius0, oc * 4, newcap * 4, 0xFF)) return false;  // 0xFFFFFFFF is a NaN: "no state"
    if(!dev_grow(ix, (void **)&ix->d_upper_off, oc * 4, newcap * 4, 0xFF)) return false;
    ix->cap = newcap;
    // the visited bitmaps are sized by capacity
    if(ix->d_bitmaps) { (void)hipFree(ix->d_bitmaps); ix->d_bitmaps = nullptr; }
    ix->bitmap_slots = 0;
    ix->bm_words = 0;
    return true;
}

static bool reserve_upper(Index *ix, size_t need_blocks)
{
    if(need_blocks <= ix->upper_cap) return true;
    size_t nc = std::max<size_t>(std::max<size_t>(ix->upper_cap * 2, need_blocks), 1024);
    if(!dev_grow(ix, (void **)&ix->d_upper_nbr, ix->upper_cap * ix->M * 4, nc * ix->M * 4, 0xFF)) return false;
    if(!dev_grow(ix, (void **)&ix->d_radius_upper, ix->upper_cap * 4, nc * 4, 0xFF)) return false;
    ix->upper_cap = nc;
    return true;
}

bool fill_norms(Index *ix, size_t first, size_t count)
{
    if(!ix->d_norm2 || count == 0) return true;
    HIPCHK(ix, launch_fill_norms(ix->mcode, ix->view(), (uint32_t)first, (uint32_t)count, ix->d_norm2, ix->stream));
    return true;
}

// ---- product quantisation -------------------------------------------------------------------------------------------
// Rows [first, first + count) of the vector block hold the caller's f32 vectors: replace each by its quantisation.
// Per subvector the nearest centroid under the index metric, the first minimum winning -- the rule of the reference's
// quantize_vector / assign_to_clusters (product_quantization.c:80-124, :207-240: a strict `<` scan over usearch_distance)
// -- found by the exact k-NN machinery (fp32-MFMA contraction, exact re-rank in the pair kernel's reduction order).
bool pq_encode_rows(Index *ix, size_t first, size_t count)
{
    if(!ix->pq || count == 0) return true;
    const uint32_t sub_chunks = (ix->pq_subdim + 3) / 4, sub_floats = sub_chunks * 4, row_floats = ix->chunks * 4;
    float    *d_sub = nullptr;
    uint32_t *d_near = nullptr;
    bool ok = hipMalloc((void **)&d_sub, count * (size_t)sub_floats * 4) == hipSuccess && hipMalloc((void **)&d_near, count * 8) == hipSuccess;
    for(uint32_t sv = 0; ok && sv < ix->pq_S; ++sv) {
        ok = ok && launch_pq_take((const float *)ix->d_vec, row_floats, (uint32_t)first, (uint32_t)count, sv, ix->pq_subdim, sub_floats, d_sub, ix->stream) == hipSuccess;
        ok = ok && ::exact_knn_device(ix->metric, sub_chunks, (const uint4 *)(ix->d_centers + (size_t)sv * ix->pq_C * sub_floats), ix->pq_C, (const uint4 *)d_sub,
                                    count, 1, d_near, (float *)(d_near + count), ix->stream);
        ok = ok && launch_pq_put((float *)ix->d_vec, row_floats, (uint32_t)first, (uint32_t)count, sv, ix->pq_subdim, ix->pq_S, d_near, ix->d_codebook,
                                 (uint32_t)ix->opts.dimensions, ix->d_codes, ix->stream) == hipSuccess;
    }
    ok = ok && hipStreamSynchronize(ix->stream) == hipSuccess;
    if(d_sub) (void)hipFree(d_sub);
    if(d_near) (void)hipFree(d_near);
    if(!ok) set_err(ix, "lantern_gpu: HIP failure while quantising vectors");
    return ok;
}

bool pq_decode_rows(Index *ix, size_t first, size_t count)
{
    if(!ix->pq || count == 0) return true;
    HIPCHK(ix, launch_pq_decode((float *)ix->d_vec, ix->chunks * 4, (uint32_t)first, (uint32_t)count, ix->pq_subdim, ix->pq_S, ix->d_codebook,
                                (uint32_t)ix->opts.dimensions, ix->d_codes, ix->stream));
    return true;
}

// A pq index needs its decodings only to ADD (the walk of a new node, the selection heuristic and the re-prunes evaluate
// stored row against stored row).  A read-mostly index -- the scan-side mirror of a built index -- drops them: num_subvectors
// bytes per row stay resident (10M x 768 at 96 subvectors: 0.96 GB instead of 30.7 GB) and searches evaluate rows by ADC.
// ceil(2^16 / cps) if (chunk * that) >> 16 == chunk / cps for every chunk of a row (checked, not assumed), else 0
static uint32_t pqd_inverse(uint32_t cps, uint32_t chunks)
{
    if(cps == 0 || chunks == 0 || chunks > 4096) return 0;
    const uint32_t inv = (65536u + cps - 1) / cps;
    for(uint32_t ch = 0; ch < chunks; ++ch)
        if(((ch * inv) >> 16) != ch / cps) return 0;
    return inv;
}

bool pq_compact_locked(Index *ix)
{
    if(!ix->pq) { set_err(ix, "lantern_gpu: not a pq index"); return false; }
    if(ix->pq_compact) return true;
    if(!flush_locked(ix)) return false;
    uint32_t S16 = (ix->pq_S + 15) / 16 * 16;
    // Code rows of 65 .. 127 bytes (96 subvectors: Lantern's usual 768 / 8) are stored at a 128-BYTE STRIDE, zero padded, for the reason
    // bit rows are (usearch_init): the fabric fetches 128-byte lines, and a 96-byte row at a 96-byte stride straddles two of them three
    // times in four.  Code bytes are addressed by subvector, so the decoding walk never looks at the padding; the table walk adds the
    // padding's all-zero table rows at the end of its chains (x + 0.0f: the same bits).  HBM only: files and the ABI keep num_subvectors bytes.
    if(S16 > 64 && S16 < 128) S16 = 128;
    if(S16 > 128 || ix->pq_C > (uint32_t)ADC_LUT_STRIDE) { set_err(ix, "lantern_gpu: the compact form takes up to 128 subvectors and 256 centroids"); return false; }
    HIPCHK(ix, hipStreamSynchronize(ix->stream));
    for(int l = 0; l < Index::kLanes; ++l)
        if(ix->lane_stream[ l ]) HIPCHK(ix, hipStreamSynchronize(ix->lane_stream[ l ]));
    if(ix->d_codes16) { (void)hipFree(ix->d_codes16); ix->d_codes16 = nullptr; }
    const size_t rows = std::max<size_t>(ix->n, 1);
    if(hipMalloc((void **)&ix->d_codes16, rows * S16) != hipSuccess) { set_err(ix, "lantern_gpu: out of device memory (code rows)"); return false; }
    HIPCHK(ix, hipMemset(ix->d_codes16, 0, rows * S16));
    if(ix->n) HIPCHK(ix, hipMemcpy2D(ix->d_codes16, S16, ix->d_codes, ix->pq_S, ix->pq_S, ix->n, hipMemcpyDeviceToDevice));
    ix->pq_S16 = S16;
    // rows decoded on the fly (device_common.hpp PqdRow) need subvectors of whole 16-byte chunks
    ix->pqd_inv = ix->pq_subdim % 4 == 0 && ix->pq_subdim >= 4 ? pqd_inverse(ix->pq_subdim / 4, ix->chunks) : 0;
    if(ix->d_vec) { HIPCHK(ix, hipFree(ix->d_vec)); ix->d_vec = nullptr; }
    ix->pq_compact = true;
    return true;
}

bool pq_expand_locked(Index *ix)
{
    if(!ix->pq_compact) return true;
    // decode into a buffer of its own: the index stays compact -- and searchable by ADC -- unless every step succeeded
    const size_t row = (size_t)ix->chunks * 16, bytes = std::max<size_t>(ix->cap, 1) * row;
    void        *fresh = nullptr;
    if(hipMalloc(&fresh, bytes) != hipSuccess) {
        set_err(ix, "lantern_gpu: out of device memory (decoding a compact pq index)");
        return false;
    }
    ix->d_vec = (uint4 *)fresh;  // pq_decode_rows writes through the index's view
    const bool ok = hipMemsetAsync(fresh, 0, bytes, ix->stream) == hipSuccess && pq_decode_rows(ix, 0, ix->n) &&
                    hipStreamSynchronize(ix->stream) == hipSuccess;
    if(!ok) {
        ix->d_vec = nullptr;
        (void)hipFree(fresh);
        if(ix->err.empty()) set_err(ix, "lantern_gpu: HIP failure while decoding a compact pq index");
        return false;
    }
    ix->pq_compact = false;
    if(ix->d_codes16) { (void)hipFree(ix->d_codes16); ix->d_codes16 = nullptr; }
    return true;
}

bool ensure_bitmaps(Index *ix, size_t slots)
{
    const size_t words = ((std::max<size_t>(ix->cap, 1) + 31) / 32 + 3) / 4 * 4;
    if(ix->d_bitmaps && ix->bitmap_slots >= slots && ix->bm_words == words) return true;
    if(ix->d_bitmaps) { (void)hipFree(ix->d_bitmaps); ix->d_bitmaps = nullptr; }
    slots = std::max(slots, ix->bitmap_slots);
    // a workgroup's region: its bitmap + the undo log that lets every walk leave the bitmap ALL-ZERO (walk.hpp VisUndo) -- zeroed here, once
    const size_t bytes = slots * (words + kVisUndoWords) * 4;
    HIPCHK(ix, hipMalloc((void **)&ix->d_bitmaps, bytes));
    HIPCHK(ix, hipMemset(ix->d_bitmaps, 0, bytes));
    HIPCHK(ix, hipDeviceSynchronize());  // (launches on non-blocking streams do not wait for the null stream's memset)
    ix->bitmap_slots = slots;
    ix->bm_words = words;
    return true;
}

// entries of the bitmaps' undo logs a walk may use: all of them, or LANTERN_GPU_VIS_UNDO (tests of the overflow path: a walk that
// records more ids than that clears its whole bitmap when it ends)
static uint32_t vis_undo_cap()
{
    static const uint32_t cap = [] {
        const char *e = std::getenv("LANTERN_GPU_VIS_UNDO");
        return e ? (uint32_t)std::min<long>(std::max<long>(std::atol(e), 0), (long)kVisUndoWords) : kVisUndoWords;
    }();
    return cap;
}

// bytes of one caller-side vector of scalar kind `kind_in`
size_t input_bytes(const Index *ix, int kind_in)
{
    if(kind_in == usearch_scalar_b1_k) return (ix->opts.dimensions + 7) / 8;
    if(kind_in == usearch_scalar_f16_k) return ix->opts.dimensions * 2;
    if(kind_in == usearch_scalar_i8_k) return ix->opts.dimensions;
    return ix->opts.dimensions * 4;
}

// which caller-side scalar kinds an index takes: its own storage kind, and f32 for an f16 / i8 index (Lantern
// hands f32 arrays to usearch_add / usearch_search_ef whatever quant_bits says: build.c:128, scan.c:220)
bool kind_accepted(const Index *ix, int kind_in)
{
    if(ix->b1_from_f32) return kind_in == usearch_scalar_f32_k;  // quant_bits = 1: real[] in, bits stored
    return kind_in == ix->scalar ||
           ((ix->scalar == usearch_scalar_f16_k || ix->scalar == usearch_scalar_i8_k) && kind_in == usearch_scalar_f32_k);
}

// f32 -> i8 as usearch's i8 storage does it: x * 100, clamped to [-100, 100], truncated toward zero
// (lantern_hnsw/test/sql/hnsw_sq.sql:33-34: "i8 uniform [-1-1]=>[-100,100] quantization"); NaN -> 0
static inline int8_t quantize_i8(float x)
{
    float v = x * 100.0f;
    if(!(v == v)) return 0;
    if(v > 100.0f) v = 100.0f;
    if(v < -100.0f) v = -100.0f;
    return (int8_t)(int)v;
}

// caller vector -> stored row (zero padded to whole 16-byte chunks).  f32 -> f16 is round-to-nearest-even,
// the cast usearch applies at add and at search time for a quant_bits=16 index.
bool pad_row(const Index *ix, const void *vec, int kind_in, uint32_t *dst)
{
    const size_t row_words = (size_t)ix->chunks * 4;
    std::memset(dst, 0, row_words * 4);
    if(ix->b1_from_f32) {
        // usearch's cast to b1x8: bit i = (x_i > 0), most significant bit of each byte first ("binary > 0 quantization",
        // lantern_hnsw/test/sql/hnsw_sq.sql:33-34); NaN compares false
        const float *f = (const float *)vec;
        uint8_t     *b = (uint8_t *)dst;
        for(size_t i = 0; i < ix->opts.dimensions; ++i)
            if(f[ i ] > 0.f) b[ i >> 3 ] |= (uint8_t)(128u >> (i & 7));
    } else if(ix->scalar == usearch_scalar_f16_k && kind_in == usearch_scalar_f32_k) {
        const float *f = (const float *)vec;
        _Float16    *h = (_Float16 *)dst;
        for(size_t i = 0; i < ix->opts.dimensions; ++i) h[ i ] = (_Float16)f[ i ];
    } else if(ix->scalar == usearch_scalar_i8_k && kind_in == usearch_scalar_f32_k) {
        const float *f = (const float *)vec;
        int8_t      *q = (int8_t *)dst;
        for(size_t i = 0; i < ix->opts.dimensions; ++i) q[ i ] = quantize_i8(f[ i ]);
    } else {
        std::memcpy(dst, vec, input_bytes(ix, kind_in));
    }
    return true;
}

// `count` caller rows into their padded, stored form (pad_row each).  A batch in the thousands is the host's largest share of a
// lantern_gpu_search_batch call -- 8192 x 768-d rows are 25 MB through one core: 2.5 ms beside a 6.6 ms search -- so large batches are
// split over a few short-lived threads (rows are independent; measured in round 5: one call at a time 0.77 -> see DESIGN.md 4.6b).
static void pad_rows(const Index *ix, const void *rows, int kind, size_t count, uint32_t *padded)
{
    const size_t row_words = (size_t)ix->chunks * 4, in_bytes = input_bytes(ix, kind);
    auto span = [&](size_t lo, size_t hi) {
        for(size_t i = lo; i < hi; ++i) pad_row(ix, (const char *)rows + i * in_bytes, kind, &padded[ i * row_words ]);
    };
    const size_t bytes = count * row_words * 4;
    size_t       T = bytes >= ((size_t)4 << 20) ? std::min<size_t>(4, count / 512) : 1;
    if(T <= 1) { span(0, count); return; }
    std::vector<std::thread> th;
    size_t                   made = 0;  // threads 1 .. made exist; the calling thread takes share 0 and whatever got no thread
    try {
        th.reserve(T);
        for(size_t t = 1; t < T; ++t, ++made) th.emplace_back(span, count * t / T, count * (t + 1) / T);
    } catch(...) {
    }
    span(0, count / T);
    if(made + 1 < T) span(count * (made + 1) / T, count);
    for(auto &x : th) x.join();
}

// a zeroed work ticket for one launch on `stream` (ring: concurrent launches on different streams get different slots)
static const uint32_t kTicketRing = 64;
static uint32_t *next_ticket(Index *ix, size_t work, int grid, hipStream_t stream)
{
    if(!ix->use_tickets || !ix->d_tickets || work <= (size_t)grid) return nullptr;
    uint32_t *t = ix->d_tickets + (ix->ticket_next++ % kTicketRing);
    if(hipMemsetAsync(t, 0, 4, stream) != hipSuccess) return nullptr;
    return t;
}

// Ordering of launches across streams (index.hpp "launch slots").  lantern_gpu_search_batch_device returns as soon as its
// kernel is queued on the caller's stream and the index mutex is released with the kernel still running; same-stream launches
// are ordered by the stream itself, everything else by the events below.
static bool slot_idle(Index *ix, int s)
{
    if(ix->slot_pending[ s ] && hipEventQuery(ix->slot_done[ s ]) == hipSuccess) ix->slot_pending[ s ] = false;
    return !ix->slot_pending[ s ];
}
bool order_launch(Index *ix, hipStream_t stream)  // an insert batch: after every search in flight
{
    for(int s = 0; s < Index::kSearchSlots; ++s)
        if(!slot_idle(ix, s) && ix->slot_stream[ s ] != stream) HIPCHK(ix, hipStreamWaitEvent(stream, ix->slot_done[ s ], 0));
    return true;
}
bool record_launch(Index *ix, hipStream_t stream)
{
    if(!ix->insert_done) HIPCHK(ix, hipEventCreateWithFlags(&ix->insert_done, hipEventDisableTiming));
    HIPCHK(ix, hipEventRecord(ix->insert_done, stream));
    ix->insert_pending = true;
    return true;
}
int acquire_search_slot(Index *ix, hipStream_t stream, size_t grid)
{
    // a search on another stream than the index's own runs behind the insert batches queued there
    if(ix->insert_pending) {
        if(hipEventQuery(ix->insert_done) == hipSuccess) ix->insert_pending = false;
        else if(stream != ix->stream && hipStreamWaitEvent(stream, ix->insert_done, 0) != hipSuccess) { set_err(ix, "lantern_gpu: hipStreamWaitEvent failed"); return -1; }
    }
    int pick = -1;
    for(int s = 0; s < Index::kSearchSlots && pick < 0; ++s)  // the slot this stream used last: the stream orders the two launches
        if(!slot_idle(ix, s) && ix->slot_stream[ s ] == stream) pick = s;
    for(int s = 0; s < Index::kSearchSlots && pick < 0; ++s)
        if(slot_idle(ix, s)) pick = s;
    if(pick < 0) {  // both slabs busy on other streams: queue behind one of them, alternating
        pick = (int)(ix->slot_next++ % Index::kSearchSlots);
        if(hipStreamWaitEvent(stream, ix->slot_done[ pick ], 0) != hipSuccess) { set_err(ix, "lantern_gpu: hipStreamWaitEvent failed"); return -1; }
    }
    if(pick == 0) {
        if(!ensure_bitmaps(ix, grid)) return -1;
        ix->slot_bitmaps[ 0 ] = ix->d_bitmaps;
        ix->slot_rows[ 0 ] = ix->bitmap_slots;
        ix->slot_words[ 0 ] = ix->bm_words;
    } else {
        const size_t words = ((std::max<size_t>(ix->cap, 1) + 31) / 32 + 3) / 4 * 4;
        if(!ix->slot_bitmaps[ pick ] || ix->slot_rows[ pick ] < grid || ix->slot_words[ pick ] != words) {
            if(ix->slot_bitmaps[ pick ]) (void)hipFree(ix->slot_bitmaps[ pick ]);  // (hipFree waits for the device)
            ix->slot_bitmaps[ pick ] = nullptr;
            const size_t rows = std::max(grid, ix->slot_rows[ pick ]);
            const size_t bytes = rows * (words + kVisUndoWords) * 4;  // (bitmap + undo log per workgroup, all-zero: ensure_bitmaps)
            if(hipMalloc((void **)&ix->slot_bitmaps[ pick ], bytes) != hipSuccess || hipMemset(ix->slot_bitmaps[ pick ], 0, bytes) != hipSuccess ||
               hipDeviceSynchronize() != hipSuccess) {
                (void)hipGetLastError();
                set_err(ix, "lantern_gpu: out of device memory (visited bitmaps)");
                return -1;
            }
            ix->slot_rows[ pick ] = rows;
            ix->slot_words[ pick ] = words;
        }
    }
    return pick;
}
bool release_search_slot(Index *ix, int s, hipStream_t stream)
{
    if(!ix->slot_done[ s ]) HIPCHK(ix, hipEventCreateWithFlags(&ix->slot_done[ s ], hipEventDisableTiming));
    HIPCHK(ix, hipEventRecord(ix->slot_done[ s ], stream));
    ix->slot_stream[ s ] = stream;
    ix->slot_pending[ s ] = true;
    return true;
}

// Resident workgroups of a walk kernel.  k_search is compiled for six waves per SIMD (<= 80 VGPRs:
// __launch_bounds__(512, 6)) and its LDS is sized for six workgroups per CU, i.e. 24 waves per CU; k_insert (wider
// lists, a larger visited set) runs at five.  LANTERN_GPU_WAVES_PER_CU overrides (tuning).
int search_grid(const Index *ix, size_t nq, int waves, int waves_per_cu)
{
    static const int forced = std::getenv("LANTERN_GPU_WAVES_PER_CU") ? std::atoi(std::getenv("LANTERN_GPU_WAVES_PER_CU")) : 0;
    if(forced > 0) waves_per_cu = forced;
    int per_cu = std::max(1, waves_per_cu / std::max(1, waves));
    size_t g = (size_t)ix->num_cus * per_cu;
    if(ix->search_max_wg > 0) g = (size_t)ix->search_max_wg;
    if(g > nq) g = nq;
    return (int)std::max<size_t>(g, 1);
}

// ---------------------------------------------------------------------------------------------------
// inserts
// ---------------------------------------------------------------------------------------------------

// One device pass over `b` new vectors whose rows / labels / levels / upper offsets are ALREADY in HBM at
// slots [first, first + b) (flush_locked uploads everything pending up front: an unlinked node is
// unreachable, so its row may sit in the table before its batch runs).  Nothing in here waits for the device on one
// GPU: layout, walk, selection, the grouping of the reverse-link requests (grouping.hip) and the reverse-link kernels
// are queued on the index stream, and the next batch is queued right behind them (the host knows sizes, levels and the
// entry point of every batch in advance; only the graph itself is device state).
// Work-sharded build (comm != nullptr): the batch is the same as on one GPU, but every rank
// walks and connects only its share [b_lo, b_hi) of the new nodes, the ranks all-gather the resulting top-M
// neighbour lists (= the reverse-link requests: 16 bytes per pick), each rank applies the reverse links of the
// nodes it owns (close % world), and the re-written adjacency rows are all-gathered.  Every step is
// deterministic and independent of who executes it, so all replicas end up bit-identical to the graph one GPU
// builds with the same batch plan.  Batches with fewer than kShardMinPerRank vectors per rank (the ramp at the
// start of a build) are executed redundantly by every rank: no exchange, same result.
static const size_t kShardMinPerRank = 8;

static bool sync_stream(Index *ix, Comm *comm)
{
    if(comm) {
        if(comm->wait(ix->stream)) return true;
        set_err(ix, comm->err);
        return false;
    }
    HIPCHK(ix, hipStreamSynchronize(ix->stream));
    return true;
}

// ---- build profile: HIP events around the phases of every batch, resolved lazily (never a wait on fresh work) -----
static const int kProfMarks = 6;  // start | walk | connect | exchange 1 + grouping | reverse links | exchange 2
static hipEvent_t prof_event(Index *ix)
{
    if(!ix->prof_free.empty()) {
        hipEvent_t e = ix->prof_free.back();
        ix->prof_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if(hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
static void prof_mark(Index *ix, int which)
{
    if(!ix->profiling) return;
    if(which == 0) ix->prof_pending.emplace_back();
    hipEvent_t e = prof_event(ix);
    ix->prof_pending.back().ev[ which ] = e;
    if(e) (void)hipEventRecord(e, ix->stream);
}
void prof_resolve(Index *ix, size_t keep)
{
    while(ix->prof_pending.size() > keep) {
        Index::ProfBatch &pb = ix->prof_pending.front();
        bool ok = true;
        for(int i = 0; i < kProfMarks; ++i) ok = ok && pb.ev[ i ] != nullptr;
        if(ok && hipEventSynchronize(pb.ev[ kProfMarks - 1 ]) == hipSuccess) {
            float ms[ kProfMarks - 1 ] = {};
            for(int i = 0; i + 1 < kProfMarks; ++i) (void)hipEventElapsedTime(&ms[ i ], pb.ev[ i ], pb.ev[ i + 1 ]);
            ix->prof.walk_ms += ms[ 0 ];
            ix->prof.connect_ms += ms[ 1 ];
            ix->prof.group_ms += ms[ 2 ];
            ix->prof.revlink_ms += ms[ 3 ];
            ix->prof.exchange_ms += ms[ 4 ];
            ix->prof.batches += 1;
        }
        for(int i = 0; i < kProfMarks; ++i)
            if(pb.ev[ i ]) ix->prof_free.push_back(pb.ev[ i ]);
        ix->prof_pending.pop_front();
    }
}

// Row-sharded build (add_row_sharded_locked below): where a batch's level-0 candidates come from
struct RowShard
{
    Index *loc;   // this rank's graph over ITS rows (labels = global slot + 1)
    Comm  *comm;
    size_t K;     // candidates a shard answers with per row
    size_t ef;    // ... found with this expansion (>= K)
};
static bool row_shard_candidates(Index *ix, const RowShard &rs, size_t first, size_t b, const uint32_t *d_link_off, uint64_t *d_tops, uint32_t *d_top_count);

static bool run_batch(Index *ix, size_t b, const int *lv, Comm *comm, const RowShard *rs = nullptr)
{
    const size_t first = ix->n;
    const int    W = comm ? comm->world : 1, R = comm ? comm->rank : 0;
    const bool   split = W > 1 && b >= (size_t)W * kShardMinPerRank;
    const size_t b_lo = split ? b * (size_t)R / (size_t)W : 0, b_hi = split ? b * ((size_t)R + 1) / (size_t)W : b;
    std::vector<uint32_t> &link_off = ix->h_link_off;  // host copy: sizes the exchanges and the launches (never uploaded)
    link_off.resize(b);
    size_t total_links = 0;
    for(size_t i = 0; i < b; ++i) {
        link_off[ i ] = (uint32_t)total_links;
        total_links += (size_t)ix->M * (size_t)(lv[ i ] + 1);
    }
    // one "item" per (new node, level): the walk result the selection kernel works on
    const size_t items = total_links / ix->M;
    uint32_t *d_link_off = (uint32_t *)scratch(ix, 0, b * 4 + items * 4 + items * 4);
    LinkReq  *d_links = (LinkReq *)scratch(ix, 1, total_links * sizeof(LinkReq));
    uint64_t *d_tops = (uint64_t *)scratch(ix, 7, items * (size_t)ix->efc * 8);
    LinkReq  *d_reqs = (LinkReq *)scratch(ix, 2, total_links * sizeof(LinkReq));
    // groups | ngroups | owner counts
    char     *d_grp = (char *)scratch(ix, 3, total_links * sizeof(uint2) + 16 + (size_t)W * 4);
    void     *d_work = scratch(ix, 4, total_links * 8 + 16);
    const size_t temp_bytes = group_temp_bytes(total_links);
    char     *d_sort = (char *)scratch(ix, 8, total_links * 24 + temp_bytes + 64);
    if(!d_link_off || !d_links || !d_tops || !d_reqs || !d_grp || !d_work || !d_sort) return false;
    uint32_t *d_item_node = d_link_off + b, *d_top_count = d_item_node + items;
    uint2    *d_groups = (uint2 *)d_grp;
    uint32_t *d_ngroups = (uint32_t *)(d_grp + total_links * sizeof(uint2));
    uint32_t *d_owner = d_ngroups + 4;
    GroupScratch gs;
    gs.keys_a = (uint64_t *)d_sort;
    gs.keys_b = gs.keys_a + total_links;
    gs.idx_a = (uint32_t *)(gs.keys_b + total_links);
    gs.idx_b = gs.idx_a + total_links;
    gs.temp = (void *)(((uintptr_t)(gs.idx_b + total_links) + 63) & ~(uintptr_t)63);
    gs.temp_bytes = temp_bytes;

    prof_mark(ix, 0);
    HIPCHK(ix, launch_batch_layout(ix->d_levels + first, (uint32_t)b, ix->M, d_link_off, d_item_node, ix->stream));

    // A handful of insertions -- ldb_aminsert's one row, the first batches of a build -- walk alone on their CUs: level 0 by the
    // lone-query walk (insert_spec_kernel.hip), up to two insertions per CU one after the other (LANTERN_GPU_INSERT_SPEC=0: off).
    static const bool ins_spec_env = !(std::getenv("LANTERN_GPU_INSERT_SPEC") && std::atoi(std::getenv("LANTERN_GPU_INSERT_SPEC")) == 0);
    const bool ins_spec = ins_spec_env && !comm && !rs && b_hi - b_lo <= (size_t)ix->num_cus * 2 && insert_spec_supported(ix->mcode, ix->efc, ix->M0) && !lds_list_env();
    const int  ins_waves = ins_spec ? 11 : ix->insert_waves;
    const int  grid = ins_spec ? (int)std::max<size_t>(1, std::min<size_t>(b_hi - b_lo, (size_t)ix->num_cus)) : search_grid(ix, b_hi - b_lo, ix->insert_waves, 20);
    if(!ensure_bitmaps(ix, (size_t)grid)) return false;
    if(!order_launch(ix, ix->stream)) return false;  // a search may still be running on a caller's stream
    auto link_at = [&](size_t i) { return i < b ? (size_t)link_off[ i ] : total_links; };

    InsertArgs ia;
    ia.view = ix->view();  // size/entry/max_level as they were BEFORE the batch
    ia.first_slot = (uint32_t)first;
    ia.b_begin = (uint32_t)b_lo;
    ia.count = (uint32_t)b_hi;
    ia.efc = ix->efc;
    ia.link_off = d_link_off;
    ia.tops = d_tops;
    ia.top_count = d_top_count;
    ia.bitmaps = ix->d_bitmaps;
    ia.bm_words = (uint32_t)ix->bm_words;
    ia.undo_cap = vis_undo_cap();
    ia.lds_list = lds_list_env();
    ia.only_upper = rs ? 1u : 0u;
    // LDS visited set for the ef_construction-wide walk (spills to the bitmap when 3/4 full); env override for tuning
    // the largest table that still lets FIVE workgroups share a CU (160 KB / 5, minus the walk's lists): 6400 slots at
    // 768-d / efc 128, enough for the ~3700 nodes such a walk visits at the 3/4 load limit
    uint32_t ivis = 8192;
    if(const char *vs = std::getenv("LANTERN_GPU_INSERT_VIS_SLOTS")) ivis = (uint32_t)std::atoi(vs) / 4 * 4;
    const int      G_ = group_lanes_for(ix->chunks), LW_ = G_ >= 32 ? 1 : G_ == 16 ? 2 : 4;  // list words a lane of a row's group fetches
    ia.spec_prefetch = ins_spec && ix->M0 % (uint32_t)LW_ == 0 && ix->M0 <= (uint32_t)(G_ * LW_) ? 1u : 0u;
    ia.spec_cache = ia.spec_prefetch ? 128u : 0u;
    auto ins_lds = [&](uint32_t vis) {
        return ins_spec ? insert_spec_lds_bytes(ix->chunks, ix->efc, ix->M0, vis, ia.spec_prefetch, ia.spec_cache) : insert_lds_bytes(ix->chunks, ix->efc, ix->M0, vis);
    };
    while(ivis && ins_lds(ivis) > (ins_spec ? 96u : 31u) * 1024) ivis = ivis > 256 ? ivis - 256 : 0;  // (one workgroup per CU in the lone-walk shape)
    if(ivis && ivis < 4 * ix->M0) ivis = 0;
    ia.vis_slots = ivis;
    ia.totals = ix->d_totals + 2;
    ia.ticket = next_ticket(ix, b_hi - b_lo, grid, ix->stream);
    if(ins_lds(ivis) > 160 * 1024) { set_err(ix, "lantern_gpu: ef_construction/dimensions exceed the 160 KiB LDS budget"); return false; }
    if(ins_spec) HIPCHK(ix, launch_insert_spec(ix->mcode, ia, ins_waves, grid, ix->stream));
    else HIPCHK(ix, launch_insert(ix->mcode, ia, ix->insert_waves, grid, ix->stream));
    if(rs && !row_shard_candidates(ix, *rs, first, b, d_link_off, d_tops, d_top_count)) return false;  // level 0: from the shards' graphs
    prof_mark(ix, 1);

    ConnectArgs ca;
    ca.view = ia.view;
    ca.first_slot = (uint32_t)first;
    ca.item_begin = (uint32_t)(link_at(b_lo) / ix->M);
    ca.items = (uint32_t)((link_at(b_hi) - link_at(b_lo)) / ix->M);
    ca.efc = ix->efc;
    ca.link_off = d_link_off;
    ca.item_node = d_item_node;
    ca.tops = d_tops;
    ca.top_count = d_top_count;
    ca.links = d_links;
    ca.totals = ix->d_totals + 4;
    HIPCHK(ix, launch_connect(ix->mcode, ca, ix->stream));
    prof_mark(ix, 2);

    if(split) {
        // exchange 1: the ranks' top-M neighbour lists (one LinkReq per pick).  Afterwards every rank holds all
        // requests of the batch; the own lists of the nodes a peer connected are rebuilt from them.
        std::vector<size_t> off((size_t)W), cnt((size_t)W);
        for(int r = 0; r < W; ++r) {
            const size_t lo = b * (size_t)r / (size_t)W, hi = b * ((size_t)r + 1) / (size_t)W;
            off[ (size_t)r ] = link_at(lo) * sizeof(LinkReq);
            cnt[ (size_t)r ] = (link_at(hi) - link_at(lo)) * sizeof(LinkReq);
        }
        if(!comm->allgatherv_device(d_links, off.data(), cnt.data(), ix->stream)) { set_err(ix, comm->err); return false; }
        HIPCHK(ix, launch_apply_own_links(ia.view, (uint32_t)first, d_link_off, d_links, (uint32_t)total_links, ix->stream));
    }

    // reverse links: group by (close, level); within a group apply in new-slot order -- on the device (grouping.hip).
    // In a sharded batch a rank keeps only the groups of the nodes it owns (close % world) and counts the others'
    // (the sizes of the second exchange's segments must be known to every rank's HOST: the one value read back).
    // (the pass also zeroes the reverse-link kernels' work counter, d_work[0]: one node less on the stream)
    std::vector<uint32_t> owner_reqs((size_t)W, 0);
    // A sharded batch sorts and gathers only the requests THIS rank owns (1 / W of them: DESIGN.md 6): their keys are appended while
    // the owners are counted, the counts come back in the batch's one host wait, and the sort runs on owner_reqs[R] keys.
    // (LANTERN_GPU_GROUP_ALL=1: the replicated pass over every request, kept for A/B runs; positions need 24 bits.)
    static const bool group_all = std::getenv("LANTERN_GPU_GROUP_ALL") && std::atoi(std::getenv("LANTERN_GPU_GROUP_ALL")) != 0;
    const bool        owned_only = split && !group_all && total_links < (1u << 24);
    if(owned_only) HIPCHK(ix, launch_group_keys_owned(d_links, (uint32_t)total_links, gs, d_ngroups, W, R, d_owner, ix->stream, (uint32_t *)d_work));
    else HIPCHK(ix, launch_group_requests(d_links, (uint32_t)total_links, gs, d_reqs, d_groups, d_ngroups, split ? W : 1, R, d_owner, ix->stream, (uint32_t *)d_work));
    if(split) {
        HIPCHK(ix, hipMemcpyAsync(owner_reqs.data(), d_owner, (size_t)W * 4, hipMemcpyDeviceToHost, ix->stream));
        if(!sync_stream(ix, comm)) return false;
        if(owned_only) HIPCHK(ix, launch_group_sort_owned(d_links, owner_reqs[ (size_t)R ], gs, d_reqs, d_groups, d_ngroups, ix->stream));
    }
    prof_mark(ix, 3);
    if(std::getenv("LANTERN_GPU_DEBUG_GROUPS")) {  // debugging aid: the size distribution of this batch's groups
        HIPCHK(ix, hipStreamSynchronize(ix->stream));
        uint32_t ng = 0;
        HIPCHK(ix, hipMemcpy(&ng, d_ngroups, 4, hipMemcpyDeviceToHost));
        std::vector<uint2> hg(ng);
        if(ng) HIPCHK(ix, hipMemcpy(hg.data(), d_groups, (size_t)ng * 8, hipMemcpyDeviceToHost));
        uint32_t mx = 0, over[ 5 ] = {};
        uint64_t sum = 0;
        for(const uint2 &g2 : hg) {
            const uint32_t sz = g2.y - g2.x;
            mx = std::max(mx, sz);
            sum += sz;
            over[ 0 ] += sz > 8;
            over[ 1 ] += sz > 32;
            over[ 2 ] += sz > 128;
            over[ 3 ] += sz > 512;
            over[ 4 ] += sz > 2048;
        }
        std::fprintf(stderr, "batch first=%zu b=%zu groups=%u reqs=%llu max=%u >8:%u >32:%u >128:%u >512:%u >2048:%u\n", first, b, ng,
                     (unsigned long long)sum, mx, over[ 0 ], over[ 1 ], over[ 2 ], over[ 3 ], over[ 4 ]);
    }
    RevlinkArgs ra;
    ra.view = ix->view();
    ra.ngroups = d_ngroups;
    ra.groups = d_groups;
    ra.max_groups = split ? owner_reqs[ (size_t)R ] : (uint32_t)total_links;
    ra.reqs = d_reqs;
    ra.totals = ix->d_totals + 5;
    // the persistent re-prune state is kept by one-GPU builds (a work-sharded build overwrites other ranks' rows wholesale);
    // whatever changes lists without maintaining it (a sharded batch, an imported graph, the switch below) marks it stale
    const char *rs_env = std::getenv("LANTERN_GPU_REPRUNE_STATE");  // =0: every full list takes the all-pairs path (A/B, tests)
    const bool  use_state = !split && !(rs_env && std::atoi(rs_env) == 0);
    if(!use_state) {
        ix->radius_stale = true;
    } else if(ix->radius_stale) {
        if(ix->d_radius0) HIPCHK(ix, hipMemsetAsync(ix->d_radius0, 0xFF, ix->cap * 4, ix->stream));
        if(ix->d_radius_upper) HIPCHK(ix, hipMemsetAsync(ix->d_radius_upper, 0xFF, ix->upper_cap * 4, ix->stream));
        ix->radius_stale = false;
    }
    ra.radius0 = use_state ? ix->d_radius0 : nullptr;
    ra.radius_upper = use_state ? ix->d_radius_upper : nullptr;
    HIPCHK(ix, launch_revlink(ix->mcode, ra, (char *)d_work + 16, (uint32_t *)d_work, ix->num_cus, ix->stream, true));
    prof_mark(ix, 4);
    if(split) {
        // exchange 2: the adjacency rows the ranks re-wrote, one record [close, level, list[0..M0)] per group.
        // A rank's segment is sized by the number of requests it owned (>= its number of groups; every rank can
        // compute it from the request array) and padded with EMPTY records.
        const size_t rec_bytes = ((size_t)ix->M0 + 2) * 4;
        std::vector<size_t> off((size_t)W), cnt((size_t)W);
        size_t total_recs = 0;
        for(int r = 0; r < W; ++r) {
            off[ (size_t)r ] = total_recs * rec_bytes;
            cnt[ (size_t)r ] = owner_reqs[ (size_t)r ] * rec_bytes;
            total_recs += owner_reqs[ (size_t)r ];
        }
        if(total_recs) {
            char *d_rec = (char *)scratch(ix, 5, total_recs * rec_bytes);
            if(!d_rec) return false;
            if(cnt[ (size_t)R ]) HIPCHK(ix, hipMemsetAsync(d_rec + off[ (size_t)R ], 0xFF, cnt[ (size_t)R ], ix->stream));
            HIPCHK(ix, launch_pack_lists(ra, (uint32_t *)(d_rec + off[ (size_t)R ]), ix->stream));
            if(!comm->allgatherv_device(d_rec, off.data(), cnt.data(), ix->stream)) { set_err(ix, comm->err); return false; }
            HIPCHK(ix, launch_apply_lists(ra.view, (const uint32_t *)d_rec, (uint32_t)total_recs, ix->stream));
        }
        // the host transport's staging buffers are reused by the next batch: it waits here.  RCCL works in place in HBM on the
        // index stream: the next batch queues behind this one, and the one wait per batch that is left (the owner counts that
        // size exchange 2, above) is where a stuck collective meets its deadline
        if(!comm->rccl && !sync_stream(ix, comm)) return false;
    }
    prof_mark(ix, 5);
    // behind the LAST kernel that rewrites lists: a search on another stream that waits for `insert_done` sees the whole batch
    if(!record_launch(ix, ix->stream)) return false;
    if(ix->profiling) prof_resolve(ix, 192);  // only batches the device finished long ago are waited for
    ix->n = first + b;
    if(b == 1 && lv[ 0 ] > ix->max_level) {  // "Updating the entry point if needed"
        ix->entry = (uint32_t)first;
        ix->max_level = lv[ 0 ];
    }
    ix->c_add_vectors += b;
    ix->c_add_batches += 1;
    return true;
}

// Levels and upper-block offsets of `count` new nodes at slots [ix->n, ix->n + count), capacity for them.
struct StagedMeta
{
    std::vector<int>      lv;
    std::vector<uint8_t>  l8;
    std::vector<uint32_t> uo;
};
static bool stage_meta(Index *ix, const int *levels_in, size_t count, StagedMeta &s)
{
    const size_t first = ix->n;
    s.lv.resize(count);
    s.l8.resize(count);
    s.uo.resize(count);
    size_t blocks = ix->upper_blocks;
    for(size_t i = 0; i < count; ++i) {
        s.lv[ i ] = (levels_in && levels_in[ i ] >= 0) ? levels_in[ i ] : level_for(ix->seed, first + i, ix->M);
        s.l8[ i ] = (uint8_t)s.lv[ i ];
        s.uo[ i ] = s.lv[ i ] > 0 ? (uint32_t)blocks : EMPTY;
        blocks += (size_t)s.lv[ i ];
    }
    if(first + count > ix->cap && !reserve_locked(ix, std::max(ix->cap * 2, first + count))) return false;
    return reserve_upper(ix, blocks);
}

// The batch loop over `count` staged nodes whose rows / labels / levels / upper offsets are in HBM.  Returns how
// many were inserted (== count unless a batch failed) and appends the host mirrors of those.
static size_t run_batches(Index *ix, const uint64_t *labels, const StagedMeta &s, size_t count, Comm *comm, bool *ok_out)
{
    size_t pi = 0;
    bool   ok = true;
    while(pi < count) {
        if(ix->n == 0) {  // "Do nothing for the first element": it only becomes the entry point
            ix->n = 1;
            ix->entry = 0;
            ix->max_level = s.lv[ 0 ];
            ix->c_add_vectors += 1;
            pi += 1;
            continue;
        }
        const size_t look = std::min(count - pi, ix->add_batch_max);
        const size_t b = plan_batch(ix->n, ix->max_level, s.lv.data() + pi, look, ix->add_batch_max, ix->add_min_ratio);
        if(!(ok = run_batch(ix, b, s.lv.data() + pi, comm))) break;
        pi += b;
    }
    // the batches were queued without waiting for the device: one synchronisation per flush surfaces a failure and makes
    // the scratch buffers safe to re-size
    if(ok && !sync_stream(ix, comm)) ok = false;
    if(ix->profiling) prof_resolve(ix, 0);
    // host mirrors of what was inserted
    ix->labels.insert(ix->labels.end(), labels, labels + pi);
    ix->levels.insert(ix->levels.end(), s.l8.begin(), s.l8.begin() + (ptrdiff_t)pi);
    ix->upper_off.insert(ix->upper_off.end(), s.uo.begin(), s.uo.begin() + (ptrdiff_t)pi);
    for(size_t i = 0; i < pi; ++i) ix->upper_blocks += (size_t)s.lv[ i ];
    *ok_out = ok;
    return pi;
}

static const size_t kStageBytes = 256 * 1024;
static bool stage_buffer(Index *ix)
{
    if(ix->h_stage) return true;
    if(hipHostMalloc((void **)&ix->h_stage, kStageBytes, hipHostMallocMapped) != hipSuccess ||
       hipHostGetDevicePointer((void **)&ix->h_stage_dev, ix->h_stage, 0) != hipSuccess) {
        (void)hipGetLastError();
        if(ix->h_stage) (void)hipHostFree(ix->h_stage);
        ix->h_stage = ix->h_stage_dev = nullptr;
        return false;  // (the pageable path still works)
    }
    ix->h_stage_bytes = kStageBytes;
    return true;
}

// `count` f32 vectors of the caller (dimensions floats each) into `dst` in the index's STORED form (f16 / i8 / sign bits): uploaded as
// they are and converted on the device (a million 768-d rows cost seconds on the host).  Returns after the stream has drained: the
// caller's buffer is pageable and borrowed.  Shared by insert_rows and the row-sharded build's stage_rows_at.
static bool upload_f32_as_stored(Index *ix, const void *f32_rows, size_t count, uint32_t *dst)
{
    const size_t d = ix->opts.dimensions;
    float       *tmp = nullptr;
    bool         up = hipMalloc((void **)&tmp, count * d * 4) == hipSuccess;
    up = up && hipMemcpyAsync(tmp, f32_rows, count * d * 4, hipMemcpyHostToDevice, ix->stream) == hipSuccess;
    up = up && launch_store_quantised(tmp, (uint32_t)d, (uint32_t)count, ix->scalar, dst, (uint32_t)ix->chunks * 4, ix->stream) == hipSuccess;
    up = up && hipStreamSynchronize(ix->stream) == hipSuccess;
    if(tmp) (void)hipFree(tmp);
    if(!up) (void)hipGetLastError();
    return up;
}

// Insert `count` vectors whose padded rows start at `rows` (row_words 4-byte words each).  levels[i] < 0
// means "draw with level_for()".  Returns how many were inserted (== count unless a batch failed).
// raw_f32: `rows` are the caller's f32 vectors (dimensions floats each) of an index with quantised storage: they are
// uploaded as they are and converted to stored rows on the device (a million 768-d rows cost seconds on the host).
static size_t insert_rows(Index *ix, const uint64_t *labels, const int *levels_in, const uint32_t *rows, size_t count, bool *ok_out,
                          bool raw_f32 = false)
{
    *ok_out = true;
    if(count == 0) return 0;
    auto fail = [&]() { *ok_out = false; return (size_t)0; };
    if(!pq_expand_locked(ix)) return fail();  // a compact pq index gets its decodings back before anything is added
    const size_t first = ix->n, row_words = (size_t)ix->chunks * 4;
    // ---- levels and upper-block offsets of everything, then ONE upload of rows and metadata: an unlinked
    // node is unreachable, so its row may sit in the table before its batch runs
    StagedMeta s;
    if(!stage_meta(ix, levels_in, count, s)) return fail();
    bool up = true, staged = false;
    if(raw_f32) {
        up = upload_f32_as_stored(ix, rows, count, (uint32_t *)ix->d_vec + first * row_words);
    } else if(count <= 64 && count * (row_words * 4 + 16) <= kStageBytes && stage_buffer(ix)) {
        // a handful of rows: one kernel reads them and their metadata from the page-locked, device-mapped block (the block is next
        // written by the next insertion, which starts after this one's closing synchronisation in run_batches)
        char *hs = ix->h_stage, *h_rows = hs, *h_lab = h_rows + count * row_words * 4, *h_uo = h_lab + count * 8, *h_lv = h_uo + count * 4;
        std::memcpy(h_rows, rows, count * row_words * 4);
        std::memcpy(h_lab, labels, count * 8);
        std::memcpy(h_uo, s.uo.data(), count * 4);
        std::memcpy(h_lv, s.l8.data(), count);
        char *dv = ix->h_stage_dev;
        up = launch_stage_small(dv, (const uint64_t *)(dv + (h_lab - hs)), (const uint32_t *)(dv + (h_uo - hs)), (const uint8_t *)(dv + (h_lv - hs)), (uint32_t)count,
                                ix->chunks, (char *)ix->d_vec + first * row_words * 4, ix->d_labels + first, ix->d_upper_off + first, ix->d_levels + first,
                                ix->stream) == hipSuccess;
        staged = true;
    } else {
        up = hipMemcpyAsync((char *)ix->d_vec + first * row_words * 4, rows, count * row_words * 4, hipMemcpyHostToDevice, ix->stream) == hipSuccess;
    }
    if(!staged) {
        up = up && hipMemcpyAsync(ix->d_labels + first, labels, count * 8, hipMemcpyHostToDevice, ix->stream) == hipSuccess;
        up = up && hipMemcpyAsync(ix->d_levels + first, s.l8.data(), count, hipMemcpyHostToDevice, ix->stream) == hipSuccess;
        up = up && hipMemcpyAsync(ix->d_upper_off + first, s.uo.data(), count * 4, hipMemcpyHostToDevice, ix->stream) == hipSuccess;
        up = up && hipStreamSynchronize(ix->stream) == hipSuccess;
    }
    // a failure below must not leave k_stage_small queued: it reads the device-mapped block the NEXT insertion overwrites, and
    // would land the failed rows in slots the host considers free
    auto fail_staged = [&]() { if(staged) (void)hipStreamSynchronize(ix->stream); return fail(); };
    if(!up) { set_err(ix, "lantern_gpu: HIP failure uploading vectors"); return fail_staged(); }
    if(!pq_encode_rows(ix, first, count)) return fail_staged();  // pq = true: the rows become their decodings, the codes go beside them
    if(!fill_norms(ix, first, count)) return fail_staged();
    const size_t done = run_batches(ix, labels, s, count, nullptr, ok_out);
    if(!*ok_out && staged) (void)hipStreamSynchronize(ix->stream);
    return done;
}

// lantern_gpu_add_sharded: a COLLECTIVE insert.  Every rank contributes the rows [shard_off, shard_off + n_shard)
// of the global slot order (rank order); the shards are all-gathered into every replica's HBM (the vector block
// is replicated: 1M x 1536 f32 is 6.1 GB of 288), then the batches run work-sharded (run_batch).
bool add_sharded_locked(Index *ix, Comm *comm, const uint64_t *labels, const void *vectors, size_t n_shard, int kind_in)
{
    if(!flush_locked(ix)) return false;
    const int    W = comm->world, R = comm->rank;
    const size_t row = (size_t)ix->chunks * 16, first = ix->n;
    // ---- shard sizes; all replicas must be in the same state
    std::vector<uint64_t> meta((size_t)W * 2, 0);
    std::vector<size_t>   off((size_t)W), cnt((size_t)W);
    meta[ (size_t)R * 2 ] = n_shard;
    meta[ (size_t)R * 2 + 1 ] = first;
    for(int r = 0; r < W; ++r) { off[ (size_t)r ] = (size_t)r * 16; cnt[ (size_t)r ] = 16; }
    if(!comm->allgatherv_host(meta.data(), off.data(), cnt.data())) { set_err(ix, comm->err); return false; }
    size_t total = 0, my_off = 0;
    for(int r = 0; r < W; ++r) {
        if(meta[ (size_t)r * 2 + 1 ] != first) { set_err(ix, "lantern_gpu: the ranks' replicas differ in size; add_sharded needs identical replicas"); return false; }
        if(r == R) my_off = total;
        total += meta[ (size_t)r * 2 ];
    }
    if(total == 0) return true;
    StagedMeta s;
    if(!stage_meta(ix, nullptr, total, s)) return false;
    // ---- own shard -> HBM, then the all-gather of the shards (vectors, labels) over the transport
    const size_t in_bytes = input_bytes(ix, kind_in);
    std::vector<uint32_t> padded;
    const void *src = vectors;
    if(n_shard && !(kind_in == ix->scalar && in_bytes == row)) {
        padded.resize(n_shard * (size_t)ix->chunks * 4);
        for(size_t i = 0; i < n_shard; ++i) pad_row(ix, (const char *)vectors + i * in_bytes, kind_in, &padded[ i * (size_t)ix->chunks * 4 ]);
        src = padded.data();
    }
    char     *vec_base = (char *)ix->d_vec + first * row;
    uint64_t *lab_base = ix->d_labels + first;
    if(n_shard) {
        HIPCHK(ix, hipMemcpyAsync(vec_base + my_off * row, src, n_shard * row, hipMemcpyHostToDevice, ix->stream));
        HIPCHK(ix, hipMemcpyAsync(lab_base + my_off, labels, n_shard * 8, hipMemcpyHostToDevice, ix->stream));
    }
    HIPCHK(ix, hipMemcpyAsync(ix->d_levels + first, s.l8.data(), total, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(ix, hipMemcpyAsync(ix->d_upper_off + first, s.uo.data(), total * 4, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(ix, hipStreamSynchronize(ix->stream));  // `padded` and the caller's buffers are free again
    size_t at = 0;
    for(int r = 0; r < W; ++r) { off[ (size_t)r ] = at * row; cnt[ (size_t)r ] = (size_t)meta[ (size_t)r * 2 ] * row; at += (size_t)meta[ (size_t)r * 2 ]; }
    if(!comm->allgatherv_device(vec_base, off.data(), cnt.data(), ix->stream)) { set_err(ix, comm->err); return false; }
    for(int r = 0; r < W; ++r) { off[ (size_t)r ] = off[ (size_t)r ] / row * 8; cnt[ (size_t)r ] = cnt[ (size_t)r ] / row * 8; }
    if(!comm->allgatherv_device(lab_base, off.data(), cnt.data(), ix->stream)) { set_err(ix, comm->err); return false; }
    std::vector<uint64_t> all_labels(total);
    HIPCHK(ix, hipMemcpyAsync(all_labels.data(), lab_base, total * 8, hipMemcpyDeviceToHost, ix->stream));
    if(!sync_stream(ix, comm)) return false;
    if(!pq_encode_rows(ix, first, total)) return false;  // every rank quantises all rows: deterministic, the replicas stay identical
    if(!fill_norms(ix, first, total)) return false;  // every rank over all rows: the replicas stay self-contained
    bool ok = true;
    run_batches(ix, all_labels.data(), s, total, comm, &ok);
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Row-sharded build (SURVEY.md 8e as written; lantern_gpu_add_row_sharded).  The WORK-sharded build above splits the walks of one graph
// and pays two exchanges per batch; this is the other partitioning: candidate generation by ROW SHARD.  Every rank keeps a graph over
// ITS rows only, grown in lock step with the global one; a batch of the global build -- the usual plan (plan_batch), its members drawn
// from all shards in proportion -- goes
//   1. every rank writes its share of the batch's rows into the global vector table and the shares are all-gathered into every rank's
//      HBM (the only time a row crosses the fabric); nothing is linked yet;
//   2. every rank searches ITS graph -- as it stood BEFORE this batch: a batch's members are invisible to each other, as in a one-GPU
//      batch -- for each of the batch's rows (k_search, ef = K answers), the per-rank lists are all-gathered in HBM (12 bytes per
//      candidate) and merged by (distance, slot) into what the level-0 walk of the insertion would have handed on
//      (k_merge_candidates); the upper levels (~1/M of the nodes) are walked in the global graph as usual (k_insert with only_upper:
//      replicated, a few per cent of the walks);
//   3. k_connect, the grouping pass and k_revlink run on every rank as in a one-GPU batch (deterministic: the replicas agree to the
//      bit, lantern_gpu_graph_checksum);
//   4. only then every rank inserts its share of the batch into its own graph (the ordinary device build, no exchange).  (Until round 4
//      the share joined first: in-batch rows then used up part of the K answers per shard and were discarded by the merge.)
//      If this step fails after step 3 committed the batch globally, the build stops with the error; the global index holds the
//      batches that completed and is consistent (searchable, saveable), the shard graph is one batch behind it and is discarded
//      with the call -- the build cannot be resumed, as with any failed collective build.
// A row so chooses among the rows inserted before its batch, as in the batch-synchronous one-GPU build: the early rows' long links
// and the reverse-link pruning are there.  The slots of the result follow the batches (within a batch rank 0's share first), not the
// caller's rank order: labels identify rows.  NOT the one-GPU build's graph edge for edge -- the candidates of a row are the union of
// W approximate searches instead of one: parity is edge for edge against the oracle's restatement of THIS procedure
// (oracle.row_sharded_build) and a recall statement against the one-GPU build (tests/test_gpu_sharded_build.py).
// ---------------------------------------------------------------------------------------------------------------------------
static bool row_shard_candidates(Index *ix, const RowShard &rs, size_t first, size_t b, const uint32_t *d_link_off, uint64_t *d_tops, uint32_t *d_top_count)
{
    Index       *loc = rs.loc;
    Comm        *comm = rs.comm;
    const int    W = comm->world, R = comm->rank;
    const size_t K = rs.K, part = b * K, row = (size_t)ix->chunks * 16;
    char *d_all = (char *)scratch(ix, 10, (size_t)W * part * 12 + 64);  // [W][b][K] labels | [W][b][K] distances
    if(!d_all) return false;
    uint64_t *g_lab = (uint64_t *)d_all;
    float    *g_dist = (float *)(d_all + (size_t)W * part * 8);
    bool ok = true;
    if(loc->n == 0) {
        std::vector<float> inf(part, __builtin_inff());
        ok = hipMemsetAsync(g_lab + (size_t)R * part, 0, part * 8, ix->stream) == hipSuccess &&
             hipMemcpyAsync(g_dist + (size_t)R * part, inf.data(), part * 4, hipMemcpyHostToDevice, ix->stream) == hipSuccess &&
             hipStreamSynchronize(ix->stream) == hipSuccess;  // `inf` is a local
    } else {
        std::lock_guard<std::mutex> gl(loc->mu);
        ok = run_search_device(loc, (const uint4 *)((const char *)ix->d_vec + first * row), b, K, rs.ef, 0, g_lab + (size_t)R * part, g_dist + (size_t)R * part,
                               nullptr, nullptr, nullptr, nullptr, ix->stream, loc->search_waves);
        if(!ok) set_err(ix, "lantern_gpu: row-sharded build, candidate search: " + loc->err);
    }
    if(ok && W > 1) {
        std::vector<size_t> off((size_t)W), cnt((size_t)W);
        for(int r = 0; r < W; ++r) { off[ (size_t)r ] = (size_t)r * part * 8; cnt[ (size_t)r ] = part * 8; }
        ok = comm->allgatherv_device(g_lab, off.data(), cnt.data(), ix->stream);
        for(int r = 0; r < W; ++r) { off[ (size_t)r ] = (size_t)r * part * 4; cnt[ (size_t)r ] = part * 4; }
        ok = ok && comm->allgatherv_device(g_dist, off.data(), cnt.data(), ix->stream);
        if(!ok) set_err(ix, "lantern_gpu: row-sharded build, exchange of the candidate lists: " + comm->err);
    }
    ok = ok && launch_merge_candidates(g_lab, g_dist, (uint32_t)W, (uint32_t)b, (uint32_t)K, (uint32_t)first, d_link_off, ix->M, ix->efc, d_tops, d_top_count,
                                       ix->stream) == hipSuccess;
    if(!ok && ix->err.empty()) set_err(ix, "lantern_gpu: HIP failure during the row-sharded build");
    return ok;
}

// `count` caller rows (scalar kind `kind_in`) into the vector table at slots [at, at + count) in their STORED form -- nothing is linked:
// a row that no list names is unreachable.  Queued on the index stream.
static bool stage_rows_at(Index *ix, size_t at, const void *vectors, size_t count, int kind_in)
{
    if(count == 0) return true;
    const size_t row_words = (size_t)ix->chunks * 4, row = row_words * 4, in_bytes = input_bytes(ix, kind_in);
    char *const  dst = (char *)ix->d_vec + at * row;
    if(kind_in == ix->scalar && in_bytes == row) {
        HIPCHK(ix, hipMemcpyAsync(dst, vectors, count * row, hipMemcpyHostToDevice, ix->stream));
        HIPCHK(ix, hipStreamSynchronize(ix->stream));  // (the caller's buffer is pageable and borrowed)
        return true;
    }
    if(kind_in == usearch_scalar_f32_k && (ix->scalar == usearch_scalar_f16_k || ix->scalar == usearch_scalar_i8_k)) {  // (quant_bits = 1 is refused above)
        const bool up = upload_f32_as_stored(ix, vectors, count, (uint32_t *)dst);
        if(!up) set_err(ix, "lantern_gpu: HIP failure uploading vectors");
        return up;
    }
    std::vector<uint32_t> padded(count * row_words);
    for(size_t i = 0; i < count; ++i) pad_row(ix, (const char *)vectors + i * in_bytes, kind_in, &padded[ i * row_words ]);
    HIPCHK(ix, hipMemcpyAsync(dst, padded.data(), count * row, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(ix, hipStreamSynchronize(ix->stream));  // (`padded` is a local)
    return true;
}

bool add_row_sharded_locked(Index *ix, Comm *comm, const uint64_t *labels, const void *vectors, size_t n_shard, int kind_in)
{
    if(ix->n || !ix->pend_labels.empty()) { set_err(ix, "lantern_gpu: the row-sharded build needs an empty index"); return false; }
    if(ix->pq || ix->b1_from_f32) { set_err(ix, "lantern_gpu: the row-sharded build does not take pq or quant_bits = 1 indexes"); return false; }
    const int    W = comm->world, R = comm->rank;
    const size_t row = (size_t)ix->chunks * 16, M0 = ix->M0, efc = ix->efc, in_bytes = input_bytes(ix, kind_in);
    // ---- shard sizes
    std::vector<uint64_t> sizes((size_t)W, 0);
    std::vector<size_t>   off((size_t)W), cnt((size_t)W);
    sizes[ (size_t)R ] = n_shard;
    for(int r = 0; r < W; ++r) { off[ (size_t)r ] = (size_t)r * 8; cnt[ (size_t)r ] = 8; }
    if(!comm->allgatherv_host(sizes.data(), off.data(), cnt.data())) { set_err(ix, comm->err); return false; }
    std::vector<size_t> lo((size_t)W + 1, 0);
    for(int r = 0; r < W; ++r) lo[ (size_t)r + 1 ] = lo[ (size_t)r ] + (size_t)sizes[ (size_t)r ];
    const size_t N = lo[ (size_t)W ];
    if(N == 0) return true;
    if(N >= 0x7FFFFFFEull) { set_err(ix, "lantern_gpu: capacity above 2^31-1 slots is not supported"); return false; }
    // ---- this rank's own graph; a row's label there is its GLOBAL slot + 1
    usearch_error_t err = nullptr;
    usearch_init_options_t lo_opts = ix->opts;
    lo_opts.retriever = lo_opts.retriever_mut = nullptr;
    lo_opts.retriever_ctx = nullptr;
    Index *loc = (Index *)usearch_init(&lo_opts, nullptr, &err);
    if(!loc) { set_err(ix, std::string("lantern_gpu: row-sharded build: ") + (err ? err : "cannot create the shard's index")); return false; }
    struct Cleanup { Index *a = nullptr; ~Cleanup() { usearch_error_t ig = nullptr; if(a) usearch_free(a, &ig); } } cleanup;
    cleanup.a = loc;
    loc->seed = ix->seed;
    loc->efc = ix->efc;
    loc->add_batch_max = ix->add_batch_max;
    loc->add_min_ratio = ix->add_min_ratio;
    if(n_shard) {
        usearch_reserve(loc, n_shard, &err);
        if(err) { set_err(ix, std::string("lantern_gpu: row-sharded build, shard graph: ") + err); return false; }
    }
    // ---- levels by the usual draw over the global slots, room for everything, the plan of the whole build
    StagedMeta s;
    if(!stage_meta(ix, nullptr, N, s)) return false;
    struct Batch { size_t first, b; };
    std::vector<Batch>  plan;
    std::vector<size_t> share;  // [batch][rank]: how many of the batch's rows come from that rank's shard
    {
        std::vector<size_t> pf, pc;
        row_shard_plan(sizes.data(), W, s.lv.data(), N, ix->add_batch_max, ix->add_min_ratio, pf, pc, share);  // (host_util.hpp)
        for(size_t t = 0; t < pf.size(); ++t) plan.push_back({ pf[ t ], pc[ t ] });
    }
    // ---- labels in slot order (every rank knows where every rank's rows go), levels and upper offsets: one upload
    std::vector<uint64_t> by_rank(N), all_labels(N);
    if(n_shard) std::memcpy(&by_rank[ lo[ (size_t)R ] ], labels, n_shard * 8);
    for(int r = 0; r < W; ++r) { off[ (size_t)r ] = lo[ (size_t)r ] * 8; cnt[ (size_t)r ] = (size_t)sizes[ (size_t)r ] * 8; }
    if(W > 1 && !comm->allgatherv_host(by_rank.data(), off.data(), cnt.data())) { set_err(ix, comm->err); return false; }
    {
        std::vector<size_t> cur(lo.begin(), lo.end() - 1);
        for(size_t t = 0; t < plan.size(); ++t) {
            size_t at = plan[ t ].first;
            for(int r = 0; r < W; ++r)
                for(size_t j = 0; j < share[ t * (size_t)W + (size_t)r ]; ++j) all_labels[ at++ ] = by_rank[ cur[ (size_t)r ]++ ];
        }
    }
    HIPCHK(ix, hipMemcpyAsync(ix->d_labels, all_labels.data(), N * 8, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(ix, hipMemcpyAsync(ix->d_levels, s.l8.data(), N, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(ix, hipMemcpyAsync(ix->d_upper_off, s.uo.data(), N * 4, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(ix, hipStreamSynchronize(ix->stream));
    // every shard answers with more than its expected share of a row's ef_construction nearest (2x; all of them for one or two ranks),
    // found with that expansion; LANTERN_GPU_ROW_SHARD_K / _EF override (tuning)
    RowShard rs;
    rs.loc = loc;
    rs.comm = comm;
    rs.K = std::min<size_t>(efc, std::max<size_t>(M0 + 1, 2 * efc / (size_t)W));
    rs.ef = rs.K;  // (measured, 200k x 768, 4 and 8 shards: recall is the same to the fourth digit from ef = ef_construction down to 33)
    if(const char *ke = std::getenv("LANTERN_GPU_ROW_SHARD_K")) rs.K = std::min<size_t>(efc, std::max<size_t>(1, (size_t)std::atoi(ke)));  // tuning
    if(const char *ee = std::getenv("LANTERN_GPU_ROW_SHARD_EF")) rs.ef = std::min<size_t>(efc, std::max<size_t>(rs.K, (size_t)std::atoi(ee)));
    // ---- the batches
    size_t mine = 0;  // rows of this rank's shard handed out so far
    bool   ok = true;
    std::vector<uint64_t> gl;
    // (a failure leaves the index with the batches that completed: the host mirrors below describe exactly those)
    auto batch = [&](size_t t) -> bool {
        const size_t first = plan[ t ].first, b = plan[ t ].b, *sh = &share[ t * (size_t)W ];
        size_t my_off = 0;
        for(int r = 0; r < R; ++r) my_off += sh[ r ];
        const size_t my_n = sh[ R ];
        // 1. this rank's share goes into the GLOBAL table (stored form); the shards' shares follow by all-gather
        if(my_n && !stage_rows_at(ix, first + my_off, (const char *)vectors + mine * in_bytes, my_n, kind_in)) return false;
        if(W > 1) {
            size_t at = 0;
            for(int r = 0; r < W; ++r) { off[ (size_t)r ] = at * row; cnt[ (size_t)r ] = sh[ r ] * row; at += sh[ r ]; }
            if(!comm->allgatherv_device((char *)ix->d_vec + first * row, off.data(), cnt.data(), ix->stream)) { set_err(ix, comm->err); return false; }
        }
        if(!fill_norms(ix, first, b)) return false;
        // 2. + 3. candidates from every shard's graph AS IT STOOD BEFORE THIS BATCH (a batch's members are invisible to each other, as
        // in a one-GPU batch: every one of a shard's K answers is usable), selection, reverse links
        if(ix->n == 0) {  // "Do nothing for the first element": it only becomes the entry point
            ix->n = 1;
            ix->entry = 0;
            ix->max_level = s.lv[ 0 ];
            ix->c_add_vectors += 1;
        } else if(!run_batch(ix, b, s.lv.data() + first, nullptr, &rs)) {
            return false;
        }
        // 4. only now this rank's share joins its own graph (until round 4 it joined first: in-batch rows then used up part of the K
        // answers per shard and were discarded by the merge -- a row could be left with far fewer than ef_construction candidates)
        if(my_n) {
            gl.resize(my_n);
            for(size_t i = 0; i < my_n; ++i) gl[ i ] = (uint64_t)(first + my_off + i) + 1;
            lantern_gpu_add_many(loc, gl.data(), (const char *)vectors + mine * in_bytes, my_n, (usearch_scalar_kind_t)kind_in, &err);
            if(!err) lantern_gpu_flush(loc, &err);
            if(err) { set_err(ix, std::string("lantern_gpu: row-sharded build, shard graph: ") + err); return false; }
            mine += my_n;
        }
        // the host transport's staging buffers and the candidate scratch are reused by the next batch
        return sync_stream(ix, comm);
    };
    for(size_t t = 0; t < plan.size() && ok; ++t) ok = batch(t);
    if(ix->profiling) prof_resolve(ix, 0);
    const size_t done = ix->n;
    ix->labels.assign(all_labels.begin(), all_labels.begin() + (ptrdiff_t)done);
    ix->levels.assign(s.l8.begin(), s.l8.begin() + (ptrdiff_t)done);
    ix->upper_off.assign(s.uo.begin(), s.uo.begin() + (ptrdiff_t)done);
    ix->upper_blocks = 0;
    for(size_t i = 0; i < done; ++i) ix->upper_blocks += (size_t)s.lv[ i ];
    return ok;
}

bool flush_locked(Index *ix)
{
    const size_t pending = ix->pend_labels.size();
    if(pending == 0) return true;
    const size_t row_words = (size_t)ix->chunks * 4;
    bool         ok = true;
    const size_t pi = insert_rows(ix, ix->pend_labels.data(), ix->pend_levels.data(), ix->pend_rows.data(), pending, &ok);
    ix->pend_labels.erase(ix->pend_labels.begin(), ix->pend_labels.begin() + (ptrdiff_t)pi);
    ix->pend_levels.erase(ix->pend_levels.begin(), ix->pend_levels.begin() + (ptrdiff_t)pi);
    ix->pend_rows.erase(ix->pend_rows.begin(), ix->pend_rows.begin() + (ptrdiff_t)(pi * row_words));
    if(ix->pend_labels.empty()) { std::vector<uint32_t>().swap(ix->pend_rows); }
    return ok;
}

// ---------------------------------------------------------------------------------------------------
// search
// ---------------------------------------------------------------------------------------------------

bool run_search_device(Index *ix, const uint4 *d_queries, size_t nq, size_t k, size_t ef, size_t skip, uint64_t *d_labels,
                       float *d_dists, uint32_t *d_slots, uint32_t *d_counts, uint64_t *d_D, uint64_t *d_E, hipStream_t stream,
                       int waves, uint32_t *done, uint32_t *done_flags)
{
    if(nq == 0 || k == 0) return true;
    size_t expansion = ef ? ef : ix->ef;
    if(expansion < k + skip) expansion = k + skip;  // usearch: expansion = max(expansion, wanted)
    // ---- a compact pq index whose subvectors are whole 16-byte chunks: the f32 walk below over rows DECODED ON THE FLY from the
    // L2-resident centroid tables (device_common.hpp PqdRow) -- the arithmetic, and so every bit of every answer, of the expanded
    // form of the same index.  (LANTERN_GPU_PQ_ADC=1, or subvectors of another width: the table walk of search_adc_kernel.hip.)
    const char *const adc_e = std::getenv("LANTERN_GPU_PQ_ADC");
    const bool        pq_adc_env = adc_e && std::atoi(adc_e) != 0;
    const bool pqd = ix->pq_compact && !pq_adc_env && ix->pqd_inv != 0;
    if(ix->pq_compact && !pqd) {
        // ---- a compact pq index: ADC over the code rows (search_adc_kernel.hip).  One 8-wave workgroup per query; the per-query
        // table takes num_subvectors16 x 256 floats of LDS (98 KB at 96 subvectors: one workgroup per CU, three at 32).
        const uint32_t code_chunks = ix->pq_S16 / 16;
        uint32_t       vis_slots = 2048;
        if(ix->search_vis_slots >= 0) vis_slots = (uint32_t)ix->search_vis_slots / 4 * 4;
        while(vis_slots && search_adc_lds_bytes(code_chunks, ix->chunks, (uint32_t)expansion, ix->M0, vis_slots) > 150 * 1024) vis_slots = vis_slots > 256 ? vis_slots - 256 : 0;
        if(vis_slots && vis_slots < 4 * ix->M0) vis_slots = 0;
        size_t lds = search_adc_lds_bytes(code_chunks, ix->chunks, (uint32_t)expansion, ix->M0, vis_slots);
        if(lds > 160 * 1024) { set_err(ix, "lantern_gpu: ef/k exceed the 160 KiB LDS budget of the ADC search kernel"); return false; }
        int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / lds));
        // A table that leaves room for one workgroup per CU anyway (96 subvectors x 256 centroids): every query walks alone on
        // its CU, so it runs the walk that is fastest alone -- walk_spec.hpp's lone-query shape, 3 role + 8 row waves
        // (LANTERN_GPU_ADC_SPEC=0|1 overrides; an explicit wave count selects the classic kernel as for the f32 walk).
        bool adc_spec = false;
        const uint32_t adc_prefetch = ix->M0 % 4 == 0 && ix->M0 <= 32 ? 1u : 0u, adc_cache = adc_prefetch ? 128u : 0u;  // (8-lane groups: four list words per lane)
        if(waves <= 0 && ix->M0 >= 2 && ix->M0 <= 64 && expansion <= 128 && !lds_list_env()) {
            const size_t with_spec = lds + search_spec_lds_bytes(ix->M0, adc_prefetch, adc_cache);
            const char  *se = std::getenv("LANTERN_GPU_ADC_SPEC");
            adc_spec = with_spec <= 160 * 1024 && (se ? std::atoi(se) != 0 : (per_cu == 1 || nq <= (size_t)ix->num_cus * 2));  // (small batches: as the f32 walk)
            if(adc_spec) {
                lds = with_spec;
                per_cu = 1;
            }
        }
        // (rows of at most 8 chunks: a row wave holds eight 8-lane groups, so FOUR row waves cover a 32-entry list in one pass --
        // measured, LANTERN_GPU_SPEC_WAVES=7: 1.84 M queries/s against 1.96 M with eight: the shorter row waves matter more)
        int aw = adc_spec ? 11 : waves > 0 ? std::min(waves, 8) : 8;
        if(adc_spec) {
            if(const char *sw = std::getenv("LANTERN_GPU_SPEC_WAVES")) {
                const int w = std::atoi(sw);
                if(w >= 4 && w <= 11) aw = w;
            }
        }
        size_t     g = (size_t)ix->num_cus * (size_t)per_cu;
        if(ix->search_max_wg > 0) g = (size_t)ix->search_max_wg;
        const int grid = (int)std::max<size_t>(1, std::min(g, nq));
        const int slot = acquire_search_slot(ix, stream, (size_t)grid);
        if(slot < 0) return false;
        SearchArgs a{};
        a.view = ix->view();
        a.view.vec = (const uint4 *)ix->d_codes16;
        a.view.chunks = code_chunks;
        a.queries = d_queries;
        a.nq = (uint32_t)nq;
        a.k = (uint32_t)k;
        a.ef = (uint32_t)expansion;
        a.skip = (uint32_t)skip;
        a.labels = ix->d_labels;
        a.out_labels = d_labels;
        a.out_dists = d_dists;
        a.out_slots = d_slots;
        a.out_counts = d_counts;
        a.out_D = d_D;
        a.out_E = d_E;
        a.bitmaps = ix->slot_bitmaps[ slot ];
        a.bm_words = (uint32_t)ix->slot_words[ slot ];
        a.undo_cap = vis_undo_cap();
        a.vis_slots = vis_slots;
        a.totals = ix->d_totals;
        a.ticket = next_ticket(ix, nq, grid, stream);
        a.done = done;
        a.done_flags = done_flags;
        a.lds_list = lds_list_env();
        a.adc_centers = ix->d_centers;
        a.adc_S = ix->pq_S;
        a.adc_C = ix->pq_C;
        a.adc_subdim = ix->pq_subdim;
        a.adc_qchunks = ix->chunks;
        a.spec = adc_spec ? 2 : 0;
        a.spec_prefetch = adc_spec ? adc_prefetch : 0;
        a.spec_cache = adc_spec ? adc_cache : 0;
        HIPCHK(ix, launch_search_adc(ix->metric + M_ADC, a, aw, grid, stream));
        if(done) ix->slot_pending[ slot ] = false;
        else if(!release_search_slot(ix, slot, stream)) return false;
        ix->c_search_queries += nq;
        return true;
    }
    // Launch shape.  Batches that fill the chip: four waves per query, six workgroups per CU -- the walk is HBM-bound and its
    // serial phases hide behind other walks' row loads.  Batches that cannot (and the lone query): the latency-bound walk of
    // walk_spec.hpp -- one barrier per hop, speculative row loads, neighbour lists fetched with the rows:
    //   spec 2: at most one query per CU: three role waves + eight row waves per query (a whole list in one pass);
    //   spec 1: up to four four-wave workgroups per CU (on request: LANTERN_GPU_SPEC=1).
    // An explicit wave count (lantern_gpu_set_search_shape; tests, tuning) selects the classic kernel; LANTERN_GPU_SPEC=0|1|2
    // overrides the automatic choice.
    int spec = 0;
    if(waves <= 0) {
        const char *se = std::getenv("LANTERN_GPU_SPEC");
        const bool  can = ix->M0 >= 2 && ix->M0 <= 64 && expansion <= 128 && !ix->phase_profile && !lds_list_env();
        if(can) {
            // (measured, 1M x 768 cosine, one 1024-query batch: the four-wave latency-bound shape 796 k QPS, the classic kernel
            // 819 k -- with every walk of the batch resident the row loads saturate HBM for most of the launch and the speculative
            // rows cost bandwidth; so spec 1 is chosen only on request, spec 2 up to two queries per CU -- one workgroup per CU, the
            // second query after the first: two workgroups side by side measure the same, 631 vs 627 k at 512 queries)
            if(se) spec = std::atoi(se);
            else if(nq <= (size_t)ix->num_cus * 2) spec = 2;  // (1M x 768 cosine: 384 queries 528 k vs 389 k, 512: 624 k vs 500 k, 768: 658 k vs 691 k)
            if(spec < 0 || spec > 4) spec = 0;
#if !LGPU_EXPERIMENTAL
            if(spec >= 3) spec = 2;  // the variants behind 3 / 4 are not in this library (LANTERN_BUILD_EXPERIMENTAL=1 builds them)
#else
            // 4: the ONE-WAVE walk (walk_solo.hpp): no barrier, no hand-over between waves -- f32 l2sq / cos rows of < 64 chunks,
            // M <= 16, ef <= 64, an index whose visited bitmap fits LDS.  ON REQUEST ONLY (LANTERN_GPU_SPEC=4, or LANTERN_GPU_SOLO=1 for
            // every launch it applies to): measured in round 5 on the lone 100k x 128 query it is SLOWER than the 3 + 8 wave shape --
            // 117.7 us against 102.0 us per query on one box (its first form: 152.6 against 106.0), 1.70 against 1.47 us per hop -- because
            // one wave has to issue all ~540 instructions of a hop itself; parity-green in every regime it takes.
            // Anything it does not take falls back to spec 2.
            static const bool solo_auto = std::getenv("LANTERN_GPU_SOLO") && std::atoi(std::getenv("LANTERN_GPU_SOLO")) != 0;
            if(spec == 2 && !se && solo_auto && nq <= (size_t)ix->num_cus) spec = 4;
            if(spec == 4) {
                const size_t words = ((std::max<size_t>(ix->n, 1) + 31) / 32 + 3) & ~(size_t)3;
                uint32_t     ne_log2 = 9;
                while(ne_log2 > 5 && search_solo_lds_bytes(ne_log2, (uint32_t)words) > 160 * 1024) --ne_log2;
                // (the instrumented instantiation -- lantern_gpu_spec_profile -- exists for f32 l2sq rows of exactly 32 chunks)
                const bool prof_ok = !ix->spec_profile || (ix->mcode == M_L2SQ && ix->chunks == 32);
                if(pqd || !prof_ok || !search_solo_supported(ix->mcode, ix->chunks, ix->M, ix->M0, (uint32_t)expansion) ||
                   search_solo_lds_bytes(ne_log2, (uint32_t)words) > 160 * 1024)
                    spec = 2;
                else {
                    const size_t lds = search_solo_lds_bytes(ne_log2, (uint32_t)words);
                    const size_t per_cu = std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / lds));
                    size_t       gmax = (size_t)ix->num_cus * per_cu;
                    if(ix->search_max_wg > 0) gmax = (size_t)ix->search_max_wg;
                    const int grid = (int)std::max<size_t>(1, std::min(gmax, nq));
                    const int slot = acquire_search_slot(ix, stream, (size_t)grid);
                    if(slot < 0) return false;
                    SearchArgs a{};
                    a.view = ix->view();
                    a.queries = d_queries;
                    a.nq = (uint32_t)nq;
                    a.k = (uint32_t)k;
                    a.ef = (uint32_t)expansion;
                    a.skip = (uint32_t)skip;
                    a.labels = ix->d_labels;
                    a.out_labels = d_labels;
                    a.out_dists = d_dists;
                    a.out_slots = d_slots;
                    a.out_counts = d_counts;
                    a.out_D = d_D;
                    a.out_E = d_E;
                    a.vis_slots = (uint32_t)words;   // the LDS bitmap (nothing of the slot's HBM slab is touched)
                    a.spec_cache = ne_log2;
                    a.totals = ix->d_totals;
                    a.ticket = next_ticket(ix, nq, grid, stream);
                    a.done = done;
                    a.done_flags = done_flags;
                    a.spec = 4;
                    a.phase_cycles = ix->spec_profile ? ix->d_totals + 16 : nullptr;
                    HIPCHK(ix, launch_search_solo(ix->mcode, a, grid, stream));
                    ix->c_solo_launches += 1;
                    if(done) ix->slot_pending[ slot ] = false;
                    else if(!release_search_slot(ix, slot, stream)) return false;
                    ix->c_search_queries += nq;
                    return true;
                }
            }
            // (3: two nodes per round, the second speculative -- walk_twin.hpp; on request only: measured slower)
            if(spec == 3 && !(expansion <= 64 && (ix->mcode == M_L2SQ || ix->mcode == M_COS) && (group_lanes_for(ix->chunks) == 64 || (ix->mcode == M_L2SQ && group_lanes_for(ix->chunks) == 16))))
                spec = 2;
#endif
        }
        // (measured, classic kernel, 1M x 768 cosine, 1024 queries: 4 waves 693 k QPS, 6 waves 525 k, 8 waves 594 k -- more waves
        // only make its serial phases costlier; so four waves per query whatever the batch size)
        waves = spec >= 2 ? 11 : spec == 1 ? 4 : waves < 0 ? -waves : 4;  // (a negative count: the caller's classic fallback)
        if(spec) {  // tuning: LANTERN_GPU_SPEC_WAVES = waves per query of the latency-bound shapes (spec 1: 2..8; spec 2: 4..11)
            if(const char *sw = std::getenv("LANTERN_GPU_SPEC_WAVES")) {
                const int w = std::atoi(sw);
                if(w >= (spec >= 2 ? 4 : 2) && w <= (spec >= 2 ? 11 : 8)) waves = w;
            }
        }
    }
    // list prefetch of the latency-bound walk: every lane of a row's group fetches LW words of the row's own list
    const int      G_ = group_lanes_for(ix->chunks), LW_ = G_ >= 32 ? 1 : G_ == 16 ? 2 : 4;
    const uint32_t spec_prefetch = spec && ix->M0 % (uint32_t)LW_ == 0 && ix->M0 <= (uint32_t)(G_ * LW_) ? 1u : 0u;
    const uint32_t spec_cache = !spec_prefetch ? 0u : spec >= 2 ? 128u : 64u;
    const size_t   spec_lds = spec ? search_spec_lds_bytes(ix->M0, spec_prefetch, spec_cache, spec == 3) : 0;
    // LDS visited set: sized for ~3x the planner's estimate of visited nodes per query (hnsw.c:89-132 puts it at
    // about 2 M ef S with S ~ 3), capped so that SIX workgroups fit on a CU (more walks in flight beat a roomier
    // set: 1.106 -> 1.17 M QPS at 1M x 768) -- four for the four-wave latency-bound shape, one for the lone-query shape;
    // it spills to the bitmap beyond
    const size_t lds_budget = spec >= 2 ? 96 * 1024 : spec == 1 ? 39 * 1024 : 26 * 1024;
    uint32_t vis_slots = 1024;
    while(vis_slots < 8192 && vis_slots / 4 * 3 < expansion * ix->M0 * 2) vis_slots <<= 1;
    if(ix->search_vis_slots >= 0) vis_slots = (uint32_t)ix->search_vis_slots / 4 * 4;
    while(vis_slots && search_lds_bytes(ix->chunks, (uint32_t)expansion, ix->M0, vis_slots) + spec_lds > lds_budget)
        vis_slots = vis_slots > 256 ? vis_slots - 256 : 0;
    if(vis_slots && vis_slots < 4 * ix->M0) vis_slots = 0;
    if(search_lds_bytes(ix->chunks, (uint32_t)expansion, ix->M0, vis_slots) + spec_lds > 160 * 1024) {
        set_err(ix, "lantern_gpu: ef/k exceed the 160 KiB LDS budget of the search kernel");
        return false;
    }
    const int grid = search_grid(ix, nq, waves, spec >= 2 ? waves : spec == 1 ? 16 : 24);
    ix->last_search_grid = grid;
    const int slot = acquire_search_slot(ix, stream, (size_t)grid);
    if(slot < 0) return false;
    SearchArgs a{};
    a.view = ix->view();
    if(pqd) {
        a.view.vec = (const uint4 *)ix->d_codes16;
        a.view.pq_centers = (const uint4 *)ix->d_centers;
        a.view.pq_cps = ix->pq_subdim / 4;
        a.view.pq_C = ix->pq_C;
        a.view.pq_inv = ix->pqd_inv;
        a.view.pq_row_bytes = ix->pq_S16;
    }
    a.queries = d_queries;
    a.nq = (uint32_t)nq;
    a.k = (uint32_t)k;
    a.ef = (uint32_t)expansion;
    a.skip = (uint32_t)skip;
    a.labels = ix->d_labels;
    a.out_labels = d_labels;
    a.out_dists = d_dists;
    a.out_slots = d_slots;
    a.out_counts = d_counts;
    a.out_D = d_D;
    a.out_E = d_E;
    a.bitmaps = ix->slot_bitmaps[ slot ];
    a.bm_words = (uint32_t)ix->slot_words[ slot ];
    a.undo_cap = vis_undo_cap();
    a.vis_slots = vis_slots;
    a.totals = ix->d_totals;
    a.ticket = next_ticket(ix, nq, grid, stream);
    // (there is no instrumented instantiation of the decoding walk: a compact pq launch ignores phase_profile)
    const bool prof_walk = ix->phase_profile && !pqd;
    a.phase_cycles = spec ? (ix->spec_profile ? ix->d_totals + 16 : nullptr) : prof_walk ? ix->d_totals + 8 : nullptr;
    // the row bitmap only when unique-rows mode asked for it AND it covers every slot the walk can name (a reserve / add since
    // it was sized would otherwise let mark_touched write past it)
    a.touched = (!spec && prof_walk && ix->unique_rows_on && ix->d_touched && ix->touched_words * 32 >= ix->cap) ? ix->d_touched : nullptr;
    if(!spec && prof_walk && ix->trace_on && ix->d_trace && nq <= ix->trace_nq) {  // (lantern_gpu_search_row_trace: the launch's own counts start at zero)
        HIPCHK(ix, hipMemsetAsync(ix->d_trace_count, 0, nq * 4, stream));
        a.trace = ix->d_trace;
        a.trace_count = ix->d_trace_count;
        a.trace_cap = (uint32_t)ix->trace_cap;
    }
    a.done = done;
    a.done_flags = done_flags;
    a.lds_list = lds_list_env();
    // small batch (at most four 4-wave workgroups per CU would be resident anyway): four rows in flight per group
    static const int wide_env = std::getenv("LANTERN_GPU_WIDE_ROWS") ? std::atoi(std::getenv("LANTERN_GPU_WIDE_ROWS")) : -1;
    a.wide_rows = spec ? 0 : wide_env >= 0 ? wide_env : (nq * (size_t)waves <= (size_t)ix->num_cus * 16 && nq >= 64);
    a.spec = spec;
    a.spec_prefetch = spec_prefetch;
    a.spec_cache = spec_cache;
    HIPCHK(ix, launch_search(pqd ? ix->metric + M_PQD : ix->mcode, a, waves, grid, stream));
    if(done) ix->slot_pending[ slot ] = false;  // the caller waits for the kernel itself: nothing to order later launches against
    else if(!release_search_slot(ix, slot, stream)) return false;
    ix->c_search_queries += nq;
    return true;
}

// One usearch_search_ef (scan.c:220-228, :273-281) on behalf of one scan.  The query row and the answer live in ONE
// pinned, device-mapped block: the kernel reads the row over the host link and writes labels | distances | slots |
// count straight back, so the call is one launch + one stream synchronisation (no H2D / D2H copy commands, whose
// queueing latency is of the order of the walk itself for a lone query).
size_t search_one_locked(Index *ix, Cursor *cur, const void *query, int kind, size_t k, size_t ef, bool streaming, uint64_t *labels,
                         float *distances)
{
    if(!streaming) cur->seen.clear();
    if(ix->n == 0 || k == 0) return 0;
    const size_t want = std::min(cur->seen.size() + k, ix->n);  // enough to find k unseen ones
    const size_t row = (size_t)ix->chunks * 16;
    const size_t need = row + want * 16 + 32;
    if(ix->h_single_bytes < need) {
        if(ix->h_single) (void)hipHostFree(ix->h_single);
        ix->h_single = ix->h_single_dev = nullptr;
        ix->h_single_bytes = 0;
        const size_t cap = need * 2 + 4096;
        if(hipHostMalloc((void **)&ix->h_single, cap, hipHostMallocMapped) != hipSuccess) {
            set_err(ix, "lantern_gpu: cannot allocate the pinned single-query block");
            return 0;
        }
        if(hipHostGetDevicePointer((void **)&ix->h_single_dev, ix->h_single, 0) != hipSuccess) {
            set_err(ix, "lantern_gpu: the pinned block is not device-mapped");
            return 0;
        }
        ix->h_single_bytes = cap;
    }
    char *dev = ix->h_single_dev;
    pad_row(ix, query, kind, (uint32_t *)ix->h_single);
    uint64_t *d_lab = (uint64_t *)(dev + row);
    float    *d_dist = (float *)(dev + row + want * 8);
    uint32_t *d_slot = (uint32_t *)(dev + row + want * 12);
    uint32_t *d_cnt = (uint32_t *)(dev + row + want * 16);
    // completion: the kernel bumps a counter in this block after its answers (system-scope release); the host spins on it.
    // No event, no stream synchronisation: of a ~200 us call those cost ~5 us.  The stream is consulted only as a watchdog
    // (a kernel that died leaves the counter at 0 and the stream idle or in error).
    const size_t       flag_off = (row + want * 16 + 4 + 15) / 16 * 16;
    volatile uint32_t *h_done = (volatile uint32_t *)(ix->h_single + flag_off);
    *h_done = 0;
    // one query: the lone-query shape (three role waves + eight row waves: walk_spec.hpp); an explicit wave count
    // (lantern_gpu_set_search_shape) or a list beyond its limits selects the classic eight-wave workgroup
    bool ok = run_search_device(ix, (const uint4 *)dev, 1, want, ef, 0, d_lab, d_dist, d_slot, d_cnt, nullptr, nullptr, ix->stream,
                                ix->search_waves > 0 ? 8 : -8, (uint32_t *)(dev + flag_off));
    if(ok) {
        for(unsigned spins = 0; *h_done == 0; ++spins) {
            cpu_relax();
            if((spins & 0x3FFF) == 0x3FFF) {  // every ~100 us: is the kernel still there?
                const hipError_t st = hipStreamQuery(ix->stream);
                if(st == hipErrorNotReady) continue;
                if(*h_done != 0) break;
                ok = false;  // the stream drained (or failed) without the counter moving
                if(st != hipSuccess) set_err(ix, std::string("lantern_gpu: ") + hipGetErrorString(st));
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if(!ok) {
        if(ix->err.empty()) set_err(ix, "lantern_gpu: HIP failure during search");
        return 0;
    }
    ix->err.clear();
    const char     *host = ix->h_single + row;
    const uint64_t *h_lab = (const uint64_t *)host;
    const float    *h_dist = (const float *)(host + want * 8);
    const uint32_t *h_slot = (const uint32_t *)(host + want * 12);
    uint32_t        got;
    std::memcpy(&got, host + want * 16, 4);
    size_t out = 0;
    for(uint32_t i = 0; i < got && out < k; ++i) {
        if(!cur->seen.insert(h_slot[ i ]).second) continue;  // returned by an earlier call of this scan
        labels[ out ] = h_lab[ i ];
        distances[ out ] = h_dist[ i ];
        ++out;
    }
    return out;
}

bool import_graph_locked(Index *ix, size_t size, const void *vectors, const uint64_t *labels, const uint8_t *levels,
                         const uint32_t *nbr0, const uint32_t *upper_off, const uint32_t *upper_nbr, uint32_t entry_slot,
                         int32_t max_level, bool vectors_are_codes)
{
    if(ix->n || !ix->pend_labels.empty()) { set_err(ix, "lantern_gpu: import needs an empty index"); return false; }
    if(size == 0) return true;
    if(!pq_expand_locked(ix)) return false;  // an empty index that was compacted has no row block: the import writes rows
    size_t blocks = 0;
    for(size_t i = 0; i < size; ++i) blocks += levels[ i ];
    if(!reserve_locked(ix, std::max(size, ix->cap)) || !reserve_upper(ix, blocks)) return false;
    if(ix->pq && vectors_are_codes) {  // a file / the pages carry the codes: the rows are their decodings
        HIPCHK(ix, hipMemcpy(ix->d_codes, vectors, size * (size_t)ix->pq_S, hipMemcpyHostToDevice));
        if(!pq_decode_rows(ix, 0, size)) return false;
    } else if(ix->chunks * 4 == ix->words) {
        HIPCHK(ix, hipMemcpy(ix->d_vec, vectors, size * (size_t)ix->words * 4, hipMemcpyHostToDevice));
    } else {
        HIPCHK(ix, hipMemset(ix->d_vec, 0, size * (size_t)ix->chunks * 16));
        HIPCHK(ix, hipMemcpy2D(ix->d_vec, (size_t)ix->chunks * 16, vectors, (size_t)ix->words * 4, (size_t)ix->words * 4, size,
                               hipMemcpyHostToDevice));
    }
    if(ix->pq && !vectors_are_codes && !pq_encode_rows(ix, 0, size)) return false;  // raw f32 rows: quantise them
    if(!fill_norms(ix, 0, size)) return false;
    ix->labels.resize(size);
    for(size_t i = 0; i < size; ++i) ix->labels[ i ] = labels ? labels[ i ] : (uint64_t)i;
    ix->levels.assign(levels, levels + size);
    ix->upper_off.assign(upper_off, upper_off + size);
    HIPCHK(ix, hipMemcpy(ix->d_labels, ix->labels.data(), size * 8, hipMemcpyHostToDevice));
    HIPCHK(ix, hipMemcpy(ix->d_levels, levels, size, hipMemcpyHostToDevice));
    HIPCHK(ix, hipMemcpy(ix->d_nbr0, nbr0, size * ix->M0 * 4, hipMemcpyHostToDevice));
    HIPCHK(ix, hipMemcpy(ix->d_upper_off, upper_off, size * 4, hipMemcpyHostToDevice));
    if(blocks) HIPCHK(ix, hipMemcpy(ix->d_upper_nbr, upper_nbr, blocks * ix->M * 4, hipMemcpyHostToDevice));
    ix->n = size;
    ix->upper_blocks = blocks;
    ix->radius_stale = true;  // lists from outside: no re-prune state
    ix->entry = entry_slot;
    ix->max_level = max_level;
    // a pq index that was loaded or mirrored (not built here) is read-mostly: LANTERN_GPU_PQ_COMPACT=1 drops its decodings at once
    if(ix->pq) {
        const char *pc = std::getenv("LANTERN_GPU_PQ_COMPACT");
        if(pc && std::atoi(pc) != 0 && !pq_compact_locked(ix)) return false;
    }
    return true;
}

}  // namespace lgpu

// =====================================================================================================
// C ABI
// =====================================================================================================
using namespace lgpu;

#define CLEAR(e) do { if(e) *(e) = nullptr; } while(0)
#define FAIL(e, msg) do { if(e) *(e) = (msg); } while(0)

static Index *H(usearch_index_t h, usearch_error_t *e)
{
    if(!h) { FAIL(e, "lantern_gpu: null index handle"); return nullptr; }
    if(((const Index *)h)->magic != kIndexMagic) { FAIL(e, "lantern_gpu: not an index handle (stale, freed or foreign pointer)"); return nullptr; }
    // HIP's current device is per host thread: an index lives on the device it was created on, whichever thread calls
    // (one thread per GPU is how a single process drives a node: lantern_gpu_comm_init_local)
    (void)hipSetDevice(((Index *)h)->device);
    return (Index *)h;
}

extern "C" {

const char *lantern_gpu_version(void) { return LGPU_EXPERIMENTAL ? "lantern_gpu 0.1 (gfx950) +experimental" : "lantern_gpu 0.1 (gfx950)"; }

int lantern_gpu_device_count(void)
try {
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
LANTERN_ABI_CATCH(nullptr)

usearch_index_t usearch_init(usearch_init_options_t *o, float *pq_codebook, usearch_error_t *e)
try {
    CLEAR(e);
    if(!o) { FAIL(e, "lantern_gpu: null init options"); return nullptr; }
    if(o->metric != nullptr) { FAIL(e, "lantern_gpu: custom metric functions are not supported"); return nullptr; }
    if(o->pq != (pq_codebook != nullptr)) { FAIL(e, "lantern_gpu: pq = true needs a codebook, and a codebook needs pq = true"); return nullptr; }
    if(o->metric_kind != usearch_metric_cos_k && o->metric_kind != usearch_metric_l2sq_k &&
       o->metric_kind != usearch_metric_hamming_k) {
        FAIL(e, "lantern_gpu: unsupported metric kind (expected cos, l2sq or hamming)");  // options.c:119-127
        return nullptr;
    }
    if(o->dimensions == 0) { FAIL(e, "lantern_gpu: dimensions must be positive"); return nullptr; }
    if(o->connectivity < 2 || o->connectivity > 128) { FAIL(e, "lantern_gpu: connectivity (M) must be in [2, 128]"); return nullptr; }  // options.c:165-179
    const bool ham = o->metric_kind == usearch_metric_hamming_k;
    if(ham && o->quantization != usearch_scalar_b1_k) { FAIL(e, "lantern_gpu: hamming needs b1 scalars"); return nullptr; }
    // quant_bits = 1 on real[] (options.c:154-155 returns b1 for ANY metric): bits = (x > 0).  The metrics over b1 storage are
    // the f32 metrics over the {0, 1} values the bits stand for: sum (a - b)^2 is exactly the Hamming distance (an l2sq index
    // runs on the Hamming kernels), the cosine is 1 - |a & b| / (sqrt |a| sqrt |b|) (M_COS_B1: three popcounts).  What the fork
    // computes for cosine cannot be read off the tree (upstream usearch has no such metric): that half is PARITY UNPINNED.
    const bool b1f = !ham && o->quantization == usearch_scalar_b1_k;
    if(!ham && !b1f && o->quantization != usearch_scalar_f32_k && o->quantization != usearch_scalar_f16_k && o->quantization != usearch_scalar_i8_k) {
        FAIL(e, "lantern_gpu: cos/l2sq indexes take f32, f16, i8 or b1 storage (quant_bits=32, 16, 8 or 1)");  // options.c:137-158
        return nullptr;
    }
    if(o->pq) {  // build.c:497-500, scan.c:75-81, pqtable.c:194-240
        if(ham || o->quantization != usearch_scalar_f32_k) { FAIL(e, "lantern_gpu: pq indexes are cos / l2sq over f32 vectors"); return nullptr; }
        if(o->num_centroids < 1 || o->num_centroids > 256) { FAIL(e, "lantern_gpu: num_centroids must be in [1, 256]"); return nullptr; }
        if(o->num_subvectors < 1 || o->num_subvectors > o->dimensions || o->dimensions % o->num_subvectors != 0) {
            FAIL(e, "lantern_gpu: num_subvectors must divide the dimensions");
            return nullptr;
        }
    }
    if(lantern_gpu_device_count() <= 0) { FAIL(e, kNoDevice); return nullptr; }

    Index *ix = new Index();
    ix->opts = *o;
    ix->metric = (int)o->metric_kind;
    ix->scalar = (int)o->quantization;
    const bool f16 = o->quantization == usearch_scalar_f16_k, i8 = o->quantization == usearch_scalar_i8_k;
    ix->b1_from_f32 = b1f;
    ix->mcode = b1f ? (ix->metric == (int)usearch_metric_cos_k ? M_COS_B1 : M_HAMMING) : ix->metric + (f16 ? M_F16 : i8 ? M_I8 : 0);
    ix->words = (ham || b1f) ? (uint32_t)((o->dimensions + 31) / 32)
                : f16 ? (uint32_t)((o->dimensions + 1) / 2)
                : i8 ? (uint32_t)((o->dimensions + 3) / 4)
                     : (uint32_t)o->dimensions;
    ix->chunks = ix->natural_chunks = (ix->words + 3) / 4;
    // Bit rows of 65 .. 127 bytes (768 bits = 96 bytes: quant_bits = 1 on 768-d, hamming over 24 words) are stored at a 128-BYTE STRIDE,
    // zero padded: the fabric fetches 128-byte lines, and a 96-byte row at a 96-byte stride straddles two of them three times in four.
    // Measured in round 5 (1M rows, 8192-query launches): 3.23 GB of fabric traffic per launch for 96-byte rows against 2.22 GB for
    // 128-byte rows, at the same 6.4 M queries/s -- so the padding costs a third more HBM for the rows and takes a third off the
    // traffic.  Zero words change no popcount: every distance, id and counter is the same.  (lantern_gpu_row_bytes tells a caller
    // that keeps queries in device memory the stride; rows still enter and leave the ABI at their own length.)
    if((ham || b1f) && ix->chunks > 4 && ix->chunks < 8) ix->chunks = 8;
    ix->M = (uint32_t)o->connectivity;
    ix->M0 = 2 * ix->M;  // validate_index.c:140-151
    ix->efc = o->expansion_add ? (uint32_t)o->expansion_add : 128;      // options.h:18-24
    ix->ef = o->expansion_search ? (uint32_t)o->expansion_search : 64;
    if(const char *dv = std::getenv("LANTERN_GPU_DEVICE")) (void)hipSetDevice(std::atoi(dv));
    if(const char *vs = std::getenv("LANTERN_GPU_VIS_SLOTS")) ix->search_vis_slots = std::atoi(vs);  // tuning/debug: 0 = bitmap only
    (void)hipGetDevice(&ix->device);
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, ix->device) == hipSuccess) ix->num_cus = prop.multiProcessorCount;
    if(const char *tk = std::getenv("LANTERN_GPU_TICKETS")) ix->use_tickets = std::atoi(tk) != 0;
    if(hipMalloc((void **)&ix->d_totals, 64 * sizeof(unsigned long long)) != hipSuccess ||
       hipMalloc((void **)&ix->d_tickets, kTicketRing * sizeof(uint32_t)) != hipSuccess ||
       hipMemset(ix->d_totals, 0, 64 * sizeof(unsigned long long)) != hipSuccess) {
        if(ix->d_totals) (void)hipFree(ix->d_totals);
        if(ix->d_tickets) (void)hipFree(ix->d_tickets);
        delete ix;
        FAIL(e, "lantern_gpu: device allocation failed");
        return nullptr;
    }
    if(o->pq) {
        // the codebook as Lantern hands it over: num_centroids rows of `dimensions` floats, row c = centroid c of every
        // subvector, concatenated (pqtable.c:194-240).  Besides it, per subvector, the table the nearest-centroid search reads.
        ix->pq = true;
        ix->pq_S = (uint32_t)o->num_subvectors;
        ix->pq_C = (uint32_t)o->num_centroids;
        ix->pq_subdim = (uint32_t)(o->dimensions / o->num_subvectors);
        const size_t d = o->dimensions, subf = (size_t)((ix->pq_subdim + 3) / 4) * 4;
        ix->h_codebook.assign(pq_codebook, pq_codebook + (size_t)ix->pq_C * d);
        std::vector<float> centers((size_t)ix->pq_S * ix->pq_C * subf, 0.f);
        for(size_t sv = 0; sv < ix->pq_S; ++sv)
            for(size_t c = 0; c < ix->pq_C; ++c)
                std::memcpy(&centers[ (sv * ix->pq_C + c) * subf ], pq_codebook + c * d + sv * ix->pq_subdim, (size_t)ix->pq_subdim * 4);
        const bool ok = hipMalloc((void **)&ix->d_codebook, ix->h_codebook.size() * 4) == hipSuccess &&
                        hipMalloc((void **)&ix->d_centers, centers.size() * 4) == hipSuccess &&
                        hipMemcpy(ix->d_codebook, ix->h_codebook.data(), ix->h_codebook.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                        hipMemcpy(ix->d_centers, centers.data(), centers.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
        if(!ok) {
            usearch_error_t ignore = nullptr;
            usearch_free(ix, &ignore);
            FAIL(e, "lantern_gpu: device allocation failed (codebook)");
            return nullptr;
        }
    }
    return ix;
}
LANTERN_ABI_CATCH(e)

void usearch_free(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    void *ptrs[] = { ix->d_vec, ix->d_norm2, ix->d_labels, ix->d_levels, ix->d_nbr0, ix->d_upper_off, ix->d_upper_nbr, ix->d_bitmaps, ix->d_totals, ix->d_tickets,
                     ix->d_radius0, ix->d_radius_upper,
                     ix->d_codebook, ix->d_centers, ix->d_codes, ix->d_codes16, ix->d_touched, ix->d_trace, ix->d_trace_count };
    for(void *p : ptrs)
        if(p) (void)hipFree(p);
    for(void *p : ix->d_scratch)
        if(p) (void)hipFree(p);
    prof_resolve(ix, 0);
    for(hipEvent_t ev : ix->prof_free) (void)hipEventDestroy(ev);
    if(ix->h_single) (void)hipHostFree(ix->h_single);
    if(ix->h_stage) (void)hipHostFree(ix->h_stage);
    for(int sl = 0; sl < Index::kSearchSlots; ++sl) {
        if(sl > 0 && ix->slot_bitmaps[ sl ]) (void)hipFree(ix->slot_bitmaps[ sl ]);
        if(ix->slot_done[ sl ]) (void)hipEventDestroy(ix->slot_done[ sl ]);
    }
    if(ix->insert_done) (void)hipEventDestroy(ix->insert_done);
    for(hipStream_t ls : ix->lane_stream)
        if(ls) (void)hipStreamDestroy(ls);
    for(char *lh : ix->lane_host)
        if(lh) (void)hipHostFree(lh);
    ix->magic = 0;  // a use after free is refused by H() for as long as the allocator leaves the word alone
    delete ix;
}
LANTERN_ABI_CATCH_VOID(e)

void usearch_reserve(usearch_index_t h, size_t capacity, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!reserve_locked(ix, capacity)) FAIL(e, ix->err.c_str());
}
LANTERN_ABI_CATCH_VOID(e)

size_t usearch_size(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return 0;
    std::lock_guard<std::mutex> g(ix->mu);
    return logical_size(ix) + ix->pend_labels.size();  // logical size; build.c:117 polls this per tuple
}
LANTERN_ABI_CATCH(e)

size_t usearch_capacity(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return 0;
    std::lock_guard<std::mutex> g(ix->mu);
    return std::max(ix->cap, ix->n + ix->pend_labels.size());
}
LANTERN_ABI_CATCH(e)

size_t usearch_dimensions(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    return ix ? ix->opts.dimensions : 0;
}
LANTERN_ABI_CATCH(e)

static void add_common(Index *ix, const usearch_label_t *labels, const void *vectors, size_t n, usearch_scalar_kind_t kind,
                       int level, usearch_error_t *e)
try {
    if(!kind_accepted(ix, (int)kind)) { FAIL(e, "lantern_gpu: scalar kind of the vector does not match the index"); return; }
    if(!vectors || !labels) { FAIL(e, "lantern_gpu: null vector or label pointer"); return; }
    std::lock_guard<std::mutex> g(ix->mu);
    const size_t row_words = (size_t)ix->chunks * 4;
    const size_t in_bytes = input_bytes(ix, (int)kind);
    if(n >= ix->add_batch_max && (int)kind == ix->scalar && in_bytes == row_words * 4 && level < 0) {
        // bulk insert of rows that need no padding: upload straight from the caller's buffer (it is borrowed
        // for the duration of this call) instead of staging a copy
        if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return; }
        bool ok = true;
        insert_rows(ix, labels, nullptr, (const uint32_t *)vectors, n, &ok);
        if(!ok) FAIL(e, ix->err.c_str());
        return;
    }
    if(n >= ix->add_batch_max && (int)kind == usearch_scalar_f32_k && level < 0 &&
       (ix->scalar == usearch_scalar_f16_k || ix->scalar == usearch_scalar_i8_k || ix->b1_from_f32)) {
        // bulk insert of f32 rows into quantised storage: one upload, the conversion runs on the device
        if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return; }
        bool ok = true;
        insert_rows(ix, labels, nullptr, (const uint32_t *)vectors, n, &ok, true);
        if(!ok) FAIL(e, ix->err.c_str());
        return;
    }
    const size_t base = ix->pend_labels.size();
    try {
        ix->pend_labels.insert(ix->pend_labels.end(), labels, labels + n);
        ix->pend_levels.insert(ix->pend_levels.end(), n, level);
        ix->pend_rows.resize((base + n) * row_words);
    } catch(...) {  // an allocation failure half way: the three pending arrays go back to describing the same `base` vectors
        ix->pend_labels.resize(base);
        ix->pend_levels.resize(base);
        ix->pend_rows.resize(base * row_words);
        throw;      // ... and the caller gets the error string (abi_guard.hpp)
    }
    for(size_t i = 0; i < n; ++i) pad_row(ix, (const char *)vectors + i * in_bytes, (int)kind, &ix->pend_rows[ (base + i) * row_words ]);
    if(ix->pend_labels.size() >= ix->add_batch_max && !flush_locked(ix)) FAIL(e, ix->err.c_str());
}
LANTERN_ABI_CATCH_VOID(e)

void usearch_add(usearch_index_t h, usearch_label_t label, const void *vector, usearch_scalar_kind_t kind, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(ix) add_common(ix, &label, vector, 1, kind, -1, e);
}
LANTERN_ABI_CATCH_VOID(e)

void *lantern_gpu_host_alloc(size_t bytes)
try {
    void *p = nullptr;
    if(lantern_gpu_device_count() <= 0 || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
LANTERN_ABI_CATCH(nullptr)

void lantern_gpu_host_free(void *p)
try {
    if(p) (void)hipHostFree(p);
}
LANTERN_ABI_CATCH_VOID(nullptr)

void lantern_gpu_add_many(usearch_index_t h, const usearch_label_t *labels, const void *vectors, size_t n,
                          usearch_scalar_kind_t kind, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(ix && n) add_common(ix, labels, vectors, n, kind, -1, e);
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_add_with_level(usearch_index_t h, usearch_label_t label, const void *vector, usearch_scalar_kind_t kind,
                                int level, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    if(level < 0 || level > 255) { FAIL(e, "lantern_gpu: level out of range"); return; }
    add_common(ix, &label, vector, 1, kind, level, e);
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_add_row_sharded(usearch_index_t h, lantern_gpu_comm_t *comm, const usearch_label_t *labels, const void *vectors,
                                 size_t n_shard, usearch_scalar_kind_t kind, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    if(!comm) { FAIL(e, "lantern_gpu: null communicator"); return; }
    if(!kind_accepted(ix, (int)kind)) { FAIL(e, "lantern_gpu: scalar kind of the vector does not match the index"); return; }
    if(n_shard && (!vectors || !labels)) { FAIL(e, "lantern_gpu: null vector or label pointer"); return; }
    std::lock_guard<std::mutex> g(ix->mu);
    if(!add_row_sharded_locked(ix, (Comm *)comm, labels, vectors, n_shard, (int)kind)) FAIL(e, ix->err.c_str());
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_add_sharded(usearch_index_t h, lantern_gpu_comm_t *comm, const usearch_label_t *labels, const void *vectors,
                             size_t n_shard, usearch_scalar_kind_t kind, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    if(!comm) { FAIL(e, "lantern_gpu: null communicator"); return; }
    if(!kind_accepted(ix, (int)kind)) { FAIL(e, "lantern_gpu: scalar kind of the vector does not match the index"); return; }
    if(n_shard && (!vectors || !labels)) { FAIL(e, "lantern_gpu: null vector or label pointer"); return; }
    std::lock_guard<std::mutex> g(ix->mu);
    if(!add_sharded_locked(ix, (Comm *)comm, labels, vectors, n_shard, (int)kind)) FAIL(e, ix->err.c_str());
}
LANTERN_ABI_CATCH_VOID(e)

// Row-partitioned search (include/lantern_gpu.h): COLLECTIVE.  Every rank searches the SAME queries in its OWN index -- a
// disjoint share of the rows, built independently -- the per-rank top-k lists are all-gathered in place in HBM (RCCL: over
// xGMI) and merged on the device by (distance, label).
void lantern_gpu_search_partitioned(usearch_index_t h, lantern_gpu_comm_t *comm_, const void *queries, size_t nq, usearch_scalar_kind_t kind,
                                    size_t k, size_t ef, usearch_label_t *labels, float *distances, uint32_t *counts, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    Comm *comm = (Comm *)comm_;
    if(!comm) { FAIL(e, "lantern_gpu: null communicator"); return; }
    if(!kind_accepted(ix, (int)kind)) { FAIL(e, "lantern_gpu: scalar kind of the queries does not match the index"); return; }
    if(nq == 0 || k == 0) return;
    if(!queries || !labels || !distances) { FAIL(e, "lantern_gpu: null buffer"); return; }
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return; }
    const int    W = comm->world, R = comm->rank;
    const size_t part = nq * k;
    const size_t row_words = (size_t)ix->chunks * 4;
    std::vector<uint32_t> padded(nq * row_words);
    pad_rows(ix, queries, (int)kind, nq, padded.data());
    char *dq = (char *)scratch(ix, 5, nq * row_words * 4);
    // [W][nq][k] labels | [W][nq][k] distances | merged labels | merged distances | merged counts
    char *dall = (char *)scratch(ix, 6, (size_t)W * part * 12 + part * 12 + nq * 4 + 64);
    if(!dq || !dall) { FAIL(e, ix->err.c_str()); return; }
    uint64_t *g_lab = (uint64_t *)dall;
    float    *g_dist = (float *)(dall + (size_t)W * part * 8);
    uint64_t *m_lab = (uint64_t *)(dall + (size_t)W * part * 12);
    float    *m_dist = (float *)((char *)m_lab + part * 8);
    uint32_t *m_cnt = (uint32_t *)((char *)m_dist + part * 4);
    bool      ok = hipMemcpyAsync(dq, padded.data(), nq * row_words * 4, hipMemcpyHostToDevice, ix->stream) == hipSuccess;
    if(ix->n == 0) {  // an empty share contributes only unused entries
        ok = ok && hipMemsetAsync(g_lab + (size_t)R * part, 0, part * 8, ix->stream) == hipSuccess;
        std::vector<float> inf(part, __builtin_inff());
        ok = ok && hipMemcpyAsync(g_dist + (size_t)R * part, inf.data(), part * 4, hipMemcpyHostToDevice, ix->stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(ix->stream) == hipSuccess;  // `inf` is a local
    } else {
        ok = ok && run_search_device(ix, (const uint4 *)dq, nq, k, ef, 0, g_lab + (size_t)R * part, g_dist + (size_t)R * part, nullptr, nullptr,
                                     nullptr, nullptr, ix->stream, ix->search_waves);
    }
    if(ok && W > 1) {
        std::vector<size_t> off(W), cnt(W);
        for(int r = 0; r < W; ++r) { off[ r ] = (size_t)r * part * 8; cnt[ r ] = part * 8; }
        ok = comm->allgatherv_device(g_lab, off.data(), cnt.data(), ix->stream);
        for(int r = 0; r < W; ++r) { off[ r ] = (size_t)r * part * 4; cnt[ r ] = part * 4; }
        ok = ok && comm->allgatherv_device(g_dist, off.data(), cnt.data(), ix->stream);
        if(!ok) set_err(ix, "lantern_gpu: exchange of the per-rank results failed: " + comm->err);
    }
    ok = ok && launch_merge_parts(g_lab, g_dist, (uint32_t)W, (uint32_t)nq, (uint32_t)k, m_lab, m_dist, m_cnt, ix->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(labels, m_lab, part * 8, hipMemcpyDeviceToHost, ix->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(distances, m_dist, part * 4, hipMemcpyDeviceToHost, ix->stream) == hipSuccess;
    if(counts) ok = ok && hipMemcpyAsync(counts, m_cnt, nq * 4, hipMemcpyDeviceToHost, ix->stream) == hipSuccess;
    ok = ok && comm->wait(ix->stream);
    if(!ok) {
        if(ix->err.empty()) set_err(ix, comm->err.empty() ? "lantern_gpu: HIP failure during the partitioned search" : comm->err);
        FAIL(e, ix->err.c_str());
    }
}
LANTERN_ABI_CATCH_VOID(e)

uint64_t lantern_gpu_graph_checksum(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return 0;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return 0; }
    const size_t n = ix->n;
    std::vector<uint32_t> nbr0(n * ix->M0), upper(ix->upper_blocks * ix->M);
    bool ok = true;
    if(n) ok = ok && hipMemcpy(nbr0.data(), ix->d_nbr0, nbr0.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if(!upper.empty()) ok = ok && hipMemcpy(upper.data(), ix->d_upper_nbr, upper.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if(!ok) { FAIL(e, "lantern_gpu: HIP failure reading the graph"); return 0; }
    uint64_t hsh = splitmix64((uint64_t)n ^ ((uint64_t)ix->entry << 32) ^ ((uint64_t)(uint32_t)ix->max_level << 24));
    for(size_t i = 0; i < n; ++i) hsh = splitmix64(hsh ^ ix->levels[ i ] ^ (ix->labels[ i ] << 8));
    for(uint32_t v : nbr0) hsh = splitmix64(hsh ^ v);
    for(uint32_t v : upper) hsh = splitmix64(hsh ^ v);
    return hsh;
}
LANTERN_ABI_CATCH(e)

// the host-side rules a second builder (the test oracle, a CPU fallback on the reference side) must share to
// reproduce a device build: the stateless level draw and the batch plan (host_util.hpp)
int lantern_gpu_level_for(uint64_t seed, uint64_t slot, uint32_t connectivity) { return level_for(seed, slot, connectivity < 2 ? 2 : connectivity); }

size_t lantern_gpu_row_shard_plan(const uint64_t *shard_sizes, int world, uint64_t seed, uint32_t connectivity, size_t max_batch, size_t min_ratio,
                                  size_t *first, size_t *count, size_t *share, size_t capacity)
try {
    if(!shard_sizes || world < 1) return 0;
    size_t N = 0;
    for(int r = 0; r < world; ++r) N += (size_t)shard_sizes[ r ];
    std::vector<int> lv(N);
    for(size_t i = 0; i < N; ++i) lv[ i ] = level_for(seed, i, connectivity < 2 ? 2 : connectivity);
    std::vector<size_t> pf, pc, ps;
    row_shard_plan(shard_sizes, world, lv.data(), N, max_batch ? max_batch : 1, min_ratio ? min_ratio : 1, pf, pc, ps);
    for(size_t t = 0; t < pf.size() && t < capacity; ++t) {
        if(first) first[ t ] = pf[ t ];
        if(count) count[ t ] = pc[ t ];
        if(share)
            for(int r = 0; r < world; ++r) share[ t * (size_t)world + (size_t)r ] = ps[ t * (size_t)world + (size_t)r ];
    }
    return pf.size();
}
LANTERN_ABI_CATCH(nullptr)

size_t lantern_gpu_plan_batch(size_t current_size, int max_level, const int *pending_levels, size_t pending, size_t max_batch,
                              size_t min_ratio)
try {
    if(!pending_levels) return 0;
    return plan_batch(current_size, max_level, pending_levels, pending, max_batch ? max_batch : 1, min_ratio ? min_ratio : 1);
}
LANTERN_ABI_CATCH(nullptr)

void lantern_gpu_flush(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) FAIL(e, ix->err.c_str());
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_set_seed(usearch_index_t h, uint64_t seed, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(ix) ix->seed = seed;
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_set_add_batch(usearch_index_t h, size_t max_batch, size_t min_ratio, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    if(max_batch == 0 || min_ratio == 0) { FAIL(e, "lantern_gpu: batch parameters must be positive"); return; }
    std::lock_guard<std::mutex> g(ix->mu);
    ix->add_batch_max = max_batch;
    ix->add_min_ratio = min_ratio;
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_set_search_shape(usearch_index_t h, int waves, int max_wg, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    if(waves < 0 || waves > 8) { FAIL(e, "lantern_gpu: waves_per_query must be in [1, 8], or 0 = automatic"); return; }
    ix->search_waves = waves;
    ix->insert_waves = waves ? waves : 4;
    ix->search_max_wg = max_wg;
}
LANTERN_ABI_CATCH_VOID(e)

size_t usearch_search_ef(usearch_index_t h, const void *query, usearch_scalar_kind_t kind, size_t k, size_t ef, bool streaming,
                         usearch_label_t *labels, float *distances, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return 0;
    if(!kind_accepted(ix, (int)kind)) { FAIL(e, "lantern_gpu: scalar kind of the query does not match the index"); return 0; }
    if(k == 0) return 0;
    if(!query || !labels || !distances) { FAIL(e, "lantern_gpu: null query or result pointer"); return 0; }
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return 0; }
    ix->err.clear();
    const size_t out = search_one_locked(ix, &ix->default_cursor, query, (int)kind, k, ef, streaming, labels, distances);
    if(!ix->err.empty()) FAIL(e, ix->err.c_str());
    return out;
}
LANTERN_ABI_CATCH(e)

// ---- cursors: the per-scan half of usearch_search_ef's streaming contract ------------------------------------------
struct lantern_gpu_cursor
{
    Index *ix;
    Cursor cur;
};

lantern_gpu_cursor_t *lantern_gpu_cursor_open(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return nullptr;
    lantern_gpu_cursor *c = new(std::nothrow) lantern_gpu_cursor();
    if(!c) { FAIL(e, "lantern_gpu: out of host memory"); return nullptr; }
    c->ix = ix;
    return c;
}
LANTERN_ABI_CATCH(e)

size_t lantern_gpu_cursor_search(lantern_gpu_cursor_t *c, const void *query, usearch_scalar_kind_t kind, size_t k, size_t ef,
                                 bool streaming, usearch_label_t *labels, float *distances, usearch_error_t *e)
try {
    CLEAR(e);
    if(!c) { FAIL(e, "lantern_gpu: null cursor"); return 0; }
    Index *ix = H(c->ix, e);
    if(!ix) return 0;
    if(!kind_accepted(ix, (int)kind)) { FAIL(e, "lantern_gpu: scalar kind of the query does not match the index"); return 0; }
    if(k == 0) return 0;
    if(!query || !labels || !distances) { FAIL(e, "lantern_gpu: null query or result pointer"); return 0; }
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return 0; }
    ix->err.clear();
    const size_t out = search_one_locked(ix, &c->cur, query, (int)kind, k, ef, streaming, labels, distances);
    if(!ix->err.empty()) FAIL(e, ix->err.c_str());
    return out;
}
LANTERN_ABI_CATCH(e)

size_t lantern_gpu_cursor_seen(lantern_gpu_cursor_t *c) { return c ? c->cur.seen.size() : 0; }

void lantern_gpu_cursor_close(lantern_gpu_cursor_t *c) { delete c; }

// Device-resident queries come with their row stride: the kernel reads query i at d_queries + i * (stored row stride), and a caller
// that laid its rows out at any other stride would get wrong answers and a read past the end of its buffer with no error.  The
// stride is therefore part of the call (`_strided`) and a mismatch is refused; the form without it is accepted only where the
// stride is unambiguous -- the index stores rows at the vector's own length rounded up to 16 bytes.
static const char *kStrideMismatch = "lantern_gpu: the query row stride does not match the index's stored row stride (lantern_gpu_row_bytes)";
static const char *kStrideAmbiguous =
    "lantern_gpu: this index stores rows at a stride wider than the vector's own length (lantern_gpu_row_bytes): device-resident "
    "queries must be handed over with their stride, through lantern_gpu_search_batch_device_strided";

void lantern_gpu_search_batch_device_strided(usearch_index_t h, const void *d_queries, size_t query_stride_bytes, size_t nq, size_t k, size_t ef,
                                             size_t skip, uint64_t *d_labels, float *d_distances, uint32_t *d_slots, uint32_t *d_counts,
                                             uint64_t *d_D, uint64_t *d_E, void *stream, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(query_stride_bytes != (size_t)ix->chunks * 16) { FAIL(e, kStrideMismatch); return; }
    if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return; }
    if(!run_search_device(ix, (const uint4 *)d_queries, nq, k, ef, skip, d_labels, d_distances, d_slots, d_counts, d_D, d_E,
                          (hipStream_t)stream, ix->search_waves))
        FAIL(e, ix->err.c_str());
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_search_batch_device(usearch_index_t h, const void *d_queries, size_t nq, size_t k, size_t ef, size_t skip,
                                     uint64_t *d_labels, float *d_distances, uint32_t *d_slots, uint32_t *d_counts,
                                     uint64_t *d_D, uint64_t *d_E, void *stream, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(ix->chunks != ix->natural_chunks) { FAIL(e, kStrideAmbiguous); return; }
    if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return; }
    if(!run_search_device(ix, (const uint4 *)d_queries, nq, k, ef, skip, d_labels, d_distances, d_slots, d_counts, d_D, d_E,
                          (hipStream_t)stream, ix->search_waves))
        FAIL(e, ix->err.c_str());
}
LANTERN_ABI_CATCH_VOID(e)

// the page-locked staging block `which` (0 .. kLanes - 1: the lanes, kLanes: lantern_gpu_search_batch), grown on demand; nullptr on failure
static char *host_stage(Index *ix, int which, size_t need)
try {
    if(ix->lane_host_bytes[ which ] < need) {
        if(ix->lane_host[ which ]) (void)hipHostFree(ix->lane_host[ which ]);
        ix->lane_host[ which ] = nullptr;
        ix->lane_host_bytes[ which ] = 0;
        const size_t grow = need + need / 2;
        if(hipHostMalloc((void **)&ix->lane_host[ which ], grow, hipHostMallocMapped) != hipSuccess) return nullptr;  // (mapped: lane_notify's kernels write into it)
        ix->lane_host_bytes[ which ] = grow;
    }
    return ix->lane_host[ which ];
}
LANTERN_ABI_CATCH(nullptr)

void lantern_gpu_search_batch(usearch_index_t h, const void *queries, size_t nq, usearch_scalar_kind_t kind, size_t k, size_t ef,
                              usearch_label_t *labels, float *distances, uint32_t *counts, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    if(!kind_accepted(ix, (int)kind)) { FAIL(e, "lantern_gpu: scalar kind of the queries does not match the index"); return; }
    if(nq == 0 || k == 0) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return; }
    const size_t row_words = (size_t)ix->chunks * 4;
    // queries and answers pass through one page-locked block: one copy up, one down (labels | distances | counts as they lie),
    // at the link's rate instead of through the runtime's staging of pageable memory
    const size_t q_bytes = nq * row_words * 4, out_bytes = nq * k * 12 + nq * 4;
    char *const  hs = host_stage(ix, Index::kLanes, q_bytes + out_bytes + 64);
    if(!hs) { FAIL(e, "lantern_gpu: cannot allocate the page-locked staging block"); return; }
    uint32_t *const padded = (uint32_t *)hs;
    char *const     h_out = hs + ((q_bytes + 63) & ~(size_t)63);
    pad_rows(ix, queries, (int)kind, nq, padded);
    char *dq = (char *)scratch(ix, 5, q_bytes);
    char *dout = (char *)scratch(ix, 6, out_bytes + 64);
    if(!dq || !dout) { FAIL(e, ix->err.c_str()); return; }
    uint64_t *d_lab = (uint64_t *)dout;
    float    *d_dist = (float *)(dout + nq * k * 8);
    uint32_t *d_cnt = (uint32_t *)(dout + nq * k * 12);
    bool      ok = hipMemcpyAsync(dq, padded, q_bytes, hipMemcpyHostToDevice, ix->stream) == hipSuccess;
    ok = ok && run_search_device(ix, (const uint4 *)dq, nq, k, ef, 0, d_lab, d_dist, nullptr, d_cnt, nullptr, nullptr, ix->stream,
                                 ix->search_waves);
    ok = ok && hipMemcpyAsync(h_out, dout, out_bytes, hipMemcpyDeviceToHost, ix->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(ix->stream) == hipSuccess;
    if(!ok) {
        if(ix->err.empty()) set_err(ix, "lantern_gpu: HIP failure during batched search");
        FAIL(e, ix->err.c_str());
        return;
    }
    std::memcpy(labels, h_out, nq * k * 8);
    std::memcpy(distances, h_out + nq * k * 8, nq * k * 4);
    if(counts) std::memcpy(counts, h_out + nq * k * 12, nq * 4);
}
LANTERN_ABI_CATCH_VOID(e)

// The same as lantern_gpu_search_batch for a caller that keeps SEVERAL batches in flight (the scan-side service: up to eight
// dispatchers, each executing a batch while another collects the next): each lane has its own stream and staging buffers, the index mutex is
// held only while the lane's copies and its launch are queued, and the wait for the answers happens outside it -- so the two
// lanes' launches overlap on the device (each in its own visited-bitmap slab: acquire_search_slot).
void lantern_gpu_search_batch_lane(usearch_index_t h, int lane, const void *queries, size_t nq, usearch_scalar_kind_t kind, size_t k, size_t ef,
                                   usearch_label_t *labels, float *distances, uint32_t *counts, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    if(lane < 0 || lane >= Index::kLanes) { FAIL(e, "lantern_gpu: lane must be in [0, 8)"); return; }
    if(!kind_accepted(ix, (int)kind)) { FAIL(e, "lantern_gpu: scalar kind of the queries does not match the index"); return; }
    if(nq == 0 || k == 0) return;
    if(!queries || !labels || !distances) { FAIL(e, "lantern_gpu: null buffer"); return; }
    const size_t row_words = (size_t)ix->chunks * 4;
    // Queries and answers pass through ONE page-locked block per lane (a lane has one caller at a time): the padded queries go up
    // in one copy, labels + distances + counts come back in one, both at the link's rate and without the runtime's staging of
    // pageable memory (four copies of it before: ~40 us of a small batch's ~150).
    const size_t q_bytes = nq * row_words * 4, out_bytes = nq * k * 12 + nq * 4, need = q_bytes + out_bytes + 64;
    char *const hs = host_stage(ix, lane, need);
    if(!hs) { FAIL(e, "lantern_gpu: cannot allocate the lane's page-locked staging block"); return; }
    uint32_t *const padded = (uint32_t *)hs;  // (chunks and the scalar kind are fixed at init: no lock needed yet)
    char *const     h_out = hs + ((q_bytes + 63) & ~(size_t)63);
    pad_rows(ix, queries, (int)kind, nq, padded);
    hipStream_t st = nullptr;
    bool        ok = true;
    // A lane's error text belongs to the calling thread: ix->err is shared by both lanes (and by every other entry point) and
    // may be rewritten or cleared the moment the mutex is dropped, while the caller -- the scan service's dispatcher -- reads
    // the message later and without the lock.
    static thread_local std::string msg;
    msg.clear();
    {
        std::lock_guard<std::mutex> g(ix->mu);
        if(!flush_locked(ix)) { msg = ix->err; FAIL(e, msg.c_str()); return; }
        if(!ix->lane_stream[ lane ] && hipStreamCreateWithFlags(&ix->lane_stream[ lane ], hipStreamNonBlocking) != hipSuccess) {
            FAIL(e, "lantern_gpu: cannot create the lane's stream");
            return;
        }
        st = ix->lane_stream[ lane ];
        char *dq = (char *)scratch(ix, 12 + 2 * lane, nq * row_words * 4);
        char *dout = (char *)scratch(ix, 13 + 2 * lane, nq * k * 12 + nq * 4 + 64);
        if(!dq || !dout) { msg = ix->err; FAIL(e, msg.c_str()); return; }
        uint64_t *d_lab = (uint64_t *)dout;
        float    *d_dist = (float *)(dout + nq * k * 8);
        uint32_t *d_cnt = (uint32_t *)(dout + nq * k * 12);
        ok = hipMemcpyAsync(dq, padded, q_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
        ok = ok && run_search_device(ix, (const uint4 *)dq, nq, k, ef, 0, d_lab, d_dist, nullptr, d_cnt, nullptr, nullptr, st, ix->search_waves);
        ok = ok && hipMemcpyAsync(h_out, dout, out_bytes, hipMemcpyDeviceToHost, st) == hipSuccess;  // labels | distances | counts, as they lie
        if(!ok) msg = ix->err.empty() ? "lantern_gpu: HIP failure during batched search" : ix->err;
    }
    // the wait is the long part: outside the mutex, so that the other lane can queue its batch meanwhile
    if(hipStreamSynchronize(st) != hipSuccess && ok) { ok = false; msg = "lantern_gpu: HIP failure during batched search"; }
    if(!ok) { FAIL(e, msg.c_str()); return; }
    std::memcpy(labels, h_out, nq * k * 8);
    std::memcpy(distances, h_out + nq * k * 8, nq * k * 4);
    if(counts) std::memcpy(counts, h_out + nq * k * 12, nq * 4);
}
LANTERN_ABI_CATCH_VOID(e)

// lantern_gpu_search_batch_lane with the answers handed on ONE QUERY AT A TIME: the kernel writes labels | distances | counts straight
// into the lane's page-locked, device-mapped block and raises a per-query word there as each walk ends (SearchArgs::done_flags);
// the calling thread polls those words and calls `done(ctx, which, count)` for the queries that finished since its last look --
// their rows are in the caller's arrays by then.  A launch's walks differ in length by 2x and more (140 hops where the mean is
// 78): a caller that answers each client when ITS walk is over, instead of when the longest one is, halves what a client of a
// small batch waits (the scan-side service: scan_server.cpp).  Returns when every query has been handed on.
void lantern_gpu_search_batch_lane_notify(usearch_index_t h, int lane, const void *queries, size_t nq, usearch_scalar_kind_t kind, size_t k, size_t ef,
                                          usearch_label_t *labels, float *distances, uint32_t *counts, lantern_gpu_queries_done_fn done, void *done_ctx,
                                          usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    if(lane < 0 || lane >= Index::kLanes) { FAIL(e, "lantern_gpu: lane must be in [0, 8)"); return; }
    if(!kind_accepted(ix, (int)kind)) { FAIL(e, "lantern_gpu: scalar kind of the queries does not match the index"); return; }
    if(nq == 0 || k == 0) return;
    if(!queries || !labels || !distances || !done) { FAIL(e, "lantern_gpu: null buffer or callback"); return; }
    const size_t row_words = (size_t)ix->chunks * 4;
    const size_t q_bytes = nq * row_words * 4, out_bytes = nq * k * 12 + nq * 4, flag_bytes = nq * 4;
    const size_t out_at = (q_bytes + 63) & ~(size_t)63, flag_at = (out_at + out_bytes + 63) & ~(size_t)63, need = flag_at + flag_bytes + 64;
    char *const  hs = host_stage(ix, lane, need);
    if(!hs) { FAIL(e, "lantern_gpu: cannot allocate the lane's page-locked staging block"); return; }
    char *hs_dev = nullptr;  // the same block as the device names it
    if(hipHostGetDevicePointer((void **)&hs_dev, hs, 0) != hipSuccess || !hs_dev) {
        (void)hipGetLastError();
        FAIL(e, "lantern_gpu: the lane's staging block is not device-mapped");
        return;
    }
    uint32_t *const padded = (uint32_t *)hs;
    char *const     h_out = hs + out_at;
    uint32_t *const flags = (uint32_t *)(hs + flag_at);
    pad_rows(ix, queries, (int)kind, nq, padded);
    std::memset(flags, 0, flag_bytes);
    hipStream_t st = nullptr;
    bool        ok = true;
    static thread_local std::string msg;
    msg.clear();
    {
        std::lock_guard<std::mutex> g(ix->mu);
        if(!flush_locked(ix)) { msg = ix->err; FAIL(e, msg.c_str()); return; }
        if(!ix->lane_stream[ lane ] && hipStreamCreateWithFlags(&ix->lane_stream[ lane ], hipStreamNonBlocking) != hipSuccess) {
            FAIL(e, "lantern_gpu: cannot create the lane's stream");
            return;
        }
        st = ix->lane_stream[ lane ];
        // A service-sized batch's queries are read by the walks straight out of the page-locked block (each workgroup fetches its 3 KB
        // row over the host link once, as a lone usearch_search_ef does): no copy command in front of the kernel -- a DMA command costs
        // tens of microseconds of queueing, as much as a tenth of a walk.  Large batches are copied into HBM first, at the link's rate.
        const bool  direct_queries = q_bytes <= (size_t)1 << 20;
        char       *dq = direct_queries ? hs_dev : (char *)scratch(ix, 12 + 2 * lane, nq * row_words * 4);
        if(!dq) { msg = ix->err; FAIL(e, msg.c_str()); return; }
        char *const d_out = hs_dev + out_at;
        ok = direct_queries || hipMemcpyAsync(dq, padded, q_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
        ok = ok && run_search_device(ix, (const uint4 *)dq, nq, k, ef, 0, (uint64_t *)d_out, (float *)(d_out + nq * k * 8), nullptr, (uint32_t *)(d_out + nq * k * 12),
                                     nullptr, nullptr, st, ix->search_waves, nullptr, (uint32_t *)(hs_dev + flag_at));
        if(!ok) msg = ix->err.empty() ? "lantern_gpu: HIP failure during batched search" : ix->err;
    }
    if(!ok) { (void)hipStreamSynchronize(st); FAIL(e, msg.c_str()); return; }
    // hand the answers on as their flags come up (outside the mutex: the other lanes queue their batches meanwhile)
    std::vector<uint32_t> pending(nq), ready;
    for(size_t i = 0; i < nq; ++i) pending[ i ] = (uint32_t)i;
    ready.reserve(nq);
    // Waiting: the flags change in host memory, so a look is a load (a cache miss when one changed).  The lane SPINS only for a
    // bounded time after the last answer came up -- LANTERN_GPU_NOTIFY_SPIN_US, default 400: longer than a service-sized launch takes
    // to its first answers and than the spacing of walks ending in a service-sized batch (measured, 1M x 768, 256 backends:
    // 524 k / 555 k / 568 k scans/s at 100 / 400 / never sleeping -- within the run-to-run spread:
    // profiles/r06_scan_notify_spin_sweep.txt) -- and then backs off to sleeping 10 .. 100 us at a time, so a lane waiting out a
    // long launch does not hold one of the database host's cores at 100 % against the backends it serves.  The runtime is asked
    // only now and then -- hipStreamQuery takes its lock, which the other lanes' launches need -- to notice a launch that ended
    // without raising its flags (it cannot, short of a fault) or failed.
    static const long spin_ns = [] {
        const char *v = std::getenv("LANTERN_GPU_NOTIFY_SPIN_US");
        return (long)(v ? std::max(0, std::atoi(v)) : 400) * 1000L;
    }();
    auto now_ns = [] {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (long)ts.tv_sec * 1000000000L + ts.tv_nsec;
    };
    unsigned idle = 0;
    long     quiet_since = now_ns(), nap_ns = 10000;
    bool     drained = false;  // the stream has finished: whatever is still pending is complete as well
    while(!pending.empty()) {
        ready.clear();
        for(size_t i = 0; i < pending.size();) {
            const uint32_t j = pending[ i ];
            if(drained || __atomic_load_n(&flags[ j ], __ATOMIC_ACQUIRE) != 0) {
                ready.push_back(j);
                pending[ i ] = pending.back();
                pending.pop_back();
            } else {
                ++i;
            }
        }
        if(!ready.empty()) {
            for(uint32_t j : ready) {
                std::memcpy(labels + (size_t)j * k, h_out + (size_t)j * k * 8, k * 8);
                std::memcpy(distances + (size_t)j * k, h_out + nq * k * 8 + (size_t)j * k * 4, k * 4);
                if(counts) std::memcpy(counts + j, h_out + nq * k * 12 + (size_t)j * 4, 4);
            }
            done(done_ctx, ready.data(), ready.size());
            idle = 0;
            quiet_since = now_ns();
            nap_ns = 10000;
            continue;
        }
        ++idle;
        const bool napping = (idle & 0x3Fu) == 0 && now_ns() - quiet_since > spin_ns;
        if((idle & 0x3FFFu) == 0 || (napping && (idle & 0x3FFu) == 0)) {
            const hipError_t q = hipStreamQuery(st);
            if(q == hipSuccess) { (void)hipStreamSynchronize(st); drained = true; }
            else if(q != hipErrorNotReady) { (void)hipGetLastError(); ok = false; break; }
        } else if(napping) {
            timespec nap{0, nap_ns};
            nanosleep(&nap, nullptr);
            nap_ns = std::min(nap_ns * 2, 100000L);
            idle |= 0x3Fu;  // look at the flags, then nap again (one look per nap until something comes up)
        } else if((idle & 0xFFu) == 0) {
            std::this_thread::yield();
        } else {
            cpu_relax();
        }
    }
    if(hipStreamSynchronize(st) != hipSuccess) ok = false;
    if(!ok) { msg = "lantern_gpu: HIP failure during batched search"; FAIL(e, msg.c_str()); }
}
LANTERN_ABI_CATCH_VOID(e)

// ---- distances ------------------------------------------------------------------------------------

static bool metric_ok(usearch_metric_kind_t m) { return m == usearch_metric_cos_k || m == usearch_metric_l2sq_k || m == usearch_metric_hamming_k; }

void lantern_gpu_distance_matrix(const void *a, size_t na, const void *b, size_t nb, usearch_scalar_kind_t kind, size_t dims,
                                 usearch_metric_kind_t metric, int exact_order, float *out, usearch_error_t *e)
try {
    CLEAR(e);
    if(!metric_ok(metric)) { FAIL(e, "lantern_gpu: unsupported metric kind (expected cos, l2sq or hamming)"); return; }
    const bool ham = metric == usearch_metric_hamming_k;
    if(ham != (kind == usearch_scalar_b1_k) || (!ham && kind != usearch_scalar_f32_k)) { FAIL(e, "lantern_gpu: scalar kind does not fit the metric"); return; }
    if(!a || !b || !out || dims == 0) { FAIL(e, "lantern_gpu: bad arguments"); return; }
    if(lantern_gpu_device_count() <= 0) { FAIL(e, kNoDevice); return; }
    if(na == 0 || nb == 0) return;
    const size_t words = ham ? (dims + 31) / 32 : dims, in_bytes = ham ? (dims + 7) / 8 : dims * 4;
    const size_t chunks = (words + 3) / 4, row = chunks * 16;
    std::vector<char> ha(na * row, 0), hb(nb * row, 0);
    for(size_t i = 0; i < na; ++i) std::memcpy(&ha[ i * row ], (const char *)a + i * in_bytes, in_bytes);
    for(size_t i = 0; i < nb; ++i) std::memcpy(&hb[ i * row ], (const char *)b + i * in_bytes, in_bytes);
    void *da = nullptr, *db = nullptr, *dout = nullptr, *dn = nullptr;
    bool  ok = hipMalloc(&da, na * row) == hipSuccess && hipMalloc(&db, nb * row) == hipSuccess &&
              hipMalloc(&dout, na * nb * 4) == hipSuccess && hipMalloc(&dn, (na + nb) * 4) == hipSuccess;
    ok = ok && hipMemcpy(da, ha.data(), na * row, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(db, hb.data(), nb * row, hipMemcpyHostToDevice) == hipSuccess;
    if(exact_order) {
        ok = ok && launch_pairs((int)metric, (const uint4 *)da, (uint32_t)na, (const uint4 *)db, (uint32_t)nb, (uint32_t)chunks,
                                (float *)dout, nullptr) == hipSuccess;
    } else {  // the dense contraction: fp32 MFMA for l2sq / cos
        float *an = (float *)dn, *bnn = an + na;
        if(!ham) {
            ok = ok && launch_row_norms((const uint4 *)da, (uint32_t)na, (uint32_t)chunks, an, nullptr) == hipSuccess;
            ok = ok && launch_row_norms((const uint4 *)db, (uint32_t)nb, (uint32_t)chunks, bnn, nullptr) == hipSuccess;
        }
        ok = ok && launch_dense((int)metric, (const uint4 *)da, (uint32_t)na, (const uint4 *)db, (uint32_t)nb, (uint32_t)chunks, an, bnn,
                                (float *)dout, (uint32_t)nb, nullptr) == hipSuccess;
    }
    ok = ok && hipMemcpy(out, dout, na * nb * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if(da) (void)hipFree(da);
    if(db) (void)hipFree(db);
    if(dout) (void)hipFree(dout);
    if(dn) (void)hipFree(dn);
    if(!ok) FAIL(e, "lantern_gpu: HIP failure in distance_matrix");
}
LANTERN_ABI_CATCH_VOID(e)

float usearch_distance(const void *a, const void *b, usearch_scalar_kind_t kind, size_t dims, usearch_metric_kind_t metric,
                       usearch_error_t *e)
try {
    float out = 0.f;
    lantern_gpu_distance_matrix(a, 1, b, 1, kind, dims, metric, 1, &out, e);
    return out;
}
LANTERN_ABI_CATCH(e)

void lantern_gpu_distance_gather(usearch_index_t h, const void *query, const uint32_t *slots, size_t n, float *out,
                                 usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix || n == 0) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix) || !pq_expand_locked(ix)) { FAIL(e, ix->err.c_str()); return; }
    for(size_t i = 0; i < n; ++i)
        if(slots[ i ] >= ix->n) { FAIL(e, "lantern_gpu: slot out of range"); return; }
    const size_t row = (size_t)ix->chunks * 16;
    char        *buf = (char *)scratch(ix, 5, row + n * 8 + 16);
    if(!buf) { FAIL(e, ix->err.c_str()); return; }
    std::vector<uint32_t> padded((size_t)ix->chunks * 4);
    pad_row(ix, query, (ix->scalar == usearch_scalar_b1_k && !ix->b1_from_f32) ? usearch_scalar_b1_k : usearch_scalar_f32_k, padded.data());
    uint32_t *d_slots = (uint32_t *)(buf + row);
    float    *d_out = (float *)(buf + row + n * 4);
    bool      ok = hipMemcpyAsync(buf, padded.data(), row, hipMemcpyHostToDevice, ix->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(d_slots, slots, n * 4, hipMemcpyHostToDevice, ix->stream) == hipSuccess;
    hipEvent_t t0 = nullptr, t1 = nullptr;  // the kernel alone (lantern_gpu_last_gather_ms: bench.py's in-run gather ceiling)
    ok = ok && hipEventCreate(&t0) == hipSuccess && hipEventCreate(&t1) == hipSuccess;
    ok = ok && hipEventRecord(t0, ix->stream) == hipSuccess;
    ok = ok && launch_gather(ix->mcode, ix->view(), (const uint4 *)buf, d_slots, (uint32_t)n, d_out, ix->stream) == hipSuccess;
    ok = ok && hipEventRecord(t1, ix->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(out, d_out, n * 4, hipMemcpyDeviceToHost, ix->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(ix->stream) == hipSuccess;
    ix->last_gather_ms = 0.f;
    if(ok) (void)hipEventElapsedTime(&ix->last_gather_ms, t0, t1);
    if(t0) (void)hipEventDestroy(t0);
    if(t1) (void)hipEventDestroy(t1);
    if(!ok) FAIL(e, "lantern_gpu: HIP failure in distance_gather");
}
LANTERN_ABI_CATCH_VOID(e)

// Exact k-NN of nq device-resident query rows over nb device-resident base rows (both `chunks` uint4 per row):
// fp32-MFMA contraction in chunks of 64k base rows + running top-(k+16) + exact-order re-rank.
// Result (device): slots[nq][k] ascending by (distance, slot), dists[nq][k].
// `fused`: the f32 contractions (l2sq / cos) keep their distance matrix to themselves -- the tile epilogue appends what can still
// enter a query's top-kk to a candidate list, folded in after every launch -- once the first kSeedCols columns have given every
// query a radius the ordinary way.  *overflowed: a candidate list ran out of room (adversarially ordered rows): the caller repeats
// the search unfused.
static const size_t kSeedCols = 4096, kCandCap = 4096;
// lantern_gpu_dense_profile: HIP events around every launch of the fp32-MFMA contraction inside the exact k-NN (bench.py's
// c3_dense_exact_knn leg: the duration of steady full-chunk launches apart from the cold first ones and the partial last chunk).
// Off unless asked for; a process-wide diagnostic, not part of any index's state.
namespace {
struct DenseLaunch { hipEvent_t a = nullptr, b = nullptr; uint32_t rows = 0, cols = 0, fused = 0; };
struct DenseProfile
{
    std::mutex               mu;
    bool                     on = false;
    std::vector<DenseLaunch> launches;
} g_dense_profile;
struct DenseTimer  // records event `a` now and `b` on destruction, on the launch stream
{
    DenseLaunch l;
    hipStream_t st;
    bool        armed = false;
    DenseTimer(hipStream_t s, uint32_t rows, uint32_t cols, bool fused) : st(s)
    {
        if(!g_dense_profile.on) return;
        if(hipEventCreate(&l.a) != hipSuccess || hipEventCreate(&l.b) != hipSuccess) { (void)hipGetLastError(); return; }
        l.rows = rows; l.cols = cols; l.fused = fused;
        armed = hipEventRecord(l.a, st) == hipSuccess;
    }
    ~DenseTimer()
    {
        if(!armed) return;
        (void)hipEventRecord(l.b, st);
        std::lock_guard<std::mutex> g(g_dense_profile.mu);
        g_dense_profile.launches.push_back(l);
    }
};
}  // namespace

// The exact k-NN's device temporaries (norms + best lists, the seed-column matrix, the candidate lists) come from a small grow-only pool
// per device instead of hipMalloc / hipFree per call -- a free synchronises the device, and a 1024 x 1M x 768 call made five of them:
// about a millisecond of a 13.6 ms call.  Blocks above 64 MB (the whole-chunk matrix of the unfused path, the f32 copies of quantised
// rows) are still allocated per call.  One exact k-NN at a time per device holds the pool (the calls are bandwidth- or MFMA-bound on
// the whole chip: nothing is lost by not overlapping them).
namespace {
struct KnnPool
{
    std::mutex mu;
    void      *p[ 3 ] = { nullptr, nullptr, nullptr };
    size_t     bytes[ 3 ] = { 0, 0, 0 };
};
KnnPool g_knn_pool[ 16 ];
constexpr size_t kKnnPoolMax = (size_t)64 << 20;

struct KnnBlock  // pooled if small, allocated for the call otherwise; released by the destructor
{
    void *ptr = nullptr;
    bool  own = false;
    bool get(KnnPool *pool, int which, size_t need)
    {
        if(need == 0) need = 16;
        if(pool && need <= kKnnPoolMax) {
            if(pool->bytes[ which ] < need) {
                if(pool->p[ which ]) (void)hipFree(pool->p[ which ]);
                pool->p[ which ] = nullptr;
                pool->bytes[ which ] = 0;
                if(hipMalloc(&pool->p[ which ], need) != hipSuccess) { (void)hipGetLastError(); return false; }
                pool->bytes[ which ] = need;
            }
            ptr = pool->p[ which ];
            return true;
        }
        own = hipMalloc(&ptr, need) == hipSuccess;
        if(!own) { (void)hipGetLastError(); ptr = nullptr; }
        return own;
    }
    ~KnnBlock() { if(own && ptr) (void)hipFree(ptr); }
};
}  // namespace

static bool exact_knn_device_impl(int mcode, uint32_t chunks, const uint4 *d_base, size_t nb, const uint4 *d_q, size_t nq, size_t k,
                                  uint32_t *d_slots, float *d_dists, hipStream_t st, bool fused, bool *overflowed)
try {
    int dev = 0;
    (void)hipGetDevice(&dev);
    KnnPool *const               pool = (dev >= 0 && dev < 16) ? &g_knn_pool[ dev ] : nullptr;
    std::unique_lock<std::mutex> pool_lock;
    if(pool) pool_lock = std::unique_lock<std::mutex>(pool->mu);
    KnnBlock b_aux, b_dd, b_cand;
    // kk = k plus a margin: the MFMA distances (|q|^2 + |b|^2 - 2 q.b) differ from the exact-order ones in the
    // last bits, so the survivors are re-ranked exactly and only then cut to k
    const uint32_t kk = (uint32_t)k + 16;
    // column chunk.  [r3] 65536 columns x 1024 queries are 4096 tiles = 5.33 rounds of the 768 workgroups the contraction keeps
    // resident (three per CU): the last round runs a third full.  98304 columns (6144 tiles = 8 rounds) were tried: l2sq 14.9 -> 14.5 ms
    // per 1024 x 1M x 768 call, but cosine 15.4 -> 20.1 ms (the fused call repeated itself unfused), so the chunk stays.
    const size_t   QT = 1024, CH = std::min<size_t>(nb, 65536);
    const bool     i8 = mcode_is_i8(mcode);
    const bool     f16 = mcode_is_f16(mcode) || i8;  // "quantised storage": the contraction runs on an f32 copy
    const int      base_metric = mcode_base(mcode);
    const bool bits = base_metric == M_HAMMING || base_metric == M_COS_B1;  // popcount metrics: no MFMA contraction, no row norms
    if(bits || nb <= kSeedCols) fused = false;
    const uint32_t fchunks = i8 ? chunks * 4 : f16 ? chunks * 2 : chunks;  // chunks of the f32 view fed to the contraction
    auto dequant = [&](const uint4 *src, size_t nchunks, uint4 *dst) { return i8 ? launch_dequant_i8(src, nchunks, dst, st) : launch_dequant_f16(src, nchunks, dst, st); };
    char  *aux = nullptr;
    float *dd = nullptr;
    uint4 *fq = nullptr, *fb = nullptr;  // f32 copies of f16 rows (queries; one chunk of base rows)
    uint64_t *cand = nullptr;            // fused: [min(nq, QT)][kCandCap] keys, then the counters and the overflow flag
    const size_t nqt_max = std::min(nq, QT);
    // the distance matrix of the unfused launches: a whole chunk, or only the seed columns
    bool ok = b_aux.get(pool, 0, (nq + nb) * 4 + 8 + nq * kk * 8) && b_dd.get(pool, 1, nqt_max * (fused ? kSeedCols : CH) * 4);
    aux = (char *)b_aux.ptr;
    dd = (float *)b_dd.ptr;
    if(ok && fused) {
        ok = b_cand.get(pool, 2, nqt_max * kCandCap * 8 + nqt_max * 4 + 16);
        cand = (uint64_t *)b_cand.ptr;
        ok = ok && hipMemsetAsync((char *)cand + nqt_max * kCandCap * 8, 0, nqt_max * 4 + 16, st) == hipSuccess;
    }
    uint32_t *ccnt = cand ? (uint32_t *)((char *)cand + nqt_max * kCandCap * 8) : nullptr, *cover = ccnt ? ccnt + nqt_max : nullptr;
    if(ok && f16)
        ok = hipMalloc((void **)&fq, nq * (size_t)fchunks * 16) == hipSuccess && hipMalloc((void **)&fb, CH * (size_t)fchunks * 16) == hipSuccess &&
             dequant(d_q, nq * (size_t)chunks, fq) == hipSuccess;
    if(ok) {
        float    *qn = (float *)aux, *bn = qn + nq;
        uint64_t *best = (uint64_t *)(aux + (nq + nb) * 4 + ((nq + nb) % 2) * 4);
        const uint4 *qv = f16 ? fq : d_q;
        const uint32_t ldd = (uint32_t)(fused ? kSeedCols : CH);
        ok = ok && hipMemsetAsync(best, 0xFF, nq * kk * 8, st) == hipSuccess;
        if(!bits) ok = ok && launch_row_norms(qv, (uint32_t)nq, fchunks, qn, st) == hipSuccess;
        if(!bits && !f16) ok = ok && launch_row_norms(d_base, (uint32_t)nb, fchunks, bn, st) == hipSuccess;
        for(size_t c0 = 0; ok && c0 < nb; c0 += CH) {
            const size_t nc = std::min(CH, nb - c0);
            const uint4 *bv = d_base + c0 * chunks;
            if(f16) {
                ok = ok && dequant(d_base + c0 * chunks, nc * (size_t)chunks, fb) == hipSuccess;
                ok = ok && launch_row_norms(fb, (uint32_t)nc, fchunks, bn + c0, st) == hipSuccess;
                bv = fb;
            }
            // the columns of this chunk that go the ordinary way: all of them, or (fused) the seed columns of the first chunk
            const size_t plain = !fused ? nc : (c0 == 0 ? std::min(nc, kSeedCols) : 0);
            for(size_t q0 = 0; ok && q0 < nq; q0 += QT) {
                const size_t nqt = std::min(QT, nq - q0);
                if(plain) {
                    {
                        DenseTimer t(st, (uint32_t)nqt, (uint32_t)plain, false);
                        ok = ok && launch_dense(base_metric, qv + q0 * fchunks, (uint32_t)nqt, bv, (uint32_t)plain, fchunks, qn + q0, bn + c0, dd, ldd, st) == hipSuccess;
                    }
                    ok = ok && launch_select(dd, ldd, (uint32_t)nqt, (uint32_t)plain, (uint32_t)c0, best + q0 * kk, kk, st) == hipSuccess;
                }
                if(plain < nc) {
                    {
                        DenseTimer t(st, (uint32_t)nqt, (uint32_t)(nc - plain), true);
                        ok = ok && launch_dense_topk(base_metric, qv + q0 * fchunks, (uint32_t)nqt, bv + plain * fchunks, (uint32_t)(nc - plain), fchunks, qn + q0,
                                                     bn + c0 + plain, best + q0 * kk, kk, cand, ccnt, (uint32_t)kCandCap, (uint32_t)(c0 + plain), st) == hipSuccess;
                    }
                    ok = ok && launch_select_candidates((uint32_t)nqt, best + q0 * kk, kk, cand, ccnt, (uint32_t)kCandCap, cover, st) == hipSuccess;
                }
            }
        }
        ok = ok && launch_rerank(mcode, d_q, (uint32_t)nq, d_base, chunks, best, kk, (uint32_t)k, d_slots, d_dists, st) == hipSuccess;
        uint32_t over = 0;
        if(fused) ok = ok && hipMemcpyAsync(&over, cover, 4, hipMemcpyDeviceToHost, st) == hipSuccess;
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        if(overflowed) *overflowed = over != 0;
    }
    if(!ok) (void)hipStreamSynchronize(st);  // nothing queued may still name the pooled blocks when the pool's lock is released
    if(fq) (void)hipFree(fq);
    if(fb) (void)hipFree(fb);
    return ok;
}
LANTERN_ABI_CATCH(nullptr)

static bool exact_knn_device(int mcode, uint32_t chunks, const uint4 *d_base, size_t nb, const uint4 *d_q, size_t nq, size_t k,
                             uint32_t *d_slots, float *d_dists, hipStream_t st)
try {
    const char *env = std::getenv("LANTERN_GPU_DENSE_FUSED");  // =0: always the unfused path (A/B, tests)
    const bool  want_fused = !(env && std::atoi(env) == 0);
    bool over = false;
    if(!exact_knn_device_impl(mcode, chunks, d_base, nb, d_q, nq, k, d_slots, d_dists, st, want_fused, &over)) return false;
    if(!over) return true;
    return exact_knn_device_impl(mcode, chunks, d_base, nb, d_q, nq, k, d_slots, d_dists, st, false, nullptr);
}
LANTERN_ABI_CATCH(nullptr)

void lantern_gpu_exact_search(usearch_index_t h, const void *queries, size_t nq, size_t k, uint32_t *slots, float *distances,
                              usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix || nq == 0 || k == 0) return;
    if(k > 240) { FAIL(e, "lantern_gpu: exact_search supports k <= 240"); return; }
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix) || !pq_expand_locked(ix)) { FAIL(e, ix->err.c_str()); return; }
    const size_t n = ix->n, row_words = (size_t)ix->chunks * 4;
    if(n == 0) {
        for(size_t i = 0; i < nq * k; ++i) { slots[ i ] = EMPTY; distances[ i ] = INFINITY; }
        return;
    }
    const int    qkind = (ix->scalar == usearch_scalar_b1_k && !ix->b1_from_f32) ? usearch_scalar_b1_k : usearch_scalar_f32_k;  // queries arrive as f32 / bits
    const size_t in_bytes = input_bytes(ix, qkind);
    std::vector<uint32_t> padded(nq * row_words);
    for(size_t i = 0; i < nq; ++i) pad_row(ix, (const char *)queries + i * in_bytes, qkind, &padded[ i * row_words ]);
    uint4 *dq = (uint4 *)scratch(ix, 5, nq * row_words * 4);
    char  *dout = (char *)scratch(ix, 6, nq * k * 8 + 64);
    if(!dq || !dout) { FAIL(e, ix->err.c_str()); return; }
    uint32_t *d_slots = (uint32_t *)dout;
    float    *d_dists = (float *)(d_slots + nq * k);
    hipStream_t st = ix->stream;
    bool ok = hipMemcpyAsync(dq, padded.data(), nq * row_words * 4, hipMemcpyHostToDevice, st) == hipSuccess;
    ok = ok && exact_knn_device(ix->mcode, ix->chunks, ix->d_vec, n, dq, nq, k, d_slots, d_dists, st);
    ok = ok && hipMemcpy(slots, d_slots, nq * k * 4, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(distances, d_dists, nq * k * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if(!ok) FAIL(e, "lantern_gpu: HIP failure in exact_search");
}
LANTERN_ABI_CATCH_VOID(e)

// on != 0: start recording (forgets earlier records).  on == 0: stop, wait for the recorded launches and write up to `cap` of them:
// ms[i] = duration of launch i (HIP events on its stream), rows[i] x cols[i] = its queries x base rows, fused[i] = 1 for the
// launch with the fused top-k epilogue.  Returns the number of recorded launches.
size_t lantern_gpu_dense_profile(int on, float *ms, uint32_t *rows, uint32_t *cols, uint32_t *fused, size_t cap)
try {
    std::vector<DenseLaunch> got;
    {
        std::lock_guard<std::mutex> g(g_dense_profile.mu);
        if(on) {
            for(auto &l : g_dense_profile.launches) { (void)hipEventDestroy(l.a); (void)hipEventDestroy(l.b); }
            g_dense_profile.launches.clear();
            g_dense_profile.on = true;
            return 0;
        }
        g_dense_profile.on = false;
        got.swap(g_dense_profile.launches);
    }
    for(size_t i = 0; i < got.size(); ++i) {
        float t = 0.0f;
        if(hipEventSynchronize(got[ i ].b) != hipSuccess || hipEventElapsedTime(&t, got[ i ].a, got[ i ].b) != hipSuccess) { (void)hipGetLastError(); t = -1.0f; }
        if(i < cap) {
            if(ms) ms[ i ] = t;
            if(rows) rows[ i ] = got[ i ].rows;
            if(cols) cols[ i ] = got[ i ].cols;
            if(fused) fused[ i ] = got[ i ].fused;
        }
        (void)hipEventDestroy(got[ i ].a);
        (void)hipEventDestroy(got[ i ].b);
    }
    return got.size();
}
LANTERN_ABI_CATCH(nullptr)

// PQ k-means assignment (product_quantization.c:80-124 assign_to_clusters): the one dense N x k contraction in
// Lantern's C code -- N x k usearch_distance calls there, one fp32-MFMA pass + exact re-rank here.
void lantern_gpu_assign_to_clusters(const float *dataset, size_t n, size_t row_dims, size_t subvector_start, size_t subvector_dim,
                                    const float *centers, size_t k, usearch_metric_kind_t metric, uint32_t *out_cluster,
                                    float *out_distance, usearch_error_t *e)
try {
    CLEAR(e);
    if(metric != usearch_metric_cos_k && metric != usearch_metric_l2sq_k) { FAIL(e, "lantern_gpu: assign_to_clusters needs cos or l2sq"); return; }
    if(!dataset || !centers || !out_cluster || subvector_dim == 0 || subvector_start + subvector_dim > row_dims || k == 0) {
        FAIL(e, "lantern_gpu: bad arguments");
        return;
    }
    if(lantern_gpu_device_count() <= 0) { FAIL(e, kNoDevice); return; }
    if(n == 0) return;
    const size_t chunks = (subvector_dim + 3) / 4, rw = chunks * 4;
    std::vector<float> hp(n * rw, 0.f), hc(k * rw, 0.f);
    for(size_t i = 0; i < n; ++i) std::memcpy(&hp[ i * rw ], dataset + i * row_dims + subvector_start, subvector_dim * 4);
    for(size_t j = 0; j < k; ++j) std::memcpy(&hc[ j * rw ], centers + j * subvector_dim, subvector_dim * 4);
    void *dp = nullptr, *dc = nullptr, *dout = nullptr;
    bool  ok = hipMalloc(&dp, hp.size() * 4) == hipSuccess && hipMalloc(&dc, hc.size() * 4) == hipSuccess &&
              hipMalloc(&dout, n * 8) == hipSuccess;
    ok = ok && hipMemcpy(dp, hp.data(), hp.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    // first minimum wins (strict `<` in the reference loop) == smallest (distance, index)
    ok = ok && exact_knn_device((int)metric, (uint32_t)chunks, (const uint4 *)dc, k, (const uint4 *)dp, n, 1, (uint32_t *)dout,
                                (float *)((uint32_t *)dout + n), nullptr);
    ok = ok && hipMemcpy(out_cluster, dout, n * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if(out_distance) ok = ok && hipMemcpy(out_distance, (uint32_t *)dout + n, n * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if(dp) (void)hipFree(dp);
    if(dc) (void)hipFree(dc);
    if(dout) (void)hipFree(dout);
    if(!ok) FAIL(e, "lantern_gpu: HIP failure in assign_to_clusters");
}
LANTERN_ABI_CATCH_VOID(e)

// ---- SQL-callable semantics (hnsw.c:296-405) ---------------------------------------------------------

static thread_local char g_dim_msg[ 160 ];

static bool same_dims(int a_dim, int b_dim, usearch_error_t *e)
try {
    if(a_dim == b_dim) return true;
    // hnsw.c:301-303
    std::snprintf(g_dim_msg, sizeof(g_dim_msg), "expected equally sized arrays but got arrays with dimensions %d and %d", a_dim, b_dim);
    FAIL(e, g_dim_msg);
    return false;
}
LANTERN_ABI_CATCH(e)

float lantern_l2sq_dist(const float *a, int a_dim, const float *b, int b_dim, usearch_error_t *e)
try {
    CLEAR(e);
    if(!same_dims(a_dim, b_dim, e)) return 0.f;
    return usearch_distance(a, b, usearch_scalar_f32_k, (size_t)a_dim, usearch_metric_l2sq_k, e);
}
LANTERN_ABI_CATCH(e)

float lantern_cos_dist(const float *a, int a_dim, const float *b, int b_dim, usearch_error_t *e)
try {
    CLEAR(e);
    if(!same_dims(a_dim, b_dim, e)) return 0.f;
    return usearch_distance(a, b, usearch_scalar_f32_k, (size_t)a_dim, usearch_metric_cos_k, e);
}
LANTERN_ABI_CATCH(e)

int32_t lantern_hamming_dist(const int32_t *a, int a_dim, const int32_t *b, int b_dim, usearch_error_t *e)
try {
    CLEAR(e);
    if(!same_dims(a_dim, b_dim, e)) return 0;
    // hnsw.c:317-319: dims = a_dim * sizeof(int32) * CHAR_BIT bits; result cast to int32 (hnsw.c:375)
    return (int32_t)usearch_distance(a, b, usearch_scalar_b1_k, (size_t)a_dim * 32, usearch_metric_hamming_k, e);
}
LANTERN_ABI_CATCH(e)

// ---- metadata / counters / graph exchange --------------------------------------------------------------

metadata_t usearch_index_metadata(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    metadata_t m;
    std::memset(&m, 0, sizeof(m));
    Index *ix = H(h, e);
    if(!ix) return m;
    m.neighbors_bytes = 4 + (size_t)ix->M * LANTERN_SLOT_SIZE;        // [count u32][M x 6-byte slots]
    m.neighbors_base_bytes = 4 + (size_t)ix->M0 * LANTERN_SLOT_SIZE;  // level 0: 2M slots
    m.inverse_log_connectivity = 1.0 / std::log((double)ix->M);
    m.connectivity = ix->M;
    m.dimensions = ix->opts.dimensions;
    m.init_options = ix->opts;
    return m;
}
LANTERN_ABI_CATCH(e)

lantern_gpu_counters lantern_gpu_counters_get(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    lantern_gpu_counters c;
    std::memset(&c, 0, sizeof(c));
    Index *ix = H(h, e);
    if(!ix) return c;
    unsigned long long t[ 8 ] = {};
    if(hipMemcpy(t, ix->d_totals, sizeof(t), hipMemcpyDeviceToHost) != hipSuccess) { FAIL(e, "lantern_gpu: HIP failure reading counters"); return c; }
    c.search_dist_evals = t[ 0 ];
    c.search_expansions = t[ 1 ];
    c.search_queries = ix->c_search_queries;
    c.add_dist_evals = t[ 2 ] + t[ 4 ] + t[ 5 ];
    c.add_expansions = t[ 3 ];
    c.add_vectors = ix->c_add_vectors;
    c.add_batches = ix->c_add_batches;
    c.add_walk_evals = t[ 2 ];
    c.add_select_evals = t[ 4 ];
    c.add_revlink_evals = t[ 5 ];
    c.add_reprunes = t[ 6 ];
    c.search_solo_launches = ix->c_solo_launches;
    return c;
}
LANTERN_ABI_CATCH(e)

// Diagnostics: with `on`, searches run the instrumented instantiation of the walk kernel (f32 l2sq / cos at G = 64 or 16 only)
// and accumulate shader-clock cycles per phase; out[8] = visited filter + compaction | wait at the first barrier | distances |
// merge | pop | neighbour-list arrival | upper-level descent | whole query.
void lantern_gpu_search_phase_profile(usearch_index_t h, int on, unsigned long long *out6 /* [8] */, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(out6) {
        (void)hipDeviceSynchronize();
        if(hipMemcpy(out6, ix->d_totals + 8, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) FAIL(e, "lantern_gpu: HIP failure reading the phase profile");
        (void)hipMemset(ix->d_totals + 8, 0, 8 * sizeof(unsigned long long));
    }
    ix->phase_profile = on != 0;
    ix->unique_rows_on = false;  // the row bitmap belongs to lantern_gpu_search_unique_rows alone
}
LANTERN_ABI_CATCH_VOID(e)

// Diagnostics: the number of DISTINCT rows the searches launched between `on` and the read evaluated -- over all their queries.
// A launch has to bring each of them in from HBM at least once, whatever the caches do with the re-reads: unique rows x row
// bytes is the cold-miss LOWER bound of its DRAM traffic, beside the counters' fabric-side bytes (which include Infinity-Cache
// hits) and the algorithmic bytes (one row per evaluation).  Runs the instrumented instantiation of the walk (as
// lantern_gpu_search_phase_profile: f32 l2sq / cos rows of >= 128 or 32..63 chunks); same walk, same D and E.
//   on = 1: zero the row bitmap and switch the instrumented kernel on;  on = 0 with `rows`: read the count, switch it off.
void lantern_gpu_search_unique_rows(usearch_index_t h, int on, uint64_t *rows, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return; }
    const size_t words = (std::max<size_t>(ix->cap, 1) + 31) / 32;
    (void)hipDeviceSynchronize();
    if(rows) {
        *rows = 0;
        if(ix->d_touched) {
            const size_t have = std::min(words, ix->touched_words);  // the index may have grown since the bitmap was sized
            std::vector<uint32_t> bits(have);
            if(have && hipMemcpy(bits.data(), ix->d_touched, have * 4, hipMemcpyDeviceToHost) != hipSuccess) { FAIL(e, "lantern_gpu: HIP failure reading the row bitmap"); return; }
            uint64_t n = 0;
            for(uint32_t w : bits) n += (uint64_t)__builtin_popcount(w);
            *rows = n;
        }
    }
    if(on) {
        const int G_ = group_lanes_for(ix->chunks);
        if(!((ix->mcode == M_L2SQ && (G_ == 64 || G_ == 16)) || (ix->mcode == M_COS && G_ == 64))) {
            FAIL(e, "lantern_gpu: the instrumented walk exists for f32 l2sq (rows of >= 128 or 32..63 chunks) and f32 cos (>= 128 chunks) only");
            return;
        }
        if(ix->d_touched && ix->touched_words < words) { (void)hipFree(ix->d_touched); ix->d_touched = nullptr; }
        if(!ix->d_touched) {
            if(hipMalloc((void **)&ix->d_touched, words * 4) != hipSuccess) { ix->d_touched = nullptr; FAIL(e, "lantern_gpu: out of device memory (row bitmap)"); return; }
            ix->touched_words = words;
        }
        if(hipMemset(ix->d_touched, 0, ix->touched_words * 4) != hipSuccess) { FAIL(e, "lantern_gpu: HIP failure clearing the row bitmap"); return; }
    }
    ix->phase_profile = on != 0;
    ix->unique_rows_on = on != 0;
}
LANTERN_ABI_CATCH_VOID(e)

// The memory objects every query of a launch asks for, in order: the input of the cache model behind bench.py's frac_dram_model
// (lantern_amd/tools/cache_model.c).  Runs the instrumented instantiation of the walk (as lantern_gpu_search_unique_rows); same walk,
// same D and E.
//   on = 1: allocate [nq][per_query_cap] entries and switch the instrumented, tracing kernel on for launches of <= nq queries;
//   on = 0: wait for the device, copy the LAST traced launch's trace (nq x per_query_cap u32, row-major) and counts (nq u32; a
//           count above per_query_cap means the tail of that query's trace was dropped) to the host buffers (either may be NULL),
//           free the device buffers and switch the instrumented kernel off.
// An entry is a row's slot (a distance evaluation), slot | 0x80000000 (the node's level-0 adjacency list was read) or
// slot | 0xC0000000 (an upper-level list); indexes of at most 2^30 slots.
void lantern_gpu_search_row_trace(usearch_index_t h, int on, size_t nq, size_t per_query_cap, uint32_t *trace, uint32_t *counts, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return; }
    (void)hipDeviceSynchronize();
    if(on) {
        const int G_ = group_lanes_for(ix->chunks);
        if(!((ix->mcode == M_L2SQ && (G_ == 64 || G_ == 16)) || (ix->mcode == M_COS && G_ == 64))) {
            FAIL(e, "lantern_gpu: the instrumented walk exists for f32 l2sq (rows of >= 128 or 32..63 chunks) and f32 cos (>= 128 chunks) only");
            return;
        }
        if(ix->cap > ((size_t)1 << 30) || nq == 0 || per_query_cap == 0) { FAIL(e, "lantern_gpu: row trace: bad arguments, or an index above 2^30 slots"); return; }
        if(ix->d_trace) { (void)hipFree(ix->d_trace); ix->d_trace = nullptr; }
        if(ix->d_trace_count) { (void)hipFree(ix->d_trace_count); ix->d_trace_count = nullptr; }
        if(hipMalloc((void **)&ix->d_trace, nq * per_query_cap * 4) != hipSuccess || hipMalloc((void **)&ix->d_trace_count, nq * 4) != hipSuccess ||
           hipMemset(ix->d_trace_count, 0, nq * 4) != hipSuccess) {
            (void)hipGetLastError();
            if(ix->d_trace) (void)hipFree(ix->d_trace);
            if(ix->d_trace_count) (void)hipFree(ix->d_trace_count);
            ix->d_trace = ix->d_trace_count = nullptr;
            FAIL(e, "lantern_gpu: out of device memory (row trace)");
            return;
        }
        ix->trace_nq = nq;
        ix->trace_cap = per_query_cap;
        ix->trace_on = true;
        ix->phase_profile = true;
        return;
    }
    bool ok = true;
    if(ix->d_trace && ix->d_trace_count) {
        const size_t m = std::min(nq, ix->trace_nq);
        if(counts && m) ok = hipMemcpy(counts, ix->d_trace_count, m * 4, hipMemcpyDeviceToHost) == hipSuccess;
        if(ok && trace && m && per_query_cap == ix->trace_cap) ok = hipMemcpy(trace, ix->d_trace, m * ix->trace_cap * 4, hipMemcpyDeviceToHost) == hipSuccess;
        else if(trace && m && per_query_cap != ix->trace_cap) ok = false;
    }
    if(ix->d_trace) (void)hipFree(ix->d_trace);
    if(ix->d_trace_count) (void)hipFree(ix->d_trace_count);
    ix->d_trace = ix->d_trace_count = nullptr;
    ix->trace_nq = ix->trace_cap = 0;
    ix->trace_on = false;
    ix->phase_profile = ix->unique_rows_on;
    if(!ok) FAIL(e, "lantern_gpu: HIP failure reading the row trace (or per_query_cap differs from the one the trace was started with)");
}
LANTERN_ABI_CATCH_VOID(e)

float lantern_gpu_last_gather_ms(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return 0.f;
    std::lock_guard<std::mutex> g(ix->mu);
    return ix->last_gather_ms;
}
LANTERN_ABI_CATCH(e)

// workgroups of the last search launch (k_search / k_search_spec): the number of walks resident at a time -- the cache model's `walkers`
int lantern_gpu_last_search_grid(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return 0;
    std::lock_guard<std::mutex> g(ix->mu);
    return ix->last_search_grid;
}
LANTERN_ABI_CATCH(e)

// the instrumented latency-bound walk (walk_spec.hpp PROF): out32[8 * wave + i], waves 0..3 = visit | list | fill | a row wave;
// i: 0 decision, 1 neighbour list, 2 issue, 3 role section, 4 loads + distances, 5 barrier wait, 6 hops, 7 list source count
// (wave 0: staging area, wave 3: cache, wave 2: HBM).  Reading resets the counters.
void lantern_gpu_spec_profile(usearch_index_t h, int on, unsigned long long *out32, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(out32) {
        (void)hipDeviceSynchronize();
        if(hipMemcpy(out32, ix->d_totals + 16, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) FAIL(e, "lantern_gpu: HIP failure reading the profile");
        (void)hipMemset(ix->d_totals + 16, 0, 32 * sizeof(unsigned long long));
    }
    ix->spec_profile = on != 0;
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_set_profiling(usearch_index_t h, int on, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    ix->profiling = on != 0;
}
LANTERN_ABI_CATCH_VOID(e)

lantern_gpu_build_profile lantern_gpu_build_profile_get(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    lantern_gpu_build_profile out;
    std::memset(&out, 0, sizeof(out));
    Index *ix = H(h, e);
    if(!ix) return out;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return out; }
    prof_resolve(ix, 0);
    return ix->prof;
}
LANTERN_ABI_CATCH(e)

lantern_gpu_graph_info lantern_gpu_graph_info_get(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    lantern_gpu_graph_info gi;
    std::memset(&gi, 0, sizeof(gi));
    Index *ix = H(h, e);
    if(!ix) return gi;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return gi; }
    gi.size = ix->n;
    gi.upper_blocks = ix->upper_blocks;
    gi.connectivity = ix->M;
    gi.entry_slot = ix->entry;
    gi.max_level = ix->max_level;
    gi.vector_words = ix->words;
    return gi;
}
LANTERN_ABI_CATCH(e)

void lantern_gpu_export_graph(usearch_index_t h, uint8_t *levels, uint32_t *nbr0, uint32_t *upper_off, uint32_t *upper_nbr,
                              uint64_t *labels, void *vectors, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return; }
    const size_t n = ix->n;
    bool         ok = true;
    if(levels && n) std::memcpy(levels, ix->levels.data(), n);
    if(upper_off && n) std::memcpy(upper_off, ix->upper_off.data(), n * 4);
    if(labels && n) std::memcpy(labels, ix->labels.data(), n * 8);
    if(nbr0 && n) ok = ok && hipMemcpy(nbr0, ix->d_nbr0, n * ix->M0 * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if(upper_nbr && ix->upper_blocks)
        ok = ok && hipMemcpy(upper_nbr, ix->d_upper_nbr, ix->upper_blocks * ix->M * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if(vectors && n && !pq_expand_locked(ix)) { FAIL(e, ix->err.c_str()); return; }
    if(vectors && n) {
        if(ix->chunks * 4 == ix->words) {
            ok = ok && hipMemcpy(vectors, ix->d_vec, n * (size_t)ix->words * 4, hipMemcpyDeviceToHost) == hipSuccess;
        } else {
            ok = ok && hipMemcpy2D(vectors, (size_t)ix->words * 4, ix->d_vec, (size_t)ix->chunks * 16, (size_t)ix->words * 4, n,
                                   hipMemcpyDeviceToHost) == hipSuccess;
        }
    }
    if(!ok) FAIL(e, "lantern_gpu: HIP failure exporting the graph");
}
LANTERN_ABI_CATCH_VOID(e)

// pq = true indexes: drop / restore the decoded rows (index.hpp pq_compact)
void lantern_gpu_pq_compact(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!pq_compact_locked(ix)) FAIL(e, ix->err.c_str());
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_pq_expand(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!pq_expand_locked(ix)) FAIL(e, ix->err.c_str());
}
LANTERN_ABI_CATCH_VOID(e)

// HBM held by the index: the vector block (or the code rows of a compact pq index) | everything else that grows with the
// number of nodes (adjacency, labels, levels, norms, re-prune state, codes)
// bytes of one STORED row in device memory (16-byte chunks; bit rows of 65 .. 127 bytes padded to 128): the stride of the `d_queries`
// a caller hands to lantern_gpu_search_batch_device
size_t lantern_gpu_row_bytes(usearch_index_t h, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    return ix ? (size_t)ix->chunks * 16 : 0;
}
LANTERN_ABI_CATCH(e)

void lantern_gpu_memory_usage(usearch_index_t h, size_t *row_bytes, size_t *other_bytes, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    const size_t cap = ix->cap;
    if(row_bytes) *row_bytes = ix->pq_compact ? std::max<size_t>(ix->n, 1) * ix->pq_S16 : (ix->d_vec ? cap * (size_t)ix->chunks * 16 : 0);
    if(other_bytes)
        *other_bytes = cap * ((size_t)ix->M0 * 4 + 8 + 1 + 4 + 4) + ix->upper_cap * ((size_t)ix->M * 4 + 4) + (ix->d_norm2 ? cap * 4 : 0) +
                       (ix->pq ? cap * (size_t)ix->pq_S : 0);
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_export_codes(usearch_index_t h, uint8_t *codes, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { FAIL(e, ix->err.c_str()); return; }
    if(!ix->pq) { FAIL(e, "lantern_gpu: not a pq index"); return; }
    if(ix->n && hipMemcpy(codes, ix->d_codes, ix->n * (size_t)ix->pq_S, hipMemcpyDeviceToHost) != hipSuccess) FAIL(e, "lantern_gpu: HIP failure exporting the codes");
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_import_graph(usearch_index_t h, size_t size, const void *vectors, const uint64_t *labels, const uint8_t *levels,
                              const uint32_t *nbr0, const uint32_t *upper_off, const uint32_t *upper_nbr, uint32_t entry_slot,
                              int32_t max_level, usearch_error_t *e)
try {
    CLEAR(e);
    Index *ix = H(h, e);
    if(!ix) return;
    std::lock_guard<std::mutex> g(ix->mu);
    if(!import_graph_locked(ix, size, vectors, labels, levels, nbr0, upper_off, upper_nbr, entry_slot, max_level)) FAIL(e, ix->err.c_str());
}
LANTERN_ABI_CATCH_VOID(e)

}  // extern "C"
