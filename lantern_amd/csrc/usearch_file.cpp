// usearch_file.cpp -- usearch-format serialisation of the device index.
//
// What Lantern consumes (lantern_hnsw/src/hnsw/external_index.c:298-372, usearch_storage.cpp:19-118,
// validate_index.c:105-226, external_index.h:29-66):
//
//   bytes [0, 136)   opaque usearch header = 80-byte file header + 56-byte graph header
//   then             node tapes back to back, node i is the i-th tape (external_index.c:137,161-164)
//   node tape        [label u64][level u16] { [count u32][cap x 6-byte slot] } x (level+1) [vector]
//                    cap = 2M on level 0, M above (validate_index.c:140-151); unused slots zero;
//                    a slot carries the neighbour's u32 sequential id in its low 4 bytes
//                    (external_index.c:399-403 rewrites them to ItemPointers on import)
//
// The 136 header bytes are parsed only inside usearch (usearch_view_mem_lazy, usearch_update_header,
// usearch_header_get/set_entry_slot), whose source is not in the reference tree.  Field OFFSETS and VALUES below
// follow upstream usearch 2.x:
//
//   index_dense_head_t (index_dense.hpp), packed, no alignment:
//     [ 0,  7)  magic "usearch"
//     [ 7, 13)  version_major, version_minor, version_patch          u16 x 3   (load refuses another MAJOR)
//     [13]      kind_metric           metric_kind_t  -- ASCII codes: cos 'c', l2sq 'e', hamming 'b', ip 'i' ...
//     [14]      kind_scalar           scalar_kind_t  -- b1x8 1, u40 2, uuid 3, f64 10, f32 11, f16 12, f8 13,
//                                                       u64 14, u32 15, u16 16, u8 17, i64 20, i32 21, i16 22, i8 23
//     [15]      kind_key              scalar_kind_t of the key type      (u64 for usearch_label_t: 14)
//     [16]      kind_compressed_slot  scalar_kind_t of the slot type     (load refuses a mismatch; see below)
//     [17, 25)  count_present   u64
//     [25, 33)  count_deleted   u64
//     [33, 41)  dimensions      u64
//     [41]      multi           bool
//     upstream reserves 64 bytes for this block; Lantern's fork 80 (its PQ fields; zero here)
//   index_serialized_header_t (index.hpp), five u64: size, connectivity, connectivity_base, max_level, entry_slot;
//     upstream 40 bytes, the fork 56 (external_index.h:59-66) -- at [80, 136) here
//
// (The C-API numerals of include/lantern_gpu.h -- cos 1, l2sq 3, f32 1 ... -- are a different enum: the C layer
// converts them, as usearch_storage.cpp:44-60 does for the scalar kind.)  kind_compressed_slot: the fork's slot type is
// the 48-bit lantern_slot_t (usearch_storage.cpp:16); upstream's 40-bit slot is scalar_kind_t::u40_k and a type with
// no scalar_kind<> specialisation reports unknown_k -- which of the two the fork stores cannot be read off the tree, so
// u40_k ("the wide custom slot") is written and ANY value is accepted on read.
// STATUS: UNVERIFIED against the pinned fork (rev aa4f91d); no fixture of a header produced by real usearch can be
// made offline.  The reader accepts both upstream codes and this library's round-1 numerals.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <unordered_map>
#include <vector>

#include "index.hpp"
#include "abi_guard.hpp"

namespace lgpu {

namespace {
constexpr size_t OFF_MAGIC = 0;        // "usearch" (7 bytes)
constexpr size_t OFF_VERSION = 7;      // u16 x 3
constexpr size_t OFF_KIND_METRIC = 13; // u8
constexpr size_t OFF_KIND_SCALAR = 14; // u8
constexpr size_t OFF_KIND_KEY = 15;    // u8
constexpr size_t OFF_KIND_SLOT = 16;   // u8
constexpr size_t OFF_COUNT_PRESENT = 17;  // u64
constexpr size_t OFF_COUNT_DELETED = 25;  // u64
constexpr size_t OFF_DIMENSIONS = 33;     // u64
constexpr size_t OFF_MULTI = 41;          // u8
// [42, 80): padding + the fork's PQ fields (zero)
constexpr size_t OFF_G_SIZE = 80;               // u64
constexpr size_t OFF_G_CONNECTIVITY = 88;       // u64
constexpr size_t OFF_G_CONNECTIVITY_BASE = 96;  // u64
constexpr size_t OFF_G_MAX_LEVEL = 104;         // u64
constexpr size_t OFF_G_ENTRY_SLOT = 112;        // u64
// [120, 136): reserved (zero)

// upstream metric_kind_t / scalar_kind_t codes (index_plugins.hpp)
constexpr uint8_t SK_B1X8 = 1, SK_U40 = 2, SK_F64 = 10, SK_F32 = 11, SK_F16 = 12, SK_U64 = 14, SK_I8 = 23;
uint8_t metric_code(int metric) { return metric == usearch_metric_cos_k ? 'c' : metric == usearch_metric_l2sq_k ? 'e' : metric == usearch_metric_hamming_k ? 'b' : 0; }
uint8_t scalar_code(int scalar)
{
    switch(scalar) {
        case usearch_scalar_f32_k: return SK_F32;
        case usearch_scalar_f64_k: return SK_F64;
        case usearch_scalar_f16_k: return SK_F16;
        case usearch_scalar_i8_k: return SK_I8;
        case usearch_scalar_b1_k: return SK_B1X8;
        default: return 0;
    }
}
// a header written by upstream-coded usearch OR by round 1 of this library (C-API numerals)
bool metric_matches(uint8_t stored, int metric) { return stored == metric_code(metric) || stored == (uint8_t)metric; }
bool scalar_matches(uint8_t stored, int scalar) { return stored == scalar_code(scalar) || stored == (uint8_t)scalar; }

template <typename T> void put(char *p, size_t off, T v) { std::memcpy(p + off, &v, sizeof(T)); }
template <typename T> T    get(const char *p, size_t off) { T v; std::memcpy(&v, p + off, sizeof(T)); return v; }

// bytes of one stored vector on the tape: dimensions * bits_per_scalar / 8 (usearch_storage.cpp:63-81)
// pq: num_subvectors code bytes, "assuming at most 2 ** 8 centroids (= 1 byte) per subvector" (usearch_storage.cpp:29-31)
size_t vector_bytes(const Index *ix) { return ix->pq ? (size_t)ix->pq_S : input_bytes(ix, ix->scalar); }
size_t node_bytes(const Index *ix, int level)
{
    return 8 + 2 + (4 + (size_t)ix->M0 * LANTERN_SLOT_SIZE) + (size_t)level * (4 + (size_t)ix->M * LANTERN_SLOT_SIZE) + vector_bytes(ix);
}
}  // namespace

size_t serialized_length(Index *ix)
{
    size_t total = USEARCH_HEADER_SIZE;
    for(size_t i = 0; i < ix->n; ++i) total += node_bytes(ix, ix->levels[ i ]);
    return total;
}

static void write_header(const Index *ix, char *h)
{
    std::memset(h, 0, USEARCH_HEADER_SIZE);
    std::memcpy(h + OFF_MAGIC, "usearch", 7);
    put<uint16_t>(h, OFF_VERSION, 2);  // usearch 2.x: load refuses a different major
    put<uint16_t>(h, OFF_VERSION + 2, 8);
    put<uint16_t>(h, OFF_VERSION + 4, 15);
    put<uint8_t>(h, OFF_KIND_METRIC, metric_code(ix->metric));
    put<uint8_t>(h, OFF_KIND_SCALAR, scalar_code(ix->scalar));
    put<uint8_t>(h, OFF_KIND_KEY, SK_U64);   // usearch_label_t = u64
    put<uint8_t>(h, OFF_KIND_SLOT, SK_U40);  // the wide custom slot (see the file comment)
    put<uint64_t>(h, OFF_COUNT_PRESENT, ix->n);
    put<uint64_t>(h, OFF_COUNT_DELETED, 0);
    put<uint64_t>(h, OFF_DIMENSIONS, ix->opts.dimensions);
    put<uint8_t>(h, OFF_MULTI, 0);
    put<uint64_t>(h, OFF_G_SIZE, ix->n);
    put<uint64_t>(h, OFF_G_CONNECTIVITY, ix->M);
    put<uint64_t>(h, OFF_G_CONNECTIVITY_BASE, ix->M0);
    put<uint64_t>(h, OFF_G_MAX_LEVEL, ix->n ? (uint64_t)ix->max_level : 0);
    put<uint64_t>(h, OFF_G_ENTRY_SLOT, ix->n ? (uint64_t)ix->entry : 0);
}

// The file as a stream of spans: header, then per node a formatted prefix (label, level, neighbour lists) and its vector bytes
// straight out of a page-locked staging buffer.  Rows leave the device in chunks of ~64 MB on a stream of their own, two buffers
// deep, so the copy of chunk c + 1 runs while chunk c is formatted and consumed; nothing the size of the index is ever
// allocated on the host (the r2 form built a pageable copy of all rows and then the whole file in memory: 5 of the 9 seconds
// of a 1M x 1536 build through the indexing server).  `sink` gets up to 1024 spans per call and returns false to abort.
bool serialize_stream(Index *ix, const SpanSink &sink)
{
    char header[ USEARCH_HEADER_SIZE ];
    write_header(ix, header);
    {
        const lantern_gpu_span h = { header, USEARCH_HEADER_SIZE };
        if(!sink(&h, 1)) { set_err(ix, "lantern_gpu: the serialisation sink failed"); return false; }
    }
    const size_t n = ix->n;
    if(n == 0) return true;
    const size_t row = (size_t)ix->chunks * 16, vb = vector_bytes(ix);
    const size_t row_stride = ix->pq ? (size_t)ix->pq_S : row;
    const char  *d_rows = ix->pq ? (const char *)ix->d_codes : (const char *)ix->d_vec;
    std::vector<uint32_t> nbr0(n * ix->M0), upper(ix->upper_blocks * ix->M + 1);
    bool ok = hipMemcpy(nbr0.data(), ix->d_nbr0, nbr0.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if(ix->upper_blocks) ok = ok && hipMemcpy(upper.data(), ix->d_upper_nbr, ix->upper_blocks * ix->M * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if(!ok) { set_err(ix, "lantern_gpu: HIP failure while serialising"); return false; }

    size_t chunk_bytes = (size_t)64 << 20;
    if(const char *cb = std::getenv("LANTERN_GPU_SAVE_CHUNK_BYTES")) chunk_bytes = std::max<size_t>(1, (size_t)std::strtoull(cb, nullptr, 10));  // (tests: many small chunks)
    const size_t per = std::min(n, std::max<size_t>(1, chunk_bytes / std::max<size_t>(row_stride, 1)));
    const size_t max_prefix = 8 + 2 + (4 + (size_t)ix->M0 * LANTERN_SLOT_SIZE) + (size_t)255 * (4 + (size_t)ix->M * LANTERN_SLOT_SIZE);
    char        *stage[ 2 ] = { nullptr, nullptr };
    hipStream_t  st = nullptr;
    hipEvent_t   ev[ 2 ] = { nullptr, nullptr };
    std::vector<char>             prefix;
    std::vector<lantern_gpu_span> spans;
    const char                   *fail = nullptr;
    ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    for(int b = 0; ok && b < 2; ++b)
        ok = hipHostMalloc((void **)&stage[ b ], per * row_stride, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&ev[ b ], hipEventDisableTiming) == hipSuccess;
    auto issue = [&](size_t c) {  // chunk c -> stage[c & 1]
        const size_t i0 = c * per, cnt = std::min(per, n - i0);
        return hipMemcpyAsync(stage[ c & 1 ], d_rows + i0 * row_stride, cnt * row_stride, hipMemcpyDeviceToHost, st) == hipSuccess &&
               hipEventRecord(ev[ c & 1 ], st) == hipSuccess;
    };
    const size_t nchunks = (n + per - 1) / per;
    if(!ok || !issue(0)) fail = "lantern_gpu: HIP failure while serialising";
    for(size_t c = 0; !fail && c < nchunks; ++c) {
        if(c + 1 < nchunks && !issue(c + 1)) { fail = "lantern_gpu: HIP failure while serialising"; break; }  // (its buffer was consumed with chunk c - 1)
        const size_t i0 = c * per, cnt = std::min(per, n - i0);
        // prefixes of the chunk's nodes while its rows are still on their way
        size_t need = 0;
        for(size_t i = i0; i < i0 + cnt; ++i) need += node_bytes(ix, ix->levels[ i ]) - vb;
        if(need > cnt * max_prefix) { fail = "lantern_gpu: node level out of range while serialising"; break; }
        prefix.assign(need, 0);
        spans.resize(cnt * 2);
        char *p = prefix.data();
        for(size_t i = i0; i < i0 + cnt; ++i) {
            const int level = ix->levels[ i ];
            char     *q = p + 10;
            put<uint64_t>(p, 0, ix->labels[ i ]);
            put<uint16_t>(p, 8, (uint16_t)level);
            for(int l = 0; l <= level; ++l) {
                const uint32_t  cap = l == 0 ? ix->M0 : ix->M;
                const uint32_t *list = l == 0 ? &nbr0[ i * ix->M0 ] : &upper[ ((size_t)ix->upper_off[ i ] + (size_t)(l - 1)) * ix->M ];
                uint32_t        k = 0;
                while(k < cap && list[ k ] != EMPTY) ++k;
                put<uint32_t>(q, 0, k);
                for(uint32_t j = 0; j < k; ++j) put<uint32_t>(q, 4 + (size_t)j * LANTERN_SLOT_SIZE, list[ j ]);  // low 4 of 6 bytes
                q += 4 + (size_t)cap * LANTERN_SLOT_SIZE;
            }
            spans[ (i - i0) * 2 ] = { p, (size_t)(q - p) };
            spans[ (i - i0) * 2 + 1 ] = { stage[ c & 1 ] + (i - i0) * row_stride, vb };
            p = q;
        }
        if(hipEventSynchronize(ev[ c & 1 ]) != hipSuccess) { fail = "lantern_gpu: HIP failure while serialising"; break; }
        for(size_t o = 0; o < spans.size() && !fail; o += 1024)
            if(!sink(&spans[ o ], std::min<size_t>(1024, spans.size() - o))) fail = "lantern_gpu: the serialisation sink failed";
    }
    if(st) (void)hipStreamSynchronize(st);  // (an aborted run may still have a copy in flight into a buffer about to be freed)
    for(int b = 0; b < 2; ++b) {
        if(ev[ b ]) (void)hipEventDestroy(ev[ b ]);
        if(stage[ b ]) (void)hipHostFree(stage[ b ]);
    }
    if(st) (void)hipStreamDestroy(st);
    if(fail) { set_err(ix, fail); return false; }
    return true;
}

bool serialize(Index *ix, char *buf, size_t len)
{
    const size_t need = serialized_length(ix);
    if(len < need) { set_err(ix, "lantern_gpu: serialisation buffer too small"); return false; }
    char *p = buf;
    return serialize_stream(ix, [&](const lantern_gpu_span *sp, size_t cnt) {
        for(size_t i = 0; i < cnt; ++i) {
            std::memcpy(p, sp[ i ].data, sp[ i ].size);
            p += sp[ i ].size;
        }
        return true;
    });
}

bool deserialize(Index *ix, const char *buf, size_t len)
{
    if(len < USEARCH_HEADER_SIZE || std::memcmp(buf + OFF_MAGIC, "usearch", 7) != 0) { set_err(ix, "lantern_gpu: not a usearch index file"); return false; }
    const uint64_t n = get<uint64_t>(buf, OFF_G_SIZE);
    if(get<uint64_t>(buf, OFF_G_CONNECTIVITY) != ix->M || get<uint64_t>(buf, OFF_DIMENSIONS) != ix->opts.dimensions ||
       !metric_matches(get<uint8_t>(buf, OFF_KIND_METRIC), ix->metric) || !scalar_matches(get<uint8_t>(buf, OFF_KIND_SCALAR), ix->scalar)) {
        set_err(ix, "lantern_gpu: index file does not match the index options (metric, scalar kind, dimensions or connectivity)");
        return false;
    }
    if(n == 0) return true;
    // the file is untrusted input (the index server hands back what a socket peer built; usearch_load takes any path):
    // every count is bounded by what `len` can hold BEFORE anything is allocated from it
    if(get<uint64_t>(buf, OFF_G_CONNECTIVITY_BASE) != ix->M0) { set_err(ix, "lantern_gpu: index file's level-0 connectivity is not 2 x connectivity"); return false; }
    if(n > (len - USEARCH_HEADER_SIZE) / node_bytes(ix, 0) || n >= 0x7FFFFFFFull) { set_err(ix, "lantern_gpu: index file declares more nodes than it can hold"); return false; }
    const uint64_t entry = get<uint64_t>(buf, OFF_G_ENTRY_SLOT), top = get<uint64_t>(buf, OFF_G_MAX_LEVEL);
    if(entry >= n || top > 255) { set_err(ix, "lantern_gpu: index file's entry slot or top level is out of range"); return false; }
    const size_t vb = vector_bytes(ix);
    std::vector<uint64_t> labels(n);
    std::vector<uint8_t>  levels(n);
    std::vector<uint32_t> nbr0(n * ix->M0, EMPTY), upper_off(n, EMPTY), upper;
    std::vector<char>     vecs(n * vb);
    const char *p = buf + USEARCH_HEADER_SIZE, *end = buf + len;
    for(size_t i = 0; i < n; ++i) {
        if(p + 10 > end) { set_err(ix, "lantern_gpu: truncated index file"); return false; }
        labels[ i ] = get<uint64_t>(p, 0);
        const int level = get<uint16_t>(p, 8);
        if(level > 255 || p + node_bytes(ix, level) > end) { set_err(ix, "lantern_gpu: corrupt node tape"); return false; }
        levels[ i ] = (uint8_t)level;
        if(level > 0) {
            upper_off[ i ] = (uint32_t)(upper.size() / ix->M);
            upper.resize(upper.size() + (size_t)level * ix->M, EMPTY);
        }
        const char *q = p + 10;
        for(int l = 0; l <= level; ++l) {
            const uint32_t cap = l == 0 ? ix->M0 : ix->M;
            uint32_t      *list = l == 0 ? &nbr0[ i * ix->M0 ] : &upper[ ((size_t)upper_off[ i ] + (size_t)(l - 1)) * ix->M ];
            const uint32_t cnt = get<uint32_t>(q, 0);
            if(cnt > cap) { set_err(ix, "lantern_gpu: corrupt neighbour count"); return false; }
            for(uint32_t j = 0; j < cnt; ++j) {
                list[ j ] = get<uint32_t>(q, 4 + (size_t)j * LANTERN_SLOT_SIZE);
                if(list[ j ] >= n) { set_err(ix, "lantern_gpu: neighbour slot out of range"); return false; }
            }
            q += 4 + (size_t)cap * LANTERN_SLOT_SIZE;
        }
        std::memcpy(&vecs[ i * vb ], q, vb);
        p += node_bytes(ix, level);
    }
    if(upper.empty()) upper.push_back(EMPTY);
    if(levels[ entry ] != top) { set_err(ix, "lantern_gpu: the entry node's level is not the index's top level"); return false; }
    for(size_t i = 0; i < n; ++i)
        if(levels[ i ] > top) { set_err(ix, "lantern_gpu: a node's level exceeds the index's top level"); return false; }
    // a neighbour listed on level l must itself reach level l (a walk reads its level-l list)
    for(size_t i = 0; i < n; ++i)
        for(int l = 1; l <= levels[ i ]; ++l) {
            const uint32_t *list = &upper[ ((size_t)upper_off[ i ] + (size_t)(l - 1)) * ix->M ];
            for(uint32_t j = 0; j < ix->M && list[ j ] != EMPTY; ++j)
                if(levels[ list[ j ] ] < l) { set_err(ix, "lantern_gpu: an upper-level list names a node that does not reach that level"); return false; }
        }
    // bit / f16 rows are stored as bytes in the file; the importer wants whole u32 words per row
    if(!ix->pq && vb != (size_t)ix->words * 4) {
        std::vector<char> w(n * (size_t)ix->words * 4, 0);
        for(size_t i = 0; i < n; ++i) std::memcpy(&w[ i * (size_t)ix->words * 4 ], &vecs[ i * vb ], vb);
        vecs.swap(w);
    }
    if(!import_graph_locked(ix, n, vecs.data(), labels.data(), levels.data(), nbr0.data(), upper_off.data(), upper.data(),
                            (uint32_t)get<uint64_t>(buf, OFF_G_ENTRY_SLOT), (int32_t)get<uint64_t>(buf, OFF_G_MAX_LEVEL), ix->pq))
        return false;
    return true;
}

// ---- HBM mirror of an index that lives in PostgreSQL pages ---------------------------------------------
// Lantern's scan and insert paths never load the index: they hand usearch the 136-byte header and two
// callbacks, slot -> pointer to the node tape inside a pinned shared buffer
// (lantern_hnsw/src/hnsw/external_index.c:613-697, scan.c:93-110, insert.c:130-151).  On the page every
// neighbour slot is a 6-byte ItemPointer (external_index.c:380-409) and the header's entry slot is one too
// (:411-418).  The device cannot chase host callbacks per hop, so the graph is walked ONCE from the entry
// slot through the callback, breadth first over every level's lists, into dense device ids.  Nodes that are
// unreachable from the entry point are unreachable for any search as well, so the mirror is search-equivalent.
bool mirror_from_retriever(Index *ix, const char *header)
{
    if(ix->n || !ix->pend_labels.empty()) { set_err(ix, "lantern_gpu: the mirror needs an empty index"); return false; }
    if(!ix->opts.retriever) { set_err(ix, "lantern_gpu: usearch_view_mem_lazy needs init_options.retriever"); return false; }
    if(std::memcmp(header + OFF_MAGIC, "usearch", 7) != 0) { set_err(ix, "lantern_gpu: not a usearch header"); return false; }
    const uint64_t declared = get<uint64_t>(header, OFF_G_SIZE);
    if(get<uint64_t>(header, OFF_G_CONNECTIVITY) != ix->M) { set_err(ix, "lantern_gpu: header connectivity does not match the index options"); return false; }
    if(declared == 0) { ix->page_mode = true; return true; }
    if(declared >= 0x7FFFFFFFull) { set_err(ix, "lantern_gpu: the header declares more nodes than the device index supports"); return false; }
    const uint64_t mask48 = 0xFFFFFFFFFFFFull;
    const uint64_t entry = get<uint64_t>(header, OFF_G_ENTRY_SLOT) & mask48;
    const size_t   vb = vector_bytes(ix), wbytes = ix->pq ? vb : (size_t)ix->words * 4;
    std::unordered_map<uint64_t, uint32_t> id_of;
    std::vector<uint64_t> slot_of, labels;
    std::vector<uint8_t>  levels;
    std::vector<uint32_t> nbr0, upper_off, upper;
    std::vector<char>     vecs;
    id_of.reserve((size_t)std::min<uint64_t>(declared, 1u << 20) * 2);  // the header is untrusted: it only hints at the table's size
    auto intern = [&](uint64_t slot) -> uint32_t {
        auto it = id_of.find(slot);
        if(it != id_of.end()) return it->second;
        const uint32_t id = (uint32_t)slot_of.size();
        id_of.emplace(slot, id);
        slot_of.push_back(slot);
        return id;
    };
    intern(entry);
    for(size_t head = 0; head < slot_of.size(); ++head) {
        if(slot_of.size() > declared) { set_err(ix, "lantern_gpu: the page graph has more nodes than its header declares"); return false; }
        const char *tape = (const char *)ix->opts.retriever(ix->opts.retriever_ctx, slot_of[ head ]);
        if(!tape) { set_err(ix, "lantern_gpu: retriever returned NULL"); return false; }
        labels.push_back(get<uint64_t>(tape, 0));
        const int level = get<uint16_t>(tape, 8);
        if(level > 255) { set_err(ix, "lantern_gpu: corrupt node level"); return false; }
        levels.push_back((uint8_t)level);
        nbr0.resize(nbr0.size() + ix->M0, EMPTY);
        upper_off.push_back(level > 0 ? (uint32_t)(upper.size() / ix->M) : EMPTY);
        if(level > 0) upper.resize(upper.size() + (size_t)level * ix->M, EMPTY);
        const char *q = tape + 10;
        for(int l = 0; l <= level; ++l) {
            const uint32_t cap = l == 0 ? ix->M0 : ix->M;
            const uint32_t cnt = get<uint32_t>(q, 0);
            if(cnt > cap) { set_err(ix, "lantern_gpu: corrupt neighbour count"); return false; }
            for(uint32_t j = 0; j < cnt; ++j) {
                uint64_t slot = 0;
                std::memcpy(&slot, q + 4 + (size_t)j * LANTERN_SLOT_SIZE, LANTERN_SLOT_SIZE);
                const uint32_t id = intern(slot);  // may grow the vectors below: index, do not keep pointers
                if(l == 0) nbr0[ head * ix->M0 + j ] = id;
                else upper[ ((size_t)upper_off[ head ] + (size_t)(l - 1)) * ix->M + j ] = id;
            }
            q += 4 + (size_t)cap * LANTERN_SLOT_SIZE;
        }
        vecs.resize(vecs.size() + wbytes, 0);
        std::memcpy(&vecs[ head * wbytes ], q, vb);
    }
    if(upper.empty()) upper.push_back(EMPTY);
    // the pages are as untrusted as a file: the walk kernels start at (entry, max_level) and read the level-l list of
    // every node a level-l list names
    const uint64_t top = get<uint64_t>(header, OFF_G_MAX_LEVEL);
    if(top > 255 || levels[ 0 ] != top) { set_err(ix, "lantern_gpu: the entry node's level is not the header's top level"); return false; }
    for(size_t i = 0; i < levels.size(); ++i) {
        if(levels[ i ] > top) { set_err(ix, "lantern_gpu: a node's level exceeds the header's top level"); return false; }
        for(int l = 1; l <= levels[ i ]; ++l) {
            const uint32_t *list = &upper[ ((size_t)upper_off[ i ] + (size_t)(l - 1)) * ix->M ];
            for(uint32_t j = 0; j < ix->M && list[ j ] != EMPTY; ++j)
                if(levels[ list[ j ] ] < l) { set_err(ix, "lantern_gpu: an upper-level list names a node that does not reach that level"); return false; }
        }
    }
    if(!import_graph_locked(ix, slot_of.size(), vecs.data(), labels.data(), levels.data(), nbr0.data(), upper_off.data(), upper.data(),
                            0 /* the entry slot was interned first */, (int32_t)top, ix->pq))
        return false;
    ix->page_mode = true;  // only a mirror that was built is in page mode
    ix->page_slots.swap(slot_of);
    ix->page_ids.swap(id_of);
    ix->page_declared = (size_t)declared;
    ix->page_attach_n = ix->n;
    return true;
}

// ---- usearch_add_external: the aminsert path (insert.c:200-214) -----------------------------------------------------
// Byte offset of level l's list inside a node tape (usearch_storage.cpp:19-32: [label u64][level u16] then one
// [count u32][cap x 6-byte slot] block per level, cap = 2M on level 0 and M above).
static size_t tape_list_offset(const Index *ix, int l)
{
    return 10 + (l == 0 ? 0 : (4 + (size_t)ix->M0 * LANTERN_SLOT_SIZE) + (size_t)(l - 1) * (4 + (size_t)ix->M * LANTERN_SLOT_SIZE));
}

// Write the device list of (id, level l) into a node tape, neighbour ids as the slots the pages use.
static bool write_list_to_tape(Index *ix, uint32_t id, int l, char *tape)
{
    const uint32_t cap = l == 0 ? ix->M0 : ix->M;
    std::vector<uint32_t> list(cap);
    const uint32_t *src = l == 0 ? ix->d_nbr0 + (size_t)id * ix->M0 : ix->d_upper_nbr + ((size_t)ix->upper_off[ id ] + (size_t)(l - 1)) * ix->M;
    if(hipMemcpy(list.data(), src, (size_t)cap * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_err(ix, "lantern_gpu: HIP failure reading a neighbour list"); return false; }
    char *q = tape + tape_list_offset(ix, l);
    uint32_t cnt = 0;
    while(cnt < cap && list[ cnt ] != EMPTY) ++cnt;
    std::memset(q, 0, 4 + (size_t)cap * LANTERN_SLOT_SIZE);
    put<uint32_t>(q, 0, cnt);
    for(uint32_t j = 0; j < cnt; ++j) {
        const uint64_t slot = ix->page_mode ? ix->page_slots[ list[ j ] ] : (uint64_t)list[ j ];
        std::memcpy(q + 4 + (size_t)j * LANTERN_SLOT_SIZE, &slot, LANTERN_SLOT_SIZE);
    }
    return true;
}

// After the node `id` (the newest) has been linked on the device: its own tape gets label, level, lists and the stored
// vector; every node it linked to gets its (possibly re-pruned) list of that level re-written through retriever_mut.
bool write_back_insert(Index *ix, uint32_t id, char *node_tape)
{
    const int level = ix->levels[ id ];
    put<uint64_t>(node_tape, 0, ix->labels[ id ]);
    put<uint16_t>(node_tape, 8, (uint16_t)level);
    std::vector<uint32_t> own((size_t)ix->M0);
    for(int l = 0; l <= level; ++l) {
        if(!write_list_to_tape(ix, id, l, node_tape)) return false;
        const uint32_t cap = l == 0 ? ix->M0 : ix->M;
        const uint32_t *src = l == 0 ? ix->d_nbr0 + (size_t)id * ix->M0 : ix->d_upper_nbr + ((size_t)ix->upper_off[ id ] + (size_t)(l - 1)) * ix->M;
        if(hipMemcpy(own.data(), src, (size_t)cap * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_err(ix, "lantern_gpu: HIP failure reading a neighbour list"); return false; }
        for(uint32_t j = 0; j < cap && own[ j ] != EMPTY; ++j) {
            if(!ix->page_mode) continue;  // no pages behind this index: the device lists are the only copy
            const Index::HolderBinding hb = ix->current_holder();
            if(!hb.retriever_mut) { set_err(ix, "lantern_gpu: usearch_add_external needs init_options.retriever_mut"); return false; }
            char *tape = (char *)hb.retriever_mut(hb.ctx, ix->page_slots[ own[ j ] ]);
            if(!tape) { set_err(ix, "lantern_gpu: retriever_mut returned NULL"); return false; }
            if(!write_list_to_tape(ix, own[ j ], l, tape)) return false;
        }
    }
    // the stored vector sits behind the last list (usearch_init_node leaves it zeroed: usearch_storage.cpp:34-44)
    const size_t row = ix->pq ? (size_t)ix->pq_S : (size_t)ix->chunks * 16;
    std::vector<char> stored(row);
    const char *src = ix->pq ? (const char *)ix->d_codes + (size_t)id * row : (const char *)ix->d_vec + (size_t)id * row;
    if(hipMemcpy(stored.data(), src, row, hipMemcpyDeviceToHost) != hipSuccess) { set_err(ix, "lantern_gpu: HIP failure reading a row"); return false; }
    std::memcpy(node_tape + tape_list_offset(ix, level + 1), stored.data(), vector_bytes(ix));
    return true;
}

}  // namespace lgpu

using namespace lgpu;

extern "C" {

// scan.c:110, insert.c:151.  "Lazy" in usearch (nodes are fetched per hop); here the whole reachable graph is
// mirrored into HBM once.  A scan-side shim keeps the mirror alive across scans.
void usearch_view_mem_lazy(usearch_index_t h, char *header136, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    Index *ix = (Index *)h;
    if(ix) (void)hipSetDevice(ix->device);
    if(!ix || !header136) { if(e) *e = "lantern_gpu: null index handle or header"; return; }
    std::lock_guard<std::mutex> g(ix->mu);
    try {
        if(!mirror_from_retriever(ix, header136) && e) *e = ix->err.c_str();
    } catch(const std::exception &) {  // the pages are untrusted input: nothing may unwind through the C boundary into a backend
        if(e) *e = set_err(ix, "lantern_gpu: out of host memory while mirroring the page graph");
    }
}
LANTERN_ABI_CATCH_VOID(e)

// insert.c:214: write size / max level / entry slot back into the header page copy
void usearch_update_header(usearch_index_t h, char *header136, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    Index *ix = (Index *)h;
    if(ix) (void)hipSetDevice(ix->device);
    if(!ix || !header136) { if(e) *e = "lantern_gpu: null index handle or header"; return; }
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { if(e) *e = ix->err.c_str(); return; }
    put<uint64_t>(header136, OFF_COUNT_PRESENT, logical_size(ix));
    put<uint64_t>(header136, OFF_G_SIZE, logical_size(ix));
    put<uint64_t>(header136, OFF_G_MAX_LEVEL, ix->n ? (uint64_t)ix->max_level : 0);
    // the entry point moves when an insert raises the top level; in the pages it is a 48-bit page slot
    // (external_index.c:411-418), in a file a sequential id
    if(ix->n) put<uint64_t>(header136, OFF_G_ENTRY_SLOT, ix->page_mode ? ix->page_slots[ ix->entry ] : (uint64_t)ix->entry);
}
LANTERN_ABI_CATCH_VOID(e)

// insert.c:209.  `node_tape` is the new node's tape inside a PostgreSQL page, already sized and headed by
// usearch_init_node (usearch_storage.cpp:34-44); `slot` its 48-bit page slot.  The node is linked into the HBM mirror
// exactly as usearch_add would (one sequential insertion at the caller's level), then everything the insertion changed
// is written to where PostgreSQL keeps it: the node's own lists and vector into `node_tape`, the re-written lists of
// the nodes it linked to through init_options.retriever_mut (external_index.c:673-697 marks those buffers dirty).
void usearch_add_external(usearch_index_t h, usearch_label_t label, const void *vector, void *node_tape, usearch_scalar_kind_t kind,
                          int16_t level, uint64_t slot, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    Index *ix = (Index *)h;
    if(ix) (void)hipSetDevice(ix->device);
    if(!ix || !vector || !node_tape) { if(e) *e = "lantern_gpu: null index handle, vector or node tape"; return; }
    if(level < 0 || level > 255) { if(e) *e = "lantern_gpu: level out of range"; return; }
    usearch_error_t err = nullptr;
    lantern_gpu_flush(h, &err);
    if(err) { if(e) *e = err; return; }
    uint32_t id;
    {
        std::lock_guard<std::mutex> g(ix->mu);
        // everything that can refuse the insertion is checked BEFORE the mirror changes: a mirror whose callbacks were unbound
        // (lantern_mirror_release by another holder) must fail here, not after the node has been linked on the device
        if(ix->page_mode && !ix->current_holder().retriever_mut) { if(e) *e = set_err(ix, "lantern_gpu: usearch_add_external needs init_options.retriever_mut"); return; }
        id = (uint32_t)ix->n;
        if(ix->page_mode) {
            const uint64_t s48 = slot & 0xFFFFFFFFFFFFull;
            if(ix->page_ids.count(s48)) { if(e) *e = set_err(ix, "lantern_gpu: usearch_add_external: the slot is already in the index"); return; }
            if(ix->page_slots.size() != ix->n) { if(e) *e = set_err(ix, "lantern_gpu: the mirror and its page slots are out of step"); return; }
            ix->page_slots.push_back(s48);
            ix->page_ids.emplace(s48, id);
        }
    }
    lantern_gpu_add_with_level(h, label, vector, kind, (int)level, &err);
    if(!err) lantern_gpu_flush(h, &err);
    std::lock_guard<std::mutex> g(ix->mu);
    if(err || ix->n != (size_t)id + 1) {
        if(ix->page_mode && ix->page_slots.size() > ix->n) {  // the node did not make it in
            ix->page_ids.erase(ix->page_slots.back());
            ix->page_slots.pop_back();
        }
        if(e) *e = err ? err : set_err(ix, "lantern_gpu: usearch_add_external: the insertion did not complete");
        return;
    }
    if(!write_back_insert(ix, id, (char *)node_tape) && e) *e = ix->err.c_str();
}
LANTERN_ABI_CATCH_VOID(e)


uint64_t usearch_header_get_entry_slot(char *h) { return get<uint64_t>(h, OFF_G_ENTRY_SLOT); }
void     usearch_header_set_entry_slot(char *h, uint64_t slot) { put<uint64_t>(h, OFF_G_ENTRY_SLOT, slot); }

size_t usearch_serialized_length(usearch_index_t h, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    Index *ix = (Index *)h;
    if(ix) (void)hipSetDevice(ix->device);
    if(!ix) { if(e) *e = "lantern_gpu: null index handle"; return 0; }
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { if(e) *e = ix->err.c_str(); return 0; }
    return serialized_length(ix);
}
LANTERN_ABI_CATCH(e)

void usearch_save_buffer(usearch_index_t h, char *buffer, size_t length, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    Index *ix = (Index *)h;
    if(ix) (void)hipSetDevice(ix->device);
    if(!ix) { if(e) *e = "lantern_gpu: null index handle"; return; }
    std::lock_guard<std::mutex> g(ix->mu);
    try {
        if(!flush_locked(ix) || !serialize(ix, buffer, length)) { if(e) *e = ix->err.c_str(); }
    } catch(const std::exception &ex) {  // (an allocation failure must not leave through the C boundary)
        if(e) *e = set_err(ix, std::string("lantern_gpu: ") + ex.what());
    }
}
LANTERN_ABI_CATCH_VOID(e)

void usearch_save(usearch_index_t h, const char *path, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    Index *ix = (Index *)h;
    if(ix) (void)hipSetDevice(ix->device);
    if(!ix) { if(e) *e = "lantern_gpu: null index handle"; return; }
    std::lock_guard<std::mutex> g(ix->mu);
    if(!flush_locked(ix)) { if(e) *e = ix->err.c_str(); return; }
    FILE *f = std::fopen(path, "wb");
    if(!f) { if(e) *e = set_err(ix, std::string("lantern_gpu: cannot write index file ") + path); return; }
    bool wrote = true, ok = false;
    try {
        ok = serialize_stream(ix, [&](const lantern_gpu_span *sp, size_t cnt) {
            for(size_t i = 0; i < cnt && wrote; ++i) wrote = std::fwrite(sp[ i ].data, 1, sp[ i ].size, f) == sp[ i ].size;
            return wrote;
        });
    } catch(const std::exception &ex) {
        set_err(ix, std::string("lantern_gpu: ") + ex.what());
    }
    if(std::fclose(f) != 0) wrote = false;
    if(!wrote) { if(e) *e = set_err(ix, std::string("lantern_gpu: cannot write index file ") + path); return; }
    if(!ok && e) *e = ix->err.c_str();
}
LANTERN_ABI_CATCH_VOID(e)

void lantern_gpu_save_stream(usearch_index_t h, lantern_gpu_write_fn fn, void *ctx, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    Index *ix = (Index *)h;
    if(ix) (void)hipSetDevice(ix->device);
    if(!ix) { if(e) *e = "lantern_gpu: null index handle"; return; }
    if(!fn) { if(e) *e = "lantern_gpu: null write callback"; return; }
    std::lock_guard<std::mutex> g(ix->mu);
    try {
        if(!flush_locked(ix) || !serialize_stream(ix, [&](const lantern_gpu_span *sp, size_t cnt) { return fn(ctx, sp, cnt) == 0; })) {
            if(e) *e = ix->err.c_str();
        }
    } catch(const std::exception &ex) {
        if(e) *e = set_err(ix, std::string("lantern_gpu: ") + ex.what());
    }
}
LANTERN_ABI_CATCH_VOID(e)

void usearch_load_buffer(usearch_index_t h, const char *buffer, size_t length, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    Index *ix = (Index *)h;
    if(ix) (void)hipSetDevice(ix->device);
    if(!ix) { if(e) *e = "lantern_gpu: null index handle"; return; }
    std::lock_guard<std::mutex> g(ix->mu);
    if(!deserialize(ix, buffer, length) && e) *e = ix->err.c_str();
}
LANTERN_ABI_CATCH_VOID(e)

void usearch_load(usearch_index_t h, const char *path, usearch_error_t *e)
try {
    if(e) *e = nullptr;
    Index *ix = (Index *)h;
    if(ix) (void)hipSetDevice(ix->device);
    if(!ix) { if(e) *e = "lantern_gpu: null index handle"; return; }
    FILE *f = std::fopen(path, "rb");
    if(!f) { if(e) *e = set_err(ix, std::string("lantern_gpu: cannot open index file ") + path); return; }
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)(sz > 0 ? sz : 0));
    const bool rd = std::fread(buf.data(), 1, buf.size(), f) == buf.size();
    std::fclose(f);
    if(!rd) { if(e) *e = set_err(ix, std::string("lantern_gpu: short read on ") + path); return; }
    usearch_load_buffer(h, buf.data(), buf.size(), e);
}
LANTERN_ABI_CATCH_VOID(e)

}  // extern "C"
