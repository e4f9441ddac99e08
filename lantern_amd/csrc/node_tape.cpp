// Lantern's node-tape helpers, as plain byte arithmetic.
//
// In the reference these seven functions are Lantern's own (lantern_hnsw/src/hnsw/usearch_storage.hpp:9-23) but they are C++
// compiled against usearch's templates (node_at<>, precomputed_constants_t: usearch_storage.cpp:2-16), so a library that takes
// usearch's place has to bring them: external_index.c:96-97,394-398,488, insert.c:207, delete.c:54-58 and utils.c:93 call them.
// Layout (validate_index.c:105-226, usearch_storage.cpp:19-32):
//     [key u64][level u16] { [count u32][slot 6 B x cap] } x (level + 1) [vector bytes],   cap = 2M at level 0, M above
// metadata_t carries the two list sizes in bytes (neighbors_base_bytes = 4 + 2M*6, neighbors_bytes = 4 + M*6).
// Host code only: nothing here touches a device.
#include <cstring>

#include "../../include/lantern_gpu.h"

namespace
{
constexpr size_t NODE_HEAD_BYTES = sizeof(usearch_label_t) + sizeof(uint16_t);

// bits of one stored scalar (usearch_storage.cpp:44-81: bits_per_scalar of the quantization kind; b1 = one bit per dimension)
unsigned scalar_bits(usearch_scalar_kind_t k)
{
    switch(k) {
        case usearch_scalar_f64_k: return 64;
        case usearch_scalar_f16_k: return 16;
        case usearch_scalar_i8_k: return 8;
        case usearch_scalar_b1_k: return 1;
        case usearch_scalar_f32_k: return 32;
        default: return 0;
    }
}

size_t lists_bytes(const metadata_t *m, unsigned level) { return m->neighbors_base_bytes + m->neighbors_bytes * (size_t)level; }
}  // namespace

extern "C" {

// usearch_storage.cpp:19-32
uint32_t UsearchNodeBytes(const metadata_t *metadata, int vector_bytes, int level)
{
    const size_t vec = metadata->init_options.pq ? metadata->init_options.num_subvectors : (size_t)vector_bytes;
    return (uint32_t)(NODE_HEAD_BYTES + lists_bytes(metadata, (unsigned)level) + vec);
}

// usearch_storage.cpp:34-44: a zeroed tape with key and level set; the lists and the vector are usearch_add_external's to fill
void usearch_init_node(metadata_t *meta, char *tape, usearch_key_t key, uint32_t level, uint64_t /*slot_id*/, void * /*vector*/, size_t vector_len)
{
    std::memset(tape, 0, UsearchNodeBytes(meta, (int)vector_len, (int)level));
    const uint16_t l16 = (uint16_t)level;
    std::memcpy(tape, &key, sizeof key);
    std::memcpy(tape + sizeof key, &l16, sizeof l16);
}

unsigned long level_from_node(char *node)
{
    uint16_t l;
    std::memcpy(&l, node + sizeof(usearch_label_t), sizeof l);
    return l;
}

usearch_label_t label_from_node(char *node)
{
    usearch_label_t k;
    std::memcpy(&k, node, sizeof k);
    return k;
}

// delete.c:58: label 0 = INVALID_ELEMENT_LABEL (hnsw.h:40), skipped by the scan (scan.c:296-300)
void reset_node_label(char *node) { std::memset(node, 0, sizeof(usearch_label_t)); }

// usearch_storage.cpp:63-81: the node's own level decides its size; vector bytes = dimensions * bits / 8, or the code bytes of a pq index
uint32_t node_tuple_size(char *node, uint32_t vector_dim, const metadata_t *meta)
{
    size_t vec = (size_t)vector_dim * scalar_bits(meta->init_options.quantization) / 8;
    if(meta->init_options.pq) vec = meta->init_options.num_subvectors;
    return (uint32_t)(NODE_HEAD_BYTES + lists_bytes(meta, (unsigned)level_from_node(node)) + vec);
}

// usearch_storage.cpp:101-118: the slots of one level's list (6 bytes each, unaligned) and how many are in use
void *get_node_neighbors_mut(const metadata_t *meta, char *node, uint32_t level, uint32_t *neighbors_count)
{
    char *list = node + NODE_HEAD_BYTES + (level == 0 ? 0 : meta->neighbors_base_bytes + meta->neighbors_bytes * (size_t)(level - 1));
    std::memcpy(neighbors_count, list, sizeof(uint32_t));
    return list + sizeof(uint32_t);
}

// The reloption `quant_bits` -> the scalar kind usearch_init takes (lantern_hnsw/src/hnsw/options.c:137-158), with the reference's
// error texts: a value that is not one of 1, 2, 4, 8, 16, 32 is rejected by the enum reloption (options.c:37-42,301-309; the text is
// pinned by test/expected/hnsw_sq.out:30-35), 4 and 2 are "unimplemented quantization" (options.c:150-153).  0 bits = unset = f32
// only through `unset` (the SQL surface spells an explicit 0 as an error: hnsw_sq.out:33-35).
usearch_scalar_kind_t lantern_quant_bits_scalar_kind(int quant_bits, bool unset, usearch_error_t *e)
{
    if(e) *e = nullptr;
    if(unset) return usearch_scalar_f32_k;
    switch(quant_bits) {
        case 32: return usearch_scalar_f32_k;
        case 16: return usearch_scalar_f16_k;
        case 8: return usearch_scalar_i8_k;
        case 1: return usearch_scalar_b1_k;
        case 4:
        case 2:
            if(e) *e = "unimplemented quantization";
            return usearch_scalar_unknown_k;
        default:
            if(e) *e = "Unsupported quantization bits. Supported values are 1, 2, 4, 8, 16 and 32";
            return usearch_scalar_unknown_k;
    }
}

}  // extern "C"
