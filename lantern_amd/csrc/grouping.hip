// grouping.hip -- the reverse-link requests of one insertion batch, grouped on the device.
//
// k_connect leaves one request (close, level, new_slot, d) per selected neighbour, new-slot-major, with EMPTY entries
// where a node kept fewer than M (kernels.hpp LinkReq).  The reverse-link kernels want them grouped by (close, level)
// with every group in new-slot order -- the order the sequential algorithm (usearch reconnect_neighbor_nodes_, reached
// from usearch_add: lantern_hnsw/src/hnsw/build.c:128) applies them in.  Round 1 did this on the host: D2H of the
// requests, a stable LSD radix sort, H2D, one stream synchronisation per batch (284 per 1M build).  Here:
//
//   k_link_keys     key = close << 8 | level (40 bits; EMPTY or not-owned -> all ones), value = position
//   rocPRIM         stable radix sort of (key, position) over the 40 key bits   [library sort: not a hot op]
//   k_gather_heads  sorted[j] = links[position[j]]; a thread whose key differs from its left neighbour's opens a
//                   group: it scans to the group's end and appends {begin, end} to the group list (atomic counter;
//                   groups are independent of one another, so their order is immaterial)
//   k_batch_layout  link_off / item_node of a batch from the levels already in HBM
//
// so a build's batches queue up on the stream back to back.
#include <cstring>  // rocPRIM's headers use memset without including it
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "kernels.hpp"

namespace lgpu {

namespace {
constexpr uint64_t KEY_NONE = (1ull << 40) - 1;  // sorts behind every real key (close < 2^31)

__global__ void __launch_bounds__(256) k_link_keys(const LinkReq *links, uint32_t n, uint64_t *keys, uint32_t *idx, uint32_t *ngroups, int world,
                                                   int rank, uint32_t *owner_counts)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i == 0) *ngroups = 0;
    if(i >= n) return;
    const LinkReq r = links[ i ];
    uint64_t      k = KEY_NONE;
    if(r.close != EMPTY) {
        bool mine = true;
        if(world > 1) {  // a rank of a work-sharded build applies the groups of the nodes it owns and counts the others'
            const uint32_t o = r.close % (uint32_t)world;
            atomicAdd(&owner_counts[ o ], 1u);
            mine = (int)o == rank;
        }
        if(mine) k = ((uint64_t)r.close << 8) | (uint64_t)(r.level & 0xFFu);
    }
    keys[ i ] = k;
    idx[ i ] = i;
}

// A rank of a work-sharded build (world > 1) sorts ONLY the requests it owns (close % world == rank): the append position is the
// value the owner count's atomicAdd returns, and the key carries the request's position in its low 24 bits -- keys are unique,
// so the order inside a group is the new-slot order whatever order the appends landed in.  1 / world of the sort and of the
// gather per rank instead of a replicated pass over every rank's requests.
constexpr uint32_t POS_BITS = 24;
__global__ void __launch_bounds__(256) k_link_keys_owned(const LinkReq *links, uint32_t n, uint64_t *keys, uint32_t *ngroups, int world, int rank,
                                                         uint32_t *owner_counts)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i == 0) *ngroups = 0;
    if(i >= n) return;
    const LinkReq r = links[ i ];
    if(r.close == EMPTY) return;
    const uint32_t o = r.close % (uint32_t)world;
    const uint32_t at = atomicAdd(&owner_counts[ o ], 1u);
    if((int)o == rank) keys[ at ] = (((uint64_t)r.close << 8 | (uint64_t)(r.level & 0xFFu)) << POS_BITS) | (uint64_t)i;
}
__global__ void __launch_bounds__(256) k_gather_heads_owned(const LinkReq *links, uint32_t m, const uint64_t *keys, LinkReq *sorted, uint2 *groups,
                                                            uint32_t *ngroups)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= m) return;
    const uint64_t k = keys[ j ], k40 = k >> POS_BITS;
    sorted[ j ] = links[ (uint32_t)(k & ((1u << POS_BITS) - 1u)) ];
    if(j != 0 && (keys[ j - 1 ] >> POS_BITS) == k40) return;
    uint32_t e = j + 1;
    while(e < m && (keys[ e ] >> POS_BITS) == k40) ++e;
    groups[ atomicAdd(ngroups, 1u) ] = make_uint2(j, e);
}

__global__ void __launch_bounds__(256) k_gather_heads(const LinkReq *links, uint32_t n, const uint64_t *keys, const uint32_t *idx, LinkReq *sorted,
                                                      uint2 *groups, uint32_t *ngroups)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= n) return;
    const uint64_t k = keys[ j ];
    if(k == KEY_NONE) return;
    sorted[ j ] = links[ idx[ j ] ];
    if(j != 0 && keys[ j - 1 ] == k) return;
    uint32_t e = j + 1;
    while(e < n && keys[ e ] == k) ++e;
    const uint32_t g = atomicAdd(ngroups, 1u);
    groups[ g ] = make_uint2(j, e);
}

// A batch of a handful of insertions (ldb_aminsert's one row: <= 32 requests) does not need three kernels and a library sort:
// ONE workgroup sorts (key << 11 | position) -- unique, so any sort is the stable sort -- bitonically in LDS and writes the
// sorted requests and the groups.  Same outputs as the pipeline above (the order of the GROUP LIST is immaterial there too).
constexpr uint32_t SMALL_N = 2048;
template <uint32_t N, uint32_t THREADS>  // N: keys sorted (a power of two >= n); a lone insertion's <= 64 requests take one wave
__global__ void __launch_bounds__(THREADS) k_group_small(const LinkReq *links, uint32_t n, LinkReq *sorted, uint2 *groups, uint32_t *ngroups, uint32_t *zero_me)
{
    __shared__ uint64_t key[ N ];
    const uint32_t tid = threadIdx.x;
    if(tid == 0) {
        *ngroups = 0;
        if(zero_me) *zero_me = 0;  // (the reverse-link kernels' work counter: saves its own memset node)
    }
    for(uint32_t i = tid; i < N; i += THREADS) {
        uint64_t k = ~0ull;
        if(i < n) {
            const LinkReq r = links[ i ];
            const uint64_t k40 = r.close != EMPTY ? (((uint64_t)r.close << 8) | (uint64_t)(r.level & 0xFFu)) : KEY_NONE;
            k = (k40 << 11) | (uint64_t)i;
        }
        key[ i ] = k;
    }
    __syncthreads();
    for(uint32_t size = 2; size <= N; size <<= 1) {
        for(uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for(uint32_t t = tid; t < N / 2; t += THREADS) {
                const uint32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool     up = (lo & size) == 0;
                const uint64_t a = key[ lo ], b = key[ hi ];
                if((a > b) == up) { key[ lo ] = b; key[ hi ] = a; }
            }
            __syncthreads();
        }
    }
    for(uint32_t j = tid; j < n; j += THREADS) {
        const uint64_t k = key[ j ], k40 = k >> 11;
        if(k40 == KEY_NONE) continue;
        sorted[ j ] = links[ (uint32_t)(k & 2047u) ];
        if(j != 0 && (key[ j - 1 ] >> 11) == k40) continue;
        uint32_t e = j + 1;
        while(e < n && (key[ e ] >> 11) == k40) ++e;
        groups[ atomicAdd(ngroups, 1u) ] = make_uint2(j, e);
    }
}

// one block: exclusive scan of (level + 1) over the batch
__global__ void __launch_bounds__(1024) k_batch_layout(const uint8_t *levels, uint32_t b, uint32_t M, uint32_t *link_off, uint32_t *item_node)
{
    __shared__ uint32_t part[ 1024 ];
    const uint32_t tid = threadIdx.x, T = blockDim.x, per = (b + T - 1) / T;  // (T = 64 for a handful of nodes: one wave, six scan steps)
    const uint32_t lo = tid * per, hi = lo + per < b ? lo + per : b;
    uint32_t       sum = 0;
    for(uint32_t i = lo; i < hi; ++i) sum += (uint32_t)levels[ i ] + 1u;
    part[ tid ] = sum;
    __syncthreads();
    for(uint32_t off = 1; off < T; off <<= 1) {  // Hillis-Steele inclusive scan
        const uint32_t v = tid >= off ? part[ tid - off ] : 0u;
        __syncthreads();
        part[ tid ] += v;
        __syncthreads();
    }
    uint32_t item = part[ tid ] - sum;  // exclusive prefix: first item of this thread's first node
    for(uint32_t i = lo; i < hi; ++i) {
        link_off[ i ] = item * M;
        const uint32_t cnt = (uint32_t)levels[ i ] + 1u;
        for(uint32_t l = 0; l < cnt; ++l) item_node[ item + l ] = i;
        item += cnt;
    }
}
// ---- product quantisation (pq = true): per-subvector staging either side of the nearest-centroid search -----------
// take: sub[i][0 .. subdim) = row(first + i)[s * subdim ..], zero padded to whole chunks
__global__ void __launch_bounds__(256) k_pq_take(const float *rows, uint32_t row_floats, uint32_t first, uint32_t count, uint32_t s, uint32_t subdim,
                                                 uint32_t sub_floats, float *sub)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = (uint32_t)(t / sub_floats), j = (uint32_t)(t % sub_floats);
    if(i >= count) return;
    sub[ (size_t)i * sub_floats + j ] = j < subdim ? rows[ (size_t)(first + i) * row_floats + (size_t)s * subdim + j ] : 0.f;
}
// put: codes[first + i][s] = nearest[i]; row(first + i)[s * subdim ..] = the centroid's values (the row becomes its decoding)
__global__ void __launch_bounds__(256) k_pq_put(float *rows, uint32_t row_floats, uint32_t first, uint32_t count, uint32_t s, uint32_t subdim, uint32_t S,
                                                const uint32_t *nearest, const float *codebook, uint32_t dims, uint8_t *codes)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = (uint32_t)(t / subdim), j = (uint32_t)(t % subdim);
    if(i >= count) return;
    const uint32_t c = nearest[ i ];
    rows[ (size_t)(first + i) * row_floats + (size_t)s * subdim + j ] = codebook[ (size_t)c * dims + (size_t)s * subdim + j ];
    if(j == 0) codes[ (size_t)(first + i) * S + s ] = (uint8_t)c;
}
// decode: row(first + i) = concatenation of the centroids its codes name
__global__ void __launch_bounds__(256) k_pq_decode(float *rows, uint32_t row_floats, uint32_t first, uint32_t count, uint32_t subdim, uint32_t S,
                                                   const float *codebook, uint32_t dims, const uint8_t *codes)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = (uint32_t)(t / row_floats), j = (uint32_t)(t % row_floats);
    if(i >= count) return;
    float v = 0.f;
    if(j < dims) v = codebook[ (size_t)codes[ (size_t)(first + i) * S + j / subdim ] * dims + j ];
    rows[ (size_t)(first + i) * row_floats + j ] = v;
}
// ---- quantised storage of f32 input, on the device: a bulk add uploads the caller's f32 rows once and this kernel writes
// the stored rows -- the rules of pad_row (index.cpp), element for element: f16 = round to nearest even; i8 =
// trunc(clamp(x * 100, -100, 100)), NaN -> 0; b1 = bit (x > 0), most significant bit of each byte first.  Rows are zero
// padded to whole 16-byte chunks.  kind: 3 = f16, 4 = i8, 5 = b1 (usearch_scalar_kind_t).
__global__ void __launch_bounds__(256) k_store_quantised(const float *src, uint32_t dims, uint32_t count, int kind, uint32_t *rows, uint32_t row_words)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = (uint32_t)(t / row_words), w = (uint32_t)(t % row_words);
    if(i >= count) return;
    const float *f = src + (size_t)i * dims;
    uint32_t     out = 0;
    if(kind == 3) {  // two halves per word, low half first
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 v = { (_Float16)0.f, (_Float16)0.f };
        if(2 * w < dims) v[ 0 ] = (_Float16)f[ 2 * w ];
        if(2 * w + 1 < dims) v[ 1 ] = (_Float16)f[ 2 * w + 1 ];
        out = __builtin_bit_cast(uint32_t, v);
    } else if(kind == 4) {  // four signed bytes per word
        for(uint32_t b = 0; b < 4; ++b) {
            const uint32_t e = 4 * w + b;
            int            q = 0;
            if(e < dims) {
                float v = f[ e ] * 100.0f;
                if(!(v == v)) v = 0.f;
                v = v > 100.0f ? 100.0f : v;
                v = v < -100.0f ? -100.0f : v;
                q = (int)v;
            }
            out |= ((uint32_t)q & 0xFFu) << (8 * b);
        }
    } else {  // 32 bits per word; byte b of the word holds elements 32 w + 8 b .. + 7, most significant bit first
        for(uint32_t b = 0; b < 32; ++b) {
            const uint32_t e = 32 * w + b;
            if(e < dims && f[ e ] > 0.f) out |= (128u >> (b & 7)) << (8 * (b >> 3));
        }
    }
    rows[ (size_t)i * row_words + w ] = out;
}
}  // namespace

hipError_t launch_store_quantised(const float *src, uint32_t dims, uint32_t count, int kind, uint32_t *rows, uint32_t row_words, hipStream_t stream)
{
    if(count == 0) return hipSuccess;
    const uint64_t threads = (uint64_t)count * row_words;
    hipLaunchKernelGGL(k_store_quantised, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream, src, dims, count, kind, rows, row_words);
    return hipGetLastError();
}

hipError_t launch_pq_take(const float *rows, uint32_t row_floats, uint32_t first, uint32_t count, uint32_t s, uint32_t subdim, uint32_t sub_floats,
                          float *sub, hipStream_t stream)
{
    if(count == 0) return hipSuccess;
    const uint64_t threads = (uint64_t)count * sub_floats;
    hipLaunchKernelGGL(k_pq_take, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream, rows, row_floats, first, count, s, subdim, sub_floats, sub);
    return hipGetLastError();
}
hipError_t launch_pq_put(float *rows, uint32_t row_floats, uint32_t first, uint32_t count, uint32_t s, uint32_t subdim, uint32_t S, const uint32_t *nearest,
                         const float *codebook, uint32_t dims, uint8_t *codes, hipStream_t stream)
{
    if(count == 0) return hipSuccess;
    const uint64_t threads = (uint64_t)count * subdim;
    hipLaunchKernelGGL(k_pq_put, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream, rows, row_floats, first, count, s, subdim, S, nearest, codebook,
                       dims, codes);
    return hipGetLastError();
}
hipError_t launch_pq_decode(float *rows, uint32_t row_floats, uint32_t first, uint32_t count, uint32_t subdim, uint32_t S, const float *codebook, uint32_t dims,
                            const uint8_t *codes, hipStream_t stream)
{
    if(count == 0) return hipSuccess;
    const uint64_t threads = (uint64_t)count * row_floats;
    hipLaunchKernelGGL(k_pq_decode, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream, rows, row_floats, first, count, subdim, S, codebook, dims,
                       codes);
    return hipGetLastError();
}

size_t group_temp_bytes(size_t n)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                    n ? n : 1, 0u, 40u, (hipStream_t) nullptr);
    size_t owned = 0;  // (the keys-only sort of a sharded batch's owned requests: launch_group_sort_owned)
    (void)rocprim::radix_sort_keys(nullptr, owned, (const uint64_t *)nullptr, (uint64_t *)nullptr, n ? n : 1, 0u, 63u, (hipStream_t) nullptr);
    return bytes > owned ? bytes : owned;
}

hipError_t launch_group_requests(const LinkReq *links, uint32_t n, const GroupScratch &gs, LinkReq *sorted, uint2 *groups, uint32_t *ngroups,
                                 int world, int rank, uint32_t *owner_counts, hipStream_t stream, uint32_t *zero_me)
{
    if(n == 0) {
        hipError_t e0 = hipMemsetAsync(ngroups, 0, 4, stream);
        return e0 != hipSuccess || !zero_me ? e0 : hipMemsetAsync(zero_me, 0, 4, stream);
    }
    if(world <= 1 && n <= SMALL_N) {
        if(n <= 64) hipLaunchKernelGGL((k_group_small<64, 64>), dim3(1), dim3(64), 0, stream, links, n, sorted, groups, ngroups, zero_me);
        else if(n <= 512) hipLaunchKernelGGL((k_group_small<512, 256>), dim3(1), dim3(256), 0, stream, links, n, sorted, groups, ngroups, zero_me);
        else hipLaunchKernelGGL((k_group_small<SMALL_N, 1024>), dim3(1), dim3(1024), 0, stream, links, n, sorted, groups, ngroups, zero_me);
        return hipGetLastError();
    }
    hipError_t e = hipSuccess;
    if(zero_me && (e = hipMemsetAsync(zero_me, 0, 4, stream)) != hipSuccess) return e;
    if(world > 1 && (e = hipMemsetAsync(owner_counts, 0, (size_t)world * 4, stream)) != hipSuccess) return e;
    const uint32_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_link_keys, dim3(blocks), dim3(256), 0, stream, links, n, gs.keys_a, gs.idx_a, ngroups, world, rank, owner_counts);
    size_t temp = gs.temp_bytes;
    e = rocprim::radix_sort_pairs(gs.temp, temp, (const uint64_t *)gs.keys_a, gs.keys_b, (const uint32_t *)gs.idx_a, gs.idx_b, (size_t)n, 0u, 40u, stream);
    if(e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gather_heads, dim3(blocks), dim3(256), 0, stream, links, n, (const uint64_t *)gs.keys_b, (const uint32_t *)gs.idx_b, sorted, groups,
                       ngroups);
    return hipGetLastError();
}

// The same for a rank of a work-sharded build, in two steps either side of the one host wait a sharded batch has anyway (the owner
// counts size the second exchange): keys of the OWNED requests, appended; then -- `m` = owner_counts[rank], now known to the host --
// the sort of those m keys (unique: 39 key bits + 24 position bits) and the gather.  n < 2^24 (the caller checks).
hipError_t launch_group_keys_owned(const LinkReq *links, uint32_t n, const GroupScratch &gs, uint32_t *ngroups, int world, int rank, uint32_t *owner_counts,
                                   hipStream_t stream, uint32_t *zero_me)
{
    hipError_t e = hipSuccess;
    if(zero_me && (e = hipMemsetAsync(zero_me, 0, 4, stream)) != hipSuccess) return e;
    if((e = hipMemsetAsync(owner_counts, 0, (size_t)world * 4, stream)) != hipSuccess) return e;
    if(n == 0) return hipMemsetAsync(ngroups, 0, 4, stream);
    hipLaunchKernelGGL(k_link_keys_owned, dim3((n + 255) / 256), dim3(256), 0, stream, links, n, gs.keys_a, ngroups, world, rank, owner_counts);
    return hipGetLastError();
}
hipError_t launch_group_sort_owned(const LinkReq *links, uint32_t m, const GroupScratch &gs, LinkReq *sorted, uint2 *groups, uint32_t *ngroups,
                                   hipStream_t stream)
{
    if(m == 0) return hipSuccess;
    size_t     temp = gs.temp_bytes;
    hipError_t e = rocprim::radix_sort_keys(gs.temp, temp, (const uint64_t *)gs.keys_a, gs.keys_b, (size_t)m, 0u, 39u + POS_BITS, stream);
    if(e != hipSuccess) return e;
    hipLaunchKernelGGL(k_gather_heads_owned, dim3((m + 255) / 256), dim3(256), 0, stream, links, m, (const uint64_t *)gs.keys_b, sorted, groups, ngroups);
    return hipGetLastError();
}

// A handful of new rows (ldb_aminsert's one): rows | labels | upper offsets | levels go from a page-locked, device-mapped block to
// their places in ONE kernel that reads the host block over the bus -- instead of four queued copies (index.cpp insert_rows).
__global__ void __launch_bounds__(256) k_stage_small(const uint4 *rows, const uint64_t *labels, const uint32_t *upper_off, const uint8_t *levels,
                                                     uint32_t chunks, uint4 *d_rows, uint64_t *d_labels, uint32_t *d_upper_off, uint8_t *d_levels)
{
    const uint32_t i = blockIdx.x, tid = threadIdx.x;
    for(uint32_t ch = tid; ch < chunks; ch += 256) d_rows[ (size_t)i * chunks + ch ] = rows[ (size_t)i * chunks + ch ];
    if(tid == 0) {
        d_labels[ i ] = labels[ i ];
        d_upper_off[ i ] = upper_off[ i ];
        d_levels[ i ] = levels[ i ];
    }
}

hipError_t launch_stage_small(const void *rows, const uint64_t *labels, const uint32_t *upper_off, const uint8_t *levels, uint32_t count, uint32_t chunks,
                              void *d_rows, uint64_t *d_labels, uint32_t *d_upper_off, uint8_t *d_levels, hipStream_t stream)
{
    if(count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_stage_small, dim3(count), dim3(256), 0, stream, (const uint4 *)rows, labels, upper_off, levels, chunks, (uint4 *)d_rows, d_labels,
                       d_upper_off, d_levels);
    return hipGetLastError();
}

hipError_t launch_batch_layout(const uint8_t *levels, uint32_t b, uint32_t M, uint32_t *link_off, uint32_t *item_node, hipStream_t stream)
{
    if(b == 0) return hipSuccess;
    hipLaunchKernelGGL(k_batch_layout, dim3(1), dim3(b <= 64 ? 64 : 1024), 0, stream, levels, b, M, link_off, item_node);
    return hipGetLastError();
}

}  // namespace lgpu
