// device_common.hpp -- gfx950 device primitives shared by every kernel of the HNSW path.
//
// Numerics contract: a row of d f32 scalars is zero-padded to whole
// 16-byte chunks; G lanes (8/16/32/64, chosen from the chunk count: group_lanes_for) cooperate on one row; lane l
// owns chunks l, l+G, l+2G... and runs ONE fmaf chain per accumulator over its scalars in memory
// order; the G partials are combined by a butterfly with offsets 1, 2, 4 .. G/2 (each step adds the
// partner's running sum), executed as DPP adds (quad_perm, row_half_mirror, row_mirror, row_bcast15,
// row_bcast31) -- one VALU op per step instead of a ds_bpermute round trip.  The complete sum is
// guaranteed in the LAST lane of the group (lane G-1).  The oracle models exactly this tree
// (oracle/metrics.c, LO_SUM_WAVE64), so device results are compared bit-for-bit.  Built with
// -ffp-contract=off: every fma below is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lgpu {

constexpr uint32_t EMPTY = 0xFFFFFFFFu;
// steps of a row pair's chunk loop whose loads are issued together (group_dist2_n); 0 = one step at a time (the form of rounds 1 - 5)
#ifndef LGPU_ROW_BLOCK
#define LGPU_ROW_BLOCK 3
#endif
#ifndef LGPU_ROW_BLOCK_COS
#define LGPU_ROW_BLOCK_COS 2
#endif
#ifndef LGPU_LIST_PREFETCH  // speculative fetch of the front's neighbour list (walk.hpp search_level_reg): 0 off, 1 rows of < 128 chunks, 2 all
#define LGPU_LIST_PREFETCH 1
#endif
#ifndef LGPU_ROW_BLOCK1
#define LGPU_ROW_BLOCK1 4
#endif

// values follow usearch_metric_kind_t (include/lantern_gpu.h)
constexpr int M_COS = 1;
constexpr int M_L2SQ = 3;
constexpr int M_HAMMING = 8;
// COSINE over b1 STORAGE (quant_bits = 1 on a dist_cos_ops index: options.c:154-155 returns b1 for any metric, and the
// reference's integration test builds such indexes, scripts/integration_tests.py:664-667).  The metrics over b1 storage are the
// f32 metrics over the {0, 1} values the bits stand for: sum (a - b)^2 is the Hamming distance (M_HAMMING serves l2sq), and the
// cosine is 1 - |a & b| / (sqrt |a| sqrt |b|) with the f32 metric's zero-norm rules -- three popcounts, integer-exact.  What the
// fork computes there cannot be read off the tree (upstream usearch has no cos metric for b1x8): PARITY UNPINNED.
constexpr int M_COS_B1 = 9;
// the same metrics over f16 STORAGE (quant_bits = 16, lantern_hnsw/src/hnsw/options.c:137-158): rows hold 8
// halves per 16-byte chunk; every element is converted to f32 (exactly) and the arithmetic is the f32
// arithmetic above -- usearch's metric_*_gt<f16_t, f32>.  Internal codes = metric + 100.
constexpr int M_F16 = 100;
constexpr int M_COS_F16 = M_COS + M_F16;
constexpr int M_L2SQ_F16 = M_L2SQ + M_F16;
// i8 STORAGE (quant_bits = 8): 16 scalars per 16-byte chunk.  usearch quantises f32 -> i8 as trunc(x * 100) clamped
// to [-100, 100] (lantern_hnsw/test/sql/hnsw_sq.sql:33-34: "i8 uniform [-1-1]=>[-100,100] quantization") and its
// cos_i8_t / l2sq_i8_t metrics accumulate in int32 -- integer-exact, so the summation order is immaterial.
// Internal codes = metric + 200.
constexpr int M_I8 = 200;
constexpr int M_COS_I8 = M_COS + M_I8;
constexpr int M_L2SQ_I8 = M_L2SQ + M_I8;
// PQ CODE storage searched by asymmetric distance computation (pq = true, compact: usearch_storage.cpp:29-31 -- a node carries
// num_subvectors code bytes): a row is its code bytes, zero padded to 16-byte chunks; the "query" is a per-query table in LDS,
// lut[s][c] = the metric's partial sum between subvector s of the query and centroid c (search_adc_kernel.hip).  Internal
// codes = metric + 400; only the search kernel is instantiated for them.
constexpr int M_ADC = 400;
constexpr int M_COS_ADC = M_COS + M_ADC;
constexpr int M_L2SQ_ADC = M_L2SQ + M_ADC;
// entries of the undo log behind every workgroup's HBM visited bitmap (walk.hpp VisUndo): a walk that records more ids than this in the
// bitmap clears the whole bitmap at its end instead.  A walk at ef = 128 evaluates ~4 000 rows, most of them after its LDS set spilled.
constexpr uint32_t kVisUndoWords = 8192;
constexpr int ADC_LUT_STRIDE = 256;  // table row stride in floats (num_centroids <= 256: external_index.c:283-296)
// A compact pq index searched by DECODING rows on the fly: a row is its code bytes in HBM, and every 16-byte chunk of its decoding
// is fetched from the per-subvector centroid tables (786 KB at 96 x 256 x 8 floats: L2-resident) -- PqdRow below.  The arithmetic
// is the f32 metric's over the decoded row, chunk for chunk: bit-identical to the expanded form of the same index.
constexpr int M_PQD = 600;
constexpr int M_COS_PQD = M_COS + M_PQD;
constexpr int M_L2SQ_PQD = M_L2SQ + M_PQD;
__host__ __device__ inline bool mcode_is_f16(int m) { return m >= M_F16 && m < M_I8; }
__host__ __device__ inline bool mcode_is_i8(int m) { return m >= M_I8 && m < M_ADC; }
__host__ __device__ inline bool mcode_is_adc(int m) { return m >= M_ADC && m < M_PQD; }
__host__ __device__ inline bool mcode_is_pqd(int m) { return m >= M_PQD; }
__host__ __device__ inline int  mcode_base(int m) { return m >= M_PQD ? m - M_PQD : m >= M_ADC ? m - M_ADC : m >= M_I8 ? m - M_I8 : m >= M_F16 ? m - M_F16 : m; }

// every kernel's dynamic LDS; the ADC accumulators read their table from its first bytes
extern __shared__ __attribute__((aligned(16))) unsigned char lgpu_smem[];

// ---- order-preserving float <-> u32, and the (distance, slot) candidate key ---------------------
__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(b);
}
// key = distance (ordered) : slot : expanded-flag.  Total order (distance, slot); the flag is the
// lowest bit so it never changes the relative order of two different slots.
__device__ __forceinline__ uint64_t make_key(float d, uint32_t slot) { return ((uint64_t)f2ord(d) << 32) | ((uint64_t)slot << 1); }
__device__ __forceinline__ uint32_t key_slot(uint64_t k) { return (uint32_t)(k & 0xFFFFFFFFu) >> 1; }
__device__ __forceinline__ float    key_dist(uint64_t k) { return ord2f((uint32_t)(k >> 32)); }
__device__ __forceinline__ bool     key_expanded(uint64_t k) { return (k & 1u) != 0; }

// per-centre pseudo-random tie order used by the neighbour-selection heuristic (oracle/hnsw.c tie_mix)
__device__ __forceinline__ uint32_t tie_mix(uint32_t id, uint32_t centre) { return (id ^ (centre * 0x9E3779B1u)) * 0x85EBCA6Bu; }

// ---- G-lane sum in DPP adds -------------------------------------------------------------------------
// Step k adds the running sum of the lane 2^k away (butterfly, offsets ascending).  After the quad
// steps every lane of a quad holds the quad's sum, so row_half_mirror / row_mirror deliver the same
// values an xor-4 / xor-8 exchange would; row_bcast15 / row_bcast31 then carry a row's (half-wave's)
// sum into the next one.  Only lane G-1 is guaranteed to hold the full sum.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_take(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_take(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <int G, typename T> __device__ __forceinline__ T group_sum(T s)
{
    s = s + dpp_take<0xB1, 0xF>(s);                  // quad_perm [1,0,3,2]  : offset 1
    s = s + dpp_take<0x4E, 0xF>(s);                  // quad_perm [2,3,0,1]  : offset 2
    if(G >= 8) s = s + dpp_take<0x141, 0xF>(s);      // row_half_mirror      : offset 4
    if(G >= 16) s = s + dpp_take<0x140, 0xF>(s);     // row_mirror           : offset 8
    if(G >= 32) s = s + dpp_take<0x142, 0xA>(s);     // row_bcast15 -> rows 1,3 : offset 16
    if(G >= 64) s = s + dpp_take<0x143, 0xC>(s);     // row_bcast31 -> rows 2,3 : offset 32
    return s;
}

// ---- per-lane accumulation ----------------------------------------------------------------------
template <int METRIC> struct Acc;

// two f32 lanes of one VGPR pair: the differences of an l2sq chunk are independent of one another, so they are formed
// with packed subtracts (v_pk_add_f32: two IEEE subtractions per instruction, same bits); the fma CHAIN stays scalar
// and in memory order, which is what the reduction-order contract fixes.
typedef float lgpu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void l2sq_chunk(const uint4 &xa, const uint4 &yb, float &s)
{
    const lgpu_f32x2 x01 = { __uint_as_float(xa.x), __uint_as_float(xa.y) }, x23 = { __uint_as_float(xa.z), __uint_as_float(xa.w) };
    const lgpu_f32x2 y01 = { __uint_as_float(yb.x), __uint_as_float(yb.y) }, y23 = { __uint_as_float(yb.z), __uint_as_float(yb.w) };
    const lgpu_f32x2 t01 = x01 - y01, t23 = x23 - y23;
    s = __builtin_fmaf(t01[ 0 ], t01[ 0 ], s);
    s = __builtin_fmaf(t01[ 1 ], t01[ 1 ], s);
    s = __builtin_fmaf(t23[ 0 ], t23[ 0 ], s);
    s = __builtin_fmaf(t23[ 1 ], t23[ 1 ], s);
}

template <> struct Acc<M_L2SQ>
{
    float s = 0.f;
    __device__ __forceinline__ void add(const uint4 &xa, const uint4 &yb) { l2sq_chunk(xa, yb, s); }
    template <int G> __device__ __forceinline__ float finish()
    {
        return group_sum<G>(s);
    }
};

template <> struct Acc<M_COS>
{
    float ab = 0.f, a2 = 0.f, b2 = 0.f;
    __device__ __forceinline__ void one(float x, float y)
    {
        ab = __builtin_fmaf(x, y, ab);
        a2 = __builtin_fmaf(x, x, a2);
        b2 = __builtin_fmaf(y, y, b2);
    }
    __device__ __forceinline__ void add(const uint4 &xa, const uint4 &yb)
    {
        one(__uint_as_float(xa.x), __uint_as_float(yb.x));
        one(__uint_as_float(xa.y), __uint_as_float(yb.y));
        one(__uint_as_float(xa.z), __uint_as_float(yb.z));
        one(__uint_as_float(xa.w), __uint_as_float(yb.w));
    }
    template <int G> __device__ __forceinline__ float finish()
    {
        ab = group_sum<G>(ab);
        a2 = group_sum<G>(a2);
        b2 = group_sum<G>(b2);
        // zero-norm rules pinned by the reference's tests (hnsw_vector.out:205-210,
        // hnsw_dist_func.out:58-61): both zero -> 0, one zero -> 1
        if(a2 == 0.f && b2 == 0.f) return 0.f;
        if(a2 == 0.f || b2 == 0.f) return 1.f;
        return 1.f - ab / (__builtin_sqrtf(a2) * __builtin_sqrtf(b2));  // IEEE sqrt and divide (hipcc default: correctly rounded)
    }
};

template <> struct Acc<M_HAMMING>
{
    uint32_t s = 0;
    __device__ __forceinline__ void add(const uint4 &xa, const uint4 &yb)
    {
        s += __popc(xa.x ^ yb.x) + __popc(xa.y ^ yb.y) + __popc(xa.z ^ yb.z) + __popc(xa.w ^ yb.w);
    }
    template <int G> __device__ __forceinline__ float finish()
    {
        return (float)group_sum<G>(s);
    }
};

template <> struct Acc<M_COS_B1>
{
    uint32_t ab = 0, a2 = 0, b2 = 0;
    __device__ __forceinline__ void add(const uint4 &xa, const uint4 &yb)
    {
        ab += __popc(xa.x & yb.x) + __popc(xa.y & yb.y) + __popc(xa.z & yb.z) + __popc(xa.w & yb.w);
        a2 += __popc(xa.x) + __popc(xa.y) + __popc(xa.z) + __popc(xa.w);
        b2 += __popc(yb.x) + __popc(yb.y) + __popc(yb.z) + __popc(yb.w);
    }
    template <int G> __device__ __forceinline__ float finish()
    {
        const float fab = (float)group_sum<G>(ab), fa2 = (float)group_sum<G>(a2), fb2 = (float)group_sum<G>(b2);
        if(fa2 == 0.f && fb2 == 0.f) return 0.f;  // the zero-norm rules of the f32 metric
        if(fa2 == 0.f || fb2 == 0.f) return 1.f;
        return 1.f - fab / (__builtin_sqrtf(fa2) * __builtin_sqrtf(fb2));
    }
};

// ---- f16 storage: two halves per 32-bit word, low half first ---------------------------------------------
typedef _Float16 lgpu_half2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void unpack_h2(uint32_t w, float &lo, float &hi)
{
    const lgpu_half2 v = __builtin_bit_cast(lgpu_half2, w);
    lo = (float)v[ 0 ];
    hi = (float)v[ 1 ];
}

template <> struct Acc<M_L2SQ_F16>
{
    float s = 0.f;
    __device__ __forceinline__ void word(uint32_t xw, uint32_t yw)
    {
        float x0, x1, y0, y1, t;
        unpack_h2(xw, x0, x1);
        unpack_h2(yw, y0, y1);
        t = x0 - y0; s = __builtin_fmaf(t, t, s);
        t = x1 - y1; s = __builtin_fmaf(t, t, s);
    }
    __device__ __forceinline__ void add(const uint4 &xa, const uint4 &yb)
    {
        word(xa.x, yb.x); word(xa.y, yb.y); word(xa.z, yb.z); word(xa.w, yb.w);
    }
    template <int G> __device__ __forceinline__ float finish() { return group_sum<G>(s); }
};

template <> struct Acc<M_COS_F16>
{
    Acc<M_COS> f;
    __device__ __forceinline__ void word(uint32_t xw, uint32_t yw)
    {
        float x0, x1, y0, y1;
        unpack_h2(xw, x0, x1);
        unpack_h2(yw, y0, y1);
        f.one(x0, y0);
        f.one(x1, y1);
    }
    __device__ __forceinline__ void add(const uint4 &xa, const uint4 &yb)
    {
        word(xa.x, yb.x); word(xa.y, yb.y); word(xa.z, yb.z); word(xa.w, yb.w);
    }
    template <int G> __device__ __forceinline__ float finish() { return f.template finish<G>(); }
};

// ---- i8 storage: four signed bytes per 32-bit word, V_DOT4_I32_I8 -----------------------------------------
__device__ __forceinline__ int dot4_i8(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }

template <> struct Acc<M_L2SQ_I8>
{
    // sum (a - b)^2 = sum a^2 + sum b^2 - 2 sum ab, every term an exact int32 (|.| <= 2000 * 100^2)
    int ab = 0, a2 = 0, b2 = 0;
    __device__ __forceinline__ void word(uint32_t x, uint32_t y)
    {
        ab = dot4_i8(x, y, ab);
        a2 = dot4_i8(x, x, a2);
        b2 = dot4_i8(y, y, b2);
    }
    __device__ __forceinline__ void add(const uint4 &xa, const uint4 &yb)
    {
        word(xa.x, yb.x); word(xa.y, yb.y); word(xa.z, yb.z); word(xa.w, yb.w);
    }
    template <int G> __device__ __forceinline__ float finish()
    {
        const uint32_t s = group_sum<G>((uint32_t)(a2 + b2 - 2 * ab));
        return (float)(int)s;
    }
};

template <> struct Acc<M_COS_I8>
{
    Acc<M_L2SQ_I8> f;
    __device__ __forceinline__ void add(const uint4 &xa, const uint4 &yb) { f.add(xa, yb); }
    template <int G> __device__ __forceinline__ float finish()
    {
        const float ab = (float)(int)group_sum<G>((uint32_t)f.ab);
        const float a2 = (float)(int)group_sum<G>((uint32_t)f.a2);
        const float b2 = (float)(int)group_sum<G>((uint32_t)f.b2);
        if(a2 == 0.f && b2 == 0.f) return 0.f;  // the zero-norm rules of the f32 metric
        if(a2 == 0.f || b2 == 0.f) return 1.f;
        return 1.f - ab / (__builtin_sqrtf(a2) * __builtin_sqrtf(b2));
    }
};

// One distance by one G-lane group: a and b each `chunks` uint4 long; gl = lane index in group.
// The value is complete in the LAST lane of the group (gl == G-1); other lanes hold partial sums.
template <int METRIC, int G, typename PA, typename PB>
__device__ __forceinline__ float group_dist(PA a, PB b, int chunks, int gl)
{
    Acc<METRIC> acc;
#pragma unroll 4
    for(int ch = gl; ch < chunks; ch += G) {
        uint4 x = a[ ch ];
        uint4 y = b[ ch ];
        acc.add(x, y);
    }
    return acc.template finish<G>();
}

// Two rows against the same `a` at once (twice the loads in flight per lane).
template <int METRIC, int G, typename PA, typename PB>
__device__ __forceinline__ void group_dist2(PA a, PB b0, PB b1, int chunks, int gl, float &d0, float &d1)
{
    Acc<METRIC> acc0, acc1;
#pragma unroll 4
    for(int ch = gl; ch < chunks; ch += G) {
        uint4 x = a[ ch ];
        uint4 y0 = b0[ ch ];
        uint4 y1 = b1[ ch ];
        acc0.add(x, y0);
        acc1.add(x, y1);
    }
    d0 = acc0.template finish<G>();
    d1 = acc1.template finish<G>();
}

// ---- cached row norms for the cosine metrics ----------------------------------------------------------------
// Acc<M_COS> runs three independent fma chains per lane (ab, a2, b2) and three G-lane sums.  a2 depends on the first
// operand only and b2 on the second only, so for STORED rows both are properties of the row: they are computed once,
// when the row enters the index, by the very chain and tree Acc<M_COS> would run (NormAcc + group_sum<G>, G fixed per
// index), and kept -- already square-rooted: IEEE sqrt is a function of its argument -- in View::norm2.  A row
// evaluation is then ONE chain (ab), ONE G-lane sum, one multiply and one divide: the bits of
// 1 - ab / (sqrt(a2) * sqrt(b2)) are those of the one-pass accumulator (usearch metric_cos_gt).  The query's
// sqrt(a2) is computed once per query the same way.
template <int METRIC> constexpr bool kCachedNorms = (METRIC == M_COS || METRIC == M_COS_F16 || METRIC == M_COS_ADC || METRIC == M_COS_PQD);

__device__ __forceinline__ float cos_finish(float ab, float a2, float b2)
{
    // zero-norm rules pinned by the reference's tests (hnsw_vector.out:205-210, hnsw_dist_func.out:58-61)
    if(a2 == 0.f && b2 == 0.f) return 0.f;
    if(a2 == 0.f || b2 == 0.f) return 1.f;
    return 1.f - ab / (__builtin_sqrtf(a2) * __builtin_sqrtf(b2));
}

// the same with ra = sqrt(a2), rb = sqrt(b2) already taken (sqrt(x) == 0 iff x == 0)
__device__ __forceinline__ float cos_finish_rooted(float ab, float ra, float rb)
{
    if(ra == 0.f && rb == 0.f) return 0.f;
    if(ra == 0.f || rb == 0.f) return 1.f;
    return 1.f - ab / (ra * rb);
}

// ||row||^2 of one operand: the a2 (= b2) chain of Acc<M_COS> / Acc<M_COS_F16>
template <int METRIC> struct NormAcc
{
    float s = 0.f;
    __device__ __forceinline__ void add(const uint4 &z)
    {
        if constexpr(METRIC == M_COS_F16) {
            const uint32_t w[ 4 ] = { z.x, z.y, z.z, z.w };
#pragma unroll
            for(int i = 0; i < 4; ++i) {
                float lo, hi;
                unpack_h2(w[ i ], lo, hi);
                s = __builtin_fmaf(lo, lo, s);
                s = __builtin_fmaf(hi, hi, s);
            }
        } else {
            s = __builtin_fmaf(__uint_as_float(z.x), __uint_as_float(z.x), s);
            s = __builtin_fmaf(__uint_as_float(z.y), __uint_as_float(z.y), s);
            s = __builtin_fmaf(__uint_as_float(z.z), __uint_as_float(z.z), s);
            s = __builtin_fmaf(__uint_as_float(z.w), __uint_as_float(z.w), s);
        }
    }
};
// sqrt(||row||^2), complete in the LAST lane of the group
template <int METRIC, int G, typename PA> __device__ __forceinline__ float group_norm(PA a, int chunks, int gl)
{
    NormAcc<METRIC> acc;
#pragma unroll 4
    for(int ch = gl; ch < chunks; ch += G) {
        uint4 x = a[ ch ];
        acc.add(x);
    }
    return __builtin_sqrtf(group_sum<G>(acc.s));
}

// The accumulator of one (row, row) evaluation when both norms are known: the ab chain only for the cosine metrics,
// the full Acc otherwise (the norms are ignored).
template <int METRIC> struct RowAcc : Acc<METRIC>
{
    template <int G> __device__ __forceinline__ float finish_n(float, float) { return this->template finish<G>(); }
};
template <> struct RowAcc<M_COS>
{
    float s = 0.f;
    __device__ __forceinline__ void add(const uint4 &xa, const uint4 &yb)
    {
        s = __builtin_fmaf(__uint_as_float(xa.x), __uint_as_float(yb.x), s);
        s = __builtin_fmaf(__uint_as_float(xa.y), __uint_as_float(yb.y), s);
        s = __builtin_fmaf(__uint_as_float(xa.z), __uint_as_float(yb.z), s);
        s = __builtin_fmaf(__uint_as_float(xa.w), __uint_as_float(yb.w), s);
    }
    template <int G> __device__ __forceinline__ float finish_n(float ra, float rb) { return cos_finish_rooted(group_sum<G>(s), ra, rb); }
};
template <> struct RowAcc<M_COS_F16>
{
    float s = 0.f;
    __device__ __forceinline__ void word(uint32_t xw, uint32_t yw)
    {
        float x0, x1, y0, y1;
        unpack_h2(xw, x0, x1);
        unpack_h2(yw, y0, y1);
        s = __builtin_fmaf(x0, y0, s);
        s = __builtin_fmaf(x1, y1, s);
    }
    __device__ __forceinline__ void add(const uint4 &xa, const uint4 &yb) { word(xa.x, yb.x); word(xa.y, yb.y); word(xa.z, yb.z); word(xa.w, yb.w); }
    template <int G> __device__ __forceinline__ float finish_n(float ra, float rb) { return cos_finish_rooted(group_sum<G>(s), ra, rb); }
};

// ---- ADC: the "row" is 16 code bytes per chunk, the first operand only says WHICH chunk (AdcQuery below); every code adds
// one table entry, in code order -- a plain f32 addition chain per lane, then the G-lane tree (the oracle: lo_adc_distance).
// Padding codes (chunks are 16 codes wide) are 0 and hit table rows whose entry 0 is +0.0: adding it changes nothing.
struct AdcQuery
{
    __device__ __forceinline__ uint4 operator[](int ch) const { return make_uint4((uint32_t)ch, 0u, 0u, 0u); }
};
__device__ __forceinline__ float adc_chunk_sum(float s, uint32_t chunk, const uint4 &codes)
{
    const float   *lut = (const float *)lgpu_smem + (size_t)chunk * 16 * ADC_LUT_STRIDE;
    const uint32_t w[ 4 ] = { codes.x, codes.y, codes.z, codes.w };
#pragma unroll
    for(int i = 0; i < 4; ++i) {
#pragma unroll
        for(int b = 0; b < 4; ++b) s = s + lut[ (i * 4 + b) * ADC_LUT_STRIDE + ((w[ i ] >> (8 * b)) & 255u) ];
    }
    return s;
}
template <> struct RowAcc<M_L2SQ_ADC>
{
    float s = 0.f;
    __device__ __forceinline__ void add(const uint4 &which, const uint4 &codes) { s = adc_chunk_sum(s, which.x, codes); }
    template <int G> __device__ __forceinline__ float finish_n(float, float) { return group_sum<G>(s); }
};
template <> struct RowAcc<M_COS_ADC>
{
    float s = 0.f;
    __device__ __forceinline__ void add(const uint4 &which, const uint4 &codes) { s = adc_chunk_sum(s, which.x, codes); }
    template <int G> __device__ __forceinline__ float finish_n(float ra, float rb) { return cos_finish_rooted(group_sum<G>(s), ra, rb); }
};

// decode-on-the-fly: the f32 accumulators, fed with decoded chunks
template <> struct Acc<M_L2SQ_PQD> : Acc<M_L2SQ> {};
template <> struct Acc<M_COS_PQD> : Acc<M_COS> {};
template <> struct RowAcc<M_L2SQ_PQD> : RowAcc<M_L2SQ> {};
template <> struct RowAcc<M_COS_PQD> : RowAcc<M_COS> {};

// group_dist with known (rooted) norms of `a` and `b`; complete in the LAST lane of the group
template <int METRIC, int G, typename PA, typename PB>
__device__ __forceinline__ float group_dist_n(PA a, PB b, int chunks, int gl, float a2, float b2)
{
    RowAcc<METRIC> acc;
#if LGPU_ROW_BLOCK1 > 0
    // one row: both operands' loads of a block of four steps are in flight together (either may be a row in HBM)
    constexpr int B = LGPU_ROW_BLOCK1;
    for(int base = gl; base < chunks; base += B * G) {
        uint4 x[ B ], y[ B ];
#pragma unroll
        for(int c = 0; c < B; ++c)
            if(base + c * G < chunks) {
                x[ c ] = a[ base + c * G ];
                y[ c ] = b[ base + c * G ];
            }
#pragma unroll
        for(int c = 0; c < B; ++c)
            if(base + c * G < chunks) acc.add(x[ c ], y[ c ]);
    }
#else
#pragma unroll 4
    for(int ch = gl; ch < chunks; ch += G) {
        uint4 x = a[ ch ];
        uint4 y = b[ ch ];
        acc.add(x, y);
    }
#endif
    return acc.template finish_n<G>(a2, b2);
}
template <int METRIC, int G, typename PA, typename PB>
__device__ __forceinline__ void group_dist2_n(PA a, PB b0, PB b1, int chunks, int gl, float a2, float b2_0, float b2_1, float &d0, float &d1)
{
    RowAcc<METRIC> acc0, acc1;
#if LGPU_ROW_BLOCK > 0
    // All loads of a block of LGPU_ROW_BLOCK steps are issued before the first is consumed (2 x LGPU_ROW_BLOCK x 16 bytes per lane in
    // flight instead of 2 x 16: a 768-d row pair is ONE memory round trip, not three).  The additions keep their order: same bits.
    constexpr int B = (METRIC % 100 == M_COS) ? LGPU_ROW_BLOCK_COS : LGPU_ROW_BLOCK;
    for(int base = gl; base < chunks; base += B * G) {
        uint4 y0[ B ], y1[ B ];
#pragma unroll
        for(int c = 0; c < B; ++c)
            if(base + c * G < chunks) {
                y0[ c ] = b0[ base + c * G ];
                y1[ c ] = b1[ base + c * G ];
            }
#pragma unroll
        for(int c = 0; c < B; ++c)
            if(base + c * G < chunks) {
                uint4 x = a[ base + c * G ];
                acc0.add(x, y0[ c ]);
                acc1.add(x, y1[ c ]);
            }
    }
#else
#pragma unroll 4
    for(int ch = gl; ch < chunks; ch += G) {
        uint4 x = a[ ch ];
        uint4 y0 = b0[ ch ];
        uint4 y1 = b1[ ch ];
        acc0.add(x, y0);
        acc1.add(x, y1);
    }
#endif
    d0 = acc0.template finish_n<G>(a2, b2_0);
    d1 = acc1.template finish_n<G>(a2, b2_1);
}

// R rows against the same `a` at once (R x the loads in flight per lane): the small-batch shape of the walk, where a CU
// holds fewer workgroups and each must keep more bytes in flight.  Every row's chain and tree are those of group_dist_n.
template <int METRIC, int G, int R, typename PA, typename PB>
__device__ __forceinline__ void group_distR_n(PA a, const PB (&b)[ R ], int chunks, int gl, float a2, const float (&b2)[ R ], float (&d)[ R ])
{
    RowAcc<METRIC> acc[ R ];
#if LGPU_ROW_BLOCK > 0
    constexpr int B = R <= 2 ? 3 : 2;  // R x B x 16 bytes per lane in flight
    for(int base = gl; base < chunks; base += B * G) {
        uint4 y[ B ][ R ];
#pragma unroll
        for(int c = 0; c < B; ++c)
            if(base + c * G < chunks) {
#pragma unroll
                for(int r = 0; r < R; ++r) y[ c ][ r ] = b[ r ][ base + c * G ];
            }
#pragma unroll
        for(int c = 0; c < B; ++c)
            if(base + c * G < chunks) {
                uint4 x = a[ base + c * G ];
#pragma unroll
                for(int r = 0; r < R; ++r) acc[ r ].add(x, y[ c ][ r ]);
            }
    }
#else
#pragma unroll 2
    for(int ch = gl; ch < chunks; ch += G) {
        uint4 x = a[ ch ];
        uint4 y[ R ];
#pragma unroll
        for(int r = 0; r < R; ++r) y[ r ] = b[ r ][ ch ];
#pragma unroll
        for(int r = 0; r < R; ++r) acc[ r ].add(x, y[ r ]);
    }
#endif
#pragma unroll
    for(int r = 0; r < R; ++r) d[ r ] = acc[ r ].template finish_n<G>(a2, b2[ r ]);
}

// lanes per row for a row of `chunks` 16-byte chunks (oracle: lo_wave_group_lanes)
// Every lane should own at least two chunks (two 16-byte loads in flight per row per lane), so short rows are
// shared by fewer lanes and a wave works on several rows at once: G = 64 from 128 chunks (d >= 512 f32 /
// 1024 f16), 32 from 64, 16 from 32, else 8.
__host__ __device__ inline int group_lanes_for(uint32_t chunks) { return chunks >= 128 ? 64 : chunks >= 64 ? 32 : chunks >= 32 ? 16 : 8; }

// ---- read-only view of the index in HBM -----------------------------------------------------------
struct View
{
    const uint4    *vec;       // [n][chunks] rows, zero padded
    uint32_t        chunks;    // uint4 per row
    uint32_t        M, M0;     // neighbours per upper level / level 0 (2M)
    uint32_t       *nbr0;      // [cap][M0], EMPTY-terminated
    const uint32_t *upper_off; // [cap] first upper block of the node (levels 1..L are consecutive)
    uint32_t       *upper_nbr; // [blocks][M]
    const uint8_t  *levels;    // [cap]
    const float    *norm2;     // [cap] sqrt(||row||^2) for the cosine metrics (the a2 / b2 chain of Acc<M_COS>, then IEEE sqrt); else NULL
    uint32_t        n;
    uint32_t        entry;
    int32_t         max_level;
    // decode-on-the-fly metrics (M_*_PQD) only: vec = the CODE rows (pq_row_bytes each), chunks = chunks of a DECODED row
    const uint4    *pq_centers;   // [S][C][pq_cps] the per-subvector centroid tables, a centroid = pq_cps whole chunks
    uint32_t        pq_cps;       // chunks per subvector (subvector dimensions / 4)
    uint32_t        pq_C;         // centroids per subvector
    uint32_t        pq_inv;       // ceil(2^16 / pq_cps): chunk / pq_cps == (chunk * pq_inv) >> 16 for chunk < 2^11 (host-checked)
    uint32_t        pq_row_bytes; // bytes of one code row (num_subvectors padded to 16)
};

__device__ __forceinline__ const uint4 *row_of(const View &v, uint32_t slot) { return v.vec + (size_t)slot * v.chunks; }

// A row of a compact pq index as the walk sees it: indexable by decoded chunk.  Chunk ch lies in subvector ch / cps; its
// bytes are chunk ch % cps of centroid `code` of that subvector: one byte from the row's codes (HBM: the whole 96-byte row is
// one or two cache lines), then 16 bytes from the centroid table (L2).  Which subvector and which offset a lane's chunks fall
// into does not depend on the row: the compiler hoists that arithmetic out of the row loops.
struct PqdRow
{
    const uint8_t *codes;
    const uint4   *centers;
    uint32_t       cps, C, inv;
    __device__ __forceinline__ uint4 operator[](int ch) const
    {
        const uint32_t sv = ((uint32_t)ch * inv) >> 16, off = (uint32_t)ch - sv * cps;
        const uint32_t code = codes[ sv ];
        return centers[ ((size_t)sv * C + code) * cps + off ];
    }
};
template <int METRIC> __device__ __forceinline__ auto row_of_m(const View &v, uint32_t slot)
{
    if constexpr(METRIC >= M_PQD) return PqdRow{ (const uint8_t *)v.vec + (size_t)slot * v.pq_row_bytes, v.pq_centers, v.pq_cps, v.pq_C, v.pq_inv };
    else return v.vec + (size_t)slot * v.chunks;
}
// cached ||row||^2 (cosine metrics); 0 and no memory access for the others
template <int METRIC> __device__ __forceinline__ float row_norm(const View &v, uint32_t slot)
{
    if constexpr(kCachedNorms<METRIC>) return v.norm2[ slot ];
    else return 0.f;
}

__device__ __forceinline__ uint32_t *neighbors_of(const View &v, uint32_t slot, int level, uint32_t &cap)
{
    if(level == 0) {
        cap = v.M0;
        return v.nbr0 + (size_t)slot * v.M0;
    }
    cap = v.M;
    return v.upper_nbr + ((size_t)v.upper_off[ slot ] + (size_t)(level - 1)) * v.M;
}

}  // namespace lgpu
