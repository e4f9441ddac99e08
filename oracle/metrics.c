/*
 * metrics.c -- oracle pairwise metrics (TEST INFRASTRUCTURE; see lantern_oracle.h).
 *
 * Restates usearch's metric_l2sq_gt / metric_cos_gt / metric_hamming_gt as reached through
 * usearch_distance from lantern_hnsw/src/hnsw.c:296-345 (array_dist, vector_dist) and
 * product_quantization.c:102,185.  Compile this file with -ffp-contract=off so that the
 * SEQ order really is "one multiply, one add per element" and the WAVE64 order really is
 * the explicit fmaf() chain it spells out.
 */
#include "lantern_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <string.h>

float lo_distance_fast(const void *a, const void *b, size_t dims, int metric); /* metrics_fast.c */

/* ---- LO_SUM_SEQ: the textbook usearch loops ------------------------------------------- */

static float l2sq_seq(const float *a, const float *b, size_t d)
{
    float s = 0.f;
    for(size_t i = 0; i != d; ++i) {
        float t = a[ i ] - b[ i ];
        s += t * t;
    }
    return s;
}

/* cosine zero-norm rules are pinned by the reference's tests:
 *   both zero -> 0   (expected/hnsw_vector.out:205-210)
 *   one zero  -> 1   (expected/hnsw_dist_func.out:58-61,90; hnsw_operators.out:99-103) */
static float cos_finish(float ab, float a2, float b2)
{
    if(a2 == 0.f && b2 == 0.f) return 0.f;
    if(a2 == 0.f || b2 == 0.f) return 1.f;
    return 1.f - ab / (sqrtf(a2) * sqrtf(b2));
}

static float cos_seq(const float *a, const float *b, size_t d)
{
    float ab = 0.f, a2 = 0.f, b2 = 0.f;
    for(size_t i = 0; i != d; ++i) {
        ab += a[ i ] * b[ i ];
        a2 += a[ i ] * a[ i ];
        b2 += b[ i ] * b[ i ];
    }
    return cos_finish(ab, a2, b2);
}

/* hamming over b1x8: dims is a BIT count (hnsw.c:317-319 passes a_dim*32); popcount of XOR */
static float hamming_bits(const uint8_t *a, const uint8_t *b, size_t bits)
{
    size_t   bytes = (bits + 7) / 8;
    uint64_t total = 0;
    for(size_t i = 0; i != bytes; ++i) total += (uint64_t)__builtin_popcount((unsigned)(a[ i ] ^ b[ i ]));
    return (float)total;
}

/* cosine of the {0, 1} vectors: integer-exact popcounts, then the f32 metric's finish (device_common.hpp Acc<M_COS_B1>) */
static float cos_bits(const uint8_t *a, const uint8_t *b, size_t bits)
{
    size_t   bytes = (bits + 7) / 8;
    uint32_t ab = 0, a2 = 0, b2 = 0;
    for(size_t i = 0; i != bytes; ++i) {
        ab += (uint32_t)__builtin_popcount((unsigned)(a[ i ] & b[ i ]));
        a2 += (uint32_t)__builtin_popcount((unsigned)a[ i ]);
        b2 += (uint32_t)__builtin_popcount((unsigned)b[ i ]);
    }
    return cos_finish((float)ab, (float)a2, (float)b2);
}

/* ---- LO_SUM_WAVE64: the device reduction tree ------------------- */
/*
 * A row of d f32 scalars is zero-padded to d4 = 4*ceil(d/4).  G lanes cooperate
 * (G = lo_wave_group_lanes(d)).  Lane l owns the float4 chunks l, l+G, l+2G, ... and runs
 * one fmaf chain per accumulator over its scalars in memory order.  The G partials are then
 * combined by a butterfly with ascending offsets: for off = 1, 2, .. G/2: p[l] = p[l] + p[l ^ off]
 * (the device executes it as DPP adds and reads the sum from lane G-1; every lane of a true xor
 * butterfly holds the same bits, so p[0] below is that value).
 */
static int group_lanes(size_t dims, size_t epc)
{
    size_t chunks = (dims + epc - 1) / epc; /* device rule (device_common.hpp group_lanes_for): >= 2 chunks per lane */
    if(chunks >= 128) return 64;
    if(chunks >= 64) return 32;
    if(chunks >= 32) return 16;
    return 8;
}
int lo_wave_group_lanes(size_t dims) { return group_lanes(dims, 4); }

static void butterfly(float *p, int G)
{
#if defined(__AVX2__)
    /* p[l] + p[l ^ off] for all l, one vector add per eight lanes: the same IEEE additions as the loop below */
    __m256 v[ 8 ];
    const int nv = G / 8;
    for(int i = 0; i < nv; ++i) v[ i ] = _mm256_loadu_ps(p + 8 * i);
    for(int i = 0; i < nv; ++i) v[ i ] = _mm256_add_ps(v[ i ], _mm256_permute_ps(v[ i ], 0xB1));        /* off 1 */
    for(int i = 0; i < nv; ++i) v[ i ] = _mm256_add_ps(v[ i ], _mm256_permute_ps(v[ i ], 0x4E));        /* off 2 */
    for(int i = 0; i < nv; ++i) v[ i ] = _mm256_add_ps(v[ i ], _mm256_permute2f128_ps(v[ i ], v[ i ], 1)); /* off 4 */
    for(int off = 1; off < nv; off <<= 1) {                                                              /* off 8, 16, 32 */
        __m256 t[ 8 ];
        for(int i = 0; i < nv; ++i) t[ i ] = _mm256_add_ps(v[ i ], v[ i ^ off ]);
        for(int i = 0; i < nv; ++i) v[ i ] = t[ i ];
    }
    for(int i = 0; i < nv; ++i) _mm256_storeu_ps(p + 8 * i, v[ i ]);
#else
    float t[ 64 ];
    for(int off = 1; off < G; off <<= 1) {
        for(int l = 0; l < G; ++l) t[ l ] = p[ l ] + p[ l ^ off ];
        memcpy(p, t, sizeof(float) * (size_t)G);
    }
#endif
}

/* Eight lanes of the tree at once (AVX2 + FMA; the oracle is built for x86-64-v3 or better): lanes l..l+7 own eight
 * consecutive chunks, i.e. 32 consecutive floats; a 4 x 4 transpose inside each 128-bit half turns them into four
 * vectors "element c of lanes l..l+7", and one vfmadd per element advances the eight lanes' chains -- each lane's chain
 * still runs over its own scalars in memory order, and vfmadd IS fmaf, so the bits are those of the scalar loops below
 * (tests/test_oracle_golden.py compares the two on ragged shapes; the golden regression file pins both). */
#if defined(__AVX2__) && defined(__FMA__)
#define LO_WAVE_SIMD 1
static inline void load8x4(const float *p, __m256 *c0, __m256 *c1, __m256 *c2, __m256 *c3)
{
    /* r_k = [lane l+k | lane l+4+k], each half one 16-byte chunk */
    __m256 r0 = _mm256_insertf128_ps(_mm256_castps128_ps256(_mm_loadu_ps(p)), _mm_loadu_ps(p + 16), 1);
    __m256 r1 = _mm256_insertf128_ps(_mm256_castps128_ps256(_mm_loadu_ps(p + 4)), _mm_loadu_ps(p + 20), 1);
    __m256 r2 = _mm256_insertf128_ps(_mm256_castps128_ps256(_mm_loadu_ps(p + 8)), _mm_loadu_ps(p + 24), 1);
    __m256 r3 = _mm256_insertf128_ps(_mm256_castps128_ps256(_mm_loadu_ps(p + 12)), _mm_loadu_ps(p + 28), 1);
    __m256 t0 = _mm256_unpacklo_ps(r0, r1), t1 = _mm256_unpackhi_ps(r0, r1);
    __m256 t2 = _mm256_unpacklo_ps(r2, r3), t3 = _mm256_unpackhi_ps(r2, r3);
    *c0 = _mm256_shuffle_ps(t0, t2, 0x44);
    *c1 = _mm256_shuffle_ps(t0, t2, 0xEE);
    *c2 = _mm256_shuffle_ps(t1, t3, 0x44);
    *c3 = _mm256_shuffle_ps(t1, t3, 0xEE);
}
#else
#define LO_WAVE_SIMD 0
#endif
int lo_wave_simd = LO_WAVE_SIMD; /* tests may clear it to run the scalar restatement (lo_set_wave_simd) */
void lo_set_wave_simd(int on) { lo_wave_simd = on && LO_WAVE_SIMD; }

/* epc = scalars per 16-byte chunk: 4 for f32 storage, 8 for f16 storage (LO_SUM_WAVE64_F16).  For f16
 * storage the caller passes values already rounded to f16 (usearch casts f32 -> f16 at add and at search
 * and its metric_*_gt<f16_t, f32> converts every element back to f32 before the arithmetic). */
static float l2sq_wave(const float *a, const float *b, size_t d, size_t epc)
{
    int    G = group_lanes(d, epc);
    size_t chunks = (d + epc - 1) / epc;
    float  p[ 64 ];
#if LO_WAVE_SIMD
    if(epc == 4 && lo_wave_simd) {
        const size_t whole = d / 4; /* chunks that are complete in memory; a ragged last chunk takes the scalar tail */
        for(int l = 0; l < G; l += 8) {
            __m256 acc = _mm256_setzero_ps();
            size_t ch = (size_t)l;
            for(; ch + 8 <= whole; ch += (size_t)G) {
                /* the differences are element-wise: subtract in memory layout, transpose once */
                float  df[ 32 ];
                for(int v = 0; v < 4; ++v)
                    _mm256_storeu_ps(df + 8 * v, _mm256_sub_ps(_mm256_loadu_ps(a + ch * 4 + 8 * v), _mm256_loadu_ps(b + ch * 4 + 8 * v)));
                __m256 t0, t1, t2, t3;
                load8x4(df, &t0, &t1, &t2, &t3);
                acc = _mm256_fmadd_ps(t0, t0, acc);
                acc = _mm256_fmadd_ps(t1, t1, acc);
                acc = _mm256_fmadd_ps(t2, t2, acc);
                acc = _mm256_fmadd_ps(t3, t3, acc);
            }
            _mm256_storeu_ps(p + l, acc);
            for(; ch < chunks; ch += (size_t)G) /* the rounds in which the eight lanes are not all whole (a ragged tail) */
                for(int j = 0; j < 8; ++j)
                    for(size_t c = 0; c < 4; ++c) {
                        size_t i = (ch + (size_t)j) * 4 + c;
                        if(i < d) {
                            float t = a[ i ] - b[ i ];
                            p[ l + j ] = fmaf(t, t, p[ l + j ]);
                        }
                    }
        }
        butterfly(p, G);
        return p[ 0 ];
    }
#endif
    for(int l = 0; l < G; ++l) {
        float acc = 0.f;
        for(size_t ch = (size_t)l; ch < chunks; ch += (size_t)G) {
            for(size_t c = 0; c < epc; ++c) {
                size_t i = ch * epc + c;
                float  x = i < d ? a[ i ] : 0.f, y = i < d ? b[ i ] : 0.f;
                float  t = x - y;
                acc = fmaf(t, t, acc);
            }
        }
        p[ l ] = acc;
    }
    butterfly(p, G);
    return p[ 0 ];
}

static float cos_wave(const float *a, const float *b, size_t d, size_t epc)
{
    int    G = group_lanes(d, epc);
    size_t chunks = (d + epc - 1) / epc;
    float  pab[ 64 ], pa2[ 64 ], pb2[ 64 ];
#if LO_WAVE_SIMD
    if(epc == 4 && lo_wave_simd) {
        const size_t whole = d / 4;
        for(int l = 0; l < G; l += 8) {
            __m256 ab = _mm256_setzero_ps(), a2 = _mm256_setzero_ps(), b2 = _mm256_setzero_ps();
            size_t ch = (size_t)l;
            for(; ch + 8 <= whole; ch += (size_t)G) {
                __m256 x[ 4 ], y[ 4 ];
                load8x4(a + ch * 4, &x[ 0 ], &x[ 1 ], &x[ 2 ], &x[ 3 ]);
                load8x4(b + ch * 4, &y[ 0 ], &y[ 1 ], &y[ 2 ], &y[ 3 ]);
                for(int c = 0; c < 4; ++c) {
                    ab = _mm256_fmadd_ps(x[ c ], y[ c ], ab);
                    a2 = _mm256_fmadd_ps(x[ c ], x[ c ], a2);
                    b2 = _mm256_fmadd_ps(y[ c ], y[ c ], b2);
                }
            }
            _mm256_storeu_ps(pab + l, ab);
            _mm256_storeu_ps(pa2 + l, a2);
            _mm256_storeu_ps(pb2 + l, b2);
            for(; ch < chunks; ch += (size_t)G)
                for(int j = 0; j < 8; ++j)
                    for(size_t c = 0; c < 4; ++c) {
                        size_t i = (ch + (size_t)j) * 4 + c;
                        if(i < d) {
                            pab[ l + j ] = fmaf(a[ i ], b[ i ], pab[ l + j ]);
                            pa2[ l + j ] = fmaf(a[ i ], a[ i ], pa2[ l + j ]);
                            pb2[ l + j ] = fmaf(b[ i ], b[ i ], pb2[ l + j ]);
                        }
                    }
        }
        butterfly(pab, G);
        butterfly(pa2, G);
        butterfly(pb2, G);
        return cos_finish(pab[ 0 ], pa2[ 0 ], pb2[ 0 ]);
    }
#endif
    for(int l = 0; l < G; ++l) {
        float ab = 0.f, a2 = 0.f, b2 = 0.f;
        for(size_t ch = (size_t)l; ch < chunks; ch += (size_t)G) {
            for(size_t c = 0; c < epc; ++c) {
                size_t i = ch * epc + c;
                float  x = i < d ? a[ i ] : 0.f, y = i < d ? b[ i ] : 0.f;
                ab = fmaf(x, y, ab);
                a2 = fmaf(x, x, a2);
                b2 = fmaf(y, y, b2);
            }
        }
        pab[ l ] = ab;
        pa2[ l ] = a2;
        pb2[ l ] = b2;
    }
    butterfly(pab, G);
    butterfly(pa2, G);
    butterfly(pb2, G);
    return cos_finish(pab[ 0 ], pa2[ 0 ], pb2[ 0 ]);
}

/* sqrt(||a||^2) in the device's order: the a2 chain and tree of cos_wave (the cached row norms of the cosine kernels, and the
 * query norm of the ADC search) */
float lo_norm_wave(const float *a, size_t d)
{
    int    G = group_lanes(d, 4);
    size_t chunks = (d + 3) / 4;
    float  p[ 64 ];
    for(int l = 0; l < G; ++l) {
        float acc = 0.f;
        for(size_t ch = (size_t)l; ch < chunks; ch += (size_t)G)
            for(size_t c = 0; c < 4; ++c) {
                size_t i = ch * 4 + c;
                float  x = i < d ? a[ i ] : 0.f;
                acc = fmaf(x, x, acc);
            }
        p[ l ] = acc;
    }
    butterfly(p, G);
    return sqrtf(p[ 0 ]);
}
/* the G-lane tree on its own (ADC row sums: eight lanes) */
float lo_tree_sum(float *p, int G)
{
    butterfly(p, G);
    return p[ 0 ];
}

/* ---- LO_SUM_I8: usearch l2sq_i8_t / cos_i8_t -- int32 accumulators over the quantised integers ------------- */
static float l2sq_i8(const float *a, const float *b, size_t d)
{
    int32_t s = 0;
    for(size_t i = 0; i != d; ++i) {
        int32_t t = (int32_t)a[ i ] - (int32_t)b[ i ];
        s += t * t;
    }
    return (float)s;
}

static float cos_i8(const float *a, const float *b, size_t d)
{
    int32_t ab = 0, a2 = 0, b2 = 0;
    for(size_t i = 0; i != d; ++i) {
        int32_t x = (int32_t)a[ i ], y = (int32_t)b[ i ];
        ab += x * y;
        a2 += x * x;
        b2 += y * y;
    }
    return cos_finish((float)ab, (float)a2, (float)b2); /* same zero-norm rules as the f32 metric */
}

float lo_distance(const void *a, const void *b, size_t dims, int metric, int sum_mode)
{
    if(metric == LO_METRIC_HAMMING) return hamming_bits((const uint8_t *)a, (const uint8_t *)b, dims);
    if(metric == LO_METRIC_COS_B1) return cos_bits((const uint8_t *)a, (const uint8_t *)b, dims);
    if(sum_mode == LO_SUM_I8) {
        if(metric == LO_METRIC_L2SQ) return l2sq_i8((const float *)a, (const float *)b, dims);
        if(metric == LO_METRIC_COS) return cos_i8((const float *)a, (const float *)b, dims);
        return NAN;
    }
    if(sum_mode == LO_SUM_FAST) return lo_distance_fast(a, b, dims, metric);
    const float *x = (const float *)a, *y = (const float *)b;
    const int wave = sum_mode == LO_SUM_WAVE64 || sum_mode == LO_SUM_WAVE64_F16;
    const size_t epc = sum_mode == LO_SUM_WAVE64_F16 ? 8 : 4;
    if(metric == LO_METRIC_L2SQ) return wave ? l2sq_wave(x, y, dims, epc) : l2sq_seq(x, y, dims);
    if(metric == LO_METRIC_COS) return wave ? cos_wave(x, y, dims, epc) : cos_seq(x, y, dims);
    return NAN;
}
