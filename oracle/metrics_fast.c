/*
 * metrics_fast.c -- the same usearch metric loops as metrics.c's SEQ order, but built with
 * the flags the reference builds usearch with (lantern_hnsw/CMakeLists.txt:122-136:
 * -ftree-vectorize -fassociative-math -fno-signed-zeros -fno-trapping-math [-march=native];
 * SimSIMD off, CMakeLists.txt:26).  Association order is then whatever the vectoriser picks,
 * exactly as in the reference.  This is the variant the CPU baseline times.
 * TEST INFRASTRUCTURE (see lantern_oracle.h).
 */
#include "lantern_oracle.h"

#include <math.h>

float lo_distance_fast(const void *pa, const void *pb, size_t d, int metric)
{
    const float *restrict a = (const float *)pa;
    const float *restrict b = (const float *)pb;
    if(metric == LO_METRIC_L2SQ) {
        float s = 0.f;
        for(size_t i = 0; i != d; ++i) {
            float t = a[ i ] - b[ i ];
            s += t * t;
        }
        return s;
    }
    if(metric == LO_METRIC_COS) {
        float ab = 0.f, a2 = 0.f, b2 = 0.f;
        for(size_t i = 0; i != d; ++i) {
            ab += a[ i ] * b[ i ];
            a2 += a[ i ] * a[ i ];
            b2 += b[ i ] * b[ i ];
        }
        if(a2 == 0.f && b2 == 0.f) return 0.f;
        if(a2 == 0.f || b2 == 0.f) return 1.f;
        return 1.f - ab / (sqrtf(a2) * sqrtf(b2));
    }
    return NAN;
}

/* hamming over whole u64 words for the timed baseline (bits must be a multiple of 8) */
float lo_hamming_fast(const void *pa, const void *pb, size_t bits)
{
    const unsigned char *a = (const unsigned char *)pa, *b = (const unsigned char *)pb;
    size_t               bytes = (bits + 7) / 8, i = 0;
    unsigned long long   total = 0;
    for(; i + 8 <= bytes; i += 8) {
        unsigned long long x, y;
        __builtin_memcpy(&x, a + i, 8);
        __builtin_memcpy(&y, b + i, 8);
        total += (unsigned long long)__builtin_popcountll(x ^ y);
    }
    for(; i < bytes; ++i) total += (unsigned long long)__builtin_popcount((unsigned)(a[ i ] ^ b[ i ]));
    return (float)total;
}
