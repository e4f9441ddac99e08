/*
 * lantern_c_caller.c -- a plain C11 translation unit that uses include/lantern_gpu.h the way lantern_hnsw does
 * (lantern_hnsw/src/hnsw/build.c:495-597, scan.c:60-131,207-228, hnsw.c:296-345): same calls, same error
 * convention.  Compiled with gcc and linked against liblantern_gpu.so by tests/test_c_abi.py -- the header must be
 * valid C and the library must link from C, because the reference's caller is C.
 *
 * Without a HIP device every compute entry point must fail with an error string (exit code 3, "no HIP device");
 * with one, a 64-row index is built and searched (exit code 0).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lantern_gpu.h"

#define DIM 8
#define ROWS 64

static int fail(const char *what, usearch_error_t error)
{
    fprintf(stderr, "%s: %s\n", what, error ? error : "(no message)");
    return 1;
}

int main(void)
{
    usearch_error_t        error = NULL;
    usearch_init_options_t opts;
    memset(&opts, 0, sizeof(opts));
    /* utils.c:57-67 PopulateUsearchOpts */
    opts.metric_kind = usearch_metric_l2sq_k;
    opts.metric = NULL;
    opts.quantization = usearch_scalar_f32_k;
    opts.dimensions = DIM;
    opts.connectivity = 4;
    opts.expansion_add = 16;
    opts.expansion_search = 16;
    opts.num_threads = 1;

    /* header helpers work anywhere */
    char header[ USEARCH_HEADER_SIZE ];
    memset(header, 0, sizeof(header));
    usearch_header_set_entry_slot(header, 42);
    if(usearch_header_get_entry_slot(header) != 42) return fail("entry slot round trip", NULL);

    /* argument validation precedes device use: hnsw.c:301-303 */
    float a3[ 3 ] = { 0, 1, 0 }, b2[ 2 ] = { 1, 1 };
    (void)lantern_l2sq_dist(b2, 2, a3, 3, &error);
    if(!error || !strstr(error, "expected equally sized arrays but got arrays with dimensions 2 and 3")) return fail("dimension check", error);

    usearch_index_t index = usearch_init(&opts, NULL, &error);
    if(lantern_gpu_device_count() <= 0) {
        if(index != NULL || !error || !strstr(error, "no HIP device")) return fail("expected the no-device error", error);
        printf("no device: %s\n", error);
        return 3;
    }
    if(error) return fail("usearch_init", error);
    usearch_reserve(index, ROWS, &error);
    if(error) return fail("usearch_reserve", error);
    float rows[ ROWS ][ DIM ];
    for(int i = 0; i < ROWS; ++i)
        for(int j = 0; j < DIM; ++j) rows[ i ][ j ] = (float)((i * 7 + j * 3) % 11) - 5.0f + 0.01f * (float)i;
    for(int i = 0; i < ROWS; ++i) {
        usearch_add(index, (usearch_label_t)(i + 1), rows[ i ], usearch_scalar_f32_k, &error); /* build.c:128 */
        if(error) return fail("usearch_add", error);
    }
    if(usearch_size(index, &error) != ROWS) return fail("usearch_size", error);
    usearch_label_t labels[ 5 ];
    float           distances[ 5 ];
    size_t          n = usearch_search_ef(index, rows[ 17 ], usearch_scalar_f32_k, 5, 0, false, labels, distances, &error); /* scan.c:220 */
    if(error) return fail("usearch_search_ef", error);
    if(n != 5 || labels[ 0 ] != 18 || distances[ 0 ] != 0.0f) return fail("row 17 must find itself first", NULL);
    for(size_t i = 1; i < n; ++i)
        if(distances[ i ] < distances[ i - 1 ]) return fail("distances must ascend", NULL);
    float d = usearch_distance(rows[ 0 ], rows[ 1 ], usearch_scalar_f32_k, DIM, usearch_metric_l2sq_k, &error); /* hnsw.c:340 */
    if(error) return fail("usearch_distance", error);
    float ref = 0.f;
    for(int j = 0; j < DIM; ++j) ref += (rows[ 0 ][ j ] - rows[ 1 ][ j ]) * (rows[ 0 ][ j ] - rows[ 1 ][ j ]);
    if(d < ref * 0.9999f || d > ref * 1.0001f) return fail("usearch_distance value", NULL);
    metadata_t meta = usearch_index_metadata(index, &error); /* build.c:561 */
    if(meta.neighbors_base_bytes != 4 + 2 * 4 * LANTERN_SLOT_SIZE) return fail("metadata", error);
    usearch_free(index, &error);
    printf("ok: %zu results, nearest label %llu\n", n, (unsigned long long)labels[ 0 ]);
    return 0;
}
