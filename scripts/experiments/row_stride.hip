// row_stride.hip -- does the random-row fetch rate of the part depend on where 3 KiB rows START?
//
// The clustered co-headline draws its rows from DRAM at the rate of a uniformly random gather of 3 KiB rows (5.15 TB/s of the 8 TB/s
// peak, 6.29 TB/s streaming: DESIGN.md section 5).  Rows of 768 f32 sit at a 3072-byte stride, so a row starts at any multiple of
// 1 KiB and straddles 4 KiB boundaries in two cases out of four.  If the memory system interleaves channels / banks on a coarser
// grain than 1 KiB, rows padded to a 4096-byte stride (one row = one aligned 4 KiB block, +33 % HBM footprint -- affordable at
// 288 GB) could fetch faster.  This measures it before anything is built: the distance phase of a hop alone, in the walk's launch
// shape (four-wave workgroups, six per CU, two rows in flight per 64-lane group), over uniformly random rows of a table far larger
// than the Infinity Cache, at strides 3072 / 3200 / 3328 / 3584 / 4096 / 8192, and at 4096 with every row start shifted by 2 / 3 KiB.
//
//   hipcc --offload-arch=gfx950 -O3 -o row_stride row_stride.hip && ./row_stride
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while(0)

constexpr int CHUNKS = 192;  // 768 f32 = 192 x 16 bytes

__device__ __forceinline__ float l2(const uint4 u, const float4 q)
{
    const float a = __uint_as_float(u.x) - q.x, b = __uint_as_float(u.y) - q.y, c = __uint_as_float(u.z) - q.z, d = __uint_as_float(u.w) - q.w;
    return a * a + b * b + c * c + d * d;
}

__global__ void __launch_bounds__(256, 6) k_rows(const char *table, size_t stride, size_t shift, const uint32_t *ids, uint32_t n, const float4 *query, float *out)
{
    const uint32_t lane = threadIdx.x & 63, group = blockIdx.x * 4 + (threadIdx.x >> 6), ngroups = gridDim.x * 4;
    float4 q[ 3 ];
#pragma unroll
    for(int j = 0; j < 3; ++j) q[ j ] = query[ lane + 64 * j ];
    for(uint32_t i = group; i < n; i += 2 * ngroups) {
        const uint32_t j = i + ngroups < n ? i + ngroups : i;
        const uint4   *r0 = (const uint4 *)(table + (size_t)ids[ i ] * stride + shift), *r1 = (const uint4 *)(table + (size_t)ids[ j ] * stride + shift);
        uint4          a[ 3 ], b[ 3 ];
#pragma unroll
        for(int c = 0; c < 3; ++c) { a[ c ] = r0[ lane + 64 * c ]; b[ c ] = r1[ lane + 64 * c ]; }
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for(int c = 0; c < 3; ++c) { d0 += l2(a[ c ], q[ c ]); d1 += l2(b[ c ], q[ c ]); }
        for(int off = 32; off > 0; off >>= 1) { d0 += __shfl_xor(d0, off); d1 += __shfl_xor(d1, off); }
        if(lane == 63) { out[ i ] = d0; out[ j ] = d1; }
    }
}

static uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }

int main()
{
    const uint32_t rows = 1000000, evals = 17600000;  // one launch of the clustered line evaluates ~6.5 M rows, of the Gaussian line 17.6 M
    const size_t   max_stride = 8192;
    char          *table;
    CHECK(hipMalloc(&table, (size_t)rows * max_stride + 4096));
    CHECK(hipMemset(table, 0x3c, (size_t)rows * max_stride + 4096));
    std::vector<uint32_t> h(evals);
    uint32_t *ids;
    float    *out;
    float4   *query;
    CHECK(hipMalloc(&ids, (size_t)evals * 4));
    CHECK(hipMalloc(&out, (size_t)evals * 4));
    CHECK(hipMalloc(&query, CHUNKS * 16));
    CHECK(hipMemset(query, 0, CHUNKS * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * 6;
    std::printf("{\"device\": \"%s\", \"grid\": %d, \"rows\": %u, \"evaluations_per_launch\": %u, \"row_bytes\": 3072, \"cases\": [\n", prop.gcnArchName, grid, rows, evals);
    const size_t strides[] = {3072, 3200, 3328, 3584, 4096, 4096, 4096, 8192};
    const size_t shifts[] = {0, 0, 0, 0, 0, 2048, 3072, 0};  // 4096 + 2048 / 3072: every row straddles a 4 KiB boundary
    bool         first = true;
    for(int pass = 0; pass < 2; ++pass)  // pass 1 repeats pass 0 with other ids: the spread between them is the noise
        for(size_t c = 0; c < sizeof(strides) / sizeof(strides[ 0 ]); ++c) {
            float best = 1e30f, sum = 0.f;
            const int reps = 5;
            for(int r = 0; r < reps + 1; ++r) {  // the first launch warms up
                for(uint32_t i = 0; i < evals; ++i) h[ i ] = mix(i * 2654435761u + 977u * (uint32_t)(r + 16 * c + 256 * pass)) % rows;
                CHECK(hipMemcpy(ids, h.data(), (size_t)evals * 4, hipMemcpyHostToDevice));
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_rows, dim3(grid), dim3(256), 0, 0, table, strides[ c ], shifts[ c ], ids, evals, query, out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if(r == 0) continue;
                best = ms < best ? ms : best;
                sum += ms;
            }
            const double bytes = (double)evals * 3072.0;
            std::printf("%s  {\"pass\": %d, \"stride\": %zu, \"shift\": %zu, \"mean_ms\": %.4f, \"best_ms\": %.4f, \"algorithmic_TBps_mean\": %.3f, \"algorithmic_TBps_best\": %.3f}", first ? "" : ",\n",
                        pass, strides[ c ], shifts[ c ], sum / reps, best, bytes / (sum / reps * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
            first = false;
        }
    std::printf("\n]}\n");
    return 0;
}
