// gather_mlp.hip -- is the random-row fetch rate limited by the bytes a CU keeps in flight?
//
// The walk keeps 2 rows x 3 x 16 bytes per lane in flight (24 VGPRs of row data, 24 waves per CU: 147 KB per CU) and runs at the rate
// of a gather of the same shape (bench.py roofline.gather).  Before building anything that lifts that limit (LDS-direct loads, fewer
// workgroups with more registers), this measures what more bytes in flight would buy: the same uniformly random gather of 3 KiB rows with
// R = 1, 2, 3, 4 rows per 64-lane group in flight, at 4 / 6 / 8 four-wave workgroups per CU.
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_mlp gather_mlp.hip && ./gather_mlp
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while(0)

__device__ __forceinline__ float l2(const uint4 u, const float4 q)
{
    const float a = __uint_as_float(u.x) - q.x, b = __uint_as_float(u.y) - q.y, c = __uint_as_float(u.z) - q.z, d = __uint_as_float(u.w) - q.w;
    return a * a + b * b + c * c + d * d;
}

template <int R, int WGS>
__global__ void __launch_bounds__(256, WGS) k_rows(const uint4 *table, const uint32_t *ids, uint32_t n, const float4 *query, float *out)
{
    const uint32_t lane = threadIdx.x & 63, group = blockIdx.x * 4 + (threadIdx.x >> 6), ngroups = gridDim.x * 4;
    float4 q[ 3 ];
#pragma unroll
    for(int j = 0; j < 3; ++j) q[ j ] = query[ lane + 64 * j ];
    for(uint32_t i = group; i < n; i += R * ngroups) {
        uint4    y[ R ][ 3 ];
        uint32_t at[ R ];
#pragma unroll
        for(int r = 0; r < R; ++r) {
            at[ r ] = i + r * ngroups < n ? i + r * ngroups : i;
            const uint4 *row = table + (size_t)ids[ at[ r ] ] * 192;
#pragma unroll
            for(int c = 0; c < 3; ++c) y[ r ][ c ] = row[ lane + 64 * c ];
        }
#pragma unroll
        for(int r = 0; r < R; ++r) {
            float d = 0.f;
#pragma unroll
            for(int c = 0; c < 3; ++c) d += l2(y[ r ][ c ], q[ c ]);
            for(int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off);
            if(lane == 63) out[ at[ r ] ] = d;
        }
    }
}

static uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }

template <int R, int WGS>
static void run(const uint4 *table, uint32_t *ids, std::vector<uint32_t> &h, uint32_t rows, uint32_t evals, const float4 *query, float *out, int cus, bool &first)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float sum = 0.f, best = 1e30f;
    const int reps = 4;
    for(int r = 0; r < reps + 1; ++r) {
        for(uint32_t i = 0; i < evals; ++i) h[ i ] = mix(i * 2654435761u + 977u * (uint32_t)(r + 16 * R + 256 * WGS)) % rows;
        CHECK(hipMemcpy(ids, h.data(), (size_t)evals * 4, hipMemcpyHostToDevice));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_rows<R, WGS>), dim3(cus * WGS), dim3(256), 0, 0, table, ids, evals, query, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if(r == 0) continue;
        sum += ms;
        best = ms < best ? ms : best;
    }
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_rows<R, WGS>, 256, 0));
    const double bytes = (double)evals * 3072.0;
    std::printf("%s  {\"rows_in_flight_per_group\": %d, \"workgroups_per_cu_launched\": %d, \"resident_per_cu_possible\": %d, \"bytes_in_flight_per_cu\": %d, \"mean_ms\": %.4f, "
                "\"algorithmic_TBps_mean\": %.3f, \"algorithmic_TBps_best\": %.3f}",
                first ? "" : ",\n", R, WGS, occ, R * 3 * 16 * 64 * 4 * (WGS < occ ? WGS : occ), sum / reps, bytes / (sum / reps * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
    first = false;
}

int main(int argc, char **argv)
{
    // rows: 1000000 (3 GB: DRAM) by default; 60000 (184 MB: inside the 256 MiB Infinity Cache); 8000 (24.6 MB: inside the eight 4 MiB L2s)
    const uint32_t rows = argc > 1 ? (uint32_t)std::atoi(argv[ 1 ]) : 1000000, evals = 17600000;
    uint4 *table;
    CHECK(hipMalloc(&table, (size_t)rows * 3072));
    CHECK(hipMemset(table, 0x3c, (size_t)rows * 3072));
    std::vector<uint32_t> h(evals);
    uint32_t *ids;
    float    *out;
    float4   *query;
    CHECK(hipMalloc(&ids, (size_t)evals * 4));
    CHECK(hipMalloc(&out, (size_t)evals * 4));
    CHECK(hipMalloc(&query, 3072));
    CHECK(hipMemset(query, 0, 3072));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"rows\": %u, \"evaluations_per_launch\": %u, \"cases\": [\n", prop.gcnArchName, cus, rows, evals);
    bool first = true;
    run<1, 6>(table, ids, h, rows, evals, query, out, cus, first);
    run<2, 4>(table, ids, h, rows, evals, query, out, cus, first);
    run<2, 6>(table, ids, h, rows, evals, query, out, cus, first);
    run<2, 8>(table, ids, h, rows, evals, query, out, cus, first);
    run<3, 4>(table, ids, h, rows, evals, query, out, cus, first);
    run<3, 6>(table, ids, h, rows, evals, query, out, cus, first);
    run<3, 8>(table, ids, h, rows, evals, query, out, cus, first);
    run<4, 4>(table, ids, h, rows, evals, query, out, cus, first);
    run<4, 6>(table, ids, h, rows, evals, query, out, cus, first);
    run<4, 8>(table, ids, h, rows, evals, query, out, cus, first);
    run<2, 6>(table, ids, h, rows, evals, query, out, cus, first);
    std::printf("\n]}\n");
    return 0;
}
