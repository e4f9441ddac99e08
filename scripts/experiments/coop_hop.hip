// coop_hop.hip -- the KILL-CRITERION measurement of a cooperative multi-CU hop.
//
// Question: can the rows of ONE hop of a lone 768-d walk (32 neighbour rows x 3 KiB = 96 KiB, today pulled by one CU at its
// ~11 B/cycle fetch rate: ~3.0 us per hop in walk_spec.hpp) be gathered by a leader workgroup + H helper workgroups on other CUs
// fast enough to bring the hop under 2.2 us -- INCLUDING the two dependent hand-offs per hop (leader -> helpers: "here are the
// ids"; helpers -> leader: "here are the distances") that a dependent chain of hops cannot hide?
//
// The microbenchmark is the cooperative hop with everything else of the walk taken away (no candidate list, no visited set): per hop
//   leader : publishes the hop's 32 ids + a sequence word (8-byte agent-scope atomics: no fence, MI355X_MICROARCH.md "8-B agent
//            atomics both sides"), evaluates its own share of the rows, waits for every helper's completion word, reads the
//            32 distances back, derives the NEXT hop's ids from them (the dependence of a real walk), repeats;
//   helper : polls the sequence word, reads the ids, gathers its share of the rows (plain loads; the rows are read-only),
//            reduces to one distance per row, stores (distance bits) with agent-scope atomics, bumps its completion word.
// H = 0 is the one-CU hop of today (all 32 rows by the leader, no hand-off).  Placement: participants are the workgroups with
// blockIdx % stride == 0 -- stride 8 puts them on ONE XCD (block b runs on XCD b % 8), stride 1 spreads them over XCDs.
//
//   hipcc --offload-arch=gfx950 -O3 -o coop_hop coop_hop.hip && ./coop_hop
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while(0)

constexpr int ROWS = 32, CHUNKS = 192 /* 768 f32 */, WAVES = 8, T = WAVES * 64;

struct Mail
{
    unsigned long long seq;            // hop number published by the leader (1-based)
    unsigned long long ids[ ROWS / 2 ];  // 32 ids, two per 8-byte word
    unsigned long long dist[ ROWS ];   // (hop << 32) | distance bits, one per row
    unsigned long long done[ 16 ];     // per helper: last hop finished
    unsigned long long xcc[ 16 ];      // diagnostics: XCC id of each participant
};

__device__ __forceinline__ unsigned long long ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one row per wave at a time: 192 uint4 = 3 per lane; l2sq against the query held in registers
__device__ __forceinline__ float row_dist(const uint4 *row, const float4 q[ 3 ], int lane)
{
    float acc = 0.f;
#pragma unroll
    for(int j = 0; j < 3; ++j) {
        const uint4  u = row[ lane + 64 * j ];
        const float4 v = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
        const float  a = v.x - q[ j ].x, b = v.y - q[ j ].y, c = v.z - q[ j ].z, d = v.w - q[ j ].w;
        acc += a * a + b * b + c * c + d * d;
    }
    for(int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    return acc;
}

__global__ void __launch_bounds__(T) k_coop(const uint4 *table, uint32_t n, const float4 *query, Mail *m, int hops, int helpers, int stride, float *sink)
{
    if(blockIdx.x % stride != 0) return;
    const int who = blockIdx.x / stride;  // 0 = leader, 1..helpers
    if(who > helpers) return;
    const int tid = threadIdx.x, wave = tid / 64, lane = tid % 64, parts = helpers + 1;
    __shared__ uint32_t s_ids[ ROWS ];
    __shared__ float    s_d[ ROWS ];
    float4 q[ 3 ];
    for(int j = 0; j < 3; ++j) q[ j ] = query[ lane + 64 * j ];
    if(tid == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        st(&m->xcc[ who ], (unsigned long long)(x & 0xF));
    }
    // rows [lo, hi) of every hop are this participant's
    const int lo = ROWS * who / parts, hi = ROWS * (who + 1) / parts;
    uint32_t  state = 12345u;
    float     total = 0.f;
    for(int h = 1; h <= hops; ++h) {
        if(who == 0) {
            // the hop's ids depend on the previous hop's distances (state)
            if(tid < ROWS) {
                uint32_t x = state * 2654435761u + (uint32_t)tid * 40503u + (uint32_t)h * 97u;
                x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
                s_ids[ tid ] = x % n;
            }
            __syncthreads();
            if(helpers) {
                if(tid < ROWS / 2) st(&m->ids[ tid ], (unsigned long long)s_ids[ 2 * tid ] | ((unsigned long long)s_ids[ 2 * tid + 1 ] << 32));
                __syncthreads();
                if(tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st(&m->seq, (unsigned long long)h); }
            }
        } else {
            if(tid == 0) while(ld(&m->seq) < (unsigned long long)h) {}
            __syncthreads();
            if(tid < ROWS / 2) {
                const unsigned long long w = ld(&m->ids[ tid ]);
                s_ids[ 2 * tid ] = (uint32_t)w;
                s_ids[ 2 * tid + 1 ] = (uint32_t)(w >> 32);
            }
            __syncthreads();
        }
        for(int r = lo + wave; r < hi; r += WAVES) {
            const float d = row_dist(table + (size_t)s_ids[ r ] * CHUNKS, q, lane);
            if(lane == 0) {
                if(who == 0) s_d[ r ] = d;
                else st(&m->dist[ r ], ((unsigned long long)h << 32) | (unsigned long long)__float_as_uint(d));
            }
        }
        if(who != 0) {
            __syncthreads();
            if(tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st(&m->done[ who ], (unsigned long long)h); }
        } else {
            // wait for the helpers' rows, read them back
            if(tid >= hi && tid < ROWS) {
                unsigned long long w;
                do { w = ld(&m->dist[ tid ]); } while((w >> 32) != (unsigned long long)h);
                s_d[ tid ] = __uint_as_float((uint32_t)w);
            }
            __syncthreads();
            if(tid == 0) {
                float    best = s_d[ 0 ];
                uint32_t arg = 0;
                for(int i = 1; i < ROWS; ++i) if(s_d[ i ] < best) { best = s_d[ i ]; arg = (uint32_t)i; }
                s_ids[ 0 ] = s_ids[ arg ] ^ __float_as_uint(best);
                total += best;
            }
            __syncthreads();
            state = s_ids[ 0 ];
            __syncthreads();
        }
    }
    if(who == 0 && tid == 0) *sink = total;
}

int main(int argc, char **argv)
{
    const uint32_t n = 1000000;
    const int      hops = argc > 1 ? std::atoi(argv[ 1 ]) : 2000;
    uint4 *table;
    CHECK(hipMalloc(&table, (size_t)n * CHUNKS * 16));
    {
        std::vector<float> h((size_t)1 << 22);
        for(size_t i = 0; i < h.size(); ++i) h[ i ] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.f;
        for(size_t off = 0; off < (size_t)n * CHUNKS * 16; off += h.size() * 4)
            CHECK(hipMemcpy((char *)table + off, h.data(), std::min(h.size() * 4, (size_t)n * CHUNKS * 16 - off), hipMemcpyHostToDevice));
    }
    float4 *query;
    CHECK(hipMalloc(&query, CHUNKS * 16));
    CHECK(hipMemset(query, 0, CHUNKS * 16));
    Mail  *mail;
    float *sink;
    CHECK(hipMalloc(&mail, sizeof(Mail)));
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    std::printf("{\"hops\": %d, \"rows_per_hop\": %d, \"row_bytes\": %d, \"results\": [", hops, ROWS, CHUNKS * 16);
    bool first = true;
    for(int stride : { 8, 1 })
        for(int helpers : { 0, 1, 3, 7 }) {
            if(helpers == 0 && stride == 1) continue;
            float best = 1e30f;
            Mail  hm;
            for(int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemset(mail, 0, sizeof(Mail)));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(a));
                hipLaunchKernelGGL(k_coop, dim3((helpers + 1) * stride), dim3(T), 0, 0, table, n, query, mail, hops, helpers, stride, sink);
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms;
                CHECK(hipEventElapsedTime(&ms, a, b));
                best = std::min(best, ms);
            }
            CHECK(hipMemcpy(&hm, mail, sizeof(Mail), hipMemcpyDeviceToHost));
            std::printf("%s{\"participants\": %d, \"placement\": \"%s\", \"us_per_hop\": %.3f, \"xcc\": [", first ? "" : ", ", helpers + 1,
                        stride == 8 ? "one XCD (blockIdx % 8 == 0)" : "consecutive blocks (one per XCD)", best * 1000.f / hops);
            for(int i = 0; i <= helpers; ++i) std::printf("%s%llu", i ? ", " : "", hm.xcc[ i ]);
            std::printf("]}");
            first = false;
        }
    std::printf("]}\n");
    return 0;
}
