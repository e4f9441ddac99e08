// host_util.hpp -- small host-side helpers of the device library (no HIP in here).
#pragma once
#include <algorithm>
#include <vector>
#include <cmath>
#include <cstddef>
#include <cstdint>

namespace lgpu {

// one turn of a spin-wait loop, on whatever the host is
inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
    __asm__ __volatile__("yield" ::: "memory");
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}

// Level draw: floor(-ln(U) / ln(M)) -- usearch choose_random_level_, Lantern's copy at
// lantern_hnsw/src/hnsw/insert.c:32-46.  U is a stateless hash of (seed, slot) so that every
// builder (this library on any batch plan, the test oracle) draws the same level for a slot.
inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

inline int level_for(uint64_t seed, uint64_t slot, uint32_t connectivity)
{
    const uint64_t h = splitmix64(seed ^ splitmix64(slot + 0x632BE59BD9B4E019ull));
    const double   u = ((double)(h >> 11) + 1.0) * (1.0 / 9007199254740992.0);  // (0,1]
    double         level = -std::log(u) * (1.0 / std::log((double)connectivity));
    if(level > 255.0) level = 255.0;
    return (int)level;
}

// Which prefix of the pending vectors forms the next device batch: at most max_batch, at most
// size/min_ratio (early inserts stay near-sequential), and a vector that raises the top level is
// inserted alone so the entry point never changes inside a batch.
inline size_t plan_batch(size_t current_size, int max_level, const int *pending_levels, size_t pending, size_t max_batch,
                         size_t min_ratio)
{
    if(pending == 0) return 0;
    if(current_size == 0) return 1;
    size_t b = current_size / (min_ratio ? min_ratio : 1);
    if(b < 1) b = 1;
    if(b > max_batch) b = max_batch;
    if(b > pending) b = pending;
    for(size_t i = 0; i < b; ++i)
        if(pending_levels[ i ] > max_level) return i == 0 ? 1 : i;
    return b;
}

// The batches of the row-sharded build (index.cpp add_row_sharded_locked) and where their rows come from: the usual plan over the
// global level draw; position p of the global order goes to the shard that is furthest behind its proportional share
// n_r (p + 1) / N of the rows handed out so far (ties to the lower rank), so every prefix of the order holds every shard's rows
// in proportion, to within one row, whatever the batch sizes are.  first / count: the batches; share[batch * world + r]: how
// many of a batch's rows come from shard r (within a batch rank 0's rows take the first slots).
inline void row_shard_plan(const uint64_t *sizes, int world, const int *levels, size_t N, size_t max_batch, size_t min_ratio, std::vector<size_t> &first,
                           std::vector<size_t> &count, std::vector<size_t> &share)
{
    std::vector<size_t> taken((size_t)world, 0);
    int    max_level = 0;
    size_t pi = 0;
    while(pi < N) {
        const size_t b = plan_batch(pi, max_level, levels + pi, std::min(N - pi, max_batch), max_batch, min_ratio);
        if(pi == 0 || (b == 1 && levels[ pi ] > max_level)) max_level = levels[ pi ];
        const size_t at = share.size();
        share.resize(at + (size_t)world, 0);
        for(size_t j = 0; j < b; ++j) {
            int      best = 0;
            __int128 lead = 0;
            for(int r = 0; r < world; ++r) {
                const __int128 behind = (__int128)sizes[ r ] * (__int128)(pi + j + 1) - (__int128)taken[ (size_t)r ] * (__int128)N;
                if(r == 0 || behind > lead) { lead = behind; best = r; }
            }
            taken[ (size_t)best ] += 1;
            share[ at + (size_t)best ] += 1;
        }
        first.push_back(pi);
        count.push_back(b);
        pi += b;
    }
}

}  // namespace lgpu
