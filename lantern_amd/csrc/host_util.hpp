// host_util.hpp -- small host-side helpers of the device library (no HIP in here).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

namespace lgpu {

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

}  // namespace lgpu
