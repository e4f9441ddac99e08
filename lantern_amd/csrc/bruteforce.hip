// bruteforce.hip -- the dense batched-query x candidate contraction (BASELINE config[2]) and the
// exact k-NN built on it (the seq-scan `ORDER BY v <op> q LIMIT k`; ground truth for recall@k as
// defined by lantern_cli/src/index_autotune/mod.rs:196-203,239-247).
//
//   k_row_norms    ||x||^2 per row (one G-lane group per row, same reduction as the walk)
//   k_dense_f32    D[q][c] = metric(Q[q], B[c]) for a 128 x 128 tile per workgroup on fp32 MFMA
//                  (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate -- bf16/fp16 would lose the 1e-5
//                  tolerance).  l2sq = |q|^2 + |b|^2 - 2 q.b ; cos = 1 - q.b / (|q||b|) with the
//                  reference's zero-norm rules.  This is the only place MFMA is used: it is the one
//                  true dense contraction on the path.
//   k_dense_ham    the same tile shape for hamming (popcount of XOR; integer, VALU)
//   k_select       per query: merge a chunk of the distance matrix into a running top-k'
//   k_rerank       recompute the k' survivors in the graph walk's exact reduction order and emit
//                  the final top-k by (distance, slot)
#include "kernels.hpp"
#include "walk.hpp"

namespace lgpu {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------
template <int G>
__global__ void __launch_bounds__(256) k_row_norms(const uint4 *rows, uint32_t n, uint32_t chunks, float *out)
{
    const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) / G, gl = threadIdx.x % G;
    const uint32_t ngroups = gridDim.x * blockDim.x / G;
    for(uint32_t i = gid; i < n; i += ngroups) {
        const uint4 *r = rows + (size_t)i * chunks;
        float        s = 0.f;
        for(uint32_t ch = gl; ch < chunks; ch += G) {
            uint4 x = r[ ch ];
            s = __builtin_fmaf(__uint_as_float(x.x), __uint_as_float(x.x), s);
            s = __builtin_fmaf(__uint_as_float(x.y), __uint_as_float(x.y), s);
            s = __builtin_fmaf(__uint_as_float(x.z), __uint_as_float(x.z), s);
            s = __builtin_fmaf(__uint_as_float(x.w), __uint_as_float(x.w), s);
        }
        s = group_sum<G>(s);
        if(gl == G - 1) out[ i ] = s;
    }
}

// ---------------------------------------------------------------------------------------------------
// 128 x 128 output tile per 256-thread workgroup; 4 waves as 2 x 2, each wave 2 x 2 MFMA tiles of 32 x 32 (64 accumulator
// registers per lane); K staged through LDS 32 floats at a time.
constexpr int BM = 128, BN = 128, BK = 32;

// FUSED: the tile does not write its distances; an output that can still enter its query's running top-kk -- ordered distance
// <= the kk-th best so far, read once per tile -- is appended to that query's candidate list (cand[q][CAP], cnt[q]), which
// k_select folds into `best` after the launch.  Once a few thousand columns have been seen that is a handful of appends per
// query and launch instead of a 268 MB matrix written here and read back there.
struct DenseTopk
{
    const uint64_t *best;  // [nq][kk] running top-kk keys (ordered distance << 32 | column), ascending
    uint32_t        kk;
    uint64_t       *cand;  // [nq][cap]
    uint32_t       *cnt;   // [nq] appended so far (may exceed cap: k_select reports the overflow)
    uint32_t        cap;
    uint32_t        c_base;  // column id of this launch's first base row
};

// Persistent workgroups over a stream of K steps.  A launch starts DENSE_WGS_PER_CU workgroups per CU; each walks its own list
// of tiles, and the K steps of all its tiles form ONE software pipeline over two LDS buffers: while step s (buffer P) is on the
// matrix cores, step s + 1 is landing in the other buffer and, from the step's barrier on, step s + 2 is being loaded into
// buffer P itself -- across tile boundaries too, so only a workgroup's first two steps ever wait for HBM with nothing to do.
// The loads are `buffer_load_dwordx4 ... lds`: global memory to LDS without a register in between and without a ds_write (gfx950
// moves 16 bytes per lane that way).  A step is eight half-rounds of 8 MFMAs, and everything else it has to do is dealt out
// BETWEEN them, in the shadow of the MFMA just issued:
//   even half-rounds   read the next round's fragments (one 16-byte read per lane and row block) into the second register set
//   half-round 4       wait for this wave's loads of step s + 1, then the step's one barrier: the other buffer is complete for
//                      everyone, and this one has no reader left (its last fragment reads were issued at the start of the
//                      half-round)
//   half-rounds 5, 6   issue the loads of step s + 2 into this buffer; read round 0's fragments of step s + 1 from the other
// What this replaced (r1-r2: loads into registers, 64 MFMAs, stash, two barriers per step) left every wave standing between a
// barrier and its first MFMA once per step, and the two waves of a SIMD fell into step with each other instead of covering for
// one another: 0.79 of the fp32-matrix peak where the same loop without its loads ran at 0.89 and with nothing but MFMAs at
// 0.94; the same pipeline with the tiles staged through registers (ds_write_b128 into padded rows) reached 0.87
// (profiles/r03_dense_variants.md has every step of that).
// Tile order: workgroup b runs on XCD b % 8; every XCD gets a contiguous range of the tn-major tile order and its workgroups
// take consecutive tiles of it, so the tiles_m workgroups sharing a B tile run on ONE XCD at about the same time and its L2
// serves all but the first read.  Loads are branch-free: the descriptor of a tile starts at its first row and ends with its
// last one that exists, so rows past the end land as zeros, and so does a k past the end (offset pushed out of range).
constexpr int DENSE_WGS_PER_CU = 2;  // 67 KB of LDS each

template <int METRIC, bool FUSED = false>
__global__ void __launch_bounds__(256, 2) k_dense_f32(const float *Q, uint32_t nq, const float *B, uint32_t nb, uint32_t stride /* floats per row */,
                                                   const float *qn, const float *bn, float *out, uint32_t ldo, DenseTopk tk)
{
    // two buffers of a 128 x 32 slab per matrix, rows unpadded (128 B), filled by buffer loads that write LDS directly: a load
    // instruction of a wave lands as 1 KB of consecutive 16-byte units (lane i -> unit i), i.e. 8 rows; WHICH k-quad of its row
    // a lane fetches is the lane's choice, and it picks quad (i & 7) ^ ((row >> 1) & 7) -- an XOR swizzle that makes the
    // 16-byte fragment reads of 16 consecutive rows hit 16 different bank groups.
    __shared__ float A_0[ BM * BK ], A_1[ BM * BK ], B_0[ BN * BK ], B_1[ BN * BK ];
    __shared__ float Ns[ 2 ][ 3 * 128 ];  // per tile parity: 128 query norms, 128 base norms (cosine: inverse roots), 128 query radii
    const int        tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int        wm = wave >> 1, wn = wave & 1;
    const uint32_t   tiles_m = (nq + BM - 1) / BM, T = tiles_m * ((nb + BN - 1) / BN);
    const uint32_t   x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const uint32_t   wx = (gridDim.x - x + 7) >> 3;  // workgroups on this XCD
    const uint32_t   cnt = (T >> 3) + (x < (T & 7) ? 1u : 0u), start = x * (T >> 3) + (x < (T & 7) ? x : (T & 7));  // its tiles
    const uint32_t   my_n = cnt > j ? (cnt - j + wx - 1) / wx : 0u;
    if(my_n == 0) return;
    auto tile_origin = [&](uint32_t i, uint32_t &q0, uint32_t &c0) {
        const uint32_t lin = start + j + i * wx;
        q0 = (lin % tiles_m) * BM;
        c0 = (lin / tiles_m) * BN;
    };

    floatx16 acc[ 2 ][ 2 ];
    auto     zero_acc = [&]() {
#pragma unroll
        for(int i = 0; i < 2; ++i)
#pragma unroll
            for(int jj = 0; jj < 2; ++jj)
#pragma unroll
                for(int r = 0; r < 16; ++r) acc[ i ][ jj ][ r ] = 0.f;
    };
    zero_acc();

    // ---- the load side of the pipeline: (tile li, k position lk), up to two steps ahead of the MFMAs
    const int      wu = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t kq = (((uint32_t)lane & 7u) ^ (((uint32_t)wu * 4u + ((uint32_t)lane >> 4)) & 7u)) * 4u;  // the k-quad this lane fetches
    const uint32_t voff = ((uint32_t)(wu * 8 + (lane >> 3)) * stride + kq) * 4u, vstep = 32u * stride * 4u;
    uint32_t               li = 0, lk = 0;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 rq, rb;  // buffer descriptors as words: base, base_hi (stride 0), bytes, flags
    auto  make_desc = [](const float *base, uint32_t bytes) {
        const uint64_t a = (uint64_t)(uintptr_t)base;
        u32x4          d;
        d.x = (uint32_t)a;
        d.y = (uint32_t)(a >> 32) & 0xFFFFu;
        d.z = bytes;
        d.w = 0x00020000u;
        return d;
    };
    auto set_load_tile = [&](uint32_t i) {
        if(i < my_n) {
            uint32_t q0, c0;
            tile_origin(i, q0, c0);
            const uint32_t rows_q = nq - q0 < (uint32_t)BM ? nq - q0 : (uint32_t)BM, rows_b = nb - c0 < (uint32_t)BN ? nb - c0 : (uint32_t)BN;
            rq = make_desc(Q + (size_t)q0 * stride, rows_q * stride * 4u);
            rb = make_desc(B + (size_t)c0 * stride, rows_b * stride * 4u);
        } else {  // past this workgroup's last tile: empty descriptors, every load returns zeros
            rq = make_desc(Q, 0);
            rb = make_desc(B, 0);
        }
    };
    // rows 8 (4 it + wave) .. + 8 of step (li, lk), 16 bytes per lane, straight into LDS.  Inline assembly, because the
    // compiler's own bookkeeping of such loads makes EVERY later LDS read wait for all of them (it cannot tell the two
    // buffers apart once there are more than a few load instructions); the wait that is needed -- all of a step's loads, before
    // the barrier that precedes their first read -- is dma_wait() below.  (M0, the LDS base of such a load, is written and used
    // inside one statement; nothing else in this file's kernels touches it.)
    auto lds_addr = [](const float *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float *)p; };
    auto dma1 = [&](int it, const float *Ap, const float *Bp) {
        const uint32_t off = lk + kq < stride ? voff + (uint32_t)it * vstep + lk * 4u : 0x80000000u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr(Ap + (it * 4 + wu) * 256)), "v"(off), "s"(rq) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr(Bp + (it * 4 + wu) * 256)), "v"(off), "s"(rb) : "memory");
    };
    auto dma_wait = []() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); };
    auto advance = [&]() {
        lk += BK;
        if(lk >= stride) {
            lk = 0;
            set_load_tile(++li);
        }
    };
    // ---- a tile's norms and radii: loaded a tile ahead into two registers, written to LDS in the tile's first step
    float    nreg = 0.f;
    uint32_t rreg = 0;
    auto     load_norms = [&](uint32_t i) {
        uint32_t q0, c0;
        tile_origin(i, q0, c0);
        nreg = tid < BM ? (q0 + tid < nq ? qn[ q0 + tid ] : 0.f) : (c0 + (tid - BM) < nb ? bn[ c0 + (tid - BM) ] : 0.f);
        if(FUSED && tid < BM)  // the query's current radius: the distance of its kk-th best so far (all ones while the list is short)
            rreg = q0 + tid < nq ? (uint32_t)(tk.best[ (size_t)(q0 + tid) * tk.kk + tk.kk - 1 ] >> 32) : 0u;
    };
    auto store_norms = [&](uint32_t par) {
        // Cosine: the norms become 1 / sqrt(norm) here (0 stays 0: the zero-norm rules of the epilogue test for it), so that an
        // output costs two multiplies instead of two IEEE square roots and a divide.  (These distances pick candidates / meet the
        // 1e-5 tolerance; exact results are re-ranked in the walk's own order.)
        float nv = nreg;
        if(METRIC != M_L2SQ) nv = nv == 0.f ? 0.f : 1.f / __builtin_sqrtf(nv);
        Ns[ par ][ tid ] = nv;
        if(FUSED && tid < BM) Ns[ par ][ 256 + tid ] = rreg == 0xFFFFFFFFu ? __builtin_inff() : ord2f(rreg);  // (rows past nq: ord2f(0) = NaN, nothing passes)
    };
    // ---- fragments: one 16-byte read per lane, row block and round j: half h = lane / 32 reads k-quad 2 j + h of its row, and
    // MFMA e (0..3) of the round contracts the pair {8 j + e, 8 j + 4 + e} -- a permutation of the contraction index, which a
    // dot product does not notice
    uint32_t foff[ 4 ];
#pragma unroll
    for(int jq = 0; jq < 4; ++jq) foff[ jq ] = (((uint32_t)(2 * jq) + ((uint32_t)lane >> 5)) ^ (((uint32_t)lane >> 1) & 7u)) * 4u;
    const uint32_t rowa = (uint32_t)(wm * 64 + (lane & 31)) * BK, rowb = (uint32_t)(wn * 64 + (lane & 31)) * BK;
    float4         fa[ 2 ][ 2 ], fb[ 2 ][ 2 ];  // [fragment set][row block]
    auto           frag = [&](int set, const float *Ap, const float *Bp, int jq) {
        fa[ set ][ 0 ] = *(const float4 *)(Ap + rowa + foff[ jq ]);
        fa[ set ][ 1 ] = *(const float4 *)(Ap + rowa + 32 * BK + foff[ jq ]);
        fb[ set ][ 0 ] = *(const float4 *)(Bp + rowb + foff[ jq ]);
        fb[ set ][ 1 ] = *(const float4 *)(Bp + rowb + 32 * BK + foff[ jq ]);
    };
    auto mfma4 = [&](float a0, float a1, float b0, float b1) {
        acc[ 0 ][ 0 ] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[ 0 ][ 0 ], 0, 0, 0);
        acc[ 0 ][ 1 ] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[ 0 ][ 1 ], 0, 0, 0);
        acc[ 1 ][ 0 ] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[ 1 ][ 0 ], 0, 0, 0);
        acc[ 1 ][ 1 ] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[ 1 ][ 1 ], 0, 0, 0);
    };
    auto mfma8 = [&](int set, int h) {  // h = 0: e = 0, 1;  h = 1: e = 2, 3
        if(h == 0) {
            mfma4(fa[ set ][ 0 ].x, fa[ set ][ 1 ].x, fb[ set ][ 0 ].x, fb[ set ][ 1 ].x);
            mfma4(fa[ set ][ 0 ].y, fa[ set ][ 1 ].y, fb[ set ][ 0 ].y, fb[ set ][ 1 ].y);
        } else {
            mfma4(fa[ set ][ 0 ].z, fa[ set ][ 1 ].z, fb[ set ][ 0 ].z, fb[ set ][ 1 ].z);
            mfma4(fa[ set ][ 0 ].w, fa[ set ][ 1 ].w, fb[ set ][ 0 ].w, fb[ set ][ 1 ].w);
        }
    };
    // plain-output variant: the finished tile's distances on their way out (see the epilogue), its origin, and how many of its
    // eight parts (row block x column block x half of the 16 registers) have left
    floatx16 stash[ 2 ][ 2 ];
    uint32_t st_q0 = 0, st_c0 = 0;
    int      st_done = 8;
    auto     flush_part = [&](auto part_c) {
        // only INTERIOR tiles take this way out (every row below nq, every column below nb): plain stores, no bounds, no branch
        constexpr int part = decltype(part_c)::value, i = part >> 2, jj = (part >> 1) & 1, h = part & 1;
        const uint32_t c = st_c0 + (uint32_t)(wn * 64 + jj * 32 + (lane & 31));
#pragma unroll
        for(int r = 8 * h; r < 8 * h + 8; ++r) {
            const uint32_t q = st_q0 + (uint32_t)(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
            out[ (size_t)q * ldo + c ] = stash[ i ][ jj ][ r ];
        }
    };
    // (st_done only ever holds uniform values; it goes through readfirstlane so that the compiler KNOWS, and keeps the branches -- and
    // everything live across them, the buffer descriptors of the load side above all -- scalar)
    auto flush_range = [&](int from, int to) {  // parts [from, to) that have not left yet
        const int d = __builtin_amdgcn_readfirstlane(st_done);
        if(d <= 0 && 0 >= from && 0 < to) flush_part(std::integral_constant<int, 0>{});
        if(d <= 1 && 1 >= from && 1 < to) flush_part(std::integral_constant<int, 1>{});
        if(d <= 2 && 2 >= from && 2 < to) flush_part(std::integral_constant<int, 2>{});
        if(d <= 3 && 3 >= from && 3 < to) flush_part(std::integral_constant<int, 3>{});
        if(d <= 4 && 4 >= from && 4 < to) flush_part(std::integral_constant<int, 4>{});
        if(d <= 5 && 5 >= from && 5 < to) flush_part(std::integral_constant<int, 5>{});
        if(d <= 6 && 6 >= from && 6 < to) flush_part(std::integral_constant<int, 6>{});
        if(d <= 7 && 7 >= from && 7 < to) flush_part(std::integral_constant<int, 7>{});
    };
    auto flush_next = [&]() {
        const int d = __builtin_amdgcn_readfirstlane(st_done);
        if(d >= 8) return;
        flush_range(d, d + 1);
        st_done = d + 1;
    };
    auto flush_rest = [&]() {
        flush_range(0, 8);
        st_done = 8;
    };
    auto epilogue = [&](uint32_t ti) {
        // ---- the tile's epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
        // Its norms and radii are in Ns[ti & 1] since the barrier of its first step; the next tile's first step is already in
        // the other LDS buffer and its fragments in registers, so the MFMAs resume right after.
        uint32_t q0, c0;
        tile_origin(ti, q0, c0);
        if constexpr(!FUSED) {
            flush_rest();  // (a tile of fewer than eight K steps: the previous one's parts have not all left yet)
            st_q0 = q0;
            st_c0 = c0;
        }
        const float *Nq = Ns[ ti & 1 ], *Nb = Ns[ ti & 1 ] + 128, *Nr = Ns[ ti & 1 ] + 256;
        const bool   interior = q0 + (uint32_t)BM <= nq && c0 + (uint32_t)BN <= nb;  // (uniform)
        if constexpr(!FUSED) st_done = interior ? 0 : 8;
#pragma unroll
        for(int i = 0; i < 2; ++i)
#pragma unroll
            for(int jj = 0; jj < 2; ++jj) {
                const int      cl = wn * 64 + jj * 32 + (lane & 31);
                const uint32_t c = c0 + (uint32_t)cl;
                const float    nb2 = Nb[ cl ];
                auto           dist_of = [&](int r, float dot) {
                    const float nq2 = Nq[ wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) ];
                    float       d;
                    if(METRIC == M_L2SQ) {
                        d = nq2 + nb2 - 2.f * dot;
                        d = d < 0.f ? 0.f : d;
                    } else {
                        if(nq2 == 0.f && nb2 == 0.f) d = 0.f;  // (nq2 / nb2 hold the INVERSE roots here)
                        else if(nq2 == 0.f || nb2 == 0.f) d = 1.f;
                        else d = 1.f - dot * (nq2 * nb2);
                    }
                    return d;
                };
                if constexpr(FUSED) {
                    // straight-line pass over the 16 outputs: which of them are inside their query's radius?  (A float compare
                    // orders like the keys' f2ord; NaN never passes; rows past nq carry a NaN radius.)  Then the rare appends.
                    uint32_t pass = 0;
#pragma unroll
                    for(int r = 0; r < 16; ++r) {
                        const int ql = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        pass |= (dist_of(r, acc[ i ][ jj ][ r ]) <= Nr[ ql ] ? 1u : 0u) << r;
                    }
                    if(c >= nb) pass = 0;
                    while(pass) {
                        const int r = __builtin_ctz(pass);
                        pass &= pass - 1;
                        float dot = 0.f;
#pragma unroll
                        for(int rr = 0; rr < 16; ++rr)
                            if(rr == r) dot = acc[ i ][ jj ][ rr ];
                        const float    d = dist_of(r, dot);
                        const uint32_t q = q0 + (uint32_t)(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
                        const uint32_t p = atomicAdd(&tk.cnt[ q ], 1u);
                        if(p < tk.cap) tk.cand[ (size_t)q * tk.cap + p ] = ((uint64_t)f2ord(d) << 32) | (uint64_t)(tk.c_base + c);
                    }
                } else {
                    // The plain-output variant does not store an interior tile here: 64 stores per lane issued at once sit in the SAME
                    // in-order counter as the LDS-direct loads of the next tile's first steps, whose `s_waitcnt vmcnt(0)` then waits for
                    // the writes to reach memory -- once per tile the matrix pipe stood still for a write latency (0.80 of the peak where
                    // the fused variant runs at 0.905).  The distances go to a register stash and leave eight at a time, one part per K
                    // step of the NEXT tile, right behind that step's barrier (flush_part above): a part has a whole step to drain.  A
                    // tile on the matrix's edge (rows past nq or columns past nb) stores at once, under its bounds checks, as before.
                    if(interior) {
#pragma unroll
                        for(int r = 0; r < 16; ++r) stash[ i ][ jj ][ r ] = dist_of(r, acc[ i ][ jj ][ r ]);
                    } else {
#pragma unroll
                        for(int r = 0; r < 16; ++r) {
                            const uint32_t q = q0 + (uint32_t)(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
                            const float    d = dist_of(r, acc[ i ][ jj ][ r ]);
                            if(q < nq && c < nb) out[ (size_t)q * ldo + c ] = d;
                        }
                    }
                }
            }
    };
    // ---- fill the pipeline: steps 0 and 1 into the two LDS buffers
    set_load_tile(0);
    load_norms(0);
#pragma unroll
    for(int it = 0; it < 4; ++it) dma1(it, A_0, B_0);
    advance();
#pragma unroll
    for(int it = 0; it < 4; ++it) dma1(it, A_1, B_1);
    advance();
    dma_wait();
    __syncthreads();
    frag(0, A_0, B_0, 0);
    const uint32_t KS = (stride + BK - 1) / BK;
    uint32_t       ti = 0, ks = 0;
    // One step on buffer P.  Its barrier sits after half-round 4: by then every wave has read the last fragments of this buffer
    // (so it can be refilled, with step s + 2, from half-round 5 on) and the loads of step s + 1 into the other buffer --
    // issued a step ago -- are waited for (so its first fragments can be read in half-round 6).
#define LGPU_FENCE __builtin_amdgcn_sched_barrier(0)
#define LGPU_DENSE_STEP(P)                                                          \
    {                                                                               \
        float *const Ap = (P) ? A_1 : A_0, *const Bp = (P) ? B_1 : B_0;             \
        float *const Ao = (P) ? A_0 : A_1, *const Bo = (P) ? B_0 : B_1;             \
        frag(1, Ap, Bp, 1);                                                         \
        LGPU_FENCE;                                                                 \
        mfma8(0, 0);                                                                \
        mfma8(0, 1);                                                                \
        LGPU_FENCE;                                                                 \
        frag(0, Ap, Bp, 2);                                                         \
        LGPU_FENCE;                                                                 \
        mfma8(1, 0);                                                                \
        mfma8(1, 1);                                                                \
        LGPU_FENCE;                                                                 \
        frag(1, Ap, Bp, 3);                                                         \
        LGPU_FENCE;                                                                 \
        mfma8(0, 0);                                                                \
        LGPU_FENCE;                                                                 \
        dma_wait();                                                                 \
        __syncthreads();                                                            \
        if constexpr(!FUSED) flush_next(); /* one part of the previous tile's distances: a whole step to drain */ \
        if(ks == 0) { /* (behind the wait for the loads: the norms' own load is long done) */ \
            store_norms(ti & 1);                                                    \
            if(ti + 1 < my_n) load_norms(ti + 1);                                   \
        }                                                                           \
        mfma8(0, 1);                                                                \
        LGPU_FENCE;                                                                 \
        dma1(0, Ap, Bp);                                                            \
        dma1(1, Ap, Bp);                                                            \
        frag(0, Ao, Bo, 0);                                                         \
        LGPU_FENCE;                                                                 \
        mfma8(1, 0);                                                                \
        LGPU_FENCE;                                                                 \
        dma1(2, Ap, Bp);                                                            \
        dma1(3, Ap, Bp);                                                            \
        advance();                                                                  \
        LGPU_FENCE;                                                                 \
        mfma8(1, 1);                                                                \
        LGPU_FENCE;                                                                 \
        if(++ks == KS) {                                                            \
            if(KS == 1) __syncthreads(); /* the norms were written after this step's only barrier */ \
            epilogue(ti);                                                           \
            zero_acc();                                                             \
            ks = 0;                                                                 \
            if(++ti == my_n) break;                                                 \
        }                                                                           \
    }
    for(;;) {
        LGPU_DENSE_STEP(0)
        LGPU_DENSE_STEP(1)
    }
    if constexpr(!FUSED) flush_rest();  // the workgroup's last tile
#undef LGPU_DENSE_STEP
#undef LGPU_FENCE
}

// hamming (COSB1: the cosine of the {0, 1} vectors, device_common.hpp M_COS_B1): 16 queries in LDS per workgroup, one base row per thread
template <bool COSB1>
__global__ void __launch_bounds__(256) k_dense_ham(const uint32_t *Q, uint32_t nq, const uint32_t *B, uint32_t nb, uint32_t stride /* words */,
                                                   float *out, uint32_t ldo)
{
    extern __shared__ uint32_t qs[];  // [16][stride]
    const uint32_t q0 = blockIdx.y * 16;
    for(uint32_t i = threadIdx.x; i < 16 * stride; i += blockDim.x) {
        const uint32_t q = q0 + i / stride;
        qs[ i ] = q < nq ? Q[ (size_t)q * stride + i % stride ] : 0u;
    }
    __syncthreads();
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if(c >= nb) return;
    uint32_t acc[ 16 ], qpop[ 16 ], bpop = 0;
#pragma unroll
    for(int j = 0; j < 16; ++j) acc[ j ] = qpop[ j ] = 0;
    const uint32_t *row = B + (size_t)c * stride;
    for(uint32_t w = 0; w < stride; ++w) {
        const uint32_t x = row[ w ];
        if(COSB1) bpop += __popc(x);
#pragma unroll
        for(int j = 0; j < 16; ++j) {
            const uint32_t qw = qs[ j * stride + w ];
            acc[ j ] += COSB1 ? __popc(x & qw) : __popc(x ^ qw);
            if(COSB1) qpop[ j ] += __popc(qw);
        }
    }
#pragma unroll
    for(int j = 0; j < 16; ++j)
        if(q0 + j < nq) {
            float d = (float)acc[ j ];
            if(COSB1) {
                const float a2 = (float)qpop[ j ], b2 = (float)bpop;
                d = (a2 == 0.f && b2 == 0.f) ? 0.f : (a2 == 0.f || b2 == 0.f) ? 1.f : 1.f - d / (__builtin_sqrtf(a2) * __builtin_sqrtf(b2));
            }
            out[ (size_t)(q0 + j) * ldo + c ] = d;
        }
}

// ---------------------------------------------------------------------------------------------------
// k_select: one workgroup per query.  best[q][0..kk) holds the running candidates as keys
// (ordered distance << 32 | slot); a chunk of `ncols` distances (columns = slots c_base..) is merged in.
constexpr int SEL_BUF = 2048;

__device__ void bitonic_sort_lds(uint64_t *a, int n /* power of two */)
{
    for(int size = 2; size <= n; size <<= 1)
        for(int stride = size >> 1; stride > 0; stride >>= 1) {
            for(int i = threadIdx.x; i < n / 2; i += blockDim.x) {
                const int lo = (i / stride) * stride * 2 + (i % stride), hi = lo + stride;
                const bool up = ((lo / size) & 1) == 0;
                const uint64_t x = a[ lo ], y = a[ hi ];
                if((x > y) == up) { a[ lo ] = y; a[ hi ] = x; }
            }
            __syncthreads();
        }
}

// With `cand` (the fused contraction's candidate lists) the source is cand[q][0 .. min(cand_cnt[q], cap)) -- ready-made keys --
// and cand_cnt[q] is reset; a list that overflowed its capacity raises *overflow (the caller then repeats the search unfused).
__global__ void __launch_bounds__(256) k_select(const float *dist, uint32_t ldo, uint32_t ncols, uint32_t c_base, uint64_t *best, uint32_t kk,
                                                const uint64_t *cand, uint32_t *cand_cnt, uint32_t cap, uint32_t *overflow)
{
    __shared__ uint64_t buf[ SEL_BUF ];
    __shared__ int      cnt;
    __shared__ uint64_t tau;
    const uint32_t q = blockIdx.x;
    const float   *row = dist + (size_t)q * ldo;
    const uint64_t *keys = nullptr;
    if(cand) {
        keys = cand + (size_t)q * cap;
        const uint32_t have = cand_cnt[ q ];
        if(have > cap && threadIdx.x == 0) atomicOr(overflow, 1u);
        ncols = have < cap ? have : cap;
        if(ncols == 0) return;  // nothing to fold in (cand_cnt[q] is already 0)
    }
    uint64_t      *mine = best + (size_t)q * kk;
    for(int i = threadIdx.x; i < SEL_BUF; i += blockDim.x) buf[ i ] = i < (int)kk ? mine[ i ] : ~0ull;
    if(threadIdx.x == 0) { cnt = (int)kk; tau = mine[ kk - 1 ]; }
    __syncthreads();
    for(uint32_t base = 0; base < ncols; base += blockDim.x * 4) {
        // 4 columns per thread per round; anything below the current k-th best goes to the buffer
        for(int u = 0; u < 4; ++u) {
            const uint32_t c = base + u * blockDim.x + threadIdx.x;
            if(c < ncols) {
                const uint64_t key = keys ? keys[ c ] : ((uint64_t)f2ord(row[ c ]) << 32) | (uint64_t)(c_base + c);
                if(key < tau) {
                    const int p = atomicAdd(&cnt, 1);
                    if(p < SEL_BUF) buf[ p ] = key;
                }
            }
        }
        __syncthreads();
        // at most 1024 new entries per round, so sorting whenever the buffer is more than half full
        // guarantees the next round fits
        if(cnt > SEL_BUF / 2 || base + blockDim.x * 4 >= ncols) {
            int n2 = 64;  // the buffer holds cnt keys, ~0 beyond: sort the smallest power of two that covers them
            while(n2 < cnt && n2 < SEL_BUF) n2 <<= 1;
            bitonic_sort_lds(buf, n2);
            for(int i = (int)kk + threadIdx.x; i < SEL_BUF; i += blockDim.x) buf[ i ] = ~0ull;
            if(threadIdx.x == 0) { cnt = (int)kk; tau = buf[ kk - 1 ]; }
            __syncthreads();
        }
    }
    for(int i = threadIdx.x; i < (int)kk; i += blockDim.x) mine[ i ] = buf[ i ];
    if(cand && threadIdx.x == 0) cand_cnt[ q ] = 0;
}

// k_rerank: exact-order distances of the kk survivors, then the k smallest by (distance, slot)
template <int METRIC, int G>
__global__ void __launch_bounds__(256) k_rerank(const uint4 *Q, const uint4 *B, uint32_t chunks, const uint64_t *best, uint32_t kk, uint32_t k,
                                                uint32_t *out_slots, float *out_dists)
{
    __shared__ uint64_t keys[ 256 ];
    const uint32_t q = blockIdx.x;
    const int      tid = threadIdx.x, T = blockDim.x, g = tid / G, gl = tid % G, NG = T / G;
    for(int i = tid; i < 256; i += T) keys[ i ] = ~0ull;
    __syncthreads();
    for(uint32_t i = g; i < kk; i += NG) {
        const uint64_t cand = best[ (size_t)q * kk + i ];
        if(cand == ~0ull) continue;
        const uint32_t slot = (uint32_t)(cand & 0xFFFFFFFFu);
        float          d = group_dist<METRIC, G>(Q + (size_t)q * chunks, B + (size_t)slot * chunks, (int)chunks, gl);
        if(gl == G - 1) keys[ i ] = ((uint64_t)f2ord(d) << 32) | slot;
    }
    __syncthreads();
    bitonic_sort_lds(keys, 256);
    for(uint32_t i = tid; i < k; i += T) {
        const uint64_t key = keys[ i ];
        out_slots[ (size_t)q * k + i ] = key == ~0ull ? EMPTY : (uint32_t)(key & 0xFFFFFFFFu);
        out_dists[ (size_t)q * k + i ] = key == ~0ull ? __builtin_inff() : ord2f((uint32_t)(key >> 32));
    }
}

// k_merge_parts: the merge step of the row-partitioned search (index.cpp lantern_gpu_search_partitioned): per query, the
// world x k per-rank results (label, distance; unused tails are label 0 / +inf) -> the global k smallest by (distance, label).
// One wave per query; every candidate's rank is counted against all the others (world * k <= a few hundred).
__global__ void __launch_bounds__(256) k_merge_parts(const uint64_t *labels, const float *dists, uint32_t world, uint32_t nq, uint32_t k,
                                                     uint64_t *out_labels, float *out_dists, uint32_t *out_counts)
{
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if(q >= nq) return;
    const uint32_t total = world * k;
    const size_t   part = (size_t)nq * k;  // entries per rank
    uint32_t       valid = 0;
    for(uint32_t i = lane; i < total; i += 64) {
        const size_t   at = (size_t)(i / k) * part + (size_t)q * k + (i % k);
        const float    d = dists[ at ];
        const uint64_t l = labels[ at ];
        const bool     live = !(d == __builtin_inff() && l == 0);  // an unused tail entry
        uint32_t       rank = 0;
        if(live) {
            valid++;
            const uint32_t dk = f2ord(d);
            for(uint32_t j = 0; j < total; ++j) {
                const size_t   aj = (size_t)(j / k) * part + (size_t)q * k + (j % k);
                const float    dj = dists[ aj ];
                const uint64_t lj = labels[ aj ];
                if(dj == __builtin_inff() && lj == 0) continue;
                const uint32_t djk = f2ord(dj);
                rank += (djk < dk || (djk == dk && (lj < l || (lj == l && j < i)))) ? 1u : 0u;
            }
            if(rank < k) {
                out_labels[ (size_t)q * k + rank ] = l;
                out_dists[ (size_t)q * k + rank ] = d;
            }
        }
    }
    // how many came out, and the unused tail in the library's convention
    for(int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o);
    const uint32_t got = valid < k ? valid : k;
    for(uint32_t i = got + lane; i < k; i += 64) {
        out_labels[ (size_t)q * k + i ] = 0;
        out_dists[ (size_t)q * k + i ] = __builtin_inff();
    }
    if(lane == 0 && out_counts) out_counts[ q ] = got;
}

hipError_t launch_merge_parts(const uint64_t *labels, const float *dists, uint32_t world, uint32_t nq, uint32_t k, uint64_t *out_labels,
                              float *out_dists, uint32_t *out_counts, hipStream_t stream)
{
    if(nq == 0 || k == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_parts, dim3((nq + 3) / 4), dim3(256), 0, stream, labels, dists, world, nq, k, out_labels, out_dists, out_counts);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Row-sharded build (index.cpp add_row_sharded_locked): the candidate lists the ranks found in THEIR shards for row
// first_slot + q -- [world][nq][k] (label = slot + 1, distance), unused entries (0, inf) -- become what the insertion walk hands
// the selection kernel for level 0 of batch member q (item link_off[q] / M): up to `stride` keys (distance, slot) in ascending
// order.  The members of the batch itself (slots >= first_slot: the shards' graphs hold them already) are left out, as they are
// invisible to one another in a one-GPU batch -- two of them choosing each other would ask for a link that is there already.
// One wave per row; ranking by counting, as in k_merge_parts.
__global__ void __launch_bounds__(256) k_merge_candidates(const uint64_t *labels, const float *dists, uint32_t world, uint32_t nq, uint32_t k,
                                                          uint32_t first_slot, const uint32_t *link_off, uint32_t M, uint32_t stride, uint64_t *tops,
                                                          uint32_t *top_count)
{
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if(q >= nq) return;
    const uint32_t item = link_off[ q ] / M;
    const uint32_t total = world * k;
    const size_t   part = (size_t)nq * k;
    const uint64_t batch = (uint64_t)first_slot + 1;  // labels from here on are members of this batch (the row itself among them)
    uint32_t       valid = 0;
    for(uint32_t i = lane; i < total; i += 64) {
        const size_t   at = (size_t)(i / k) * part + (size_t)q * k + (i % k);
        const float    d = dists[ at ];
        const uint64_t l = labels[ at ];
        if(l == 0 || l >= batch) continue;
        valid++;
        const uint32_t dk = f2ord(d);
        uint32_t       rank = 0;
        for(uint32_t j = 0; j < total; ++j) {
            const size_t   aj = (size_t)(j / k) * part + (size_t)q * k + (j % k);
            const uint64_t lj = labels[ aj ];
            if(lj == 0 || lj >= batch) continue;
            const uint32_t djk = f2ord(dists[ aj ]);
            rank += (djk < dk || (djk == dk && (lj < l || (lj == l && j < i)))) ? 1u : 0u;
        }
        if(rank < stride) tops[ (size_t)item * stride + rank ] = make_key(d, (uint32_t)(l - 1));
    }
    for(int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o);
    if(lane == 0) top_count[ item ] = valid < stride ? valid : stride;
}

hipError_t launch_merge_candidates(const uint64_t *labels, const float *dists, uint32_t world, uint32_t nq, uint32_t k, uint32_t first_slot,
                                   const uint32_t *link_off, uint32_t M, uint32_t stride, uint64_t *tops, uint32_t *top_count, hipStream_t stream)
{
    if(nq == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_candidates, dim3((nq + 3) / 4), dim3(256), 0, stream, labels, dists, world, nq, k, first_slot, link_off, M, stride, tops,
                       top_count);
    return hipGetLastError();
}

// f16 rows (8 halves per chunk) -> f32 rows (4 floats per chunk) for the MFMA contraction
__global__ void __launch_bounds__(256) k_dequant_f16(const uint4 *src, size_t nchunks, uint4 *dst)
{
    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[ i ];
        float       f[ 8 ];
        unpack_h2(v.x, f[ 0 ], f[ 1 ]);
        unpack_h2(v.y, f[ 2 ], f[ 3 ]);
        unpack_h2(v.z, f[ 4 ], f[ 5 ]);
        unpack_h2(v.w, f[ 6 ], f[ 7 ]);
        dst[ 2 * i ] = make_uint4(__float_as_uint(f[ 0 ]), __float_as_uint(f[ 1 ]), __float_as_uint(f[ 2 ]), __float_as_uint(f[ 3 ]));
        dst[ 2 * i + 1 ] = make_uint4(__float_as_uint(f[ 4 ]), __float_as_uint(f[ 5 ]), __float_as_uint(f[ 6 ]), __float_as_uint(f[ 7 ]));
    }
}

// i8 rows (16 bytes per chunk) -> f32 rows (four chunks) for the MFMA contraction; integers up to 100 are exact in f32
__global__ void __launch_bounds__(256) k_dequant_i8(const uint4 *src, size_t nchunks, uint4 *dst)
{
    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * blockDim.x) {
        const uint4    v = src[ i ];
        const uint32_t w[ 4 ] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for(int j = 0; j < 4; ++j) {
            const float f0 = (float)(int8_t)(w[ j ] & 0xFF), f1 = (float)(int8_t)((w[ j ] >> 8) & 0xFF);
            const float f2 = (float)(int8_t)((w[ j ] >> 16) & 0xFF), f3 = (float)(int8_t)(w[ j ] >> 24);
            dst[ 4 * i + j ] = make_uint4(__float_as_uint(f0), __float_as_uint(f1), __float_as_uint(f2), __float_as_uint(f3));
        }
    }
}

hipError_t launch_dequant_i8(const uint4 *src, size_t nchunks, uint4 *dst, hipStream_t stream)
{
    if(nchunks == 0) return hipSuccess;
    size_t blocks = (nchunks + 255) / 256;
    if(blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_dequant_i8, dim3((uint32_t)blocks), dim3(256), 0, stream, src, nchunks, dst);
    return hipGetLastError();
}

hipError_t launch_dequant_f16(const uint4 *src, size_t nchunks, uint4 *dst, hipStream_t stream)
{
    if(nchunks == 0) return hipSuccess;
    size_t blocks = (nchunks + 255) / 256;
    if(blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_dequant_f16, dim3((uint32_t)blocks), dim3(256), 0, stream, src, nchunks, dst);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
hipError_t launch_row_norms(const uint4 *rows, uint32_t n, uint32_t chunks, float *out, hipStream_t stream)
{
    if(n == 0) return hipSuccess;
    const int G_ = group_lanes_for(chunks);
    uint32_t  blocks = (uint32_t)(((uint64_t)n * G_ + 255) / 256);
    if(blocks > 16384) blocks = 16384;
    switch(G_) {
        case 64: hipLaunchKernelGGL((k_row_norms<64>), dim3(blocks), dim3(256), 0, stream, rows, n, chunks, out); break;
        case 32: hipLaunchKernelGGL((k_row_norms<32>), dim3(blocks), dim3(256), 0, stream, rows, n, chunks, out); break;
        case 16: hipLaunchKernelGGL((k_row_norms<16>), dim3(blocks), dim3(256), 0, stream, rows, n, chunks, out); break;
        default: hipLaunchKernelGGL((k_row_norms<8>), dim3(blocks), dim3(256), 0, stream, rows, n, chunks, out);
    }
    return hipGetLastError();
}

// persistent grid of k_dense_f32: DENSE_WGS_PER_CU workgroups per CU of the current device (fewer if there are fewer tiles)
static uint32_t dense_grid(uint32_t tiles)
{
    static int cus[ 64 ];  // per device ordinal; filled once (racing fillers write the same value)
    int        dev = 0;
    if(hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if(cus[ dev ] == 0) {
        int v = 0;
        cus[ dev ] = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0 ? v : 256;
    }
    const uint32_t g = (uint32_t)DENSE_WGS_PER_CU * (uint32_t)cus[ dev ];
    return tiles < g ? tiles : g;
}

hipError_t launch_dense(int metric, const uint4 *Q, uint32_t nq, const uint4 *B, uint32_t nb, uint32_t chunks, const float *qn,
                        const float *bn, float *out, uint32_t ldo, hipStream_t stream)
{
    if(nq == 0 || nb == 0) return hipSuccess;
    const uint32_t stride = chunks * 4;
    if(metric == M_HAMMING || metric == M_COS_B1) {
        dim3 grid((nb + 255) / 256, (nq + 15) / 16);
        if(metric == M_HAMMING)
            hipLaunchKernelGGL(k_dense_ham<false>, grid, dim3(256), 16 * stride * 4, stream, (const uint32_t *)Q, nq, (const uint32_t *)B, nb, stride, out, ldo);
        else
            hipLaunchKernelGGL(k_dense_ham<true>, grid, dim3(256), 16 * stride * 4, stream, (const uint32_t *)Q, nq, (const uint32_t *)B, nb, stride, out, ldo);
        return hipGetLastError();
    }
    const uint32_t tiles = ((nq + BM - 1) / BM) * ((nb + BN - 1) / BN);
    const DenseTopk none = { nullptr, 0, nullptr, nullptr, 0, 0 };
    if(metric == M_L2SQ)
        hipLaunchKernelGGL((k_dense_f32<M_L2SQ>), dim3(dense_grid(tiles)), dim3(256), 0, stream, (const float *)Q, nq, (const float *)B, nb, stride, qn, bn,
                           out, ldo, none);
    else
        hipLaunchKernelGGL((k_dense_f32<M_COS>), dim3(dense_grid(tiles)), dim3(256), 0, stream, (const float *)Q, nq, (const float *)B, nb, stride, qn, bn,
                           out, ldo, none);
    return hipGetLastError();
}

// the contraction with the top-kk filter fused into its epilogue (l2sq / cos): candidates go to cand / cnt, see k_dense_f32
hipError_t launch_dense_topk(int metric, const uint4 *Q, uint32_t nq, const uint4 *B, uint32_t nb, uint32_t chunks, const float *qn,
                             const float *bn, const uint64_t *best, uint32_t kk, uint64_t *cand, uint32_t *cnt, uint32_t cap, uint32_t c_base,
                             hipStream_t stream)
{
    if(nq == 0 || nb == 0) return hipSuccess;
    if(metric != M_L2SQ && metric != M_COS) return hipErrorInvalidValue;
    const uint32_t  stride = chunks * 4;
    const uint32_t  tiles = ((nq + BM - 1) / BM) * ((nb + BN - 1) / BN);
    const DenseTopk tk = { best, kk, cand, cnt, cap, c_base };
    if(metric == M_L2SQ)
        hipLaunchKernelGGL((k_dense_f32<M_L2SQ, true>), dim3(dense_grid(tiles)), dim3(256), 0, stream, (const float *)Q, nq, (const float *)B, nb, stride, qn,
                           bn, (float *)nullptr, 0u, tk);
    else
        hipLaunchKernelGGL((k_dense_f32<M_COS, true>), dim3(dense_grid(tiles)), dim3(256), 0, stream, (const float *)Q, nq, (const float *)B, nb, stride, qn,
                           bn, (float *)nullptr, 0u, tk);
    return hipGetLastError();
}

hipError_t launch_select(const float *dist, uint32_t ldo, uint32_t nq, uint32_t ncols, uint32_t c_base, uint64_t *best, uint32_t kk,
                         hipStream_t stream)
{
    if(nq == 0 || ncols == 0) return hipSuccess;
    hipLaunchKernelGGL(k_select, dim3(nq), dim3(256), 0, stream, dist, ldo, ncols, c_base, best, kk, (const uint64_t *)nullptr, (uint32_t *)nullptr, 0u,
                       (uint32_t *)nullptr);
    return hipGetLastError();
}

hipError_t launch_select_candidates(uint32_t nq, uint64_t *best, uint32_t kk, const uint64_t *cand, uint32_t *cnt, uint32_t cap, uint32_t *overflow,
                                    hipStream_t stream)
{
    if(nq == 0) return hipSuccess;
    hipLaunchKernelGGL(k_select, dim3(nq), dim3(256), 0, stream, (const float *)nullptr, 0u, 0u, 0u, best, kk, cand, cnt, cap, overflow);
    return hipGetLastError();
}

hipError_t launch_rerank(int metric, const uint4 *Q, uint32_t nq, const uint4 *B, uint32_t chunks, const uint64_t *best, uint32_t kk,
                         uint32_t k, uint32_t *out_slots, float *out_dists, hipStream_t stream)
{
    if(nq == 0) return hipSuccess;
    const int G_ = group_lanes_for(chunks);
#define RR(MM, GG) hipLaunchKernelGGL((k_rerank<MM, GG>), dim3(nq), dim3(256), 0, stream, Q, B, chunks, best, kk, k, out_slots, out_dists)
#define RRG(MM) switch(G_) { case 64: RR(MM, 64); break; case 32: RR(MM, 32); break; case 16: RR(MM, 16); break; default: RR(MM, 8); }
    switch(metric) {
        case M_L2SQ: RRG(M_L2SQ); break;
        case M_COS: RRG(M_COS); break;
        case M_HAMMING: RRG(M_HAMMING); break;
        case M_COS_B1: RRG(M_COS_B1); break;
        case M_L2SQ_F16: RRG(M_L2SQ_F16); break;
        case M_COS_F16: RRG(M_COS_F16); break;
        case M_L2SQ_I8: RRG(M_L2SQ_I8); break;
        case M_COS_I8: RRG(M_COS_I8); break;
        default: return hipErrorInvalidValue;
    }
#undef RRG
#undef RR
    return hipGetLastError();
}

}  // namespace lgpu
