// Key/value radix sort of the (bucket, entry) pairs of the MSM: least significant digit first, up to
// eight key bits per pass, every pass stable.
//
// A pass over the pairs is three launches:
//   k_radix_hist     one block per tile of kTile pairs: how many keys of the tile fall into each bin
//   k_radix_scan_*   exclusive prefix of those counts over (bin, tile) in bin-major order: where the
//                    keys of bin b of tile t go (two levels: chunks of tiles, then the chunk totals)
//   k_radix_scatter  one block per tile again: a stable rank for every key inside its tile (wave
//                    level: a counter per wave and bin in LDS, advanced by one add-with-return per
//                    key, or by ballots: see k_rank_selftest), the tile reordered by bin in LDS, then
//                    written out in runs of consecutive addresses
// The pairs cross HBM twice per pass (plus the keys once more for the counts); with 2^13 pairs per
// tile and 2^8 bins a run is 128 bytes for uniformly distributed digits.  Skewed inputs (all keys
// equal) only make the runs longer.
#include <algorithm>
#include <cstring>
#include <vector>

#include "context.hpp"
#include "msm_core.cuh"

namespace plonk {

#ifndef PLONK_SORT_ITEMS
#define PLONK_SORT_ITEMS 32
#endif
#ifndef PLONK_SORT_THREADS
#define PLONK_SORT_THREADS 256
#endif
static constexpr int kSortThreads = PLONK_SORT_THREADS, kSortItems = PLONK_SORT_ITEMS, kTile = kSortThreads * kSortItems;
static constexpr int kSortWaves = kSortThreads / 64, kWaveSpan = 64 * kSortItems;
static constexpr int kMaxBins = 256, kChunkTiles = 128;

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// Lanes of one instruction that hold the same digit hit the same LDS counter, and the LDS serves them one after
// the other: with skewed keys (every scalar equal, most digits zero) a plain add per lane takes 64 turns per
// instruction.  While the first pending lane's digit is shared by at least kCrowd pending lanes, those lanes are
// served together -- one add of their number by the first of them, ranks by position among them -- and what is
// left goes through the per-lane adds.  Evenly spread digits pay one failed test per key.
// Returns (WANT_RANK) the counter's value before this lane's key, as if the valid lanes had added 1 each in
// ascending lane order.
static constexpr uint32_t kCrowd = 16;

template <bool WANT_RANK>
__device__ __forceinline__ uint32_t count_digit(uint32_t *counters, uint32_t d, bool valid, uint32_t lane) {
    uint64_t todo = __ballot(valid);
    uint32_t rank = 0;
    while (todo) {  // wave-uniform
        const int first = __ffsll((unsigned long long)todo) - 1;
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, first);
        const uint64_t same = __ballot(d == d0) & todo;
        const uint32_t cnt = (uint32_t)__popcll(same);
        if (cnt < kCrowd) break;
        uint32_t prior = 0;
        if ((int)lane == first) {
            if (WANT_RANK) prior = atomicAdd(&counters[d0], cnt);
            else atomicAdd(&counters[d0], cnt);
        }
        if (WANT_RANK) {
            prior = (uint32_t)__builtin_amdgcn_readlane((int)prior, first);
            if ((same >> lane) & 1)
                rank = prior + __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
        }
        todo &= ~same;
    }
    if ((todo >> lane) & 1) {
        if (WANT_RANK) rank = atomicAdd(&counters[d], 1u);
        else atomicAdd(&counters[d], 1u);
    }
    return rank;
}

// hist[tile][bin] = number of keys of the tile whose digit (key >> shift) & mask is bin
__global__ __launch_bounds__(kSortThreads) void k_radix_hist(const uint32_t *__restrict__ keys, size_t n, uint32_t shift,
                                                            uint32_t mask, uint32_t *__restrict__ hist) {
    __shared__ uint32_t count[kMaxBins];
    if (threadIdx.x < kMaxBins) count[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const size_t base = (size_t)blockIdx.x * kTile;
    if (base + kTile <= n) {
        // whole tile: 16-byte loads, four keys per lane and load
        const uint4 *src = reinterpret_cast<const uint4 *>(keys + base);
#pragma unroll
        for (int r = 0; r < kSortItems / 4; r++) {
            const uint4 k = src[r * kSortThreads + threadIdx.x];
            count_digit<false>(count, (k.x >> shift) & mask, true, lane);
            count_digit<false>(count, (k.y >> shift) & mask, true, lane);
            count_digit<false>(count, (k.z >> shift) & mask, true, lane);
            count_digit<false>(count, (k.w >> shift) & mask, true, lane);
        }
    } else {
        // the same trip count for every lane of a wave (count_digit votes)
        for (size_t i0 = base; i0 < n; i0 += kSortThreads) {
            const size_t i = i0 + threadIdx.x;
            const bool valid = i < n;
            count_digit<false>(count, valid ? (keys[i] >> shift) & mask : 0u, valid, lane);
        }
    }
    __syncthreads();
    if (threadIdx.x < kMaxBins) hist[(size_t)blockIdx.x * kMaxBins + threadIdx.x] = count[threadIdx.x];
}

// The two scan levels work on counts laid out as [tile][bins] with bins = 64, 128 or 256.  A block is kScanThreads lanes:
// kScanThreads / bins groups of `bins` lanes, lane b of a group owns bin b (the more groups, the shorter the
// chain of dependent loads each lane walks).
//
// Level 1: one block per chunk of chunk_tiles = kGroupTiles * groups tiles; group g scans the chunk's g-th run
// of kGroupTiles tiles: hist[tile][b] becomes the number of keys of bin b in the earlier tiles of the chunk;
// chunk_total[chunk][b] their number in the whole chunk.
static constexpr uint32_t kScanThreads = 1024, kGroupTiles = 32;
static_assert(kChunkTiles == kGroupTiles * (kScanThreads / kMaxBins), "chunk of a pass with kMaxBins bins");

// values[i * stride] for i in [first, last) become their exclusive running sum (starting from `run`); returns
// the total.  Sixteen loads in flight at a time: the loop is bound by memory latency.
__device__ __forceinline__ uint32_t scan_column(uint32_t *__restrict__ values, uint32_t first, uint32_t last,
                                                size_t stride, uint32_t run) {
    for (uint32_t i = first; i < last; i += 16) {
        uint32_t v[16];
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) v[j] = i + j < last ? values[(size_t)(i + j) * stride] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            if (i + j < last) values[(size_t)(i + j) * stride] = run;
            run += v[j];
        }
    }
    return run;
}
__device__ __forceinline__ void add_to_column(uint32_t *__restrict__ values, uint32_t first, uint32_t last, size_t stride,
                                              uint32_t add) {
#pragma unroll 16
    for (uint32_t i = first; i < last; i++) values[(size_t)i * stride] += add;
}

__global__ __launch_bounds__(kScanThreads) void k_radix_scan_tiles(uint32_t *__restrict__ hist, uint32_t ntiles, uint32_t bins,
                                                         uint32_t *__restrict__ chunk_total) {
    __shared__ uint32_t group_total[kScanThreads];
    const uint32_t t = threadIdx.x, b = t % bins, g = t / bins, groups = kScanThreads / bins;
    uint32_t first = (blockIdx.x * groups + g) * kGroupTiles;
    if (first > ntiles) first = ntiles;
    const uint32_t last = first + kGroupTiles < ntiles ? first + kGroupTiles : ntiles;
    const uint32_t run = scan_column(hist + b, first, last, bins, 0u);
    if (groups == 1) {
        chunk_total[(size_t)blockIdx.x * bins + b] = run;
        return;
    }
    group_total[t] = run;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t j = 0; j < groups; j++) {
        const uint32_t v = group_total[j * bins + b];
        before += j < g ? v : 0u;
        total += v;
    }
    if (before) add_to_column(hist + b, first, last, bins, before);
    if (g == 0) chunk_total[(size_t)blockIdx.x * bins + b] = total;
}

// Level 2: one block: chunk_total[chunk][b] becomes the position of the first key of bin b of that chunk in
// the sorted order (all keys of smaller bins, then the earlier chunks' keys of bin b).  Group g takes the g-th
// run of chunks.
__global__ __launch_bounds__(kScanThreads) void k_radix_scan_chunks(uint32_t *__restrict__ chunk_total, uint32_t nchunks,
                                                          uint32_t bins) {
    __shared__ uint32_t group_total[kScanThreads];
    __shared__ uint32_t bin_total[kMaxBins];
    const uint32_t t = threadIdx.x, b = t % bins, g = t / bins, groups = kScanThreads / bins;
    const uint32_t per_group = (nchunks + groups - 1) / groups;
    const uint32_t first = g * per_group < nchunks ? g * per_group : nchunks;
    const uint32_t last = first + per_group < nchunks ? first + per_group : nchunks;
    const uint32_t run = scan_column(chunk_total + b, first, last, bins, 0u);
    group_total[t] = run;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t j = 0; j < groups; j++) {
        const uint32_t v = group_total[j * bins + b];
        before += j < g ? v : 0u;
        total += v;
    }
    // exclusive prefix of the bin totals (entries from `bins` on are zero; a logarithmic scan keeps it short)
    if (t < kMaxBins) bin_total[t] = t < bins ? total : 0u;
    __syncthreads();
    for (uint32_t off = 1; off < kMaxBins; off <<= 1) {
        const uint32_t v = t >= off && t < kMaxBins ? bin_total[t - off] : 0u;
        __syncthreads();
        if (t < kMaxBins) bin_total[t] += v;
        __syncthreads();
    }
    const uint32_t base = bin_total[b] - total + before;  // inclusive - own = exclusive
    add_to_column(chunk_total + b, first, last, bins, base);
}

// Lanes of the wave (among those with `valid`) whose digit equals this lane's.
__device__ __forceinline__ uint64_t same_digit_lanes(uint32_t digit, uint32_t bits, bool valid) {
    uint64_t m = __ballot(valid);
    for (uint32_t i = 0; i < bits; i++) {
        const bool bit = (digit >> i) & 1u;
        const uint64_t set = __ballot(bit);
        m &= bit ? set : ~set;
    }
    return m;
}

// Stable scatter of one tile of WAVES * 64 * ITEMS pairs held in registers, by every lane of a block of
// WAVES waves (WAVES * 64 >= kMaxBins, WAVES >= kMaxBins / 64).  Element e of the tile belongs to wave e / (64 * ITEMS), round
// (e % (64 * ITEMS)) / 64, lane e % 64: ranks follow that order.  tile_hist: the tile's row of scanned counts,
// chunk_row: its chunk's row of bin positions (both `bins`-strided rows).  The shared memory may be used again
// at once by the next call.
template <int WAVES, int ITEMS>
struct ScatterMem {
    uint32_t wave_count[WAVES][kMaxBins];  // per wave and bin: running count, then start position
    uint32_t bin_start[kMaxBins];          // first position of the bin in the reordered tile
    uint32_t bin_dest[kMaxBins];           // where that position goes in the output
    uint32_t skey[WAVES * 64 * ITEMS], sval[WAVES * 64 * ITEMS];
};

template <int WAVES, int ITEMS, bool ATOMIC_RANK>
__device__ __forceinline__ void tile_scatter(ScatterMem<WAVES, ITEMS> &m, const uint32_t (&key)[ITEMS],
                                             const uint32_t (&val)[ITEMS], uint32_t tile_count, uint32_t shift,
                                             uint32_t bits, const uint32_t *__restrict__ tile_hist,
                                             const uint32_t *__restrict__ chunk_row,
                                             uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
    constexpr uint32_t kThreads = WAVES * 64, kSpan = 64 * ITEMS;
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t mask = (1u << bits) - 1;
    for (uint32_t i = t; i < WAVES * kMaxBins; i += kThreads) (&m.wave_count[0][0])[i] = 0;
    __syncthreads();

    uint32_t rank[ITEMS];
    const uint32_t wave_first = w * kSpan;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t e = wave_first + r * 64 + lane;
        const bool valid = e < tile_count;
        const uint32_t d = (key[r] >> shift) & mask;
        if (ATOMIC_RANK) {
            // one LDS operation: the counter's value before this lane's increment (see k_rank_selftest for the
            // order of the lanes of one instruction that this relies on)
            rank[r] = count_digit<true>(m.wave_count[w], d, valid, lane);
        } else {
            const uint64_t same = same_digit_lanes(d, bits, valid);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
            uint32_t prior = 0;
            if (valid) {
                prior = m.wave_count[w][d];
                // the wave's LDS operations complete in program order: every lane of the group has read the count
                // before its first lane replaces it
                if (below == 0) m.wave_count[w][d] = prior + (uint32_t)__popcll(same);
            }
            rank[r] = prior + below;
        }
    }
    __syncthreads();
    // per bin: start in the reordered tile, start of every wave's share, destination in the output.  Lane b of
    // the block's first kMaxBins lanes owns bin b: a prefix sum inside each wave (64 bins) by shuffles, the
    // waves' totals through LDS.
    {
        const uint32_t b = t;
        const bool owner = t < kMaxBins;
        uint32_t c[WAVES], total = 0;
        if (owner) {
#pragma unroll
            for (int i = 0; i < WAVES; i++) {
                c[i] = m.wave_count[i][b];
                total += c[i];
            }
        }
        uint32_t incl = total;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
            if ((int)lane >= off) incl += v;
        }
        if (owner && lane == 63) m.bin_start[w] = incl;  // the wave's 64 bins together (read back below)
        __syncthreads();
        uint32_t start = incl - total;
        if (owner) {
            for (uint32_t j = 0; j < w; j++) start += m.bin_start[j];
        }
        __syncthreads();
        if (owner) {
            m.bin_start[b] = start;
            uint32_t run = start;
#pragma unroll
            for (int i = 0; i < WAVES; i++) {
                m.wave_count[i][b] = run;
                run += c[i];
            }
            if (b <= mask) m.bin_dest[b] = chunk_row[b] + tile_hist[b];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const uint32_t e = wave_first + r * 64 + lane;
        if (e < tile_count) {
            const uint32_t d = (key[r] >> shift) & mask;
            const uint32_t pos = m.wave_count[w][d] + rank[r];
            m.skey[pos] = key[r];
            m.sval[pos] = val[r];
        }
    }
    __syncthreads();
    for (uint32_t i = t; i < tile_count; i += kThreads) {
        const uint32_t k = m.skey[i];
        const uint32_t d = (k >> shift) & mask;
        const size_t dest = (size_t)m.bin_dest[d] + (i - m.bin_start[d]);
        keys_out[dest] = k;
        vals_out[dest] = m.sval[i];
    }
}

// Consecutive workgroups go to the eight XCDs in turn, each with an L2 of its own; tiles that follow each other
// write neighbouring runs of every bin, i.e. halves of the same 128-byte lines.  Numbering the tiles so that an XCD
// works through a contiguous range lets those lines fill up inside one L2.
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t block, uint32_t nblocks) {
    const uint32_t per = nblocks / 8;
    return block < 8 * per ? (block % 8) * per + block / 8 : block;
}

template <bool ATOMIC_RANK>
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(const uint32_t *__restrict__ keys_in,
                                                               const uint32_t *__restrict__ vals_in, size_t n,
                                                               uint32_t shift, uint32_t bits,
                                                               const uint32_t *__restrict__ hist,
                                                               const uint32_t *__restrict__ chunk_base,
                                                               uint32_t *__restrict__ keys_out,
                                                               uint32_t *__restrict__ vals_out) {
    __shared__ ScatterMem<kSortWaves, kSortItems> mem;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t tile = xcd_contiguous(blockIdx.x, gridDim.x);
    const size_t tile_base = (size_t)tile * kTile;
    const uint32_t tile_count = (uint32_t)(n - tile_base < (size_t)kTile ? n - tile_base : (size_t)kTile);
    uint32_t key[kSortItems], val[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const uint32_t e = w * kWaveSpan + r * 64 + lane;
        const bool valid = e < tile_count;
        key[r] = valid ? keys_in[tile_base + e] : 0u;
        val[r] = valid ? vals_in[tile_base + e] : 0u;
    }
    tile_scatter<kSortWaves, kSortItems, ATOMIC_RANK>(mem, key, val, tile_count, shift, bits, hist + (size_t)tile * kMaxBins,
                                         chunk_base + (size_t)(tile / kChunkTiles) * kMaxBins, keys_out, vals_out);
}

// ---- first pass straight from the scalars ------------------------------------------------------------------
// A block takes kDigitTile consecutive scalars; its digits of window w are sort tile w * (n / kDigitTile) + block
// of the first pass (any numbering of the tiles will do for the first pass of a sort from the lowest digit up;
// both kernels use this one).  Pair of scalar i and window w: key = |digit| - 1, or nbuckets for digit 0 (a pair
// that contributes nothing: the sort looks at the bucket bits only, so it travels with bucket 0 and the
// accumulation passes over it), val = w * srs_n + first + i with the digit's sign in bit 31.
static constexpr int kDigitItems = 8, kDigitWaves = 4, kDigitTile = kDigitWaves * 64 * kDigitItems;

// The next window's pair of a scalar that is shifted down as its windows are used up (the lowest c bits of k
// are the window; extracting window w from a scalar kept whole costs a chain of selects per limb).
__device__ __forceinline__ void next_digit_pair(Fr &k, bool opposite, uint32_t &carry, int c, uint32_t nbuckets,
                                                uint32_t entry, uint32_t &key, uint32_t &val) {
    int32_t d = (int32_t)((k.l[0] & ((1u << c) - 1)) + carry);
#pragma unroll
    for (int i = 0; i + 1 < FrCfg::N; i++) k.l[i] = __builtin_amdgcn_alignbit(k.l[i + 1], k.l[i], (uint32_t)c);
    k.l[FrCfg::N - 1] >>= c;
    // the signed digit of msm_core.cuh: d > 2^(c-1) -> d - 2^c and a carry into the next window
    carry = d > (int32_t)(1u << (c - 1)) ? 1u : 0u;
    if (carry) d -= (int32_t)(1u << c);
    const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
    key = mag == 0 ? nbuckets : mag - 1;
    val = entry | (((d < 0) != opposite) ? 0x80000000u : 0u);
}

// counts[(w * (n / kDigitTile) + block) * bins + b] = digits of window w of the block's scalars with key & mask == b
__global__ __launch_bounds__(kDigitWaves * 64) void k_digits_hist(const Fr *__restrict__ scalars, uint32_t n, int c,
                                                                  int nwin, uint32_t nbuckets, uint32_t bins,
                                                                  uint32_t *__restrict__ counts) {
    extern __shared__ uint32_t count[];  // [nwin][bins]
    for (uint32_t j = threadIdx.x; j < (uint32_t)nwin * bins; j += blockDim.x) count[j] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kDigitTile, mask = (bins - 1) & (nbuckets - 1);  // bucket bits only
    Fr raw[kDigitItems];
#pragma unroll
    for (int r = 0; r < kDigitItems; r++) raw[r] = scalars[base + r * (kDigitWaves * 64) + threadIdx.x];
#pragma unroll
    for (int r = 0; r < kDigitItems; r++) {
        Fr k = fe_from_mont(raw[r]);
        const bool opposite = msm_folds_scalars(c) && scalar_fold(k);
        uint32_t carry = 0, key, val;
        for (int w = 0; w < nwin; w++) {
            next_digit_pair(k, opposite, carry, c, nbuckets, 0u, key, val);
            count_digit<false>(count + (uint32_t)w * bins, key & mask, true, threadIdx.x & 63);
        }
    }
    __syncthreads();
    const uint32_t tiles_per_window = n / kDigitTile;
    for (uint32_t j = threadIdx.x; j < (uint32_t)nwin * bins; j += blockDim.x) {
        const uint32_t w = j / bins, b = j % bins;
        counts[((size_t)w * tiles_per_window + blockIdx.x) * bins + b] = count[j];
    }
}

template <bool ATOMIC_RANK>
__global__ __launch_bounds__(kDigitWaves * 64) void k_digits_scatter(const Fr *__restrict__ scalars, uint32_t n,
                                                                     uint32_t srs_n, uint32_t first, int c, int nwin,
                                                                     uint32_t nbuckets, uint32_t bits,
                                                                     const uint32_t *__restrict__ hist,
                                                                     const uint32_t *__restrict__ chunk_base,
                                                                     uint32_t chunk_tiles,
                                                                     uint32_t *__restrict__ keys_out,
                                                                     uint32_t *__restrict__ vals_out) {
    __shared__ ScatterMem<kDigitWaves, kDigitItems> mem;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, bins = 1u << bits;
    // rows of `bins` counts, of which a narrow window plan uses the first nbuckets only (bucket bits, see above)
    const uint32_t digit_bits = bins < nbuckets ? bits : (uint32_t)(c - 1);
    const uint32_t tiles_per_window = n / kDigitTile, block = xcd_contiguous(blockIdx.x, gridDim.x);
    Fr k[kDigitItems];
    uint32_t carry[kDigitItems], index[kDigitItems];
    bool opposite[kDigitItems];
#pragma unroll
    for (int r = 0; r < kDigitItems; r++) {
        index[r] = block * kDigitTile + wave * (64 * kDigitItems) + r * 64 + lane;
        k[r] = fe_from_mont(scalars[index[r]]);
        opposite[r] = msm_folds_scalars(c) && scalar_fold(k[r]);
        carry[r] = 0;
    }
    for (int w = 0; w < nwin; w++) {
        uint32_t key[kDigitItems], val[kDigitItems];
#pragma unroll
        for (int r = 0; r < kDigitItems; r++)
            next_digit_pair(k[r], opposite[r], carry[r], c, nbuckets, (uint32_t)w * srs_n + first + index[r], key[r], val[r]);
        const size_t tile = (size_t)w * tiles_per_window + block;
        tile_scatter<kDigitWaves, kDigitItems, ATOMIC_RANK>(mem, key, val, kDigitTile, 0u, digit_bits, hist + tile * bins,
                                               chunk_base + (tile / chunk_tiles) * bins, keys_out, vals_out);
    }
}

// ---- which ranking the scatter kernels use -----------------------------------------------------------------
// A stable rank needs, for lanes of one wave that hold the same digit, "how many lower lanes hold it too".  One
// LDS add-with-return per lane on a counter per digit gives exactly that IF the lanes of one instruction that
// hit the same counter are served in ascending lane order -- which this part does, but which no manual
// promises.  So it is checked, once per context: a wave ranks pseudo-random digits of every width both ways
// (the ballot form above, which relies on nothing, and the add-with-return form); the context uses the
// one-instruction form only if all ranks agree, and the ballot form (about 50 vector instructions per key
// instead of one LDS operation) otherwise.
__global__ __launch_bounds__(64) void k_rank_selftest(uint32_t *__restrict__ mismatches) {
    __shared__ uint32_t by_ballot[kMaxBins], by_add[kMaxBins];
    const uint32_t lane = threadIdx.x;
    uint32_t bad = 0;
    for (uint32_t bits = 0; bits <= 8; bits++) {
        for (uint32_t i = lane; i < (uint32_t)kMaxBins; i += 64) by_ballot[i] = by_add[i] = 0;
        __syncthreads();
        for (uint32_t round = 0; round < 24; round++) {
            uint32_t h = (lane * 2654435761u) ^ (round * 40503u + bits * 977u);
            h ^= h >> 13;
            h *= 0x5bd1e995u;
            h ^= h >> 15;
            const uint32_t d = h & ((1u << bits) - 1);
            const bool valid = round % 5 != 4 || lane % 3 != 1;  // some rounds with idle lanes
            const uint64_t same = same_digit_lanes(d, bits, valid);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
            uint32_t want = 0, got = 0;
            if (valid) {
                const uint32_t prior = by_ballot[d];
                if (below == 0) by_ballot[d] = prior + (uint32_t)__popcll(same);
                want = prior + below;
                got = atomicAdd(&by_add[d], 1u);
            }
            bad += want != got;
        }
        __syncthreads();
    }
    if (bad) atomicAdd(mismatches, bad);
}

// The second half of the check runs the real thing: both production scatter kernels (pairs in, scalars in) on
// whole four-wave tiles with keys that are partly spread and partly crowded, two passes each, and the host looks
// at what came out: keys ascending and, among equal keys, entries in their original order -- which is all a
// stable sort can produce, whatever ranking it uses.
static int selftest_real_kernels(plonk_hip_ctx *ctx, bool *ok) {
    *ok = false;
    const size_t n = 2 * (size_t)kTile + 1234;           // pairs of the plain sort: two whole tiles and a ragged one
    const size_t ns = 2 * (size_t)kDigitTile;            // scalars of the sort that starts from the scalars
    const int c = 13, nwin = (255 + c - 1) / c;         // 20 windows of 13 bits: two passes (6 + 6 bucket bits)
    const uint32_t nbuckets = 1u << (c - 1);
    const size_t total = std::max(n, ns * (size_t)nwin), tmp_bytes = sort_pairs_temp_bytes(total, 16);
    const uint32_t bucket_mask = nbuckets - 1;
    if (!sort_pairs_of_scalars_applies(ns, c, nwin)) return PLONK_HIP_ERR_INVALID_ARG;
    std::vector<uint32_t> keys(n), vals(n), out_k(total), out_v(total);
    std::vector<Fr> scalars(ns);
    uint32_t h = 12345u;
    auto next = [&h]() {
        h ^= h << 13;
        h ^= h >> 17;
        h ^= h << 5;
        return h;
    };
    for (size_t i = 0; i < n; i++) {
        const uint32_t r = next();
        // stretches of spread keys, of two alternating keys and of one key
        keys[i] = (i / 700) % 3 == 0 ? (r & 0xffffu) : ((i / 700) % 3 == 1 ? 0x1234u + (r & 1u) * 0x0100u : 0x00ffu);
        vals[i] = (uint32_t)i;
    }
    for (size_t i = 0; i < ns; i++) {
        for (int j = 0; j < FrCfg::N; j++) scalars[i].l[j] = (i / 300) % 2 ? next() : 0x9e3779b9u * (uint32_t)(j + 1);
        scalars[i].l[FrCfg::N - 1] &= 0x3fffffffu;  // below the modulus
    }
    uint32_t *d[4] = {};
    void *d_tmp = nullptr;
    Fr *d_sc = nullptr;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipMalloc((void **)&d[i], (total + 64) * 4);
    if (e == hipSuccess) e = hipMalloc(&d_tmp, tmp_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_sc, ns * sizeof(Fr));
    int rc = PLONK_HIP_OK;
    bool good = e == hipSuccess;
    if (good) {
        // plain sort, 16 key bits: two passes, result back in the first pair of buffers
        e = hipMemcpyAsync(d[0], keys.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d[2], vals.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) rc = sort_pairs(ctx, d[0], d[1], d[2], d[3], n, 16, d_tmp, tmp_bytes);
        if (e == hipSuccess && rc == PLONK_HIP_OK) e = hipMemcpyAsync(out_k.data(), d[0], n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && rc == PLONK_HIP_OK) e = hipMemcpyAsync(out_v.data(), d[2], n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && rc == PLONK_HIP_OK) e = hipStreamSynchronize(ctx->stream);
        good = e == hipSuccess && rc == PLONK_HIP_OK;
        for (size_t i = 0; good && i < n; i++) {
            good = out_v[i] < n && keys[out_v[i]] == out_k[i] &&
                   (i == 0 || out_k[i - 1] < out_k[i] || (out_k[i - 1] == out_k[i] && out_v[i - 1] < out_v[i]));
        }
    }
    if (good) {
        // from the scalars: entry = window * ns + index, so equal keys must come out with ascending entries
        const size_t pairs = ns * (size_t)nwin;
        e = hipMemcpyAsync(d_sc, scalars.data(), ns * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            rc = sort_pairs_of_scalars(ctx, d_sc, ns, ns, 0, c, nwin, nbuckets, d[0], d[1], d[2], d[3], d_tmp, tmp_bytes);
        const int res = (sort_pairs_passes(c - 1) & 1) ? 1 : 0;  // which pair of buffers holds the result
        if (e == hipSuccess && rc == PLONK_HIP_OK) e = hipMemcpyAsync(out_k.data(), d[res], pairs * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && rc == PLONK_HIP_OK) e = hipMemcpyAsync(out_v.data(), d[2 + res], pairs * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && rc == PLONK_HIP_OK) e = hipStreamSynchronize(ctx->stream);
        good = e == hipSuccess && rc == PLONK_HIP_OK;
        for (size_t i = 0; good && i < pairs; i++) {
            const uint32_t entry = out_v[i] & 0x7fffffffu;
            // ordered by bucket (a digit-0 pair, key nbuckets, counts as bucket 0), entries ascending inside one
            const uint32_t b = out_k[i] & bucket_mask, before = i ? out_k[i - 1] & bucket_mask : 0u;
            good = out_k[i] <= nbuckets && entry < pairs &&
                   (i == 0 || before < b || (before == b && (out_v[i - 1] & 0x7fffffffu) < entry));
        }
    }
    for (int i = 0; i < 4; i++)
        if (d[i]) (void)hipFree(d[i]);
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_sc) (void)hipFree(d_sc);
    PLONK_HIP_TRY(ctx, e);
    PLONK_TRY(rc);
    *ok = good;
    return PLONK_HIP_OK;
}

int sort_selftest(plonk_hip_ctx *ctx) {
    uint32_t *d_bad = nullptr, bad = 1;
    PLONK_HIP_TRY(ctx, hipMalloc(&d_bad, 4));
    hipError_t e = hipMemsetAsync(d_bad, 0, 4, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_rank_selftest, dim3(4), dim3(64), 0, ctx->stream, d_bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_bad);
    PLONK_HIP_TRY(ctx, e);
    ctx->lds_rank_in_lane_order = bad == 0;
    if (ctx->lds_rank_in_lane_order) {
        bool ok = false;
        PLONK_TRY(selftest_real_kernels(ctx, &ok));
        ctx->lds_rank_in_lane_order = ok;
    }
    return PLONK_HIP_OK;
}

int sort_pairs_passes(int end_bit) { return end_bit <= 0 ? 0 : (end_bit + 7) / 8; }

// Key bits of the first pass when it is made from the scalars: as few as leave the later passes at most eight
// bits each (small tiles: few bins keep the runs written out long), at least six.
static int fused_first_bits(int end_bit) {
    const int later = sort_pairs_passes(end_bit) - 1;
    return std::max(6, end_bit - 8 * later);
}

bool sort_pairs_of_scalars_applies(size_t n, int c, int nwin) {
    const int end_bit = c - 1;  // the bucket bits of a pair's key
    return n > 0 && n % kDigitTile == 0 && end_bit >= 1 && fused_first_bits(end_bit) <= 8 &&
           (size_t)nwin * (1u << fused_first_bits(end_bit)) * 4 <= 48 * 1024 && n * (size_t)nwin < 0xffffffffull;
}

size_t sort_pairs_temp_bytes(size_t n, int end_bit) {
    const size_t ntiles = (n + kTile - 1) / kTile, nchunks = (ntiles + kChunkTiles - 1) / kChunkTiles;
    size_t words = (ntiles + nchunks) * kMaxBins;
    // first pass from the scalars: smaller tiles, rows of 2^bits counts
    const size_t bins = (size_t)1 << std::min(8, fused_first_bits(end_bit));
    const size_t dtiles = (n + kDigitTile - 1) / kDigitTile, dchunks = dtiles / kGroupTiles + 1;
    words = std::max(words, (dtiles + dchunks) * bins);
    return words * sizeof(uint32_t) + 256;
}

// Sorts the pairs by the low end_bit key bits.  The passes alternate between the two buffer pairs, starting
// from (keys_a, vals_a): the result is in (keys_b, vals_b) after an odd number of passes (sort_pairs_passes),
// in (keys_a, vals_a) after an even one.  The key bits spread evenly over the passes.
// Passes first_pass .. npasses - 1 over pairs of whole-tile lists; key bits [shift, end_bit) spread evenly.
static void later_passes(plonk_hip_ctx *ctx, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b,
                         size_t n, int shift, int end_bit, int first_pass, int npasses, void *temp) {
    const uint32_t ntiles = cdiv(n, kTile), nchunks = cdiv(ntiles, kChunkTiles);
    uint32_t *hist = (uint32_t *)temp, *chunk_total = hist + (size_t)ntiles * kMaxBins;
    for (int p = first_pass; p < npasses; p++) {
        const int bits = (end_bit - shift + (npasses - p) - 1) / (npasses - p);
        const uint32_t *kin = (p & 1) ? keys_b : keys_a, *vin = (p & 1) ? vals_b : vals_a;
        uint32_t *kout = (p & 1) ? keys_a : keys_b, *vout = (p & 1) ? vals_a : vals_b;
        hipLaunchKernelGGL(k_radix_hist, dim3(ntiles), dim3(kSortThreads), 0, ctx->stream, kin, n, (uint32_t)shift,
                           (1u << bits) - 1, hist);
        hipLaunchKernelGGL(k_radix_scan_tiles, dim3(nchunks), dim3(kScanThreads), 0, ctx->stream, hist, ntiles, (uint32_t)kMaxBins,
                           chunk_total);
        hipLaunchKernelGGL(k_radix_scan_chunks, dim3(1), dim3(kScanThreads), 0, ctx->stream, chunk_total, nchunks,
                           (uint32_t)kMaxBins);
        hipLaunchKernelGGL(ctx->lds_rank_in_lane_order ? k_radix_scatter<true> : k_radix_scatter<false>, dim3(ntiles),
                           dim3(kSortThreads), 0, ctx->stream, kin, vin, n, (uint32_t)shift, (uint32_t)bits, hist,
                           chunk_total, kout, vout);
        shift += bits;
    }
}

int sort_pairs(plonk_hip_ctx *ctx, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, size_t n,
               int end_bit, void *temp, size_t temp_bytes) {
    if (n == 0) return PLONK_HIP_OK;
    if (n >= 0xffffffffull || temp_bytes < sort_pairs_temp_bytes(n, end_bit)) return PLONK_HIP_ERR_INVALID_ARG;
    later_passes(ctx, keys_a, keys_b, vals_a, vals_b, n, 0, end_bit, 0, sort_pairs_passes(end_bit), temp);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

// The sorted (bucket, entry) pairs of n scalars (sort_pairs_of_scalars_applies), in the buffers sort_pairs would
// leave them in; the first pass writes to (keys_b, vals_b) what it makes of the scalars.
int sort_pairs_of_scalars(plonk_hip_ctx *ctx, const Fr *d_scalars, size_t n, size_t srs_n, size_t first, int c, int nwin,
                          uint32_t nbuckets, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b,
                          void *temp, size_t temp_bytes) {
    const size_t total = n * (size_t)nwin;
    const int end_bit = c - 1;
    if (!sort_pairs_of_scalars_applies(n, c, nwin) || temp_bytes < sort_pairs_temp_bytes(total, end_bit))
        return PLONK_HIP_ERR_INVALID_ARG;
    const int bits = fused_first_bits(end_bit), npasses = sort_pairs_passes(end_bit);
    const uint32_t bins = 1u << bits, groups = kScanThreads / bins, chunk_tiles = kGroupTiles * groups;
    const uint32_t blocks = (uint32_t)(n / kDigitTile), ntiles = blocks * (uint32_t)nwin, nchunks = cdiv(ntiles, chunk_tiles);
    uint32_t *hist = (uint32_t *)temp, *chunk_total = hist + (size_t)ntiles * bins;
    hipLaunchKernelGGL(k_digits_hist, dim3(blocks), dim3(kDigitWaves * 64), (size_t)nwin * bins * 4, ctx->stream, d_scalars,
                       (uint32_t)n, c, nwin, nbuckets, bins, hist);
    hipLaunchKernelGGL(k_radix_scan_tiles, dim3(nchunks), dim3(kScanThreads), 0, ctx->stream, hist, ntiles, bins, chunk_total);
    hipLaunchKernelGGL(k_radix_scan_chunks, dim3(1), dim3(kScanThreads), 0, ctx->stream, chunk_total, nchunks, bins);
    hipLaunchKernelGGL(ctx->lds_rank_in_lane_order ? k_digits_scatter<true> : k_digits_scatter<false>, dim3(blocks),
                       dim3(kDigitWaves * 64), 0, ctx->stream, d_scalars, (uint32_t)n,
                       (uint32_t)srs_n, (uint32_t)first, c, nwin, nbuckets, (uint32_t)bits, hist, chunk_total, chunk_tiles,
                       keys_b, vals_b);
    later_passes(ctx, keys_a, keys_b, vals_a, vals_b, total, std::min(bits, end_bit), end_bit, 1, npasses, temp);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

}  // namespace plonk
