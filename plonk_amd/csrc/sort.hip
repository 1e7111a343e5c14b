// Key/value radix sort of the (bucket, entry) pairs of the MSM: least significant digit first, up to
// eight key bits per pass, every pass stable.
//
// A pass over the pairs is three launches:
//   k_radix_hist     one block per tile of kTile pairs: how many keys of the tile fall into each bin
//   k_radix_scan_*   exclusive prefix of those counts over (bin, tile) in bin-major order: where the
//                    keys of bin b of tile t go (two levels: chunks of tiles, then the chunk totals)
//   k_radix_scatter  one block per tile again: a stable rank for every key inside its tile (wave
//                    level: the lanes holding the same digit are found with one ballot per key bit, a
//                    counter per wave and bin in LDS carries the rank from round to round), the tile
//                    reordered by bin in LDS, then written out in runs of consecutive addresses
// The pairs cross HBM twice per pass (plus the keys once more for the counts); with 2^13 pairs per
// tile and 2^8 bins a run is 128 bytes for uniformly distributed digits.  Skewed inputs (all keys
// equal) only make the runs longer.
#include <cstring>

#include "context.hpp"

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

// hist[tile][bin] = number of keys of the tile whose digit (key >> shift) & mask is bin
__global__ __launch_bounds__(kSortThreads) void k_radix_hist(const uint32_t *__restrict__ keys, size_t n, uint32_t shift,
                                                            uint32_t mask, uint32_t *__restrict__ hist) {
    __shared__ uint32_t count[kMaxBins];
    if (threadIdx.x < kMaxBins) count[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kTile;
    if (base + kTile <= n) {
        // whole tile: 16-byte loads, four keys per lane and load
        const uint4 *src = reinterpret_cast<const uint4 *>(keys + base);
#pragma unroll
        for (int r = 0; r < kSortItems / 4; r++) {
            const uint4 k = src[r * kSortThreads + threadIdx.x];
            atomicAdd(&count[(k.x >> shift) & mask], 1u);
            atomicAdd(&count[(k.y >> shift) & mask], 1u);
            atomicAdd(&count[(k.z >> shift) & mask], 1u);
            atomicAdd(&count[(k.w >> shift) & mask], 1u);
        }
    } else {
        for (size_t i = base + threadIdx.x; i < n; i += kSortThreads) atomicAdd(&count[(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kMaxBins) hist[(size_t)blockIdx.x * kMaxBins + threadIdx.x] = count[threadIdx.x];
}

// Level 1: one block per chunk of kChunkTiles tiles, lane b owns bin b: hist[tile][b] becomes the number of
// keys of bin b in the earlier tiles of the chunk; chunk_total[chunk][b] their number in the whole chunk.
__global__ __launch_bounds__(kMaxBins) void k_radix_scan_tiles(uint32_t *__restrict__ hist, uint32_t ntiles,
                                                              uint32_t *__restrict__ chunk_total) {
    const uint32_t b = threadIdx.x, first = blockIdx.x * kChunkTiles;
    const uint32_t last = first + kChunkTiles < ntiles ? first + kChunkTiles : ntiles;
    uint32_t run = 0;
#pragma unroll 8
    for (uint32_t t = first; t < last; t++) {
        uint32_t *p = hist + (size_t)t * kMaxBins + b;
        const uint32_t c = *p;
        *p = run;
        run += c;
    }
    chunk_total[(size_t)blockIdx.x * kMaxBins + b] = run;
}

// Level 2: one block, lane b owns bin b: chunk_total[chunk][b] becomes the position of the first key of bin b
// of that chunk in the sorted order (all keys of smaller bins, then the earlier chunks' keys of bin b).
__global__ __launch_bounds__(kMaxBins) void k_radix_scan_chunks(uint32_t *__restrict__ chunk_total, uint32_t nchunks) {
    __shared__ uint32_t bin_total[kMaxBins];
    const uint32_t b = threadIdx.x;
    uint32_t run = 0;
#pragma unroll 8
    for (uint32_t c = 0; c < nchunks; c++) {
        uint32_t *p = chunk_total + (size_t)c * kMaxBins + b;
        const uint32_t v = *p;
        *p = run;
        run += v;
    }
    bin_total[b] = run;
    __syncthreads();
    // exclusive prefix over the bins (256 values: a serial pass by every lane over the values below its own
    // would do; a logarithmic scan keeps it short)
    for (uint32_t off = 1; off < kMaxBins; off <<= 1) {
        const uint32_t v = b >= off ? bin_total[b - off] : 0u;
        __syncthreads();
        bin_total[b] += v;
        __syncthreads();
    }
    const uint32_t base = bin_total[b] - run;  // inclusive - own = exclusive
#pragma unroll 8
    for (uint32_t c = 0; c < nchunks; c++) chunk_total[(size_t)c * kMaxBins + b] += base;
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

// Stable scatter of one tile.  Element e of the tile (global position tile * kTile + e) belongs to wave
// e / kWaveSpan, round (e % kWaveSpan) / 64, lane e % 64: ranks follow that order.
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(const uint32_t *__restrict__ keys_in,
                                                               const uint32_t *__restrict__ vals_in, size_t n,
                                                               uint32_t shift, uint32_t bits,
                                                               const uint32_t *__restrict__ hist,
                                                               const uint32_t *__restrict__ chunk_base,
                                                               uint32_t *__restrict__ keys_out,
                                                               uint32_t *__restrict__ vals_out) {
    __shared__ uint32_t wave_count[kSortWaves][kMaxBins];  // per wave and bin: running count, then start position
    __shared__ uint32_t bin_start[kMaxBins];               // first position of the bin in the reordered tile
    __shared__ uint32_t bin_dest[kMaxBins];                // where that position goes in the output
    __shared__ uint32_t skey[kTile], sval[kTile];
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t mask = (1u << bits) - 1;
    const size_t tile_base = (size_t)blockIdx.x * kTile;
    const uint32_t tile_count = (uint32_t)(n - tile_base < (size_t)kTile ? n - tile_base : (size_t)kTile);
    for (uint32_t i = t; i < kSortWaves * kMaxBins; i += kSortThreads) (&wave_count[0][0])[i] = 0;
    __syncthreads();

    uint32_t key[kSortItems], val[kSortItems], rank[kSortItems];
    const uint32_t wave_first = w * kWaveSpan;
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const uint32_t e = wave_first + r * 64 + lane;
        const bool valid = e < tile_count;
        key[r] = valid ? keys_in[tile_base + e] : 0u;
        val[r] = valid ? vals_in[tile_base + e] : 0u;
    }
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const uint32_t e = wave_first + r * 64 + lane;
        const bool valid = e < tile_count;
        const uint32_t d = (key[r] >> shift) & mask;
        const uint64_t same = same_digit_lanes(d, bits, valid);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
        uint32_t prior = 0;
        if (valid) {
            prior = wave_count[w][d];
            // the wave's LDS operations complete in program order: every lane of the group has read the count
            // before its first lane replaces it
            if (below == 0) wave_count[w][d] = prior + (uint32_t)__popcll(same);
        }
        rank[r] = prior + below;
    }
    __syncthreads();
    // per bin: start in the reordered tile, start of every wave's share, destination in the output
    // (lane b < kMaxBins owns bin b; the other lanes only keep the barriers company)
    {
        const uint32_t b = t & (kMaxBins - 1);
        const bool owner = t < kMaxBins;
        uint32_t c[kSortWaves], total = 0;
#pragma unroll
        for (int i = 0; i < kSortWaves; i++) {
            c[i] = wave_count[i][b];
            total += c[i];
        }
        if (owner) bin_start[b] = total;
        __syncthreads();
        for (uint32_t off = 1; off < kMaxBins; off <<= 1) {
            const uint32_t v = b >= off ? bin_start[b - off] : 0u;
            __syncthreads();
            if (owner) bin_start[b] += v;
            __syncthreads();
        }
        const uint32_t start = bin_start[b] - total;
        __syncthreads();
        if (owner) {
            bin_start[b] = start;
            uint32_t run = start;
#pragma unroll
            for (int i = 0; i < kSortWaves; i++) {
                wave_count[i][b] = run;
                run += c[i];
            }
            bin_dest[b] = chunk_base[(size_t)(blockIdx.x / kChunkTiles) * kMaxBins + b] + hist[(size_t)blockIdx.x * kMaxBins + b];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const uint32_t e = wave_first + r * 64 + lane;
        if (e < tile_count) {
            const uint32_t d = (key[r] >> shift) & mask;
            const uint32_t pos = wave_count[w][d] + rank[r];
            skey[pos] = key[r];
            sval[pos] = val[r];
        }
    }
    __syncthreads();
    for (uint32_t i = t; i < tile_count; i += kSortThreads) {
        const uint32_t k = skey[i];
        const uint32_t d = (k >> shift) & mask;
        const size_t dest = (size_t)bin_dest[d] + (i - bin_start[d]);
        keys_out[dest] = k;
        vals_out[dest] = sval[i];
    }
}

int sort_pairs_passes(int end_bit) { return end_bit <= 0 ? 0 : (end_bit + 7) / 8; }

size_t sort_pairs_temp_bytes(size_t n, int end_bit) {
    (void)end_bit;
    const size_t ntiles = (n + kTile - 1) / kTile, nchunks = (ntiles + kChunkTiles - 1) / kChunkTiles;
    return (ntiles + nchunks) * kMaxBins * sizeof(uint32_t) + 256;
}

// Sorts the pairs by the low end_bit key bits.  The passes alternate between the two buffer pairs, starting
// from (keys_a, vals_a): the result is in (keys_b, vals_b) after an odd number of passes (sort_pairs_passes),
// in (keys_a, vals_a) after an even one.
// The first pass sorts by the lowest key bits (the key bits spread evenly over the passes, see sort_pairs).
SortFirstPass sort_pairs_first_pass(int end_bit, void *temp) {
    const int npasses = sort_pairs_passes(end_bit);
    const int bits = npasses ? (end_bit + npasses - 1) / npasses : 0;
    return SortFirstPass{(uint32_t)kTile, (uint32_t)kMaxBins, (1u << bits) - 1, (uint32_t *)temp};
}

int sort_pairs(plonk_hip_ctx *ctx, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, size_t n,
               int end_bit, void *temp, size_t temp_bytes, bool first_counts_done) {
    if (n == 0) return PLONK_HIP_OK;
    if (n >= 0xffffffffull || temp_bytes < sort_pairs_temp_bytes(n, end_bit)) return PLONK_HIP_ERR_INVALID_ARG;
    const uint32_t ntiles = cdiv(n, kTile), nchunks = cdiv(ntiles, kChunkTiles);
    uint32_t *hist = (uint32_t *)temp, *chunk_total = hist + (size_t)ntiles * kMaxBins;
    const int npasses = sort_pairs_passes(end_bit);
    int shift = 0;
    for (int p = 0; p < npasses; p++) {
        // the key bits spread evenly over the passes
        const int bits = (end_bit - shift + (npasses - p) - 1) / (npasses - p);
        const uint32_t *kin = (p & 1) ? keys_b : keys_a, *vin = (p & 1) ? vals_b : vals_a;
        uint32_t *kout = (p & 1) ? keys_a : keys_b, *vout = (p & 1) ? vals_a : vals_b;
        if (p > 0 || !first_counts_done) {
            hipLaunchKernelGGL(k_radix_hist, dim3(ntiles), dim3(kSortThreads), 0, ctx->stream, kin, n, (uint32_t)shift,
                               (1u << bits) - 1, hist);
        }
        hipLaunchKernelGGL(k_radix_scan_tiles, dim3(nchunks), dim3(kMaxBins), 0, ctx->stream, hist, ntiles, chunk_total);
        hipLaunchKernelGGL(k_radix_scan_chunks, dim3(1), dim3(kMaxBins), 0, ctx->stream, chunk_total, nchunks);
        hipLaunchKernelGGL(k_radix_scatter, dim3(ntiles), dim3(kSortThreads), 0, ctx->stream, kin, vin, n,
                           (uint32_t)shift, (uint32_t)bits, hist, chunk_total, kout, vout);
        shift += bits;
    }
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

}  // namespace plonk
