// G1 multi-scalar multiplication behind kzg_commit / kzg_evaluate: signed-digit bucket method
// (Pippenger) over resident window tables.
//
// The SRS is fixed, so the context keeps [2^(c*w)] P_i for every window w (srs.hip).  All
// windows then share ONE set of 2^(c-1) buckets and no window is ever shifted by doublings:
//     sum_i k_i P_i = sum_b (b+1) * B_b,   B_b = sum over (i, w) with |digit_w(k_i)| = b+1
//                                               of sign * table[w][i].
// Pipeline (all on the context's stream, no host synchronisation):
//   1. k_digits      scalars out of Montgomery form, nwin signed digits each -> (bucket, entry)
//   2. sort_pairs    radix sort of the n*nwin pairs by bucket (sort.hip); large inputs: steps 1 and 2 as
//                    sort_pairs_of_scalars, whose first pass takes its keys from the scalars themselves
//   3. k_bounds      first sorted position of every bucket
//   4. k_accumulate  one lane per fixed-length SEGMENT of the sorted list: gathers the affine
//                    table points (one 128-byte line each) and adds them into an XYZZ
//                    accumulator (8M + 2S per point) on 28-bit limbs with lazy reduction
//                    (field28.cuh, g1_28.cuh).
//                    Buckets lying wholly inside a segment are written directly; runs cut by a
//                    segment border are written as partials.  Work per lane is uniform whatever
//                    the scalar distribution.
//   5. k_fixup       one lane per bucket that spans several segments: sums its partials
//   6. k_reduce_*    sum_b (b+1) B_b by recursive 4-bucket chunks:
//                    S(A, off) = sum_t LW[t] + 4 * S(R, 0) with per-chunk weighted sums LW and
//                    plain sums R, LDS tree sums of every level's LW, Horner over the levels.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "context.hpp"
#define PLONK_MONT28_ASM 1  // the lazy Fq products as hand-scheduled sequences
#include "coop28.cuh"
#include "g1_28.cuh"
#include "msm_core.cuh"

namespace plonk {

static constexpr int kBlock = 128;

// Out-of-line doublings: one copy of each per code object.  They are taken only when two equal
// points meet in one bucket.  Arguments by index / value and results by value, so that no
// caller variable has its address taken (an address-taken accumulator would live in scratch
// memory for the whole loop).
__device__ __noinline__ G1Xyzz28 dbl_entry_cold(const G1Affine28 *__restrict__ table, uint32_t v) {
    G1Affine28 pt = table[v & 0x7fffffffu];
    Fq28 y = (v >> 31) ? q28_neg(pt.y) : pt.y;
    return xyzz28_dbl_affine(pt.x, y);
}
__device__ __noinline__ G1Xyzz28 dbl_cold(const G1Xyzz28 *a) { return xyzz28_dbl(*a); }
// The accumulation loop's rare case (the incoming point equals or is opposite to the
// accumulator): redo the addition on canonical values with the complete formulas.
__device__ __noinline__ G1Xyzz28 madd_slow_cold(const G1Xyzz28 *acc_lazy, const G1Affine28 *__restrict__ table,
                                                uint32_t v) {
    G1Xyzz28 acc = xyzz28_canon(*acc_lazy);
    xyzz28_madd(acc, table[v & 0x7fffffffu], (v >> 31) != 0);
    return acc;
}

// ------------------------------------------------------------------------------------------
// 1. digits
// ------------------------------------------------------------------------------------------
// scalar i multiplies SRS point first + i
__global__ __launch_bounds__(256) void k_digits(const Fr *__restrict__ scalars, uint32_t n, uint32_t srs_n,
                                               uint32_t first, int c, int nwin, uint32_t nbuckets,
                                               uint32_t *__restrict__ keys, uint32_t *__restrict__ vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr k = fe_from_mont(scalars[i]);
    // digits of r - k where the width asks for it: every entry then takes the opposite point
    const bool opposite = msm_folds_scalars(c) && scalar_fold(k);
    uint32_t carry = 0;
    for (int w = 0; w < nwin; w++) {
        int32_t d = signed_digit(k, w, c, carry);
        uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
        // digit 0 contributes nothing: key nbuckets, which the sort (bucket bits only) files under bucket 0 and the
        // accumulation passes over
        uint32_t key = mag == 0 ? nbuckets : mag - 1;
        uint32_t val = ((uint32_t)w * srs_n + first + i) | (((d < 0) != opposite) ? 0x80000000u : 0u);
        size_t at = (size_t)w * n + i;
        keys[at] = key;
        vals[at] = val;
    }
}

// ------------------------------------------------------------------------------------------
// 3. bucket bounds: start[b] = first sorted position whose bucket (key mod nbuckets) is >= b, b = 0..nbuckets;
//    start[nbuckets] = start[nbuckets + 1] = total.
// ------------------------------------------------------------------------------------------
// One lane per bucket, a binary search each: the cost does not depend on how the keys are spread (a lane
// per sorted position that fills in the buckets between its key and its neighbour's would loop over all
// buckets when a short input meets a wide window plan).
__global__ __launch_bounds__(256) void k_bounds(const uint32_t *__restrict__ keys, size_t total, uint32_t nbuckets,
                                               uint32_t *__restrict__ start) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nbuckets + 1) return;
    uint32_t lo = 0, hi = (uint32_t)total;  // first position with key >= b lies in [lo, hi]
    if (b > nbuckets) lo = hi;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((keys[mid] & (nbuckets - 1)) < b) lo = mid + 1; else hi = mid;
    }
    start[b] = lo;
}

// ------------------------------------------------------------------------------------------
// 4. accumulate
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void flush_run(uint32_t key, uint32_t run_begin, uint32_t run_end, uint32_t seg_begin,
                                          size_t seg, const uint32_t *__restrict__ start,
                                          G1Xyzz28 *__restrict__ buckets, G1Xyzz28 *__restrict__ part,
                                          const G1Xyzz28 &acc) {
    if (start[key] == run_begin && start[key + 1] == run_end) {
        buckets[key] = acc;  // the whole bucket lies inside this segment
    } else if (run_begin == seg_begin) {
        part[2 * seg] = acc;  // run touches the segment's first position
    } else {
        part[2 * seg + 1] = acc;  // run touches only the segment's last position
    }
}

// Whole-line gather through LDS.  A lane reading its own 128-byte table line with eight 16-byte
// loads makes every wave-load touch 64 different lines; instead the wave fetches its 64 lines with
// eight loads that each cover eight whole lines (eight lanes per line), written straight into the
// wave's LDS buffer, and every lane then picks up its own line from there.  Piece s of the line of
// lane j is stored at slot 8j + (s ^ (j & 7)) so that the read-back spreads over the banks.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void global_void_t;

__device__ __forceinline__ void issue_lines(const G1Affine28 *__restrict__ table, uint4 *buf, uint32_t lane,
                                            uint32_t my_index) {
#pragma unroll
    for (uint32_t g = 0; g < 8; g++) {
        const uint32_t owner = 8 * g + (lane >> 3);
        const uint32_t idx = (uint32_t)__shfl((int)my_index, (int)owner, 64);
        const uint32_t piece = (lane & 7) ^ (owner & 7);
        const uint4 *src = reinterpret_cast<const uint4 *>(table + idx) + piece;
        // destination: wave-uniform base; the hardware adds lane * 16
        __builtin_amdgcn_global_load_lds((global_void_t *)src, (lds_void_t *)(buf + 64 * g), 16, 0, 0);
    }
}

__device__ __forceinline__ G1Affine28 read_line(const uint4 *buf, uint32_t lane) {
    G1Affine28 pt;
    uint4 w[8];
#pragma unroll
    for (uint32_t s = 0; s < 8; s++) w[s] = buf[8 * lane + (s ^ (lane & 7))];
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        const uint4 &a = w[i >> 2], &b = w[4 + (i >> 2)];
        pt.x.l[i] = (i & 3) == 0 ? a.x : (i & 3) == 1 ? a.y : (i & 3) == 2 ? a.z : a.w;
        pt.y.l[i] = (i & 3) == 0 ? b.x : (i & 3) == 1 ? b.y : (i & 3) == 2 ? b.z : b.w;
    }
    return pt;
}

// Two waves per SIMD: the whole addition then lives in registers.  A third wave (168 registers)
// spills about 60 bytes per lane and iteration, which showed up as 13 GB of write traffic per
// 2^24-point launch and cost more than the extra wave hid.
// LDS_PAIRS (what the library uses): a lane's (bucket, entry) pairs come in kGroup positions at a time (32 bytes of
// each list, on 16-byte boundaries) by direct global-to-LDS loads into a buffer of the wave in which every lane has
// a place of its own; the loop reads one pair per iteration from there.  Without it (kept for measurements) a
// lane fetches its pairs one position at a time, two positions ahead of the addition: the lanes of a wave then
// walk 64 different 128-byte lines of each list for 32 iterations, and the lines are fetched from memory about
// thirteen times over (50 GB instead of 30 GB per 2^24-point launch) for the same speed.
// The block's LDS stays at 24 KiB: from 32 KiB per block of two waves on, the kernel measured 5 % slower whatever
// its loads looked like (which is what made groups of sixteen positions look slower than single loads).
static constexpr uint32_t kGroup = 8;  // positions per group (a multiple of four)

template <bool LDS_PAIRS>
__global__ __launch_bounds__(kBlock, 2) void k_accumulate(const uint32_t *__restrict__ keys,
                                                       const uint32_t *__restrict__ vals,
                                                       const uint32_t *__restrict__ start, uint32_t nbuckets,
                                                       uint32_t seg_len, const G1Affine28 *__restrict__ table,
                                                       G1Xyzz28 *__restrict__ buckets,
                                                       G1Xyzz28 *__restrict__ part, uint32_t *status) {
    __shared__ uint4 line_buf[kBlock / 64][64 * 8];                           // 8 KiB per wave
    __shared__ uint4 pair_buf[LDS_PAIRS ? kBlock / 64 : 1][2][kGroup / 4][64];  // 4 KiB per wave
    const uint32_t lane = threadIdx.x & 63;
    uint4 *buf = line_buf[threadIdx.x >> 6];
    uint4(*pbuf)[kGroup / 4][64] = pair_buf[LDS_PAIRS ? threadIdx.x >> 6 : 0];
    const size_t seg = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nvalid = start[nbuckets];
    const size_t begin64 = seg * seg_len;
    const bool live = begin64 < nvalid;
    if (__ballot(live) == 0) return;  // the whole wave lies behind the list
    // lanes without a segment stay in the loop: they carry part of the other lanes' line loads
    const uint32_t begin = live ? (uint32_t)begin64 : 0u;
    const size_t end64 = begin64 + seg_len;
    const uint32_t end = live ? (end64 < nvalid ? (uint32_t)end64 : nvalid) : 0u;

    // pair at position begin + o (zeros behind the segment's end), straight from the lists
    auto pair_direct = [&](uint32_t o, uint32_t &key, uint32_t &val) {
        key = 0;
        val = 0;
        if (begin + o < end) {
            key = keys[begin + o];
            val = vals[begin + o];
        }
    };
    // Group g of a lane: the kGroup positions from base + kGroup g on, base = begin rounded down to a multiple of
    // four (the lists have sixteen positions of slack behind their end).  The wave's buffer holds one group per lane.
    const uint32_t base = begin & ~3u;
    auto group_issue = [&](uint32_t g) {
        const uint32_t first = base + kGroup * g;
        if (first < end) {
#pragma unroll
            for (uint32_t q = 0; q < kGroup / 4; q++) {
                // destination: wave-uniform base; the hardware adds lane * 16
                __builtin_amdgcn_global_load_lds((global_void_t *)(keys + first + 4 * q), (lds_void_t *)&pbuf[0][q][0], 16, 0, 0);
                __builtin_amdgcn_global_load_lds((global_void_t *)(vals + first + 4 * q), (lds_void_t *)&pbuf[1][q][0], 16, 0, 0);
            }
        }
    };
    auto pair_at = [&](uint32_t o, uint32_t &key, uint32_t &val) {
        if (!LDS_PAIRS) {
            pair_direct(o, key, val);
            return;
        }
        key = 0;
        val = 0;
        if (begin + o < end) {
            const uint32_t j = begin + o - base;
            key = reinterpret_cast<const uint32_t *>(&pbuf[0][(j >> 2) % (kGroup / 4)][lane])[j & 3];
            val = reinterpret_cast<const uint32_t *>(&pbuf[1][(j >> 2) % (kGroup / 4)][lane])[j & 3];
        }
    };

    // (bucket, entry) of the current position and of the next one; the line of the current
    // position is always on its way to (or in) the wave's buffer
    uint32_t k, v, kn, vn;
    pair_direct(0, k, v);
    pair_direct(1, kn, vn);
    if (LDS_PAIRS) group_issue((begin + 2 - base) / kGroup);  // the group of the first position the loop reads
    // (a pair to pass over fetches line 0 instead of its entry's: many of them then cost no memory traffic)
    issue_lines(table, buf, lane, k < nbuckets ? v & 0x7fffffffu : 0u);
    const uint32_t bucket_mask = nbuckets - 1;  // a key is a bucket, or nbuckets: a pair of bucket 0 to pass over
    uint32_t cur = k & bucket_mask;
    uint32_t run_begin = begin;
    G1Xyzz28 acc = xyzz28_identity();
    // The bucket bounds are only meaningful for keys in ascending order; a sort that lost it (see sort.hip on
    // what its ranks rely on) is reported, not trusted: every change of key is checked, and the segment's first
    // key against the one before it.
    bool out_of_order = live && begin < end && ((begin > 0 && (keys[begin - 1] & bucket_mask) > cur) || k > nbuckets);
    for (uint32_t it = 0; it < seg_len; it++) {  // the same trip count for every lane of the grid
        const uint32_t p = begin + it;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this position's lines have landed
        G1Affine28 pt = read_line(buf, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every lane holds its line: the buffer is free
        issue_lines(table, buf, lane, kn < nbuckets ? vn & 0x7fffffffu : 0u);
        uint32_t k2, v2;
        pair_at(it + 2, k2, v2);
        if (LDS_PAIRS) {
            // the next position opens a new group: this lane's group is used up, fetch the next one into its place
            // (it lands with the wait at the top of the loop)
            const uint32_t jn = begin + it + 3 - base;
            if (jn % kGroup == 0 && it + 3 < seg_len) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                group_issue(jn / kGroup);
            }
        }
        if (p < end) {
            const uint32_t kb = k & bucket_mask;
            out_of_order = out_of_order || k > nbuckets;
            if (kb != cur) {
                out_of_order = out_of_order || kb < cur;
                flush_run(cur, run_begin, p, begin, seg, start, buckets, part, acc);
                cur = kb;
                run_begin = p;
                acc = xyzz28_identity();
            }
            if (k < nbuckets && !affine28_is_identity(pt)) {
                const bool neg = (v >> 31) != 0;
                if (xyzz28_is_identity(acc)) {
                    acc.x = pt.x;
                    acc.y = neg ? q28_neg(pt.y) : pt.y;
                    acc.zz = q28_one();
                    acc.zzz = q28_one();
                } else if (!xyzz28_madd_lazy(acc, pt.x, pt.y, neg)) {
                    G1Xyzz28 t = acc;  // a private copy: acc itself keeps its registers
                    acc = madd_slow_cold(&t, table, v);
                }
            }
        }
        k = kn;
        v = vn;
        kn = k2;
        vn = v2;
    }
    if (live) flush_run(cur, run_begin, end, begin, seg, start, buckets, part, acc);
    if (out_of_order) __hip_atomic_store(status, kStatusKeysOutOfOrder, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------
// 5. fix-up: buckets cut by segment borders
// ------------------------------------------------------------------------------------------
// Full XYZZ additions with the (rare) equal-or-opposite-operands case out of line: one inlined
// copy per kernel keeps the serial tails short without blowing up the code object.
// Operands and result in the lazy form the accumulation leaves behind: the tree sums never
// canonicalise in between.
__device__ __noinline__ G1Xyzz28 add_slow_cold(const G1Xyzz28 *a, const G1Xyzz28 *b) {
    return xyzz28_add(xyzz28_canon(*a), xyzz28_canon(*b));
}
__device__ __forceinline__ G1Xyzz28 add_inline(const G1Xyzz28 &a, const G1Xyzz28 &b) {
    if (xyzz28_is_identity(a)) return b;
    if (xyzz28_is_identity(b)) return a;
    G1Xyzz28 r;
    if (!xyzz28_add_lazy(r, a, b)) {
        G1Xyzz28 ta = a, tb = b;  // private copies: the caller's values keep their registers
        r = add_slow_cold(&ta, &tb);
    }
    return r;
}

// Partial j (0-based) of a bucket whose run starts at sorted position lo: segment s0 + j;
// only the first partial can sit in the "touches the segment end" slot.
__device__ __forceinline__ const G1Xyzz28 *partial_of(const G1Xyzz28 *part, uint32_t s0, uint32_t j, uint32_t lo,
                                                    uint32_t seg_len) {
    size_t s = (size_t)s0 + j;
    return part + 2 * s + ((j == 0 && lo != s0 * seg_len) ? 1 : 0);
}

// ---- wave-level sums -----------------------------------------------------------------------
__device__ __forceinline__ G1Xyzz28 shfl_down_point(const G1Xyzz28 &v, int off, int width) {
    G1Xyzz28 o;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        o.x.l[i] = (uint32_t)__shfl_down((int)v.x.l[i], off, width);
        o.y.l[i] = (uint32_t)__shfl_down((int)v.y.l[i], off, width);
        o.zz.l[i] = (uint32_t)__shfl_down((int)v.zz.l[i], off, width);
        o.zzz.l[i] = (uint32_t)__shfl_down((int)v.zzz.l[i], off, width);
    }
    return o;
}
__device__ __forceinline__ G1Xyzz28 shfl_point(const G1Xyzz28 &v, int src) {
    G1Xyzz28 o;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        o.x.l[i] = (uint32_t)__shfl((int)v.x.l[i], src, 64);
        o.y.l[i] = (uint32_t)__shfl((int)v.y.l[i], src, 64);
        o.zz.l[i] = (uint32_t)__shfl((int)v.zz.l[i], src, 64);
        o.zzz.l[i] = (uint32_t)__shfl((int)v.zzz.l[i], src, 64);
    }
    return o;
}

// The serial tails below are chains of dependent point additions (about 15 microseconds each for
// a lone wave), so every sum is organised for small depth: a few strided terms per lane, then a
// shuffle tree inside the wave, then a two-level tree across the block's four waves.  All stages
// of one sum go through a single inlined addition (selected operands, one loop).
static constexpr int kTreeBlock = 256;  // largest block of the tree kernels

// Sum of `cnt` points (lazy form in, lazy form out) fetched through `at(j)` by one block of 64, 128
// or 256 lanes; valid in lane 0.  Wide blocks give the smallest depth, narrow ones the least idle
// work in the tree levels.
template <class At>
__device__ __forceinline__ G1Xyzz28 block_sum(uint32_t cnt, G1Xyzz28 *sh, At at) {
    const uint32_t t = threadIdx.x, lane = t & 63, width = blockDim.x;
    const uint32_t waves = width >> 6;                         // 1, 2 or 4
    const uint32_t wave_levels = waves == 4 ? 2 : (waves == 2 ? 1 : 0);
    const uint32_t strided = (cnt + width - 1) / width;        // uniform over the block
    G1Xyzz28 acc = xyzz28_identity();
    const uint32_t steps = strided + 6 + wave_levels;
    for (uint32_t step = 0; step < steps; step++) {
        G1Xyzz28 o = xyzz28_identity();
        if (step < strided) {
            uint32_t j = step * width + t;
            if (j < cnt) o = *at(j);
        } else if (step < strided + 6) {
            const int off = 32 >> (step - strided);
            o = shfl_down_point(acc, off, 64);
            if ((int)lane >= off) o = xyzz28_identity();
        } else {
            if (step == strided + 6) {
                if (lane == 0) sh[t >> 6] = acc;
                __syncthreads();
                acc = t < waves ? sh[t] : xyzz28_identity();
            }
            const int off = (int)(waves >> 1) >> (step - strided - 6);
            o = shfl_down_point(acc, off, 4);
            if (t >= waves || (int)t >= off) o = xyzz28_identity();
        }
        acc = add_inline(acc, o);
    }
    __syncthreads();  // sh may be reused by the caller's next sum
    return acc;
}

// ---- sums by four cooperating waves ----------------------------------------------------------
// Where only a few blocks are left to run (the per-bit sums, the final chain, the row and column
// sums of a small bucket array) a block of four waves works on 64 slots at once with the staged
// formulas of coop28.cuh: values live in LDS as [value][limb][slot] (conflict-free for a wave).
struct CoopMem {
    uint32_t v[8 + kCoopTemps][Q28::N][64];  // values 0-3: point A, 4-7: point B, 8-: temporaries
    uint32_t slow[64];
};
static constexpr int kCoopA = 0, kCoopB = 4, kCoopT = 8;

struct CoopSlot {
    CoopMem *m;
    uint32_t s, sb;  // slot of a (and of the temporaries), slot of b
    int bbase;       // kCoopB, or kCoopA when b is another slot of the same array
    __device__ __forceinline__ Fq28 ld(int val, uint32_t slot) const {
        Fq28 r;
#pragma unroll
        for (int i = 0; i < Q28::N; i++) r.l[i] = m->v[val][i][slot];
        return r;
    }
    __device__ __forceinline__ void st(int val, const Fq28 &x) {
#pragma unroll
        for (int i = 0; i < Q28::N; i++) m->v[val][i][s] = x.l[i];
    }
    __device__ __forceinline__ Fq28 a(int i) const { return ld(kCoopA + i, s); }
    __device__ __forceinline__ Fq28 b(int i) const { return ld(bbase + i, sb); }
    __device__ __forceinline__ Fq28 t(int i) const { return ld(kCoopT + i, s); }
    __device__ __forceinline__ void set_a(int i, const Fq28 &x) { st(kCoopA + i, x); }
    __device__ __forceinline__ void set_t(int i, const Fq28 &x) { st(kCoopT + i, x); }
    __device__ __forceinline__ bool slow() const { return m->slow[s] != 0; }
    __device__ __forceinline__ void set_slow(bool f) { m->slow[s] = f ? 1u : 0u; }
};

__device__ __forceinline__ void coop_store(CoopMem &m, int base, uint32_t slot, const G1Xyzz28 &p) {
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        m.v[base + 0][i][slot] = p.x.l[i];
        m.v[base + 1][i][slot] = p.y.l[i];
        m.v[base + 2][i][slot] = p.zz.l[i];
        m.v[base + 3][i][slot] = p.zzz.l[i];
    }
}
__device__ __forceinline__ G1Xyzz28 coop_load(const CoopMem &m, int base, uint32_t slot) {
    G1Xyzz28 p;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        p.x.l[i] = m.v[base + 0][i][slot];
        p.y.l[i] = m.v[base + 1][i][slot];
        p.zz.l[i] = m.v[base + 2][i][slot];
        p.zzz.l[i] = m.v[base + 3][i][slot];
    }
    return p;
}
__device__ __forceinline__ bool coop_is_identity(const CoopMem &m, int base, uint32_t slot) {
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) z |= m.v[base + 2][i][slot];
    return z == 0;
}

// A[s] += (b_self ? A : B)[s + boff] for the slots s with (s & seg_mask) < nact (s + boff < 64; seg_mask = 63: the
// slots s < nact; seg_mask = 31: the same in each half of the slots).  Called by all 256
// threads of the block after a barrier that made the operands visible; ends with a barrier.
// Whether the staged loop (which contains barriers) runs is decided by a block-wide vote that is itself
// a barrier (__syncthreads_or): all four waves take the same branch by construction, whatever each of
// them read, and nothing below writes a point before every wave has taken its decisions.
// Inlined, and every kernel calls it from ONE place (a loop over its steps): as a function of its own it saved and
// restored about a hundred registers per call through scratch memory, 7 000 of the 22 000 cycles of a call.
__device__ __forceinline__ void coop_add(CoopMem &m, bool b_self, uint32_t boff, uint32_t nact, uint32_t seg_mask = 63u) {
    const uint32_t w = threadIdx.x >> 6, s = threadIdx.x & 63;
    CoopSlot c{&m, s, (s + boff) & 63u, b_self ? kCoopA : kCoopB};
    const bool act = (s & seg_mask) < nact;
    const bool ida = act && coop_is_identity(m, kCoopA, s);
    const bool idb = !act || coop_is_identity(m, c.bbase, c.sb);
    const bool live = act && !ida && !idb;
    if (__syncthreads_or(live ? 1 : 0)) {
#pragma unroll 1
        for (int stage = 0; stage < kCoopAddStages; stage++) {
            if (live) coop_add_stage(c, stage, (int)w);
            __syncthreads();
        }
    }
    // what the stages leave to wave 0: equal or opposite points, and a = identity
    if (w == 0 && act) {
        if (live && c.slow()) {
            G1Xyzz28 pa = coop_load(m, kCoopA, s), pb = coop_load(m, c.bbase, c.sb);
            coop_store(m, kCoopA, s, add_slow_cold(&pa, &pb));
        } else if (ida && !idb) {
            coop_store(m, kCoopA, s, coop_load(m, c.bbase, c.sb));
        }
    }
    __syncthreads();
}

// A[s] <- 2 A[s] in the slots that `want` it; same calling rules as coop_add.
__device__ __forceinline__ void coop_dbl(CoopMem &m, bool want) {
    const uint32_t w = threadIdx.x >> 6, s = threadIdx.x & 63;
    CoopSlot c{&m, s, s, kCoopA};
    const bool live = want && !coop_is_identity(m, kCoopA, s);
    if (__syncthreads_or(live ? 1 : 0)) {
#pragma unroll 1
        for (int stage = 0; stage < kCoopDblStages; stage++) {
            if (live) coop_dbl_stage(c, stage, (int)w);
            __syncthreads();
        }
    }
    __syncthreads();
}

// Sum of `cnt` points fetched through `at(j)` by a block of exactly 256 threads; valid in thread 0.
// Every thread first adds up its own strided share; the 256 partial sums are then joined by
// cooperative additions: three rounds bring the four waves' values into the 64 slots, six more
// halve the slots.
template <class At>
__device__ __forceinline__ G1Xyzz28 block_sum_coop(uint32_t cnt, CoopMem &m, At at) {
    const uint32_t t = threadIdx.x, w = t >> 6, s = t & 63;
    G1Xyzz28 acc = xyzz28_identity();
    for (uint32_t j = t; j < cnt; j += 256) acc = add_inline(acc, *at(j));
    if (w == 0) coop_store(m, kCoopA, s, acc);
    // steps 0-2: the values of waves 1-3 join the 64 slots; steps 3-8: the slots are halved
#pragma unroll 1
    for (uint32_t step = 0; step < 9; step++) {
        const bool join = step < 3;
        const uint32_t off = join ? 0u : 32u >> (step - 3);
        if (join ? cnt <= 64 * (step + 1) : cnt <= off) continue;  // nothing there (the same for every thread)
        if (join && w == step + 1) coop_store(m, kCoopB, s, acc);
        __syncthreads();
        coop_add(m, !join, off, join ? 64u : off);
    }
    __syncthreads();
    G1Xyzz28 r = coop_load(m, kCoopA, 0);
    __syncthreads();  // the caller may start the next sum
    return r;
}

// Two sums of exactly 256 points each by one block of 256 threads; the first is valid in slot 0, the second in
// slot 32 of the A values (coop_load).  With more sums than compute units a block per sum runs in two rounds of
// nine additions, six of them on 32, 16, ... 1 of the 64 slots; two sums side by side share the slots instead:
// slots 0-31 belong to the first sum, 32-63 to the second, every thread brings in two points of its half's sum
// (eight loads of 64 slots: seven additions), and five halvings work on both halves at once: twelve additions
// per pair of sums instead of eighteen.
template <class At0, class At1>
__device__ __forceinline__ void block_sum_coop_pair(CoopMem &m, At0 at0, At1 at1) {
    const uint32_t t = threadIdx.x, w = t >> 6, s = t & 63, e = s & 31;
    // point (4 r + w) * 32 + e of this half's sum goes in at step 4 r + w: a thread's second point is fetched
    // while the other waves take their turns
    auto fetch = [&](uint32_t r) { return s < 32 ? *at0((4 * r + w) * 32 + e) : *at1((4 * r + w) * 32 + e); };
    G1Xyzz28 mine = fetch(0);
    // steps 0-7: the eight loads (the first fills the A values, the others are added to them); steps 8-12: halvings
#pragma unroll 1
    for (uint32_t step = 0; step < 13; step++) {
        const bool load = step < 8;
        if (load && w == (step & 3)) {
            coop_store(m, step == 0 ? kCoopA : kCoopB, s, mine);
            if (step < 4) mine = fetch(1);
        }
        __syncthreads();
        if (step == 0) continue;
        const uint32_t off = load ? 0u : 16u >> (step - 8);
        coop_add(m, !load, off, load ? 64u : off, load ? 63u : 31u);
    }
}

// Buckets cut into more partials than a lane group handles go to the block path, in chunks of chunk_parts
// partials (chosen by the host so that all chunks of the worst case run at once: the strided part of a block's
// sum keeps every lane busy, its tree does not): one block sums a chunk, and a bucket of several chunks gets one
// more sum over its chunk results (k_fixup_join).  A bucket that holds a whole window's digits (every scalar equal) is cut into tens of
// thousands of partials: by chunks its sum is two short trees spread over the GPU instead of one block's long
// chain.  The queue is built before the accumulation (it depends only on the bucket bounds).
// Queue words: [0] work items, [1] joins, [2] chunk slots handed out, [3] unused; max_joins x {bucket, first
// slot, chunks}; then the items {bucket, chunk, slot or kDirectSlot}.
static constexpr uint32_t kDirectSlot = 0xffffffffu, kQueueHead = 4;

__global__ __launch_bounds__(256) void k_classify(const uint32_t *__restrict__ start, uint32_t nbuckets,
                                                 uint32_t seg_len, uint32_t group_cap, uint32_t chunk_parts,
                                                 uint32_t *__restrict__ queue,
                                                 uint32_t max_joins, uint32_t max_items, uint32_t max_slots,
                                                 uint32_t *status) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbuckets) return;
    uint32_t lo = start[b], hi = start[b + 1];
    if (lo >= hi) return;
    uint32_t cnt = (hi - 1) / seg_len - lo / seg_len + 1;
    if (cnt <= group_cap) return;
    uint32_t *joins = queue + kQueueHead, *items = joins + 3 * (size_t)max_joins;
    const uint32_t nch = (cnt + chunk_parts - 1) / chunk_parts;
    const uint32_t item0 = atomicAdd(&queue[0], nch);
    // the host sized the queue for any bounds of ascending keys; anything else is reported, not written
    bool fits = item0 + nch <= max_items;
    uint32_t slot0 = kDirectSlot;
    if (fits && nch > 1) {
        slot0 = atomicAdd(&queue[2], nch);
        const uint32_t j = atomicAdd(&queue[1], 1u);
        fits = slot0 + nch <= max_slots && j < max_joins;
        if (fits) {
            joins[3 * j] = b;
            joins[3 * j + 1] = slot0;
            joins[3 * j + 2] = nch;
        }
    }
    if (!fits) {
        __hip_atomic_store(status, kStatusKeysOutOfOrder, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    for (uint32_t i = 0; i < nch; i++) {
        uint32_t *it = items + 3 * (size_t)(item0 + i);
        it[0] = b;
        it[1] = i;
        it[2] = nch > 1 ? slot0 + i : kDirectSlot;
    }
}

// Blocks [0, group_blocks): a group of 2^g_log lanes per bucket; each lane adds up to per_lane of
// the bucket's partials (stride 2^g_log), then a shuffle tree joins the group.  Whole buckets
// (written by k_accumulate in its lazy form) and empty ones pass through the same code.
// Blocks behind them: one block per queued chunk of a long bucket (grid-stride over the queue's items).
__global__ __launch_bounds__(kTreeBlock) void k_fixup(TailBatch tb, uint32_t nbuckets) {
    __shared__ G1Xyzz28 sh[kTreeBlock / 64];
    // one MSM of the batch per grid row; blocks past this MSM's lane groups all serve its queue
    const TailDesc &d = tb.d[blockIdx.y];
    const uint32_t *__restrict__ start = d.start;
    const G1Xyzz28 *__restrict__ part = (const G1Xyzz28 *)d.part;
    G1Xyzz28 *__restrict__ buckets = (G1Xyzz28 *)d.buckets;
    const uint32_t seg_len = d.seg_len, g_log = d.g_log, per_lane = d.per_lane, group_blocks = d.group_blocks;
    if (blockIdx.x >= group_blocks) {
        const uint32_t *__restrict__ queue = d.queue;
        G1Xyzz28 *__restrict__ chunk_out = (G1Xyzz28 *)d.chunk_out;
        const uint32_t count = queue[0] < d.max_items ? queue[0] : d.max_items;
        const uint32_t *__restrict__ items = queue + kQueueHead + 3 * (size_t)d.max_joins;
        for (uint32_t i = blockIdx.x - group_blocks; i < count; i += gridDim.x - group_blocks) {
            const uint32_t b = items[3 * (size_t)i], first = items[3 * (size_t)i + 1] * d.chunk_parts;
            const uint32_t slot = items[3 * (size_t)i + 2];
            const uint32_t lo = start[b], hi = start[b + 1];
            const uint32_t s0 = lo / seg_len, cnt = (hi - 1) / seg_len - s0 + 1;
            const uint32_t m = cnt - first < d.chunk_parts ? cnt - first : d.chunk_parts;
            G1Xyzz28 acc =
                block_sum(m, sh, [&](uint32_t j) { return partial_of(part, s0, first + j, lo, seg_len); });
            if (threadIdx.x == 0) (slot == kDirectSlot ? buckets[b] : chunk_out[slot]) = acc;
        }
        return;
    }
    const uint32_t G = 1u << g_log;
    const size_t gl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = (uint32_t)(gl >> g_log), r = (uint32_t)gl & (G - 1);
    const bool valid = b < nbuckets;
    uint32_t lo = 0, hi = 0, s0 = 0, cnt = 0;
    if (valid) {
        lo = start[b];
        hi = start[b + 1];
        if (lo != hi) {
            s0 = lo / seg_len;
            cnt = (hi - 1) / seg_len - s0 + 1;
        }
    }
    const bool queued = cnt > per_lane * G;
    const bool whole = cnt == 1;  // stored whole by k_accumulate: nothing to add
    G1Xyzz28 acc = xyzz28_identity();
    // strided terms: as many rounds as the longest bucket of this wave needs
    const uint32_t need = (queued || whole) ? 0u : (cnt + G - 1) >> g_log;
    uint32_t step = 0;
    int off = (int)(G >> 1);
    for (;;) {
        G1Xyzz28 o = xyzz28_identity();
        if (__any(step < need)) {  // the same answer in every lane of the wave
            const uint32_t j = step * G + r;
            if (!queued && !whole && j < cnt) o = *partial_of(part, s0, j, lo, seg_len);
            step++;
        } else {
            if (off == 0) break;
            o = shfl_down_point(acc, off, (int)G);
            if ((int)r >= off) o = xyzz28_identity();
            off >>= 1;
        }
        acc = add_inline(acc, o);
    }
    if (valid && r == 0 && !queued && !whole) buckets[b] = acc;  // identity for an empty bucket
}

// The buckets of several chunks: one block each sums the chunk results (grid-stride over the queue's joins;
// grid row = MSM of the batch).
__global__ __launch_bounds__(kTreeBlock) void k_fixup_join(TailBatch tb) {
    __shared__ G1Xyzz28 sh[kTreeBlock / 64];
    const TailDesc &d = tb.d[blockIdx.y];
    const uint32_t *__restrict__ queue = d.queue;
    const G1Xyzz28 *__restrict__ chunk_out = (const G1Xyzz28 *)d.chunk_out;
    G1Xyzz28 *__restrict__ buckets = (G1Xyzz28 *)d.buckets;
    const uint32_t count = queue[1] < d.max_joins ? queue[1] : d.max_joins;
    const uint32_t *__restrict__ joins = queue + kQueueHead;
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t b = joins[3 * (size_t)i], slot0 = joins[3 * (size_t)i + 1], nch = joins[3 * (size_t)i + 2];
        G1Xyzz28 acc = block_sum(nch, sh, [&](uint32_t j) { return chunk_out + slot0 + j; });
        if (threadIdx.x == 0) buckets[b] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// 6. reduction  S = sum_b (b + 1) B_b  over nb = H * L buckets, b = hi * L + lo:
//      S = sum_i 2^i T_i + T_all,
//      T_i        = sum of the column sums C[lo] with bit i of lo set            (i < lbits)
//      T_(lbits+i) = sum of the row sums R[hi] with bit i of hi set               (i < hbits)
//      T_all      = sum of all buckets,
//    with R[hi] = sum_lo B[hi*L + lo] and C[lo] = sum_hi B[hi*L + lo]: two additions per bucket,
//    then sums over at most L or H elements, then one short Horner chain.
// ------------------------------------------------------------------------------------------
// Reduction workspace of one MSM: H row sums, L column sums, then the bit sums.
// blocks [0, H): row sums; blocks [H, H + L): column sums; grid row = MSM of the batch
__global__ __launch_bounds__(kTreeBlock) void k_row_col_sums(TailBatch tb, uint32_t H, uint32_t L) {
    __shared__ G1Xyzz28 sh[kTreeBlock / 64];
    const TailDesc &d = tb.d[blockIdx.y];
    if (!d.out) return;
    const G1Xyzz28 *__restrict__ B = (const G1Xyzz28 *)d.buckets;
    G1Xyzz28 *__restrict__ R = (G1Xyzz28 *)d.red, *__restrict__ C = R + H;
    const uint32_t i = blockIdx.x;
    const bool row = i < H;
    const G1Xyzz28 *base = row ? B + (size_t)i * L : B + (i - H);
    const size_t stride = row ? 1 : L;
    G1Xyzz28 acc = block_sum(row ? L : H, sh, [&](uint32_t j) { return base + (size_t)j * stride; });
    if (threadIdx.x == 0) (row ? R[i] : C[i - H]) = acc;
}

// The same with 256 threads per sum and cooperative additions: for bucket arrays small enough that
// the sums do not fill the GPU anyway.
__global__ __launch_bounds__(256) void k_row_col_sums_coop(TailBatch tb, uint32_t H, uint32_t L) {
    __shared__ CoopMem mem;
    const TailDesc &d = tb.d[blockIdx.y];
    if (!d.out) return;
    const G1Xyzz28 *__restrict__ B = (const G1Xyzz28 *)d.buckets;
    G1Xyzz28 *__restrict__ R = (G1Xyzz28 *)d.red, *__restrict__ C = R + H;
    const uint32_t i = blockIdx.x;
    const bool row = i < H;
    const G1Xyzz28 *base = row ? B + (size_t)i * L : B + (i - H);
    const size_t stride = row ? 1 : L;
    G1Xyzz28 acc = block_sum_coop(row ? L : H, mem, [&](uint32_t j) { return base + (size_t)j * stride; });
    if (threadIdx.x == 0) (row ? R[i] : C[i - H]) = acc;
}

// Large bucket arrays: the sums in two levels.  With one wave per row or column the strided part of a sum is a
// chain of L / 64 (or H / 64) dependent additions and the H + L waves are neither few enough to run at once nor
// many enough to even out.  Level 1 puts one lane on every m buckets instead: lane (h, j) of the first H * L / m
// adds up B[h][j], B[h][j + L/m], ... (m terms of row h), lane (k, l) of the second H * L / m adds up B[k][l],
// B[k + H/m][l], ... (m terms of column l); neighbouring lanes read neighbouring buckets, every lane is busy for
// the whole chain, and m is chosen so that the grid is about what the device keeps resident.  Level 2 is the
// one-wave-per-sum kernel over the L / m partials of a row and the H / m partials of a column.
// Workspace (behind the bit sums): RP[H][L/m], then CP[H/m][L].
__device__ __forceinline__ size_t reduce_partials_offset(uint32_t H, uint32_t L) {
    return (size_t)H + L + (uint32_t)(__ffs((int)H) - 1) + (uint32_t)(__ffs((int)L) - 1) + 1;  // R, C, bit sums
}

__global__ __launch_bounds__(128, 2) void k_row_col_partials(TailBatch tb, uint32_t H, uint32_t L, uint32_t m) {
    const TailDesc &d = tb.d[blockIdx.y];
    if (!d.out) return;
    const G1Xyzz28 *__restrict__ B = (const G1Xyzz28 *)d.buckets;
    G1Xyzz28 *__restrict__ P = (G1Xyzz28 *)d.red + reduce_partials_offset(H, L);
    const uint32_t per_kind = H * (L / m);  // = (H / m) * L
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * per_kind) return;
    const G1Xyzz28 *src;
    size_t stride;
    if (i < per_kind) {
        const uint32_t lr = L / m, h = i / lr, j = i % lr;
        src = B + (size_t)h * L + j;
        stride = lr;
    } else {
        src = B + (i - per_kind);
        stride = (size_t)(H / m) * L;
    }
    G1Xyzz28 acc = src[0];
#pragma unroll 1
    for (uint32_t t = 1; t < m; t++) acc = add_inline(acc, src[t * stride]);
    P[i] = acc;
}

// blocks [0, H): row sums over RP; blocks [H, H + L): column sums over CP
__global__ __launch_bounds__(kTreeBlock) void k_row_col_sums_of_partials(TailBatch tb, uint32_t H, uint32_t L, uint32_t m) {
    __shared__ G1Xyzz28 sh[kTreeBlock / 64];
    const TailDesc &d = tb.d[blockIdx.y];
    if (!d.out) return;
    G1Xyzz28 *__restrict__ R = (G1Xyzz28 *)d.red, *__restrict__ C = R + H;
    const G1Xyzz28 *__restrict__ RP = R + reduce_partials_offset(H, L), *__restrict__ CP = RP + (size_t)H * (L / m);
    const uint32_t i = blockIdx.x;
    const bool row = i < H;
    const G1Xyzz28 *base = row ? RP + (size_t)i * (L / m) : CP + (i - H);
    const size_t stride = row ? 1 : L;
    G1Xyzz28 acc = block_sum(row ? L / m : H / m, sh, [&](uint32_t j) { return base + (size_t)j * stride; });
    if (threadIdx.x == 0) (row ? R[i] : C[i - H]) = acc;
}

// H = L = 256: two sums per block (rows 2i and 2i + 1, or columns 2i and 2i + 1)
__global__ __launch_bounds__(256) void k_row_col_sums_coop_pair(TailBatch tb) {
    __shared__ CoopMem mem;
    constexpr uint32_t H = 256, L = 256;
    const TailDesc &d = tb.d[blockIdx.y];
    if (!d.out) return;
    const G1Xyzz28 *__restrict__ B = (const G1Xyzz28 *)d.buckets;
    G1Xyzz28 *__restrict__ R = (G1Xyzz28 *)d.red, *__restrict__ C = R + H;
    const uint32_t i = 2 * blockIdx.x;
    const bool row = i < H;
    const G1Xyzz28 *base = row ? B + (size_t)i * L : B + (i - H);
    const size_t stride = row ? 1 : L, next = row ? L : 1;
    block_sum_coop_pair(mem, [&](uint32_t j) { return base + (size_t)j * stride; },
                        [&](uint32_t j) { return base + next + (size_t)j * stride; });
    if (threadIdx.x == 0) {
        (row ? R[i] : C[i - H]) = coop_load(mem, kCoopA, 0);
        (row ? R[i + 1] : C[i + 1 - H]) = coop_load(mem, kCoopA, 32);
    }
}

// block i < lbits: T_i over C; lbits <= i < lbits + hbits: T_i over R; block lbits + hbits: T_all
__global__ __launch_bounds__(256) void k_bit_sums(TailBatch tb, uint32_t hbits, uint32_t lbits) {
    __shared__ CoopMem mem;
    const TailDesc &d = tb.d[blockIdx.y];
    if (!d.out) return;
    const G1Xyzz28 *__restrict__ R = (const G1Xyzz28 *)d.red, *__restrict__ C = R + ((size_t)1 << hbits);
    G1Xyzz28 *__restrict__ T = (G1Xyzz28 *)d.red + ((size_t)1 << hbits) + ((size_t)1 << lbits);
    const uint32_t i = blockIdx.x;
    const bool all = i == lbits + hbits, low = all || i < lbits;
    const G1Xyzz28 *A = low ? C : R;
    const uint32_t nbits = low ? lbits : hbits;
    const uint32_t bit = all ? 31u : (low ? i : i - lbits);
    // j runs over the indices with `bit` removed; put a one back in at that position
    G1Xyzz28 acc = block_sum_coop(all ? 1u << nbits : 1u << (nbits - 1), mem, [&](uint32_t j) {
        return A + (all ? j : ((((j >> bit) << 1) | 1u) << bit | (j & ((1u << bit) - 1))));
    });
    if (threadIdx.x == 0) T[i] = acc;
}

// One block of four waves, slot d < ndig for the radix-8 digit U_d = T_3d + 2 T_(3d+1) + 4 T_(3d+2),
// which is then doubled 3d times; slot ndig holds T_all; a tree adds the ndig + 1 slots:
//   S = sum_d 8^d U_d + T_all.
// All steps are cooperative (staged) doublings and additions on lazy values.  Jacobian result.
// For a slice [first, first + 2^sbits) of a larger bucket array (first a multiple of 2^sbits, T the
// bit sums of the slice, T[sbits] its total) the bits of `first` stand for the whole slice:
// T_i = T_all for a set bit i >= sbits of first, O for a clear one.
__global__ __launch_bounds__(256) void k_finish(TailBatch tb, uint32_t sbits, uint32_t nbits, uint32_t first) {
    __shared__ CoopMem m;
    const TailDesc &desc = tb.d[blockIdx.x];
    if (!desc.out) return;
    const uint32_t lbits_ = (sbits + 1) / 2, hbits_ = sbits - lbits_;
    const G1Xyzz28 *__restrict__ T = (const G1Xyzz28 *)desc.red + ((size_t)1 << hbits_) + ((size_t)1 << lbits_);
    G1Jac *__restrict__ out = (G1Jac *)desc.out;
    const uint32_t w = threadIdx.x >> 6, d = threadIdx.x & 63;
    const uint32_t ndig = (nbits + 2) / 3;  // at most 8: nbits <= 23
    auto term = [&](uint32_t bit) {
        if (d >= ndig || bit >= nbits) return xyzz28_identity();
        if (bit < sbits) return T[bit];
        return ((first >> bit) & 1u) ? T[sbits] : xyzz28_identity();
    };
    if (w == 0) coop_store(m, kCoopA, d, term(3 * d + 2));
    __syncthreads();
    // The steps as one loop with one doubling and one addition in its body:
    //   0, 2: double the digits;  1, 3: add term 3d + 1, then 3d;  4 .. 4 + 3 (ndig - 1) - 1: digit d doubled 3d times;
    //   then T_all joins as slot ndig and four steps of a tree (offsets 8, 4, 2, 1) add the ndig + 1 slots
    const uint32_t shifts = 3 * (ndig - 1), nsteps = 4 + shifts + 4;
#pragma unroll 1
    for (uint32_t i = 0; i < nsteps; i++) {
        if (i < 4 + shifts && (i >= 4 || (i & 1) == 0)) {
            coop_dbl(m, d < ndig && (i < 4 || i - 4 < 3 * d));
            continue;
        }
        bool b_self = false, run = true;
        uint32_t boff = 0, nact = ndig;
        if (i < 4) {
            if (w == 0) coop_store(m, kCoopB, d, term(3 * d + (i == 1 ? 1u : 0u)));
        } else {
            const uint32_t off = 8u >> (i - 4 - shifts);
            if (off == 8 && w == 0 && d == ndig) coop_store(m, kCoopA, d, T[sbits]);
            // slots s < off whose partner s + off is one of the ndig + 1 terms
            b_self = true;
            boff = off;
            nact = ndig + 1 - off < off ? ndig + 1 - off : off;
            run = ndig + 1 > off;
        }
        __syncthreads();
        if (run) coop_add(m, b_self, boff, nact);
    }
    if (threadIdx.x == 0) *out = xyzz_to_jac(xyzz28_to(xyzz28_canon(coop_load(m, kCoopA, 0))));
}

// B[b] = sum over p of parts[p * count + b]: the bucket partials of `nparts` GPUs for one slice of
// the bucket array, one lane per bucket.
__global__ __launch_bounds__(128) void k_sum_parts(const G1Xyzz28 *__restrict__ parts, uint32_t nparts, uint32_t count,
                                                   G1Xyzz28 *__restrict__ B) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= count) return;
    G1Xyzz28 acc = parts[b];
#pragma unroll 1
    for (uint32_t p = 1; p < nparts; p++) acc = add_inline(acc, parts[(size_t)p * count + b]);
    B[b] = acc;
}

// out = sum of n Jacobian points (n is the number of GPUs: a handful)
__global__ void k_sum_jac(const G1Jac *__restrict__ pts, uint32_t n, G1Jac *__restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    G1Xyzz acc = xyzz_identity();
    for (uint32_t i = 0; i < n; i++) acc = xyzz_add(acc, jac_to_xyzz(pts[i]));
    *out = xyzz_to_jac(acc);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
MsmPlan msm_choose_plan(size_t srs_n, int override_bits) {
    MsmPlan p;
    int best = 0;
    if (override_bits >= 2 && override_bits <= 24) {
        best = override_bits;
    } else if (srs_n > 1024) {
        // Measured on MI355X.  Widths whose top window is narrow (255 mod c small but not 0) pile that
        // window's digits into a few buckets and pay for it in the fix-up, hence 13, 20, 22 and the two
        // divisors of 255, whose windows are all full: 15 and 17.
        int lg = 0;
        while (((size_t)1 << lg) < srs_n) lg++;
        best = lg <= 13 ? 13 : (lg <= 18 ? 15 : (lg <= 20 ? 17 : (lg <= 23 ? 20 : 22)));
    } else {
        // Small inputs: cost in "bucket additions"; the accumulation runs ~2^18 lanes at once, the
        // reduction is a serial tail whose depth grows with c.
        double best_cost = 1e300;
        const double lanes = 262144.0;
        for (int c = 4; c <= 13; c++) {
            double nwin = msm_num_windows(c);
            double nb = std::ldexp(1.0, c - 1);
            double total = (double)srs_n * nwin;
            double per_bucket = total / nb;
            double seg = std::max(std::sqrt(per_bucket), total / 1048576.0);
            double accumulate = std::max(total / lanes, seg);
            double fixup = 1.4 * std::max(per_bucket / seg, nb * (per_bucket / seg) / lanes);
            double reduce_work = 1.4 * 3.0 * nb / lanes;
            double reduce_depth = 1.4 * (8.0 * std::ceil((c - 1) / 2.0) + 18.0 + 1.5 * (c - 1));
            double cost = accumulate + fixup + reduce_work + reduce_depth;
            if (cost < best_cost) {
                best_cost = cost;
                best = c;
            }
        }
    }
    // table index (w * srs_n + i) must fit in 31 bits
    while (best < 24 && (double)srs_n * msm_num_windows(best) >= 2147483648.0) best++;
    p.c = best;
    p.nwin = msm_num_windows(best);
    p.nbuckets = 1u << (best - 1);
    return p;
}

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

static void record(plonk_hip_ctx *ctx, int i) {
    if (ctx->profile) (void)hipEventRecord(ctx->ev[i], ctx->stream);
}

int g1_sum_dev(plonk_hip_ctx *ctx, const G1Jac *d_pts, size_t n, G1Jac *d_out) {
    if (n == 0 || n > 0xffffffffull) return PLONK_HIP_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_sum_jac, dim3(1), dim3(1), 0, ctx->stream, d_pts, (uint32_t)n, d_out);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

size_t msm_bucket_bytes() { return sizeof(G1Xyzz28); }

// Elements of reduction workspace one MSM needs: row sums, column sums, bit sums.
// Buckets per lane of the first level of the two-level row and column sums (0: one level).
static uint32_t reduce_partial_terms(uint32_t sbits) { return sbits >= 18 ? 1u << (sbits - 16) : 0u; }

static size_t reduce_ws_elems(uint32_t sbits) {
    const uint32_t lbits = (sbits + 1) / 2, hbits = sbits - lbits;
    const uint32_t m = reduce_partial_terms(sbits);
    return ((size_t)1 << hbits) + ((size_t)1 << lbits) + sbits + 1 + (m ? 2 * (((size_t)1 << sbits) / m) : 0);
}

// For every MSM of the batch with a result pointer: S = sum_b (first + b + 1) B[b] over its 2^sbits
// buckets as a Jacobian point; `first` is a multiple of 2^sbits below 2^nbits (0 and sbits == nbits for a
// whole bucket array).
static int reduce_buckets(plonk_hip_ctx *ctx, const TailBatch &tb, uint32_t count, uint32_t sbits, uint32_t nbits,
                          uint32_t first) {
    const uint32_t lbits = (sbits + 1) / 2, hbits = sbits - lbits;
    const uint32_t L = 1u << lbits, H = 1u << hbits;
    hipStream_t st = ctx->stream;
    // one sum per block: wide blocks (small depth) while they are few, one wave per sum once
    // there is a sum for every SIMD (the tree levels of a wide block are mostly idle lanes)
    const uint32_t sums = H + L;
    const size_t all_sums = (size_t)sums * count;
    const uint32_t width = all_sums >= 1024 ? 64u : 128u;
    const uint32_t m = reduce_partial_terms(sbits);
    if (m) {
        const uint32_t lanes = 2 * ((1u << sbits) / m);
        hipLaunchKernelGGL(k_row_col_partials, dim3(lanes / 128, count), dim3(128), 0, st, tb, H, L, m);
        hipLaunchKernelGGL(k_row_col_sums_of_partials, dim3(sums, count), dim3(64), 0, st, tb, H, L, m);
    } else if (H == 256 && L == 256 && count == 1) {  // more sums than compute units: two per block
        hipLaunchKernelGGL(k_row_col_sums_coop_pair, dim3(sums / 2, count), dim3(256), 0, st, tb);
    } else if (all_sums <= 512) {  // (at 512 sums: 0.60 against 0.62 ms for the whole reduction of 2^16 buckets)
        hipLaunchKernelGGL(k_row_col_sums_coop, dim3(sums, count), dim3(256), 0, st, tb, H, L);
    } else {
        hipLaunchKernelGGL(k_row_col_sums, dim3(sums, count), dim3(width), 0, st, tb, H, L);
    }
    hipLaunchKernelGGL(k_bit_sums, dim3(sbits + 1, count), dim3(256), 0, st, tb, hbits, lbits);
    hipLaunchKernelGGL(k_finish, dim3(count), dim3(256), 0, st, tb, sbits, nbits, first);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

// Multi-GPU exchange of bucket partials: rank g receives, from every rank, that rank's partial
// sums of the buckets [first, first + count) and finishes that slice.  parts: nparts arrays of
// count buckets each (the layout msm_run_dev leaves in d_buckets_out), count a power of two >= 4,
// first a multiple of count.
int msm_reduce_slice_dev(plonk_hip_ctx *ctx, const void *d_parts, size_t nparts, size_t first, size_t count,
                         G1Jac *d_out) {
    if (ctx->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    const uint32_t nb = ctx->plan.nbuckets;
    if (nparts == 0 || nparts > 1024 || count < 4 || (count & (count - 1)) || count > nb || first % count ||
        first + count > nb)
        return PLONK_HIP_ERR_INVALID_ARG;
    PLONK_TRY(msm_join(ctx));  // the workspace below is shared with pending tails
    uint32_t sbits = 0;
    while (((size_t)1 << sbits) < count) sbits++;
    const G1Xyzz28 *B = (const G1Xyzz28 *)d_parts;
    if (nparts > 1) {
        PLONK_TRY(buf_reserve(ctx, ctx->buckets, (size_t)nb * sizeof(G1Xyzz28)));
        hipLaunchKernelGGL(k_sum_parts, dim3(cdiv(count, 128)), dim3(128), 0, ctx->stream, B, (uint32_t)nparts,
                           (uint32_t)count, (G1Xyzz28 *)ctx->buckets.ptr);
        B = (const G1Xyzz28 *)ctx->buckets.ptr;
    }
    PLONK_TRY(buf_reserve(ctx, ctx->red_a, reduce_ws_elems(sbits) * sizeof(G1Xyzz28)));
    TailBatch tb = {};
    tb.d[0].buckets = (void *)B;
    tb.d[0].red = ctx->red_a.ptr;
    tb.d[0].out = d_out;
    return reduce_buckets(ctx, tb, 1, sbits, (uint32_t)ctx->plan.c - 1, (uint32_t)first);
}

// d_out: the MSM as a Jacobian point.  With d_buckets_out set (nbuckets x sizeof(G1Xyzz28) bytes of
// device memory) the pipeline stops after the fix-up and leaves the bucket sums there instead.
int msm_run_dev(plonk_hip_ctx *ctx, const Fr *d_scalars, size_t n, G1Jac *d_out, void *d_buckets_out, bool overlap,
                size_t first) {
    if (ctx->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    if (n == 0) return PLONK_HIP_ERR_EMPTY_POLY;
    if (first > ctx->srs_n || n > ctx->srs_n - first) return PLONK_HIP_ERR_SRS_TOO_SHORT;
    const MsmPlan &pl = ctx->plan;
    const size_t total = n * (size_t)pl.nwin;
    if (total >= 0xfffffff0ull) return PLONK_HIP_ERR_INVALID_ARG;
    const uint32_t nb = pl.nbuckets;
    hipStream_t st = ctx->stream;

    // Segment length: for small inputs balance the accumulation depth (seg_len mixed additions in
    // a row) against the fix-up depth; for large ones keep about 2^20 segments (several rounds of
    // resident lanes, which evens out the tail of the kernel); in between, at least 2^18 segments
    // of up to 32 positions (fewer partial sums for the fix-up).
    const double per_bucket = (double)total / nb;
    const double fill = std::min(std::ceil((double)total / 262144.0),
                                 std::max(32.0, std::ceil((double)total / 1048576.0)));
    uint32_t seg_len = (uint32_t)std::max<double>(1.0, std::max(std::ceil(std::sqrt(per_bucket)), fill));
    // With a few rounds of resident lanes the last round's fill shows: the shortest length that fills a whole
    // number of rounds is taken.
    const double resident = (double)ctx->num_cus * 4 * 2 * 64;
    const double rounds = (double)total / seg_len / resident;
    if (rounds >= 1.5 && rounds <= 16.0) {
        seg_len = (uint32_t)std::ceil((double)total / (std::floor(rounds + 0.5) * resident));
    }
    if (ctx->tune_seg_len) seg_len = ctx->tune_seg_len;
    const size_t nseg = (total + seg_len - 1) / seg_len;

    // A deferred tail (overlap) takes the next free slot; anything else first finishes what is pending
    // and uses slot 0, so that a single MSM never waits behind a batch it does not belong to.
    if (d_buckets_out || ctx->profile) overlap = false;
    if (!overlap || ctx->npending == kMaxTailSlots) PLONK_TRY(msm_join(ctx));
    const int slot = ctx->npending;

    const int key_bits = pl.c - 1;  // the sort orders the pairs by bucket
    size_t tmp_bytes = sort_pairs_temp_bytes(total, key_bits);
    PLONK_TRY(buf_reserve(ctx, ctx->keys_a, (total + 16) * 4));
    PLONK_TRY(buf_reserve(ctx, ctx->keys_b, (total + 16) * 4));
    PLONK_TRY(buf_reserve(ctx, ctx->vals_a, (total + 16) * 4));
    PLONK_TRY(buf_reserve(ctx, ctx->vals_b, (total + 16) * 4));
    PLONK_TRY(buf_reserve(ctx, ctx->sort_tmp, tmp_bytes));
    DevBuf &b_start = slot ? ctx->tail_bufs[slot - 1].bucket_start : ctx->bucket_start;
    DevBuf &b_buckets = slot ? ctx->tail_bufs[slot - 1].buckets : ctx->buckets;
    DevBuf &b_part = slot ? ctx->tail_bufs[slot - 1].seg_part : ctx->seg_part;
    DevBuf &b_big = slot ? ctx->tail_bufs[slot - 1].big_list : ctx->big_list;
    PLONK_TRY(buf_reserve(ctx, b_start, ((size_t)nb + 2) * 4));
    if (!d_buckets_out) PLONK_TRY(buf_reserve(ctx, b_buckets, (size_t)nb * sizeof(G1Xyzz28)));
    PLONK_TRY(buf_reserve(ctx, b_part, nseg * 2 * sizeof(G1Xyzz28)));

    uint32_t *keys_a = (uint32_t *)ctx->keys_a.ptr, *keys_b = (uint32_t *)ctx->keys_b.ptr;
    uint32_t *vals_a = (uint32_t *)ctx->vals_a.ptr, *vals_b = (uint32_t *)ctx->vals_b.ptr;
    uint32_t *start = (uint32_t *)b_start.ptr;
    G1Xyzz28 *buckets = d_buckets_out ? (G1Xyzz28 *)d_buckets_out : (G1Xyzz28 *)b_buckets.ptr;
    G1Xyzz28 *part = (G1Xyzz28 *)b_part.ptr;
    const G1Affine28 *table = (const G1Affine28 *)ctx->table.ptr;

    record(ctx, PH_DIGITS);
    // Large inputs: the sort's first pass is made from the scalars themselves, the unsorted pairs never exist
    // (sort.hip).  Small ones: pairs first, then the sort -- fewer and shorter launches.
    const bool from_scalars = ctx->front_end != kFrontEndPairs && sort_pairs_of_scalars_applies(n, pl.c, pl.nwin) &&
                              (ctx->front_end == kFrontEndScalars || n >= kFrontEndScalarsMin);
    if (ctx->debug_skip_sort) {
        // fault injection: the pairs as the digits come, in the buffers the sort would have left them in
        const bool odd = (sort_pairs_passes(key_bits) & 1) != 0;
        hipLaunchKernelGGL(k_digits, dim3(cdiv(n, 256)), dim3(256), 0, st, d_scalars, (uint32_t)n, (uint32_t)ctx->srs_n,
                           (uint32_t)first, pl.c, pl.nwin, nb, odd ? keys_b : keys_a, odd ? vals_b : vals_a);
        record(ctx, PH_SORT);
    } else if (from_scalars) {
        record(ctx, PH_SORT);
        PLONK_TRY(sort_pairs_of_scalars(ctx, d_scalars, n, ctx->srs_n, first, pl.c, pl.nwin, nb, keys_a, keys_b, vals_a,
                                        vals_b, ctx->sort_tmp.ptr, tmp_bytes));
    } else {
        hipLaunchKernelGGL(k_digits, dim3(cdiv(n, 256)), dim3(256), 0, st, d_scalars, (uint32_t)n, (uint32_t)ctx->srs_n,
                           (uint32_t)first, pl.c, pl.nwin, nb, keys_a, vals_a);
        record(ctx, PH_SORT);
        PLONK_TRY(sort_pairs(ctx, keys_a, keys_b, vals_a, vals_b, total, key_bits, ctx->sort_tmp.ptr, tmp_bytes));
    }
    if ((sort_pairs_passes(key_bits) & 1) == 0) {  // an even number of passes ends in the first pair of buffers
        keys_b = keys_a;
        vals_b = vals_a;
    }
    record(ctx, PH_BOUNDS);
    hipLaunchKernelGGL(k_bounds, dim3(cdiv((size_t)nb + 2, 256)), dim3(256), 0, st, keys_b, total, nb, start);
    // Fix-up shape: a typical bucket is cut into about per_bucket / seg_len + 1 partials; a group
    // of 2^g_log lanes takes up to per_lane partials per lane, anything longer (skewed digit
    // distributions: the narrow top window, repeated scalars) goes to the block path.
    const double typical = per_bucket / seg_len + 1.0;
    // wider groups shorten the chain of dependent additions but leave lanes idle in the tree
    // levels: widen only while the kernel stays below about two waves per SIMD
    uint32_t g_log = 0;
    while (g_log < 5 && (double)(2u << g_log) < typical && ((size_t)nb << (g_log + 1)) <= 131072) g_log++;
    // heavy buckets sit next to each other (the narrow top window fills the lowest bucket numbers),
    // so with many buckets a generous serial bound costs little and the block path is kept for the
    // really long ones; with few buckets depth is all that matters
    const uint32_t per_lane = std::max<uint32_t>(((size_t)nb << g_log) > 65536 ? 16u : 3u,
                                                 (uint32_t)std::ceil(2.0 * typical / (1u << g_log)));
    const uint32_t group_blocks = cdiv((size_t)nb << g_log, kTreeBlock);
    // Queue of the long buckets: the (bucket, segment) incidences number at most nseg + nb, a queued bucket has
    // more than `cap` of them, one of several chunks more than chunk_parts.  Chunks of at least 256 partials, and
    // long enough that about 512 blocks (what the GPU runs at once) cover every segment.
    const uint32_t cap = per_lane << g_log;
    const size_t incidences = nseg + nb;
    uint32_t chunk_parts = 256;
    while (chunk_parts < 8192 && (size_t)chunk_parts * 512 < nseg) chunk_parts *= 2;
    const uint32_t max_joins = (uint32_t)(incidences / chunk_parts + 1);
    const uint32_t max_items = (uint32_t)(incidences / chunk_parts + incidences / cap + 2);
    const uint32_t max_slots = 2 * max_joins;
    DevBuf &b_chunk = slot ? ctx->tail_bufs[slot - 1].chunk_out : ctx->chunk_out;
    PLONK_TRY(buf_reserve(ctx, b_big, (kQueueHead + 3 * ((size_t)max_joins + max_items)) * 4));
    PLONK_TRY(buf_reserve(ctx, b_chunk, (size_t)max_slots * sizeof(G1Xyzz28)));
    uint32_t *queue = (uint32_t *)b_big.ptr;
    PLONK_HIP_TRY(ctx, hipMemsetAsync(queue, 0, kQueueHead * 4, st));
    hipLaunchKernelGGL(k_classify, dim3(cdiv(nb, 256)), dim3(256), 0, st, start, nb, seg_len, cap, chunk_parts, queue, max_joins,
                       max_items, max_slots, ctx->status_dev);
    record(ctx, PH_ACCUM);
    // segments shorter than a group: nothing to bring in ahead
    const bool lds_pairs = ctx->tune_pair_loads ? ctx->tune_pair_loads == 2 : seg_len >= kGroup;
    hipLaunchKernelGGL(lds_pairs ? k_accumulate<true> : k_accumulate<false>, dim3(cdiv(nseg, kBlock)), dim3(kBlock), 0, st,
                       keys_b, vals_b, start, nb, seg_len, table, buckets, part, ctx->status_dev);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    TailDesc &d = ctx->pending[slot];
    d.start = start;
    d.part = part;
    d.buckets = buckets;
    d.queue = queue;
    d.chunk_out = b_chunk.ptr;
    d.max_joins = max_joins;
    d.max_items = max_items;
    d.chunk_parts = chunk_parts;
    d.red = nullptr;  // assigned when the tails run
    d.out = d_buckets_out ? nullptr : (void *)d_out;
    d.seg_len = seg_len;
    d.g_log = g_log;
    d.per_lane = per_lane;
    d.group_blocks = group_blocks;
    ctx->npending = slot + 1;
    if (!overlap) PLONK_TRY(msm_join(ctx));
    return PLONK_HIP_OK;
}

// The tails of all pending MSMs: one fix-up launch and one reduction (three launches) for all of them.
int msm_join(plonk_hip_ctx *ctx) {
    const uint32_t count = (uint32_t)ctx->npending;
    if (count == 0) return PLONK_HIP_OK;
    ctx->npending = 0;  // whatever happens below, the slots are free again
    const MsmPlan &pl = ctx->plan;
    const uint32_t nb = pl.nbuckets, sbits = (uint32_t)pl.c - 1;
    const size_t ws = reduce_ws_elems(sbits);
    PLONK_TRY(buf_reserve(ctx, ctx->red_a, ws * count * sizeof(G1Xyzz28)));
    TailBatch tb = {};
    uint32_t group_blocks = 0;
    bool any_out = false;
    for (uint32_t i = 0; i < count; i++) {
        tb.d[i] = ctx->pending[i];
        tb.d[i].red = (G1Xyzz28 *)ctx->red_a.ptr + ws * i;
        group_blocks = std::max(group_blocks, tb.d[i].group_blocks);
        any_out = any_out || tb.d[i].out;
    }
    // blocks for the queue of long buckets: usually empty, at most a few thousand chunks
    const uint32_t big_blocks = std::min<uint32_t>(nb, 2048u);
    record(ctx, PH_FIXUP);
    hipLaunchKernelGGL(k_fixup, dim3(group_blocks + big_blocks, count), dim3(kTreeBlock), 0, ctx->stream, tb, nb);
    hipLaunchKernelGGL(k_fixup_join, dim3(std::min<uint32_t>(nb, 256u), count), dim3(kTreeBlock), 0, ctx->stream, tb);
    record(ctx, PH_REDUCE);
    if (any_out) PLONK_TRY(reduce_buckets(ctx, tb, count, sbits, sbits, 0));
    record(ctx, PH_COUNT);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

}  // namespace plonk
