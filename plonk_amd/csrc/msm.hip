// G1 multi-scalar multiplication behind kzg_commit / kzg_evaluate: signed-digit bucket method
// (Pippenger) over resident window tables.
//
// The SRS is fixed, so the context keeps [2^(c*w)] P_i for every window w (srs.hip).  All
// windows then share ONE set of 2^(c-1) buckets and no window is ever shifted by doublings:
//     sum_i k_i P_i = sum_b (b+1) * B_b,   B_b = sum over (i, w) with |digit_w(k_i)| = b+1
//                                               of sign * table[w][i].
// Pipeline (all on the context's stream, no host synchronisation):
//   1. k_digits      scalars out of Montgomery form, nwin signed digits each -> (bucket, entry)
//   2. sort_pairs    radix sort of the n*nwin pairs by bucket
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

#include "context.hpp"
#include "g1_28.cuh"
#include "msm_core.cuh"

namespace plonk {

static constexpr int kBlock = 128;
static constexpr int kChunk = 4;       // bucket-reduction chunk (the "4" in the Horner step)
static constexpr int kChunkLog = 2;
static constexpr int kMaxLevels = 16;  // 2^(c-1) buckets, c <= 24 -> 12 levels
static constexpr int kSumBlock = 128;

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
__global__ __launch_bounds__(256) void k_digits(const Fr *__restrict__ scalars, uint32_t n, uint32_t srs_n, int c,
                                               int nwin, uint32_t nbuckets, uint32_t *__restrict__ keys,
                                               uint32_t *__restrict__ vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr k = fe_from_mont(scalars[i]);
    uint32_t carry = 0;
    for (int w = 0; w < nwin; w++) {
        int32_t d = signed_digit(k, w, c, carry);
        uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
        // digit 0 contributes nothing: key nbuckets sorts behind every real bucket
        uint32_t key = mag == 0 ? nbuckets : mag - 1;
        uint32_t val = ((uint32_t)w * srs_n + i) | (d < 0 ? 0x80000000u : 0u);
        size_t at = (size_t)w * n + i;
        keys[at] = key;
        vals[at] = val;
    }
}

// ------------------------------------------------------------------------------------------
// 3. bucket bounds: start[b] = first sorted position with key >= b, b = 0..nbuckets;
//    start[nbuckets] = number of non-zero digits, start[nbuckets + 1] = total.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bounds(const uint32_t *__restrict__ keys, size_t total, uint32_t nbuckets,
                                               uint32_t *__restrict__ start) {
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    uint32_t k = keys[p];
    uint32_t first = p == 0 ? 0u : keys[p - 1] + 1u;
    for (uint32_t b = first; b <= k; b++) start[b] = (uint32_t)p;
    if (p == total - 1) {
        for (uint32_t b = k + 1; b <= nbuckets + 1; b++) start[b] = (uint32_t)total;
    }
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

__global__ __launch_bounds__(kBlock, 3) void k_accumulate(const uint32_t *__restrict__ keys,
                                                       const uint32_t *__restrict__ vals,
                                                       const uint32_t *__restrict__ start, uint32_t nbuckets,
                                                       uint32_t seg_len, const G1Affine28 *__restrict__ table,
                                                       G1Xyzz28 *__restrict__ buckets,
                                                       G1Xyzz28 *__restrict__ part) {
    __shared__ uint4 line_buf[kBlock / 64][64 * 8];  // 8 KiB per wave
    const uint32_t lane = threadIdx.x & 63;
    uint4 *buf = line_buf[threadIdx.x >> 6];
    const size_t seg = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nvalid = start[nbuckets];
    const size_t begin64 = seg * seg_len;
    const bool live = begin64 < nvalid;
    if (__ballot(live) == 0) return;  // the whole wave lies behind the list
    // lanes without a segment stay in the loop: they carry part of the other lanes' line loads
    const uint32_t begin = live ? (uint32_t)begin64 : 0u;
    const size_t end64 = begin64 + seg_len;
    const uint32_t end = live ? (end64 < nvalid ? (uint32_t)end64 : nvalid) : 0u;

    // (bucket, entry) of the current position and of the next one; the line of the current
    // position is always on its way to (or in) the wave's buffer
    uint32_t k = 0, v = 0, kn = 0, vn = 0;
    if (begin < end) {
        k = keys[begin];
        v = vals[begin];
    }
    if (begin + 1 < end) {
        kn = keys[begin + 1];
        vn = vals[begin + 1];
    }
    issue_lines(table, buf, lane, v & 0x7fffffffu);
    uint32_t cur = k;
    uint32_t run_begin = begin;
    G1Xyzz28 acc = xyzz28_identity();
    for (uint32_t it = 0; it < seg_len; it++) {  // the same trip count for every lane of the grid
        const uint32_t p = begin + it;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this position's lines have landed
        G1Affine28 pt = read_line(buf, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every lane holds its line: the buffer is free
        issue_lines(table, buf, lane, vn & 0x7fffffffu);
        uint32_t k2 = 0, v2 = 0;
        if (p + 2 < end) {
            k2 = keys[p + 2];
            v2 = vals[p + 2];
        }
        if (p < end) {
            if (k != cur) {
                flush_run(cur, run_begin, p, begin, seg, start, buckets, part, acc);
                cur = k;
                run_begin = p;
                acc = xyzz28_identity();
            }
            if (!affine28_is_identity(pt)) {
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
}

// ------------------------------------------------------------------------------------------
// 5. fix-up: buckets cut by segment borders
// ------------------------------------------------------------------------------------------
// Full XYZZ addition with the (rare) equal-operands case out of line: one inlined copy per
// kernel keeps the serial tails short without blowing up the code object.
__device__ __forceinline__ G1Xyzz28 add_inline(const G1Xyzz28 &a, const G1Xyzz28 &b) {
    G1Xyzz28 r;
    if (!xyzz28_add_nodbl(r, a, b)) {
        G1Xyzz28 t = a;  // a private copy: the caller's value keeps its registers
        r = dbl_cold(&t);
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

// One lane per bucket.  Buckets with more than serial_max partials (skewed digit
// distributions: the narrow top window, repeated scalars) are queued for k_fixup_big.
__global__ __launch_bounds__(kBlock) void k_fixup(const uint32_t *__restrict__ start, uint32_t nbuckets,
                                                  uint32_t seg_len, const G1Xyzz28 *__restrict__ part,
                                                  G1Xyzz28 *__restrict__ buckets, uint32_t serial_max,
                                                  uint32_t *__restrict__ big_count,
                                                  uint32_t *__restrict__ big_list) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbuckets) return;
    uint32_t lo = start[b], hi = start[b + 1];
    if (lo == hi) {
        buckets[b] = xyzz28_identity();
        return;
    }
    uint32_t s0 = lo / seg_len, s1 = (hi - 1) / seg_len;
    if (s0 == s1) {
        // written whole by k_accumulate, in its lazy form: canonicalise in place
        buckets[b] = xyzz28_canon(buckets[b]);
        return;
    }
    uint32_t cnt = s1 - s0 + 1;
    if (cnt > serial_max) {
        big_list[atomicAdd(big_count, 1u)] = b;
        return;
    }
    G1Xyzz28 acc = xyzz28_canon(*partial_of(part, s0, 0, lo, seg_len));
    for (uint32_t j = 1; j < cnt; j++) acc = add_inline(acc, xyzz28_canon(*partial_of(part, s0, j, lo, seg_len)));
    buckets[b] = acc;
}

// Sum of `cnt` points fetched through `at(j)` by one block: strided partial sums, then an LDS
// tree.  The result is valid in lane 0.  One inlined addition serves both stages.  LAZY_IN:
// the fetched points come straight from k_accumulate and are canonicalised first.
template <bool LAZY_IN, class At>
__device__ __forceinline__ G1Xyzz28 block_sum(uint32_t cnt, G1Xyzz28 *sh, At at) {
    const uint32_t t = threadIdx.x;
    const uint32_t strided = (cnt + kSumBlock - 1) / kSumBlock;  // uniform over the block
    G1Xyzz28 acc = xyzz28_identity();
    const uint32_t steps = strided + 7;  // 7 = log2(kSumBlock) tree levels
    for (uint32_t step = 0; step < steps; step++) {
        const G1Xyzz28 *src = nullptr;
        if (step < strided) {
            uint32_t j = step * kSumBlock + t;
            if (j < cnt) src = at(j);
        } else {
            uint32_t stride = (kSumBlock / 2) >> (step - strided);
            if (step == strided) {
                sh[t] = acc;
                __syncthreads();
            }
            if (t < stride) src = &sh[t + stride];
        }
        if (src) {
            G1Xyzz28 o = *src;
            if (LAZY_IN && step < strided) o = xyzz28_canon(o);
            acc = add_inline(acc, o);
        }
        if (step >= strided) {
            __syncthreads();  // every read of this level is done
            sh[t] = acc;
            __syncthreads();
        }
    }
    return acc;
}

// One block per queued bucket (grid-stride over the queue).
__global__ __launch_bounds__(kSumBlock) void k_fixup_big(const uint32_t *__restrict__ start, uint32_t seg_len,
                                                         const G1Xyzz28 *__restrict__ part,
                                                         G1Xyzz28 *__restrict__ buckets,
                                                         const uint32_t *__restrict__ big_count,
                                                         const uint32_t *__restrict__ big_list) {
    __shared__ G1Xyzz28 sh[kSumBlock];
    const uint32_t count = *big_count;
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        uint32_t b = big_list[i];
        uint32_t lo = start[b], hi = start[b + 1];
        uint32_t s0 = lo / seg_len, s1 = (hi - 1) / seg_len;
        G1Xyzz28 acc =
            block_sum<true>(s1 - s0 + 1, sh, [&](uint32_t j) { return partial_of(part, s0, j, lo, seg_len); });
        if (threadIdx.x == 0) buckets[b] = acc;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// 6. reduction  sum_b (b + 1) B_b
// ------------------------------------------------------------------------------------------
// One lane per chunk of kChunk consecutive elements of A:
//   R[t]  = sum_j A[t*kChunk + j]
//   LW[t] = sum_j (j + off) A[t*kChunk + j]          (off = 1 on the bucket level, else 0)
// by the running-sum recurrence run += A[b]; sum += run, highest b first.  Both updates go
// through one inlined addition with selected operands.
__global__ __launch_bounds__(kBlock) void k_reduce_level(const G1Xyzz28 *__restrict__ A, uint32_t n, int off,
                                                         G1Xyzz28 *__restrict__ LW, G1Xyzz28 *__restrict__ R,
                                                         uint32_t nchunks) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    uint32_t lo = t * kChunk;
    uint32_t hi = lo + kChunk < n ? lo + kChunk : n;
    G1Xyzz28 run = xyzz28_identity(), sum = xyzz28_identity();
    const uint32_t steps = 2 * (hi - lo);
    for (uint32_t step = 0; step < steps; step++) {
        uint32_t b = hi - 1 - (step >> 1);
        bool to_run = (step & 1) == 0;
        if (!to_run && b == lo && !off) break;  // weight 0 for the chunk's first element
        G1Xyzz28 x = to_run ? run : sum;
        G1Xyzz28 y = to_run ? A[b] : run;
        G1Xyzz28 r = add_inline(x, y);
        if (to_run) {
            run = r;
        } else {
            sum = r;
        }
    }
    LW[t] = sum;
    R[t] = run;
}

struct SumDesc {
    uint32_t in_off[kMaxLevels];
    uint32_t n[kMaxLevels];
    uint32_t out_off[kMaxLevels];
};

static constexpr uint32_t kSumPerBlock = 4 * kSumBlock;

// Plain sums: block (x, level) adds elements [x*kSumPerBlock, (x+1)*kSumPerBlock) of that
// level's array.
__global__ __launch_bounds__(kSumBlock) void k_sum_pass(const G1Xyzz28 *__restrict__ in, G1Xyzz28 *__restrict__ out,
                                                        SumDesc d) {
    __shared__ G1Xyzz28 sh[kSumBlock];
    const int lev = blockIdx.y;
    const uint32_t n = d.n[lev];
    const uint32_t base = blockIdx.x * kSumPerBlock;
    if (base >= n) return;  // uniform per block
    const G1Xyzz28 *src = in + d.in_off[lev] + base;
    uint32_t cnt = n - base < kSumPerBlock ? n - base : kSumPerBlock;
    G1Xyzz28 acc = block_sum<false>(cnt, sh, [&](uint32_t j) { return src + j; });
    if (threadIdx.x == 0) out[d.out_off[lev] + blockIdx.x] = acc;
}

// Horner over the level sums: total = s_0 + 4 (s_1 + 4 (s_2 + ...)); Jacobian result.
__global__ void k_finish(const G1Xyzz28 *__restrict__ sums, SumDesc d, int nlev, G1Jac *__restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    G1Xyzz28 acc = sums[d.in_off[nlev - 1]];
    for (int l = nlev - 2; l >= 0; l--) {
        for (int k = 0; k < kChunkLog; k++) acc = xyzz28_dbl(acc);
        acc = add_inline(acc, sums[d.in_off[l]]);
    }
    *out = xyzz_to_jac(xyzz28_to(acc));
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
    } else {
        // Cost in "bucket additions": the accumulation runs ~2^18 lanes at once; the
        // reduction is a serial tail whose depth grows with c.
        double best_cost = 1e300;
        const double lanes = 262144.0;
        for (int c = 4; c <= 22; c++) {
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

int msm_run_dev(plonk_hip_ctx *ctx, const Fr *d_scalars, size_t n, G1Jac *d_out) {
    if (ctx->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    if (n == 0) return PLONK_HIP_ERR_EMPTY_POLY;
    if (n > ctx->srs_n) return PLONK_HIP_ERR_SRS_TOO_SHORT;
    const MsmPlan &pl = ctx->plan;
    const size_t total = n * (size_t)pl.nwin;
    if (total >= 0xfffffff0ull) return PLONK_HIP_ERR_INVALID_ARG;
    const uint32_t nb = pl.nbuckets;
    hipStream_t st = ctx->stream;

    // segment length: balance accumulation depth against fix-up depth for small inputs, keep
    // about 2^20 segments for large ones
    const double per_bucket = (double)total / nb;
    uint32_t seg_len = (uint32_t)std::max<double>(
        1.0, std::max(std::ceil(std::sqrt(per_bucket)), std::ceil((double)total / 1048576.0)));
    const size_t nseg = (total + seg_len - 1) / seg_len;

    size_t tmp_bytes = sort_pairs_temp_bytes(total, pl.c);
    PLONK_TRY(buf_reserve(ctx, ctx->keys_a, total * 4));
    PLONK_TRY(buf_reserve(ctx, ctx->keys_b, total * 4));
    PLONK_TRY(buf_reserve(ctx, ctx->vals_a, total * 4));
    PLONK_TRY(buf_reserve(ctx, ctx->vals_b, total * 4));
    PLONK_TRY(buf_reserve(ctx, ctx->sort_tmp, tmp_bytes));
    PLONK_TRY(buf_reserve(ctx, ctx->bucket_start, ((size_t)nb + 2) * 4));
    PLONK_TRY(buf_reserve(ctx, ctx->buckets, (size_t)nb * sizeof(G1Xyzz28)));
    PLONK_TRY(buf_reserve(ctx, ctx->seg_part, nseg * 2 * sizeof(G1Xyzz28)));
    PLONK_TRY(buf_reserve(ctx, ctx->big_list, ((size_t)nb + 1) * 4));

    uint32_t *keys_a = (uint32_t *)ctx->keys_a.ptr, *keys_b = (uint32_t *)ctx->keys_b.ptr;
    uint32_t *vals_a = (uint32_t *)ctx->vals_a.ptr, *vals_b = (uint32_t *)ctx->vals_b.ptr;
    uint32_t *start = (uint32_t *)ctx->bucket_start.ptr;
    G1Xyzz28 *buckets = (G1Xyzz28 *)ctx->buckets.ptr;
    G1Xyzz28 *part = (G1Xyzz28 *)ctx->seg_part.ptr;
    const G1Affine28 *table = (const G1Affine28 *)ctx->table.ptr;
    uint32_t *big_count = (uint32_t *)ctx->big_list.ptr, *big_list = big_count + 1;

    record(ctx, PH_DIGITS);
    hipLaunchKernelGGL(k_digits, dim3(cdiv(n, 256)), dim3(256), 0, st, d_scalars, (uint32_t)n, (uint32_t)ctx->srs_n,
                       pl.c, pl.nwin, nb, keys_a, vals_a);
    record(ctx, PH_SORT);
    PLONK_TRY(sort_pairs(ctx, keys_a, keys_b, vals_a, vals_b, total, pl.c, ctx->sort_tmp.ptr, tmp_bytes));
    record(ctx, PH_BOUNDS);
    hipLaunchKernelGGL(k_bounds, dim3(cdiv(total, 256)), dim3(256), 0, st, keys_b, total, nb, start);
    record(ctx, PH_ACCUM);
    hipLaunchKernelGGL(k_accumulate, dim3(cdiv(nseg, kBlock)), dim3(kBlock), 0, st, keys_b, vals_b, start, nb,
                       seg_len, table, buckets, part);
    record(ctx, PH_FIXUP);
    PLONK_HIP_TRY(ctx, hipMemsetAsync(big_count, 0, 4, st));
    // a typical bucket is cut into per_bucket / seg_len partials; only buckets well above
    // that (skew) take the block-per-bucket path
    const uint32_t serial_max = std::max<uint32_t>(16u, 2u * (uint32_t)std::ceil(per_bucket / seg_len) + 2u);
    hipLaunchKernelGGL(k_fixup, dim3(cdiv(nb, kBlock)), dim3(kBlock), 0, st, start, nb, seg_len, part, buckets,
                       serial_max, big_count, big_list);
    hipLaunchKernelGGL(k_fixup_big, dim3(std::min<uint32_t>(nb, 1024u)), dim3(kSumBlock), 0, st, start, seg_len, part,
                       buckets, big_count, big_list);
    record(ctx, PH_REDUCE);

    // ---- reduction levels
    uint32_t level_n[kMaxLevels + 1];
    int nlev = 0;
    level_n[0] = nb;
    size_t lw_total = 0;
    while (true) {
        uint32_t nchunks = (level_n[nlev] + kChunk - 1) / kChunk;
        level_n[nlev + 1] = nchunks;
        lw_total += nchunks;
        nlev++;
        if (nchunks <= 1) break;
    }
    if (nlev > kMaxLevels) return PLONK_HIP_ERR_INVALID_ARG;
    PLONK_TRY(buf_reserve(ctx, ctx->red_a, (size_t)level_n[1] * sizeof(G1Xyzz28)));
    PLONK_TRY(buf_reserve(ctx, ctx->red_b, (size_t)(nlev > 1 ? level_n[2] : 1) * sizeof(G1Xyzz28)));
    PLONK_TRY(buf_reserve(ctx, ctx->red_lw, lw_total * sizeof(G1Xyzz28)));
    G1Xyzz28 *lw = (G1Xyzz28 *)ctx->red_lw.ptr;
    SumDesc desc = {};
    {
        const G1Xyzz28 *A = buckets;
        uint32_t off = 0;
        for (int l = 0; l < nlev; l++) {
            G1Xyzz28 *Rbuf = (G1Xyzz28 *)((l & 1) ? ctx->red_b.ptr : ctx->red_a.ptr);
            uint32_t nchunks = level_n[l + 1];
            hipLaunchKernelGGL(k_reduce_level, dim3(cdiv(nchunks, kBlock)), dim3(kBlock), 0, st, A, level_n[l],
                               l == 0 ? 1 : 0, lw + off, Rbuf, nchunks);
            desc.in_off[l] = off;
            desc.n[l] = nchunks;
            off += nchunks;
            A = Rbuf;
        }
    }
    // ---- plain sums of every level's LW array
    size_t max_blocks_total = 0;
    for (int l = 0; l < nlev; l++) max_blocks_total += cdiv(desc.n[l], kSumPerBlock);
    PLONK_TRY(buf_reserve(ctx, ctx->red_sum_a, max_blocks_total * sizeof(G1Xyzz28)));
    PLONK_TRY(buf_reserve(ctx, ctx->red_sum_b, max_blocks_total * sizeof(G1Xyzz28)));
    const G1Xyzz28 *in = lw;
    G1Xyzz28 *outs[2] = {(G1Xyzz28 *)ctx->red_sum_a.ptr, (G1Xyzz28 *)ctx->red_sum_b.ptr};
    int flip = 0;
    while (true) {
        uint32_t max_n = 0, max_blocks = 0, off = 0;
        for (int l = 0; l < nlev; l++) {
            uint32_t blocks = cdiv(desc.n[l], kSumPerBlock);
            desc.out_off[l] = off;
            off += blocks;
            max_n = std::max(max_n, desc.n[l]);
            max_blocks = std::max(max_blocks, blocks);
        }
        if (max_n <= 1) break;
        hipLaunchKernelGGL(k_sum_pass, dim3(max_blocks, nlev), dim3(kSumBlock), 0, st, in, outs[flip], desc);
        for (int l = 0; l < nlev; l++) {
            desc.in_off[l] = desc.out_off[l];
            desc.n[l] = cdiv(desc.n[l], kSumPerBlock);
        }
        in = outs[flip];
        flip ^= 1;
    }
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(1), 0, st, in, desc, nlev, d_out);
    record(ctx, PH_COUNT);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

}  // namespace plonk
