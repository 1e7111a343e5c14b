// Fr number-theoretic transform over Omega = <omega>, omega = 7^((r-1)/n), and the coset
// quotient by Z_Omega(X) = X^k - 1.
//
// A transform of 2^log_n points runs as ceil(log_n / 10) passes.  A pass owns the butterfly
// stages [s0, s1): one workgroup loads a tile of 2^(s1-s0) elements
//     i(j) = (t_hi << s1) | (j << s0) | t_lo,   j = 0 .. 2^(s1-s0) - 1
// into LDS, runs its stages there and writes the tile back, so HBM is touched once per pass.
// Two butterfly orders share the tiling:
//   DIF  natural order in  -> bit-reversed order out   (stages high to low)
//   DIT  bit-reversed in   -> natural order out        (stages low to high)
// The public transform is natural order on both sides (DIF + bit-reversal, or bit-reversal
// + DIT); the quotient pipeline chains DIF -> pointwise -> DIT and never permutes.
// Inside a pass the butterflies run on 29-bit limbs (field29.cuh) with lazy reduction; every lane
// keeps 8 elements in registers and does up to 3 stages (12 butterflies) per LDS round trip.
//
// Twiddles (four-step form).  A pass over [s0, s1) is a plain transform of 2^(s1-s0) points on
// every tile: its butterflies use the roots of unity of THAT size, which depend on the in-tile
// index j only (a table of 2^(s1-s0-1) entries that stays in the caches; the stages on the three
// lowest in-tile bits use 1, omega_4, omega_8 and omega_8^3 from scalar registers, and the
// multiplications by 1 are skipped: 7 of every 40 butterflies of a 10-stage pass).  What the true
// twiddle omega_(2^s1)^((j_low << s0 | t_lo) << ...) has on top of that is collected into ONE factor
// per element, omega_(2^s1)^(t_lo * rev(j)), applied when a strided pass stores (DIF) or, for DIT, when the
// pass before it stores.  Those factors come from a table laid out like the data ((j << s0) | t_lo), so they are
// read in the same contiguous rows as the data, and a batch is dealt to the eight XCDs by tile
// rather than by vector, so that one XCD's L2 serves a table row to all vectors of the batch.
// A coset shift (f(gX) before a forward transform, g^-i / n after an inverse one) is one more
// table of per-element factors applied at the first load or the last store: no separate pass.
// All tables hold w * 2^261 in 29-bit limbs and are cached in the context per size and direction.
#include "context.hpp"
#define PLONK_MONT29_ASM 1  // the transforms' products as one hand-scheduled sequence
#include "field29.cuh"
#include "poly.hpp"

namespace plonk {

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------
// host-side field helpers (same __host__ __device__ arithmetic as the kernels)
// ------------------------------------------------------------------------------------------
static Fr fr_from_u64(uint64_t v) {
    Fr c = fe_zero<FrCfg>();
    c.l[0] = (uint32_t)v;
    c.l[1] = (uint32_t)(v >> 32);
    return fe_to_mont(c);
}

// omega_n = 7^((r-1) >> log_n)   (common/utils.rs:11-38 of the reference for k = 2^log_n)
static Fr fr_root_of_unity(uint32_t log_n) {
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = FrCfg::mod(i);
    e[0] -= 1;  // r - 1 (r is odd, no borrow)
    // shift right by log_n (< 32)
    if (log_n) {
        for (int i = 0; i < 8; i++) e[i] = (e[i] >> log_n) | (i + 1 < 8 ? e[i + 1] << (32 - log_n) : 0u);
    }
    Fr base = fr_from_u64(7), acc = fe_one<FrCfg>();
    for (int i = 7; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            acc = fe_sqr(acc);
            if ((e[i] >> b) & 1) acc = fe_mul(acc, base);
        }
    return acc;
}

// ------------------------------------------------------------------------------------------
// power tables
// ------------------------------------------------------------------------------------------
struct PowSeed {
    Fr pw[32];  // base^(2^k)
};
// out[i] = base^i, i < count (square and multiply per entry; tables are small)
__global__ __launch_bounds__(256) void k_pow_table(PowSeed seed, uint32_t count, Fr *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fr acc = fe_one<FrCfg>();
#pragma unroll 1
    for (int k = 0; k < 32; k++) {
        if ((i >> k) & 1) acc = fe_mul(acc, seed.pw[k]);
    }
    out[i] = acc;
}
// out[e] = lo[e & 1023] * hi[e >> 10]
__global__ __launch_bounds__(256) void k_pow_combine(const Fr *__restrict__ lo, const Fr *__restrict__ hi, size_t count,
                                                    Fr *__restrict__ out) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    Fr v = lo[e & 1023];
    if (e >> 10) v = fe_mul(v, hi[e >> 10]);
    out[e] = v;
}
// twiddle entry: 9 x 29-bit limbs padded to 48 bytes (three 16-byte loads)
struct alignas(16) Tw29 {
    Fr29 v;
    uint32_t pad[3];
};
// out[e] = (lo[e & 1023] * hi[e >> 10]) * 2^261 in 29-bit limbs
__global__ __launch_bounds__(256) void k_twiddle29(const Fr *__restrict__ lo, const Fr *__restrict__ hi, size_t count,
                                                  Tw29 *__restrict__ out) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    Fr v = lo[e & 1023];
    if (e >> 10) v = fe_mul(v, hi[e >> 10]);
    Tw29 t;
    t.v = fr29_twiddle_from(v);
    t.pad[0] = t.pad[1] = t.pad[2] = 0;
    out[e] = t;
}
// the same with every entry multiplied by `factor` (a coset table with 1/n folded in)
__global__ __launch_bounds__(256) void k_twiddle29_scaled(const Fr *__restrict__ lo, const Fr *__restrict__ hi,
                                                         size_t count, Fr factor, Tw29 *__restrict__ out) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    Fr v = fe_mul(lo[e & 1023], factor);
    if (e >> 10) v = fe_mul(v, hi[e >> 10]);
    Tw29 t;
    t.v = fr29_twiddle_from(v);
    t.pad[0] = t.pad[1] = t.pad[2] = 0;
    out[e] = t;
}
// Correction factors of a strided pass over the stages [s0, s1): out[(j << s0) | t] = w^(t * rev(j)),
// w the primitive 2^s1-th root the lo / hi tables are powers of, rev over the s1 - s0 bits of j.
// With `scaled` the rows j of the lower half are multiplied by f_lower and those of the upper half by
// f_upper on top of that (a 1/n, or the two values of 1/Z on a coset, folded into the same product).
__global__ __launch_bounds__(256) void k_correction29(const Fr *__restrict__ lo, const Fr *__restrict__ hi, uint32_t s0,
                                                     uint32_t s1, int scaled, Fr f_lower, Fr f_upper,
                                                     Tw29 *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> s1) return;
    const uint32_t k = s1 - s0;
    const uint64_t t = i & (((size_t)1 << s0) - 1);
    const uint64_t jr = __brev((uint32_t)(i >> s0)) >> (32 - k);
    const uint64_t e = (t * jr) & (((uint64_t)1 << s1) - 1);
    Fr v = lo[e & 1023];
    if (e >> 10) v = fe_mul(v, hi[e >> 10]);
    if (scaled) v = fe_mul(v, (i >> (s1 - 1)) ? f_upper : f_lower);
    Tw29 o;
    o.v = fr29_twiddle_from(v);
    o.pad[0] = o.pad[1] = o.pad[2] = 0;
    out[i] = o;
}
// out[i] = base^i * 2^261 in 29-bit limbs (the multiplier form of the lazy product)
__global__ __launch_bounds__(256) void k_pow_table29(PowSeed seed, uint32_t count, Tw29 *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fr acc = fe_one<FrCfg>();
#pragma unroll 1
    for (int k = 0; k < 32; k++) {
        if ((i >> k) & 1) acc = fe_mul(acc, seed.pw[k]);
    }
    Tw29 t;
    t.v = fr29_twiddle_from(acc);
    t.pad[0] = t.pad[1] = t.pad[2] = 0;
    out[i] = t;
}
// data[b][i] *= lo[i & 1023] * hi[i >> 10] for the vectors b of one batch group: the factor is
// formed once per lane and reused for every vector of the group.  Inputs canonical, outputs canonical.
static constexpr unsigned kScaleGroup = 8;
__global__ __launch_bounds__(256) void k_scale_two_level(Fr *__restrict__ data, size_t n, size_t batch_stride,
                                                        unsigned batch, const Tw29 *__restrict__ lo,
                                                        const Tw29 *__restrict__ hi) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr29 v = lo[i & 1023].v;
    if (i >> 10) v = fr29_cond_sub(fr29_mul_lazy(v, hi[i >> 10].v));
    const unsigned b0 = blockIdx.y * kScaleGroup, b1 = min(batch, b0 + kScaleGroup);
    for (unsigned b = b0; b < b1; b++) {
        Fr *p = data + (size_t)b * batch_stride + i;
        *p = fr29_unpack(fr29_cond_sub(fr29_mul_lazy(fr29_pack(*p), v)));
    }
}

static PowSeed make_seed(const Fr &base) {
    PowSeed s;
    s.pw[0] = base;
    for (int k = 1; k < 32; k++) s.pw[k] = fe_sqr(s.pw[k - 1]);
    return s;
}

// lo = base^i (i < 1024), hi = (base^1024)^i (i < ceil(count / 1024)) in ctx->scratch_b.
static int build_two_level(plonk_hip_ctx *ctx, const Fr &base, size_t count, const Fr **lo, const Fr **hi) {
    size_t nhi = (count + 1023) / 1024;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_b, (1024 + nhi) * sizeof(Fr)));
    Fr *d_lo = (Fr *)ctx->scratch_b.ptr, *d_hi = d_lo + 1024;
    PowSeed s = make_seed(base);
    hipLaunchKernelGGL(k_pow_table, dim3(4), dim3(256), 0, ctx->stream, s, 1024u, d_lo);
    PowSeed sh = make_seed(s.pw[10]);
    hipLaunchKernelGGL(k_pow_table, dim3(cdiv(nhi, 256)), dim3(256), 0, ctx->stream, sh, (uint32_t)nhi, d_hi);
    *lo = d_lo;
    *hi = d_hi;
    return PLONK_HIP_OK;
}
// the same tables in the 29-bit multiplier form
static int build_two_level29(plonk_hip_ctx *ctx, const Fr &base, size_t count, const Tw29 **lo, const Tw29 **hi) {
    size_t nhi = (count + 1023) / 1024;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_b, (1024 + nhi) * sizeof(Tw29)));
    Tw29 *d_lo = (Tw29 *)ctx->scratch_b.ptr, *d_hi = d_lo + 1024;
    PowSeed s = make_seed(base);
    hipLaunchKernelGGL(k_pow_table29, dim3(4), dim3(256), 0, ctx->stream, s, 1024u, d_lo);
    PowSeed sh = make_seed(s.pw[10]);
    hipLaunchKernelGGL(k_pow_table29, dim3(cdiv(nhi, 256)), dim3(256), 0, ctx->stream, sh, (uint32_t)nhi, d_hi);
    *lo = d_lo;
    *hi = d_hi;
    return PLONK_HIP_OK;
}

int scale_by_powers(plonk_hip_ctx *ctx, Fr *d_data, size_t n, size_t batch, const Fr &w, size_t stride) {
    if (stride == 0) stride = n;
    if (n == 0 || batch == 0) return PLONK_HIP_OK;
    if (n > 0xffffffffull || batch > 65535) return PLONK_HIP_ERR_INVALID_ARG;
    const Tw29 *lo, *hi;
    PLONK_TRY(build_two_level29(ctx, w, n, &lo, &hi));
    hipLaunchKernelGGL(k_scale_two_level, dim3(cdiv(n, 256), cdiv(batch, kScaleGroup)), dim3(256), 0, ctx->stream,
                       d_data, n, stride, (unsigned)batch, lo, hi);
    return PLONK_HIP_OK;
}

static int get_twiddles(plonk_hip_ctx *ctx, uint32_t log_n, bool inverse, const Tw29 **out) {
    DevBuf &b = ctx->twiddle[inverse ? 1 : 0][log_n];
    size_t count = log_n ? ((size_t)1 << (log_n - 1)) : 1;
    if (!b.ptr) {
        Fr w = fr_root_of_unity(log_n);
        if (inverse) w = fe_inv(w);
        const Fr *lo, *hi;
        PLONK_TRY(build_two_level(ctx, w, count, &lo, &hi));
        PLONK_TRY(buf_reserve(ctx, b, count * sizeof(Tw29)));
        hipLaunchKernelGGL(k_twiddle29, dim3(cdiv(count, 256)), dim3(256), 0, ctx->stream, lo, hi, count,
                           (Tw29 *)b.ptr);
    }
    *out = (const Tw29 *)b.ptr;
    return PLONK_HIP_OK;
}

// w_(2^s1)^(t * rev(j)) at (j << s0) | t for the strided pass over [s0, s1).  Kinds:
//   kCorrForward / kCorrInverse   the plain factors for w = omega or omega^-1
//   kCorrInverseScaled            omega^-1 factors times 2^-s1: an inverse transform whose last pass is
//                                 this one needs no separate multiplication by 1/n
//   kCorrForwardOverZ             omega factors times 1 / (g^k (-1)^e - 1), k = 2^(s1-1), g = 7: the values of
//                                 1 / (X^k - 1) on the coset g * <omega>; the exponent e of the point an
//                                 element ends up at is even exactly for the rows j of the lower half
enum CorrKind { kCorrForward = 0, kCorrInverse = 1, kCorrInverseScaled = 2, kCorrForwardOverZ = 3 };
static int get_correction(plonk_hip_ctx *ctx, uint32_t s0, uint32_t s1, CorrKind kind, const Fr *f_lower,
                          const Fr *f_upper, const Tw29 **out) {
    DevBuf &b = ctx->ntt_corr[kind][s1][s0];
    const size_t count = (size_t)1 << s1;
    if (!b.ptr) {
        Fr w = fr_root_of_unity(s1);
        if (kind == kCorrInverse || kind == kCorrInverseScaled) w = fe_inv(w);
        const Fr *lo, *hi;
        PLONK_TRY(build_two_level(ctx, w, count, &lo, &hi));
        PLONK_TRY(buf_reserve(ctx, b, count * sizeof(Tw29)));
        const Fr one = fe_one<FrCfg>();
        hipLaunchKernelGGL(k_correction29, dim3(cdiv(count, 256)), dim3(256), 0, ctx->stream, lo, hi, s0, s1,
                           f_lower ? 1 : 0, f_lower ? *f_lower : one, f_upper ? *f_upper : one, (Tw29 *)b.ptr);
    }
    *out = (const Tw29 *)b.ptr;
    return PLONK_HIP_OK;
}

// Coset factors for vectors of 2^log_n elements: pre[i] = shift^i (forward transform of f(shift X)),
// post[i] = shift^-i / n (after an inverse transform).  One shift is cached per size and direction.
static int get_coset_table(plonk_hip_ctx *ctx, uint32_t log_n, const Fr &shift, bool inverse, const Tw29 **out) {
    const int d = inverse ? 1 : 0;
    DevBuf &b = ctx->coset_tab[d][log_n];
    const size_t n = (size_t)1 << log_n;
    if (!b.ptr || !ctx->coset_ok[d][log_n] || !fe_eq(ctx->coset_shift[d][log_n], shift)) {
        ctx->coset_ok[d][log_n] = false;
        const Fr *lo, *hi;
        PLONK_TRY(build_two_level(ctx, inverse ? fe_inv(shift) : shift, n, &lo, &hi));
        PLONK_TRY(buf_reserve(ctx, b, n * sizeof(Tw29)));
        if (inverse) {
            hipLaunchKernelGGL(k_twiddle29_scaled, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, lo, hi, n,
                               fe_inv(fr_from_u64(n)), (Tw29 *)b.ptr);
        } else {
            hipLaunchKernelGGL(k_twiddle29, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, lo, hi, n, (Tw29 *)b.ptr);
        }
        ctx->coset_shift[d][log_n] = shift;
        ctx->coset_ok[d][log_n] = true;
    }
    *out = (const Tw29 *)b.ptr;
    return PLONK_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// butterfly passes
// ------------------------------------------------------------------------------------------
// LDS image of a tile: element j at word 9*j + (j >> 5) (odd stride plus one pad word per 32
// elements keeps the strided accesses of every round spread over the banks).
__device__ __forceinline__ uint32_t lds_word(uint32_t j) { return 9u * j + (j >> 5); }
__device__ __forceinline__ Fr29 lds_get(const uint32_t *tile, uint32_t j) {
    Fr29 v;
    const uint32_t *p = tile + lds_word(j);
#pragma unroll
    for (int i = 0; i < R29::N; i++) v.l[i] = p[i];
    return v;
}
__device__ __forceinline__ void lds_put(uint32_t *tile, uint32_t j, const Fr29 &v) {
    uint32_t *p = tile + lds_word(j);
#pragma unroll
    for (int i = 0; i < R29::N; i++) p[i] = v.l[i];
}

// One round: G butterfly stages on the in-tile bits [b0, b0 + G) of a tile of 2^k elements.  A call
// handles NE elements: NE / 2^G groups of 2^G elements whose tile indices differ exactly in those
// bits; `item` is the work item's index inside the tile (tile_n / 8 of them).  Stage b0 + bb of a
// group multiplies by w^((j mod 2^(b0 + bb)) << (k - 1 - b0 - bb)), w the primitive 2^k-th root the
// table `tw` holds the powers of.
// Z marks the round on the lowest bits (b0 == 0): there j mod 2^bb is the group position itself, so
// the factors are the same for every lane (scalar loads) and those with exponent 0 are skipped.
//
// Bounds (units of 2^255, capacity 64; r = 0.906).
// DIF: every element enters a round below B = 4.2; sums double per stage (8 B = 33.6 at most), a
//   subtrahend is at most 4 B < 19 r, a product is below 1.4, and at the end of the round every value
//   that is not a fresh product takes one reduction step, which brings it below 1 + 0.0943 * 33 < B.
//   A difference that is NOT multiplied (Z round, exponent 0) is below B' + 17.2 with B' the bound
//   of its operands (4.2, 8.4 or 16.8): it is reduced at once (below 3.4) except after the last
//   stage, where the round's closing reduction takes care of it.
// DIT: elements enter the first (Z) round below 2 (fresh from memory, or a product); its unmultiplied
//   differences add 3 r, 6 r, 12 r (each at least the subtrahend's bound 2, 4.72, 10.15): below 21.1
//   after the round.  Every later stage adds at most 2 r = 1.81: below 36 after 11 stages.
template <int BB>
struct DitZeroK {
    static constexpr int value = BB == 0 ? 3 : (BB == 1 ? 6 : 12);
};

template <bool DIF, int G, int NE, bool Z>
__device__ __forceinline__ void ntt_round(uint32_t *tile, uint32_t item, uint32_t b0, uint32_t k,
                                          const Tw29 *__restrict__ tw) {
    constexpr uint32_t GROUP = 1u << G, GPT = (uint32_t)NE >> G, TW_PER_GROUP = GROUP - 1;
    Fr29 x[NE];
    uint32_t jj[NE];
    const uint32_t low_mask = (1u << b0) - 1;
#pragma unroll
    for (uint32_t u = 0; u < GPT; u++) {
        const uint32_t gid = item * GPT + u;
#pragma unroll
        for (uint32_t c = 0; c < GROUP; c++) {
            const uint32_t j = ((gid >> b0) << (b0 + G)) | (c << b0) | (gid & low_mask);
            jj[u * GROUP + c] = j;
            x[u * GROUP + c] = lds_get(tile, j);
        }
    }
    // Twiddles of the whole round, requested up front so that their latencies overlap: stage bb
    // of a group needs 2^bb distinct factors (one per value of the group bits below bb).  In the Z
    // round they do not depend on the lane: one set for all groups, from uniform addresses.
    Fr29 twr[Z ? TW_PER_GROUP : GPT * TW_PER_GROUP];
#pragma unroll
    for (uint32_t u = 0; u < (Z ? 1u : GPT); u++) {
#pragma unroll
        for (int bb = 0; bb < G; bb++) {
            const uint32_t emask = (1u << (b0 + bb)) - 1;
#pragma unroll
            for (uint32_t cl = 0; cl < (1u << bb); cl++) {
                if (Z && cl == 0) continue;  // exponent 0
                uint32_t e;
                if (Z) {
                    e = (uint32_t)__builtin_amdgcn_readfirstlane((int)(cl << (k - 1 - bb)));
                } else {
                    e = (jj[u * GROUP + cl] & emask) << (k - 1 - b0 - bb);
                }
                twr[u * TW_PER_GROUP + ((1u << bb) - 1) + cl] = tw[e].v;
            }
        }
    }
#pragma unroll
    for (int step = 0; step < G; step++) {
        const int bb = DIF ? (G - 1 - step) : step;
#pragma unroll
        for (uint32_t u = 0; u < GPT; u++) {
#pragma unroll
            for (uint32_t c = 0; c < GROUP; c++) {
                if (c & (1u << bb)) continue;
                const uint32_t lo = u * GROUP + c, hi = lo | (1u << bb);
                const uint32_t cl = c & ((1u << bb) - 1);
                if (Z && cl == 0) {
                    // factor 1
                    if (DIF) {
                        Fr29 sum = fr29_add_lazy(x[lo], x[hi]);
                        Fr29 d = fr29_sub_lazy<19>(x[lo], x[hi]);
                        x[hi] = bb == 0 ? d : fr29_reduce_step(d);
                        x[lo] = sum;
                    } else {
                        Fr29 t = x[hi];
                        if (bb == 0) x[hi] = fr29_sub_lazy<DitZeroK<0>::value>(x[lo], t);
                        if (bb == 1) x[hi] = fr29_sub_lazy<DitZeroK<1>::value>(x[lo], t);
                        if (bb == 2) x[hi] = fr29_sub_lazy<DitZeroK<2>::value>(x[lo], t);
                        x[lo] = fr29_add_lazy(x[lo], t);
                    }
                    continue;
                }
                const Fr29 &w = twr[(Z ? 0u : u * TW_PER_GROUP) + ((1u << bb) - 1) + cl];
                if (DIF) {
                    Fr29 sum = fr29_add_lazy(x[lo], x[hi]);
                    x[hi] = fr29_mul_lazy(fr29_sub_lazy<19>(x[lo], x[hi]), w);
                    x[lo] = sum;
                } else {
                    Fr29 t = fr29_mul_lazy(x[hi], w);
                    x[hi] = fr29_sub_lazy<2>(x[lo], t);
                    x[lo] = fr29_add_lazy(x[lo], t);
                }
            }
        }
    }
    if (DIF) {
        // what is not a fresh product: the sums of the last stage (group bit 0 clear), and in the Z
        // round its differences as well
#pragma unroll
        for (uint32_t e = 0; e < (uint32_t)NE; e += (Z ? 1 : 2)) x[e] = fr29_reduce_step(x[e]);
    }
#pragma unroll
    for (uint32_t e = 0; e < (uint32_t)NE; e++) lds_put(tile, jj[e], x[e]);
}

struct PassArgs {
    size_t batch_stride;
    uint32_t log_n, s0, s1, tpb_log;
    uint32_t batch, tile_blocks;  // vectors in this launch; blocks per vector
    int do_scale, canonical_out, in_map, out_map, xcd_map;
    uint32_t xcd_group_log;  // with xcd_map: 2^xcd_group_log neighbouring tile blocks stay on one XCD
    Fr29 scale, scale_upper;
    // DIT: the factors of the NEXT pass (stages [next_s0, next_s1), its load map next_map) ride on this pass's store
    uint32_t next_s0, next_s1;
    int next_map;
};

// One pass over the stages [s0, s1) (3 <= s1 - s0 <= 11).  A block owns `tiles_per_block`
// consecutive tiles (several when tiles are small, so that all lanes have work).  Elements
// cross HBM as 8 x 32-bit limbs (values below 2^256; canonical when `canonical_out` is set, as
// on the last pass).
//   tw    powers of the primitive 2^(s1-s0)-th root (direction as the transform's)
//   corr  strided DIF passes: the per-element factors described at the top of the file, at (j << s0) | t_lo,
//         multiplied in when the pass stores; null otherwise
//   corr_next  DIT: the factors of the strided pass that comes NEXT, multiplied in when this pass stores (the
//         product is below 2 r, so it fits the 256-bit format without the two reduction steps a plain store
//         needs, and the next pass loads plain values); the element stored at address A is element
//         j = row(A) (bit-reversed when the next pass loads through map 1) of its tile there; null on the last pass
//   pre   a factor per input element, by address (coset shift of a forward transform), or null
//   post  a factor per output element, by address (coset shift and 1/n of an inverse transform), or null
// `scale` (w * 2^261 form) multiplies every output when do_scale is 1 and post is null; with do_scale 2
// (only on the pass with s0 = 0) the upper half of each vector takes `scale_upper`.
// Grid: tile_blocks * batch blocks.  With xcd_map the block index is read as (slot, xcd) = (L / 8,
// L % 8), slot = (group * batch + vector) * G + g, and the block works on tile block (8 * group + xcd) * G + g:
// blocks that the hardware deals to one XCD (L % 8 equal) and that follow each other then share their
// table rows (the same tile block of every vector) and their memory lines (G neighbouring tile blocks:
// a strided pass touches rows of 32 bytes per column of the block, and two blocks that split a 128-byte
// line between them would each fetch all of it if they sat on XCDs with separate L2s).
template <bool DIF, int GMAX>
__global__ __launch_bounds__(256, GMAX == 2 ? 4 : 2) void k_ntt_pass(const Fr *src, Fr *dst, const Tw29 *__restrict__ tw,
                                                  const Tw29 *__restrict__ corr, const Tw29 *__restrict__ corr_next,
                                                  const Tw29 *__restrict__ pre, const Tw29 *__restrict__ post,
                                                  PassArgs a) {
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t log_n = a.log_n, s0 = a.s0, s1 = a.s1, tpb_log = a.tpb_log;
    const uint32_t k = s1 - s0;
    const uint32_t tile_n = 1u << k;
    const uint32_t tile_words = 9u * tile_n + (tile_n >> 5) + 1;
    const uint32_t block_elems = tile_n << tpb_log;
    uint32_t tile_block, vec_index;
    if (a.xcd_map) {
        const uint32_t slot = blockIdx.x >> 3, rest = slot >> a.xcd_group_log;
        tile_block = ((((rest / a.batch) << 3) | (blockIdx.x & 7)) << a.xcd_group_log) | (slot & ((1u << a.xcd_group_log) - 1));
        vec_index = rest % a.batch;
    } else {
        tile_block = blockIdx.x % a.tile_blocks;
        vec_index = blockIdx.x / a.tile_blocks;
    }
    const Fr *vec_in = src + (size_t)vec_index * a.batch_stride;
    Fr *vec = dst + (size_t)vec_index * a.batch_stride;
    const size_t first_tile = (size_t)tile_block << tpb_log;
    const size_t s0_mask = ((size_t)1 << s0) - 1;
    const int in_map = a.in_map, out_map = a.out_map;

    // Neighbouring lanes take neighbouring addresses: in the pass with s0 = 0 a tile is contiguous
    // (j runs fastest); in a strided pass the tiles of a block are adjacent columns (the tile
    // index runs fastest), so that each row of the block is one contiguous piece of memory.
    // Address maps (in_map for the load, out_map for the store) fold the bit-reversal permutation
    // of a natural-order transform into its two passes:
    //   0  plain: element j of tile T
    //   1  the tile's rows in bit-reversed order (strided pass)
    //   2  pass with s0 = 0 only: element j of tile T lives at (rev(j) << (log_n - k)) | T
    auto address = [&](int map, size_t T, uint32_t j) -> size_t {
        const uint32_t jr = map ? (__brev(j) >> (32 - k)) : j;
        if (map == 2) return ((size_t)jr << (log_n - k)) | T;
        return ((T >> s0) << s1) | ((size_t)jr << s0) | (T & s0_mask);
    };
    const bool tile_fastest_in = s0 != 0 || in_map == 2, tile_fastest_out = s0 != 0 || out_map == 2;
    // element q of the block in load order: its place in memory, and its way into the LDS image with the
    // factors that ride on a load
    auto take = [&](const Fr &raw, uint32_t q) {
        const uint32_t tb = tile_fastest_in ? q & ((1u << tpb_log) - 1) : q >> k;
        const uint32_t j = tile_fastest_in ? q >> tpb_log : q & (tile_n - 1);
        const size_t T = first_tile + tb;
        Fr29 v = fr29_pack(raw);
        if (pre) v = fr29_mul_lazy(v, pre[address(in_map, T, j)].v);
        lds_put(lds + tb * tile_words, j, v);
    };
    auto where = [&](uint32_t q) -> size_t {
        const uint32_t tb = tile_fastest_in ? q & ((1u << tpb_log) - 1) : q >> k;
        const uint32_t j = tile_fastest_in ? q >> tpb_log : q & (tile_n - 1);
        return address(in_map, first_tile + tb, j);
    };
    // (One load in flight per thread: with all eight of a thread issued together, blocks that share 128-byte
    // lines ask for them at the same moment and both fetch them -- a quarter more read traffic for 2 % of time.)
    for (uint32_t q = threadIdx.x; q < block_elems; q += blockDim.x) take(vec_in[where(q)], q);
    __syncthreads();

    // a work item is 2^GMAX elements: GMAX = 3 keeps more stages in registers per LDS round trip,
    // GMAX = 2 halves the register footprint (four waves per SIMD instead of two)
    constexpr uint32_t NE = 1u << GMAX;
    const uint32_t items_per_tile = tile_n >> GMAX, items = items_per_tile << tpb_log;
    const uint32_t nrounds = (k + GMAX - 1) / GMAX;
    for (uint32_t r = 0; r < nrounds; r++) {
        // round q works on the in-tile bits [GMAX q, GMAX q + g): DIT walks them upwards, DIF
        // downwards; the round on the highest bits takes what is left (g <= GMAX)
        const uint32_t b0 = GMAX * (DIF ? nrounds - 1 - r : r);
        const uint32_t g = k - b0 < (uint32_t)GMAX ? k - b0 : (uint32_t)GMAX;
        for (uint32_t w = threadIdx.x; w < items; w += blockDim.x) {
            const uint32_t tb = w / items_per_tile, item = w % items_per_tile;
            uint32_t *tile = lds + tb * tile_words;
            if (GMAX == 3 && g == 3) {
                if (b0 == 0) {
                    ntt_round<DIF, 3, 8, true>(tile, item, 0, k, tw);
                } else {
                    ntt_round<DIF, 3, 8, false>(tile, item, b0, k, tw);
                }
            } else if (g == 2) {
                // groups of four, one after the other (for GMAX = 3 this keeps the register
                // footprint of the rarely taken branch below that of the three-stage round)
#pragma unroll 1
                for (uint32_t h = 0; h < NE / 4; h++)
                    ntt_round<DIF, 2, 4, false>(tile, (NE / 4) * item + h, b0, k, tw);
            } else {
                ntt_round<DIF, 1, (int)NE, false>(tile, item, b0, k, tw);
            }
        }
        __syncthreads();
    }

    for (uint32_t q = threadIdx.x; q < block_elems; q += blockDim.x) {
        const uint32_t tb = tile_fastest_out ? q & ((1u << tpb_log) - 1) : q >> k;
        const uint32_t j = tile_fastest_out ? q >> tpb_log : q & (tile_n - 1);
        const size_t T = first_tile + tb;
        const size_t idx = address(out_map, T, j);
        Fr29 v = lds_get(lds + tb * tile_words, j);
        bool product = false;  // a product is below 2 r
        if (DIF && corr) {
            v = fr29_mul_lazy(v, corr[((size_t)j << s0) | (T & s0_mask)].v);
            product = true;
        }
        if (!DIF && corr_next) {
            const uint32_t kn = a.next_s1 - a.next_s0;
            const uint32_t row = (uint32_t)(idx >> a.next_s0) & ((1u << kn) - 1);
            const uint32_t jn = a.next_map ? (__brev(row) >> (32 - kn)) : row;
            const size_t at = (((size_t)jn << a.next_s0) | (idx & (((size_t)1 << a.next_s0) - 1))) &
                              (((size_t)1 << a.next_s1) - 1);
            v = fr29_mul_lazy(v, corr_next[at].v);
            product = true;
        }
        if (post) {
            v = fr29_mul_lazy(v, post[idx].v);
        } else if (a.do_scale == 2) {
            v = fr29_mul_lazy(v, (idx >> (log_n - 1)) ? a.scale_upper : a.scale);
        } else if (a.do_scale) {
            v = fr29_mul_lazy(v, a.scale);
        } else if (!product) {
            v = fr29_reduce_below_2_256(v);
        }
        if (a.canonical_out) v = fr29_cond_sub(v);
        vec[idx] = fr29_unpack(v);
    }
}

// Transforms of 2 or 4 points (tiles too small for the register rounds): one lane per vector,
// natural order in and out, on the 8 x 32-bit Montgomery form; w4 is the primitive 4th root.
__global__ void k_ntt_tiny(Fr *__restrict__ data, uint32_t log_n, size_t batch, Fr w4, Fr scale, int do_scale) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    Fr *v = data + (b << log_n);
    if (log_n == 1) {
        Fr x = v[0], y = v[1];
        v[0] = fe_add(x, y);
        v[1] = fe_sub(x, y);
    } else {
        // natural order in and out: X_j = sum_i x_i w4^(ij)
        Fr x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
        Fr e0 = fe_add(x0, x2), e1 = fe_sub(x0, x2);
        Fr o0 = fe_add(x1, x3), o1 = fe_mul(fe_sub(x1, x3), w4);
        v[0] = fe_add(e0, o0);
        v[1] = fe_add(e1, o1);
        v[2] = fe_sub(e0, o0);
        v[3] = fe_sub(e1, o1);
    }
    if (do_scale) {
        for (uint32_t i = 0; i < (1u << log_n); i++) v[i] = fe_mul(v[i], scale);
    }
}

// in-place bit-reversal permutation
__global__ __launch_bounds__(256) void k_bitrev_swap(Fr *__restrict__ data, uint32_t log_n, size_t batch_stride) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << log_n)) return;
    uint32_t r = __brev((uint32_t)i) >> (32 - log_n);
    if (i < r) {
        Fr *p = data + (size_t)blockIdx.y * batch_stride;
        Fr a = p[i], b = p[r];
        p[i] = b;
        p[r] = a;
    }
}

// Evaluations on the coset g*<omega_2k> in bit-reversed order: position p holds the point
// g * omega^rev(p); (g omega^e)^k - 1 = g^k (-1)^e - 1 and e is even exactly when p < k.
__global__ __launch_bounds__(256) void k_mul_halves(Fr *__restrict__ data, size_t n, size_t batch_stride, Fr lo_half,
                                                   Fr hi_half) {
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    Fr *q = data + (size_t)blockIdx.y * batch_stride + p;
    *q = fe_mul(*q, p < n / 2 ? lo_half : hi_half);
}

static constexpr int kMaxTileLog = 11;  // 2048 elements = 74 KiB of LDS

struct PassPlan {
    int npass;
    uint32_t s0[6], s1[6];
};
// log_n >= 3.  The pass over the low stages works on contiguous tiles and takes up to 11 stages.
// The stages above it are strided: their blocks hold several adjacent columns so that memory is
// still touched in contiguous rows, which limits a strided pass to 9 stages (4 columns of 512) --
// or 10 (2 columns) when that saves a whole pass.
static PassPlan make_plan(uint32_t log_n) {
    PassPlan p;
    uint32_t low = log_n < (uint32_t)kMaxTileLog ? log_n : (uint32_t)kMaxTileLog;
    uint32_t rest = log_n - low;
    if (rest > 0 && rest < 3) {  // every pass needs at least 3 stages
        low -= 3 - rest;
        rest = 3;
    }
    p.npass = 1;
    p.s0[0] = 0;
    p.s1[0] = low;
    if (rest > 0) {
        uint32_t np = rest <= 11 ? 1 : (rest + 8) / 9;
        uint32_t base = rest / np, extra = rest % np, s = low;
        for (uint32_t i = 0; i < np; i++) {
            uint32_t k = base + (i < extra ? 1 : 0);
            p.s0[p.npass] = s;
            p.s1[p.npass] = s + k;
            s += k;
            p.npass++;
        }
    }
    return p;
}

// DIF (natural -> bit-reversed) or DIT (bit-reversed -> natural) butterflies over `batch`
// vectors of 2^log_n elements; `scale` (if given, radix-2^256 Montgomery form) multiplies
// every output element in the last pass.  Outputs are canonical.
// pre / post (log_n >= 3): tables of one factor per element, by address, applied when the first pass
// loads (its input is in natural order for DIF) or the last pass stores (natural order for DIT); post
// replaces `scale`.
// With `natural` set (3 <= log_n <= kMaxNaturalLog) the permutation is folded into the passes:
// DIF then takes natural order to natural order, DIT likewise; transforms of more than one pass
// go through the context's transform workspace (the scattering pass cannot work in place).
static constexpr uint32_t kMaxNaturalLog = 20;
// scale_is: what `scale` (and `scale_upper`) are known to be, which lets them be folded into a cached table:
// kScaleInverseOfN = 2^-log_n; kScaleInverseOfZ = the two values of 1 / (X^(n/2) - 1) on the coset 7 * <omega>.
enum ScaleIs { kScaleArbitrary = 0, kScaleInverseOfN = 1, kScaleInverseOfZ = 2 };
static int run_passes(plonk_hip_ctx *ctx, Fr *d, uint32_t log_n, size_t batch, bool inverse_tw, bool dif,
                      const Fr *scale, const Fr *scale_upper = nullptr, bool natural = false,
                      const Tw29 *pre = nullptr, const Tw29 *post = nullptr, ScaleIs scale_is = kScaleArbitrary) {
    const bool scale_is_inverse_of_n = scale_is == kScaleInverseOfN, scale_is_inverse_of_z = scale_is == kScaleInverseOfZ;
    const size_t n = (size_t)1 << log_n;
    if (log_n <= 2) {
        if (pre || post) return PLONK_HIP_ERR_INVALID_ARG;
        // 2 or 4 points: natural-order transform plus the permutation the caller expects;
        // omega_4 (or its inverse) is the only non-trivial factor
        Fr w4 = fr_root_of_unity(2);
        if (inverse_tw) w4 = fe_inv(w4);
        Fr one = fe_one<FrCfg>();
        if (!dif) hipLaunchKernelGGL(k_bitrev_swap, dim3(1, (unsigned)batch), dim3(256), 0, ctx->stream, d, log_n, n);
        hipLaunchKernelGGL(k_ntt_tiny, dim3(cdiv(batch, 64)), dim3(64), 0, ctx->stream, d, log_n, batch, w4,
                           scale ? *scale : one, (scale && !scale_upper) ? 1 : 0);
        if (dif) hipLaunchKernelGGL(k_bitrev_swap, dim3(1, (unsigned)batch), dim3(256), 0, ctx->stream, d, log_n, n);
        if (scale_upper) {
            hipLaunchKernelGGL(k_mul_halves, dim3(1, (unsigned)batch), dim3(256), 0, ctx->stream, d, n, n, *scale,
                               *scale_upper);
        }
        return PLONK_HIP_OK;
    }
    if (batch > 0x7fffffffull >> 3) return PLONK_HIP_ERR_INVALID_ARG;
    PassPlan pl = make_plan(log_n);
    Fr *ws = nullptr;
    if (natural) {
        if (log_n < 3 || log_n > kMaxNaturalLog || scale_upper) return PLONK_HIP_ERR_INVALID_ARG;
        if (log_n > (uint32_t)kMaxTileLog) {
            // two passes of at most 10 stages: both sides keep rows of at least 64 bytes
            const uint32_t low = log_n / 2;
            pl.npass = 2;
            pl.s0[0] = 0;
            pl.s1[0] = low;
            pl.s0[1] = low;
            pl.s1[1] = log_n;
            PLONK_TRY(buf_reserve(ctx, ctx->ntt_ws, batch * n * sizeof(Fr)));
            ws = (Fr *)ctx->ntt_ws.ptr;
        }
    }
    // the split scale needs the s0 = 0 pass to come last, as it does for DIF
    if (scale_upper && (!dif || !scale)) return PLONK_HIP_ERR_INVALID_ARG;
    if (!ctx->ntt_lds_opt_in) {
        // tiles of 2^11 elements need more than the default 64 KiB of dynamic LDS (set once per
        // context, i.e. per device)
        PLONK_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_ntt_pass<true, 3>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        PLONK_HIP_TRY(ctx, hipFuncSetAttribute((const void *)k_ntt_pass<false, 3>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        ctx->ntt_lds_opt_in = true;
    }
    // every table first: building one may grow a buffer, which waits for the stream
    const Tw29 *tws[6], *corrs[6];
    // A constant factor on every output can ride in the correction table of the strided pass that ends
    // at log_n instead of costing a product of its own: 1/n of an inverse transform (DIT: that pass
    // comes last), and the two values of 1/Z of the coset quotient (DIF: which one applies is known
    // from the row of that pass already).
    bool scale_folded = false;
    for (int i = 0; i < pl.npass; i++) {
        PLONK_TRY(get_twiddles(ctx, pl.s1[i] - pl.s0[i], inverse_tw, &tws[i]));
        corrs[i] = nullptr;
        if (pl.s0[i] == 0) continue;
        CorrKind kind = inverse_tw ? kCorrInverse : kCorrForward;
        const Fr *f_lower = nullptr, *f_upper = nullptr;
        if (pl.s1[i] == log_n && scale && !post) {
            if (!dif && inverse_tw && !scale_upper && scale_is_inverse_of_n) {
                kind = kCorrInverseScaled;
                f_lower = f_upper = scale;
            } else if (dif && !inverse_tw && scale_upper && scale_is_inverse_of_z) {
                kind = kCorrForwardOverZ;
                f_lower = scale;
                f_upper = scale_upper;
            }
            scale_folded = f_lower != nullptr;
        }
        PLONK_TRY(get_correction(ctx, pl.s0[i], pl.s1[i], kind, f_lower, f_upper, &corrs[i]));
    }
    if (scale_folded) scale = scale_upper = nullptr;
    for (int q = 0; q < pl.npass; q++) {
        int i = dif ? pl.npass - 1 - q : q;
        uint32_t k = pl.s1[i] - pl.s0[i];
        bool first = q == 0, last = q == pl.npass - 1;
        // small tiles: several per block so that a block has 1024 elements (128 work items)
        // tiles per block: up to 1024 elements per block in the contiguous pass (small tiles),
        // up to 2048 in a strided pass (adjacent columns: 128-byte rows from 4 columns on)
        const uint32_t block_log = (pl.s0[i] == 0 && !(natural && pl.npass == 2)) ? 10 : 11;
        uint32_t tpb_log = k < block_log ? block_log - k : 0;
        if (pl.s0[i] != 0 && tpb_log > pl.s0[i]) tpb_log = pl.s0[i];
        if ((n >> k) < ((size_t)1 << tpb_log)) tpb_log = (uint32_t)(log_n - k);
        const uint32_t tile_n = 1u << k;
        const size_t lds = (size_t)(9 * tile_n + (tile_n >> 5) + 1) * 4 << tpb_log;
        const unsigned threads = k + tpb_log >= 11 ? 256 : 128;
        PassArgs a;
        a.batch_stride = n;
        a.log_n = log_n;
        a.s0 = pl.s0[i];
        a.s1 = pl.s1[i];
        a.tpb_log = tpb_log;
        a.batch = (uint32_t)batch;
        a.tile_blocks = (uint32_t)((n >> k) >> tpb_log);
        if ((size_t)a.tile_blocks * batch > 0x7fffffffull) return PLONK_HIP_ERR_INVALID_ARG;
        a.do_scale = (last && scale && !post) ? (scale_upper ? 2 : 1) : 0;
        a.canonical_out = last ? 1 : 0;
        // a batch shares the per-element tables through one XCD's L2 when its blocks are dealt by tile
        const Tw29 *pass_pre = first ? pre : nullptr, *pass_post = last ? post : nullptr;
        a.xcd_map = 0;
        a.xcd_group_log = 0;
        a.scale = (scale && !post) ? fr29_twiddle_from(*scale) : Fr29{};
        a.scale_upper = scale_upper ? fr29_twiddle_from(*scale_upper) : Fr29{};
        // natural order: DIF = strided pass (rows stored reversed) then the s0 = 0 pass (scattered
        // store); DIT is the mirror image with the maps on the loads
        a.in_map = 0;
        a.out_map = 0;
        const Fr *src = d;
        Fr *dst = d;
        if (natural) {
            const int map = pl.s0[i] == 0 ? 2 : 1;
            if (dif) a.out_map = map; else a.in_map = map;
            if (pl.npass == 2) {
                if (q == 0) dst = ws; else src = ws;
            }
        }
        // rows of a strided or scattered side are 32 bytes per column of the block: neighbours that
        // complete a 128-byte line go to the same XCD
        const bool rows = pl.s0[i] != 0 || a.in_map == 2 || a.out_map == 2;
        uint32_t group_log = 0;
        while (rows && (32u << (tpb_log + group_log)) < 128u) group_log++;
        // DIF multiplies its factors in when the strided pass stores; DIT hands the factors of pass q + 1 to
        // the store of pass q
        const Tw29 *corr_store = dif ? corrs[i] : nullptr, *corr_next = nullptr;
        a.next_s0 = a.next_s1 = 0;
        a.next_map = 0;
        if (!dif && !last) {
            corr_next = corrs[i + 1];
            a.next_s0 = pl.s0[i + 1];
            a.next_s1 = pl.s1[i + 1];
            a.next_map = natural ? 1 : 0;
        }
        const bool tables = batch > 1 && (corr_store || corr_next || pass_pre || pass_post);
        if ((group_log || tables) && a.tile_blocks % (8u << group_log) == 0) {
            a.xcd_map = 1;
            a.xcd_group_log = group_log;
        }
        const dim3 grid(a.tile_blocks * (unsigned)batch);
        // (a two-stage variant, GMAX = 2, runs four waves per SIMD at the same speed)
        if (dif) {
            hipLaunchKernelGGL((k_ntt_pass<true, 3>), grid, dim3(threads), lds, ctx->stream, src, dst, tws[i], corr_store,
                               corr_next, pass_pre, pass_post, a);
        } else {
            hipLaunchKernelGGL((k_ntt_pass<false, 3>), grid, dim3(threads), lds, ctx->stream, src, dst, tws[i], corr_store,
                               corr_next, pass_pre, pass_post, a);
        }
    }
    return PLONK_HIP_OK;
}

// The pass structure of a transform of 2^log_n points, for inspection: stage ranges [s0, s1) in
// ascending order of s0.  natural: the two-pass form with the permutation folded in.
int ntt_plan_describe(uint32_t log_n, bool natural, int max_passes, uint32_t *s0, uint32_t *s1) {
    if (log_n < 3 || log_n > 30) return 0;
    PassPlan pl = make_plan(log_n);
    if (natural && log_n <= kMaxNaturalLog && log_n > (uint32_t)kMaxTileLog) {
        pl.npass = 2;
        pl.s0[0] = 0;
        pl.s1[0] = log_n / 2;
        pl.s0[1] = log_n / 2;
        pl.s1[1] = log_n;
    }
    for (int i = 0; i < pl.npass && i < max_passes; i++) {
        s0[i] = pl.s0[i];
        s1[i] = pl.s1[i];
    }
    return pl.npass;
}

int ntt_butterflies(plonk_hip_ctx *ctx, Fr *d_data, uint32_t log_n, size_t batch, bool inverse) {
    if (log_n > 30) return PLONK_HIP_ERR_INVALID_ARG;
    if (batch == 0 || log_n == 0) return PLONK_HIP_OK;
    const size_t n = (size_t)1 << log_n;
    const Fr ninv = fe_inv(fr_from_u64(n));
    // the batch index is a grid dimension: long batches go in slabs
    for (size_t b0 = 0; b0 < batch; b0 += 32768) {
        size_t nb = batch - b0 < 32768 ? batch - b0 : 32768;
        if (inverse) {
            PLONK_TRY(run_passes(ctx, d_data + b0 * n, log_n, nb, true, false, &ninv, nullptr, false, nullptr, nullptr,
                                 kScaleInverseOfN));
        } else {
            PLONK_TRY(run_passes(ctx, d_data + b0 * n, log_n, nb, false, true, nullptr));
        }
    }
    return PLONK_HIP_OK;
}

int ntt_dev(plonk_hip_ctx *ctx, Fr *d_data, uint32_t log_n, size_t batch, bool inverse, const Fr *coset_shift) {
    if (log_n > 30 || batch > 65535) return PLONK_HIP_ERR_INVALID_ARG;
    if (batch == 0 || log_n == 0) return PLONK_HIP_OK;  // one point: f(shift * 1) = f_0 either way
    const size_t n = (size_t)1 << log_n;
    Fr ninv = fe_inv(fr_from_u64(n));
    if (coset_shift && inverse && fe_is_zero(*coset_shift)) return PLONK_HIP_ERR_ZERO_DENOMINATOR;
    // the coset factors ride on the first load (forward) or the last store (inverse, with 1/n);
    // transforms of 2 or 4 points scale in a launch of their own
    const Tw29 *pre = nullptr, *post = nullptr;
    const bool tiny = log_n < 3;
    if (coset_shift && !tiny) {
        if (fe_is_zero(*coset_shift)) {
            // f(0 * X) = f_0 everywhere: keep the constant coefficient only (no table of powers of 0)
            PLONK_HIP_TRY(ctx, hipMemset2DAsync(d_data + 1, n * sizeof(Fr), 0, (n - 1) * sizeof(Fr), batch, ctx->stream));
        } else if (inverse) {
            PLONK_TRY(get_coset_table(ctx, log_n, *coset_shift, true, &post));
        } else {
            PLONK_TRY(get_coset_table(ctx, log_n, *coset_shift, false, &pre));
        }
    }
    if (!inverse && coset_shift && tiny) PLONK_TRY(scale_by_powers(ctx, d_data, n, batch, *coset_shift));
    if (log_n >= 3 && log_n <= kMaxNaturalLog) {
        if (!inverse) {
            // natural order to natural order both ways: the DIT butterflies are the cheaper ones (no
            // reduction steps between the stages), so the forward transform uses them as well
            PLONK_TRY(run_passes(ctx, d_data, log_n, batch, false, false, nullptr, nullptr, true, pre, nullptr));
        } else {
            PLONK_TRY(run_passes(ctx, d_data, log_n, batch, true, false, &ninv, nullptr, true, nullptr, post,
                                 kScaleInverseOfN));
        }
    } else if (!inverse) {
        PLONK_TRY(run_passes(ctx, d_data, log_n, batch, false, true, nullptr, nullptr, false, pre, nullptr));
        hipLaunchKernelGGL(k_bitrev_swap, dim3(cdiv(n, 256), (unsigned)batch), dim3(256), 0, ctx->stream, d_data, log_n,
                           n);
    } else {
        hipLaunchKernelGGL(k_bitrev_swap, dim3(cdiv(n, 256), (unsigned)batch), dim3(256), 0, ctx->stream, d_data, log_n,
                           n);
        PLONK_TRY(run_passes(ctx, d_data, log_n, batch, true, false, &ninv, nullptr, false, nullptr, post,
                             kScaleInverseOfN));
    }
    if (inverse && coset_shift && tiny) PLONK_TRY(scale_by_powers(ctx, d_data, n, batch, fe_inv(*coset_shift)));
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// domains of size 3 * 2^k (the reference lays a circuit of G gates out on 3G + ... points:
// setup_global_params/mod.rs:57; its example uses d = 12): one radix-3 step around the
// power-of-two transform
// ------------------------------------------------------------------------------------------
// omega_n = 7^((r-1)/n) for any n dividing r - 1 (common/utils.rs:11-38); false when n does not
static bool fr_root_of_unity_general(uint64_t n, Fr *out) {
    if (n == 0 || n > 0xffffffffull) return false;
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = FrCfg::mod(i);
    e[0] -= 1;
    uint64_t rem = 0;
    for (int i = 7; i >= 0; i--) {  // (r - 1) / n by long division on 32-bit limbs
        uint64_t cur = (rem << 32) | e[i];
        e[i] = (uint32_t)(cur / n);
        rem = cur % n;
    }
    if (rem != 0) return false;
    Fr base = fr_from_u64(7), acc = fe_one<FrCfg>();
    for (int i = 7; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            acc = fe_sqr(acc);
            if ((e[i] >> b) & 1) acc = fe_mul(acc, base);
        }
    *out = acc;
    return true;
}

// y[c][i] = x[3 i + c]
__global__ __launch_bounds__(256) void k_deinterleave3(const Fr *__restrict__ x, size_t m, Fr *__restrict__ y) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * m) return;
    const Fr *src = x + (size_t)blockIdx.y * 3 * m;
    Fr *dst = y + (size_t)blockIdx.y * 3 * m;
    dst[(i % 3) * m + i / 3] = src[i];
}
// X[j] = (Y0[j mod m] + w^j Y1[j mod m] + w^2j Y2[j mod m]) * scale, w^j = lo[j & 1023] * hi[j >> 10]
__global__ __launch_bounds__(256) void k_combine3(const Fr *__restrict__ y, size_t m, const Fr *__restrict__ lo,
                                                 const Fr *__restrict__ hi, Fr scale, int do_scale,
                                                 Fr *__restrict__ x) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 3 * m) return;
    const Fr *src = y + (size_t)blockIdx.y * 3 * m;
    Fr w = lo[j & 1023];
    if (j >> 10) w = fe_mul(w, hi[j >> 10]);
    const size_t i = j % m;
    Fr acc = fe_add(src[i], fe_mul(w, fe_add(src[m + i], fe_mul(w, src[2 * m + i]))));
    if (do_scale) acc = fe_mul(acc, scale);
    x[(size_t)blockIdx.y * 3 * m + j] = acc;
}

// Natural order in and out over Omega = construct_Omega(n), n = 2^k or 3 * 2^k.
int ntt_mixed_dev(plonk_hip_ctx *ctx, Fr *d_data, size_t n, size_t batch, bool inverse) {
    if (n == 0 || batch == 0) return PLONK_HIP_OK;
    size_t m = n;
    uint32_t log_m = 0;
    while ((m & 1) == 0) {
        m >>= 1;
        log_m++;
    }
    if (m == 1) return ntt_dev(ctx, d_data, log_m, batch, inverse, nullptr);
    if (m != 3 || log_m > 29 || batch * 3 > 65535) return PLONK_HIP_ERR_INVALID_ARG;
    m = (size_t)1 << log_m;
    Fr w;
    if (!fr_root_of_unity_general(n, &w)) return PLONK_HIP_ERR_INVALID_ARG;
    if (inverse) w = fe_inv(w);
    PLONK_TRY(buf_reserve(ctx, ctx->ntt3_ws, batch * n * sizeof(Fr)));
    Fr *y = (Fr *)ctx->ntt3_ws.ptr;
    hipLaunchKernelGGL(k_deinterleave3, dim3(cdiv(n, 256), (unsigned)batch), dim3(256), 0, ctx->stream, d_data, m, y);
    // the three sub-transforms use omega_n^3 = omega_m (the inverse ones also divide by m)
    PLONK_TRY(ntt_dev(ctx, y, log_m, batch * 3, inverse, nullptr));
    const Fr *lo, *hi;
    PLONK_TRY(build_two_level(ctx, w, n, &lo, &hi));
    const Fr third = fe_inv(fr_from_u64(3));
    hipLaunchKernelGGL(k_combine3, dim3(cdiv(n, 256), (unsigned)batch), dim3(256), 0, ctx->stream, y, m, lo, hi, third,
                       inverse ? 1 : 0, d_data);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// coset quotient by X^k - 1
// ------------------------------------------------------------------------------------------
__global__ void k_set_int(int *__restrict__ flags, size_t count, int value) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) flags[i] = value;
}

// flags[b] = 0 if any of data[b][first .. first+count) is non-zero
__global__ __launch_bounds__(256) void k_check_zero(const Fr *__restrict__ data, size_t first, size_t count,
                                                   size_t batch_stride, int *__restrict__ flags) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if (!fe_is_zero(data[(size_t)blockIdx.y * batch_stride + first + i])) flags[blockIdx.y] = 0;
}

int quotient_coset_dev(plonk_hip_ctx *ctx, Fr *d_f, uint32_t log_k, size_t batch, int *d_divisible) {
    if (log_k > 29 || batch > 65535) return PLONK_HIP_ERR_INVALID_ARG;
    if (batch == 0) return PLONK_HIP_OK;
    const uint32_t log_n = log_k + 1;
    const size_t k = (size_t)1 << log_k, n = 2 * k;
    const Fr g = fr_from_u64(7);  // multiplicative generator: g^k is never +-1 for k | 2^31
    // g^k
    Fr gk = g;
    for (uint32_t i = 0; i < log_k; i++) gk = fe_sqr(gk);
    Fr one = fe_one<FrCfg>();
    Fr z_even = fe_sub(gk, one);             // Z on points with even exponent
    Fr z_odd = fe_sub(fe_neg(gk), one);      // Z on points with odd exponent
    if (fe_is_zero(z_even) || fe_is_zero(z_odd)) return PLONK_HIP_ERR_ZERO_DENOMINATOR;
    Fr inv_even = fe_inv(z_even), inv_odd = fe_inv(z_odd);

    Fr ninv = fe_inv(fr_from_u64(n));
    if (log_n < 3) {
        PLONK_TRY(scale_by_powers(ctx, d_f, n, batch, g));
        PLONK_TRY(run_passes(ctx, d_f, log_n, batch, false, true, &inv_even, &inv_odd));
        PLONK_TRY(run_passes(ctx, d_f, log_n, batch, true, false, &ninv));
        PLONK_TRY(scale_by_powers(ctx, d_f, k, batch, fe_inv(g), n));
    } else {
        // g^i rides on the first load, 1/Z on the store of the last forward pass, g^-i / n on the last
        // store of the way back (the upper half is zero exactly when it was zero before: every factor is
        // a non-zero power, and that is all the divisibility test below reads)
        const Tw29 *pre, *post;
        PLONK_TRY(get_coset_table(ctx, log_n, g, false, &pre));
        PLONK_TRY(get_coset_table(ctx, log_n, g, true, &post));
        PLONK_TRY(run_passes(ctx, d_f, log_n, batch, false, true, &inv_even, &inv_odd, false, pre, nullptr,
                             kScaleInverseOfZ));
        PLONK_TRY(run_passes(ctx, d_f, log_n, batch, true, false, &ninv, nullptr, false, nullptr, post));
    }
    if (d_divisible) {
        hipLaunchKernelGGL(k_set_int, dim3(cdiv(batch, 256)), dim3(256), 0, ctx->stream, d_divisible, batch, 1);
        hipLaunchKernelGGL(k_check_zero, dim3(cdiv(k, 256), (unsigned)batch), dim3(256), 0, ctx->stream, d_f, k, k, n,
                           d_divisible);
    }
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

}  // namespace plonk
