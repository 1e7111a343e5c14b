// Fr on 9 unsaturated 29-bit limbs: the arithmetic of the NTT butterflies.
//
// Same idea as field28.cuh: a 29-bit limb product is < 2^58, so a whole column of the schoolbook
// product (at most 18 products) accumulates in one 64-bit register with single in-place
// v_mad_u64_u32 instructions -- no carry flags, no operand re-pairing.  162 multiply-adds per
// Montgomery product (radix R29 = 2^261).
//
// Data elements stay in the caller's Montgomery domain (x * 2^256, as ark-ff stores them) and
// are only re-sliced into 29-bit limbs; the twiddle factors are kept as w * 2^261, so that
//     mont29(x * 2^256, w * 2^261) = x * w * 2^256
// needs no domain conversion.  Arithmetic is lazy: limbs are kept normalised (< 2^29) while values
// may exceed r; 261 bits of capacity allow values up to 64 * 2^255 (about 70 r).  Bounds are
// tracked in units of 2^255 where the functions are used (ntt.hip).
#pragma once
#include "field.cuh"

namespace plonk {

struct R29 {
    static constexpr int N = 9;
    static constexpr int BITS = 29;
    static constexpr uint32_t MASK = (1u << BITS) - 1;
    // r
    static PLONK_HD constexpr uint32_t mod(int i) {
        constexpr uint32_t m[N] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu,
                                   0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
        return m[i];
    }
    // -r^-1 mod 2^29
    static constexpr uint32_t INV = 0x1fffffffu;
};

struct Fr29 {
    uint32_t l[R29::N];  // 36 bytes; odd word stride keeps LDS accesses spread over the banks
};

// limb i of K*r (compile time)
template <int K>
PLONK_HD constexpr uint32_t r29_kr(int i) {
    uint64_t c = 0;
    uint32_t out = 0;
    for (int j = 0; j <= i; j++) {
        c += (uint64_t)K * R29::mod(j);
        out = (uint32_t)(c & R29::MASK);
        c >>= R29::BITS;
    }
    return out;
}

// re-slice a 256-bit integer (8 x 32-bit limbs) into 9 x 29-bit limbs and back
PLONK_HD Fr29 fr29_pack(const Fr &a) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < R29::N; i++) {
        const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
        uint64_t lo = a.l[w];
        uint64_t hi = (w + 1 < FrCfg::N) ? a.l[w + 1] : 0u;
        r.l[i] = (uint32_t)((lo | (hi << 32)) >> sh) & R29::MASK;
    }
    return r;
}
// value must be < 2^256
PLONK_HD Fr fr29_unpack(const Fr29 &a) {
    Fr r;
#pragma unroll
    for (int w = 0; w < FrCfg::N; w++) {
        uint64_t v = 0;
        const int first = (32 * w) / 29;
#pragma unroll
        for (int i = first; i < first + 3 && i < R29::N; i++) {
            int shift = 29 * i - 32 * w;
            if (shift >= 32) continue;
            if (shift >= 0) {
                v |= (uint64_t)a.l[i] << shift;
            } else {
                v |= (uint64_t)a.l[i] >> (-shift);
            }
        }
        r.l[w] = (uint32_t)v;
    }
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__) && defined(PLONK_MONT29_ASM)
// The product below as one hand-scheduled instruction sequence (mont29_asm.inc, written and checked by
// scripts/gen_mont_asm.py): 189 instructions against about 230 from the compiler, which splits the column
// sums into independent chains and joins them with 64-bit additions.  The digits m_k are subtracted
// (signed multiply-adds with -r_j), so that m_k is just the low 29 bits of the running column and no
// product with r_0 = 1 is formed; the last digit is taken 2^29 lower, which adds r: the value is exactly
// the one the portable form returns whenever that form's digits are not all zero.
// Needs a's limbs below 2^30 (signed column sums), b's below 2^29.  Accumulator: v[2:3].
__device__ __forceinline__ Fr29 fr29_mul_asm(const Fr29 &a, const Fr29 &b) {
    Fr29 r;
    asm(
#include "mont29_asm.inc"
        : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]),
          "=&v"(r.l[7]), "=&v"(r.l[8])
        : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
          "v"(a.l[8]), "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]),
          "v"(b.l[7]), "v"(b.l[8]), "s"(0u - R29::mod(1)), "s"(0u - R29::mod(2)), "s"(0u - R29::mod(3)),
          "s"(0u - R29::mod(4)), "s"(0u - R29::mod(5)), "s"(0u - R29::mod(6)), "s"(0u - R29::mod(7)),
          "s"(0u - R29::mod(8)), "s"(R29::MASK), "s"(~R29::MASK)
        : "vcc", "v2", "v3");
    return r;
}
#endif

// Montgomery product a*b/2^261 mod r, no final subtraction:
// a < A * 2^255, b < r  =>  result < r * (1 + A / 64) ... precisely a*b/2^261 + r.
PLONK_HD Fr29 fr29_mul_lazy(const Fr29 &a, const Fr29 &b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(PLONK_MONT29_ASM)
    return fr29_mul_asm(a, b);
#else
    constexpr int N = R29::N;
    uint64_t acc = 0;
    uint32_t m[N];
    Fr29 r;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * R29::mod(k - i);
        m[k] = ((uint32_t)acc * R29::INV) & R29::MASK;
        acc += (uint64_t)m[k] * R29::mod(0);
        acc >>= R29::BITS;
    }
#pragma unroll
    for (int k = N; k < 2 * N; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)m[i] * R29::mod(k - i);
        }
        r.l[k - N] = (uint32_t)acc & R29::MASK;
        acc >>= R29::BITS;
    }
    return r;
#endif
}

// a + b, carries propagated, no reduction.
PLONK_HD Fr29 fr29_add_lazy(const Fr29 &a, const Fr29 &b) {
    Fr29 s;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < R29::N; i++) {
        uint32_t t = a.l[i] + b.l[i] + c;
        s.l[i] = t & R29::MASK;
        c = t >> R29::BITS;
    }
    return s;
}

// a - b + K*r in one carry pass; needs K*r >= b.
template <int K>
PLONK_HD Fr29 fr29_sub_lazy(const Fr29 &a, const Fr29 &b) {
    Fr29 r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < R29::N; i++) {
        int32_t t = (int32_t)a.l[i] - (int32_t)b.l[i] + (int32_t)r29_kr<K>(i) + c;
        r.l[i] = (uint32_t)t & R29::MASK;
        c = t >> R29::BITS;
    }
    return r;
}

// v - floor(v / 2^255) * r: any v < 2^261 comes down below 2^255 * (1 + 0.1 * floor(v / 2^255)).
// Applying it twice brings every representable value below 1.2 * 2^255 < 2^256.
PLONK_HD Fr29 fr29_reduce_step(const Fr29 &v) {
    const int64_t k = v.l[8] >> 23;  // bits 255.. of the value (limb 8 starts at bit 232)
    Fr29 r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < R29::N; i++) {
        int64_t t = (int64_t)v.l[i] - k * (int64_t)R29::mod(i) + c;
        r.l[i] = (uint32_t)t & R29::MASK;
        c = t >> R29::BITS;
    }
    return r;
}
PLONK_HD Fr29 fr29_reduce_below_2_256(const Fr29 &v) { return fr29_reduce_step(fr29_reduce_step(v)); }

// v < 2r  ->  v mod r
PLONK_HD Fr29 fr29_cond_sub(const Fr29 &a) {
    uint32_t t[R29::N];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < R29::N; i++) {
        int32_t d = (int32_t)a.l[i] - (int32_t)R29::mod(i) + c;
        t[i] = (uint32_t)d & R29::MASK;
        c = d >> R29::BITS;
    }
    Fr29 r;
#pragma unroll
    for (int i = 0; i < R29::N; i++) r.l[i] = c ? a.l[i] : t[i];
    return r;
}
// any representable value -> canonical (< r): below 1.2 * 2^255 < 2r after the two steps
PLONK_HD Fr29 fr29_canon(const Fr29 &v) { return fr29_cond_sub(fr29_reduce_below_2_256(v)); }

// Twiddle form: w (canonical, Montgomery radix 2^256, i.e. the integer w*2^256 mod r) ->
// the integer w * 2^261 mod r in 29-bit limbs.
PLONK_HD Fr29 fr29_twiddle_from(const Fr &w_mont256) {
    // 32 in the radix-2^256 Montgomery domain: fe_mul(w*2^256, 32*2^256) = w*32*2^256 = w*2^261
    Fr c32;
    constexpr uint32_t k[8] = {0xffffffbau, 0x00000045u, 0x0072d846u, 0x1a25272eu,
                               0x5dbeee8bu, 0xfe2eedcdu, 0x9eefbe41u, 0x4d043f42u};
#pragma unroll
    for (int i = 0; i < 8; i++) c32.l[i] = k[i];
    return fr29_pack(fe_mul(w_mont256, c32));
}

}  // namespace plonk
