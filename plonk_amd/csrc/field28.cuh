// Fq on 14 unsaturated 28-bit limbs: the arithmetic of the MSM's bucket-accumulation loop.
//
// Why a second representation: on gfx950 every carry-flag link (v_add_co -> v_addc_co, or the
// carry-out of v_mad_u64_u32) costs wait states, and a saturated 32-bit limb product leaves no
// room to add a second product into the same 64-bit accumulator.  With 28-bit limbs a product
// is < 2^56, so a whole column of the schoolbook product (at most 28 products) fits one 64-bit
// accumulator: every multiply-add is ONE v_mad_u64_u32 accumulating in place -- no carry
// flags, no operand re-pairing, no hazards -- and additions/subtractions propagate carries with
// shifts and masks instead of flags.
//
// Value = sum l[i] * 2^(28 i).  Montgomery radix R28 = 2^392.  Every element handed between
// functions is canonical: each limb < 2^28 and the value < q.
#pragma once
#include "field.cuh"

namespace plonk {

struct Q28 {
    static constexpr int N = 14;
    static constexpr int BITS = 28;
    static constexpr uint32_t MASK = (1u << BITS) - 1;
    // q
    static PLONK_HD constexpr uint32_t mod(int i) {
        constexpr uint32_t m[N] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u,
                                   0xf38512bu, 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
        return m[i];
    }
    // -q^-1 mod 2^28
    static constexpr uint32_t INV = 0xffcfffdu;
    // R28 mod q
    static PLONK_HD constexpr uint32_t one(int i) {
        constexpr uint32_t m[N] = {0x347fcb8u, 0xd800000u, 0x002b119u, 0x0cde6d2u, 0xc7212e0u, 0x83a2090u, 0x037669fu,
                                   0xda0f73eu, 0x9b09b42u, 0x1297bb0u, 0x515d98fu, 0x012ca7cu, 0x659fcfau, 0x000577au};
        return m[i];
    }
    // 2^400 mod q: Montgomery-multiplying x*2^384 by it gives x*2^392 (radix 2^384 -> 2^392)
    static PLONK_HD constexpr uint32_t to28(int i) {
        constexpr uint32_t m[N] = {0x80e6299u, 0x3500034u, 0xeb12856u, 0xdeb2699u, 0xc988670u, 0x4ef6697u, 0x70983e8u,
                                   0xa4e6fe9u, 0x3e8a053u, 0xecf271eu, 0xc20d323u, 0x6eb6385u, 0x47f1286u, 0x00156dau};
        return m[i];
    }
    // 2^384 mod q: Montgomery-multiplying x*2^392 by it gives x*2^384 (radix 2^392 -> 2^384)
    static PLONK_HD constexpr uint32_t from28(int i) {
        constexpr uint32_t m[N] = {0x002fffdu, 0x0900000u, 0xc000276u, 0x000bc40u, 0x8baebf4u, 0x5753c75u, 0x55f4898u,
                                   0x7052574u, 0x7ce5853u, 0x56ec6d7u, 0x71a97a2u, 0xe4935c0u, 0xec3fa80u, 0x0015f65u};
        return m[i];
    }
};

// 14 limbs = 56 bytes, padded to 64 so that an affine point is exactly one 128-byte line.
struct alignas(16) Fq28 {
    uint32_t l[Q28::N];
};

PLONK_HD Fq28 q28_zero() {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) r.l[i] = 0;
    return r;
}
PLONK_HD Fq28 q28_one() {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) r.l[i] = Q28::one(i);
    return r;
}
PLONK_HD bool q28_is_zero(const Fq28 &a) {
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) x |= a.l[i];
    return x == 0;
}

// a in [0, 2q) with normalised limbs -> a mod q.
PLONK_HD Fq28 q28_reduce_once(const Fq28 &a) {
    uint32_t t[Q28::N];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        int32_t d = (int32_t)a.l[i] - (int32_t)Q28::mod(i) + c;
        t[i] = (uint32_t)d & Q28::MASK;
        c = d >> Q28::BITS;  // 0 or -1
    }
    Fq28 r;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) r.l[i] = c ? a.l[i] : t[i];  // c == -1 <=> a < q
    return r;
}

PLONK_HD Fq28 q28_add(const Fq28 &a, const Fq28 &b) {
    Fq28 s;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        uint32_t t = a.l[i] + b.l[i] + c;
        s.l[i] = t & Q28::MASK;
        c = t >> Q28::BITS;
    }
    return q28_reduce_once(s);  // a + b < 2q < 2^392: no carry out of the top limb
}

PLONK_HD Fq28 q28_sub(const Fq28 &a, const Fq28 &b) {
    uint32_t t[Q28::N];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        int32_t d = (int32_t)a.l[i] - (int32_t)b.l[i] + c;
        t[i] = (uint32_t)d & Q28::MASK;
        c = d >> Q28::BITS;
    }
    // c == -1: the difference is negative, add q back
    uint32_t mask = (uint32_t)c;
    Fq28 r;
    uint32_t cc = 0;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        uint32_t u = t[i] + (Q28::mod(i) & mask) + cc;
        r.l[i] = u & Q28::MASK;
        cc = u >> Q28::BITS;
    }
    return r;
}

PLONK_HD Fq28 q28_neg(const Fq28 &a) { return q28_sub(q28_zero(), a); }

PLONK_HD Fq28 q28_dbl(const Fq28 &a) { return q28_add(a, a); }

// Montgomery product a*b/2^392 mod q by product scanning: column k sums a[i]*b[k-i] and
// m[i]*q[k-i] in one 64-bit accumulator, m[k] clears the column's low 28 bits, the accumulator
// shifts down one limb.  392 multiply-adds, each a single in-place v_mad_u64_u32.
PLONK_HD Fq28 q28_mul(const Fq28 &a, const Fq28 &b) {
    constexpr int N = Q28::N;
    uint64_t acc = 0;
    uint32_t m[N];
    Fq28 r;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Q28::mod(k - i);
        m[k] = ((uint32_t)acc * Q28::INV) & Q28::MASK;
        acc += (uint64_t)m[k] * Q28::mod(0);
        acc >>= Q28::BITS;
    }
#pragma unroll
    for (int k = N; k < 2 * N; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)m[i] * Q28::mod(k - i);
        }
        r.l[k - N] = (uint32_t)acc & Q28::MASK;
        acc >>= Q28::BITS;
    }
    // a, b < q  =>  result < q + q*q/2^392 < 2q
    return q28_reduce_once(r);
}

// a^2: the off-diagonal products are taken once against the doubled operand.
PLONK_HD Fq28 q28_sqr(const Fq28 &a) {
    constexpr int N = Q28::N;
    uint32_t a2[N];
#pragma unroll
    for (int i = 0; i < N; i++) a2[i] = a.l[i] << 1;  // < 2^29: products < 2^57, columns still fit
    uint64_t acc = 0;
    uint32_t m[N];
    Fq28 r;
#pragma unroll
    for (int k = 0; k < 2 * N; k++) {
        // sum_{i+j=k, i<j} 2 a_i a_j  (+ a_{k/2}^2 when k is even)
#pragma unroll
        for (int i = 0; i < N; i++) {
            int j = k - i;
            if (j > i && j < N) acc += (uint64_t)a2[i] * a.l[j];
        }
        if ((k & 1) == 0 && k / 2 < N) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
        if (k < N) {
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Q28::mod(k - i);
            m[k] = ((uint32_t)acc * Q28::INV) & Q28::MASK;
            acc += (uint64_t)m[k] * Q28::mod(0);
        } else {
#pragma unroll
            for (int i = k - N + 1; i < N; i++) acc += (uint64_t)m[i] * Q28::mod(k - i);
            r.l[k - N] = (uint32_t)acc & Q28::MASK;
        }
        acc >>= Q28::BITS;
    }
    return q28_reduce_once(r);
}

// ---------------------------------------------------------------------------------------------
// Lazy forms for the bucket-accumulation loop: limbs stay normalised (< 2^28) but values are only
// bounded by a small multiple of q, so the conditional subtractions disappear.  Bounds are in
// units of q and are tracked where the functions are used (g1_28.cuh).
// ---------------------------------------------------------------------------------------------
// limb i of K*q
template <int K>
PLONK_HD constexpr uint32_t q28_kq(int i) {
    uint64_t c = 0;
    uint32_t out = 0;
    for (int j = 0; j <= i; j++) {
        c += (uint64_t)K * Q28::mod(j);
        out = (uint32_t)(c & Q28::MASK);
        c >>= Q28::BITS;
    }
    return out;
}

#if defined(__HIP_DEVICE_COMPILE__) && defined(PLONK_MONT28_ASM)
// The lazy products below as hand-scheduled instruction sequences (mont28_*_asm.inc, written and checked by
// scripts/gen_mont_asm.py): one dependent chain per product, the digits m_k subtracted with signed
// multiply-adds (m_k = low limb of the running column times q^-1), the last digit taken 2^28 lower so that
// the result comes out as a*b/2^392 + q at most, like the portable forms'.  The compiler's code for the
// portable forms splits the column sums into independent chains and joins them with 64-bit additions.
// Needs normalised limbs (below 2^28; the doubled operand of the squaring below 2^29).  Accumulator: v[2:3].
#define PLONK_I14(x)                                                                                                  \
    "v"(x.l[0]), "v"(x.l[1]), "v"(x.l[2]), "v"(x.l[3]), "v"(x.l[4]), "v"(x.l[5]), "v"(x.l[6]), "v"(x.l[7]), "v"(x.l[8]), \
        "v"(x.l[9]), "v"(x.l[10]), "v"(x.l[11]), "v"(x.l[12]), "v"(x.l[13])
#define PLONK_O14(x)                                                                                                  \
    "=&v"(x.l[0]), "=&v"(x.l[1]), "=&v"(x.l[2]), "=&v"(x.l[3]), "=&v"(x.l[4]), "=&v"(x.l[5]), "=&v"(x.l[6]),            \
        "=&v"(x.l[7]), "=&v"(x.l[8]), "=&v"(x.l[9]), "=&v"(x.l[10]), "=&v"(x.l[11]), "=&v"(x.l[12]), "=&v"(x.l[13])
#define PLONK_Q14_CONSTANTS                                                                                            \
    "s"(0u - Q28::mod(0)), "s"(0u - Q28::mod(1)), "s"(0u - Q28::mod(2)), "s"(0u - Q28::mod(3)), "s"(0u - Q28::mod(4)),   \
        "s"(0u - Q28::mod(5)), "s"(0u - Q28::mod(6)), "s"(0u - Q28::mod(7)), "s"(0u - Q28::mod(8)),                      \
        "s"(0u - Q28::mod(9)), "s"(0u - Q28::mod(10)), "s"(0u - Q28::mod(11)), "s"(0u - Q28::mod(12)),                   \
        "s"(0u - Q28::mod(13)), "s"(Q28::MASK), "s"(~Q28::MASK), "s"((0u - Q28::INV) & Q28::MASK)
__device__ __forceinline__ Fq28 q28_mul_asm(const Fq28 &a, const Fq28 &b) {
    Fq28 r;
    asm(
#include "mont28_mul_asm.inc"
        : PLONK_O14(r)
        : PLONK_I14(a), PLONK_I14(b), PLONK_Q14_CONSTANTS
        : "vcc", "v2", "v3");
    return r;
}
__device__ __forceinline__ Fq28 q28_sqr_asm(const Fq28 &a) {
    Fq28 a2, r;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) a2.l[i] = a.l[i] << 1;
    asm(
#include "mont28_sqr_asm.inc"
        : PLONK_O14(r)
        : PLONK_I14(a), PLONK_I14(a2), PLONK_Q14_CONSTANTS
        : "vcc", "v2", "v3");
    return r;
}
__device__ __forceinline__ Fq28 q28_mul2_asm(const Fq28 &a, const Fq28 &b, const Fq28 &c, const Fq28 &d) {
    Fq28 r;
    asm(
#include "mont28_mul2_asm.inc"
        : PLONK_O14(r)
        : PLONK_I14(a), PLONK_I14(b), PLONK_I14(c), PLONK_I14(d), PLONK_Q14_CONSTANTS
        : "vcc", "v2", "v3");
    return r;
}
#undef PLONK_I14
#undef PLONK_O14
#undef PLONK_Q14_CONSTANTS
#endif

// Montgomery product without the final conditional subtraction:
// a < A*q, b < B*q  =>  result < (1 + A*B/2500) * q   (q / 2^392 < 1/2500).
PLONK_HD Fq28 q28_mul_lazy(const Fq28 &a, const Fq28 &b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(PLONK_MONT28_ASM)
    return q28_mul_asm(a, b);
#else
    constexpr int N = Q28::N;
    uint64_t acc = 0;
    uint32_t m[N];
    Fq28 r;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Q28::mod(k - i);
        m[k] = ((uint32_t)acc * Q28::INV) & Q28::MASK;
        acc += (uint64_t)m[k] * Q28::mod(0);
        acc >>= Q28::BITS;
    }
#pragma unroll
    for (int k = N; k < 2 * N; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)m[i] * Q28::mod(k - i);
        }
        r.l[k - N] = (uint32_t)acc & Q28::MASK;
        acc >>= Q28::BITS;
    }
    return r;
#endif
}

// a * a / 2^392 mod q, lazy like q28_mul_lazy.
PLONK_HD Fq28 q28_sqr_lazy(const Fq28 &a) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(PLONK_MONT28_ASM)
    return q28_sqr_asm(a);
#else
    return q28_mul_lazy(a, a);  // (the compiler takes the off-diagonal products once)
#endif
}

// (a*b + c*d) / 2^392 mod q with one reduction for both products:
// a < A*q, b < B*q, c < C*q, d < D*q  =>  result < (1 + (A*B + C*D)/2500) * q.
// A column holds at most 28 products and 14 reduction terms, each below 2^56: under 2^62.
PLONK_HD Fq28 q28_mul2_lazy(const Fq28 &a, const Fq28 &b, const Fq28 &c, const Fq28 &d) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(PLONK_MONT28_ASM)
    return q28_mul2_asm(a, b, c, d);
#else
    constexpr int N = Q28::N;
    uint64_t acc = 0;
    uint32_t m[N];
    Fq28 r;
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)c.l[i] * d.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Q28::mod(k - i);
        m[k] = ((uint32_t)acc * Q28::INV) & Q28::MASK;
        acc += (uint64_t)m[k] * Q28::mod(0);
        acc >>= Q28::BITS;
    }
#pragma unroll
    for (int k = N; k < 2 * N; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)c.l[i] * d.l[k - i];
            acc += (uint64_t)m[i] * Q28::mod(k - i);
        }
        r.l[k - N] = (uint32_t)acc & Q28::MASK;
        acc >>= Q28::BITS;
    }
    return r;
#endif
}

// (sign ? -a : a) - b + K*q in one carry pass.  Needs K*q >= b (+ a when sign is set).
// Result < a + K*q (sign clear) or <= K*q (sign set).
template <int K>
PLONK_HD Fq28 q28_sub_lazy(const Fq28 &a, const Fq28 &b, bool sign = false) {
    const int32_t sm = sign ? -1 : 0;
    Fq28 r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        int32_t ai = ((int32_t)a.l[i] ^ sm) - sm;  // +-a_i
        int32_t t = ai - (int32_t)b.l[i] + (int32_t)q28_kq<K>(i) + c;
        r.l[i] = (uint32_t)t & Q28::MASK;
        c = t >> Q28::BITS;
    }
    return r;
}

// a - b - 2c + K*q in one carry pass.  Needs K*q >= b + 2c.
template <int K>
PLONK_HD Fq28 q28_sub2_lazy(const Fq28 &a, const Fq28 &b, const Fq28 &c2) {
    Fq28 r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        int32_t t = (int32_t)a.l[i] - (int32_t)b.l[i] - (int32_t)(c2.l[i] << 1) + (int32_t)q28_kq<K>(i) + c;
        r.l[i] = (uint32_t)t & Q28::MASK;
        c = t >> Q28::BITS;
    }
    return r;
}

// x < 2q with normalised limbs: is x = 0 mod q?
PLONK_HD bool q28_is_zero_mod_q(const Fq28 &x) {
    uint32_t z = 0, d = 0;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        z |= x.l[i];
        d |= x.l[i] ^ Q28::mod(i);
    }
    return z == 0 || d == 0;
}

// Any value below ~2000 q with normalised limbs -> its canonical representative.
PLONK_HD Fq28 q28_canon(const Fq28 &x) { return q28_reduce_once(q28_mul_lazy(x, q28_one())); }

// ---------------------------------------------------------------------------------------------
// conversions to and from the 12 x 32-bit Montgomery (radix 2^384) form of field.cuh
// ---------------------------------------------------------------------------------------------
// plain re-slicing of a 384-bit integer (< 2^392 trivially)
PLONK_HD Fq28 q28_pack(const Fq &a) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        const int bit = 28 * i, w = bit >> 5, sh = bit & 31;
        uint64_t lo = a.l[w];
        uint64_t hi = (w + 1 < FqCfg::N) ? a.l[w + 1] : 0u;
        r.l[i] = (uint32_t)(((lo | (hi << 32)) >> sh)) & Q28::MASK;
    }
    return r;
}
PLONK_HD Fq q28_unpack(const Fq28 &a) {
    Fq r;
#pragma unroll
    for (int w = 0; w < FqCfg::N; w++) {
        // bits [32w, 32w+32) come from limbs floor(32w/28) and the next one (and maybe a third)
        uint64_t v = 0;
        const int first = (32 * w) / 28;
#pragma unroll
        for (int i = first; i < first + 3 && i < Q28::N; i++) {
            int shift = 28 * i - 32 * w;  // position of limb i relative to the word
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

// Fq (radix 2^384 Montgomery) -> Fq28 (radix 2^392 Montgomery) and back
PLONK_HD Fq28 q28_from_fq(const Fq &a) {
    Fq28 c;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) c.l[i] = Q28::to28(i);
    return q28_mul(q28_pack(a), c);
}
PLONK_HD Fq q28_to_fq(const Fq28 &a) {
    Fq28 c;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) c.l[i] = Q28::from28(i);
    return q28_unpack(q28_mul(a, c));
}

}  // namespace plonk
