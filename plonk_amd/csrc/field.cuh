// BLS12-381 prime-field arithmetic on 32-bit limbs for the gfx950 integer VALU.
//
// Fq (381 bits) = 12 limbs, Montgomery radix R = 2^384; Fr (255 bits) = 8 limbs, R = 2^256.
// Two u32 limbs are one u64 limb of ark-ff's in-memory form, so buffers handed over by the
// caller (`[u64; 6]` / `[u64; 4]` Montgomery limbs, little endian) are used as they are.
//
// Every function keeps values fully reduced (< p): the curve formulas compare field elements
// for equality (P == Q, P == -Q, Z == 0), which needs one representative per residue.
//
// All routines are __host__ __device__ so that the same code is unit-tested on the CPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PLONK_HD __host__ __device__ __forceinline__
#else
#define PLONK_HD inline __attribute__((always_inline))
#endif

namespace plonk {

// ---------------------------------------------------------------------------------------------
// Field descriptors.  Constants are returned by constexpr functions so that, after full
// unrolling, every use becomes an instruction literal (no constant-memory loads).
// ---------------------------------------------------------------------------------------------
struct FqCfg {
    static constexpr int N = 12;
    // q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    static PLONK_HD constexpr uint32_t mod(int i) {
        constexpr uint32_t m[N] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                                   0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
        return m[i];
    }
    // -q^-1 mod 2^32
    static constexpr uint32_t INV = 0xfffcfffdu;
    // R mod q
    static PLONK_HD constexpr uint32_t one(int i) {
        constexpr uint32_t m[N] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u,
                                   0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
        return m[i];
    }
    // R^2 mod q
    static PLONK_HD constexpr uint32_t r2(int i) {
        constexpr uint32_t m[N] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu,
                                   0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};
        return m[i];
    }
};

struct FrCfg {
    static constexpr int N = 8;
    // r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    static PLONK_HD constexpr uint32_t mod(int i) {
        constexpr uint32_t m[N] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                   0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        return m[i];
    }
    static constexpr uint32_t INV = 0xffffffffu;
    static PLONK_HD constexpr uint32_t one(int i) {
        constexpr uint32_t m[N] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                   0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
        return m[i];
    }
    static PLONK_HD constexpr uint32_t r2(int i) {
        constexpr uint32_t m[N] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                   0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
        return m[i];
    }
};

// A field element: N little-endian 32-bit limbs, Montgomery form unless stated otherwise.
// 16-byte alignment lets the compiler move elements with dwordx4 loads and stores.
template <class C>
struct alignas(16) Fe {
    uint32_t l[C::N];
};
using Fq = Fe<FqCfg>;
using Fr = Fe<FrCfg>;

template <class C>
PLONK_HD Fe<C> fe_zero() {
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = 0;
    return r;
}
template <class C>
PLONK_HD Fe<C> fe_one() {
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < C::N; i++) r.l[i] = C::one(i);
    return r;
}
template <class C>
PLONK_HD bool fe_is_zero(const Fe<C> &a) {
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) x |= a.l[i];
    return x == 0;
}
template <class C>
PLONK_HD bool fe_eq(const Fe<C> &a, const Fe<C> &b) {
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) x |= a.l[i] ^ b.l[i];
    return x == 0;
}

// r = a - p if a >= p else a   (a < 2p)
template <class C>
PLONK_HD void fe_reduce_once(Fe<C> &a) {
    uint32_t t[C::N];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) {
        uint64_t d = (uint64_t)a.l[i] - C::mod(i) - borrow;
        t[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    // borrow == 0  <=>  a >= p
#pragma unroll
    for (int i = 0; i < C::N; i++) a.l[i] = borrow ? a.l[i] : t[i];
}

template <class C>
PLONK_HD Fe<C> fe_add(const Fe<C> &a, const Fe<C> &b) {
    Fe<C> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) {
        c += (uint64_t)a.l[i] + b.l[i];
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    // both moduli leave the top bit of the top limb clear, so a + b < 2p never carries out
    fe_reduce_once(r);
    return r;
}

template <class C>
PLONK_HD Fe<C> fe_sub(const Fe<C> &a, const Fe<C> &b) {
    Fe<C> r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) {
        uint64_t d = (uint64_t)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    uint32_t mask = borrow ? 0xffffffffu : 0u;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) {
        c += (uint64_t)r.l[i] + (C::mod(i) & mask);
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

template <class C>
PLONK_HD Fe<C> fe_neg(const Fe<C> &a) {
    return fe_sub(fe_zero<C>(), a);
}

template <class C>
PLONK_HD Fe<C> fe_dbl(const Fe<C> &a) {
    Fe<C> r;
    uint32_t prev = 0;
#pragma unroll
    for (int i = 0; i < C::N; i++) {
        r.l[i] = (a.l[i] << 1) | (prev >> 31);
        prev = a.l[i];
    }
    fe_reduce_once(r);
    return r;
}

// Montgomery product a*b/R mod p, portable form: coarsely integrated operand scanning on 32-bit
// limbs.  Both moduli are below 2^(32N-1), so the running value stays below 2p and N+1 limbs
// suffice ("no-carry" variant).
template <class C>
PLONK_HD Fe<C> fe_mul_cios(const Fe<C> &a, const Fe<C> &b) {
    constexpr int N = C::N;
    uint32_t t[N + 1];
#pragma unroll
    for (int i = 0; i <= N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            uint64_t s = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (uint32_t)s;
            c = s >> 32;
        }
        t[N] = (uint32_t)c;
        uint32_t m = t[0] * C::INV;
        uint64_t s = (uint64_t)m * C::mod(0) + t[0];
        c = s >> 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            s = (uint64_t)m * C::mod(j) + t[j] + c;
            t[j - 1] = (uint32_t)s;
            c = s >> 32;
        }
        s = (uint64_t)t[N] + c;
        t[N - 1] = (uint32_t)s;
    }
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    fe_reduce_once(r);
    return r;
}

template <class C>
PLONK_HD Fe<C> fe_mul(const Fe<C> &a, const Fe<C> &b) {
    return fe_mul_cios(a, b);
}

template <class C>
PLONK_HD Fe<C> fe_sqr(const Fe<C> &a) {
    return fe_mul(a, a);
}

// Montgomery form -> canonical integer limbs: a / R mod p, i.e. the reduction half of a Montgomery product
// alone (N rounds of "add the multiple of p that clears the lowest limb, drop that limb"; the running value
// stays below 2^(32N) as both moduli leave the top bit of the top limb clear, and ends at most p).
template <class C>
PLONK_HD Fe<C> fe_from_mont(const Fe<C> &a) {
    constexpr int N = C::N;
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = a.l[i];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint32_t m = t[0] * C::INV;
        uint64_t c = ((uint64_t)m * C::mod(0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            const uint64_t s = (uint64_t)m * C::mod(j) + t[j] + c;
            t[j - 1] = (uint32_t)s;
            c = s >> 32;
        }
        t[N - 1] = (uint32_t)c;
    }
    Fe<C> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    fe_reduce_once(r);
    return r;
}
template <class C>
PLONK_HD Fe<C> fe_to_mont(const Fe<C> &a) {
    Fe<C> r2;
#pragma unroll
    for (int i = 0; i < C::N; i++) r2.l[i] = C::r2(i);
    return fe_mul(a, r2);
}

// a^(p-2): inverse by Fermat (0 -> 0).  Serial: ~1.5 N*32 multiplications.
template <class C>
PLONK_HD Fe<C> fe_inv(const Fe<C> &a) {
    Fe<C> acc = fe_one<C>();
    // exponent p - 2 (limb-wise with borrow), most significant bit first
    uint32_t ex[C::N];
    uint32_t borrow = 2;
#pragma unroll
    for (int i = 0; i < C::N; i++) {
        uint32_t m = C::mod(i);
        ex[i] = m - borrow;
        borrow = m < borrow ? 1u : 0u;
    }
    for (int i = C::N - 1; i >= 0; i--) {
        uint32_t e = 0;
#pragma unroll
        for (int j = 0; j < C::N; j++) e = (i == j) ? ex[j] : e;
        for (int b = 31; b >= 0; b--) {
            acc = fe_sqr(acc);
            if ((e >> b) & 1) acc = fe_mul(acc, a);
        }
    }
    return acc;
}

// Small exponent power (square and multiply, low bit first).
template <class C>
PLONK_HD Fe<C> fe_pow_u64(Fe<C> base, uint64_t e) {
    Fe<C> acc = fe_one<C>();
    while (e) {
        if (e & 1) acc = fe_mul(acc, base);
        base = fe_sqr(base);
        e >>= 1;
    }
    return acc;
}

}  // namespace plonk
