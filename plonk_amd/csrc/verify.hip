// Verifier side of KZG: kzg_verify (plonk/src/common/kzg.rs:91-98) and the folding of many such checks into
// one pairing-product check.  Host code only: a proof's 15 openings cost two Miller loops and one final
// exponentiation after folding; there is nothing here for a GPU to do.
//
//   e(com - v G, H) == e(proof, tau H - u H)   <=>   e(com - v G + u proof, H) * e(-proof, tau H) == 1
//
// so every check -- and every random linear combination of checks,
//   L = sum rho_i (com_i - v_i G + u_i proof_i),   R = sum rho_i proof_i,   e(L, H) * e(-R, tau H) == 1 --
// pairs G1 points with the two FIXED G2 points H and tau H only.  Their Miller-loop lines are computed once
// (per distinct tau H) and kept; G2 arithmetic never runs per check.
//
// The pairing is the optimal ate pairing of BLS12-381 raised to three times the usual final exponent (the
// addition chain in the curve parameter x computes that power; 3 does not divide r, so it is as good a
// pairing): the reference's ark-bls12-381 only ever compares pairings for equality (kzg.rs:97).
// Plain arithmetic on purpose: Fq2 = Fq[u] / (u^2 + 1); Fq12 as polynomials of degree < 6 in w over Fq2 with
// w^6 = xi = 1 + u (the tower Fq6 = Fq2[v] / (v^3 - xi), Fq12 = Fq6[w] / (w^2 - v) read as even and odd
// coefficients); the point T of the Miller loop affine on the twist y^2 = x^3 + 4 xi.  Frobenius constants
// are computed at first use, not tabulated.
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/plonk_hip.h"
#include "g1.cuh"

namespace plonk {
namespace verify {

// ---------------------------------------------------------------------------------------- Fq on the host
// Six 64-bit limbs, Montgomery radix 2^384: the same bytes as the 12 x 32-bit form of field.cuh (this is
// little-endian host code), about four times faster here than that portable form.
struct Fp {
    uint64_t l[6];
};
typedef unsigned __int128 u128;
static const uint64_t kP[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                               0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const uint64_t kPInv = 0x89f3fffcfffcfffdull;  // -p^-1 mod 2^64
static const Fp kFpOne = {{0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull, 0x77ce585370525745ull,
                           0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull}};
static const Fp kFpR2 = {{0xf4df1f341c341746ull, 0x0a76e6a609d104f1ull, 0x8de5476c4c95b6d5ull, 0x67eb88a9939d83c0ull,
                          0x9a793e85b519952dull, 0x11988fe592cae3aaull}};
static bool fp_geq_p(const uint64_t a[6]) {
    for (int i = 5; i >= 0; i--) {
        if (a[i] != kP[i]) return a[i] > kP[i];
    }
    return true;
}
static void fp_sub_p(uint64_t a[6]) {
    u128 borrow = 0;
    for (int i = 0; i < 6; i++) {
        const u128 d = (u128)a[i] - kP[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
}
static Fp fe_add(const Fp &a, const Fp &b) {
    Fp r;
    u128 c = 0;
    for (int i = 0; i < 6; i++) {
        c += (u128)a.l[i] + b.l[i];
        r.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (fp_geq_p(r.l)) fp_sub_p(r.l);  // a + b < 2 p < 2^384
    return r;
}
static Fp fe_sub(const Fp &a, const Fp &b) {
    Fp r;
    u128 borrow = 0;
    for (int i = 0; i < 6; i++) {
        const u128 d = (u128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    if (borrow) {
        u128 c = 0;
        for (int i = 0; i < 6; i++) {
            c += (u128)r.l[i] + kP[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    return r;
}
static bool fe_is_zero(const Fp &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }
static bool fe_eq(const Fp &a, const Fp &b) { return memcmp(a.l, b.l, sizeof(a.l)) == 0; }
static Fp fe_neg(const Fp &a) {
    const Fp zero = {{0, 0, 0, 0, 0, 0}};
    return fe_is_zero(a) ? a : fe_sub(zero, a);
}
static Fp fe_dbl(const Fp &a) { return fe_add(a, a); }
static Fp fe_mul(const Fp &a, const Fp &b) {
    // coarsely integrated operand scanning, fully unrolled: t stays in registers
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const uint64_t bi = b.l[i];
        u128 c = (u128)a.l[0] * bi + t0;
        const uint64_t lo0 = (uint64_t)c;
        c = (c >> 64) + (u128)a.l[1] * bi + t1;
        const uint64_t lo1 = (uint64_t)c;
        c = (c >> 64) + (u128)a.l[2] * bi + t2;
        const uint64_t lo2 = (uint64_t)c;
        c = (c >> 64) + (u128)a.l[3] * bi + t3;
        const uint64_t lo3 = (uint64_t)c;
        c = (c >> 64) + (u128)a.l[4] * bi + t4;
        const uint64_t lo4 = (uint64_t)c;
        c = (c >> 64) + (u128)a.l[5] * bi + t5;
        const uint64_t lo5 = (uint64_t)c;
        c = (c >> 64) + t6;
        const uint64_t lo6 = (uint64_t)c, lo7 = (uint64_t)(c >> 64);
        const uint64_t m = lo0 * kPInv;
        u128 d = ((u128)m * kP[0] + lo0) >> 64;
        d += (u128)m * kP[1] + lo1;
        t0 = (uint64_t)d;
        d = (d >> 64) + (u128)m * kP[2] + lo2;
        t1 = (uint64_t)d;
        d = (d >> 64) + (u128)m * kP[3] + lo3;
        t2 = (uint64_t)d;
        d = (d >> 64) + (u128)m * kP[4] + lo4;
        t3 = (uint64_t)d;
        d = (d >> 64) + (u128)m * kP[5] + lo5;
        t4 = (uint64_t)d;
        d = (d >> 64) + lo6;
        t5 = (uint64_t)d;
        t6 = lo7 + (uint64_t)(d >> 64);
    }
    Fp r = {{t0, t1, t2, t3, t4, t5}};
    if (t6 || fp_geq_p(r.l)) fp_sub_p(r.l);
    return r;
}
static Fp fe_sqr(const Fp &a) { return fe_mul(a, a); }
static Fp fe_to_mont(const Fp &a) { return fe_mul(a, kFpR2); }
static Fp fe_from_mont(const Fp &a) {
    const Fp one = {{1, 0, 0, 0, 0, 0}};
    return fe_mul(a, one);
}
static Fp fe_inv(const Fp &a) {  // a^(p - 2)
    uint64_t e[6];
    memcpy(e, kP, sizeof(e));
    e[0] -= 2;
    Fp acc = kFpOne;
    for (int i = 5; i >= 0; i--) {
        for (int b = 63; b >= 0; b--) {
            acc = fe_mul(acc, acc);
            if ((e[i] >> b) & 1) acc = fe_mul(acc, a);
        }
    }
    return acc;
}
static Fp fp_from(const Fq &a) {
    Fp r;
    memcpy(r.l, a.l, sizeof(r.l));
    return r;
}

// ---------------------------------------------------------------------------------------- Fq2
struct Fq2 {
    Fp c0, c1;
};
static Fp fq_zero() { return Fp{{0, 0, 0, 0, 0, 0}}; }
static Fp fq_one() { return kFpOne; }
static Fq2 f2_zero() { return {fq_zero(), fq_zero()}; }
static Fq2 f2_one() { return {fq_one(), fq_zero()}; }
static bool f2_is_zero(const Fq2 &a) { return fe_is_zero(a.c0) && fe_is_zero(a.c1); }
static bool f2_eq(const Fq2 &a, const Fq2 &b) { return fe_eq(a.c0, b.c0) && fe_eq(a.c1, b.c1); }
static Fq2 f2_add(const Fq2 &a, const Fq2 &b) { return {fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)}; }
static Fq2 f2_sub(const Fq2 &a, const Fq2 &b) { return {fe_sub(a.c0, b.c0), fe_sub(a.c1, b.c1)}; }
static Fq2 f2_neg(const Fq2 &a) { return {fe_neg(a.c0), fe_neg(a.c1)}; }
static Fq2 f2_conj(const Fq2 &a) { return {a.c0, fe_neg(a.c1)}; }
static Fq2 f2_mul(const Fq2 &a, const Fq2 &b) {
    const Fp t0 = fe_mul(a.c0, b.c0), t1 = fe_mul(a.c1, b.c1);
    const Fp s = fe_mul(fe_add(a.c0, a.c1), fe_add(b.c0, b.c1));
    return {fe_sub(t0, t1), fe_sub(fe_sub(s, t0), t1)};
}
static Fq2 f2_sqr(const Fq2 &a) {
    // (a0 + a1)(a0 - a1) + 2 a0 a1 u
    const Fp t = fe_mul(a.c0, a.c1);
    return {fe_mul(fe_add(a.c0, a.c1), fe_sub(a.c0, a.c1)), fe_dbl(t)};
}
static Fq2 f2_scale(const Fq2 &a, const Fp &k) { return {fe_mul(a.c0, k), fe_mul(a.c1, k)}; }
static Fq2 f2_mul_xi(const Fq2 &a) { return {fe_sub(a.c0, a.c1), fe_add(a.c0, a.c1)}; }
static Fq2 f2_inv(const Fq2 &a) {
    const Fp n = fe_inv(fe_add(fe_sqr(a.c0), fe_sqr(a.c1)));
    return {fe_mul(a.c0, n), fe_neg(fe_mul(a.c1, n))};
}
// exponent: 6 little-endian 64-bit words
static Fq2 f2_pow(const Fq2 &a, const uint64_t e[6]) {
    Fq2 acc = f2_one();
    for (int i = 5; i >= 0; i--) {
        for (int b = 63; b >= 0; b--) {
            acc = f2_sqr(acc);
            if ((e[i] >> b) & 1) acc = f2_mul(acc, a);
        }
    }
    return acc;
}
static Fp fq_from_u64(uint64_t v) {
    Fp a = fq_zero();
    a.l[0] = v;
    return fe_to_mont(a);
}
// (p - 1) / 6, (p - 3) / 4, (p - 1) / 2
static const uint64_t kExpSixth[6] = {0x49aa7ffffffff1c7ull, 0x051caaaa72e35555ull, 0xe688231ad3c82906ull,
                                      0xe613e1eb7deb831full, 0x0c849bf3b5e1f223ull, 0x045582fc5eeaa66full};
static const uint64_t kExpSqrt1[6] = {0xee7fbfffffffeaaaull, 0x07aaffffac54ffffull, 0xd9cc34a83dac3d89ull,
                                      0xd91dd2e13ce144afull, 0x92c6e9ed90d2eb35ull, 0x0680447a8e5ff9a6ull};
static const uint64_t kExpSqrt2[6] = {0xdcff7fffffffd555ull, 0x0f55ffff58a9ffffull, 0xb39869507b587b12ull,
                                      0xb23ba5c279c2895full, 0x258dd3db21a5d66bull, 0x0d0088f51cbff34dull};

// a square root of a, if there is one (p = 3 mod 4: the complex method)
static bool f2_sqrt(const Fq2 &a, Fq2 &root) {
    if (f2_is_zero(a)) {
        root = a;
        return true;
    }
    const Fq2 a1 = f2_pow(a, kExpSqrt1);
    const Fq2 alpha = f2_mul(a1, f2_mul(a1, a));
    const Fq2 x0 = f2_mul(a1, a);
    const Fq2 minus_one = f2_neg(f2_one());
    if (f2_eq(alpha, minus_one)) {
        root = f2_mul(Fq2{fq_zero(), fq_one()}, x0);
    } else {
        root = f2_mul(f2_pow(f2_add(f2_one(), alpha), kExpSqrt2), x0);
    }
    return f2_eq(f2_sqr(root), a);
}

// ---------------------------------------------------------------------------------------- Fq12
struct F12 {
    Fq2 c[6];  // coefficient of w^i
};
static F12 f12_one() {
    F12 r;
    for (int i = 0; i < 6; i++) r.c[i] = f2_zero();
    r.c[0] = f2_one();
    return r;
}
static bool f12_is_one(const F12 &a) {
    if (!f2_eq(a.c[0], f2_one())) return false;
    for (int i = 1; i < 6; i++) {
        if (!f2_is_zero(a.c[i])) return false;
    }
    return true;
}
static F12 f12_mul(const F12 &a, const F12 &b) {
    Fq2 t[11];
    for (int i = 0; i < 11; i++) t[i] = f2_zero();
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j < 6; j++) t[i + j] = f2_add(t[i + j], f2_mul(a.c[i], b.c[j]));
    }
    F12 r;
    for (int k = 0; k < 6; k++) r.c[k] = k < 5 ? f2_add(t[k], f2_mul_xi(t[k + 6])) : t[k];
    return r;
}
// times l0 + l2 w^2 + l3 w^3 with l3 in Fq: what a Miller-loop line looks like
static F12 f12_mul_line(const F12 &a, const Fq2 &l0, const Fq2 &l2, const Fp &l3) {
    Fq2 t[11];
    for (int i = 0; i < 11; i++) t[i] = f2_zero();
    for (int i = 0; i < 6; i++) {
        t[i] = f2_add(t[i], f2_mul(a.c[i], l0));
        t[i + 2] = f2_add(t[i + 2], f2_mul(a.c[i], l2));
        t[i + 3] = f2_add(t[i + 3], f2_scale(a.c[i], l3));
    }
    F12 r;
    for (int k = 0; k < 6; k++) r.c[k] = k < 5 ? f2_add(t[k], f2_mul_xi(t[k + 6])) : t[k];
    return r;
}
// a^(p^6): w -> -w
static F12 f12_conj(const F12 &a) {
    F12 r = a;
    for (int i = 1; i < 6; i += 2) r.c[i] = f2_neg(a.c[i]);
    return r;
}

// Fq6 = Fq2[v] / (v^3 - xi) for the inversion
struct F6 {
    Fq2 a0, a1, a2;
};
static F6 f6_mul(const F6 &x, const F6 &y) {
    return {f2_add(f2_mul(x.a0, y.a0), f2_mul_xi(f2_add(f2_mul(x.a1, y.a2), f2_mul(x.a2, y.a1)))),
            f2_add(f2_add(f2_mul(x.a0, y.a1), f2_mul(x.a1, y.a0)), f2_mul_xi(f2_mul(x.a2, y.a2))),
            f2_add(f2_add(f2_mul(x.a0, y.a2), f2_mul(x.a1, y.a1)), f2_mul(x.a2, y.a0))};
}
static F6 f6_sub(const F6 &x, const F6 &y) { return {f2_sub(x.a0, y.a0), f2_sub(x.a1, y.a1), f2_sub(x.a2, y.a2)}; }
static F6 f6_mul_v(const F6 &x) { return {f2_mul_xi(x.a2), x.a0, x.a1}; }
static F6 f6_inv(const F6 &x) {
    const Fq2 t0 = f2_sub(f2_sqr(x.a0), f2_mul_xi(f2_mul(x.a1, x.a2)));
    const Fq2 t1 = f2_sub(f2_mul_xi(f2_sqr(x.a2)), f2_mul(x.a0, x.a1));
    const Fq2 t2 = f2_sub(f2_sqr(x.a1), f2_mul(x.a0, x.a2));
    const Fq2 d = f2_add(f2_mul(x.a0, t0), f2_mul_xi(f2_add(f2_mul(x.a2, t1), f2_mul(x.a1, t2))));
    const Fq2 di = f2_inv(d);
    return {f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di)};
}
// (A + B w)^-1 = (A - B w) / (A^2 - v B^2)
static F12 f12_inv(const F12 &a) {
    const F6 A = {a.c[0], a.c[2], a.c[4]}, B = {a.c[1], a.c[3], a.c[5]};
    const F6 d = f6_inv(f6_sub(f6_mul(A, A), f6_mul_v(f6_mul(B, B))));
    const F6 ra = f6_mul(A, d), rb = f6_mul(B, d);
    F12 r;
    r.c[0] = ra.a0;
    r.c[2] = ra.a1;
    r.c[4] = ra.a2;
    r.c[1] = f2_neg(rb.a0);
    r.c[3] = f2_neg(rb.a1);
    r.c[5] = f2_neg(rb.a2);
    return r;
}

// w^(p - 1) = xi^((p - 1) / 6) and w^(p^2 - 1) = its norm; their powers
struct FrobeniusTables {
    Fq2 g1[6], g2[6];
};
static const FrobeniusTables &frobenius_tables() {
    static FrobeniusTables t;
    static std::once_flag once;
    std::call_once(once, [] {
        const Fq2 xi = {fq_one(), fq_one()};
        const Fq2 gamma1 = f2_pow(xi, kExpSixth);
        const Fq2 gamma2 = f2_mul(gamma1, f2_conj(gamma1));
        t.g1[0] = t.g2[0] = f2_one();
        for (int i = 1; i < 6; i++) {
            t.g1[i] = f2_mul(t.g1[i - 1], gamma1);
            t.g2[i] = f2_mul(t.g2[i - 1], gamma2);
        }
    });
    return t;
}
static F12 f12_frobenius(const F12 &a, int power) {  // power 1 or 2
    const FrobeniusTables &t = frobenius_tables();
    F12 r;
    for (int i = 0; i < 6; i++) r.c[i] = power == 1 ? f2_mul(f2_conj(a.c[i]), t.g1[i]) : f2_mul(a.c[i], t.g2[i]);
    return r;
}

static constexpr uint64_t kXAbs = 0xd201000000010000ull;  // the curve parameter is -kXAbs

// a^x for a in the cyclotomic subgroup (where the inverse is the conjugate)
static F12 exp_by_x(const F12 &a) {
    F12 acc = a;
    for (int b = 62; b >= 0; b--) {
        acc = f12_mul(acc, acc);
        if ((kXAbs >> b) & 1) acc = f12_mul(acc, a);
    }
    return f12_conj(acc);
}

// f^(3 (p^12 - 1) / r)
static F12 final_exponentiation(const F12 &f) {
    // easy part: (p^6 - 1)(p^2 + 1)
    F12 r = f12_mul(f12_conj(f), f12_inv(f));
    r = f12_mul(f12_frobenius(r, 2), r);
    // hard part: 3 (p^4 - p^2 + 1) / r by an addition chain in x
    F12 y0 = f12_mul(r, r);
    F12 y1 = exp_by_x(r);
    F12 y2 = f12_conj(r);
    y1 = f12_mul(y1, y2);
    y2 = exp_by_x(y1);
    y1 = f12_conj(y1);
    y1 = f12_mul(y1, y2);
    y2 = exp_by_x(y1);
    y1 = f12_frobenius(y1, 1);
    y1 = f12_mul(y1, y2);
    r = f12_mul(r, y0);
    y0 = exp_by_x(y1);
    y2 = exp_by_x(y0);
    y0 = f12_frobenius(y1, 2);
    y1 = f12_conj(y1);
    y1 = f12_mul(y1, y2);
    y1 = f12_mul(y1, y0);
    return f12_mul(r, y1);
}

// ---------------------------------------------------------------------------------------- G2, prepared
struct G2Affine {
    Fq2 x, y;
    bool infinity;
};
static Fq2 twist_b() {
    const Fp four = fq_from_u64(4);
    return {four, four};
}
static bool g2_on_curve(const G2Affine &q) {
    if (q.infinity) return true;
    return f2_eq(f2_sqr(q.y), f2_add(f2_mul(f2_sqr(q.x), q.x), twist_b()));
}
static G2Affine g2_double(const G2Affine &q, Fq2 *slope = nullptr) {
    if (q.infinity || f2_is_zero(q.y)) return {f2_zero(), f2_zero(), true};
    const Fq2 x2 = f2_sqr(q.x);
    const Fq2 lam = f2_mul(f2_add(f2_add(x2, x2), x2), f2_inv(f2_add(q.y, q.y)));
    if (slope) *slope = lam;
    const Fq2 x3 = f2_sub(f2_sub(f2_sqr(lam), q.x), q.x);
    return {x3, f2_sub(f2_mul(lam, f2_sub(q.x, x3)), q.y), false};
}
static G2Affine g2_add(const G2Affine &q, const G2Affine &s, Fq2 *slope = nullptr) {
    if (q.infinity) return s;
    if (s.infinity) return q;
    if (f2_eq(q.x, s.x)) {
        if (f2_eq(q.y, s.y)) return g2_double(q, slope);
        return {f2_zero(), f2_zero(), true};
    }
    const Fq2 lam = f2_mul(f2_sub(s.y, q.y), f2_inv(f2_sub(s.x, q.x)));
    if (slope) *slope = lam;
    const Fq2 x3 = f2_sub(f2_sub(f2_sqr(lam), q.x), s.x);
    return {x3, f2_sub(f2_mul(lam, f2_sub(q.x, x3)), q.y), false};
}
// r Q == identity (r as the 8 x 32-bit limbs of FrCfg)
static bool g2_in_subgroup(const G2Affine &q) {
    G2Affine acc = {f2_zero(), f2_zero(), true};
    for (int i = FrCfg::N - 1; i >= 0; i--) {
        for (int b = 31; b >= 0; b--) {
            acc = g2_double(acc);
            if ((FrCfg::mod(i) >> b) & 1) acc = g2_add(acc, q);
        }
    }
    return acc.infinity;
}

static Fp fq_from_be48(const uint8_t *b) {
    Fp a;
    for (int i = 0; i < 6; i++) {
        const uint8_t *w = b + 40 - 8 * i;
        uint64_t v = 0;
        for (int k = 0; k < 8; k++) v = (v << 8) | w[k];
        a.l[i] = v;
    }
    return a;
}
static bool fq_canonical(const Fp &a) { return !fp_geq_p(a.l); }
// y > -y in the order of the encoding: c1 first, then c0, as integers
static bool f2_lexicographically_largest(const Fq2 &y) {
    const Fq2 ny = f2_neg(y);
    const Fp a1 = fe_from_mont(y.c1), b1 = fe_from_mont(ny.c1), a0 = fe_from_mont(y.c0), b0 = fe_from_mont(ny.c0);
    for (int i = 5; i >= 0; i--) {
        if (a1.l[i] != b1.l[i]) return a1.l[i] > b1.l[i];
    }
    for (int i = 5; i >= 0; i--) {
        if (a0.l[i] != b0.l[i]) return a0.l[i] > b0.l[i];
    }
    return false;
}
// The 96-byte compressed form of ark-bls12-381 (plonk/data/srs.json, "tau_g2"): x.c1 then x.c0 big-endian, top
// three bits of byte 0 = compressed, infinity, y lexicographically largest.
static bool g2_decompress(const uint8_t in[96], G2Affine &out) {
    if (!(in[0] & 0x80)) return false;
    if (in[0] & 0x40) {
        if (in[0] & 0x3f) return false;
        for (int i = 1; i < 96; i++) {
            if (in[i]) return false;
        }
        out = {f2_zero(), f2_zero(), true};
        return true;
    }
    uint8_t buf[96];
    memcpy(buf, in, 96);
    buf[0] &= 0x1f;
    const Fp c1 = fq_from_be48(buf), c0 = fq_from_be48(buf + 48);
    if (!fq_canonical(c0) || !fq_canonical(c1)) return false;
    out.x = {fe_to_mont(c0), fe_to_mont(c1)};
    out.infinity = false;
    if (!f2_sqrt(f2_add(f2_mul(f2_sqr(out.x), out.x), twist_b()), out.y)) return false;
    if (f2_lexicographically_largest(out.y) != ((in[0] & 0x20) != 0)) out.y = f2_neg(out.y);
    return true;
}

// The lines of the Miller loop for a fixed Q: at T with slope lam the line through the untwisted points,
// evaluated at P = (xP, yP) in G1 and scaled by w^3 (an element of a proper subfield, which the final
// exponentiation sends to 1), is (lam x_T - y_T) - lam x_P w^2 + y_P w^3: per step the constant term and lam.
struct Prepared {
    std::vector<Fq2> l0, lam;  // in the order the loop uses them
    bool infinity = false;
};
static Prepared prepare(const G2Affine &q) {
    Prepared p;
    if (q.infinity) {
        p.infinity = true;
        return p;
    }
    G2Affine t = q;
    for (int b = 62; b >= 0; b--) {
        Fq2 lam;
        const G2Affine t2 = g2_double(t, &lam);
        p.l0.push_back(f2_sub(f2_mul(lam, t.x), t.y));
        p.lam.push_back(lam);
        t = t2;
        if ((kXAbs >> b) & 1) {
            const G2Affine t3 = g2_add(t, q, &lam);
            p.l0.push_back(f2_sub(f2_mul(lam, t.x), t.y));
            p.lam.push_back(lam);
            t = t3;
        }
    }
    return p;
}

// f <- f * (Miller function of (P, Q)) for affine P != identity
static void miller_accumulate(F12 &f, const Fp &xp, const Fp &yp, const Prepared &q) {
    F12 g = f12_one();
    size_t k = 0;
    for (int b = 62; b >= 0; b--) {
        g = f12_mul(g, g);
        g = f12_mul_line(g, q.l0[k], f2_neg(f2_scale(q.lam[k], xp)), yp);
        k++;
        if ((kXAbs >> b) & 1) {
            g = f12_mul_line(g, q.l0[k], f2_neg(f2_scale(q.lam[k], xp)), yp);
            k++;
        }
    }
    f = f12_mul(f, f12_conj(g));  // x is negative
}

// generator of G2
static G2Affine g2_generator() {
    static const uint64_t xc0[6] = {0xd48056c8c121bdb8ull, 0x0bac0326a805bbefull, 0xb4510b647ae3d177ull,
                                    0xc6e47ad4fa403b02ull, 0x260805272dc51051ull, 0x024aa2b2f08f0a91ull};
    static const uint64_t xc1[6] = {0xe5ac7d055d042b7eull, 0x334cf11213945d57ull, 0xb5da61bbdc7f5049ull,
                                    0x596bd0d09920b61aull, 0x7dacd3a088274f65ull, 0x13e02b6052719f60ull};
    static const uint64_t yc0[6] = {0xe193548608b82801ull, 0x923ac9cc3baca289ull, 0x6d429a695160d12cull,
                                    0xadfd9baa8cbdd3a7ull, 0x8cc9cdc6da2e351aull, 0x0ce5d527727d6e11ull};
    static const uint64_t yc1[6] = {0xaaa9075ff05f79beull, 0x3f370d275cec1da1ull, 0x267492ab572e99abull,
                                    0xcb3e287e85a763afull, 0x32acd2b02bc28b99ull, 0x0606c4a02ea734ccull};
    auto load = [](const uint64_t v[6]) {
        Fp a;
        memcpy(a.l, v, sizeof(a.l));
        return fe_to_mont(a);
    };
    return {{load(xc0), load(xc1)}, {load(yc0), load(yc1)}, false};
}

struct PreparedPair {
    uint8_t tau_encoding[96];
    Prepared h, tau_h;
    bool valid = false;
};
// The prepared (H, tau H) of the last tau_g2 seen, copied out under the lock (callers on other threads may
// replace the cached pair at any time); -1: not a valid G2 encoding / not in the subgroup.
static int prepared_for(const uint8_t tau_g2[96], PreparedPair *out) {
    static PreparedPair cache;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!cache.valid || memcmp(cache.tau_encoding, tau_g2, 96) != 0) {
        G2Affine tau;
        if (!g2_decompress(tau_g2, tau) || !g2_on_curve(tau) || !g2_in_subgroup(tau)) return -1;
        cache.valid = false;
        cache.h = prepare(g2_generator());
        cache.tau_h = prepare(tau);
        memcpy(cache.tau_encoding, tau_g2, 96);
        cache.valid = true;
    }
    *out = cache;
    return 0;
}

// ---------------------------------------------------------------------------------------- G1 side
static void load_jac(const uint64_t limbs[18], G1Jac &p) { memcpy(&p, limbs, sizeof(G1Jac)); }
static Fr load_fr(const uint64_t v[4]) {
    Fr r;
    memcpy(r.l, v, sizeof(r.l));
    return r;
}
// G1 on the 64-bit form: XYZZ accumulator (x = X / ZZ, y = Y / ZZZ; ZZ = 0 for the identity), affine addends
struct P1 {
    Fp x, y, zz, zzz;
};
struct A1 {
    Fp x, y;
};
static P1 p1_identity() { return {fq_zero(), fq_zero(), fq_zero(), fq_zero()}; }
static P1 p1_dbl(const P1 &p) {
    if (fe_is_zero(p.zz) || fe_is_zero(p.y)) return p1_identity();
    const Fp u = fe_dbl(p.y), v = fe_sqr(u), w = fe_mul(u, v), s = fe_mul(p.x, v);
    const Fp x2 = fe_sqr(p.x), m = fe_add(fe_dbl(x2), x2);
    const Fp x3 = fe_sub(fe_sqr(m), fe_dbl(s));
    return {x3, fe_sub(fe_mul(m, fe_sub(s, x3)), fe_mul(w, p.y)), fe_mul(v, p.zz), fe_mul(w, p.zzz)};
}
static void p1_madd(P1 &acc, const A1 &q) {
    if (fe_is_zero(acc.zz)) {
        acc = {q.x, q.y, fq_one(), fq_one()};
        return;
    }
    const Fp pp0 = fe_sub(fe_mul(q.x, acc.zz), acc.x), r = fe_sub(fe_mul(q.y, acc.zzz), acc.y);
    if (fe_is_zero(pp0)) {
        if (fe_is_zero(r)) {
            acc = p1_dbl({q.x, q.y, fq_one(), fq_one()});
        } else {
            acc = p1_identity();
        }
        return;
    }
    const Fp pp = fe_sqr(pp0), ppp = fe_mul(pp0, pp), qq = fe_mul(acc.x, pp);
    const Fp x3 = fe_sub(fe_sub(fe_sqr(r), ppp), fe_dbl(qq));
    acc = {x3, fe_sub(fe_mul(r, fe_sub(qq, x3)), fe_mul(acc.y, ppp)), fe_mul(acc.zz, pp), fe_mul(acc.zzz, ppp)};
}
static bool p1_to_affine(const P1 &p, A1 &out) {
    if (fe_is_zero(p.zz)) return false;
    const Fp zi = fe_inv(fe_mul(p.zz, p.zzz));
    out.x = fe_mul(p.x, fe_mul(zi, p.zzz));
    out.y = fe_mul(p.y, fe_mul(zi, p.zz));
    return true;
}
// Jacobian (X, Y, Z) of the interface -> affine; false for the identity
static bool jac_to_affine(const G1Jac &p, A1 &out) {
    const Fp z = fp_from(p.z);
    if (fe_is_zero(z)) return false;
    const Fp zi = fe_inv(z), zi2 = fe_sqr(zi);
    out.x = fe_mul(fp_from(p.x), zi2);
    out.y = fe_mul(fp_from(p.y), fe_mul(zi2, zi));
    return true;
}
// sum k_i P_i by one joint double-and-add (scalars: 8 x 32-bit little-endian integers)
struct Term {
    A1 p;
    uint32_t k[8];
    bool skip;
};
static P1 joint_scalar_mul(const std::vector<Term> &terms) {
    P1 acc = p1_identity();
    for (int bit = 255; bit >= 0; bit--) {
        acc = p1_dbl(acc);
        for (const Term &t : terms) {
            if (!t.skip && ((t.k[bit >> 5] >> (bit & 31)) & 1)) p1_madd(acc, t.p);
        }
    }
    return acc;
}
// y^2 == x^3 + 4: what `G1::deserialize_compressed` guarantees for every point the reference's kzg_verify sees.
// (Membership in the prime-order subgroup is the caller's precondition: points produced by this library or decoded
// with the subgroup check are in it.)
static bool a1_on_curve(const A1 &p) {
    const Fp two = fe_dbl(fq_one()), four = fe_dbl(two);
    return fe_eq(fe_sqr(p.y), fe_add(fe_mul(fe_sqr(p.x), p.x), four));
}
static Term make_term(const G1Jac &p, const Fr &k_mont) {
    Term t;
    t.skip = !jac_to_affine(p, t.p);
    const Fr k = fe_from_mont(k_mont);
    memcpy(t.k, k.l, sizeof(t.k));
    return t;
}
static Fr fr_from_u128(const uint64_t w[2]) {
    Fr a = fe_zero<FrCfg>();
    a.l[0] = (uint32_t)w[0];
    a.l[1] = (uint32_t)(w[0] >> 32);
    a.l[2] = (uint32_t)w[1];
    a.l[3] = (uint32_t)(w[1] >> 32);
    return fe_to_mont(a);
}

static int verify_folded(const uint8_t tau_g2[96], size_t n, const uint64_t *coms, const uint64_t *us,
                         const uint64_t *vs, const uint64_t *proofs, const uint64_t *rho, int *ok) {
    PreparedPair pair;
    if (prepared_for(tau_g2, &pair) != 0) return PLONK_HIP_ERR_INVALID_POINT;
    const PreparedPair *g2 = &pair;
    // L = sum rho_i com_i + sum (rho_i u_i) proof_i - (sum rho_i v_i) G,   R = sum rho_i proof_i
    std::vector<Term> left, right;
    left.reserve(2 * n + 1);
    right.reserve(n);
    Fr v_total = fe_zero<FrCfg>();
    const uint64_t one[2] = {1, 0};
    for (size_t i = 0; i < n; i++) {
        G1Jac com, proof;
        load_jac(coms + 18 * i, com);
        load_jac(proofs + 18 * i, proof);
        const Fr w = fr_from_u128(rho ? rho + 2 * i : one);
        const Fr u = load_fr(us + 4 * i), v = load_fr(vs + 4 * i);
        left.push_back(make_term(com, w));
        left.push_back(make_term(proof, fe_mul(w, u)));
        // a pairing of points off the curve means nothing, whatever it evaluates to
        for (size_t k = left.size() - 2; k < left.size(); k++)
            if (!left[k].skip && !a1_on_curve(left[k].p)) return PLONK_HIP_ERR_INVALID_POINT;
        right.push_back(make_term(proof, w));
        v_total = fe_add(v_total, fe_mul(w, v));
    }
    {
        // - (sum rho_i v_i) G
        G1Jac gen;
        const uint64_t gx[6] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull,
                                0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull};
        const uint64_t gy[6] = {0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull,
                                0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};
        Fp x, y;
        memcpy(x.l, gx, sizeof(x.l));
        memcpy(y.l, gy, sizeof(y.l));
        x = fe_to_mont(x);
        y = fe_to_mont(y);
        memcpy(gen.x.l, x.l, sizeof(x.l));
        memcpy(gen.y.l, y.l, sizeof(y.l));
        gen.z = fe_one<FqCfg>();
        left.push_back(make_term(gen, fe_neg(v_total)));
    }
    const P1 L = joint_scalar_mul(left), R = joint_scalar_mul(right);
    F12 f = f12_one();
    A1 a;
    if (p1_to_affine(L, a)) miller_accumulate(f, a.x, a.y, g2->h);
    if (p1_to_affine(R, a) && !g2->tau_h.infinity) miller_accumulate(f, a.x, fe_neg(a.y), g2->tau_h);
    *ok = f12_is_one(final_exponentiation(f)) ? 1 : 0;
    return PLONK_HIP_OK;
}

}  // namespace verify
}  // namespace plonk

extern "C" {

int plonk_hip_kzg_setup_tau_g2(const uint64_t tau[4], uint8_t out[96]) {
    using namespace plonk;
    using namespace plonk::verify;
    if (!tau || !out) return PLONK_HIP_ERR_INVALID_ARG;
    const Fr k = fe_from_mont(load_fr(tau));
    G2Affine acc = {f2_zero(), f2_zero(), true};
    const G2Affine h = g2_generator();
    for (int i = FrCfg::N - 1; i >= 0; i--) {
        for (int b = 31; b >= 0; b--) {
            acc = g2_double(acc);
            if ((k.l[i] >> b) & 1) acc = g2_add(acc, h);
        }
    }
    memset(out, 0, 96);
    if (acc.infinity) {
        out[0] = 0xc0;
        return PLONK_HIP_OK;
    }
    const Fp c1 = fe_from_mont(acc.x.c1), c0 = fe_from_mont(acc.x.c0);
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j < 8; j++) {
            out[47 - 8 * i - j] = (uint8_t)(c1.l[i] >> (8 * j));
            out[95 - 8 * i - j] = (uint8_t)(c0.l[i] >> (8 * j));
        }
    }
    out[0] |= 0x80 | (f2_lexicographically_largest(acc.y) ? 0x20 : 0);
    return PLONK_HIP_OK;
}

int plonk_hip_kzg_verify(const uint8_t tau_g2[96], const uint64_t com[18], const uint64_t u[4], const uint64_t v[4],
                         const uint64_t proof[18], int *ok) {
    if (!tau_g2 || !com || !u || !v || !proof || !ok) return PLONK_HIP_ERR_INVALID_ARG;
    return plonk::verify::verify_folded(tau_g2, 1, com, u, v, proof, nullptr, ok);
}

int plonk_hip_kzg_verify_batch(const uint8_t tau_g2[96], size_t n, const uint64_t *coms, const uint64_t *us,
                               const uint64_t *vs, const uint64_t *proofs, const uint64_t *rho, int *ok) {
    if (!tau_g2 || !ok || (n && (!coms || !us || !vs || !proofs || !rho))) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) {
        *ok = 1;
        return PLONK_HIP_OK;
    }
    return plonk::verify::verify_folded(tau_g2, n, coms, us, vs, proofs, rho, ok);
}

}  // extern "C"
