// G1 bucket accumulation on the 28-bit limb form (field28.cuh): the group law of the MSM's hot
// loop.  Same formulas and conventions as g1.cuh (XYZZ accumulators, affine identity = all
// zero); only the field representation differs.  Window tables are stored in this form, one
// 128-byte line per affine point.
#pragma once
#include "field28.cuh"
#include "g1.cuh"

namespace plonk {

struct G1Affine28 {
    Fq28 x, y;  // 2 x 64 B
};
struct G1Xyzz28 {
    Fq28 x, y, zz, zzz;
};

PLONK_HD bool affine28_is_identity(const G1Affine28 &p) { return q28_is_zero(p.x) && q28_is_zero(p.y); }

PLONK_HD G1Xyzz28 xyzz28_identity() {
    G1Xyzz28 r;
    r.x = q28_zero();
    r.y = q28_zero();
    r.zz = q28_zero();
    r.zzz = q28_zero();
    return r;
}
PLONK_HD bool xyzz28_is_identity(const G1Xyzz28 &p) { return q28_is_zero(p.zz); }

// 2P for an affine P != identity (mdbl-2008-s-1).
PLONK_HD G1Xyzz28 xyzz28_dbl_affine(const Fq28 &x, const Fq28 &y) {
    Fq28 U = q28_dbl(y);
    Fq28 V = q28_mul(U, U);
    Fq28 W = q28_mul(U, V);
    Fq28 S = q28_mul(x, V);
    Fq28 XX = q28_mul(x, x);
    Fq28 M = q28_add(q28_dbl(XX), XX);
    G1Xyzz28 r;
    r.x = q28_sub(q28_sub(q28_mul(M, M), S), S);
    r.y = q28_sub(q28_mul(M, q28_sub(S, r.x)), q28_mul(W, y));
    r.zz = V;
    r.zzz = W;
    return r;
}

// Keeps the instruction scheduler from interleaving two steps of the addition: every step is a
// 392-multiply-add field product, so nothing is gained by overlapping them, while overlapping
// lengthens live ranges and with them the register count.
#if defined(__HIP_DEVICE_COMPILE__)
#define PLONK_STEP() __builtin_amdgcn_sched_barrier(0)
#else
#define PLONK_STEP() ((void)0)
#endif

// acc += (x2, y2) for an affine point that is not the identity (madd-2008-s: 8M + 2S), except
// when (x2, y2) equals acc: then acc is left alone and false is returned -- the caller doubles.
// The steps are ordered so that at most eight field elements are live at any time.
PLONK_HD bool xyzz28_madd_nodbl(G1Xyzz28 &acc, const Fq28 &x2, const Fq28 &y2) {
    if (xyzz28_is_identity(acc)) {
        acc.x = x2;
        acc.y = y2;
        acc.zz = q28_one();
        acc.zzz = q28_one();
        return true;
    }
    Fq28 P = q28_sub(q28_mul(x2, acc.zz), acc.x);  // U2 - X1
    PLONK_STEP();
    Fq28 R = q28_sub(q28_mul(y2, acc.zzz), acc.y);  // S2 - Y1
    PLONK_STEP();
    if (q28_is_zero(P)) {
        if (q28_is_zero(R)) return false;
        acc = xyzz28_identity();
        return true;
    }
    Fq28 PP = q28_mul(P, P);
    PLONK_STEP();
    acc.zz = q28_mul(acc.zz, PP);
    PLONK_STEP();
    Fq28 PPP = q28_mul(P, PP);
    PLONK_STEP();
    acc.zzz = q28_mul(acc.zzz, PPP);
    PLONK_STEP();
    Fq28 Q = q28_mul(acc.x, PP);
    PLONK_STEP();
    Fq28 T = q28_mul(acc.y, PPP);
    PLONK_STEP();
    acc.x = q28_sub(q28_sub(q28_sub(q28_mul(R, R), PPP), Q), Q);
    PLONK_STEP();
    acc.y = q28_sub(q28_mul(R, q28_sub(Q, acc.x)), T);
    return true;
}

// 2P, general ZZ (dbl-2008-s-1, a = 0).
PLONK_HD G1Xyzz28 xyzz28_dbl(const G1Xyzz28 &p) {
    if (xyzz28_is_identity(p)) return xyzz28_identity();
    Fq28 U = q28_dbl(p.y);
    Fq28 V = q28_mul(U, U);
    Fq28 W = q28_mul(U, V);
    Fq28 S = q28_mul(p.x, V);
    Fq28 XX = q28_mul(p.x, p.x);
    Fq28 M = q28_add(q28_dbl(XX), XX);
    G1Xyzz28 r;
    r.x = q28_sub(q28_sub(q28_mul(M, M), S), S);
    r.y = q28_sub(q28_mul(M, q28_sub(S, r.x)), q28_mul(W, p.y));
    r.zz = q28_mul(V, p.zz);
    r.zzz = q28_mul(W, p.zzz);
    return r;
}

// out = a + b, both XYZZ (add-2008-s: 12M + 2S), except when a == b (both non-identity):
// returns false and leaves out untouched -- the caller doubles.
PLONK_HD bool xyzz28_add_nodbl(G1Xyzz28 &out, const G1Xyzz28 &a, const G1Xyzz28 &b) {
    if (xyzz28_is_identity(a)) {
        out = b;
        return true;
    }
    if (xyzz28_is_identity(b)) {
        out = a;
        return true;
    }
    Fq28 U1 = q28_mul(a.x, b.zz);
    Fq28 P = q28_sub(q28_mul(b.x, a.zz), U1);
    Fq28 S1 = q28_mul(a.y, b.zzz);
    Fq28 R = q28_sub(q28_mul(b.y, a.zzz), S1);
    if (q28_is_zero(P)) {
        if (q28_is_zero(R)) return false;
        out = xyzz28_identity();
        return true;
    }
    Fq28 PP = q28_mul(P, P);
    Fq28 PPP = q28_mul(P, PP);
    Fq28 Q = q28_mul(U1, PP);
    G1Xyzz28 r;
    r.x = q28_sub(q28_sub(q28_sub(q28_mul(R, R), PPP), Q), Q);
    r.y = q28_sub(q28_mul(R, q28_sub(Q, r.x)), q28_mul(S1, PPP));
    r.zz = q28_mul(q28_mul(a.zz, b.zz), PP);
    r.zzz = q28_mul(q28_mul(a.zzz, b.zzz), PPP);
    out = r;
    return true;
}

PLONK_HD G1Xyzz28 xyzz28_add(const G1Xyzz28 &a, const G1Xyzz28 &b) {
    G1Xyzz28 r;
    if (!xyzz28_add_nodbl(r, a, b)) r = xyzz28_dbl(a);
    return r;
}

// Lazy mixed addition for the accumulation loop: acc += (x2, negate ? -y2 : y2).
// Preconditions: acc is not the identity; (x2, y2) canonical and not the identity;
//   acc.x < 5.1 q, acc.y < 3.1 q, acc.zz, acc.zzz < 1.1 q   (the same bounds hold on return).
// Returns false, with acc untouched, when x2*ZZ = X mod q (the points are equal or opposite):
// the caller resolves that rare case on canonical values.
PLONK_HD bool xyzz28_madd_lazy(G1Xyzz28 &acc, const Fq28 &x2, const Fq28 &y2, bool negate) {
    Fq28 P = q28_sub_lazy<6>(q28_mul_lazy(x2, acc.zz), acc.x);            // U2 - X1      < 7.1 q
    PLONK_STEP();
    Fq28 R = q28_sub_lazy<5>(q28_mul_lazy(y2, acc.zzz), acc.y, negate);   // +-S2 - Y1    < 6.1 q
    PLONK_STEP();
    Fq28 PP = q28_sqr_lazy(P);                                            //              < 1.03 q
    if (q28_is_zero_mod_q(PP)) return false;
    PLONK_STEP();
    acc.zz = q28_mul_lazy(acc.zz, PP);
    PLONK_STEP();
    Fq28 PPP = q28_mul_lazy(P, PP);
    PLONK_STEP();
    acc.zzz = q28_mul_lazy(acc.zzz, PPP);
    PLONK_STEP();
    Fq28 Q = q28_mul_lazy(acc.x, PP);
    PLONK_STEP();
    acc.x = q28_sub2_lazy<4>(q28_sqr_lazy(R), PPP, Q);                 // R^2 - PPP - 2Q  < 5.1 q
    PLONK_STEP();
    // Y3 = R (Q - X3) - Y1 PPP as one double-width sum with a single reduction
    Fq28 negY = q28_sub_lazy<4>(q28_zero(), acc.y);                       // 4q - Y1      <= 4 q
    acc.y = q28_mul2_lazy(R, q28_sub_lazy<6>(Q, acc.x), negY, PPP);       //              < 1.1 q
    return true;
}

// Lazy full addition for the tree sums behind the accumulation: out = a + b for two points in the
// lazy form above (neither the identity; x < 5.1 q, y < 3.1 q, zz, zzz < 1.1 q; the result obeys the
// same bounds, so sums of sums need no canonicalisation in between).  add-2008-s with 13
// reductions for its 14 products.  Returns false, with out untouched, when the points are equal or
// opposite: the caller resolves that on canonical values.
PLONK_HD bool xyzz28_add_lazy(G1Xyzz28 &out, const G1Xyzz28 &a, const G1Xyzz28 &b) {
    Fq28 U1 = q28_mul_lazy(a.x, b.zz);                                    // < 1.01 q
    Fq28 P = q28_sub_lazy<2>(q28_mul_lazy(b.x, a.zz), U1);                // U2 - U1     < 3.1 q
    Fq28 S1 = q28_mul_lazy(a.y, b.zzz);
    Fq28 R = q28_sub_lazy<2>(q28_mul_lazy(b.y, a.zzz), S1);               // S2 - S1     < 3.1 q
    Fq28 PP = q28_sqr_lazy(P);                                            //             < 1.01 q
    if (q28_is_zero_mod_q(PP)) return false;
    Fq28 PPP = q28_mul_lazy(P, PP);
    Fq28 Q = q28_mul_lazy(U1, PP);
    G1Xyzz28 r;
    r.x = q28_sub2_lazy<4>(q28_sqr_lazy(R), PPP, Q);                   // R^2 - PPP - 2Q  < 5.1 q
    Fq28 negS1 = q28_sub_lazy<2>(q28_zero(), S1);                         // 2q - S1     <= 2 q
    r.y = q28_mul2_lazy(R, q28_sub_lazy<6>(Q, r.x), negS1, PPP);          // R (Q - X3) - S1 PPP  < 1.1 q
    r.zz = q28_mul_lazy(q28_mul_lazy(a.zz, b.zz), PP);
    r.zzz = q28_mul_lazy(q28_mul_lazy(a.zzz, b.zzz), PPP);
    out = r;
    return true;
}

// Canonical representative of a lazily accumulated point.
PLONK_HD G1Xyzz28 xyzz28_canon(const G1Xyzz28 &p) {
    if (xyzz28_is_identity(p)) return xyzz28_identity();
    G1Xyzz28 r;
    r.x = q28_canon(p.x);
    r.y = q28_canon(p.y);
    r.zz = q28_canon(p.zz);
    r.zzz = q28_canon(p.zzz);
    return r;
}

PLONK_HD void xyzz28_madd(G1Xyzz28 &acc, const G1Affine28 &p, bool negate) {
    if (affine28_is_identity(p)) return;
    Fq28 y = negate ? q28_neg(p.y) : p.y;
    if (!xyzz28_madd_nodbl(acc, p.x, y)) acc = xyzz28_dbl_affine(p.x, y);
}

// representation changes (the rest of the pipeline works on the 12 x 32-bit form)
PLONK_HD G1Affine28 affine28_from(const G1Affine &p) {
    G1Affine28 r;
    r.x = q28_from_fq(p.x);  // 0 -> 0, so the all-zero identity code is preserved
    r.y = q28_from_fq(p.y);
    return r;
}
PLONK_HD G1Affine affine28_to(const G1Affine28 &p) {
    G1Affine r;
    r.x = q28_to_fq(p.x);
    r.y = q28_to_fq(p.y);
    return r;
}
PLONK_HD G1Xyzz xyzz28_to(const G1Xyzz28 &p) {
    G1Xyzz r;
    r.x = q28_to_fq(p.x);
    r.y = q28_to_fq(p.y);
    r.zz = q28_to_fq(p.zz);
    r.zzz = q28_to_fq(p.zzz);
    return r;
}

}  // namespace plonk
