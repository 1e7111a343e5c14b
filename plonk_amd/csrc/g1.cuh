// BLS12-381 G1 (y^2 = x^3 + 4 over Fq) group law for the bucket method.
//
// Three representations:
//   G1Affine   (x, y)            96 B   resident SRS tables; identity encoded as x = y = 0
//                                        ((0,0) is not on the curve, so the code is free)
//   G1Xyzz     (X, Y, ZZ, ZZZ)   192 B  x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity ZZ = 0.
//                                        Bucket accumulators: mixed add = 8M + 2S with no
//                                        inversion and no Z tracking.
//   G1Jac      (X, Y, Z)         144 B  ark-ec's G1Projective layout (Jacobian, identity Z = 0):
//                                        what the C ABI takes and returns.
// All formulas are complete by explicit branches (P = Q doubles, P = -Q gives the identity).
#pragma once
#include "field.cuh"

namespace plonk {

struct G1Affine {
    Fq x, y;
};
struct G1Xyzz {
    Fq x, y, zz, zzz;
};
struct G1Jac {
    Fq x, y, z;
};

PLONK_HD bool affine_is_identity(const G1Affine &p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
PLONK_HD G1Affine affine_identity() {
    G1Affine r;
    r.x = fe_zero<FqCfg>();
    r.y = fe_zero<FqCfg>();
    return r;
}

PLONK_HD G1Xyzz xyzz_identity() {
    G1Xyzz r;
    r.x = fe_zero<FqCfg>();
    r.y = fe_zero<FqCfg>();
    r.zz = fe_zero<FqCfg>();
    r.zzz = fe_zero<FqCfg>();
    return r;
}
PLONK_HD bool xyzz_is_identity(const G1Xyzz &p) { return fe_is_zero(p.zz); }

PLONK_HD G1Xyzz xyzz_from_affine(const G1Affine &p) {
    if (affine_is_identity(p)) return xyzz_identity();
    G1Xyzz r;
    r.x = p.x;
    r.y = p.y;
    r.zz = fe_one<FqCfg>();
    r.zzz = fe_one<FqCfg>();
    return r;
}

// 2P, general ZZ (dbl-2008-s-1, a = 0).
PLONK_HD G1Xyzz xyzz_dbl(const G1Xyzz &p) {
    if (xyzz_is_identity(p)) return xyzz_identity();
    Fq U = fe_dbl(p.y);
    Fq V = fe_sqr(U);
    Fq W = fe_mul(U, V);
    Fq S = fe_mul(p.x, V);
    Fq XX = fe_sqr(p.x);
    Fq M = fe_add(fe_dbl(XX), XX);
    G1Xyzz r;
    r.x = fe_sub(fe_sub(fe_sqr(M), S), S);
    r.y = fe_sub(fe_mul(M, fe_sub(S, r.x)), fe_mul(W, p.y));
    r.zz = fe_mul(V, p.zz);
    r.zzz = fe_mul(W, p.zzz);
    return r;
}

// 2P for an affine P != identity (mdbl-2008-s-1).
PLONK_HD G1Xyzz xyzz_dbl_affine(const Fq &x, const Fq &y) {
    Fq U = fe_dbl(y);
    Fq V = fe_sqr(U);
    Fq W = fe_mul(U, V);
    Fq S = fe_mul(x, V);
    Fq XX = fe_sqr(x);
    Fq M = fe_add(fe_dbl(XX), XX);
    G1Xyzz r;
    r.x = fe_sub(fe_sub(fe_sqr(M), S), S);
    r.y = fe_sub(fe_mul(M, fe_sub(S, r.x)), fe_mul(W, y));
    r.zz = V;
    r.zzz = W;
    return r;
}

// acc += (x2, y2) for an affine point that is not the identity (madd-2008-s), except when
// (x2, y2) equals acc: then nothing is changed and false is returned -- the caller doubles.
// Keeping the doubling out of line keeps the hot loop's code and register footprint small.
PLONK_HD bool xyzz_madd_nodbl(G1Xyzz &acc, const Fq &x2, const Fq &y2) {
    if (xyzz_is_identity(acc)) {
        acc.x = x2;
        acc.y = y2;
        acc.zz = fe_one<FqCfg>();
        acc.zzz = fe_one<FqCfg>();
        return true;
    }
    Fq U2 = fe_mul(x2, acc.zz);
    Fq S2 = fe_mul(y2, acc.zzz);
    Fq P = fe_sub(U2, acc.x);
    Fq R = fe_sub(S2, acc.y);
    if (fe_is_zero(P)) {
        if (fe_is_zero(R)) return false;
        acc = xyzz_identity();
        return true;
    }
    Fq PP = fe_sqr(P);
    Fq PPP = fe_mul(P, PP);
    Fq Q = fe_mul(acc.x, PP);
    Fq X3 = fe_sub(fe_sub(fe_sub(fe_sqr(R), PPP), Q), Q);
    Fq Y3 = fe_sub(fe_mul(R, fe_sub(Q, X3)), fe_mul(acc.y, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fe_mul(acc.zz, PP);
    acc.zzz = fe_mul(acc.zzz, PPP);
    return true;
}

PLONK_HD void xyzz_madd_xy(G1Xyzz &acc, const Fq &x2, const Fq &y2) {
    if (!xyzz_madd_nodbl(acc, x2, y2)) acc = xyzz_dbl_affine(x2, y2);
}

// acc += (negate ? -p : p); p may be the identity.
PLONK_HD void xyzz_madd(G1Xyzz &acc, const G1Affine &p, bool negate) {
    if (affine_is_identity(p)) return;
    Fq y = negate ? fe_neg(p.y) : p.y;
    xyzz_madd_xy(acc, p.x, y);
}

// out = a + b, both XYZZ (add-2008-s), except when a == b (both non-identity): returns false
// and leaves out untouched -- the caller doubles.
PLONK_HD bool xyzz_add_nodbl(G1Xyzz &out, const G1Xyzz &a, const G1Xyzz &b) {
    if (xyzz_is_identity(a)) {
        out = b;
        return true;
    }
    if (xyzz_is_identity(b)) {
        out = a;
        return true;
    }
    Fq U1 = fe_mul(a.x, b.zz);
    Fq U2 = fe_mul(b.x, a.zz);
    Fq S1 = fe_mul(a.y, b.zzz);
    Fq S2 = fe_mul(b.y, a.zzz);
    Fq P = fe_sub(U2, U1);
    Fq R = fe_sub(S2, S1);
    if (fe_is_zero(P)) {
        if (fe_is_zero(R)) return false;
        out = xyzz_identity();
        return true;
    }
    Fq PP = fe_sqr(P);
    Fq PPP = fe_mul(P, PP);
    Fq Q = fe_mul(U1, PP);
    G1Xyzz r;
    r.x = fe_sub(fe_sub(fe_sub(fe_sqr(R), PPP), Q), Q);
    r.y = fe_sub(fe_mul(R, fe_sub(Q, r.x)), fe_mul(S1, PPP));
    r.zz = fe_mul(fe_mul(a.zz, b.zz), PP);
    r.zzz = fe_mul(fe_mul(a.zzz, b.zzz), PPP);
    out = r;
    return true;
}

PLONK_HD G1Xyzz xyzz_add(const G1Xyzz &a, const G1Xyzz &b) {
    G1Xyzz r;
    if (!xyzz_add_nodbl(r, a, b)) r = xyzz_dbl(a);
    return r;
}

PLONK_HD G1Jac jac_identity() {
    // ark's G1Projective::zero() = (1, 1, 0)
    G1Jac r;
    r.x = fe_one<FqCfg>();
    r.y = fe_one<FqCfg>();
    r.z = fe_zero<FqCfg>();
    return r;
}

// XYZZ -> Jacobian without inversion: Z = ZZ gives X = x*ZZ^2 = X*ZZ and, because
// ZZ^3 = ZZZ^2, Y = y*ZZ^3 = Y*ZZZ.
PLONK_HD G1Jac xyzz_to_jac(const G1Xyzz &p) {
    if (xyzz_is_identity(p)) return jac_identity();
    G1Jac r;
    r.x = fe_mul(p.x, p.zz);
    r.y = fe_mul(p.y, p.zzz);
    r.z = p.zz;
    return r;
}

// Jacobian -> XYZZ: ZZ = Z^2, ZZZ = Z^3.
PLONK_HD G1Xyzz jac_to_xyzz(const G1Jac &p) {
    if (fe_is_zero(p.z)) return xyzz_identity();
    G1Xyzz r;
    r.x = p.x;
    r.y = p.y;
    r.zz = fe_sqr(p.z);
    r.zzz = fe_mul(r.zz, p.z);
    return r;
}

// 2P in Jacobian coordinates (dbl-2009-l, a = 0): 2M + 5S.
PLONK_HD G1Jac jac_dbl(const G1Jac &p) {
    if (fe_is_zero(p.z)) return jac_identity();
    Fq A = fe_sqr(p.x);
    Fq B = fe_sqr(p.y);
    Fq Cc = fe_sqr(B);
    Fq t = fe_sqr(fe_add(p.x, B));
    Fq D = fe_dbl(fe_sub(fe_sub(t, A), Cc));
    Fq E = fe_add(fe_dbl(A), A);
    Fq F = fe_sqr(E);
    G1Jac r;
    r.x = fe_sub(F, fe_dbl(D));
    Fq C8 = fe_dbl(fe_dbl(fe_dbl(Cc)));
    r.y = fe_sub(fe_mul(E, fe_sub(D, r.x)), C8);
    r.z = fe_dbl(fe_mul(p.y, p.z));
    return r;
}

// Jacobian -> affine given zinv = 1/Z.
PLONK_HD G1Affine jac_to_affine_with_inv(const G1Jac &p, const Fq &zinv) {
    Fq zi2 = fe_sqr(zinv);
    G1Affine r;
    r.x = fe_mul(p.x, zi2);
    r.y = fe_mul(p.y, fe_mul(zi2, zinv));
    return r;
}

}  // namespace plonk
