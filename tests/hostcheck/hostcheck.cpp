// Host build of the device arithmetic headers (test infrastructure): the same
// __host__ __device__ code that the gfx950 kernels inline is compiled with g++ and
// exposed over a C ABI so the CPU test-suite can check it against the oracle.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../plonk_amd/csrc/g1.cuh"
#include "../../plonk_amd/csrc/field29.cuh"
#include "../../plonk_amd/csrc/g1_28.cuh"
#include "../../plonk_amd/csrc/coop28.cuh"
#include "../../plonk_amd/csrc/msm_core.cuh"

using namespace plonk;

template <class C>
static Fe<C> ld(const uint32_t *p) {
    Fe<C> r;
    memcpy(r.l, p, sizeof(r.l));
    return r;
}
template <class C>
static void st(uint32_t *p, const Fe<C> &v) {
    memcpy(p, v.l, sizeof(v.l));
}

#define VEC2(name, C, fn)                                                               \
    extern "C" void name(const uint32_t *a, const uint32_t *b, uint32_t *out, size_t n) { \
        for (size_t i = 0; i < n; i++)                                                  \
            st<C>(out + C::N * i, fn(ld<C>(a + C::N * i), ld<C>(b + C::N * i)));       \
    }
#define VEC1(name, C, fn)                                                \
    extern "C" void name(const uint32_t *a, uint32_t *out, size_t n) {   \
        for (size_t i = 0; i < n; i++) st<C>(out + C::N * i, fn(ld<C>(a + C::N * i))); \
    }

VEC2(hc_fq_mul, FqCfg, fe_mul<FqCfg>)
VEC2(hc_fq_add, FqCfg, fe_add<FqCfg>)
VEC2(hc_fq_sub, FqCfg, fe_sub<FqCfg>)
VEC1(hc_fq_inv, FqCfg, fe_inv<FqCfg>)
VEC1(hc_fq_dbl, FqCfg, fe_dbl<FqCfg>)
VEC1(hc_fq_neg, FqCfg, fe_neg<FqCfg>)
VEC1(hc_fq_from_mont, FqCfg, fe_from_mont<FqCfg>)
VEC1(hc_fq_to_mont, FqCfg, fe_to_mont<FqCfg>)
VEC2(hc_fr_mul, FrCfg, fe_mul<FrCfg>)
VEC2(hc_fr_add, FrCfg, fe_add<FrCfg>)
VEC2(hc_fr_sub, FrCfg, fe_sub<FrCfg>)
VEC1(hc_fr_inv, FrCfg, fe_inv<FrCfg>)
VEC1(hc_fr_dbl, FrCfg, fe_dbl<FrCfg>)
VEC1(hc_fr_from_mont, FrCfg, fe_from_mont<FrCfg>)
VEC1(hc_fr_to_mont, FrCfg, fe_to_mont<FrCfg>)

static G1Xyzz ldx(const uint32_t *p) {
    G1Xyzz r;
    memcpy(&r, p, sizeof(r));
    return r;
}
static G1Affine lda(const uint32_t *p) {
    G1Affine r;
    memcpy(&r, p, sizeof(r));
    return r;
}
static G1Jac ldj(const uint32_t *p) {
    G1Jac r;
    memcpy(&r, p, sizeof(r));
    return r;
}

extern "C" void hc_xyzz_madd(const uint32_t *acc, const uint32_t *aff, int negate, uint32_t *out) {
    G1Xyzz a = ldx(acc);
    xyzz_madd(a, lda(aff), negate != 0);
    memcpy(out, &a, sizeof(a));
}
extern "C" void hc_xyzz_add(const uint32_t *a, const uint32_t *b, uint32_t *out) {
    G1Xyzz r = xyzz_add(ldx(a), ldx(b));
    memcpy(out, &r, sizeof(r));
}
extern "C" void hc_xyzz_dbl(const uint32_t *a, uint32_t *out) {
    G1Xyzz r = xyzz_dbl(ldx(a));
    memcpy(out, &r, sizeof(r));
}
extern "C" void hc_xyzz_to_jac(const uint32_t *a, uint32_t *out) {
    G1Jac r = xyzz_to_jac(ldx(a));
    memcpy(out, &r, sizeof(r));
}
extern "C" void hc_jac_to_xyzz(const uint32_t *a, uint32_t *out) {
    G1Xyzz r = jac_to_xyzz(ldj(a));
    memcpy(out, &r, sizeof(r));
}
extern "C" void hc_jac_dbl(const uint32_t *a, uint32_t *out) {
    G1Jac r = jac_dbl(ldj(a));
    memcpy(out, &r, sizeof(r));
}
extern "C" void hc_xyzz_from_affine(const uint32_t *a, uint32_t *out) {
    G1Xyzz r = xyzz_from_affine(lda(a));
    memcpy(out, &r, sizeof(r));
}

// Signed window digits of a Montgomery-form scalar (see msm_core.cuh).
// Returns 1 when the digits are those of r - k (widths that divide 255: the MSM then takes the opposite point),
// and the carry out of the top window in *carry_out (always 0).
extern "C" int hc_scalar_digits(const uint32_t *scalar_mont, int c, int nwin, int32_t *out, uint32_t *carry_out) {
    Fr k = fe_from_mont(ld<FrCfg>(scalar_mont));
    const bool opposite = msm_folds_scalars(c) && scalar_fold(k);
    uint32_t carry = 0;
    for (int w = 0; w < nwin; w++) out[w] = signed_digit(k, w, c, carry);
    *carry_out = carry;
    return opposite ? 1 : 0;
}
extern "C" int hc_num_windows(int c) { return msm_num_windows(c); }

// 28-bit limb form: operands and results cross this boundary in the 12 x 32-bit Montgomery form
// (converted in and out with the header's own converters).
#define VEC2_28(name, fn)                                                                 \
    extern "C" void name(const uint32_t *a, const uint32_t *b, uint32_t *out, size_t n) { \
        for (size_t i = 0; i < n; i++)                                                    \
            st<FqCfg>(out + 12 * i, q28_to_fq(fn(q28_from_fq(ld<FqCfg>(a + 12 * i)),      \
                                                 q28_from_fq(ld<FqCfg>(b + 12 * i)))));   \
    }
VEC2_28(hc_q28_mul, q28_mul)
VEC2_28(hc_q28_add, q28_add)
VEC2_28(hc_q28_sub, q28_sub)
extern "C" void hc_q28_sqr(const uint32_t *a, uint32_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) st<FqCfg>(out + 12 * i, q28_to_fq(q28_sqr(q28_from_fq(ld<FqCfg>(a + 12 * i)))));
}
extern "C" void hc_q28_roundtrip(const uint32_t *a, uint32_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) st<FqCfg>(out + 12 * i, q28_to_fq(q28_from_fq(ld<FqCfg>(a + 12 * i))));
}
// raw limbs of the converted value (14 words), to check canonical form
extern "C" void hc_q28_limbs(const uint32_t *a, uint32_t *out14, size_t n) {
    for (size_t i = 0; i < n; i++) {
        Fq28 v = q28_from_fq(ld<FqCfg>(a + 12 * i));
        memcpy(out14 + 14 * i, v.l, 56);
    }
}

// Mixed addition on the 28-bit limb form.  acc_in/out: XYZZ in the 12 x 32-bit form (the
// accumulator is re-created in 28-bit form from affine points by the caller's sequence);
// here: fold a list of n affine points (with signs) into one accumulator and convert back.
extern "C" void hc_xyzz28_fold(const uint32_t *aff, const int *negate, size_t n, uint32_t *out_xyzz) {
    G1Xyzz28 acc = xyzz28_identity();
    for (size_t i = 0; i < n; i++) {
        G1Affine28 p = affine28_from(lda(aff + 24 * i));
        xyzz28_madd(acc, p, negate[i] != 0);
    }
    G1Xyzz r = xyzz28_to(acc);
    memcpy(out_xyzz, &r, sizeof(r));
}
extern "C" void hc_affine28_roundtrip(const uint32_t *aff, uint32_t *out) {
    G1Affine r = affine28_to(affine28_from(lda(aff)));
    memcpy(out, &r, sizeof(r));
}

// The accumulation loop's lazy form: same fold, lazy additions, canonicalised at the end
// (mirrors k_accumulate + the canonicalisation of k_fixup).
extern "C" void hc_xyzz28_fold_lazy(const uint32_t *aff, const int *negate, size_t n, uint32_t *out_xyzz) {
    G1Xyzz28 acc = xyzz28_identity();
    for (size_t i = 0; i < n; i++) {
        G1Affine28 p = affine28_from(lda(aff + 24 * i));
        if (affine28_is_identity(p)) continue;
        bool neg = negate[i] != 0;
        if (xyzz28_is_identity(acc)) {
            acc.x = p.x;
            acc.y = neg ? q28_neg(p.y) : p.y;
            acc.zz = q28_one();
            acc.zzz = q28_one();
        } else if (!xyzz28_madd_lazy(acc, p.x, p.y, neg)) {
            acc = xyzz28_canon(acc);
            xyzz28_madd(acc, p, neg);
        }
    }
    G1Xyzz r = xyzz28_to(xyzz28_canon(acc));
    memcpy(out_xyzz, &r, sizeof(r));
}

// The tree sums behind the accumulation: every point enters as a lazily accumulated pair
// (p + p' by the lazy mixed addition), the partial sums are joined by lazy full additions in a
// binary tree, and only the end result is canonicalised (mirrors k_fixup / k_row_col_sums).
static G1Xyzz28 lazy_sum(G1Xyzz28 a, G1Xyzz28 b) {
    if (xyzz28_is_identity(a)) return b;
    if (xyzz28_is_identity(b)) return a;
    G1Xyzz28 r;
    if (!xyzz28_add_lazy(r, a, b)) r = xyzz28_add(xyzz28_canon(a), xyzz28_canon(b));
    return r;
}
extern "C" void hc_xyzz28_tree_lazy(const uint32_t *aff, const int *negate, size_t n, uint32_t *out_xyzz) {
    std::vector<G1Xyzz28> level;
    for (size_t i = 0; i < n; i++) {
        G1Affine28 p = affine28_from(lda(aff + 24 * i));
        G1Xyzz28 leaf = xyzz28_identity();
        if (!affine28_is_identity(p)) {
            leaf.x = p.x;
            leaf.y = negate[i] ? q28_neg(p.y) : p.y;
            leaf.zz = q28_one();
            leaf.zzz = q28_one();
        }
        level.push_back(leaf);
    }
    while (level.size() > 1) {
        std::vector<G1Xyzz28> next;
        for (size_t i = 0; i + 1 < level.size(); i += 2) next.push_back(lazy_sum(level[i], level[i + 1]));
        if (level.size() & 1) next.push_back(level.back());
        level.swap(next);
    }
    G1Xyzz r = xyzz28_to(xyzz28_canon(level.empty() ? xyzz28_identity() : level[0]));
    memcpy(out_xyzz, &r, sizeof(r));
}

// The staged (four-wave) addition and doubling, run on one slot with plain arrays: the stages in
// order, the four roles one after the other within a stage (what the barriers guarantee).
struct OneSlot {
    Fq28 A[4], B[4], T[kCoopTemps];
    bool flag = false;
    Fq28 a(int i) const { return A[i]; }
    Fq28 b(int i) const { return B[i]; }
    Fq28 t(int i) const { return T[i]; }
    void set_a(int i, const Fq28 &x) { A[i] = x; }
    void set_t(int i, const Fq28 &x) { T[i] = x; }
    bool slow() const { return flag; }
    void set_slow(bool f) { flag = f; }
};
static void slot_put(Fq28 *v, const G1Xyzz28 &p) { v[0] = p.x; v[1] = p.y; v[2] = p.zz; v[3] = p.zzz; }
static G1Xyzz28 slot_get(const Fq28 *v) { G1Xyzz28 p; p.x = v[0]; p.y = v[1]; p.zz = v[2]; p.zzz = v[3]; return p; }
static G1Xyzz28 coop_sum(const G1Xyzz28 &a, const G1Xyzz28 &b) {
    if (xyzz28_is_identity(a)) return b;
    if (xyzz28_is_identity(b)) return a;
    OneSlot m;
    slot_put(m.A, a);
    slot_put(m.B, b);
    for (int stage = 0; stage < kCoopAddStages; stage++)
        for (int w = 0; w < kCoopWaves; w++) coop_add_stage(m, stage, w);
    if (m.slow()) return xyzz28_add(xyzz28_canon(a), xyzz28_canon(b));
    return slot_get(m.A);
}
static G1Xyzz28 coop_double(const G1Xyzz28 &a) {
    if (xyzz28_is_identity(a)) return a;
    OneSlot m;
    slot_put(m.A, a);
    for (int stage = 0; stage < kCoopDblStages; stage++)
        for (int w = 0; w < kCoopWaves; w++) coop_dbl_stage(m, stage, w);
    return slot_get(m.A);
}
// sum of the points by a pairwise tree of staged additions, then `doublings` staged doublings
extern "C" void hc_xyzz28_tree_coop(const uint32_t *aff, const int *negate, size_t n, int doublings, uint32_t *out_xyzz) {
    std::vector<G1Xyzz28> level;
    for (size_t i = 0; i < n; i++) {
        G1Affine28 p = affine28_from(lda(aff + 24 * i));
        G1Xyzz28 leaf = xyzz28_identity();
        if (!affine28_is_identity(p)) {
            leaf.x = p.x;
            leaf.y = negate[i] ? q28_neg(p.y) : p.y;
            leaf.zz = q28_one();
            leaf.zzz = q28_one();
        }
        level.push_back(leaf);
    }
    while (level.size() > 1) {
        std::vector<G1Xyzz28> next;
        for (size_t i = 0; i + 1 < level.size(); i += 2) next.push_back(coop_sum(level[i], level[i + 1]));
        if (level.size() & 1) next.push_back(level.back());
        level.swap(next);
    }
    G1Xyzz28 acc = level.empty() ? xyzz28_identity() : level[0];
    for (int i = 0; i < doublings; i++) acc = coop_double(acc);
    G1Xyzz r = xyzz28_to(xyzz28_canon(acc));
    memcpy(out_xyzz, &r, sizeof(r));
}

// 29-bit limb form of Fr (NTT butterflies).  x, w: radix-2^256 Montgomery limbs (8 x u32).
extern "C" void hc_fr29_mul(const uint32_t *x, const uint32_t *w, uint32_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        Fr29 a = fr29_pack(ld<FrCfg>(x + 8 * i));
        Fr29 t = fr29_twiddle_from(ld<FrCfg>(w + 8 * i));
        st<FrCfg>(out + 8 * i, fr29_unpack(fr29_canon(fr29_mul_lazy(a, t))));
    }
}
// A lazy butterfly chain: acc <- (acc + x*w, acc - x*w + 2r) alternately, `steps` times, then
// canonicalised: checks the add / sub / reduce helpers against plain modular arithmetic.
extern "C" void hc_fr29_chain(const uint32_t *x, const uint32_t *w, int steps, uint32_t *out_sum, uint32_t *out_diff) {
    Fr29 a = fr29_pack(ld<FrCfg>(x)), b = fr29_pack(ld<FrCfg>(x + 8));
    Fr29 t = fr29_twiddle_from(ld<FrCfg>(w));
    for (int i = 0; i < steps; i++) {
        Fr29 m = fr29_mul_lazy(b, t);
        Fr29 s = fr29_add_lazy(a, m);
        Fr29 d = fr29_sub_lazy<2>(a, m);
        a = s;
        b = d;
        if ((i % 5) == 4) {
            a = fr29_reduce_below_2_256(a);
            b = fr29_reduce_below_2_256(b);
        }
    }
    st<FrCfg>(out_sum, fr29_unpack(fr29_canon(a)));
    st<FrCfg>(out_diff, fr29_unpack(fr29_canon(b)));
}
extern "C" void hc_fr29_roundtrip(const uint32_t *x, uint32_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) st<FrCfg>(out + 8 * i, fr29_unpack(fr29_pack(ld<FrCfg>(x + 8 * i))));
}
