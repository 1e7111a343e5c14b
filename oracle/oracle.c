/*
 * CPU oracle (TEST INFRASTRUCTURE ONLY) -- plain C restatement of the reference's KZG /
 * polynomial hot path on 64-bit limbs.
 *
 * It is the checker and the timed CPU baseline, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The shipped
 * library (plonk_amd/csrc) neither links nor calls it.
 *
 * Parity status: pinned.  tests/test_oracle_*.py check every routine here against
 * oracle/pyref.py, which regenerates the reference's plonk/data/verification_key.json and
 * plonk/data/proof.json byte for byte; orc_msm_naive itself reproduces com_S and com_W of
 * verification_key.json from srs.json + proving_key.json.
 *
 * The reference delegates all arithmetic to arkworks 0.5 (plonk/Cargo.toml:14-19), whose
 * sources are absent; the published algorithms are restated: Montgomery arithmetic with
 * R = 2^256 (Fr) and R = 2^384 (Fq), Jacobian G1 over y^2 = x^3 + 4, double-and-add
 * scalar multiplication, dense polynomials.  Paths in citations are relative to the
 * reference root.
 *
 * Memory formats (little-endian 64-bit limbs, Montgomery form, as ark-ff keeps them):
 *   Fr  = 4 limbs, Fq = 6 limbs, G1 projective = X|Y|Z (18 limbs, Jacobian, identity Z = 0),
 *   G1 affine = x|y (12 limbs, identity encoded as x = y = 0).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef unsigned __int128 u128;

#define FRN 4
#define FQN 6

static const u64 FR_MOD[FRN] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                                0x73eda753299d7d48ULL};
static const u64 FQ_MOD[FQN] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
/* G1 generator, canonical integers (pinned by plonk/data/srs.json:8). */
static const u64 G1X_CANON[FQN] = {0xfb3af00adb22c6bbULL, 0x6c55e83ff97a1aefULL, 0xa14e3a3f171bac58ULL,
                                   0xc3688c4f9774b905ULL, 0x2695638c4fa9ac0fULL, 0x17f1d3a73197d794ULL};
static const u64 G1Y_CANON[FQN] = {0x0caa232946c5e7e1ULL, 0xd03cc744a2888ae4ULL, 0x00db18cb2c04b3edULL,
                                   0xfcf5e095d5d00af6ULL, 0xa09e30ed741d8ae4ULL, 0x08b3f481e3aaa0f1ULL};

static u64 FR_INV, FQ_INV;        /* -p^-1 mod 2^64 */
static u64 FR_R1[FRN], FR_R2[FRN]; /* R mod p, R^2 mod p */
static u64 FQ_R1[FQN], FQ_R2[FQN];
static int g_init = 0;

/* ---------------------------------------------------------------- big integers */
static inline int bn_geq(const u64 *a, const u64 *b, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline u64 bn_add(u64 *r, const u64 *a, const u64 *b, int n) {
    u128 c = 0;
    for (int i = 0; i < n; i++) {
        c += (u128)a[i] + b[i];
        r[i] = (u64)c;
        c >>= 64;
    }
    return (u64)c;
}
static inline u64 bn_sub(u64 *r, const u64 *a, const u64 *b, int n) {
    u64 borrow = 0;
    for (int i = 0; i < n; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
}
static inline int bn_is_zero(const u64 *a, int n) {
    u64 x = 0;
    for (int i = 0; i < n; i++) x |= a[i];
    return x == 0;
}
static inline int bn_eq(const u64 *a, const u64 *b, int n) {
    u64 x = 0;
    for (int i = 0; i < n; i++) x |= a[i] ^ b[i];
    return x == 0;
}

/* ---------------------------------------------------------------- modular arithmetic */
static inline __attribute__((always_inline)) void mod_add(u64 *r, const u64 *a, const u64 *b, const u64 *p, int n) {
    u64 t[FQN];
    u64 c = bn_add(t, a, b, n);
    if (c || bn_geq(t, p, n)) bn_sub(t, t, p, n);
    memcpy(r, t, n * 8);
}
static inline __attribute__((always_inline)) void mod_sub(u64 *r, const u64 *a, const u64 *b, const u64 *p, int n) {
    u64 t[FQN];
    if (bn_sub(t, a, b, n)) bn_add(t, t, p, n);
    memcpy(r, t, n * 8);
}
static inline __attribute__((always_inline)) void mod_neg(u64 *r, const u64 *a, const u64 *p, int n) {
    if (bn_is_zero(a, n)) {
        memset(r, 0, n * 8);
    } else {
        bn_sub(r, p, a, n);
    }
}
/* Montgomery product a*b/R mod p (CIOS). */
static inline __attribute__((always_inline)) void mont_mul(u64 *r, const u64 *a, const u64 *b, const u64 *p, u64 inv,
                                                           int n) {
    u64 t[FQN + 2];
    memset(t, 0, sizeof(t));
    for (int i = 0; i < n; i++) {
        u128 c = 0;
        for (int j = 0; j < n; j++) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[n];
        t[n] = (u64)c;
        t[n + 1] = (u64)(c >> 64);
        u64 m = t[0] * inv;
        c = (u128)m * p[0] + t[0];
        c >>= 64;
        for (int j = 1; j < n; j++) {
            c += (u128)m * p[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[n];
        t[n - 1] = (u64)c;
        t[n] = t[n + 1] + (u64)(c >> 64);
    }
    if (t[n] || bn_geq(t, p, n)) bn_sub(t, t, p, n);
    memcpy(r, t, n * 8);
}

#define FR_MUL(r, a, b) mont_mul(r, a, b, FR_MOD, FR_INV, FRN)
#define FR_ADD(r, a, b) mod_add(r, a, b, FR_MOD, FRN)
#define FR_SUB(r, a, b) mod_sub(r, a, b, FR_MOD, FRN)
#define FQ_MUL(r, a, b) mont_mul(r, a, b, FQ_MOD, FQ_INV, FQN)
#define FQ_ADD(r, a, b) mod_add(r, a, b, FQ_MOD, FQN)
#define FQ_SUB(r, a, b) mod_sub(r, a, b, FQ_MOD, FQN)

static void compute_r_powers(u64 *r1, u64 *r2, const u64 *p, int n) {
    u64 t[FQN];
    memset(t, 0, sizeof(t));
    t[0] = 1;
    for (int i = 0; i < 2 * 64 * n; i++) {
        u64 c = bn_add(t, t, t, n);
        if (c || bn_geq(t, p, n)) bn_sub(t, t, p, n);
        if (i == 64 * n - 1) memcpy(r1, t, n * 8);
    }
    memcpy(r2, t, n * 8);
}
static u64 neg_inv64(u64 p0) {
    u64 x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - p0 * x;
    return (u64)0 - x;
}

void orc_init(void) {
    if (g_init) return;
    FR_INV = neg_inv64(FR_MOD[0]);
    FQ_INV = neg_inv64(FQ_MOD[0]);
    compute_r_powers(FR_R1, FR_R2, FR_MOD, FRN);
    compute_r_powers(FQ_R1, FQ_R2, FQ_MOD, FQN);
    g_init = 1;
}

/* a^e for a multi-limb exponent e (little-endian limbs). */
static void mod_pow(u64 *r, const u64 *a, const u64 *e, int elimbs, const u64 *p, u64 inv, const u64 *one, int n) {
    u64 acc[FQN], base[FQN];
    memcpy(acc, one, n * 8);
    memcpy(base, a, n * 8);
    for (int i = 0; i < elimbs * 64; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) mont_mul(acc, acc, base, p, inv, n);
        mont_mul(base, base, base, p, inv, n);
    }
    memcpy(r, acc, n * 8);
}
static void fr_inv(u64 *r, const u64 *a) {
    u64 e[FRN];
    u64 two[FRN] = {2, 0, 0, 0};
    bn_sub(e, FR_MOD, two, FRN);
    mod_pow(r, a, e, FRN, FR_MOD, FR_INV, FR_R1, FRN);
}
static void fq_inv(u64 *r, const u64 *a) {
    u64 e[FQN];
    u64 two[FQN] = {2, 0, 0, 0, 0, 0};
    bn_sub(e, FQ_MOD, two, FQN);
    mod_pow(r, a, e, FQN, FQ_MOD, FQ_INV, FQ_R1, FQN);
}

/* ---------------------------------------------------------------- exported field helpers */
void orc_fr_to_mont(const u64 *in, u64 *out, size_t n) {
    orc_init();
    for (size_t i = 0; i < n; i++) FR_MUL(out + 4 * i, in + 4 * i, FR_R2);
}
void orc_fr_from_mont(const u64 *in, u64 *out, size_t n) {
    orc_init();
    u64 one[FRN] = {1, 0, 0, 0};
    for (size_t i = 0; i < n; i++) FR_MUL(out + 4 * i, in + 4 * i, one);
}
void orc_fq_to_mont(const u64 *in, u64 *out, size_t n) {
    orc_init();
    for (size_t i = 0; i < n; i++) FQ_MUL(out + 6 * i, in + 6 * i, FQ_R2);
}
void orc_fq_from_mont(const u64 *in, u64 *out, size_t n) {
    orc_init();
    u64 one[FQN] = {1, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) FQ_MUL(out + 6 * i, in + 6 * i, one);
}
void orc_fr_mul_vec(const u64 *a, const u64 *b, u64 *out, size_t n) {
    orc_init();
    for (size_t i = 0; i < n; i++) FR_MUL(out + 4 * i, a + 4 * i, b + 4 * i);
}
void orc_fr_add_vec(const u64 *a, const u64 *b, u64 *out, size_t n) {
    orc_init();
    for (size_t i = 0; i < n; i++) FR_ADD(out + 4 * i, a + 4 * i, b + 4 * i);
}
void orc_fr_sub_vec(const u64 *a, const u64 *b, u64 *out, size_t n) {
    orc_init();
    for (size_t i = 0; i < n; i++) FR_SUB(out + 4 * i, a + 4 * i, b + 4 * i);
}
void orc_fr_inv_vec(const u64 *a, u64 *out, size_t n) {
    orc_init();
    for (size_t i = 0; i < n; i++) fr_inv(out + 4 * i, a + 4 * i);
}
void orc_fq_mul_vec(const u64 *a, const u64 *b, u64 *out, size_t n) {
    orc_init();
    for (size_t i = 0; i < n; i++) FQ_MUL(out + 6 * i, a + 6 * i, b + 6 * i);
}
void orc_fq_add_vec(const u64 *a, const u64 *b, u64 *out, size_t n) {
    orc_init();
    for (size_t i = 0; i < n; i++) FQ_ADD(out + 6 * i, a + 6 * i, b + 6 * i);
}
void orc_fq_sub_vec(const u64 *a, const u64 *b, u64 *out, size_t n) {
    orc_init();
    for (size_t i = 0; i < n; i++) FQ_SUB(out + 6 * i, a + 6 * i, b + 6 * i);
}

/* ---------------------------------------------------------------- G1 (Jacobian) */
typedef struct {
    u64 x[FQN], y[FQN], z[FQN];
} g1_t;

static void g1_set_identity(g1_t *p) {
    /* ark's G1Projective::zero() = (1, 1, 0) */
    memcpy(p->x, FQ_R1, sizeof(p->x));
    memcpy(p->y, FQ_R1, sizeof(p->y));
    memset(p->z, 0, sizeof(p->z));
}
static inline int g1_is_identity(const g1_t *p) { return bn_is_zero(p->z, FQN); }

static void g1_double(g1_t *r, const g1_t *p) {
    if (g1_is_identity(p)) {
        g1_set_identity(r);
        return;
    }
    u64 A[FQN], B[FQN], C[FQN], D[FQN], E[FQN], F[FQN], t[FQN], X3[FQN], Y3[FQN], Z3[FQN];
    FQ_MUL(A, p->x, p->x);
    FQ_MUL(B, p->y, p->y);
    FQ_MUL(C, B, B);
    FQ_ADD(t, p->x, B);
    FQ_MUL(t, t, t);
    FQ_SUB(t, t, A);
    FQ_SUB(t, t, C);
    FQ_ADD(D, t, t);
    FQ_ADD(E, A, A);
    FQ_ADD(E, E, A);
    FQ_MUL(F, E, E);
    FQ_SUB(X3, F, D);
    FQ_SUB(X3, X3, D);
    FQ_SUB(t, D, X3);
    FQ_MUL(Y3, E, t);
    FQ_ADD(C, C, C);
    FQ_ADD(C, C, C);
    FQ_ADD(C, C, C);
    FQ_SUB(Y3, Y3, C);
    FQ_MUL(Z3, p->y, p->z);
    FQ_ADD(Z3, Z3, Z3);
    memcpy(r->x, X3, sizeof(X3));
    memcpy(r->y, Y3, sizeof(Y3));
    memcpy(r->z, Z3, sizeof(Z3));
}

static void g1_add(g1_t *r, const g1_t *p, const g1_t *q) {
    if (g1_is_identity(p)) {
        *r = *q;
        return;
    }
    if (g1_is_identity(q)) {
        *r = *p;
        return;
    }
    u64 Z1Z1[FQN], Z2Z2[FQN], U1[FQN], U2[FQN], S1[FQN], S2[FQN], H[FQN], Rr[FQN], HH[FQN], HHH[FQN], V[FQN], t[FQN];
    FQ_MUL(Z1Z1, p->z, p->z);
    FQ_MUL(Z2Z2, q->z, q->z);
    FQ_MUL(U1, p->x, Z2Z2);
    FQ_MUL(U2, q->x, Z1Z1);
    FQ_MUL(S1, p->y, q->z);
    FQ_MUL(S1, S1, Z2Z2);
    FQ_MUL(S2, q->y, p->z);
    FQ_MUL(S2, S2, Z1Z1);
    if (bn_eq(U1, U2, FQN)) {
        if (bn_eq(S1, S2, FQN)) {
            g1_double(r, p);
        } else {
            g1_set_identity(r);
        }
        return;
    }
    FQ_SUB(H, U2, U1);
    FQ_SUB(Rr, S2, S1);
    FQ_MUL(HH, H, H);
    FQ_MUL(HHH, H, HH);
    FQ_MUL(V, U1, HH);
    g1_t o;
    FQ_MUL(o.x, Rr, Rr);
    FQ_SUB(o.x, o.x, HHH);
    FQ_SUB(o.x, o.x, V);
    FQ_SUB(o.x, o.x, V);
    FQ_SUB(t, V, o.x);
    FQ_MUL(o.y, Rr, t);
    FQ_MUL(t, S1, HHH);
    FQ_SUB(o.y, o.y, t);
    FQ_MUL(o.z, p->z, q->z);
    FQ_MUL(o.z, o.z, H);
    *r = o;
}

/* Mixed addition: q is affine (x, y), identity encoded as (0, 0). */
static void g1_add_affine(g1_t *r, const g1_t *p, const u64 *qx, const u64 *qy) {
    if (bn_is_zero(qx, FQN) && bn_is_zero(qy, FQN)) {
        *r = *p;
        return;
    }
    g1_t q;
    memcpy(q.x, qx, sizeof(q.x));
    memcpy(q.y, qy, sizeof(q.y));
    memcpy(q.z, FQ_R1, sizeof(q.z));
    g1_add(r, p, &q);
}

/* `G1 * Fr` as the reference performs it per coefficient (common/kzg.rs:34,56,83): the scalar
 * leaves Montgomery form, then double-and-add from the most significant set bit. */
static void g1_mul_canon(g1_t *r, const g1_t *p, const u64 *k) {
    g1_t acc;
    g1_set_identity(&acc);
    int top = -1;
    for (int i = 255; i >= 0; i--) {
        if ((k[i / 64] >> (i % 64)) & 1) {
            top = i;
            break;
        }
    }
    for (int i = top; i >= 0; i--) {
        g1_double(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) g1_add(&acc, &acc, p);
    }
    *r = acc;
}
static void g1_mul_mont(g1_t *r, const g1_t *p, const u64 *k_mont) {
    u64 one[FRN] = {1, 0, 0, 0}, k[FRN];
    FR_MUL(k, k_mont, one);
    g1_mul_canon(r, p, k);
}

void orc_g1_generator(u64 *out) {
    orc_init();
    g1_t *g = (g1_t *)out;
    FQ_MUL(g->x, G1X_CANON, FQ_R2);
    FQ_MUL(g->y, G1Y_CANON, FQ_R2);
    memcpy(g->z, FQ_R1, sizeof(g->z));
}
void orc_g1_identity(u64 *out) {
    orc_init();
    g1_set_identity((g1_t *)out);
}
void orc_g1_add(const u64 *p, const u64 *q, u64 *out) {
    orc_init();
    g1_t r;
    g1_add(&r, (const g1_t *)p, (const g1_t *)q);
    memcpy(out, &r, sizeof(r));
}
void orc_g1_double(const u64 *p, u64 *out) {
    orc_init();
    g1_t r;
    g1_double(&r, (const g1_t *)p);
    memcpy(out, &r, sizeof(r));
}
void orc_g1_neg(const u64 *p, u64 *out) {
    orc_init();
    g1_t r = *(const g1_t *)p;
    mod_neg(r.y, r.y, FQ_MOD, FQN);
    memcpy(out, &r, sizeof(r));
}
void orc_g1_mul(const u64 *p, const u64 *scalar_mont, u64 *out) {
    orc_init();
    g1_t r;
    g1_mul_mont(&r, (const g1_t *)p, scalar_mont);
    memcpy(out, &r, sizeof(r));
}
/* Jacobian -> affine (x | y Montgomery), one inversion each; identity -> (0, 0).
 * Returns the number of identity points. */
size_t orc_g1_to_affine(const u64 *pts, size_t n, u64 *out) {
    orc_init();
    size_t n_inf = 0;
    for (size_t i = 0; i < n; i++) {
        const g1_t *p = (const g1_t *)(pts + 18 * i);
        u64 *o = out + 12 * i;
        if (g1_is_identity(p)) {
            memset(o, 0, 96);
            n_inf++;
            continue;
        }
        u64 zi[FQN], zi2[FQN];
        fq_inv(zi, p->z);
        FQ_MUL(zi2, zi, zi);
        FQ_MUL(o, p->x, zi2);
        FQ_MUL(zi2, zi2, zi);
        FQ_MUL(o + 6, p->y, zi2);
    }
    return n_inf;
}
void orc_g1_from_affine(const u64 *aff, size_t n, u64 *out) {
    orc_init();
    for (size_t i = 0; i < n; i++) {
        const u64 *a = aff + 12 * i;
        g1_t *p = (g1_t *)(out + 18 * i);
        if (bn_is_zero(a, 12)) {
            g1_set_identity(p);
        } else {
            memcpy(p->x, a, 48);
            memcpy(p->y, a + 6, 48);
            memcpy(p->z, FQ_R1, 48);
        }
    }
}
int orc_g1_eq(const u64 *p, const u64 *q) {
    orc_init();
    u64 a[12], b[12];
    orc_g1_to_affine(p, 1, a);
    orc_g1_to_affine(q, 1, b);
    return bn_eq(a, b, 12) && (g1_is_identity((const g1_t *)p) == g1_is_identity((const g1_t *)q));
}
/* 48-byte compressed encoding (big-endian x; bit7 compressed, bit6 infinity, bit5 y > -y),
 * as written by serialize_compressed at common/utils.rs:70-72 and common/proof.rs:65. */
void orc_g1_compress(const u64 *pts, size_t n, uint8_t *out) {
    orc_init();
    u64 half[FQN], one[FQN] = {1, 0, 0, 0, 0, 0};
    bn_sub(half, FQ_MOD, one, FQN); /* (q-1)/2 */
    for (int i = 0; i < FQN; i++) half[i] = (half[i] >> 1) | (i + 1 < FQN ? half[i + 1] << 63 : 0);
    for (size_t i = 0; i < n; i++) {
        uint8_t *o = out + 48 * i;
        u64 aff[12], x[FQN], y[FQN];
        if (orc_g1_to_affine(pts + 18 * i, 1, aff)) {
            memset(o, 0, 48);
            o[0] = 0xC0;
            continue;
        }
        FQ_MUL(x, aff, one);
        FQ_MUL(y, aff + 6, one);
        for (int b = 0; b < 48; b++) o[b] = (uint8_t)(x[(47 - b) / 8] >> (8 * ((47 - b) % 8)));
        o[0] |= 0x80;
        /* y > (q-1)/2 */
        if (bn_geq(y, half, FQN) && !bn_eq(y, half, FQN)) o[0] |= 0x20;
    }
}

/* ---------------------------------------------------------------- KZG (reference algorithms) */
/* common/kzg.rs:24-45 with tau supplied by the caller: sequential accumulator *= tau. */
void orc_kzg_setup(const u64 *tau_mont, size_t degree, u64 *out_pts) {
    orc_init();
    g1_t acc;
    orc_g1_generator((u64 *)&acc);
    memcpy(out_pts, &acc, sizeof(acc));
    for (size_t i = 1; i <= degree; i++) {
        g1_mul_mont(&acc, &acc, tau_mont);
        memcpy(out_pts + 18 * i, &acc, sizeof(acc));
    }
}

/* common/kzg.rs:53-58 / :79-85: one full double-and-add per coefficient, then the sum.
 * `threads` > 1 splits the index range over OpenMP threads (the reference is single-threaded);
 * the group sum is associative, so the result is the same point.  n == 0 writes the identity
 * and returns -1 (the reference panics on the empty reduce). */
int orc_msm_naive(const u64 *pts_proj, const u64 *scalars_mont, size_t n, u64 *out, int threads) {
    orc_init();
    g1_t total;
    g1_set_identity(&total);
    if (n == 0) {
        memcpy(out, &total, sizeof(total));
        return -1;
    }
    if (threads < 1) threads = 1;
    g1_t *partial = (g1_t *)malloc(sizeof(g1_t) * threads);
#pragma omp parallel num_threads(threads)
    {
        int t = 0, nt = 1;
#ifdef _OPENMP
        t = omp_get_thread_num();
        nt = omp_get_num_threads();
#endif
        g1_t acc, term;
        g1_set_identity(&acc);
        size_t lo = n * (size_t)t / nt, hi = n * (size_t)(t + 1) / nt;
        for (size_t i = lo; i < hi; i++) {
            g1_mul_mont(&term, (const g1_t *)(pts_proj + 18 * i), scalars_mont + 4 * i);
            g1_add(&acc, &acc, &term);
        }
        partial[t] = acc;
        if (t == 0) {
            for (int k = nt; k < threads; k++) g1_set_identity(&partial[k]);
        }
    }
    for (int t = 0; t < threads; t++) g1_add(&total, &total, &partial[t]);
    free(partial);
    memcpy(out, &total, sizeof(total));
    return 0;
}

/* The fair CPU algorithm: signed-digit Pippenger over affine points (x|y, identity (0,0)),
 * window c, windows processed in parallel.  Same result as orc_msm_naive. */
int orc_msm_pippenger(const u64 *pts_affine, const u64 *scalars_mont, size_t n, u64 *out, int c, int threads) {
    orc_init();
    g1_t total;
    g1_set_identity(&total);
    if (n == 0) {
        memcpy(out, &total, sizeof(total));
        return -1;
    }
    if (c < 2) c = 2;
    if (c > 20) c = 20;
    if (threads < 1) threads = 1;
    int W = 255 / c + 1;
    size_t nb = (size_t)1 << (c - 1);
    /* signed digits */
    int32_t *digits = (int32_t *)malloc(sizeof(int32_t) * n * W);
    u64 one[FRN] = {1, 0, 0, 0};
#pragma omp parallel for num_threads(threads) schedule(static)
    for (size_t i = 0; i < n; i++) {
        u64 k[FRN + 1];
        FR_MUL(k, scalars_mont + 4 * i, one);
        k[FRN] = 0;
        int carry = 0;
        for (int w = 0; w < W; w++) {
            int bit = w * c;
            u64 v = k[bit / 64] >> (bit % 64);
            if (bit % 64 + c > 64) v |= k[bit / 64 + 1] << (64 - bit % 64);
            int64_t d = (int64_t)(v & (((u64)1 << c) - 1)) + carry;
            carry = 0;
            if (d > (int64_t)nb) {
                d -= (int64_t)1 << c;
                carry = 1;
            }
            digits[i * W + w] = (int32_t)d;
        }
    }
    g1_t *wsum = (g1_t *)malloc(sizeof(g1_t) * W);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int w = 0; w < W; w++) {
        g1_t *buckets = (g1_t *)malloc(sizeof(g1_t) * nb);
        for (size_t b = 0; b < nb; b++) g1_set_identity(&buckets[b]);
        for (size_t i = 0; i < n; i++) {
            int32_t d = digits[i * W + w];
            if (d == 0) continue;
            const u64 *px = pts_affine + 12 * i;
            u64 ny[FQN];
            if (d > 0) {
                g1_add_affine(&buckets[d - 1], &buckets[d - 1], px, px + 6);
            } else {
                mod_neg(ny, px + 6, FQ_MOD, FQN);
                if (bn_is_zero(px, 12)) continue;
                g1_add_affine(&buckets[-d - 1], &buckets[-d - 1], px, ny);
            }
        }
        g1_t run, sum;
        g1_set_identity(&run);
        g1_set_identity(&sum);
        for (size_t b = nb; b-- > 0;) {
            g1_add(&run, &run, &buckets[b]);
            g1_add(&sum, &sum, &run);
        }
        wsum[w] = sum;
        free(buckets);
    }
    for (int w = W - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) g1_double(&total, &total);
        g1_add(&total, &total, &wsum[w]);
    }
    free(wsum);
    free(digits);
    memcpy(out, &total, sizeof(total));
    return 0;
}

/* ---------------------------------------------------------------- polynomials over Fr */
/* Horner evaluation (`f.evaluate(&u)`, common/kzg.rs:65). */
void orc_fr_eval(const u64 *coeffs, size_t n, const u64 *u, u64 *out) {
    orc_init();
    u64 acc[FRN] = {0, 0, 0, 0};
    for (size_t i = n; i-- > 0;) {
        FR_MUL(acc, acc, u);
        FR_ADD(acc, acc, coeffs + 4 * i);
    }
    memcpy(out, acc, sizeof(acc));
}

/* Schoolbook long division a = q*b + r (`divide_with_q_and_r`, common/kzg.rs:74-76,
 * common/protocols.rs:162-163,177-178).  a has na coefficients, b has nb >= 1 with a
 * non-zero leading coefficient.  q receives na-nb+1 coefficients (caller trims), rem nb-1. */
int orc_poly_divmod(const u64 *a, size_t na, const u64 *b, size_t nb, u64 *q, u64 *rem) {
    orc_init();
    if (nb == 0 || bn_is_zero(b + 4 * (nb - 1), FRN)) return -1;
    if (na < nb) {
        if (rem) memcpy(rem, a, na * 32);
        return 0;
    }
    u64 *w = (u64 *)malloc(na * 32);
    memcpy(w, a, na * 32);
    u64 linv[FRN], coef[FRN], t[FRN];
    fr_inv(linv, b + 4 * (nb - 1));
    for (size_t s = na - nb + 1; s-- > 0;) {
        FR_MUL(coef, w + 4 * (s + nb - 1), linv);
        memcpy(q + 4 * s, coef, 32);
        if (bn_is_zero(coef, FRN)) continue;
        for (size_t i = 0; i < nb; i++) {
            FR_MUL(t, coef, b + 4 * i);
            FR_SUB(w + 4 * (s + i), w + 4 * (s + i), t);
        }
    }
    if (rem) memcpy(rem, w, (nb - 1) * 32);
    free(w);
    return 0;
}

/* Schoolbook product (ark-poly `*` semantics; the product itself is unique). */
void orc_poly_mul(const u64 *a, size_t na, const u64 *b, size_t nb, u64 *out) {
    orc_init();
    memset(out, 0, (na + nb - 1) * 32);
    u64 t[FRN];
    for (size_t i = 0; i < na; i++)
        for (size_t j = 0; j < nb; j++) {
            FR_MUL(t, a + 4 * i, b + 4 * j);
            FR_ADD(out + 4 * (i + j), out + 4 * (i + j), t);
        }
}

/* omega = 7^((r-1)/n) for n = 2^logn (common/utils.rs:11-38). */
static void fr_root_of_unity(u64 *w, unsigned logn) {
    u64 e[FRN], seven_c[FRN] = {7, 0, 0, 0}, seven[FRN];
    /* (r-1) >> logn */
    u64 rm1[FRN];
    u64 one[FRN] = {1, 0, 0, 0};
    bn_sub(rm1, FR_MOD, one, FRN);
    for (int i = 0; i < FRN; i++) {
        e[i] = logn ? (rm1[i] >> logn) | (i + 1 < FRN ? rm1[i + 1] << (64 - logn) : 0) : rm1[i];
    }
    FR_MUL(seven, seven_c, FR_R2);
    if (logn == 0) {
        memcpy(w, FR_R1, 32);
        return;
    }
    mod_pow(w, seven, e, FRN, FR_MOD, FR_INV, FR_R1, FRN);
}
void orc_fr_root_of_unity(unsigned logn, u64 *out) {
    orc_init();
    fr_root_of_unity(out, logn);
}

/* In-place radix-2 transform, natural order in and out.  Forward: data[j] = f(omega^j)
 * (the evaluations the reference takes one Horner at a time, common/protocols.rs:243-250,
 * 571-588); inverse: coefficients of the interpolant over Omega
 * (common/polynomials.rs:50-58 when x_points = construct_Omega(n)). */
void orc_fr_ntt(u64 *data, unsigned logn, int inverse) {
    orc_init();
    size_t n = (size_t)1 << logn;
    for (size_t i = 0; i < n; i++) {
        size_t j = 0;
        for (unsigned b = 0; b < logn; b++) j |= ((i >> b) & 1) << (logn - 1 - b);
        if (i < j) {
            u64 t[FRN];
            memcpy(t, data + 4 * i, 32);
            memcpy(data + 4 * i, data + 4 * j, 32);
            memcpy(data + 4 * j, t, 32);
        }
    }
    u64 wn[FRN];
    fr_root_of_unity(wn, logn);
    if (inverse) fr_inv(wn, wn);
    u64 *tw = (u64 *)malloc((n / 2 + 1) * 32);
    memcpy(tw, FR_R1, 32);
    for (size_t i = 1; i < n / 2; i++) FR_MUL(tw + 4 * i, tw + 4 * (i - 1), wn);
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t step = n / len;
        for (size_t s = 0; s < n; s += len) {
            for (size_t k = 0; k < len / 2; k++) {
                u64 *u = data + 4 * (s + k), *v = data + 4 * (s + k + len / 2), t[FRN], x[FRN];
                FR_MUL(t, v, tw + 4 * (k * step));
                FR_ADD(x, u, t);
                FR_SUB(v, u, t);
                memcpy(u, x, 32);
            }
        }
    }
    free(tw);
    if (inverse) {
        u64 nn[FRN] = {n, 0, 0, 0}, ninv[FRN];
        FR_MUL(nn, nn, FR_R2);
        fr_inv(ninv, nn);
        for (size_t i = 0; i < n; i++) FR_MUL(data + 4 * i, data + 4 * i, ninv);
    }
}

/* Lagrange interpolation exactly as common/polynomials.rs:14-58 does it: product of all
 * monomials, one long division and one O(n) denominator per point, then the weighted sum.
 * out receives n coefficients.  Returns -1 on coincident points (inverse of zero). */
int orc_interpolate_lagrange(const u64 *xs, const u64 *ys, size_t n, u64 *out, int threads) {
    orc_init();
    if (threads < 1) threads = 1;
    u64 *prod = (u64 *)calloc(n + 1, 32);
    memcpy(prod, FR_R1, 32);
    for (size_t i = 0; i < n; i++) {
        /* prod *= (X - x_i): new[j] = old[j-1] - x_i*old[j], highest j first (in place) */
        u64 t[FRN], zero[FRN] = {0, 0, 0, 0};
        for (size_t j = i + 2; j-- > 0;) {
            const u64 *lower = j > 0 ? prod + 4 * (j - 1) : zero;
            if (j <= i) {
                FR_MUL(t, prod + 4 * j, xs + 4 * i);
            } else {
                memset(t, 0, sizeof(t));
            }
            FR_SUB(prod + 4 * j, lower, t);
        }
    }
    memset(out, 0, n * 32);
    int bad = 0;
#pragma omp parallel num_threads(threads)
    {
        u64 *num = (u64 *)malloc((n + 1) * 32);
        u64 *acc = (u64 *)calloc(n, 32);
#pragma omp for schedule(static)
        for (size_t i = 0; i < n; i++) {
            /* numerator = prod / (X - x_i): synthetic division */
            u64 carry[FRN], t[FRN];
            memcpy(carry, prod + 4 * n, 32);
            for (size_t j = n; j-- > 0;) {
                memcpy(num + 4 * j, carry, 32);
                FR_MUL(t, carry, xs + 4 * i);
                FR_ADD(carry, prod + 4 * j, t);
            }
            u64 den[FRN], d[FRN];
            memcpy(den, FR_R1, 32);
            for (size_t j = 0; j < n; j++) {
                if (j == i) continue;
                FR_SUB(d, xs + 4 * i, xs + 4 * j);
                FR_MUL(den, den, d);
            }
            if (bn_is_zero(den, FRN)) {
                bad = 1;
                continue;
            }
            fr_inv(den, den);
            FR_MUL(den, den, ys + 4 * i);
            for (size_t j = 0; j < n; j++) {
                FR_MUL(t, num + 4 * j, den);
                FR_ADD(acc + 4 * j, acc + 4 * j, t);
            }
        }
#pragma omp critical
        for (size_t j = 0; j < n; j++) FR_ADD(out + 4 * j, out + 4 * j, acc + 4 * j);
        free(num);
        free(acc);
    }
    free(prod);
    return bad ? -1 : 0;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
