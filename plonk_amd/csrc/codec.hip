// Compressed G1 encoding on the device: the 48-byte form ark-serialize writes and reads for
// BLS12-381 G1 (big-endian x; in the first byte bit 7 = compressed, bit 6 = point at infinity,
// bit 5 = y is the larger of {y, q - y}).  The reference stores the SRS that way
// (plonk/src/setup_global_params/json.rs:22-48, `G1::deserialize_compressed`) and prints every
// commitment that way (plonk/src/common/utils.rs:70-72, plonk/src/common/proof.rs:58-268).
// Decoding a point costs a square root (q = 3 mod 4: one exponentiation by (q+1)/4) and, as
// `deserialize_compressed` validates, a subgroup check ([r]P = O): minutes on a CPU for a 2^24-point
// SRS, well under a second here.
#include <cstring>

#include "context.hpp"
#include "g1_28.cuh"
#include "poly.hpp"

namespace plonk {

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

struct Exponent384 {
    uint32_t w[12];
};

// base^e, e a 384-bit integer (most significant bit first, leading zeros skipped by the caller's
// choice of top_bit)
static __device__ Fq fq_pow(const Fq &base, const Exponent384 &e, int top_bit) {
    Fq acc = fe_one<FqCfg>();
#pragma unroll 1
    for (int b = top_bit; b >= 0; b--) {
        acc = fe_sqr(acc);
        if ((e.w[b >> 5] >> (b & 31)) & 1) acc = fe_mul(acc, base);
    }
    return acc;
}

// a > b for canonical (non-Montgomery) values
static __device__ bool fq_gt(const Fq &a, const Fq &b) {
    for (int i = FqCfg::N - 1; i >= 0; i--) {
        if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
    }
    return false;
}

// [r]P == O for an affine point on the curve (r = the order of the subgroup, 255 bits)
static __device__ bool in_subgroup(const G1Affine &p) {
    G1Xyzz acc = xyzz_from_affine(p);  // the top bit of r
#pragma unroll 1
    for (int b = 253; b >= 0; b--) {
        acc = xyzz_dbl(acc);
        if ((FrCfg::mod(b >> 5) >> (b & 31)) & 1) xyzz_madd(acc, p, false);
    }
    return xyzz_is_identity(acc);
}

// One lane per point.  *bad receives the smallest index of an encoding `deserialize_compressed`
// rejects: compression bit clear, x >= q, x^3 + 4 not a square, point outside the subgroup.
__global__ __launch_bounds__(128) void k_g1_decompress(const uint8_t *__restrict__ bytes, size_t n, Exponent384 sqrt_exp,
                                                      Fq half_q, int check_subgroup, G1Affine *__restrict__ out,
                                                      unsigned long long *__restrict__ bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *b = bytes + 48 * i;
    const uint32_t flags = b[0] >> 5;
    Fq x;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const uint8_t *p = b + 44 - 4 * w;  // limb w = bytes [44-4w, 48-4w), big-endian
        x.l[w] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    }
    x.l[11] &= 0x1fffffffu;
    bool ok = (flags & 4) != 0;
    Fq qm;
#pragma unroll
    for (int w = 0; w < 12; w++) qm.l[w] = FqCfg::mod(w);
    if (!fq_gt(qm, x)) ok = false;  // x must be below q
    G1Affine pt = affine_identity();
    if (ok && !(flags & 2)) {
        const Fq xm = fe_to_mont(x);
        const Fq one = fe_one<FqCfg>();
        const Fq two = fe_add(one, one);
        const Fq rhs = fe_add(fe_mul(fe_sqr(xm), xm), fe_add(two, two));  // x^3 + 4
        Fq y = fq_pow(rhs, sqrt_exp, 379);                              // (q+1)/4 has 380 bits
        if (!fe_eq(fe_sqr(y), rhs)) {
            ok = false;
        } else {
            const bool larger = fq_gt(fe_from_mont(y), half_q);  // y > (q-1)/2
            if (larger != ((flags & 1) != 0)) y = fe_neg(y);
            pt.x = xm;
            pt.y = y;
            if (check_subgroup && !in_subgroup(pt)) ok = false;
        }
    }
    out[i] = pt;
    if (!ok) atomicMin(bad, (unsigned long long)i);
}

// affine (x, y) -> Jacobian (x, y, 1); identity -> (1, 1, 0) like ark's `zero()`
__global__ __launch_bounds__(256) void k_affine_to_jac12(const G1Affine *__restrict__ in, G1Jac *__restrict__ out,
                                                        size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine a = in[i];
    out[i] = affine_is_identity(a) ? jac_identity() : xyzz_to_jac(xyzz_from_affine(a));
}

// One lane per point: Jacobian -> affine (one inversion per lane) -> 48 bytes.
__global__ __launch_bounds__(128) void k_g1_compress(const G1Jac *__restrict__ in, size_t n, Fq half_q,
                                                    uint8_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Jac p = in[i];
    uint8_t *b = out + 48 * i;
    if (fe_is_zero(p.z)) {
        b[0] = 0xc0;
        for (int k = 1; k < 48; k++) b[k] = 0;
        return;
    }
    const G1Affine a = jac_to_affine_with_inv(p, fe_inv(p.z));
    const Fq x = fe_from_mont(a.x), y = fe_from_mont(a.y);
#pragma unroll
    for (int w = 0; w < 12; w++) {
        uint8_t *q = b + 44 - 4 * w;
        q[0] = (uint8_t)(x.l[w] >> 24);
        q[1] = (uint8_t)(x.l[w] >> 16);
        q[2] = (uint8_t)(x.l[w] >> 8);
        q[3] = (uint8_t)x.l[w];
    }
    b[0] |= 0x80 | (fq_gt(y, half_q) ? 0x20 : 0);
}

static void codec_constants(Exponent384 *sqrt_exp, Fq *half_q) {
    // (q + 1) / 4 and (q - 1) / 2 from the limbs of q (q = 3 mod 4)
    uint32_t q[12];
    for (int i = 0; i < 12; i++) q[i] = FqCfg::mod(i);
    uint32_t t[12];
    uint64_t c = 1;
    for (int i = 0; i < 12; i++) {
        c += q[i];
        t[i] = (uint32_t)c;
        c >>= 32;
    }
    for (int i = 0; i < 12; i++) sqrt_exp->w[i] = (t[i] >> 2) | (i + 1 < 12 ? t[i + 1] << 30 : 0u);
    for (int i = 0; i < 12; i++) half_q->l[i] = (q[i] >> 1) | (i + 1 < 12 ? q[i + 1] << 31 : 0u);  // q odd: (q-1)/2
}

// d_bytes: n x 48 encodings; d_affine: n affine points out; *bad_index = n when every encoding was
// accepted, else the smallest rejected index (the output is then unspecified from there on).
int g1_decompress_dev(plonk_hip_ctx *ctx, const uint8_t *d_bytes, size_t n, bool check_subgroup, G1Affine *d_affine,
                      size_t *bad_index) {
    if (n == 0) {
        if (bad_index) *bad_index = 0;
        return PLONK_HIP_OK;
    }
    Exponent384 e;
    Fq half;
    codec_constants(&e, &half);
    PLONK_TRY(buf_reserve(ctx, ctx->out_stage, 64 * sizeof(G1Jac) + 4096));
    unsigned long long *d_bad = (unsigned long long *)((char *)ctx->out_stage.ptr + 64 * sizeof(G1Jac) + 64);
    unsigned long long init = ~0ull, got = 0;
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_bad, &init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_g1_decompress, dim3(cdiv(n, 128)), dim3(128), 0, ctx->stream, d_bytes, n, e, half,
                       check_subgroup ? 1 : 0, d_affine, d_bad);
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(&got, d_bad, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    PLONK_HIP_TRY(ctx, hipGetLastError());
    if (bad_index) *bad_index = got == ~0ull ? n : (size_t)got;
    return got == ~0ull ? PLONK_HIP_OK : PLONK_HIP_ERR_INVALID_POINT;
}

int g1_affine_to_jac_dev(plonk_hip_ctx *ctx, const G1Affine *d_affine, size_t n, G1Jac *d_out) {
    if (n) hipLaunchKernelGGL(k_affine_to_jac12, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, d_affine, d_out, n);
    return PLONK_HIP_OK;
}

int g1_compress_dev(plonk_hip_ctx *ctx, const G1Jac *d_pts, size_t n, uint8_t *d_bytes) {
    if (n == 0) return PLONK_HIP_OK;
    Exponent384 e;
    Fq half;
    codec_constants(&e, &half);
    hipLaunchKernelGGL(k_g1_compress, dim3(cdiv(n, 128)), dim3(128), 0, ctx->stream, d_pts, n, half, d_bytes);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

}  // namespace plonk
