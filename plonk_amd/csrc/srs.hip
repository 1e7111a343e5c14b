// SRS residency: upload / generation of tau_powers_g1, affine normalisation and the window
// tables [2^(c*w)] P_i that the bucket method reads.
#include <cstring>
#include <vector>

#include "context.hpp"
#include "g1_28.cuh"
#include "msm_core.cuh"

namespace plonk {

static constexpr int kBatch = 64;  // points per lane in the shared-inversion normalisation

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------
// Jacobian -> affine with one inversion per kBatch points (Montgomery's trick).
// scratch: n field elements.  Identity points (Z = 0) become (0, 0).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_batch_to_affine(const G1Jac *__restrict__ in, G1Affine *__restrict__ out,
                                                        size_t n, Fq *__restrict__ scratch) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t base = t * kBatch;
    if (base >= n) return;
    int cnt = (int)(n - base < (size_t)kBatch ? n - base : (size_t)kBatch);
    Fq acc = fe_one<FqCfg>();
    for (int j = 0; j < cnt; j++) {
        Fq z = in[base + j].z;
        scratch[base + j] = acc;  // product of the Z's before j
        if (!fe_is_zero(z)) acc = fe_mul(acc, z);
    }
    Fq inv = fe_inv(acc);
    for (int j = cnt - 1; j >= 0; j--) {
        G1Jac p = in[base + j];
        if (fe_is_zero(p.z)) {
            out[base + j] = affine_identity();
            continue;
        }
        Fq zinv = fe_mul(inv, scratch[base + j]);
        inv = fe_mul(inv, p.z);
        out[base + j] = jac_to_affine_with_inv(p, zinv);
    }
}

// next[i] = [2^c] prev[i] as a Jacobian point.
__global__ __launch_bounds__(128) void k_window_shift(const G1Affine *__restrict__ prev, G1Jac *__restrict__ next,
                                                     size_t n, int c) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine a = prev[i];
    G1Jac p;
    if (affine_is_identity(a)) {
        p = jac_identity();
    } else {
        p.x = a.x;
        p.y = a.y;
        p.z = fe_one<FqCfg>();
        for (int k = 0; k < c; k++) p = jac_dbl(p);
    }
    next[i] = p;
}

// One level of the resident window table: 12 x 32-bit affine points -> the 28-bit limb form the
// accumulation kernel reads (one 128-byte line per point).
__global__ __launch_bounds__(256) void k_to_table28(const G1Affine *__restrict__ in, G1Affine28 *__restrict__ out,
                                                   size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = affine28_from(in[i]);
}

struct PowTable {
    Fr pw[32];  // tau^(2^k)
};

// out[i] = first_power * tau^i
__global__ __launch_bounds__(256) void k_fr_powers(PowTable tab, Fr first_power, size_t n, Fr *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr acc = first_power;
    uint32_t e = (uint32_t)i;
#pragma unroll 1
    for (int k = 0; k < 32; k++) {
        if ((e >> k) & 1) acc = fe_mul(acc, tab.pw[k]);
    }
    out[i] = acc;
}

// out[i] = [s_i] G from the generator table gen[w][d] = [d * 256^w] G (d = 0 is the identity).
__global__ __launch_bounds__(128) void k_fixed_base(const Fr *__restrict__ scalars, size_t n,
                                                   const G1Affine *__restrict__ gen, G1Jac *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr k = fe_from_mont(scalars[i]);
    G1Xyzz acc = xyzz_identity();
#pragma unroll 1
    for (int w = 0; w < 32; w++) {
        uint32_t limb = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) limb = (w >> 2) == j ? k.l[j] : limb;
        uint32_t d = (limb >> (8 * (w & 3))) & 0xffu;
        if (d) xyzz_madd(acc, gen[w * 256 + d], false);
    }
    out[i] = xyzz_to_jac(acc);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static const uint32_t kGenX[12] = {0xfd530c16u, 0x5cb38790u, 0x9976fff5u, 0x7817fc67u, 0x143ba1c1u, 0x154f95c7u,
                                   0xf3d0e747u, 0xf0ae6acdu, 0x21dbf440u, 0xedce6eccu, 0x9e0bfb75u, 0x12017741u};
static const uint32_t kGenY[12] = {0x0ce72271u, 0xbaac93d5u, 0x7918fd8eu, 0x8c22631au, 0x570725ceu, 0xdd595f13u,
                                   0x50405194u, 0x51ac5829u, 0xad0059c0u, 0x0e1c8c3fu, 0x5008a26au, 0x0bbc3efcu};

// Generator multiples [d * 256^w] G, built once per context on the host (8 K points).
static int ensure_gen_table(plonk_hip_ctx *ctx) {
    if (ctx->gen_table.ptr) return PLONK_HIP_OK;
    std::vector<G1Xyzz> pts(32 * 256);
    G1Affine g;
    for (int i = 0; i < 12; i++) {
        g.x.l[i] = kGenX[i];
        g.y.l[i] = kGenY[i];
    }
    G1Xyzz base = xyzz_from_affine(g);
    for (int w = 0; w < 32; w++) {
        pts[w * 256] = xyzz_identity();
        G1Xyzz acc = base;
        for (int d = 1; d < 256; d++) {
            pts[w * 256 + d] = acc;
            acc = xyzz_add(acc, base);
        }
        base = acc;  // 256 * base
    }
    // normalise: x = X/ZZ, y = Y/ZZZ with one shared inversion of prod(ZZ * ZZZ)
    std::vector<Fq> pre(pts.size());
    Fq acc = fe_one<FqCfg>();
    for (size_t i = 0; i < pts.size(); i++) {
        pre[i] = acc;
        if (!xyzz_is_identity(pts[i])) acc = fe_mul(acc, fe_mul(pts[i].zz, pts[i].zzz));
    }
    Fq inv = fe_inv(acc);
    std::vector<G1Affine> aff(pts.size());
    for (size_t i = pts.size(); i-- > 0;) {
        if (xyzz_is_identity(pts[i])) {
            aff[i] = affine_identity();
            continue;
        }
        Fq zi = fe_mul(inv, pre[i]);  // 1 / (ZZ * ZZZ)
        inv = fe_mul(inv, fe_mul(pts[i].zz, pts[i].zzz));
        aff[i].x = fe_mul(pts[i].x, fe_mul(zi, pts[i].zzz));  // X / ZZ
        aff[i].y = fe_mul(pts[i].y, fe_mul(zi, pts[i].zz));   // Y / ZZZ
    }
    PLONK_TRY(buf_reserve(ctx, ctx->gen_table, aff.size() * sizeof(G1Affine)));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(ctx->gen_table.ptr, aff.data(), aff.size() * sizeof(G1Affine),
                                      hipMemcpyHostToDevice, ctx->stream));
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PLONK_HIP_OK;
}

// Choose the window plan for n points and size the table (level 0 is filled by the caller).
static int srs_begin(plonk_hip_ctx *ctx, size_t n) {
    if (n == 0) return PLONK_HIP_ERR_INVALID_ARG;
    MsmPlan pl = msm_choose_plan(n, ctx->window_bits_override);
    if ((double)n * pl.nwin >= 2147483648.0) return PLONK_HIP_ERR_INVALID_ARG;
    ctx->srs_n = 0;
    PLONK_TRY(buf_reserve(ctx, ctx->table, n * (size_t)pl.nwin * sizeof(G1Affine28)));
    PLONK_TRY(buf_reserve(ctx, ctx->level_a, n * sizeof(G1Affine)));
    PLONK_TRY(buf_reserve(ctx, ctx->level_b, n * sizeof(G1Affine)));
    ctx->plan = pl;
    return PLONK_HIP_OK;
}

// Level 0 (12 x 32-bit affine, in ctx->level_a) -> every level of the 28-bit table: c doublings
// per level, shared-inversion normalisation, conversion.
static int srs_build_levels(plonk_hip_ctx *ctx, size_t n) {
    const MsmPlan &pl = ctx->plan;
    G1Affine28 *table = (G1Affine28 *)ctx->table.ptr;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_a, n * sizeof(G1Jac)));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_b, n * sizeof(Fq)));
    G1Jac *jac = (G1Jac *)ctx->scratch_a.ptr;
    Fq *scr = (Fq *)ctx->scratch_b.ptr;
    G1Affine *prev = (G1Affine *)ctx->level_a.ptr, *next = (G1Affine *)ctx->level_b.ptr;
    hipLaunchKernelGGL(k_to_table28, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, prev, table, n);
    for (int w = 1; w < pl.nwin; w++) {
        hipLaunchKernelGGL(k_window_shift, dim3(cdiv(n, 128)), dim3(128), 0, ctx->stream, prev, jac, n, pl.c);
        hipLaunchKernelGGL(k_batch_to_affine, dim3(cdiv(cdiv(n, kBatch), 128)), dim3(128), 0, ctx->stream, jac, next,
                           n, scr);
        hipLaunchKernelGGL(k_to_table28, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, next, table + (size_t)w * n,
                           n);
        G1Affine *t = prev;
        prev = next;
        next = t;
    }
    PLONK_HIP_TRY(ctx, hipGetLastError());
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->srs_n = n;
    return PLONK_HIP_OK;
}

int srs_load_projective(plonk_hip_ctx *ctx, const uint64_t *pts, size_t n) {
    if (!pts) return PLONK_HIP_ERR_INVALID_ARG;
    PLONK_TRY(srs_begin(ctx, n));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_a, n * sizeof(G1Jac)));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_b, n * sizeof(Fq)));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch_a.ptr, pts, n * sizeof(G1Jac), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_batch_to_affine, dim3(cdiv(cdiv(n, kBatch), 128)), dim3(128), 0, ctx->stream,
                       (const G1Jac *)ctx->scratch_a.ptr, (G1Affine *)ctx->level_a.ptr, n, (Fq *)ctx->scratch_b.ptr);
    return srs_build_levels(ctx, n);
}

// SRS from its compressed encodings (n x 48 bytes in host memory), decoded and validated on the
// device.  On a rejected encoding the context is left without an SRS.
int srs_load_compressed(plonk_hip_ctx *ctx, const uint8_t *bytes, size_t n, bool check_subgroup, size_t *bad_index) {
    if (!bytes) return PLONK_HIP_ERR_INVALID_ARG;
    PLONK_TRY(srs_begin(ctx, n));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_a, n * sizeof(G1Jac)));  // also the work area of srs_build_levels
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch_a.ptr, bytes, n * 48, hipMemcpyHostToDevice, ctx->stream));
    PLONK_TRY(g1_decompress_dev(ctx, (const uint8_t *)ctx->scratch_a.ptr, n, check_subgroup,
                                (G1Affine *)ctx->level_a.ptr, bad_index));
    return srs_build_levels(ctx, n);
}

int srs_setup(plonk_hip_ctx *ctx, const uint64_t tau[4], size_t first, size_t n, uint64_t *out_pts) {
    if (!tau || n == 0) return PLONK_HIP_ERR_INVALID_ARG;
    if (n >= 0xffffffffull) return PLONK_HIP_ERR_INVALID_ARG;
    PLONK_TRY(ensure_gen_table(ctx));
    PLONK_TRY(srs_begin(ctx, n));
    PowTable tab;
    memcpy(tab.pw[0].l, tau, sizeof(tab.pw[0].l));
    for (int k = 1; k < 32; k++) tab.pw[k] = fe_sqr(tab.pw[k - 1]);
    Fr first_power = fe_pow_u64(tab.pw[0], (uint64_t)first);  // tau^first
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, n * sizeof(Fr)));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_a, n * sizeof(G1Jac)));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_b, n * sizeof(Fq)));
    Fr *pw = (Fr *)ctx->scratch_c.ptr;
    G1Jac *jac = (G1Jac *)ctx->scratch_a.ptr;
    hipLaunchKernelGGL(k_fr_powers, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, tab, first_power, n, pw);
    hipLaunchKernelGGL(k_fixed_base, dim3(cdiv(n, 128)), dim3(128), 0, ctx->stream, pw, n,
                       (const G1Affine *)ctx->gen_table.ptr, jac);
    if (out_pts) {
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(out_pts, jac, n * sizeof(G1Jac), hipMemcpyDeviceToHost, ctx->stream));
    }
    hipLaunchKernelGGL(k_batch_to_affine, dim3(cdiv(cdiv(n, kBatch), 128)), dim3(128), 0, ctx->stream, jac,
                       (G1Affine *)ctx->level_a.ptr, n, (Fq *)ctx->scratch_b.ptr);
    return srs_build_levels(ctx, n);
}

// affine table entries -> Jacobian triples with Z = 1 (identity -> (1, 1, 0))
__global__ void k_affine_to_jac(const G1Affine28 *__restrict__ in, G1Jac *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine a = affine28_to(in[i]);
    G1Jac p;
    if (affine_is_identity(a)) {
        p = jac_identity();
    } else {
        p.x = a.x;
        p.y = a.y;
        p.z = fe_one<FqCfg>();
    }
    out[i] = p;
}

int srs_read(plonk_hip_ctx *ctx, size_t first, size_t count, uint64_t *out) {
    if (ctx->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    if (!out || first + count > ctx->srs_n) return PLONK_HIP_ERR_INVALID_ARG;
    if (count == 0) return PLONK_HIP_OK;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_a, count * sizeof(G1Jac)));
    hipLaunchKernelGGL(k_affine_to_jac, dim3(cdiv(count, 256)), dim3(256), 0, ctx->stream,
                       (const G1Affine28 *)ctx->table.ptr + first, (G1Jac *)ctx->scratch_a.ptr, count);
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->scratch_a.ptr, count * sizeof(G1Jac), hipMemcpyDeviceToHost,
                                      ctx->stream));
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PLONK_HIP_OK;
}

}  // namespace plonk
