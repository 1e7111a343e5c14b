// Probe: what would batch-affine bucket additions cost on this GPU?
//
// An affine + affine addition needs 6 field products once 1 / (x2 - x1) is at hand (against 10 for the mixed
// XYZZ addition of the accumulation loop), and Montgomery's trick turns B inversions into one inversion and
// 3 (B - 1) products.  The catch on a SIMD machine is where the B independent additions come from: a lane's
// own accumulators (K of them fit in registers) times the lanes that share one inversion through LDS.  This
// kernel is that scheme at its best: operands in registers, no bucket logic, the table points from a small
// cache-resident set:
//   every lane holds K affine accumulators and adds one table point to each per step;
//   the K denominators of a lane are multiplied up (K - 1 products), the 256 lane totals of a block go through
//   a prefix and a suffix product scan in LDS (8 steps of 2 products), ONE thread inverts the block's total
//   (a Fermat inversion on the 12 x 32-bit form), every lane recovers the inverse of its own total (2 products)
//   and of its K denominators (2 (K - 1) products), then the K additions proper (3 products each).
// With `invert` = 0 the inversion is skipped (the total itself is used: wrong points, same work otherwise), which
// prices the scheme as if inversions were free.  Results with invert = 1 are checked against the oracle by
// tests/test_gpu_probes.py.
#include "context.hpp"
#define PLONK_MONT28_ASM 1
#include "g1_28.cuh"

namespace plonk {

// canonical product on the hand-scheduled sequence: inputs below q, result below q
__device__ __forceinline__ Fq28 mulc(const Fq28 &a, const Fq28 &b) { return q28_reduce_once(q28_mul_lazy(a, b)); }

struct ScanMem {
    uint32_t v[2][Q28::N][256];  // prefix / suffix running products, [limb][lane]
    uint32_t inv[Q28::N];
};
__device__ __forceinline__ Fq28 scan_get(const ScanMem &m, int which, uint32_t t) {
    Fq28 r;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) r.l[i] = m.v[which][i][t];
    return r;
}
__device__ __forceinline__ void scan_put(ScanMem &m, int which, uint32_t t, const Fq28 &x) {
#pragma unroll
    for (int i = 0; i < Q28::N; i++) m.v[which][i][t] = x.l[i];
}

template <int K>
__global__ __launch_bounds__(256, 2) void k_batch_affine(const G1Affine28 *__restrict__ table, uint32_t table_n,
                                                         uint32_t iters, int invert, G1Affine28 *__restrict__ out,
                                                         uint32_t *__restrict__ degenerate) {
    __shared__ ScanMem mem;
    const uint32_t t = threadIdx.x;
    const size_t g = (size_t)blockIdx.x * 256 + t;
    G1Affine28 acc[K];
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = table[(g * K + k) % table_n];
    const Fq28 one = q28_one();
    for (uint32_t it = 1; it <= iters; it++) {
        Fq28 d[K], p[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const G1Affine28 pt = table[(g * K + k + (size_t)it * 1021u) % table_n];
            d[k] = q28_sub(pt.x, acc[k].x);
            if (q28_is_zero(d[k])) atomicOr(degenerate, 1u);  // equal or opposite points: not handled here
            p[k] = k ? mulc(p[k - 1], d[k]) : d[k];
        }
        // running products of the lane totals over the block, from both ends
        Fq28 pre = p[K - 1], suf = p[K - 1];
        scan_put(mem, 0, t, pre);
        scan_put(mem, 1, t, suf);
        __syncthreads();
#pragma unroll 1
        for (uint32_t off = 1; off < 256; off <<= 1) {
            const Fq28 a = t >= off ? scan_get(mem, 0, t - off) : one;
            const Fq28 b = t + off < 256 ? scan_get(mem, 1, t + off) : one;
            __syncthreads();
            pre = mulc(a, pre);
            suf = mulc(suf, b);
            scan_put(mem, 0, t, pre);
            scan_put(mem, 1, t, suf);
            __syncthreads();
        }
        if (t == 255) {
            Fq28 total = pre;
            if (invert) total = q28_from_fq(fe_inv(q28_to_fq(total)));
#pragma unroll
            for (int i = 0; i < Q28::N; i++) mem.inv[i] = total.l[i];
        }
        __syncthreads();
        Fq28 run;
#pragma unroll
        for (int i = 0; i < Q28::N; i++) run.l[i] = mem.inv[i];
        // 1 / (this lane's total) = 1 / (block total) * (totals before) * (totals after)
        if (t > 0) run = mulc(run, scan_get(mem, 0, t - 1));
        if (t < 255) run = mulc(run, scan_get(mem, 1, t + 1));
        __syncthreads();  // the scan arrays are rewritten by the next step
#pragma unroll
        for (int k = K - 1; k >= 0; k--) {
            const Fq28 dinv = k ? mulc(run, p[k - 1]) : run;
            if (k) run = mulc(run, d[k]);
            const G1Affine28 pt = table[(g * K + k + (size_t)it * 1021u) % table_n];
            const Fq28 lam = mulc(q28_sub(pt.y, acc[k].y), dinv);
            const Fq28 x3 = q28_sub(q28_sub(q28_reduce_once(q28_sqr_lazy(lam)), acc[k].x), pt.x);
            acc[k].y = q28_sub(mulc(lam, q28_sub(acc[k].x, x3)), acc[k].y);
            acc[k].x = x3;
        }
    }
#pragma unroll
    for (int k = 0; k < K; k++) out[g * K + k] = acc[k];
}

// additions per second of the scheme above with K accumulators per lane (1, 2 or 4) on the context's SRS points
// (level 0 of the window tables, at least 4096 of them); first_out: if not null, the accumulators of the first
// block after `iters` steps as Jacobian points with Z = 1 (256 * K x 18 limbs), for checking.
int bench_batch_affine(plonk_hip_ctx *ctx, int K, int iters, int invert, double *adds_per_second, uint64_t *first_out);

__global__ void k_affine28_to_jac(const G1Affine28 *__restrict__ in, size_t n, G1Jac *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine a = affine28_to(in[i]);
    G1Jac j;
    j.x = a.x;
    j.y = a.y;
    j.z = fe_one<FqCfg>();
    out[i] = j;
}

int bench_batch_affine(plonk_hip_ctx *ctx, int K, int iters, int invert, double *adds_per_second, uint64_t *first_out) {
    if (ctx->srs_n < 4096) return PLONK_HIP_ERR_NO_SRS;
    if ((K != 1 && K != 2 && K != 4) || iters < 1) return PLONK_HIP_ERR_INVALID_ARG;
    const uint32_t table_n = 4096;  // cache resident: the probe prices arithmetic, not the gather
    const int blocks = 512, threads = 256;
    const size_t lanes = (size_t)blocks * threads;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_a, lanes * K * sizeof(G1Affine28)));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, 256 * (size_t)K * sizeof(G1Jac) + 256));
    G1Affine28 *out = (G1Affine28 *)ctx->scratch_a.ptr;
    uint32_t *flag = (uint32_t *)((char *)ctx->scratch_c.ptr + 256 * (size_t)K * sizeof(G1Jac));
    const G1Affine28 *table = (const G1Affine28 *)ctx->table.ptr;
    hipEvent_t e0, e1;
    PLONK_HIP_TRY(ctx, hipEventCreate(&e0));
    PLONK_HIP_TRY(ctx, hipEventCreate(&e1));
    float best = 1e30f;
    PLONK_HIP_TRY(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
    for (int rep = 0; rep < 3; rep++) {
        PLONK_HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        if (K == 1) hipLaunchKernelGGL(k_batch_affine<1>, dim3(blocks), dim3(threads), 0, ctx->stream, table, table_n, (uint32_t)iters, invert, out, flag);
        if (K == 2) hipLaunchKernelGGL(k_batch_affine<2>, dim3(blocks), dim3(threads), 0, ctx->stream, table, table_n, (uint32_t)iters, invert, out, flag);
        if (K == 4) hipLaunchKernelGGL(k_batch_affine<4>, dim3(blocks), dim3(threads), 0, ctx->stream, table, table_n, (uint32_t)iters, invert, out, flag);
        PLONK_HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        PLONK_HIP_TRY(ctx, hipEventSynchronize(e1));
        float ms = 0;
        PLONK_HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    uint32_t degenerate = 0;
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(&degenerate, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (first_out) {
        G1Jac *jac = (G1Jac *)ctx->scratch_c.ptr;
        hipLaunchKernelGGL(k_affine28_to_jac, dim3(K), dim3(256), 0, ctx->stream, out, (size_t)256 * K, jac);
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(first_out, jac, 256 * (size_t)K * sizeof(G1Jac), hipMemcpyDeviceToHost, ctx->stream));
    }
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (degenerate) return PLONK_HIP_ERR_ZERO_DENOMINATOR;
    *adds_per_second = (double)lanes * K * (double)iters / (best * 1e-3);
    return PLONK_HIP_OK;
}

}  // namespace plonk

extern "C" int plonk_hip_bench_batch_affine(plonk_hip_ctx *ctx, int accumulators_per_lane, int steps, int invert,
                                            double *adds_per_second, uint64_t *first_block_out) {
    if (!ctx || !adds_per_second) return PLONK_HIP_ERR_INVALID_ARG;
    std::lock_guard<std::recursive_mutex> guard(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) return PLONK_HIP_ERR_HIP;
    return plonk::bench_batch_affine(ctx, accumulators_per_lane, steps, invert, adds_per_second, first_block_out);
}
