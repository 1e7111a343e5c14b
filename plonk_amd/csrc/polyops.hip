// Polynomial building blocks over Fr beyond the transform itself: running products and sums,
// batch inversion, evaluation at many points, products, the vanishing polynomial of a root set,
// division by it, and interpolation at arbitrary points.  All routines take device pointers and
// run on the context stream.
#include <cstring>

#include "context.hpp"
#include "poly.hpp"

namespace plonk {

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

static inline uint32_t ceil_log2(size_t n) {
    uint32_t l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}

// integer -> Montgomery form
static __device__ __forceinline__ Fr fr_from_u64_dev(uint64_t v) {
    Fr c = fe_zero<FrCfg>();
    c.l[0] = (uint32_t)v;
    c.l[1] = (uint32_t)(v >> 32);
    return fe_to_mont(c);
}

// ------------------------------------------------------------------------------------------
// inclusive running product / sum
// ------------------------------------------------------------------------------------------
struct OpMul {
    static __device__ __forceinline__ Fr identity() { return fe_one<FrCfg>(); }
    static __device__ __forceinline__ Fr apply(const Fr &a, const Fr &b) { return fe_mul(a, b); }
};
struct OpAdd {
    static __device__ __forceinline__ Fr identity() { return fe_zero<FrCfg>(); }
    static __device__ __forceinline__ Fr apply(const Fr &a, const Fr &b) { return fe_add(a, b); }
};

static constexpr int kPrefixThreads = 256, kPrefixPer = 8, kPrefixBlock = kPrefixThreads * kPrefixPer;

// Scan of one block of 2048 elements in place; the block total goes to block_totals.
template <class OP>
__global__ __launch_bounds__(kPrefixThreads) void k_prefix_block(Fr *__restrict__ data, size_t n,
                                                                Fr *__restrict__ block_totals) {
    __shared__ Fr sh[2][kPrefixThreads];
    const unsigned tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * kPrefixBlock + (size_t)tid * kPrefixPer;
    Fr x[kPrefixPer];
    Fr run = OP::identity();
#pragma unroll
    for (int c = 0; c < kPrefixPer; c++) {
        if (base + c < n) run = c ? OP::apply(run, data[base + c]) : data[base];
        x[c] = run;
    }
    sh[0][tid] = run;
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (unsigned off = 1; off < kPrefixThreads; off <<= 1) {
        Fr v = sh[cur][tid];
        if (tid >= off) v = OP::apply(sh[cur][tid - off], v);
        sh[cur ^ 1][tid] = v;
        __syncthreads();
        cur ^= 1;
    }
    if (tid) {
        const Fr before = sh[cur][tid - 1];
#pragma unroll
        for (int c = 0; c < kPrefixPer; c++) x[c] = OP::apply(before, x[c]);
    }
#pragma unroll
    for (int c = 0; c < kPrefixPer; c++)
        if (base + c < n) data[base + c] = x[c];
    if (tid == kPrefixThreads - 1) block_totals[blockIdx.x] = sh[cur][tid];
}

// elements of block b >= 1 take the scanned total of the blocks before them
template <class OP>
__global__ __launch_bounds__(256) void k_prefix_offsets(Fr *__restrict__ data, size_t n,
                                                       const Fr *__restrict__ scanned_totals) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x + kPrefixBlock;
    if (i >= n) return;
    data[i] = OP::apply(scanned_totals[i / kPrefixBlock - 1], data[i]);
}

size_t prefix_scan_scratch_elems(size_t n) {
    size_t total = 0;
    while (n > 1) {
        n = (n + kPrefixBlock - 1) / kPrefixBlock;
        total += n;
    }
    return total + 1;
}

template <class OP>
static int prefix_scan_impl(plonk_hip_ctx *ctx, Fr *d, size_t n, Fr *d_scratch) {
    if (n == 0) return PLONK_HIP_OK;
    size_t nblocks = (n + kPrefixBlock - 1) / kPrefixBlock;
    hipLaunchKernelGGL(k_prefix_block<OP>, dim3((unsigned)nblocks), dim3(kPrefixThreads), 0, ctx->stream, d, n,
                       d_scratch);
    if (nblocks > 1) {
        PLONK_TRY(prefix_scan_impl<OP>(ctx, d_scratch, nblocks, d_scratch + nblocks));
        hipLaunchKernelGGL(k_prefix_offsets<OP>, dim3(cdiv(n - kPrefixBlock, 256)), dim3(256), 0, ctx->stream, d, n,
                           d_scratch);
    }
    return PLONK_HIP_OK;
}

int prefix_scan(plonk_hip_ctx *ctx, Fr *d, size_t n, bool product, Fr *d_scratch) {
    return product ? prefix_scan_impl<OpMul>(ctx, d, n, d_scratch) : prefix_scan_impl<OpAdd>(ctx, d, n, d_scratch);
}

// ------------------------------------------------------------------------------------------
// batch inversion: one field inversion per lane, shared by the lane's CH elements
// ------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(128) void k_batch_inverse(Fr *__restrict__ data, size_t n, size_t nthreads,
                                                      int *__restrict__ zero_flag) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nthreads || t >= n) return;
    Fr p[CH];
    Fr run = data[t];
    p[0] = run;
#pragma unroll
    for (int c = 1; c < CH; c++) {
        const size_t idx = (size_t)c * nthreads + t;
        if (idx < n) run = fe_mul(run, data[idx]);
        p[c] = run;
    }
    if (fe_is_zero(run)) {
        *zero_flag = 1;
        return;
    }
    Fr inv = fe_inv(run);
#pragma unroll
    for (int c = CH - 1; c >= 1; c--) {
        const size_t idx = (size_t)c * nthreads + t;
        if (idx < n) {
            const Fr a = data[idx];
            data[idx] = fe_mul(inv, p[c - 1]);
            inv = fe_mul(inv, a);
        }
    }
    data[t] = inv;
}

// *d_flag (an int in device memory) is set to 1 when some element is zero; it must be cleared
// by the caller beforehand.
int batch_inverse(plonk_hip_ctx *ctx, Fr *d, size_t n, int *d_flag) {
    if (n == 0) return PLONK_HIP_OK;
    if (n >= ((size_t)1 << 16)) {
        size_t nthreads = (n + 15) / 16;
        hipLaunchKernelGGL(k_batch_inverse<16>, dim3(cdiv(nthreads, 128)), dim3(128), 0, ctx->stream, d, n, nthreads,
                           d_flag);
    } else {
        size_t nthreads = (n + 3) / 4;
        hipLaunchKernelGGL(k_batch_inverse<4>, dim3(cdiv(nthreads, 128)), dim3(128), 0, ctx->stream, d, n, nthreads,
                           d_flag);
    }
    return PLONK_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// element-wise helpers
// ------------------------------------------------------------------------------------------
// out may be a or b
__global__ __launch_bounds__(256) void k_pointwise_mul(const Fr *a, const Fr *b, Fr *out, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = fe_mul(a[i], b[i]);
}

int pointwise_mul(plonk_hip_ctx *ctx, const Fr *a, const Fr *b, Fr *out, size_t count) {
    if (count) hipLaunchKernelGGL(k_pointwise_mul, dim3(cdiv(count, 256)), dim3(256), 0, ctx->stream, a, b, out, count);
    return PLONK_HIP_OK;
}

// out[i] = a x[i] + b y[i] + (i == 0 || broadcast ? c : 0) for i < max(nx, ny); coefficients past
// the end of x or y read as zero.  out may be x or y.
__global__ __launch_bounds__(256) void k_linear_combination(const Fr *x, size_t nx, const Fr *y, size_t ny, Fr a, Fr b,
                                                            Fr c, int broadcast, Fr *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (nx > ny ? nx : ny)) return;
    Fr v = (i == 0 || broadcast) ? c : fe_zero<FrCfg>();
    if (i < nx) v = fe_add(v, fe_mul(a, x[i]));
    if (i < ny) v = fe_add(v, fe_mul(b, y[i]));
    out[i] = v;
}

int linear_combination(plonk_hip_ctx *ctx, const Fr *x, size_t nx, const Fr *y, size_t ny, const Fr &a, const Fr &b,
                       const Fr &c, bool broadcast, Fr *out) {
    const size_t n = nx > ny ? nx : ny;
    if (n)
        hipLaunchKernelGGL(k_linear_combination, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, x, nx, y, ny, a, b, c,
                           broadcast ? 1 : 0, out);
    return PLONK_HIP_OK;
}

__global__ void k_set_flag(int *flag) { *flag = 1; }

// *flag |= 1 when some element is not zero
__global__ __launch_bounds__(256) void k_any_nonzero(const Fr *__restrict__ x, size_t n, int *__restrict__ flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !fe_is_zero(x[i])) atomicOr(flag, 1);
}

int any_nonzero(plonk_hip_ctx *ctx, const Fr *x, size_t n, int *d_flag) {
    if (n) hipLaunchKernelGGL(k_any_nonzero, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, x, n, d_flag);
    return PLONK_HIP_OK;
}

// *flag |= 1 when x[perm[i]] != x[i] for some i (or perm[i] is out of range)
__global__ __launch_bounds__(256) void k_permutation_mismatch(const Fr *__restrict__ x, const uint32_t *__restrict__ perm,
                                                             size_t n, int *__restrict__ flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = perm[i];
    if (j >= n || !fe_eq(x[j], x[i])) atomicOr(flag, 1);
}

int permutation_mismatch(plonk_hip_ctx *ctx, const Fr *x, const uint32_t *perm, size_t n, int *d_flag) {
    if (n) hipLaunchKernelGGL(k_permutation_mismatch, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, x, perm, n, d_flag);
    return PLONK_HIP_OK;
}

// out[j] = sum_t f[j + t*k], j < k: the residue of f modulo X^k - 1 (same values on the k-th roots of unity)
__global__ __launch_bounds__(256) void k_fold(const Fr *__restrict__ f, size_t n, size_t k, Fr *__restrict__ out) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    Fr acc = fe_zero<FrCfg>();
    for (size_t i = j; i < n; i += k) acc = fe_add(acc, f[i]);
    out[j] = acc;
}

int fold_mod_vanishing(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, size_t k, Fr *d_out) {
    if (k == 0) return PLONK_HIP_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_fold, dim3(cdiv(k, 256)), dim3(256), 0, ctx->stream, d_f, n, k, d_out);
    return PLONK_HIP_OK;
}

// out[i] = f(points[i]) by Horner, one lane per point (every lane reads the same coefficient)
__global__ __launch_bounds__(128) void k_eval_points(const Fr *__restrict__ f, size_t n, const Fr *__restrict__ pts,
                                                    size_t m, Fr *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const Fr x = pts[i];
    Fr acc = fe_zero<FrCfg>();
    for (size_t j = n; j-- > 0;) acc = fe_add(fe_mul(acc, x), f[j]);
    out[i] = acc;
}

int eval_points(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, const Fr *d_pts, size_t m, Fr *d_out) {
    if (m) hipLaunchKernelGGL(k_eval_points, dim3(cdiv(m, 128)), dim3(128), 0, ctx->stream, d_f, n, d_pts, m, d_out);
    return PLONK_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// products
// ------------------------------------------------------------------------------------------
// out[j] = sum_i a[i] * b[j-i] for short a (na <= 32): one lane per output coefficient
__global__ __launch_bounds__(256) void k_mul_short(const Fr *__restrict__ a, size_t na, const Fr *__restrict__ b,
                                                  size_t nb, Fr *__restrict__ out, size_t keep) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= keep) return;
    Fr acc = fe_zero<FrCfg>();
    for (size_t i = 0; i < na && i <= j; i++) {
        if (j - i < nb) acc = fe_add(acc, fe_mul(a[i], b[j - i]));
    }
    out[j] = acc;
}

// dst[s][j] = j < len ? src[s * src_stride + j] : 0 for j < width, s < rows (zero-padded rows)
__global__ __launch_bounds__(256) void k_spread_rows(const Fr *__restrict__ src, size_t src_stride, size_t len,
                                                    Fr *__restrict__ dst, size_t width, size_t rows) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= width * rows) return;
    size_t s = e / width, j = e % width;
    dst[e] = j < len ? src[s * src_stride + j] : fe_zero<FrCfg>();
}

// First `keep` coefficients of a*b (keep <= na + nb - 1).  d_out must not overlap the inputs.
int poly_mul_dev(plonk_hip_ctx *ctx, const Fr *d_a, size_t na, const Fr *d_b, size_t nb, Fr *d_out, size_t keep) {
    if (na == 0 || nb == 0 || keep == 0) return PLONK_HIP_OK;
    if (keep > na + nb - 1) return PLONK_HIP_ERR_INVALID_ARG;
    if (na > nb) {
        const Fr *t = d_a;
        d_a = d_b;
        d_b = t;
        size_t tn = na;
        na = nb;
        nb = tn;
    }
    if (na <= 32) {
        hipLaunchKernelGGL(k_mul_short, dim3(cdiv(keep, 256)), dim3(256), 0, ctx->stream, d_a, na, d_b, nb, d_out, keep);
        return PLONK_HIP_OK;
    }
    const uint32_t log_n = ceil_log2(na + nb - 1);
    const size_t N = (size_t)1 << log_n;
    PLONK_TRY(buf_reserve(ctx, ctx->mul_ws, 2 * N * sizeof(Fr)));
    Fr *w = (Fr *)ctx->mul_ws.ptr;
    hipLaunchKernelGGL(k_spread_rows, dim3(cdiv(N, 256)), dim3(256), 0, ctx->stream, d_a, (size_t)0, na, w, N,
                       (size_t)1);
    hipLaunchKernelGGL(k_spread_rows, dim3(cdiv(N, 256)), dim3(256), 0, ctx->stream, d_b, (size_t)0, nb, w + N, N,
                       (size_t)1);
    // values in bit-reversed order are fine for an element-wise product
    PLONK_TRY(ntt_butterflies(ctx, w, log_n, 2, false));
    PLONK_TRY(pointwise_mul(ctx, w, w + N, w, N));
    PLONK_TRY(ntt_butterflies(ctx, w, log_n, 1, true));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_out, w, keep * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
    return PLONK_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// vanishing polynomial of a root set: prod (X - rho_i)
// ------------------------------------------------------------------------------------------
static constexpr int kLeaf = 256;

// One block multiplies out the kLeaf linear factors of a leaf (roots beyond `m` count as 0, i.e.
// as factors X).  Output: the kLeaf low coefficients; the leading coefficient 1 is implicit.
__global__ __launch_bounds__(kLeaf) void k_roots_leaf(const Fr *__restrict__ roots, size_t m, Fr *__restrict__ out) {
    __shared__ Fr sh[2][kLeaf];
    const unsigned j = threadIdx.x;
    const size_t first = (size_t)blockIdx.x * kLeaf;
    sh[0][j] = j == 0 ? fe_one<FrCfg>() : fe_zero<FrCfg>();
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (unsigned i = 0; i < kLeaf; i++) {
        // degree i -> i + 1: new[j] = old[j-1] - rho * old[j]
        const bool real = first + i < m;
        Fr rho;
        if (real) rho = roots[first + i];
        Fr lower = j ? sh[cur][j - 1] : fe_zero<FrCfg>();
        Fr v = real ? fe_sub(lower, fe_mul(rho, sh[cur][j])) : lower;
        sh[cur ^ 1][j] = v;  // the coefficient of X^kLeaf (the leading 1 after the last factor) is dropped
        __syncthreads();
        cur ^= 1;
    }
    out[first + j] = sh[cur][j];
}

// prod[p][j] = A[2p][j] * A[2p+1][j] for rows of `width` elements
__global__ __launch_bounds__(256) void k_tree_pointwise(const Fr *__restrict__ A, size_t width, size_t pairs,
                                                       Fr *__restrict__ prod) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= width * pairs) return;
    size_t p = e / width, j = e % width;
    prod[e] = fe_mul(A[2 * p * width + j], A[(2 * p + 1) * width + j]);
}

// next[p][j] = prod[p][j] + (j >= d ? cur[2p][j-d] + cur[2p+1][j-d] : 0), j < 2d:
// (X^d + a)(X^d + b) = X^2d + X^d (a + b) + a*b
__global__ __launch_bounds__(256) void k_tree_combine(const Fr *__restrict__ prod, const Fr *__restrict__ cur, size_t d,
                                                     size_t pairs, Fr *__restrict__ next) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * d * pairs) return;
    size_t p = e / (2 * d), j = e % (2 * d);
    Fr v = prod[e];
    if (j >= d) v = fe_add(v, fe_add(cur[(2 * p) * d + j - d], cur[(2 * p + 1) * d + j - d]));
    next[e] = v;
}

// z[j] = j < m ? low[j + pad] : 1   (strip the pad factors X, restore the leading coefficient)
__global__ __launch_bounds__(256) void k_strip_pad(const Fr *__restrict__ low, size_t pad, size_t m, Fr *__restrict__ z) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > m) return;
    z[j] = j < m ? low[j + pad] : fe_one<FrCfg>();
}

// d_z receives the m + 1 coefficients of prod_{i<m} (X - roots[i]) (`construct_vanishing_polynomial_from_roots`,
// common/utils.rs:51-62 of the reference).  Leaves of 256 roots are multiplied out directly, then
// pairs of equal degree are multiplied level by level with batched transforms.
int vanishing_from_roots_dev(plonk_hip_ctx *ctx, const Fr *d_roots, size_t m, Fr *d_z) {
    hipStream_t st = ctx->stream;
    size_t leaves = 1;
    while (leaves * kLeaf < m) leaves *= 2;
    const size_t M = leaves * kLeaf, pad = M - m;
    // two ping-pong arrays of M low coefficients, the transform area (2M) and the products (M)
    PLONK_TRY(buf_reserve(ctx, ctx->tree_ws, 5 * M * sizeof(Fr)));
    Fr *cur = (Fr *)ctx->tree_ws.ptr, *next = cur + M, *A = next + M, *P = A + 2 * M;
    hipLaunchKernelGGL(k_roots_leaf, dim3((unsigned)leaves), dim3(kLeaf), 0, st, d_roots, m, cur);
    for (size_t d = kLeaf, cnt = leaves; cnt > 1; d *= 2, cnt /= 2) {
        const uint32_t log_w = ceil_log2(2 * d);
        const size_t pairs = cnt / 2;
        hipLaunchKernelGGL(k_spread_rows, dim3(cdiv(2 * d * cnt, 256)), dim3(256), 0, st, cur, d, d, A, 2 * d, cnt);
        PLONK_TRY(ntt_butterflies(ctx, A, log_w, cnt, false));
        hipLaunchKernelGGL(k_tree_pointwise, dim3(cdiv(2 * d * pairs, 256)), dim3(256), 0, st, A, 2 * d, pairs, P);
        PLONK_TRY(ntt_butterflies(ctx, P, log_w, pairs, true));
        hipLaunchKernelGGL(k_tree_combine, dim3(cdiv(2 * d * pairs, 256)), dim3(256), 0, st, P, cur, d, pairs, next);
        Fr *t = cur;
        cur = next;
        next = t;
    }
    hipLaunchKernelGGL(k_strip_pad, dim3(cdiv(m + 1, 256)), dim3(256), 0, st, cur, pad, m, d_z);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// quotient by a monic polynomial (remainder dropped), and by a root set
// ------------------------------------------------------------------------------------------
// h[i] = i <= m ? z[m - i] : 0 for i < len (the reversed divisor, truncated or zero-extended)
__global__ __launch_bounds__(256) void k_reverse_into(const Fr *__restrict__ src, size_t top, Fr *__restrict__ dst,
                                                     size_t len) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    dst[i] = i <= top ? src[top - i] : fe_zero<FrCfg>();
}
// e = 2 - e
__global__ __launch_bounds__(256) void k_two_minus(Fr *__restrict__ e, size_t len) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    Fr v = fe_neg(e[i]);
    if (i == 0) v = fe_add(v, fe_dbl(fe_one<FrCfg>()));
    e[i] = v;
}
__global__ void k_set_one(Fr *p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *p = fe_one<FrCfg>();
}

// q = floor(f / z) for monic z of degree m < n: reversal turns it into a truncated power-series
// division, rev(q) = rev(f) * rev(z)^-1 mod X^(n-m), with the inverse series from Newton's
// iteration g <- g * (2 - h g).  d_q receives n - m coefficients and must not overlap f or z.
int poly_div_monic_dev(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, const Fr *d_z, size_t m, Fr *d_q) {
    if (n <= m) return PLONK_HIP_OK;
    hipStream_t st = ctx->stream;
    const size_t L = n - m;
    size_t Lp = 1;
    while (Lp < L) Lp *= 2;
    PLONK_TRY(buf_reserve(ctx, ctx->div_ws, 5 * Lp * sizeof(Fr)));
    Fr *h = (Fr *)ctx->div_ws.ptr, *g = h + Lp, *e = g + Lp, *gn = e + Lp, *rf = gn + Lp;
    hipLaunchKernelGGL(k_reverse_into, dim3(cdiv(Lp, 256)), dim3(256), 0, st, d_z, m, h, Lp);
    PLONK_HIP_TRY(ctx, hipMemsetAsync(g, 0, Lp * sizeof(Fr), st));
    hipLaunchKernelGGL(k_set_one, dim3(1), dim3(1), 0, st, g);
    for (size_t t = 1; t < L; t *= 2) {
        const size_t t2 = 2 * t;  // <= Lp
        PLONK_TRY(poly_mul_dev(ctx, h, t2, g, t, e, t2));
        hipLaunchKernelGGL(k_two_minus, dim3(cdiv(t2, 256)), dim3(256), 0, st, e, t2);
        PLONK_TRY(poly_mul_dev(ctx, g, t, e, t2, gn, t2));
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(g, gn, t2 * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    }
    hipLaunchKernelGGL(k_reverse_into, dim3(cdiv(L, 256)), dim3(256), 0, st, d_f, n - 1, rf, L);
    PLONK_TRY(poly_mul_dev(ctx, rf, L, g, L, e, L));
    hipLaunchKernelGGL(k_reverse_into, dim3(cdiv(L, 256)), dim3(256), 0, st, e, L - 1, d_q, L);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

// q = floor(f / prod (X - roots[i])) (`compute_q_zero_test_from_roots`, common/protocols.rs:169-181 of
// the reference).  Few roots: one synthetic division per root (dividing by the factors one after
// the other gives the same quotient as dividing by their product).  Many roots: product tree, then
// the series division above.  d_q receives n - m coefficients.
int div_by_roots_dev(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, const Fr *d_roots, const Fr *h_roots, size_t m,
                     Fr *d_q) {
    if (n <= m) return PLONK_HIP_OK;
    hipStream_t st = ctx->stream;
    if (m <= 24) {
        const size_t scr = suffix_scan_scratch_elems(n);
        PLONK_TRY(buf_reserve(ctx, ctx->div_ws, (2 * n + scr) * sizeof(Fr)));
        Fr *bufs[2] = {(Fr *)ctx->div_ws.ptr, (Fr *)ctx->div_ws.ptr + n};
        Fr *ws = bufs[1] + n;
        const Fr *src = d_f;
        size_t len = n;
        for (size_t i = 0; i < m; i++) {
            Fr *dst = bufs[i & 1];
            PLONK_TRY(suffix_scan(ctx, src, len, h_roots[i], dst, ws));
            src = dst + 1;  // s[1..len) is the quotient by X - root, s[0] the dropped remainder
            len--;
        }
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_q, src, len * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        return PLONK_HIP_OK;
    }
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_a, (m + 1) * sizeof(Fr)));
    Fr *z = (Fr *)ctx->scratch_a.ptr;
    PLONK_TRY(vanishing_from_roots_dev(ctx, d_roots, m, z));
    return poly_div_monic_dev(ctx, d_f, n, z, m, d_q);
}

// ------------------------------------------------------------------------------------------
// interpolation at arbitrary points (Lagrange form, quadratic work)
// ------------------------------------------------------------------------------------------
// deriv[j] = (j + 1) * z[j + 1]
__global__ __launch_bounds__(256) void k_derivative(const Fr *__restrict__ z, size_t n, Fr *__restrict__ deriv) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    deriv[j] = fe_mul(fr_from_u64_dev(j + 1), z[j + 1]);
}

// Row w of `partial` = sum over the 64 points of wave w of c_i * M(X) / (X - x_i): lane i carries
// the synthetic-division value h_j = M[j+1] + x_i h_{j+1} downwards from j = n-1, and the wave adds
// the 64 terms c_i h_j with shuffles.
__global__ __launch_bounds__(64) void k_lagrange_rows(const Fr *__restrict__ M, size_t n, const Fr *__restrict__ xs,
                                                     const Fr *__restrict__ cs, Fr *__restrict__ partial) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    const bool live = i < n;
    Fr x = fe_zero<FrCfg>(), c = fe_zero<FrCfg>();
    if (live) {
        x = xs[i];
        c = cs[i];
    }
    Fr h = fe_zero<FrCfg>();
    Fr *row = partial + (size_t)blockIdx.x * n;
#pragma unroll 1
    for (size_t j = n; j-- > 0;) {
        h = fe_add(fe_mul(h, x), M[j + 1]);
        Fr term = fe_mul(c, h);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            Fr o;
#pragma unroll
            for (int l = 0; l < 8; l++) o.l[l] = __shfl_xor(term.l[l], off, 64);
            term = fe_add(term, o);
        }
        if (threadIdx.x == 0) row[j] = term;
    }
}
// out[j] = sum_w partial[w][j]
__global__ __launch_bounds__(256) void k_sum_rows(const Fr *__restrict__ partial, size_t rows, size_t n,
                                                 Fr *__restrict__ out) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    Fr acc = partial[j];
    for (size_t w = 1; w < rows; w++) acc = fe_add(acc, partial[w * n + j]);
    out[j] = acc;
}

// The n coefficients of the polynomial of degree < n through (x_i, y_i) (`interpolate_polynomial`,
// common/polynomials.rs:14-58 of the reference).  *d_flag is set when two x coincide (the reference
// panics on the zero denominator, polynomials.rs:42).
int interpolate_dev(plonk_hip_ctx *ctx, const Fr *d_x, const Fr *d_y, size_t n, Fr *d_out, int *d_flag) {
    if (n == 0) return PLONK_HIP_OK;
    if (n > kMaxInterpolationPoints) return PLONK_HIP_ERR_INVALID_ARG;
    hipStream_t st = ctx->stream;
    const size_t rows = (n + 63) / 64;
    PLONK_TRY(buf_reserve(ctx, ctx->interp_ws, (3 * n + 2 + rows * n) * sizeof(Fr)));
    Fr *M = (Fr *)ctx->interp_ws.ptr, *deriv = M + n + 1, *c = deriv + n, *partial = c + n;
    PLONK_TRY(vanishing_from_roots_dev(ctx, d_x, n, M));
    hipLaunchKernelGGL(k_derivative, dim3(cdiv(n, 256)), dim3(256), 0, st, M, n, deriv);
    // M'(x_i) = prod_{j != i} (x_i - x_j)
    PLONK_TRY(eval_points(ctx, deriv, n, d_x, n, c));
    PLONK_TRY(batch_inverse(ctx, c, n, d_flag));
    PLONK_TRY(pointwise_mul(ctx, c, d_y, c, n));
    hipLaunchKernelGGL(k_lagrange_rows, dim3((unsigned)rows), dim3(64), 0, st, M, n, d_x, c, partial);
    hipLaunchKernelGGL(k_sum_rows, dim3(cdiv(n, 256)), dim3(256), 0, st, partial, rows, n, d_out);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// interpolation at the points 1, q, q^2, ..., q^(n-1)  (O(n log n))
// ------------------------------------------------------------------------------------------
// The gate rows of a circuit are such a set (Omega[3 l] = (w^3)^l, setup_proving_key/mod.rs:36-43), and it is a
// whole subgroup -- where one inverse transform interpolates -- only for one circuit size in each family.  On a
// geometric progression the Newton basis N_i = prod_{j<i} (X - q^j) has closed forms (the q-binomial theorem):
// with u_i = prod_{k=1..i} (q^k - 1), e_i = q^(i(i-1)/2) and g_i = (-1)^i e_i / u_i,
//     values -> Newton coefficients:   c_i e_i = [X^i]  (sum_j v_j / u_j X^j) * (sum_k g_k X^k)
//     Newton -> monomial coefficients: F_k u_k = sum_{j>=0} (c_(k+j) u_(k+j)) g_j
// i.e. two products of length n, three running products and one batched inversion.
//
// a[0] = 1, a[i] = v for i >= 1 (the running product is then v^i)
__global__ __launch_bounds__(256) void k_geo_fill(Fr *__restrict__ a, size_t n, Fr v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = i ? v : fe_one<FrCfg>();
}
// from pw[i] = q^i and pwi[i] = q^-i: the factors whose running products are u, e and 1/e
__global__ __launch_bounds__(256) void k_geo_factors(const Fr *__restrict__ pw, const Fr *__restrict__ pwi, size_t n,
                                                    Fr *__restrict__ u, Fr *__restrict__ e, Fr *__restrict__ einv) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr one = fe_one<FrCfg>();
    u[i] = i ? fe_sub(pw[i], one) : one;
    e[i] = i ? pw[i - 1] : one;
    einv[i] = i ? pwi[i - 1] : one;
}
// g[i] = (-1)^i e[i] uinv[i];  w[i] = y[i] uinv[i]
__global__ __launch_bounds__(256) void k_geo_series(const Fr *__restrict__ e, const Fr *__restrict__ uinv,
                                                   const Fr *__restrict__ y, size_t n, Fr *__restrict__ g,
                                                   Fr *__restrict__ w) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr t = fe_mul(e[i], uinv[i]);
    g[i] = (i & 1) ? fe_neg(t) : t;
    w[i] = fe_mul(y[i], uinv[i]);
}
// a[n - 1 - i] = p[i] u[i] einv[i]   (c_i u_i, highest index first)
__global__ __launch_bounds__(256) void k_geo_newton(const Fr *__restrict__ p, const Fr *__restrict__ u,
                                                   const Fr *__restrict__ einv, size_t n, Fr *__restrict__ a) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[n - 1 - i] = fe_mul(p[i], fe_mul(u[i], einv[i]));
}
// out[k] = c[n - 1 - k] uinv[k]
__global__ __launch_bounds__(256) void k_geo_finish(const Fr *__restrict__ c, const Fr *__restrict__ uinv, size_t n,
                                                   Fr *__restrict__ out) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = fe_mul(c[n - 1 - k], uinv[k]);
}

// The n coefficients of the polynomial of degree < n with F(q^i) = y[i], i < n (`interpolate_polynomial` of the
// reference, common/polynomials.rs:50-58, for x_points = [1, q, q^2, ...]).  *d_flag is set when two of the
// points coincide (q^k = 1 for some 0 < k < n; the reference panics on the zero denominator, polynomials.rs:42).
// d_out may be d_y.
int interpolate_geometric_dev(plonk_hip_ctx *ctx, const Fr &q, const Fr *d_y, size_t n, Fr *d_out, int *d_flag) {
    if (n == 0) return PLONK_HIP_OK;
    if (n > ((size_t)1 << 28)) return PLONK_HIP_ERR_INVALID_ARG;
    hipStream_t st = ctx->stream;
    if (fe_is_zero(q)) {
        // the points 1, 0, 0, ...: distinct only for n <= 2
        if (n > 2) {
            hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(1), 0, st, d_flag);
            return PLONK_HIP_OK;
        }
    }
    const size_t scr = prefix_scan_scratch_elems(n);
    PLONK_TRY(buf_reserve(ctx, ctx->interp_ws, (7 * n + scr) * sizeof(Fr)));
    Fr *x0 = (Fr *)ctx->interp_ws.ptr, *x1 = x0 + n, *u = x1 + n, *e = u + n, *einv = e + n, *x5 = einv + n,
       *x6 = x5 + n, *scratch = x6 + n;
    const unsigned blocks = cdiv(n, 256);
    const Fr qinv = fe_is_zero(q) ? q : fe_inv(q);
    hipLaunchKernelGGL(k_geo_fill, dim3(blocks), dim3(256), 0, st, x0, n, q);
    hipLaunchKernelGGL(k_geo_fill, dim3(blocks), dim3(256), 0, st, x1, n, qinv);
    PLONK_TRY(prefix_scan(ctx, x0, n, true, scratch));  // q^i
    PLONK_TRY(prefix_scan(ctx, x1, n, true, scratch));  // q^-i
    hipLaunchKernelGGL(k_geo_factors, dim3(blocks), dim3(256), 0, st, x0, x1, n, u, e, einv);
    PLONK_TRY(prefix_scan(ctx, u, n, true, scratch));
    PLONK_TRY(prefix_scan(ctx, e, n, true, scratch));
    PLONK_TRY(prefix_scan(ctx, einv, n, true, scratch));
    Fr *uinv = x0, *g = x1;
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(uinv, u, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    PLONK_TRY(batch_inverse(ctx, uinv, n, d_flag));  // a zero u_i: q^k = 1 for some 0 < k <= i
    hipLaunchKernelGGL(k_geo_series, dim3(blocks), dim3(256), 0, st, e, uinv, d_y, n, g, x5);
    PLONK_TRY(poly_mul_dev(ctx, x5, n, g, n, x6, n));
    hipLaunchKernelGGL(k_geo_newton, dim3(blocks), dim3(256), 0, st, x6, u, einv, n, x5);
    PLONK_TRY(poly_mul_dev(ctx, x5, n, g, n, x6, n));
    hipLaunchKernelGGL(k_geo_finish, dim3(blocks), dim3(256), 0, st, x6, uinv, n, d_out);
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

}  // namespace plonk
