// Host-side entry points of the polynomial kernels (device pointers, context stream).
#pragma once
#include "context.hpp"

namespace plonk {

// s[j] = f[j] + u*s[j+1]; scratch must hold suffix_scan_scratch_elems(n) elements.
size_t suffix_scan_scratch_elems(size_t n);
int suffix_scan(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, const Fr &u, Fr *d_s, Fr *d_scratch);
// data[b][i] *= w^i for `batch` vectors of n elements stored back to back.
// the first n elements of `batch` vectors `stride` elements apart (0: n) are multiplied by w^i
int scale_by_powers(plonk_hip_ctx *ctx, Fr *d_data, size_t n, size_t batch, const Fr &w, size_t stride = 0);
// q[j] = sum_{m>=1} f[j + m*k], j < n - k.
int div_vanishing(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, size_t k, Fr *d_q);

// polyops.hip
// inclusive running product (or sum) in place; scratch must hold prefix_scan_scratch_elems(n) elements
size_t prefix_scan_scratch_elems(size_t n);
int prefix_scan(plonk_hip_ctx *ctx, Fr *d, size_t n, bool product, Fr *d_scratch);
// element-wise inverse in place; *d_flag (device int, cleared by the caller) becomes 1 on a zero element
int batch_inverse(plonk_hip_ctx *ctx, Fr *d, size_t n, int *d_flag);
int pointwise_mul(plonk_hip_ctx *ctx, const Fr *a, const Fr *b, Fr *out, size_t count);
// out[i] = a x[i] + b y[i] + (i == 0 || broadcast ? c : 0), i < max(nx, ny)
int linear_combination(plonk_hip_ctx *ctx, const Fr *x, size_t nx, const Fr *y, size_t ny, const Fr &a, const Fr &b,
                       const Fr &c, bool broadcast, Fr *out);
// *d_flag |= 1 when x[perm[i]] != x[i] for some i
int permutation_mismatch(plonk_hip_ctx *ctx, const Fr *x, const uint32_t *perm, size_t n, int *d_flag);
// *d_flag |= 1 when some x[i] != 0
int any_nonzero(plonk_hip_ctx *ctx, const Fr *x, size_t n, int *d_flag);
// out[j] = sum_t f[j + t*k], j < k
int fold_mod_vanishing(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, size_t k, Fr *d_out);
// out[i] = f(points[i])
int eval_points(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, const Fr *d_pts, size_t m, Fr *d_out);
// first `keep` coefficients of a*b
int poly_mul_dev(plonk_hip_ctx *ctx, const Fr *d_a, size_t na, const Fr *d_b, size_t nb, Fr *d_out, size_t keep);
// prod (X - roots[i]): m + 1 coefficients
int vanishing_from_roots_dev(plonk_hip_ctx *ctx, const Fr *d_roots, size_t m, Fr *d_z);
// floor(f / z), z monic of degree m: n - m coefficients
int poly_div_monic_dev(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, const Fr *d_z, size_t m, Fr *d_q);
// floor(f / prod (X - roots[i])); the roots are given both in device and in host memory
int div_by_roots_dev(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, const Fr *d_roots, const Fr *h_roots, size_t m,
                     Fr *d_q);
// interpolation at arbitrary points (quadratic work)
static constexpr size_t kMaxInterpolationPoints = (size_t)1 << 16;
int interpolate_dev(plonk_hip_ctx *ctx, const Fr *d_x, const Fr *d_y, size_t n, Fr *d_out, int *d_flag);
// interpolation at the points 1, q, q^2, ...: two products of length n (any n the transforms take)
int interpolate_geometric_dev(plonk_hip_ctx *ctx, const Fr &q, const Fr *d_y, size_t n, Fr *d_out, int *d_flag);

// ntt.hip
// butterflies only: forward = natural order in, bit-reversed order out, no scaling;
// inverse = bit-reversed in, natural out, scaled by 1/n
int ntt_butterflies(plonk_hip_ctx *ctx, Fr *d_data, uint32_t log_n, size_t batch, bool inverse);
int ntt_dev(plonk_hip_ctx *ctx, Fr *d_data, uint32_t log_n, size_t batch, bool inverse, const Fr *coset_shift);
int ntt_plan_describe(uint32_t log_n, bool natural, int max_passes, uint32_t *s0, uint32_t *s1);
// the same transform over Omega of size n = 2^k or 3 * 2^k
int ntt_mixed_dev(plonk_hip_ctx *ctx, Fr *d_data, size_t n, size_t batch, bool inverse);
int quotient_coset_dev(plonk_hip_ctx *ctx, Fr *d_f, uint32_t log_k, size_t batch, int *d_divisible);

}  // namespace plonk
