// Host-side entry points of the polynomial kernels (device pointers, context stream).
#pragma once
#include "context.hpp"

namespace plonk {

// s[j] = f[j] + u*s[j+1]; scratch must hold suffix_scan_scratch_elems(n) elements.
size_t suffix_scan_scratch_elems(size_t n);
int suffix_scan(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, const Fr &u, Fr *d_s, Fr *d_scratch);
// data[b][i] *= w^i for `batch` vectors of n elements stored back to back.
int scale_by_powers(plonk_hip_ctx *ctx, Fr *d_data, size_t n, size_t batch, const Fr &w);
// q[j] = sum_{m>=1} f[j + m*k], j < n - k.
int div_vanishing(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, size_t k, Fr *d_q);

// ntt.hip
int ntt_dev(plonk_hip_ctx *ctx, Fr *d_data, uint32_t log_n, size_t batch, bool inverse, const Fr *coset_shift);
int quotient_coset_dev(plonk_hip_ctx *ctx, Fr *d_f, uint32_t log_k, size_t batch, int *d_divisible);

}  // namespace plonk
