// Polynomial kernels over Fr: suffix Horner scan (evaluation at a point and division by
// X - u in one pass) and exact division by X^k - 1.
#include "context.hpp"
#include "poly.hpp"

namespace plonk {

static constexpr int kScanChunk = 64;
static constexpr int kScanSerial = 256;

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------
// Suffix Horner scan: s[j] = f[j] + u * s[j+1], s[n] = 0.
//   s[0] = f(u)                              (`f.evaluate(&u)`)
//   s[1..n) = coefficients of (f - f(u)) / (X - u)   (synthetic division)
// ------------------------------------------------------------------------------------------
__global__ void k_scan_serial(const Fr *__restrict__ f, size_t n, Fr u, Fr *__restrict__ s) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Fr acc = fe_zero<FrCfg>();
    for (size_t j = n; j-- > 0;) {
        acc = fe_add(fe_mul(acc, u), f[j]);
        s[j] = acc;
    }
}

// H[t] = Horner value of chunk t taken alone (zero carry-in).
__global__ __launch_bounds__(128) void k_scan_chunk_totals(const Fr *__restrict__ f, size_t n, Fr u,
                                                          Fr *__restrict__ H, size_t nchunks) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    size_t lo = t * kScanChunk;
    size_t hi = lo + kScanChunk < n ? lo + kScanChunk : n;
    Fr acc = fe_zero<FrCfg>();
    for (size_t j = hi; j-- > lo;) acc = fe_add(fe_mul(acc, u), f[j]);
    H[t] = acc;
}

// s over chunk t with carry-in C[t+1] (= s at the chunk's upper border).
__global__ __launch_bounds__(128) void k_scan_apply(const Fr *__restrict__ f, size_t n, Fr u,
                                                   const Fr *__restrict__ C, Fr *__restrict__ s, size_t nchunks) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    size_t lo = t * kScanChunk;
    size_t hi = lo + kScanChunk < n ? lo + kScanChunk : n;
    Fr acc = t + 1 < nchunks ? C[t + 1] : fe_zero<FrCfg>();
    for (size_t j = hi; j-- > lo;) {
        acc = fe_add(fe_mul(acc, u), f[j]);
        s[j] = acc;
    }
}

// Scratch layout for the recursion: level arrays H and C packed back to back.
size_t suffix_scan_scratch_elems(size_t n) {
    size_t total = 0;
    while (n > (size_t)kScanSerial) {
        n = (n + kScanChunk - 1) / kScanChunk;
        total += 2 * n;
    }
    return total + 1;
}

int suffix_scan(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, const Fr &u, Fr *d_s, Fr *d_scratch) {
    if (n == 0) return PLONK_HIP_OK;
    hipStream_t st = ctx->stream;
    if (n <= (size_t)kScanSerial) {
        hipLaunchKernelGGL(k_scan_serial, dim3(1), dim3(1), 0, st, d_f, n, u, d_s);
        return PLONK_HIP_OK;
    }
    size_t nchunks = (n + kScanChunk - 1) / kScanChunk;
    Fr *H = d_scratch, *C = d_scratch + nchunks;
    hipLaunchKernelGGL(k_scan_chunk_totals, dim3(cdiv(nchunks, 128)), dim3(128), 0, st, d_f, n, u, H, nchunks);
    Fr U = u;
    for (int k = 0; k < 6; k++) U = fe_sqr(U);  // u^64
    PLONK_TRY(suffix_scan(ctx, H, nchunks, U, C, d_scratch + 2 * nchunks));
    hipLaunchKernelGGL(k_scan_apply, dim3(cdiv(nchunks, 128)), dim3(128), 0, st, d_f, n, u, C, d_s, nchunks);
    return PLONK_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// Quotient of f by X^k - 1 exactly as long division produces it (remainder dropped):
//   q[j] = f[j+k] + f[j+2k] + f[j+3k] + ...        for j = 0 .. n-k-1
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_div_vanishing(const Fr *__restrict__ f, size_t n, size_t k,
                                                      Fr *__restrict__ q) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j + k >= n) return;
    Fr acc = f[j + k];
    for (size_t i = j + 2 * k; i < n; i += k) acc = fe_add(acc, f[i]);
    q[j] = acc;
}

int div_vanishing(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, size_t k, Fr *d_q) {
    if (k == 0) return PLONK_HIP_ERR_INVALID_ARG;
    if (n <= k) return PLONK_HIP_OK;
    hipLaunchKernelGGL(k_div_vanishing, dim3(cdiv(n - k, 256)), dim3(256), 0, ctx->stream, d_f, n, k, d_q);
    return PLONK_HIP_OK;
}

}  // namespace plonk
