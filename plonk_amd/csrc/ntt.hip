// Fr number-theoretic transform over Omega = <omega>, omega = 7^((r-1)/n), and the coset
// quotient by Z_Omega(X) = X^k - 1.
//
// A transform of 2^log_n points runs as ceil(log_n / 10) passes.  A pass owns the butterfly
// stages [s0, s1): one workgroup loads a tile of 2^(s1-s0) elements
//     i(j) = (t_hi << s1) | (j << s0) | t_lo,   j = 0 .. 2^(s1-s0) - 1
// into LDS, runs its stages there and writes the tile back, so HBM is touched once per pass.
// Two butterfly orders share the tiling:
//   DIF  natural order in  -> bit-reversed order out   (stages high to low)
//   DIT  bit-reversed in   -> natural order out        (stages low to high)
// The public transform is natural order on both sides (DIF + bit-reversal, or bit-reversal
// + DIT); the quotient pipeline chains DIF -> pointwise -> DIT and never permutes.
// Twiddles omega^e, e < n/2, are precomputed per size and direction and cached in the context.
#include "context.hpp"
#include "poly.hpp"

namespace plonk {

static constexpr int kMaxTileLog = 10;  // 1024 elements = 32 KiB of LDS
static constexpr int kNttBlock = 256;

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------
// host-side field helpers (same __host__ __device__ arithmetic as the kernels)
// ------------------------------------------------------------------------------------------
static Fr fr_from_u64(uint64_t v) {
    Fr c = fe_zero<FrCfg>();
    c.l[0] = (uint32_t)v;
    c.l[1] = (uint32_t)(v >> 32);
    return fe_to_mont(c);
}

// omega_n = 7^((r-1) >> log_n)   (common/utils.rs:11-38 of the reference for k = 2^log_n)
static Fr fr_root_of_unity(uint32_t log_n) {
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = FrCfg::mod(i);
    e[0] -= 1;  // r - 1 (r is odd, no borrow)
    // shift right by log_n (< 32)
    if (log_n) {
        for (int i = 0; i < 8; i++) e[i] = (e[i] >> log_n) | (i + 1 < 8 ? e[i + 1] << (32 - log_n) : 0u);
    }
    Fr base = fr_from_u64(7), acc = fe_one<FrCfg>();
    for (int i = 7; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            acc = fe_sqr(acc);
            if ((e[i] >> b) & 1) acc = fe_mul(acc, base);
        }
    return acc;
}

// ------------------------------------------------------------------------------------------
// power tables
// ------------------------------------------------------------------------------------------
struct PowSeed {
    Fr pw[32];  // base^(2^k)
};
// out[i] = base^i, i < count (square and multiply per entry; tables are small)
__global__ __launch_bounds__(256) void k_pow_table(PowSeed seed, uint32_t count, Fr *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fr acc = fe_one<FrCfg>();
#pragma unroll 1
    for (int k = 0; k < 32; k++) {
        if ((i >> k) & 1) acc = fe_mul(acc, seed.pw[k]);
    }
    out[i] = acc;
}
// out[e] = lo[e & 1023] * hi[e >> 10]
__global__ __launch_bounds__(256) void k_pow_combine(const Fr *__restrict__ lo, const Fr *__restrict__ hi, size_t count,
                                                    Fr *__restrict__ out) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    Fr v = lo[e & 1023];
    if (e >> 10) v = fe_mul(v, hi[e >> 10]);
    out[e] = v;
}
// data[b][i] *= lo[i & 1023] * hi[i >> 10]
__global__ __launch_bounds__(256) void k_scale_two_level(Fr *__restrict__ data, size_t n, size_t batch_stride,
                                                        const Fr *__restrict__ lo, const Fr *__restrict__ hi) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr v = lo[i & 1023];
    if (i >> 10) v = fe_mul(v, hi[i >> 10]);
    Fr *p = data + (size_t)blockIdx.y * batch_stride + i;
    *p = fe_mul(*p, v);
}

static PowSeed make_seed(const Fr &base) {
    PowSeed s;
    s.pw[0] = base;
    for (int k = 1; k < 32; k++) s.pw[k] = fe_sqr(s.pw[k - 1]);
    return s;
}

// lo = base^i (i < 1024), hi = (base^1024)^i (i < ceil(count / 1024)) in ctx->scratch_b.
static int build_two_level(plonk_hip_ctx *ctx, const Fr &base, size_t count, const Fr **lo, const Fr **hi) {
    size_t nhi = (count + 1023) / 1024;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_b, (1024 + nhi) * sizeof(Fr)));
    Fr *d_lo = (Fr *)ctx->scratch_b.ptr, *d_hi = d_lo + 1024;
    PowSeed s = make_seed(base);
    hipLaunchKernelGGL(k_pow_table, dim3(4), dim3(256), 0, ctx->stream, s, 1024u, d_lo);
    PowSeed sh = make_seed(s.pw[10]);
    hipLaunchKernelGGL(k_pow_table, dim3(cdiv(nhi, 256)), dim3(256), 0, ctx->stream, sh, (uint32_t)nhi, d_hi);
    *lo = d_lo;
    *hi = d_hi;
    return PLONK_HIP_OK;
}

int scale_by_powers(plonk_hip_ctx *ctx, Fr *d_data, size_t n, size_t batch, const Fr &w) {
    if (n == 0 || batch == 0) return PLONK_HIP_OK;
    if (n > 0xffffffffull || batch > 65535) return PLONK_HIP_ERR_INVALID_ARG;
    const Fr *lo, *hi;
    PLONK_TRY(build_two_level(ctx, w, n, &lo, &hi));
    hipLaunchKernelGGL(k_scale_two_level, dim3(cdiv(n, 256), (unsigned)batch), dim3(256), 0, ctx->stream, d_data, n, n,
                       lo, hi);
    return PLONK_HIP_OK;
}

static int get_twiddles(plonk_hip_ctx *ctx, uint32_t log_n, bool inverse, const Fr **out) {
    DevBuf &b = ctx->twiddle[inverse ? 1 : 0][log_n];
    size_t count = log_n ? ((size_t)1 << (log_n - 1)) : 1;
    if (!b.ptr) {
        Fr w = fr_root_of_unity(log_n);
        if (inverse) w = fe_inv(w);
        const Fr *lo, *hi;
        PLONK_TRY(build_two_level(ctx, w, count, &lo, &hi));
        PLONK_TRY(buf_reserve(ctx, b, count * sizeof(Fr)));
        hipLaunchKernelGGL(k_pow_combine, dim3(cdiv(count, 256)), dim3(256), 0, ctx->stream, lo, hi, count,
                           (Fr *)b.ptr);
    }
    *out = (const Fr *)b.ptr;
    return PLONK_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// butterfly passes
// ------------------------------------------------------------------------------------------
template <bool DIF>
__global__ __launch_bounds__(kNttBlock) void k_ntt_pass(Fr *__restrict__ data, uint32_t log_n, uint32_t s0, uint32_t s1,
                                                       const Fr *__restrict__ tw, size_t batch_stride, Fr scale,
                                                       int do_scale) {
    extern __shared__ __align__(16) unsigned char smem[];
    Fr *tile = reinterpret_cast<Fr *>(smem);
    const uint32_t k = s1 - s0;
    const uint32_t tile_n = 1u << k;
    const size_t T = blockIdx.x;
    const size_t t_lo = T & (((size_t)1 << s0) - 1);
    const size_t t_hi = T >> s0;
    Fr *base = data + (size_t)blockIdx.y * batch_stride + (t_hi << s1) + t_lo;

    for (uint32_t j = threadIdx.x; j < tile_n; j += kNttBlock) tile[j] = base[(size_t)j << s0];
    __syncthreads();

    for (uint32_t step = 0; step < k; step++) {
        const uint32_t bl = DIF ? (k - 1 - step) : step;  // in-tile bit of this stage
        const uint32_t st = s0 + bl;                      // global stage: half-span 2^st
        const uint32_t low_mask = (1u << bl) - 1;
        for (uint32_t b = threadIdx.x; b < (tile_n >> 1); b += kNttBlock) {
            uint32_t jlo = ((b >> bl) << (bl + 1)) | (b & low_mask);
            uint32_t jhi = jlo + (1u << bl);
            size_t e = ((((size_t)(jlo & low_mask)) << s0) | t_lo) << (log_n - 1 - st);
            Fr w = tw[e];
            Fr x = tile[jlo], y = tile[jhi];
            if (DIF) {
                tile[jlo] = fe_add(x, y);
                tile[jhi] = fe_mul(fe_sub(x, y), w);
            } else {
                y = fe_mul(y, w);
                tile[jlo] = fe_add(x, y);
                tile[jhi] = fe_sub(x, y);
            }
        }
        __syncthreads();
    }

    for (uint32_t j = threadIdx.x; j < tile_n; j += kNttBlock) {
        Fr v = tile[j];
        if (do_scale) v = fe_mul(v, scale);
        base[(size_t)j << s0] = v;
    }
}

// in-place bit-reversal permutation
__global__ __launch_bounds__(256) void k_bitrev_swap(Fr *__restrict__ data, uint32_t log_n, size_t batch_stride) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << log_n)) return;
    uint32_t r = __brev((uint32_t)i) >> (32 - log_n);
    if (i < r) {
        Fr *p = data + (size_t)blockIdx.y * batch_stride;
        Fr a = p[i], b = p[r];
        p[i] = b;
        p[r] = a;
    }
}

struct PassPlan {
    int npass;
    uint32_t s0[4], s1[4];
};
static PassPlan make_plan(uint32_t log_n) {
    PassPlan p;
    p.npass = (int)((log_n + kMaxTileLog - 1) / kMaxTileLog);
    uint32_t base = log_n / p.npass, extra = log_n % p.npass, s = 0;
    for (int i = 0; i < p.npass; i++) {
        uint32_t k = base + ((uint32_t)i < extra ? 1 : 0);
        p.s0[i] = s;
        p.s1[i] = s + k;
        s += k;
    }
    return p;
}

// DIF (natural -> bit-reversed) or DIT (bit-reversed -> natural) butterflies over `batch`
// vectors; `scale` (if given) multiplies every output element in the last pass.
static int run_passes(plonk_hip_ctx *ctx, Fr *d, uint32_t log_n, size_t batch, bool inverse_tw, bool dif,
                      const Fr *scale) {
    const Fr *tw;
    PLONK_TRY(get_twiddles(ctx, log_n, inverse_tw, &tw));
    PassPlan pl = make_plan(log_n);
    const size_t n = (size_t)1 << log_n;
    Fr sc = scale ? *scale : fe_one<FrCfg>();
    for (int q = 0; q < pl.npass; q++) {
        int i = dif ? pl.npass - 1 - q : q;
        uint32_t k = pl.s1[i] - pl.s0[i];
        bool last = q == pl.npass - 1;
        dim3 grid((unsigned)(n >> k), (unsigned)batch);
        size_t lds = ((size_t)1 << k) * sizeof(Fr);
        if (dif) {
            hipLaunchKernelGGL(k_ntt_pass<true>, grid, dim3(kNttBlock), lds, ctx->stream, d, log_n, pl.s0[i], pl.s1[i],
                               tw, n, sc, (last && scale) ? 1 : 0);
        } else {
            hipLaunchKernelGGL(k_ntt_pass<false>, grid, dim3(kNttBlock), lds, ctx->stream, d, log_n, pl.s0[i],
                               pl.s1[i], tw, n, sc, (last && scale) ? 1 : 0);
        }
    }
    return PLONK_HIP_OK;
}

int ntt_dev(plonk_hip_ctx *ctx, Fr *d_data, uint32_t log_n, size_t batch, bool inverse, const Fr *coset_shift) {
    if (log_n > 30 || batch > 65535) return PLONK_HIP_ERR_INVALID_ARG;
    if (batch == 0 || log_n == 0) return PLONK_HIP_OK;  // one point: f(shift * 1) = f_0 either way
    const size_t n = (size_t)1 << log_n;
    if (!inverse) {
        if (coset_shift) PLONK_TRY(scale_by_powers(ctx, d_data, n, batch, *coset_shift));
        PLONK_TRY(run_passes(ctx, d_data, log_n, batch, false, true, nullptr));
        hipLaunchKernelGGL(k_bitrev_swap, dim3(cdiv(n, 256), (unsigned)batch), dim3(256), 0, ctx->stream, d_data, log_n,
                           n);
    } else {
        hipLaunchKernelGGL(k_bitrev_swap, dim3(cdiv(n, 256), (unsigned)batch), dim3(256), 0, ctx->stream, d_data, log_n,
                           n);
        Fr ninv = fe_inv(fr_from_u64(n));
        PLONK_TRY(run_passes(ctx, d_data, log_n, batch, true, false, &ninv));
        if (coset_shift) {
            if (fe_is_zero(*coset_shift)) return PLONK_HIP_ERR_ZERO_DENOMINATOR;
            PLONK_TRY(scale_by_powers(ctx, d_data, n, batch, fe_inv(*coset_shift)));
        }
    }
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

// ------------------------------------------------------------------------------------------
// coset quotient by X^k - 1
// ------------------------------------------------------------------------------------------
// Evaluations on the coset g*<omega_2k> in bit-reversed order: position p holds the point
// g * omega^rev(p); (g omega^e)^k - 1 = g^k (-1)^e - 1 and e is even exactly when p < k.
__global__ __launch_bounds__(256) void k_mul_halves(Fr *__restrict__ data, size_t n, size_t batch_stride, Fr lo_half,
                                                   Fr hi_half) {
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    Fr *q = data + (size_t)blockIdx.y * batch_stride + p;
    *q = fe_mul(*q, p < n / 2 ? lo_half : hi_half);
}

__global__ void k_set_int(int *__restrict__ flags, size_t count, int value) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) flags[i] = value;
}

// flags[b] = 0 if any of data[b][first .. first+count) is non-zero
__global__ __launch_bounds__(256) void k_check_zero(const Fr *__restrict__ data, size_t first, size_t count,
                                                   size_t batch_stride, int *__restrict__ flags) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    if (!fe_is_zero(data[(size_t)blockIdx.y * batch_stride + first + i])) flags[blockIdx.y] = 0;
}

int quotient_coset_dev(plonk_hip_ctx *ctx, Fr *d_f, uint32_t log_k, size_t batch, int *d_divisible) {
    if (log_k > 29 || batch > 65535) return PLONK_HIP_ERR_INVALID_ARG;
    if (batch == 0) return PLONK_HIP_OK;
    const uint32_t log_n = log_k + 1;
    const size_t k = (size_t)1 << log_k, n = 2 * k;
    const Fr g = fr_from_u64(7);  // multiplicative generator: g^k is never +-1 for k | 2^31
    // g^k
    Fr gk = g;
    for (uint32_t i = 0; i < log_k; i++) gk = fe_sqr(gk);
    Fr one = fe_one<FrCfg>();
    Fr z_even = fe_sub(gk, one);             // Z on points with even exponent
    Fr z_odd = fe_sub(fe_neg(gk), one);      // Z on points with odd exponent
    if (fe_is_zero(z_even) || fe_is_zero(z_odd)) return PLONK_HIP_ERR_ZERO_DENOMINATOR;
    Fr inv_even = fe_inv(z_even), inv_odd = fe_inv(z_odd);

    PLONK_TRY(scale_by_powers(ctx, d_f, n, batch, g));
    PLONK_TRY(run_passes(ctx, d_f, log_n, batch, false, true, nullptr));
    hipLaunchKernelGGL(k_mul_halves, dim3(cdiv(n, 256), (unsigned)batch), dim3(256), 0, ctx->stream, d_f, n, n,
                       inv_even, inv_odd);
    Fr ninv = fe_inv(fr_from_u64(n));
    PLONK_TRY(run_passes(ctx, d_f, log_n, batch, true, false, &ninv));
    PLONK_TRY(scale_by_powers(ctx, d_f, n, batch, fe_inv(g)));
    if (d_divisible) {
        hipLaunchKernelGGL(k_set_int, dim3(cdiv(batch, 256)), dim3(256), 0, ctx->stream, d_divisible, batch, 1);
        hipLaunchKernelGGL(k_check_zero, dim3(cdiv(k, 256), (unsigned)batch), dim3(256), 0, ctx->stream, d_f, k, k, n,
                           d_divisible);
    }
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

}  // namespace plonk
