// Peak-rate probes for the roofline: the integer VALU multiply-add that every field
// multiplication is made of (v_mad_u64_u32), issued from registers only.
#include "context.hpp"
#include "field28.cuh"

namespace plonk {

// Each lane runs 8 independent 64-bit accumulators; every step is one 32x32+64 multiply-add.
__global__ __launch_bounds__(256) void k_mad_rate(uint32_t *__restrict__ out, uint32_t seed, int iters) {
    uint32_t a = seed ^ (threadIdx.x * 2654435761u), b = seed + blockIdx.x * 40503u + 12345u;
    uint64_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = (uint64_t)(a + j) << 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b) : "vcc");
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

// Field multiplication throughput: every lane runs a dependent chain x <- x*y, y <- y*x.
// variant 0: device form (fe_mul), 1: portable CIOS form.
// 28-bit limb form: variant 2 = q28_mul, 3 = q28_sqr + q28_mul
template <int VARIANT>
__global__ __launch_bounds__(256) void k_q28_mul_rate(uint32_t *__restrict__ out, uint32_t seed, int iters) {
    Fq28 x, y;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        x.l[i] = (seed + threadIdx.x * 977u + i * 131u) & 0x00ffffffu;
        y.l[i] = (seed ^ (blockIdx.x * 613u + i * 257u)) & 0x00ffffffu;
    }
    for (int i = 0; i < iters; i++) {
        if (VARIANT == 2) {
            x = q28_mul(x, y);
            y = q28_mul(y, x);
        } else {
            x = q28_sqr(y);
            y = q28_mul(y, x);
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) s ^= x.l[i] ^ y.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k_fq_mul_rate(uint32_t *__restrict__ out, uint32_t seed, int iters) {
    Fq x, y;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        x.l[i] = (seed + threadIdx.x * 977u + i * 131u) & 0x0fffffffu;
        y.l[i] = (seed ^ (blockIdx.x * 613u + i * 257u)) & 0x0fffffffu;
    }
    for (int i = 0; i < iters; i++) {
        if (VARIANT == 0) {
            x = fe_mul(x, y);
            y = fe_mul(y, x);
        } else {
            x = fe_mul_cios(x, y);
            y = fe_mul_cios(y, x);
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) s ^= x.l[i] ^ y.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int bench_fq_mul_rate(plonk_hip_ctx *ctx, int variant, int blocks_per_cu, double *muls_per_second) {
    const int blocks = 256 * blocks_per_cu, threads = 256, iters = 512;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1;
    PLONK_HIP_TRY(ctx, hipEventCreate(&e0));
    PLONK_HIP_TRY(ctx, hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        PLONK_HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        if (variant == 2) {
            hipLaunchKernelGGL(k_q28_mul_rate<2>, dim3(blocks), dim3(threads), 0, ctx->stream,
                               (uint32_t *)ctx->scratch_c.ptr, 0x1234567u + rep, iters);
        } else if (variant == 3) {
            hipLaunchKernelGGL(k_q28_mul_rate<3>, dim3(blocks), dim3(threads), 0, ctx->stream,
                               (uint32_t *)ctx->scratch_c.ptr, 0x1234567u + rep, iters);
        } else if (variant == 0) {
            hipLaunchKernelGGL(k_fq_mul_rate<0>, dim3(blocks), dim3(threads), 0, ctx->stream,
                               (uint32_t *)ctx->scratch_c.ptr, 0x1234567u + rep, iters);
        } else {
            hipLaunchKernelGGL(k_fq_mul_rate<1>, dim3(blocks), dim3(threads), 0, ctx->stream,
                               (uint32_t *)ctx->scratch_c.ptr, 0x1234567u + rep, iters);
        }
        PLONK_HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        PLONK_HIP_TRY(ctx, hipEventSynchronize(e1));
        float ms = 0;
        PLONK_HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *muls_per_second = (double)blocks * threads * (double)iters * 2.0 / (best * 1e-3);
    return PLONK_HIP_OK;
}

// Issue-cost probe: EXTRA cheap integer VALU instructions after every multiply-add, 8 independent
// chains per lane.  Reports multiply-adds per second and the shader clock seen by the kernel.
template <int EXTRA>
__global__ __launch_bounds__(256) void k_mad_mix(uint32_t *__restrict__ out, uint64_t *__restrict__ clocks,
                                                 uint32_t seed, int iters) {
    uint32_t a = seed ^ (threadIdx.x * 2654435761u), b = seed + blockIdx.x * 40503u + 12345u;
    uint64_t acc[8];
    uint32_t x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        acc[j] = (uint64_t)(a + j) << 7;
        x[j] = a * (j + 3);
    }
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b) : "vcc");
#pragma unroll
                for (int e = 0; e < EXTRA; e++) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(x[j]) : "v"(b));
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s ^= acc[j] ^ x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clocks[0] = t1 - t0;
        clocks[1] = r1 - r0;
    }
}

int bench_mad_mix(plonk_hip_ctx *ctx, int extra, int blocks_per_cu, double *mads_per_second, double *clock_ghz) {
    const int blocks = 256 * blocks_per_cu, threads = 256, iters = 2048;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, (size_t)blocks * threads * 4 + 64));
    uint32_t *out = (uint32_t *)ctx->scratch_c.ptr;
    uint64_t *clk = (uint64_t *)((char *)ctx->scratch_c.ptr + (size_t)blocks * threads * 4);
    hipEvent_t e0, e1;
    PLONK_HIP_TRY(ctx, hipEventCreate(&e0));
    PLONK_HIP_TRY(ctx, hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        PLONK_HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        switch (extra) {
            case 0: hipLaunchKernelGGL(k_mad_mix<0>, dim3(blocks), dim3(threads), 0, ctx->stream, out, clk, 7u + rep, iters); break;
            case 1: hipLaunchKernelGGL(k_mad_mix<1>, dim3(blocks), dim3(threads), 0, ctx->stream, out, clk, 7u + rep, iters); break;
            case 2: hipLaunchKernelGGL(k_mad_mix<2>, dim3(blocks), dim3(threads), 0, ctx->stream, out, clk, 7u + rep, iters); break;
            default: hipLaunchKernelGGL(k_mad_mix<4>, dim3(blocks), dim3(threads), 0, ctx->stream, out, clk, 7u + rep, iters); break;
        }
        PLONK_HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        PLONK_HIP_TRY(ctx, hipEventSynchronize(e1));
        float ms = 0;
        PLONK_HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    uint64_t h[2];
    PLONK_HIP_TRY(ctx, hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    *mads_per_second = (double)blocks * threads * (double)iters * 32.0 / (best * 1e-3);
    *clock_ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;  // s_memrealtime ticks at 100 MHz
    return PLONK_HIP_OK;
}

int bench_mad_rate(plonk_hip_ctx *ctx, double *mads_per_second) {
    const int blocks = 256 * 8, threads = 256, iters = 4096;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1;
    PLONK_HIP_TRY(ctx, hipEventCreate(&e0));
    PLONK_HIP_TRY(ctx, hipEventCreate(&e1));
    // The shader clock needs about 60 ms of continuous load to come up from idle: 40 launches of about 2 ms
    // untimed, then the average of 8 (the rate the part sustains, not the best burst).
    const int warm = 40, timed = 8;
    for (int rep = 0; rep < warm; rep++) {
        hipLaunchKernelGGL(k_mad_rate, dim3(blocks), dim3(threads), 0, ctx->stream, (uint32_t *)ctx->scratch_c.ptr,
                           0x9e3779b9u + rep, iters);
    }
    PLONK_HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    for (int rep = 0; rep < timed; rep++) {
        hipLaunchKernelGGL(k_mad_rate, dim3(blocks), dim3(threads), 0, ctx->stream, (uint32_t *)ctx->scratch_c.ptr,
                           0x9e3779b9u + warm + rep, iters);
    }
    PLONK_HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    PLONK_HIP_TRY(ctx, hipEventSynchronize(e1));
    float ms = 0;
    PLONK_HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    double mads = (double)blocks * threads * (double)iters * 32.0 * timed;
    *mads_per_second = mads / (ms * 1e-3);
    return PLONK_HIP_OK;
}

// Random 128-byte-line gather: every lane reads `iters` whole lines (8 x 16-byte loads, the access
// pattern of the bucket accumulation's table reads) at hashed positions of a table of `lines` lines.
__global__ __launch_bounds__(128) void k_gather_rate(const uint4 *__restrict__ table, uint32_t lines, int iters,
                                                    uint32_t seed, uint32_t *__restrict__ out) {
    uint32_t h = seed ^ ((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        uint32_t line = (uint32_t)(((uint64_t)(h ^ (h >> 15)) * lines) >> 32);
        const uint4 *p = table + (size_t)line * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint4 v = p[k];
            acc.x ^= v.x;
            acc.y += v.y;
            acc.z ^= v.z;
            acc.w += v.w;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// The same gather in the form the bucket accumulation uses: the wave fetches its 64 lines with
// eight direct global-to-LDS loads, eight lanes per line (pieces in swizzled order), and every lane
// reads its own line back from LDS.
__global__ __launch_bounds__(128) void k_gather_rate_coop(const uint4 *__restrict__ table, uint32_t lines, int iters,
                                                         uint32_t seed, uint32_t *__restrict__ out) {
    __shared__ uint4 line_buf[2][64 * 8];
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef const __attribute__((address_space(1))) void global_void_t;
    const uint32_t lane = threadIdx.x & 63;
    uint4 *buf = line_buf[threadIdx.x >> 6];
    uint32_t h = seed ^ ((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        uint32_t line = (uint32_t)(((uint64_t)(h ^ (h >> 15)) * lines) >> 32);
#pragma unroll
        for (uint32_t g = 0; g < 8; g++) {
            const uint32_t owner = 8 * g + (lane >> 3);
            const uint32_t idx = (uint32_t)__shfl((int)line, (int)owner, 64);
            const uint32_t piece = (lane & 7) ^ (owner & 7);
            __builtin_amdgcn_global_load_lds((global_void_t *)(table + (size_t)idx * 8 + piece),
                                             (lds_void_t *)(buf + 64 * g), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            uint4 v = buf[8 * lane + (k ^ (lane & 7))];
            acc.x ^= v.x;
            acc.y += v.y;
            acc.z ^= v.z;
            acc.w += v.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// Bytes per second of that gather over a table of table_bytes (rounded down to whole lines);
// table_bytes with its lowest bit set selects the cooperative form.
int bench_gather_rate(plonk_hip_ctx *ctx, size_t table_bytes, double *bytes_per_second) {
    const bool coop = (table_bytes & 1) != 0;
    const uint32_t lines = (uint32_t)std::min<size_t>(table_bytes / 128, 0x7fffffffu);
    if (lines == 0) return PLONK_HIP_ERR_INVALID_ARG;
    const int blocks = 256 * 24, threads = 128, iters = 64;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_a, (size_t)lines * 128));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, (size_t)blocks * threads * 4));
    PLONK_HIP_TRY(ctx, hipMemsetAsync(ctx->scratch_a.ptr, 1, (size_t)lines * 128, ctx->stream));
    hipEvent_t e0, e1;
    PLONK_HIP_TRY(ctx, hipEventCreate(&e0));
    PLONK_HIP_TRY(ctx, hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        PLONK_HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        if (coop) {
            hipLaunchKernelGGL(k_gather_rate_coop, dim3(blocks), dim3(threads), 0, ctx->stream,
                               (const uint4 *)ctx->scratch_a.ptr, lines, iters, 0x51ed27u + rep,
                               (uint32_t *)ctx->scratch_c.ptr);
        } else {
            hipLaunchKernelGGL(k_gather_rate, dim3(blocks), dim3(threads), 0, ctx->stream,
                               (const uint4 *)ctx->scratch_a.ptr, lines, iters, 0x51ed27u + rep,
                               (uint32_t *)ctx->scratch_c.ptr);
        }
        PLONK_HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        PLONK_HIP_TRY(ctx, hipEventSynchronize(e1));
        float ms = 0;
        PLONK_HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *bytes_per_second = (double)blocks * threads * (double)iters * 128.0 / (best * 1e-3);
    return PLONK_HIP_OK;
}

// Issue rate of single VALU instructions: 8 independent registers (pairs) per lane, 32 instructions
// per loop iteration.  OP selects the instruction (see bench_op_rate).
template <int OP>
__global__ __launch_bounds__(256) void k_op_rate(uint32_t *__restrict__ out, uint32_t seed, int iters) {
    uint32_t a = seed ^ (threadIdx.x * 2654435761u), b = (seed + blockIdx.x * 40503u + 12345u) | 1u;
    uint64_t r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = ((uint64_t)(a + j) << 17) | b;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (OP == 0) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(r[j]));
                if (OP == 1) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r[j]) : "v"(r[(j + 1) & 7]));
                if (OP == 2) asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(*(uint32_t *)&r[j]) : "v"(a));
                if (OP == 3) asm volatile("v_and_b32 %0, %1, %0" : "+v"(*(uint32_t *)&r[j]) : "v"(b));
                if (OP == 4) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(*(uint32_t *)&r[j]) : "v"(a), "v"(b));
                if (OP == 5) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(r[j]) : "v"(a), "v"(b) : "vcc");
                if (OP == 6) asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(r[j]));
                if (OP == 7) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(*(uint32_t *)&r[j]) : "v"(b));
                if (OP == 8) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[j]) : "v"(a), "v"(b) : "vcc");
                if (OP == 9) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(*(uint32_t *)&r[j]) : "v"(a));
                // the multiply-add again with every operation waiting for the one before it (10) or
                // for the one two before it (11)
                if (OP == 10) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[0]) : "v"(a), "v"(b) : "vcc");
                if (OP == 11) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[j & 1]) : "v"(a), "v"(b) : "vcc");
            }
        }
    }
    uint64_t sacc = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) sacc ^= r[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)sacc ^ (uint32_t)(sacc >> 32);
}

// Wave-instructions per second per SIMD-equivalent are derived by the caller; this returns lane
// operations per second for instruction `op`:
// 0 v_lshrrev_b64, 1 v_lshl_add_u64, 2 v_alignbit_b32, 3 v_and_b32, 4 v_add3_u32, 5 v_mad_i64_i32,
// 6 v_ashrrev_i64, 7 v_mul_lo_u32, 8 v_mad_u64_u32, 9 v_sub_u32, 10 / 11 v_mad_u64_u32 in one / two
// dependent chains (0-9 run eight independent chains).  op + 100: one wave per SIMD instead of eight.
int bench_op_rate(plonk_hip_ctx *ctx, int op, double *ops_per_second) {
    const bool lone = op >= 100;
    if (lone) op -= 100;
    const int blocks = lone ? 1024 : 256 * 8, threads = lone ? 64 : 256, iters = 2048;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, (size_t)blocks * threads * 4));
    uint32_t *out = (uint32_t *)ctx->scratch_c.ptr;
    hipEvent_t e0, e1;
    PLONK_HIP_TRY(ctx, hipEventCreate(&e0));
    PLONK_HIP_TRY(ctx, hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        PLONK_HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
#define PLONK_OP_CASE(N) case N: hipLaunchKernelGGL(k_op_rate<N>, dim3(blocks), dim3(threads), 0, ctx->stream, out, 11u + rep, iters); break;
        switch (op) {
            PLONK_OP_CASE(0) PLONK_OP_CASE(1) PLONK_OP_CASE(2) PLONK_OP_CASE(3) PLONK_OP_CASE(4)
            PLONK_OP_CASE(5) PLONK_OP_CASE(6) PLONK_OP_CASE(7) PLONK_OP_CASE(8) PLONK_OP_CASE(9)
            PLONK_OP_CASE(10) PLONK_OP_CASE(11)
            default: (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return PLONK_HIP_ERR_INVALID_ARG;
        }
#undef PLONK_OP_CASE
        PLONK_HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        PLONK_HIP_TRY(ctx, hipEventSynchronize(e1));
        float ms = 0;
        PLONK_HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ops_per_second = (double)blocks * threads * (double)iters * 32.0 / (best * 1e-3);
    return PLONK_HIP_OK;
}

}  // namespace plonk
