// Peak-rate probes for the roofline: the integer VALU multiply-add that every field
// multiplication is made of (v_mad_u64_u32), issued from registers only.
#include "context.hpp"

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

int bench_mad_rate(plonk_hip_ctx *ctx, double *mads_per_second) {
    const int blocks = 256 * 8, threads = 256, iters = 4096;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1;
    PLONK_HIP_TRY(ctx, hipEventCreate(&e0));
    PLONK_HIP_TRY(ctx, hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        PLONK_HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(k_mad_rate, dim3(blocks), dim3(threads), 0, ctx->stream, (uint32_t *)ctx->scratch_c.ptr,
                           0x9e3779b9u + rep, iters);
        PLONK_HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        PLONK_HIP_TRY(ctx, hipEventSynchronize(e1));
        float ms = 0;
        PLONK_HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    double mads = (double)blocks * threads * (double)iters * 32.0;
    *mads_per_second = mads / (best * 1e-3);
    return PLONK_HIP_OK;
}

}  // namespace plonk
