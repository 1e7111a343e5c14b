/* plonk_hip_bench.h -- hardware probes of libplonk_hip.so.
 *
 * Not part of the surface a binding of the reference needs (that is plonk_hip.h): these entry points measure
 * the machine -- the integer multiply-add rate that bounds the bucket accumulation and the transforms, the issue
 * cost of single instructions, the rate of random 128-byte line gathers -- so that a benchmark can state its
 * roofline against figures taken in the same process.  They have no counterpart in the reference.
 */
#ifndef PLONK_HIP_BENCH_H
#define PLONK_HIP_BENCH_H

#include "plonk_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Roofline denominator: sustained rate (multiply-adds per second, whole GPU) of the 32x32+64
 * integer multiply-add `v_mad_u64_u32` issued from registers only, measured after the shader clock has come
 * up (about 100 ms in all). */
int plonk_hip_bench_mad_rate(plonk_hip_ctx *ctx, double *mads_per_second);
/* Issue-cost probe: multiply-add rate with `extra` (0, 1, 2 or 4) cheap integer VALU instructions
 * after every multiply-add, and the shader clock (GHz) the kernel ran at. */
int plonk_hip_bench_mad_mix(plonk_hip_ctx *ctx, int extra, int blocks_per_cu, double *mads_per_second,
                             double *clock_ghz);
/* Sustained Fq Montgomery multiplications per second (whole GPU) of dependent chains in
 * registers; variant 0 = the form the kernels use, 1 = the portable CIOS form. */
int plonk_hip_bench_fq_mul_rate(plonk_hip_ctx *ctx, int variant, int blocks_per_cu, double *muls_per_second);
/* Measured lane-operations/s of one VALU instruction (op: 0 v_lshrrev_b64, 1 v_lshl_add_u64,
 * 2 v_alignbit_b32, 3 v_and_b32, 4 v_add3_u32, 5 v_mad_i64_i32, 6 v_ashrrev_i64, 7 v_mul_lo_u32,
 * 8 v_mad_u64_u32, 9 v_sub_u32): the per-instruction costs behind the kernels' instruction budgets. */
int plonk_hip_bench_op_rate(plonk_hip_ctx *ctx, int op, double *ops_per_second);
/* Measured bytes/s of a random gather of whole 128-byte lines from a table of table_bytes: the
 * memory-side ceiling of the bucket accumulation's table reads.  An odd table_bytes selects the
 * access form the accumulation kernel uses (eight lanes per line, direct global-to-LDS loads);
 * an even one the plain form (every lane reads its own line). */
int plonk_hip_bench_gather_rate(plonk_hip_ctx *ctx, size_t table_bytes, double *bytes_per_second);
/* How the bucket sort ranks keys on this context's device: 1 = one LDS add-with-return per key (the check made
 * when the context was created found the lanes of an instruction served in ascending order), 0 = the
 * ballot form that relies on nothing (the check failed, or PLONK_HIP_SORT_RANK=ballot was set). */
int plonk_hip_bench_sort_rank_mode(plonk_hip_ctx *ctx);
/* What batch-affine bucket additions would cost (a design probe, not a product path): every lane of 512 blocks
 * of 256 keeps accumulators_per_lane (1, 2 or 4) affine accumulators in registers and adds one point of the
 * context's SRS (its first 4096 points, which must be loaded) to each per step; the denominators of a block share
 * ONE field inversion through product scans in LDS (Montgomery's trick), so that an addition costs 6 field
 * products plus its share of the scans and of the inversion.  invert = 0 skips the inversion (wrong points, same
 * work otherwise): the scheme priced as if inversions were free.  adds_per_second: whole GPU.  first_block_out
 * (may be NULL): the first block's 256 * accumulators_per_lane accumulators after `steps` steps as Jacobian
 * points (x 18 limbs), for checking against an independent sum. */
int plonk_hip_bench_batch_affine(plonk_hip_ctx *ctx, int accumulators_per_lane, int steps, int invert,
                                 double *adds_per_second, uint64_t *first_block_out);
/* Fault injection for the tests of the library's self-checks: with on != 0 the commitments of this context
 * skip their bucket sort, so the accumulation meets keys that are not in ascending order, which it must report
 * (PLONK_HIP_ERR_INTERNAL at the next wait) instead of returning a wrong point. */
int plonk_hip_bench_skip_sort(plonk_hip_ctx *ctx, int on);
/* Measurement overrides of the bucket accumulation of this context's commitments: seg_len != 0 fixes the number
 * of sorted positions one lane walks (the library otherwise derives it from the size and from the lanes the device
 * keeps resident); pair_loads = 1 makes the lanes fetch their (bucket, entry) pairs one position at a time, 2 eight
 * at a time through LDS.  0 restores the library's choice of either. */
int plonk_hip_bench_msm_tuning(plonk_hip_ctx *ctx, uint32_t seg_len, int pair_loads);

#ifdef __cplusplus
}
#endif

#endif /* PLONK_HIP_BENCH_H */
