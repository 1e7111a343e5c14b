/*
 * libplonk_hip -- MI355X (gfx950) backend for the KZG / polynomial hot path of the
 * hjeljeli32/plonk prover.  Flat C ABI: plain pointers and sizes, no C++ or framework types.
 *
 * This is the seam a Rust shim (`cc`/`build.rs`) binds: each entry point names the reference
 * function it stands behind (paths relative to the reference repository root).
 *
 * Data formats (exactly ark-ff / ark-ec 0.5 in-memory layouts, little-endian u64 limbs,
 * Montgomery form):
 *   Fr            4 x u64   (R = 2^256)       -- `ark_bls12_381::Fr`
 *   Fq            6 x u64   (R = 2^384)
 *   G1 projective 18 x u64  X | Y | Z, Jacobian, identity <=> Z == 0 -- `G1Projective`
 *   G1 affine     12 x u64  x | y, identity encoded as all-zero
 * so `coeffs.as_ptr() as *const u64` and `tau_powers_g1.as_ptr() as *const u64` can be passed
 * straight through.  Returned G1 values are valid Jacobian representatives (not normalised to
 * Z = 1); ark's `==` and `serialize_compressed` are representation independent.
 *
 * Unless a name ends in `_dev`, pointers are HOST pointers, calls are synchronous and results
 * are valid on return.  `_dev` entry points take DEVICE pointers (memory of the context's GPU),
 * enqueue on the context's stream and return without waiting; call plonk_hip_sync().
 * A context is bound to one GPU.  Entry points may be called from any thread: calls on one
 * context are serialised by a lock inside it (`_dev` calls only enqueue, so two threads sharing
 * a context still share its stream order); use one context per thread for concurrency.
 * All entry points return 0 on success or a negative PLONK_HIP_ERR_*.
 * Nothing unwinds across this boundary.
 */
#ifndef PLONK_HIP_H
#define PLONK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct plonk_hip_ctx plonk_hip_ctx;

enum {
    PLONK_HIP_OK = 0,
    /* kzg_commit: SRS shorter than the polynomial -> CommitError::CommitFailed
     * (plonk/src/common/kzg.rs:11-15,49-50); in kzg_open the reference panics on the
     * out-of-bounds index (kzg.rs:83) */
    PLONK_HIP_ERR_SRS_TOO_SHORT = -1,
    /* empty coefficient vector / quotient: the reference panics on `.reduce(..).unwrap()`
     * (kzg.rs:57-58, :85) */
    PLONK_HIP_ERR_EMPTY_POLY = -2,
    /* inverse of zero: the reference panics on `.inverse().unwrap()`
     * (plonk/src/common/polynomials.rs:42, plonk/src/common/protocols.rs:349,583) */
    PLONK_HIP_ERR_ZERO_DENOMINATOR = -3,
    PLONK_HIP_ERR_INVALID_ARG = -4,
    PLONK_HIP_ERR_NO_SRS = -5,
    PLONK_HIP_ERR_HIP = -6,         /* a HIP runtime call failed; see plonk_hip_last_error() */
    PLONK_HIP_ERR_OUT_OF_MEMORY = -7,
    PLONK_HIP_ERR_NOT_DIVISIBLE = -8, /* coset quotient asked for a polynomial Z_Omega does not divide */
    /* a compressed G1 encoding that `G1::deserialize_compressed` rejects; the reference panics on
     * `.expect("Failed to deserialize G1")` (plonk/src/setup_global_params/json.rs:30-33) */
    PLONK_HIP_ERR_INVALID_POINT = -9,
    /* a kernel found its own intermediate data inconsistent (the bucket sort's output not in ascending
     * order) and the results queued since the last successful wait must not be used; the context
     * switches to its conservative sort and stays usable.  No counterpart in the reference. */
    PLONK_HIP_ERR_INTERNAL = -10
};

/* ----------------------------------------------------------------------------------------
 * Context
 * -------------------------------------------------------------------------------------- */
int plonk_hip_ctx_create(int device, plonk_hip_ctx **out);
void plonk_hip_ctx_destroy(plonk_hip_ctx *ctx);
const char *plonk_hip_strerror(int code);
/* Text of the last failing HIP call on this context ("" if none). */
const char *plonk_hip_last_error(plonk_hip_ctx *ctx);
/* Window width c of the signed-digit bucket method used for the NEXT SRS load (0 = choose
 * from the SRS size).  2 <= c <= 24. */
int plonk_hip_ctx_set_window_bits(plonk_hip_ctx *ctx, int c);
/* The context's HIP stream (a hipStream_t), for callers that order their own work on it. */
void *plonk_hip_ctx_stream(plonk_hip_ctx *ctx);
int plonk_hip_sync(plonk_hip_ctx *ctx);

/* Device memory of the context's GPU, for callers that keep data resident (`_dev` calls). */
int plonk_hip_malloc(plonk_hip_ctx *ctx, size_t bytes, void **dptr);
int plonk_hip_free(plonk_hip_ctx *ctx, void *dptr);
int plonk_hip_memcpy_h2d(plonk_hip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int plonk_hip_memcpy_d2h(plonk_hip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* ----------------------------------------------------------------------------------------
 * SRS -- `GlobalParameters.tau_powers_g1` (plonk/src/common/kzg.rs:17-21)
 * -------------------------------------------------------------------------------------- */
/* Uploads n Jacobian points, normalises them to affine on the GPU and builds the resident
 * window tables ([2^(c*w)]P_i for every window w).  Replaces any previous SRS. */
int plonk_hip_srs_load_projective(plonk_hip_ctx *ctx, const uint64_t *pts /* n x 18 */, size_t n);
int plonk_hip_srs_size(plonk_hip_ctx *ctx, size_t *n);
/* Reads back SRS points [first, first+count) as Jacobian triples with Z = 1. */
int plonk_hip_srs_read(plonk_hip_ctx *ctx, size_t first, size_t count, uint64_t *out /* count x 18 */);

/* The SRS from its compressed encodings, as `SetupGlobalParamsOutputJson::into_setup_output` reads
 * it (plonk/src/setup_global_params/json.rs:22-48: hex -> 48 bytes -> `G1::deserialize_compressed`;
 * the hex step stays with the caller).  bytes = n x 48.  check_subgroup != 0 validates like the
 * reference does (on the curve and in the prime-order subgroup); 0 skips the subgroup test for
 * trusted files.  On PLONK_HIP_ERR_INVALID_POINT *bad_index (may be NULL) is the first rejected
 * entry and the context holds no SRS. */
int plonk_hip_srs_load_compressed(plonk_hip_ctx *ctx, const uint8_t *bytes, size_t n, int check_subgroup,
                                  size_t *bad_index);
/* Batch forms of `G1::deserialize_compressed` / `serialize_compressed` (json.rs:30-33;
 * plonk/src/common/utils.rs:70-72, plonk/src/common/proof.rs:58-268): n x 48 bytes <-> n x 18 limbs
 * (decoded points have Z = 1; the identity is (1, 1, 0)). */
int plonk_hip_g1_decompress(plonk_hip_ctx *ctx, const uint8_t *bytes, size_t n, int check_subgroup,
                            uint64_t *out_pts /* n x 18 */, size_t *bad_index);
int plonk_hip_g1_compress(plonk_hip_ctx *ctx, const uint64_t *pts /* n x 18 */, size_t n, uint8_t *bytes);

/* `kzg_setup(degree)` (kzg.rs:24-45) with tau supplied by the caller: computes [tau^i]G for
 * i = 0..degree on the GPU, installs the result as the context's SRS and, if out_pts is not
 * NULL, returns it ((degree+1) x 18).  The reference's own tau is `Fr::rand(test_rng())`;
 * drawing it stays on the caller's side.  tau_g2 (kzg.rs:39): plonk_hip_kzg_setup_tau_g2 below. */
int plonk_hip_kzg_setup(plonk_hip_ctx *ctx, const uint64_t tau[4], size_t degree, uint64_t *out_pts);
/* The slice [tau^first .. tau^(first+count-1)]G of the same sequence: the SRS range one GPU
 * holds when a large commitment is sharded by point range (one context per GPU). */
int plonk_hip_kzg_setup_range(plonk_hip_ctx *ctx, const uint64_t tau[4], size_t first, size_t count,
                              uint64_t *out_pts);

/* ----------------------------------------------------------------------------------------
 * Commit / open -- the G1 MSM
 * -------------------------------------------------------------------------------------- */
/* `kzg_commit(gp, f)` (kzg.rs:48-60): out = sum_i coeffs[i] * tau_powers_g1[i], n = coeffs.len().
 * n > SRS size -> PLONK_HIP_ERR_SRS_TOO_SHORT; n == 0 -> PLONK_HIP_ERR_EMPTY_POLY. */
int plonk_hip_kzg_commit(plonk_hip_ctx *ctx, const uint64_t *coeffs /* n x 4 */, size_t n, uint64_t out[18]);
/* Same with coefficients and result (18 x u64) in device memory; asynchronous. */
int plonk_hip_kzg_commit_dev(plonk_hip_ctx *ctx, const void *d_coeffs, size_t n, void *d_out);
/* The batched commitments of the prover (protocols.rs:191-195, 287-288, 652-653, 761;
 * prover/part1.rs:66, part2.rs:48; setup_verification_key/mod.rs:74,108): `batch`
 * polynomials against the same SRS.  outs = batch x 18. */
int plonk_hip_kzg_commit_batch(plonk_hip_ctx *ctx, const uint64_t *const *coeffs, const size_t *ns, size_t batch,
                               uint64_t *outs);

/* d_out = sum of n Jacobian points in device memory (n x 18 -> 18): combines the per-GPU
 * partial commitments after they have been gathered (the `+` of kzg.rs:57 across shards). */
int plonk_hip_g1_sum_dev(plonk_hip_ctx *ctx, const void *d_pts, size_t n, void *d_out);

/* Exchange of bucket partials between GPUs (the other way to split the sum of kzg.rs:53-58 over
 * point ranges): every rank stops its local commitment after the bucket sums, the ranks exchange
 * slices of their bucket arrays (all-to-all of opaque bytes), and each rank adds up and reduces
 * one slice, which spreads the serial bucket reduction over the GPUs.
 *   _bucket_layout:  number of buckets of the installed SRS's window plan and bytes per bucket
 *                    (equal on ranks that hold equally many points).
 *   _buckets_dev:    the commitment of d_coeffs (n x 4, device) up to the bucket sums, written to
 *                    d_buckets (nbuckets x bytes_per_bucket of device memory).
 *   _reduce_bucket_slice_dev: d_parts = nparts arrays of `count` buckets each, all for buckets
 *                    [first, first + count) (count a power of two >= 4, first a multiple of count);
 *                    d_out (18 words, device) = this slice's share of the commitment.  The shares
 *                    of all slices add up (plonk_hip_g1_sum_dev) to what plonk_hip_kzg_commit_dev
 *                    returns for the concatenated input. */
int plonk_hip_msm_bucket_layout(plonk_hip_ctx *ctx, size_t *nbuckets, size_t *bytes_per_bucket);
int plonk_hip_msm_buckets_dev(plonk_hip_ctx *ctx, const void *d_coeffs, size_t n, void *d_buckets);
int plonk_hip_msm_reduce_bucket_slice_dev(plonk_hip_ctx *ctx, const void *d_parts, size_t nparts, size_t first,
                                          size_t count, void *d_out);

/* `kzg_evaluate(gp, f, u)` (kzg.rs:63-88): v = f(u); q = (f - v) / (X - u);
 * proof = sum_i q_i * tau_powers_g1[i].  n = coeffs.len().
 * n < 2 -> PLONK_HIP_ERR_EMPTY_POLY (empty quotient); n - 1 > SRS size -> SRS_TOO_SHORT. */
int plonk_hip_kzg_open(plonk_hip_ctx *ctx, const uint64_t *coeffs /* n x 4 */, size_t n, const uint64_t u[4],
                       uint64_t v[4], uint64_t proof[18]);
/* Several openings of one polynomial (protocols.rs:655-659, 763-767: t at w^(k-1), rp, w*rp;
 * T at r, w*r, w^2*r).  us = m x 4, vs = m x 4, proofs = m x 18. */
int plonk_hip_kzg_open_many(plonk_hip_ctx *ctx, const uint64_t *coeffs, size_t n, const uint64_t *us, size_t m,
                            uint64_t *vs, uint64_t *proofs);

/* The G2 half of `kzg_setup` (kzg.rs:38-39): out = gp.tau_g2 = G2::generator() * tau in the 96-byte compressed
 * form of plonk/data/srs.json.  Host code (no context, no GPU). */
int plonk_hip_kzg_setup_tau_g2(const uint64_t tau[4], uint8_t out[96]);

/* `kzg_verify(gp, com_f, u, v, proof)` (kzg.rs:91-98): *ok = [ e(com_f - v G, H) == e(proof, tau H - u H) ].
 * Host code (no context, no GPU): the check is taken as e(com_f - v G + u proof, H) * e(-proof, tau H) == 1, so
 * only G1 arithmetic depends on the arguments; the Miller-loop lines of H and tau H are computed once per
 * tau_g2 and kept.  tau_g2 = gp.tau_g2 (kzg.rs:20) in the 96-byte compressed form of plonk/data/srs.json; it is
 * validated (encoding, curve, subgroup) when first seen: PLONK_HIP_ERR_INVALID_POINT otherwise.  com and proof
 * are Jacobian points as everywhere in this interface; each must lie on the curve (checked:
 * PLONK_HIP_ERR_INVALID_POINT otherwise) and in the prime-order subgroup (PRECONDITION, not checked: the
 * reference's G1 type guarantees both through validated deserialisation; points made by this library or decoded
 * by plonk_hip_g1_decompress with check_subgroup != 0 satisfy it); u, v Montgomery limbs. */
int plonk_hip_kzg_verify(const uint8_t tau_g2[96], const uint64_t com[18], const uint64_t u[4], const uint64_t v[4],
                         const uint64_t proof[18], int *ok);
/* The n kzg_verify calls of one proof (15 in the reference's verifier: protocols.rs:229-231, 706-718, 804-808,
 * verifier/part4.rs:20-29) as ONE pairing-product check: with weights rho_i,
 *   L = sum rho_i (com_i - v_i G + u_i proof_i),  R = sum rho_i proof_i,  *ok = [ e(L, H) == e(R, tau H) ]:
 * two Miller loops and one final exponentiation instead of 2 n pairings.  rho = n x 2 words, the weights as
 * 128-bit integers (little-endian words); they must be drawn after the proofs are fixed and be unpredictable
 * to their author (rho_0 = 1 is fine): then *ok == 1 means every single check holds, except with probability
 * about n / 2^128.  coms, proofs = n x 18; us, vs = n x 4.  n == 0: *ok = 1. */
int plonk_hip_kzg_verify_batch(const uint8_t tau_g2[96], size_t n, const uint64_t *coms, const uint64_t *us,
                               const uint64_t *vs, const uint64_t *proofs, const uint64_t *rho, int *ok);

/* GPU time of whatever is enqueued on the context's stream between the two calls (HIP events on that stream;
 * _stop waits for the work to finish).  For measurement harnesses: a framework's own events only see the
 * framework's stream. */
int plonk_hip_timer_start(plonk_hip_ctx *ctx);
int plonk_hip_timer_stop(plonk_hip_ctx *ctx, float *ms);
/* Per-phase GPU time (ms, HIP events on the context's stream) of the most recent MSM run by
 * this context; valid after plonk_hip_profile(ctx, 1).  Phase names from
 * plonk_hip_msm_phase_name(i), i < PLONK_HIP_MSM_PHASES. */
#define PLONK_HIP_MSM_PHASES 6
int plonk_hip_profile(plonk_hip_ctx *ctx, int enable);
int plonk_hip_msm_phase_ms(plonk_hip_ctx *ctx, float ms[PLONK_HIP_MSM_PHASES]);
const char *plonk_hip_msm_phase_name(int i);
/* Window plan in force: bits per window, windows, buckets. */
int plonk_hip_msm_plan(plonk_hip_ctx *ctx, int *window_bits, int *windows, size_t *buckets);
/* Host-only planning, usable without a GPU: the window plan an SRS of srs_n points would get, and
 * the passes of a 2^log_n-point transform (returns the number of passes, stage ranges
 * [first_stage[i], end_stage[i]) in ascending order; 0 for sizes below 8, which take no tiled pass). */
int plonk_hip_msm_plan_for(size_t srs_n, int window_bits_override, int *window_bits, int *windows, size_t *buckets);
int plonk_hip_ntt_plan(uint32_t log_n, int natural_order, int max_passes, uint32_t *first_stage,
                       uint32_t *end_stage);

/* ----------------------------------------------------------------------------------------
 * Polynomials over Fr
 * -------------------------------------------------------------------------------------- */
/* Evaluations over Omega = construct_Omega(2^log_n) (plonk/src/common/utils.rs:11-38,
 * omega = 7^((r-1)/n)), natural order in and out, in place, `batch` vectors of 2^log_n
 * elements stored back to back.
 *   inverse = 0: data[j] <- f(omega^j), the evaluations the reference takes one Horner at a
 *                time (protocols.rs:243-250, 345-353, 461-468, 571-588; prover/part4.rs:36-42);
 *   inverse = 1: data <- coefficients of `interpolate_polynomial(Omega, data)`
 *                (plonk/src/common/polynomials.rs:50-58), zero padded to 2^log_n. */
int plonk_hip_fr_ntt(plonk_hip_ctx *ctx, uint64_t *data, uint32_t log_n, size_t batch, int inverse);
int plonk_hip_fr_ntt_dev(plonk_hip_ctx *ctx, void *d_data, uint32_t log_n, size_t batch, int inverse);
/* The same over Omega = construct_Omega(n) for n = 2^k or 3 * 2^k (the reference's domains have
 * 3 points per gate, plonk/src/setup_global_params/mod.rs:57; its example circuit has n = 12):
 * one radix-3 step around the power-of-two transform.  PLONK_HIP_ERR_INVALID_ARG for other n. */
int plonk_hip_fr_ntt_mixed(plonk_hip_ctx *ctx, uint64_t *data, size_t n, size_t batch, int inverse);
int plonk_hip_fr_ntt_mixed_dev(plonk_hip_ctx *ctx, void *d_data, size_t n, size_t batch, int inverse);
/* The same over the coset shift * Omega: forward evaluates f at shift*omega^j, inverse
 * interpolates from those values. */
int plonk_hip_fr_coset_ntt(plonk_hip_ctx *ctx, uint64_t *data, uint32_t log_n, size_t batch, const uint64_t shift[4],
                           int inverse);
int plonk_hip_fr_coset_ntt_dev(plonk_hip_ctx *ctx, void *d_data, uint32_t log_n, size_t batch,
                               const uint64_t shift[4], int inverse);

/* `compose_polynomials(f, [0, w])` = f(wX): coeffs[i] <- coeffs[i] * w^i
 * (polynomials.rs:80-113 as called from protocols.rs:258-270, 361-380, 596-633, 728-740;
 * prover/part3.rs:42-54). */
int plonk_hip_fr_scale_by_powers(plonk_hip_ctx *ctx, uint64_t *coeffs, size_t n, const uint64_t w[4]);

/* `f.evaluate(&u)` (kzg.rs:65). */
int plonk_hip_fr_eval(plonk_hip_ctx *ctx, const uint64_t *coeffs, size_t n, const uint64_t u[4], uint64_t out[4]);

/* `compute_q_zero_test(k, f)` (protocols.rs:157-166): quotient of f (n coefficients) by
 * X^k - 1 with the remainder dropped, exactly as long division gives it, whether or not
 * X^k - 1 divides f.  q receives n - k coefficients (n > k), untrimmed. */
int plonk_hip_fr_div_vanishing(plonk_hip_ctx *ctx, const uint64_t *f, size_t n, size_t k, uint64_t *q);
/* The same quotient for `batch` polynomials of n = 2^(log_k+1) coefficients each, k = 2^log_k,
 * by coset evaluation: coset-NTT on the 2k domain, pointwise division by Z on the coset,
 * coset-iNTT.  Equal to long division only when X^k - 1 divides f; divisible[b] reports
 * whether it did (the top k coefficients of the result vanished).  q = batch x k x 4. */
int plonk_hip_fr_quotient_coset(plonk_hip_ctx *ctx, const uint64_t *f, uint32_t log_k, size_t batch, uint64_t *q,
                                int *divisible);
int plonk_hip_fr_quotient_coset_dev(plonk_hip_ctx *ctx, void *d_f_inout, uint32_t log_k, size_t batch,
                                    int *d_divisible);


/* out[i] = f(points[i]), i < m: the `f.evaluate(&Omega[j])` calls of the t builders
 * (protocols.rs:246, 348, 464, 577-579) for any point set, e.g. an Omega whose size is not a
 * power of two. */
int plonk_hip_fr_eval_many(plonk_hip_ctx *ctx, const uint64_t *coeffs, size_t n, const uint64_t *points, size_t m,
                           uint64_t *out /* m x 4 */);
/* out[j] = f(omega^j), j < k = 2^log_k, for a polynomial of any length n (it is first reduced
 * modulo X^k - 1, which keeps its values on Omega). */
int plonk_hip_fr_evaluate_over_omega(plonk_hip_ctx *ctx, const uint64_t *coeffs, size_t n, uint32_t log_k,
                                     uint64_t *out /* k x 4 */);
/* data[i] <- data[0] * ... * data[i]  (the `.reduce(|t_eval, f_eval| t_eval * f_eval)` of
 * protocols.rs:243-250, computed once instead of per i) and the additive form (protocols.rs:461-468). */
int plonk_hip_fr_prefix_product(plonk_hip_ctx *ctx, uint64_t *data, size_t n);
int plonk_hip_fr_prefix_sum(plonk_hip_ctx *ctx, uint64_t *data, size_t n);
/* data[i] <- 1 / data[i]; PLONK_HIP_ERR_ZERO_DENOMINATOR if some element is zero (data is then
 * unspecified), where the reference panics on `.inverse().unwrap()` (protocols.rs:349, 583). */
int plonk_hip_fr_batch_inverse(plonk_hip_ctx *ctx, uint64_t *data, size_t n);
/* out[i] = a * x[i] + b * y[i] + (i == 0 || broadcast ? c : 0) for i < max(nx, ny), coefficients past
 * the end of x or y counting as zero: the sums, differences and scalar multiples of dense polynomials
 * that the gadgets form inline (`&f + &g`, `&f - &g`, `r - s W - f`: protocols.rs:272, 382, 490, 583-584,
 * 620-635, 736-749; prover/part2.rs:45, part3.rs:56-59), and with broadcast = 1 the same on value
 * vectors over Omega (numerators and denominators of protocols.rs:578-582).  a, b, c: 4 limbs each. */
int plonk_hip_fr_linear_combination(plonk_hip_ctx *ctx, const uint64_t *x, size_t nx, const uint64_t *y, size_t ny,
                                    const uint64_t a[4], const uint64_t b[4], const uint64_t c[4], int broadcast,
                                    uint64_t *out /* max(nx, ny) x 4 */);
/* out[i] = prod_{j<=i} num[j] / den[j]: the t values of the rational product check and of the
 * prescribed-permutation check (protocols.rs:345-353, 571-588). */
int plonk_hip_fr_ratio_prefix_product(plonk_hip_ctx *ctx, const uint64_t *num, const uint64_t *den, size_t n,
                                      uint64_t *out /* n x 4 */);
/* Dense product `&a * &b` (protocols.rs:272, 382, 490, 635, 749; prover/part3.rs:59):
 * na + nb - 1 coefficients, untrimmed.  Nothing is written when na or nb is 0. */
int plonk_hip_fr_poly_mul(plonk_hip_ctx *ctx, const uint64_t *a, size_t na, const uint64_t *b, size_t nb,
                          uint64_t *out);
/* `construct_vanishing_polynomial_from_roots(roots)` (plonk/src/common/utils.rs:51-62):
 * prod (X - roots[i]), m + 1 coefficients. */
int plonk_hip_fr_vanishing_from_roots(plonk_hip_ctx *ctx, const uint64_t *roots, size_t m, uint64_t *z);
/* `compute_q_zero_test_from_roots(roots, f)` (protocols.rs:169-181): quotient of f (n coefficients)
 * by prod (X - roots[i]) with the remainder dropped; q receives n - m coefficients (n > m),
 * untrimmed. */
int plonk_hip_fr_div_by_roots(plonk_hip_ctx *ctx, const uint64_t *f, size_t n, const uint64_t *roots, size_t m,
                              uint64_t *q);
/* `interpolate_polynomial(xs, ys)` (plonk/src/common/polynomials.rs:14-58) for arbitrary points:
 * n coefficients, untrimmed.  Quadratic work, n <= 65536 (use plonk_hip_fr_ntt with inverse = 1 when
 * xs = Omega).  PLONK_HIP_ERR_ZERO_DENOMINATOR when two xs coincide (the reference panics,
 * polynomials.rs:42). */
int plonk_hip_fr_interpolate(plonk_hip_ctx *ctx, const uint64_t *xs, const uint64_t *ys, size_t n, uint64_t *coeffs);
/* `interpolate_polynomial(x_points, y_points)` for x_points = [1, q, q^2, ..., q^(n-1)] in O(n log n) (two
 * products of length n; no limit on n beyond the transforms'): the gate rows of a circuit are such a set
 * (Omega[3 l] = (omega^3)^l, plonk/src/setup_proving_key/mod.rs:36-43; plonk/src/prover/part3.rs:32-33) and are a
 * whole subgroup only for one circuit size of each family.  PLONK_HIP_ERR_ZERO_DENOMINATOR when two of the points
 * coincide (q^k = 1 for some 0 < k < n), like the reference's panic (polynomials.rs:42).  The _dev form works in
 * place on n values in device memory and waits for its result. */
int plonk_hip_fr_interpolate_geometric(plonk_hip_ctx *ctx, const uint64_t q[4], const uint64_t *ys, size_t n,
                                       uint64_t *coeffs);
int plonk_hip_fr_interpolate_geometric_dev(plonk_hip_ctx *ctx, const uint64_t q[4], void *d_values, size_t n);

/* ---- Device-resident variants ------------------------------------------------------------------
 * The polynomial entry points above copy their operands from and to host memory on every call, which
 * is what a caller holding `DensePolynomial`s needs; a prover that keeps its polynomials on the GPU
 * between the steps of prover/part1-5.rs uses these instead.  Pointers named d_* are device memory
 * from plonk_hip_malloc; scalars (points, factors) and results that the transcript needs (opened
 * values, commitments) stay on the host.  Same arithmetic, same error codes as the host forms. */
int plonk_hip_memcpy_d2d(plonk_hip_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);
int plonk_hip_memset_zero(plonk_hip_ctx *ctx, void *dst_dev, size_t bytes);
/* `kzg_evaluate(gp, f, u)` (kzg.rs:63-88) for m points u of one resident polynomial: vs = m x 4,
 * proofs = m x 18 (host). */
int plonk_hip_kzg_open_many_dev(plonk_hip_ctx *ctx, const void *d_coeffs, size_t n, const uint64_t *us, size_t m,
                                uint64_t *vs, uint64_t *proofs);
/* `kzg_commit` (kzg.rs:48-60) of `batch` resident polynomials, commitments to the host (batch x 18);
 * batch <= 64. */
int plonk_hip_kzg_commit_many_dev(plonk_hip_ctx *ctx, const void *const *d_coeffs, const size_t *ns, size_t batch,
                                  uint64_t *outs);
/* A list of independent commitments and openings of resident polynomials in one call (the groups
 * of protocols.rs:191-195, 652-667, 761-771): task k commits to polynomial k (open[k] = 0) or opens it at
 * us[4k..4k+4) (open[k] = 1).  outs = m x 18 (commitment or proof), vs = m x 4 (f(u), zero for a
 * commitment); m <= 64.  Queued back to back on the context's stream: every multi-scalar multiplication of the
 * list stops after its bucket accumulation, and the serial tails (bucket fix-up and reduction: chains of
 * dependent point additions that leave most of the GPU idle) of up to 16 of them then run side by side in one
 * set of launches (so do plonk_hip_kzg_commit_batch, _commit_many_dev and _open_many{,_dev}). */
int plonk_hip_kzg_commit_open_many_dev(plonk_hip_ctx *ctx, const void *const *d_coeffs, const size_t *ns,
                                       const uint8_t *open, const uint64_t *us /* m x 4 */, size_t m,
                                       uint64_t *vs, uint64_t *outs);
/* f(wX) in place (`compose_polynomials(f, [0, w])`, polynomials.rs:80-113). */
int plonk_hip_fr_scale_by_powers_dev(plonk_hip_ctx *ctx, void *d_coeffs, size_t n, const uint64_t w[4]);
/* `&a * &b`: d_out receives na + nb - 1 coefficients and must not overlap a or b. */
int plonk_hip_fr_poly_mul_dev(plonk_hip_ctx *ctx, const void *d_a, size_t na, const void *d_b, size_t nb, void *d_out);
/* plonk_hip_fr_linear_combination on resident vectors; d_out (max(nx, ny) elements) may be d_x or d_y. */
int plonk_hip_fr_linear_combination_dev(plonk_hip_ctx *ctx, const void *d_x, size_t nx, const void *d_y, size_t ny,
                                        const uint64_t a[4], const uint64_t b[4], const uint64_t c[4], int broadcast,
                                        void *d_out);
/* `compute_q_zero_test_from_roots` (protocols.rs:169-181): d_q receives n - m coefficients, no overlap. */
int plonk_hip_fr_div_by_roots_dev(plonk_hip_ctx *ctx, const void *d_f, size_t n, const void *d_roots, size_t m,
                                  void *d_q);
/* `compute_q_zero_test` (protocols.rs:157-166): quotient by X^k - 1, n - k coefficients, no overlap. */
int plonk_hip_fr_div_vanishing_dev(plonk_hip_ctx *ctx, const void *d_f, size_t n, size_t k, void *d_q);
/* *holds = 1 when values[perm[i]] == values[i] for all i < n: the wiring assertion T(W(y)) = T(y) on
 * Omega (prover/part4.rs:36-42) on the trace values.  d_values resident, perm (n indices) on the host. */
int plonk_hip_fr_respects_permutation_dev(plonk_hip_ctx *ctx, const void *d_values, const uint32_t *perm, size_t n,
                                          int *holds);
/* *is_zero = 1 when all n resident elements are zero (the remainder of an exact division, the values
 * of a constraint polynomial on its rows: the `assert!`s of prover/part3.rs:61-64, part4.rs:58). */
int plonk_hip_fr_is_zero_dev(plonk_hip_ctx *ctx, const void *d_x, size_t n, int *is_zero);
/* d_num[i] <- prod_{j<=i} d_num[j] / d_den[j] (protocols.rs:345-353, 571-588); d_den is overwritten with
 * the inverses.  PLONK_HIP_ERR_ZERO_DENOMINATOR like the host form. */
int plonk_hip_fr_ratio_prefix_product_dev(plonk_hip_ctx *ctx, void *d_num, void *d_den, size_t n);
/* plonk_hip_fr_ratio_prefix_product_dev for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same arithmetic on the same kernels, but the call allocates and frees nothing and never
 * waits: the partial products of the scan live in a static array of the library's code object (270 336
 * bytes, held by every GPU the library runs on until the process ends), and a zero denominator is reported
 * through *d_zero_flag (a device int the caller owns: 0 = fine, 1 = some d_den[i] was zero and d_num is
 * unspecified) instead of an error code.  n <= 2^24, else PLONK_HIP_ERR_INVALID_ARG.  All contexts on one
 * GPU share that GPU's array; the library orders their uses of it with one event per GPU (waited for ahead of the
 * scan, recorded behind it), so calls on different contexts of one GPU do not race -- they serialise.  (Under
 * stream capture that wait would pull the other context's stream into the capture: capture with one context per
 * GPU.)  Work the caller queues on streams of its own is not covered. */
int plonk_hip_fr_ratio_prefix_product_dev_noalloc(plonk_hip_ctx *ctx, void *d_num, void *d_den, size_t n,
                                                  int *d_zero_flag);

/* ----------------------------------------------------------------------------------------
 * Several GPUs behind one handle (one process)
 * --------------------------------------------------------------------------------------
 * `kzg_commit(&gp, &f)` and `kzg_evaluate(&gp, &f, u)` (kzg.rs:48-88) for a `GlobalParameters` too large or a
 * commitment too slow for one GPU: the SRS is split by point range over `n_devices` sub-contexts (devices[g]
 * may repeat a device), each device receives its coefficient slice over its own PCIe link, runs the whole local
 * multi-scalar multiplication and sends its 144-byte partial to the first device (peer copy, xGMI inside a
 * node), which adds the partials up.  For an opening the suffix scan of kzg.rs:65-76 runs slice by slice with
 * the carries chained on the host.  Same data formats, error codes and blocking semantics as the single-GPU
 * entry points; a handle may be used from any thread (calls are serialised).  Every device holds at least one
 * SRS point (n >= n_devices). */
typedef struct plonk_hip_multi plonk_hip_multi;
int plonk_hip_multi_create(const int *devices, int n_devices, plonk_hip_multi **out);
void plonk_hip_multi_destroy(plonk_hip_multi *m);
const char *plonk_hip_multi_last_error(plonk_hip_multi *m);
int plonk_hip_multi_device_count(plonk_hip_multi *m);
int plonk_hip_multi_set_window_bits(plonk_hip_multi *m, int c);
/* `GlobalParameters.tau_powers_g1` (kzg.rs:17-21) / `kzg_setup` (kzg.rs:24-45), as the single-GPU forms. */
int plonk_hip_multi_srs_load_projective(plonk_hip_multi *m, const uint64_t *pts /* n x 18 */, size_t n);
int plonk_hip_multi_kzg_setup(plonk_hip_multi *m, const uint64_t tau[4], size_t degree, uint64_t *out_pts);
int plonk_hip_multi_srs_size(plonk_hip_multi *m, size_t *n);
/* SRS range [first, first + count) held by sub-context `index`. */
int plonk_hip_multi_srs_range(plonk_hip_multi *m, int index, size_t *first, size_t *count);
/* `kzg_commit` (kzg.rs:48-60), `kzg_evaluate` (kzg.rs:63-88) and the batched commitments
 * (protocols.rs:191-195, 652-653, 761). */
int plonk_hip_multi_kzg_commit(plonk_hip_multi *m, const uint64_t *coeffs /* n x 4 */, size_t n, uint64_t out[18]);
int plonk_hip_multi_kzg_commit_batch(plonk_hip_multi *m, const uint64_t *const *coeffs, const size_t *ns, size_t batch,
                                     uint64_t *outs);
int plonk_hip_multi_kzg_open(plonk_hip_multi *m, const uint64_t *coeffs /* n x 4 */, size_t n, const uint64_t u[4],
                             uint64_t v[4], uint64_t proof[18]);

#ifdef __cplusplus
}
#endif
#endif /* PLONK_HIP_H */
