// Library context: one GPU, one stream, the resident SRS window tables and a growable
// device workspace shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <string>

#include "../../include/plonk_hip.h"
#include "g1.cuh"

namespace plonk {

// A device buffer that only ever grows.
struct DevBuf {
    void *ptr = nullptr;
    size_t bytes = 0;
};

struct MsmPlan {
    int c = 0;           // bits per window
    int nwin = 0;        // windows = ceil(255 / c)
    uint32_t nbuckets = 0;  // 2^(c-1)
};

enum FrontEnd { kFrontEndAuto = 0, kFrontEndScalars, kFrontEndPairs };
// from this many scalars on the first sort pass is made from the scalars (where its shape allows)
static constexpr size_t kFrontEndScalarsMin = (size_t)1 << 21;

// Bits of the context's status word (host memory the kernels can write; checked at every host wait).
static constexpr uint32_t kStatusKeysOutOfOrder = 1u;

enum MsmPhase { PH_DIGITS = 0, PH_SORT, PH_BOUNDS, PH_ACCUM, PH_FIXUP, PH_REDUCE, PH_COUNT };

// One MSM whose tail is still to run (device pointers; bucket arrays are G1Xyzz28, see g1_28.cuh).
static constexpr int kMaxTailSlots = 16;
struct TailDesc {
    const uint32_t *start;      // bucket bounds
    const void *part;           // segment partials
    void *buckets;              // bucket sums (whole buckets already there)
    const uint32_t *queue;      // buckets whose partials take whole blocks, in chunks (msm.hip: k_classify)
    void *chunk_out;            // chunk sums of the buckets of several chunks
    uint32_t max_joins, max_items, chunk_parts;
    void *red;                  // row sums, column sums, bit sums of the reduction
    void *out;                  // G1Jac result, or null: stop after the bucket sums
    uint32_t seg_len, g_log, per_lane, group_blocks;
};
struct TailBatch {
    TailDesc d[kMaxTailSlots];
};
struct TailSlotBufs {
    DevBuf bucket_start, buckets, seg_part, big_list, chunk_out;
};

}  // namespace plonk

struct plonk_hip_ctx {
    int device = 0;
    int num_cus = 256;  // compute units of the device (read when the context is created)
    hipStream_t stream = nullptr;
    std::string last_error;
    std::recursive_mutex mu;  // serialises the entry points that use this context
    int window_bits_override = 0;
    // how an MSM gets its sorted (bucket, entry) pairs: by size, or one way always (PLONK_HIP_FRONT_END =
    // "scalars" / "pairs" when the context is created; for tests and measurements)
    int front_end = plonk::kFrontEndAuto;
    // the sort ranks keys with one LDS add-with-return each (checked by sort_selftest when the context is created;
    // PLONK_HIP_SORT_RANK=ballot keeps the form that relies on nothing)
    bool lds_rank_in_lane_order = false;
    // measurement override (plonk_hip_bench_msm_tuning): segment length of the accumulation (0: the library's choice)
    uint32_t tune_seg_len = 0;
    int tune_pair_loads = 0;  // 0: the library's choice, 1: one position at a time, 2: a group at a time through LDS
    bool debug_skip_sort = false;  // fault injection (plonk_hip_bench_skip_sort): hand the accumulation unsorted pairs

    // SRS: `table` holds plan.nwin levels of srs_n affine points in the 28-bit limb form
    // (G1Affine28, 128 B each); level w, point i = [2^(c*w)] P_i at table[w * srs_n + i].
    size_t srs_n = 0;
    plonk::MsmPlan plan;
    plonk::DevBuf table;

    // MSM workspace
    plonk::DevBuf keys_a, keys_b, vals_a, vals_b, sort_tmp;
    plonk::DevBuf bucket_start, buckets, seg_part, big_list, chunk_out, red_a, red_b, red_lw, red_sum_a, red_sum_b;
    // A batch of MSMs (several commitments / openings queued by one call) defers the serial tails: every
    // MSM of the batch stops after its accumulation, leaving its bucket bounds, whole buckets and segment
    // partials in a slot of its own, and ONE set of tail launches (fix-up, row / column sums, bit sums,
    // final chain) then finishes all pending slots side by side -- the tails are chains of dependent
    // point additions that leave most of the GPU idle, so n of them take about as long as one.
    // Everything stays on the context's single stream.  Slot 0 is the set of buffers above.
    plonk::TailSlotBufs tail_bufs[plonk::kMaxTailSlots - 1];
    plonk::TailDesc pending[plonk::kMaxTailSlots];
    int npending = 0;
    int entry_depth = 0;  // nesting of C entry points on this context (under `mu`)
    plonk::DevBuf scalars_stage, out_stage, stage_c;
    // generic scratch for SRS building and polynomial routines
    plonk::DevBuf scratch_a, scratch_b, scratch_c;
    // workspaces of the polynomial product, product tree, series division and interpolation
    plonk::DevBuf mul_ws, tree_ws, div_ws, interp_ws, ntt_ws, ntt3_ws;
    // two levels of the SRS in 12 x 32-bit affine form while the window tables are being built
    plonk::DevBuf level_a, level_b;
    // fixed-base table of the generator for kzg_setup (32 windows x 256 affine multiples)
    plonk::DevBuf gen_table;
    // twiddles for the NTT: per log_n forward and inverse tables
    plonk::DevBuf twiddle[2][33];
    // per-element factor tables of the strided passes, [kind][s1][s0] (ntt.hip: CorrKind), and of coset transforms,
    // [inverse][log_n] (one shift cached per size and direction)
    plonk::DevBuf ntt_corr[4][33][33];
    plonk::DevBuf coset_tab[2][33];
    plonk::Fr coset_shift[2][33];
    bool coset_ok[2][33] = {};
    bool ntt_lds_opt_in = false;  // large dynamic LDS enabled for the transform kernels on this device

    // One word of mapped host memory that kernels set when they find their input inconsistent (kStatus* bits):
    // no copy and no wait of its own, the host looks at it after the waits it performs anyway.
    uint32_t *status_host = nullptr, *status_dev = nullptr;

    // this device's copy of the library's static scratch array (capi.hip), resolved at creation
    void *static_scratch = nullptr;

    hipStream_t copy_stream = nullptr;  // host-to-device staging of large inputs beside the compute stream
    hipEvent_t timer_ev[2] = {};  // plonk_hip_timer_start / _stop

    bool profile = false;
    hipEvent_t ev[plonk::PH_COUNT + 1] = {};
    float phase_ms[plonk::PH_COUNT] = {};
};

namespace plonk {

#define PLONK_HIP_TRY(ctx, expr)                                                         \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);       \
            return _e == hipErrorOutOfMemory ? PLONK_HIP_ERR_OUT_OF_MEMORY : PLONK_HIP_ERR_HIP; \
        }                                                                                \
    } while (0)

#define PLONK_TRY(expr)          \
    do {                         \
        int _rc = (expr);        \
        if (_rc != 0) return _rc; \
    } while (0)

// hipStreamSynchronize on the context's stream, then the status word: PLONK_HIP_ERR_INTERNAL (and the word
// cleared) if a kernel reported inconsistent input since the last look.
int stream_sync(plonk_hip_ctx *ctx);
// the look at the status word alone, for callers that have waited for the stream's work some other way
int status_check(plonk_hip_ctx *ctx);

// Grow `b` to at least `bytes` (contents are not preserved).
int buf_reserve(plonk_hip_ctx *ctx, DevBuf &b, size_t bytes);
void buf_release(DevBuf &b);

// msm.hip
MsmPlan msm_choose_plan(size_t srs_n, int override_bits);
// d_scalars: n Montgomery scalars in device memory; d_out: one G1Jac in device memory.
// overlap: defer the tail of this MSM (its result is written when msm_join runs the pending tails; the
// caller calls msm_join before it reads d_out or returns).
// first: scalar i multiplies SRS point first + i (a slice of a longer polynomial).
int msm_run_dev(plonk_hip_ctx *ctx, const Fr *d_scalars, size_t n, G1Jac *d_out, void *d_buckets_out = nullptr,
                bool overlap = false, size_t first = 0);
// runs the tails of all pending MSMs (one set of launches for all of them)
int msm_join(plonk_hip_ctx *ctx);
int msm_reduce_slice_dev(plonk_hip_ctx *ctx, const void *d_parts, size_t nparts, size_t first, size_t count,
                         G1Jac *d_out);
size_t msm_bucket_bytes();
// d_out = sum of n Jacobian points in device memory (the gathered per-GPU partial results)
int g1_sum_dev(plonk_hip_ctx *ctx, const G1Jac *d_pts, size_t n, G1Jac *d_out);

// sort.hip
// decides how the scatter kernels rank keys on this device (sets ctx->lds_rank_in_lane_order)
int sort_selftest(plonk_hip_ctx *ctx);
size_t sort_pairs_temp_bytes(size_t n, int end_bit);
int sort_pairs_passes(int end_bit);
// result in (keys_b, vals_b) after an odd number of passes, in (keys_a, vals_a) after an even one.
int sort_pairs(plonk_hip_ctx *ctx, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, size_t n,
               int end_bit, void *temp, size_t temp_bytes);
// The same sort, by the c - 1 bucket bits of the keys, with the first pass made from the scalars (the pairs of
// scalar i: window w -> key = |digit| - 1 or nbuckets for digit 0, val = w * srs_n + first + i with the sign in
// bit 31); temp as for sort_pairs(n * nwin pairs, c - 1); result where sort_pairs leaves it.
bool sort_pairs_of_scalars_applies(size_t n, int c, int nwin);
int sort_pairs_of_scalars(plonk_hip_ctx *ctx, const Fr *d_scalars, size_t n, size_t srs_n, size_t first, int c, int nwin,
                          uint32_t nbuckets, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b,
                          void *temp, size_t temp_bytes);

// microbench.hip
int bench_mad_rate(plonk_hip_ctx *ctx, double *mads_per_second);
int bench_mad_mix(plonk_hip_ctx *ctx, int extra, int blocks_per_cu, double *mads_per_second, double *clock_ghz);
int bench_fq_mul_rate(plonk_hip_ctx *ctx, int variant, int blocks_per_cu, double *muls_per_second);
int bench_op_rate(plonk_hip_ctx *ctx, int op, double *ops_per_second);
int bench_gather_rate(plonk_hip_ctx *ctx, size_t table_bytes, double *bytes_per_second);

// srs.hip
int srs_install_affine(plonk_hip_ctx *ctx, size_t n);  // level 0 already in ctx->table
int srs_load_projective(plonk_hip_ctx *ctx, const uint64_t *pts, size_t n);
int srs_setup(plonk_hip_ctx *ctx, const uint64_t tau[4], size_t first, size_t n, uint64_t *out_pts);
int srs_read(plonk_hip_ctx *ctx, size_t first, size_t count, uint64_t *out);
int srs_load_compressed(plonk_hip_ctx *ctx, const uint8_t *bytes, size_t n, bool check_subgroup, size_t *bad_index);

// codec.hip
int g1_decompress_dev(plonk_hip_ctx *ctx, const uint8_t *d_bytes, size_t n, bool check_subgroup, G1Affine *d_affine,
                      size_t *bad_index);
int g1_affine_to_jac_dev(plonk_hip_ctx *ctx, const G1Affine *d_affine, size_t n, G1Jac *d_out);
int g1_compress_dev(plonk_hip_ctx *ctx, const G1Jac *d_pts, size_t n, uint8_t *d_bytes);

}  // namespace plonk
