// C ABI of libplonk_hip (see include/plonk_hip.h for the contract of every entry point).
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/plonk_hip_bench.h"
#include "context.hpp"
#include "poly.hpp"

using namespace plonk;

namespace plonk {

int buf_reserve(plonk_hip_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return PLONK_HIP_OK;
    if (b.ptr) {
        // kernels already queued may still read the old allocation
        PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        PLONK_HIP_TRY(ctx, hipFree(b.ptr));
        b.ptr = nullptr;
        b.bytes = 0;
    }
    size_t rounded = (bytes + 255) & ~(size_t)255;
    PLONK_HIP_TRY(ctx, hipMalloc(&b.ptr, rounded));
    b.bytes = rounded;
    return PLONK_HIP_OK;
}

int status_check(plonk_hip_ctx *ctx) {
    if (!ctx->status_host) return PLONK_HIP_OK;
    const uint32_t status = *(volatile uint32_t *)ctx->status_host;
    if (!status) return PLONK_HIP_OK;
    *(volatile uint32_t *)ctx->status_host = 0;
    if (status & kStatusKeysOutOfOrder) {
        // the one assumption the fast sort makes did not hold: from now on the form that makes none
        ctx->lds_rank_in_lane_order = false;
        ctx->last_error = "bucket sort left keys out of order; results since the last wait are invalid "
                          "(the context now ranks keys by ballots)";
    }
    return PLONK_HIP_ERR_INTERNAL;
}

int stream_sync(plonk_hip_ctx *ctx) {
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return status_check(ctx);
}

void buf_release(DevBuf &b) {
    if (b.ptr) (void)hipFree(b.ptr);
    b.ptr = nullptr;
    b.bytes = 0;
}

}  // namespace plonk

// Scratch of plonk_hip_fr_ratio_prefix_product_dev_noalloc: the block totals of the running product
// for up to kStaticMaxElems elements.  One copy per GPU, in the code object's own memory; every
// context created on a GPU resolves that GPU's address once, below.
static constexpr size_t kStaticMaxElems = (size_t)1 << 24;
static constexpr size_t kStaticScratchBytes = 8448 * sizeof(Fr);  // 8198 elements needed at 2^24
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

static std::mutex g_static_mu;
static void *g_static_ptr[256];  // per device ordinal
// Every context on a GPU shares that GPU's array, each with a stream of its own: an event per GPU, recorded behind
// each use and waited for ahead of the next, puts the uses in one order (no allocation per call: the event is made
// when the first context on the GPU is created).
static hipEvent_t g_static_done[256];

// The device must be current on the calling thread.
static hipError_t resolve_static_scratch(int device, void **out) {
    if (device < 0 || device >= 256) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> guard(g_static_mu);
    if (!g_static_ptr[device]) {
        void *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        e = hipEventCreateWithFlags(&g_static_done[device], hipEventDisableTiming);
        if (e != hipSuccess) return e;
        g_static_ptr[device] = ptr;
    }
    *out = g_static_ptr[device];
    return hipSuccess;
}

static int bind(plonk_hip_ctx *ctx) {
    PLONK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return PLONK_HIP_OK;
}

// Every entry point that touches the GPU holds the context's lock for its whole duration and
// makes the context's device current on the calling thread.
// Leaving the outermost entry point drops whatever deferred MSM tails are still pending: on the normal paths
// msm_join has run them by then; after an error in the middle of a batch they must not run later, when
// the output addresses they hold may no longer be valid.
struct EntryScope {
    plonk_hip_ctx *ctx;
    explicit EntryScope(plonk_hip_ctx *c) : ctx(c) { ctx->entry_depth++; }
    ~EntryScope() {
        if (--ctx->entry_depth == 0) ctx->npending = 0;
    }
};
#define PLONK_ENTER(ctx)                                        \
    if (!(ctx)) return PLONK_HIP_ERR_INVALID_ARG;               \
    std::lock_guard<std::recursive_mutex> _plonk_guard((ctx)->mu); \
    EntryScope _plonk_scope(ctx);                               \
    PLONK_TRY(bind(ctx))

static Fr load_fr(const uint64_t v[4]) {
    Fr r;
    memcpy(r.l, v, sizeof(r.l));
    return r;
}

extern "C" {

int plonk_hip_ctx_create(int device, plonk_hip_ctx **out) {
    if (!out) return PLONK_HIP_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return PLONK_HIP_ERR_HIP;
    plonk_hip_ctx *ctx = new (std::nothrow) plonk_hip_ctx();
    if (!ctx) return PLONK_HIP_ERR_OUT_OF_MEMORY;
    ctx->device = device;
    if (const char *fe = std::getenv("PLONK_HIP_FRONT_END")) {
        if (!std::strcmp(fe, "scalars")) ctx->front_end = kFrontEndScalars;
        if (!std::strcmp(fe, "pairs")) ctx->front_end = kFrontEndPairs;
    }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) {
        delete ctx;
        return PLONK_HIP_ERR_HIP;
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
        ctx->num_cus = cus;
    for (int i = 0; i <= PH_COUNT; i++) {
        if (hipEventCreate(&ctx->ev[i]) != hipSuccess) {
            delete ctx;
            return PLONK_HIP_ERR_HIP;
        }
    }
    if (hipHostMalloc((void **)&ctx->status_host, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&ctx->status_dev, ctx->status_host, 0) != hipSuccess) {
        plonk_hip_ctx_destroy(ctx);
        return PLONK_HIP_ERR_HIP;
    }
    *ctx->status_host = 0;
    if (resolve_static_scratch(device, &ctx->static_scratch) != hipSuccess) {
        plonk_hip_ctx_destroy(ctx);
        return PLONK_HIP_ERR_HIP;
    }
    const char *rank = std::getenv("PLONK_HIP_SORT_RANK");
    if (!(rank && !std::strcmp(rank, "ballot")) && sort_selftest(ctx) != PLONK_HIP_OK) {
        plonk_hip_ctx_destroy(ctx);
        return PLONK_HIP_ERR_HIP;
    }
    *out = ctx;
    return PLONK_HIP_OK;
}

void plonk_hip_ctx_destroy(plonk_hip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = {&ctx->table,     &ctx->keys_a,    &ctx->keys_b,       &ctx->vals_a,    &ctx->vals_b,
                      &ctx->sort_tmp,  &ctx->bucket_start, &ctx->buckets,   &ctx->seg_part,  &ctx->big_list, &ctx->chunk_out, &ctx->red_a,
                      &ctx->red_b,     &ctx->red_lw,    &ctx->red_sum_a,    &ctx->red_sum_b, &ctx->scalars_stage,
                      &ctx->out_stage, &ctx->scratch_a, &ctx->scratch_b,    &ctx->scratch_c, &ctx->gen_table,      &ctx->level_a,   &ctx->level_b,
                      &ctx->stage_c,   &ctx->mul_ws,    &ctx->tree_ws,   &ctx->div_ws,       &ctx->interp_ws, &ctx->ntt_ws, &ctx->ntt3_ws};
    for (DevBuf *b : bufs) buf_release(*b);
    for (int d = 0; d < 2; d++)
        for (int i = 0; i < 33; i++) {
            buf_release(ctx->twiddle[d][i]);
            buf_release(ctx->coset_tab[d][i]);
        }
    for (int kind = 0; kind < 4; kind++)
        for (int i = 0; i < 33; i++)
            for (int j = 0; j < 33; j++) buf_release(ctx->ntt_corr[kind][i][j]);
    for (int i = 0; i <= PH_COUNT; i++)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    for (int i = 0; i < 2; i++)
        if (ctx->timer_ev[i]) (void)hipEventDestroy(ctx->timer_ev[i]);
    for (auto &t : ctx->tail_bufs) {
        buf_release(t.bucket_start);
        buf_release(t.buckets);
        buf_release(t.seg_part);
        buf_release(t.big_list);
        buf_release(t.chunk_out);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->status_host) (void)hipHostFree(ctx->status_host);
    delete ctx;
}

const char *plonk_hip_strerror(int code) {
    switch (code) {
        case PLONK_HIP_OK: return "ok";
        case PLONK_HIP_ERR_SRS_TOO_SHORT:
            return "Length of tau_powers_g1 must be at least equal to degree of polynomial + 1";
        case PLONK_HIP_ERR_EMPTY_POLY: return "empty coefficient vector (the reference panics on unwrap)";
        case PLONK_HIP_ERR_ZERO_DENOMINATOR: return "inverse of zero";
        case PLONK_HIP_ERR_INVALID_ARG: return "invalid argument";
        case PLONK_HIP_ERR_NO_SRS: return "no SRS loaded in this context";
        case PLONK_HIP_ERR_HIP: return "HIP runtime error";
        case PLONK_HIP_ERR_OUT_OF_MEMORY: return "out of device memory";
        case PLONK_HIP_ERR_NOT_DIVISIBLE: return "polynomial is not divisible by the vanishing polynomial";
        case PLONK_HIP_ERR_INVALID_POINT: return "Failed to deserialize G1";
        case PLONK_HIP_ERR_INTERNAL: return "inconsistent intermediate data on the GPU; see plonk_hip_last_error()";
        default: return "unknown error";
    }
}

const char *plonk_hip_last_error(plonk_hip_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int plonk_hip_ctx_set_window_bits(plonk_hip_ctx *ctx, int c) {
    if (!ctx || (c != 0 && (c < 2 || c > 24))) return PLONK_HIP_ERR_INVALID_ARG;
    ctx->window_bits_override = c;
    return PLONK_HIP_OK;
}

void *plonk_hip_ctx_stream(plonk_hip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int plonk_hip_sync(plonk_hip_ctx *ctx) {
    PLONK_ENTER(ctx);
    return stream_sync(ctx);
}

int plonk_hip_malloc(plonk_hip_ctx *ctx, size_t bytes, void **dptr) {
    PLONK_ENTER(ctx);
    if (!dptr) return PLONK_HIP_ERR_INVALID_ARG;
    PLONK_HIP_TRY(ctx, hipMalloc(dptr, bytes ? bytes : 16));
    return PLONK_HIP_OK;
}

int plonk_hip_free(plonk_hip_ctx *ctx, void *dptr) {
    PLONK_ENTER(ctx);
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    PLONK_HIP_TRY(ctx, hipFree(dptr));
    return PLONK_HIP_OK;
}

int plonk_hip_memcpy_h2d(plonk_hip_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    PLONK_ENTER(ctx);
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PLONK_HIP_OK;
}

int plonk_hip_memcpy_d2h(plonk_hip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    PLONK_ENTER(ctx);
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return stream_sync(ctx);
}

// ---------------------------------------------------------------------------- SRS
int plonk_hip_srs_load_projective(plonk_hip_ctx *ctx, const uint64_t *pts, size_t n) {
    PLONK_ENTER(ctx);
    return srs_load_projective(ctx, pts, n);
}

int plonk_hip_srs_load_compressed(plonk_hip_ctx *ctx, const uint8_t *bytes, size_t n, int check_subgroup,
                                  size_t *bad_index) {
    PLONK_ENTER(ctx);
    return srs_load_compressed(ctx, bytes, n, check_subgroup != 0, bad_index);
}

int plonk_hip_g1_decompress(plonk_hip_ctx *ctx, const uint8_t *bytes, size_t n, int check_subgroup, uint64_t *out_pts,
                            size_t *bad_index) {
    PLONK_ENTER(ctx);
    if ((!bytes || !out_pts) && n) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) return PLONK_HIP_OK;
    PLONK_TRY(buf_reserve(ctx, ctx->scalars_stage, n * 48));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, n * sizeof(G1Affine)));
    PLONK_TRY(buf_reserve(ctx, ctx->stage_c, n * sizeof(G1Jac)));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(ctx->scalars_stage.ptr, bytes, n * 48, hipMemcpyHostToDevice, ctx->stream));
    PLONK_TRY(g1_decompress_dev(ctx, (const uint8_t *)ctx->scalars_stage.ptr, n, check_subgroup != 0,
                                (G1Affine *)ctx->scratch_c.ptr, bad_index));
    PLONK_TRY(g1_affine_to_jac_dev(ctx, (const G1Affine *)ctx->scratch_c.ptr, n, (G1Jac *)ctx->stage_c.ptr));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(out_pts, ctx->stage_c.ptr, n * sizeof(G1Jac), hipMemcpyDeviceToHost, ctx->stream));
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PLONK_HIP_OK;
}

int plonk_hip_g1_compress(plonk_hip_ctx *ctx, const uint64_t *pts, size_t n, uint8_t *bytes) {
    PLONK_ENTER(ctx);
    if ((!pts || !bytes) && n) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) return PLONK_HIP_OK;
    PLONK_TRY(buf_reserve(ctx, ctx->stage_c, n * sizeof(G1Jac)));
    PLONK_TRY(buf_reserve(ctx, ctx->scalars_stage, n * 48));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_c.ptr, pts, n * sizeof(G1Jac), hipMemcpyHostToDevice, ctx->stream));
    PLONK_TRY(g1_compress_dev(ctx, (const G1Jac *)ctx->stage_c.ptr, n, (uint8_t *)ctx->scalars_stage.ptr));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(bytes, ctx->scalars_stage.ptr, n * 48, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PLONK_HIP_OK;
}

int plonk_hip_srs_size(plonk_hip_ctx *ctx, size_t *n) {
    if (!ctx || !n) return PLONK_HIP_ERR_INVALID_ARG;
    *n = ctx->srs_n;
    return PLONK_HIP_OK;
}

int plonk_hip_srs_read(plonk_hip_ctx *ctx, size_t first, size_t count, uint64_t *out) {
    PLONK_ENTER(ctx);
    return srs_read(ctx, first, count, out);
}

int plonk_hip_kzg_setup(plonk_hip_ctx *ctx, const uint64_t tau[4], size_t degree, uint64_t *out_pts) {
    PLONK_ENTER(ctx);
    return srs_setup(ctx, tau, 0, degree + 1, out_pts);
}

int plonk_hip_kzg_setup_range(plonk_hip_ctx *ctx, const uint64_t tau[4], size_t first, size_t count,
                              uint64_t *out_pts) {
    PLONK_ENTER(ctx);
    return srs_setup(ctx, tau, first, count, out_pts);
}

int plonk_hip_g1_sum_dev(plonk_hip_ctx *ctx, const void *d_pts, size_t n, void *d_out) {
    PLONK_ENTER(ctx);
    if (!d_pts || !d_out) return PLONK_HIP_ERR_INVALID_ARG;
    return g1_sum_dev(ctx, (const G1Jac *)d_pts, n, (G1Jac *)d_out);
}

int plonk_hip_msm_bucket_layout(plonk_hip_ctx *ctx, size_t *nbuckets, size_t *bytes_per_bucket) {
    PLONK_ENTER(ctx);
    if (!nbuckets || !bytes_per_bucket) return PLONK_HIP_ERR_INVALID_ARG;
    if (ctx->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    *nbuckets = ctx->plan.nbuckets;
    *bytes_per_bucket = msm_bucket_bytes();
    return PLONK_HIP_OK;
}

int plonk_hip_msm_buckets_dev(plonk_hip_ctx *ctx, const void *d_coeffs, size_t n, void *d_buckets) {
    PLONK_ENTER(ctx);
    if (!d_coeffs || !d_buckets) return PLONK_HIP_ERR_INVALID_ARG;
    return msm_run_dev(ctx, (const Fr *)d_coeffs, n, nullptr, d_buckets);
}

int plonk_hip_msm_reduce_bucket_slice_dev(plonk_hip_ctx *ctx, const void *d_parts, size_t nparts, size_t first,
                                          size_t count, void *d_out) {
    PLONK_ENTER(ctx);
    if (!d_parts || !d_out) return PLONK_HIP_ERR_INVALID_ARG;
    return msm_reduce_slice_dev(ctx, d_parts, nparts, first, count, (G1Jac *)d_out);
}

// ---------------------------------------------------------------------------- commit / open
int plonk_hip_kzg_commit_dev(plonk_hip_ctx *ctx, const void *d_coeffs, size_t n, void *d_out) {
    PLONK_ENTER(ctx);
    if (!d_coeffs || !d_out) return PLONK_HIP_ERR_INVALID_ARG;
    return msm_run_dev(ctx, (const Fr *)d_coeffs, n, (G1Jac *)d_out);
}

// Host coefficients of a large polynomial go to the device in slices: while slice k + 1 crosses PCIe (copy
// stream; the host waits for it, the GPU does not) the GPU already sorts and accumulates slice k.  Every
// slice is a multi-scalar multiplication of its own against its SRS range with a deferred tail; the tails
// then run side by side and the partial results are added.  No GPU-side wait between the two streams: a
// slice's kernels are only queued once its copy has completed.
// (measured at 2^24 coefficients with equal slices: 2 slices 51 ms, 3: 50, 4: 49, 6: 51, 8: 52; one copy then
// one MSM: 58; short slices leave few points per bucket, which the accumulation pays for.  Four slices of
// an eighth, an eighth, a quarter and a half: 2.4 ms less than four equal ones.)
static constexpr size_t kSliceMin = (size_t)1 << 22;  // coefficients: below two such slices, one copy and one MSM
static constexpr size_t kSliceMax = 4;

static int commit_host_sliced(plonk_hip_ctx *ctx, const uint64_t *coeffs, size_t n, Fr *d_stage, G1Jac *d_parts,
                              G1Jac *d_out) {
    size_t slices = n / kSliceMin;
    if (slices > kSliceMax) slices = kSliceMax;
    if (slices < 2) {
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_stage, coeffs, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        return msm_run_dev(ctx, d_stage, n, d_out);
    }
    if (!ctx->copy_stream) PLONK_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    // the staging buffer may still be read by kernels of an earlier call
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // Slice lengths: the GPU idles during the first copy only, so with four slices the first two are short
    // (an eighth each, then a quarter and a half: every copy is shorter than the MSM it hides behind).
    size_t len_of[kSliceMax];
    // (whole multiples of 2048 except for the last: the length the sort's first pass from the scalars asks for)
    if (slices == kSliceMax) {
        len_of[0] = len_of[1] = (n / 8) & ~(size_t)2047;
        len_of[2] = (n / 4) & ~(size_t)2047;
        len_of[3] = n - len_of[0] - len_of[1] - len_of[2];
    } else {
        const size_t each = (n + slices - 1) / slices;
        for (size_t k = 0, left = n; k < slices; k++, left -= len_of[k - 1]) len_of[k] = left < each ? left : each;
    }
    size_t done = 0, k = 0;
    for (; k < slices && done < n; k++) {
        const size_t len = len_of[k];
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_stage + done, coeffs + 4 * done, len * sizeof(Fr), hipMemcpyHostToDevice,
                                          ctx->copy_stream));
        PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
        PLONK_TRY(msm_run_dev(ctx, d_stage + done, len, d_parts + k, nullptr, true, done));
        done += len;
    }
    PLONK_TRY(msm_join(ctx));
    return g1_sum_dev(ctx, d_parts, k, d_out);
}

int plonk_hip_kzg_commit(plonk_hip_ctx *ctx, const uint64_t *coeffs, size_t n, uint64_t out[18]) {
    PLONK_ENTER(ctx);
    if (!out || (!coeffs && n)) return PLONK_HIP_ERR_INVALID_ARG;
    if (ctx->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    if (n == 0) return PLONK_HIP_ERR_EMPTY_POLY;
    if (n > ctx->srs_n) return PLONK_HIP_ERR_SRS_TOO_SHORT;
    PLONK_TRY(buf_reserve(ctx, ctx->scalars_stage, n * sizeof(Fr)));
    PLONK_TRY(buf_reserve(ctx, ctx->out_stage, 64 * sizeof(G1Jac) + 4096));
    G1Jac *d_out = (G1Jac *)ctx->out_stage.ptr;
    PLONK_TRY(commit_host_sliced(ctx, coeffs, n, (Fr *)ctx->scalars_stage.ptr, d_out + 1, d_out));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(out, d_out, sizeof(G1Jac), hipMemcpyDeviceToHost, ctx->stream));
    return stream_sync(ctx);
}

int plonk_hip_kzg_commit_batch(plonk_hip_ctx *ctx, const uint64_t *const *coeffs, const size_t *ns, size_t batch,
                               uint64_t *outs) {
    PLONK_ENTER(ctx);
    if (!coeffs || !ns || !outs) return PLONK_HIP_ERR_INVALID_ARG;
    if (ctx->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    size_t total = 0;
    for (size_t b = 0; b < batch; b++) {
        if (ns[b] == 0 || !coeffs[b]) return PLONK_HIP_ERR_EMPTY_POLY;
        if (ns[b] > ctx->srs_n) return PLONK_HIP_ERR_SRS_TOO_SHORT;
        total += ns[b];
    }
    if (batch == 0) return PLONK_HIP_OK;
    // all coefficient vectors are staged back to back, the commitments are queued one after the
    // other on the stream, and the host waits once for the whole group
    PLONK_TRY(buf_reserve(ctx, ctx->scalars_stage, total * sizeof(Fr)));
    PLONK_TRY(buf_reserve(ctx, ctx->stage_c, batch * sizeof(G1Jac)));
    Fr *d_f = (Fr *)ctx->scalars_stage.ptr;
    G1Jac *d_out = (G1Jac *)ctx->stage_c.ptr;
    size_t off = 0;
    for (size_t b = 0; b < batch; b++) {
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_f + off, coeffs[b], ns[b] * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        off += ns[b];
    }
    off = 0;
    for (size_t b = 0; b < batch; b++) {
        PLONK_TRY(msm_run_dev(ctx, d_f + off, ns[b], d_out + b, nullptr, batch > 1));
        off += ns[b];
    }
    PLONK_TRY(msm_join(ctx));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(outs, d_out, batch * sizeof(G1Jac), hipMemcpyDeviceToHost, ctx->stream));
    return stream_sync(ctx);
}

// v_k = f(u_k) and the opening proofs for m points of one polynomial in device memory; the suffix scan
// of point k + 1 and its MSM run while the tail of MSM k finishes on the tail stream.
static int open_many_core(plonk_hip_ctx *ctx, const Fr *d_f, size_t n, const uint64_t *us, size_t m, uint64_t *vs,
                          uint64_t *proofs) {
    const size_t scr = suffix_scan_scratch_elems(n);
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, (n + scr) * sizeof(Fr)));
    PLONK_TRY(buf_reserve(ctx, ctx->out_stage, 64 * sizeof(G1Jac) + 4096));
    Fr *d_s = (Fr *)ctx->scratch_c.ptr;
    G1Jac *d_out = (G1Jac *)ctx->out_stage.ptr;
    for (size_t done = 0; done < m; done += 64) {
        const size_t group = m - done < 64 ? m - done : 64;
        for (size_t k = 0; k < group; k++) {
            // s[0] = f(u); s[1..n) = (f - f(u)) / (X - u)
            PLONK_TRY(suffix_scan(ctx, d_f, n, load_fr(us + 4 * (done + k)), d_s, d_s + n));
            PLONK_TRY(msm_run_dev(ctx, d_s + 1, n - 1, d_out + k, nullptr, group > 1));
            // the next scan overwrites d_s: fetch f(u) first (stream-ordered; the wait comes at the end)
            PLONK_HIP_TRY(ctx, hipMemcpyAsync(vs + 4 * (done + k), d_s, sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        }
        PLONK_TRY(msm_join(ctx));
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(proofs + 18 * done, d_out, group * sizeof(G1Jac), hipMemcpyDeviceToHost,
                                          ctx->stream));
        PLONK_TRY(stream_sync(ctx));
    }
    return PLONK_HIP_OK;
}

// One opening of a long polynomial in host memory, in slices like commit_host_sliced.  The scan
// s_j = f_j + u s_(j+1) runs from the top, so the slices are copied from the top down (the two short ones
// first); the scan of a slice takes the value the slice above ended with as one more coefficient behind its
// own (in the staging buffer that place holds the first coefficient of the slice above, which its scan has
// consumed by then), and the quotient coefficients s_j, j >= 1, of the slice go into a multi-scalar
// multiplication against the SRS points from j - 1 on, tail deferred.  v = s_0 comes out of the last scan.
static int open_host_sliced(plonk_hip_ctx *ctx, const uint64_t *coeffs, size_t n, const Fr &u, uint64_t v[4],
                            uint64_t proof[18]) {
    const size_t slices = kSliceMax;
    size_t len_of[kSliceMax];  // from the top: an eighth, an eighth, a quarter (whole multiples of 2048), the rest
    len_of[0] = len_of[1] = (n / 8) & ~(size_t)2047;
    len_of[2] = (n / 4) & ~(size_t)2047;
    len_of[3] = n - len_of[0] - len_of[1] - len_of[2];
    const size_t longest = len_of[3] + 1;
    PLONK_TRY(buf_reserve(ctx, ctx->scalars_stage, (n + 1) * sizeof(Fr)));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, (longest + suffix_scan_scratch_elems(longest)) * sizeof(Fr)));
    PLONK_TRY(buf_reserve(ctx, ctx->out_stage, 64 * sizeof(G1Jac) + 4096));
    Fr *d_f = (Fr *)ctx->scalars_stage.ptr, *d_s = (Fr *)ctx->scratch_c.ptr;
    G1Jac *d_out = (G1Jac *)ctx->out_stage.ptr, *d_parts = d_out + 1;
    if (!ctx->copy_stream) PLONK_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    // the staging buffer may still be read by kernels of an earlier call
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    size_t end = n;
    for (size_t k = 0; k < slices; k++) {
        const size_t len = len_of[k], begin = end - len;
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_f + begin, coeffs + 4 * begin, len * sizeof(Fr), hipMemcpyHostToDevice,
                                          ctx->copy_stream));
        PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
        const size_t carried = k ? 1 : 0;
        if (carried) {
            // s_end: the first value of the scan before this one (queued behind it, ahead of this slice's scan)
            PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_f + end, d_s, sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
        }
        PLONK_TRY(suffix_scan(ctx, d_f + begin, len + carried, u, d_s, d_s + len + carried));
        const size_t skip = begin == 0 ? 1 : 0;  // s_0 = f(u) is the value, not a coefficient of the quotient
        PLONK_TRY(msm_run_dev(ctx, d_s + skip, len - skip, d_parts + k, nullptr, true, begin + skip - 1));
        end = begin;
    }
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(v, d_s, sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    PLONK_TRY(msm_join(ctx));
    PLONK_TRY(g1_sum_dev(ctx, d_parts, slices, d_out));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(proof, d_out, sizeof(G1Jac), hipMemcpyDeviceToHost, ctx->stream));
    return stream_sync(ctx);
}

int plonk_hip_kzg_open_many(plonk_hip_ctx *ctx, const uint64_t *coeffs, size_t n, const uint64_t *us, size_t m,
                            uint64_t *vs, uint64_t *proofs) {
    PLONK_ENTER(ctx);
    if (!coeffs || !us || !vs || !proofs) return PLONK_HIP_ERR_INVALID_ARG;
    if (ctx->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    if (n < 2) return PLONK_HIP_ERR_EMPTY_POLY;
    if (n - 1 > ctx->srs_n) return PLONK_HIP_ERR_SRS_TOO_SHORT;
    if (m == 1 && n >= kSliceMax * kSliceMin) return open_host_sliced(ctx, coeffs, n, load_fr(us), vs, proofs);
    PLONK_TRY(buf_reserve(ctx, ctx->scalars_stage, n * sizeof(Fr)));
    Fr *d_f = (Fr *)ctx->scalars_stage.ptr;
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_f, coeffs, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    return open_many_core(ctx, d_f, n, us, m, vs, proofs);
}

int plonk_hip_kzg_open(plonk_hip_ctx *ctx, const uint64_t *coeffs, size_t n, const uint64_t u[4], uint64_t v[4],
                       uint64_t proof[18]) {
    return plonk_hip_kzg_open_many(ctx, coeffs, n, u, 1, v, proof);
}

int plonk_hip_bench_mad_rate(plonk_hip_ctx *ctx, double *mads_per_second) {
    PLONK_ENTER(ctx);
    if (!mads_per_second) return PLONK_HIP_ERR_INVALID_ARG;
    return bench_mad_rate(ctx, mads_per_second);
}

int plonk_hip_bench_fq_mul_rate(plonk_hip_ctx *ctx, int variant, int blocks_per_cu, double *muls_per_second) {
    PLONK_ENTER(ctx);
    if (!muls_per_second || blocks_per_cu < 1) return PLONK_HIP_ERR_INVALID_ARG;
    return bench_fq_mul_rate(ctx, variant, blocks_per_cu, muls_per_second);
}

int plonk_hip_bench_mad_mix(plonk_hip_ctx *ctx, int extra, int blocks_per_cu, double *mads_per_second,
                             double *clock_ghz) {
    PLONK_ENTER(ctx);
    if (!mads_per_second || !clock_ghz || blocks_per_cu < 1) return PLONK_HIP_ERR_INVALID_ARG;
    return bench_mad_mix(ctx, extra, blocks_per_cu, mads_per_second, clock_ghz);
}

int plonk_hip_bench_op_rate(plonk_hip_ctx *ctx, int op, double *ops_per_second) {
    PLONK_ENTER(ctx);
    if (!ops_per_second) return PLONK_HIP_ERR_INVALID_ARG;
    return bench_op_rate(ctx, op, ops_per_second);
}

int plonk_hip_bench_sort_rank_mode(plonk_hip_ctx *ctx) { return ctx && ctx->lds_rank_in_lane_order ? 1 : 0; }

int plonk_hip_bench_skip_sort(plonk_hip_ctx *ctx, int on) {
    if (!ctx) return PLONK_HIP_ERR_INVALID_ARG;
    ctx->debug_skip_sort = on != 0;
    return PLONK_HIP_OK;
}

int plonk_hip_bench_msm_tuning(plonk_hip_ctx *ctx, uint32_t seg_len, int pair_loads) {
    if (!ctx || pair_loads < 0 || pair_loads > 2) return PLONK_HIP_ERR_INVALID_ARG;
    ctx->tune_seg_len = seg_len;
    ctx->tune_pair_loads = pair_loads;
    return PLONK_HIP_OK;
}

int plonk_hip_bench_gather_rate(plonk_hip_ctx *ctx, size_t table_bytes, double *bytes_per_second) {
    PLONK_ENTER(ctx);
    if (!bytes_per_second) return PLONK_HIP_ERR_INVALID_ARG;
    return bench_gather_rate(ctx, table_bytes, bytes_per_second);
}

int plonk_hip_timer_start(plonk_hip_ctx *ctx) {
    PLONK_ENTER(ctx);
    for (int i = 0; i < 2; i++)
        if (!ctx->timer_ev[i]) PLONK_HIP_TRY(ctx, hipEventCreate(&ctx->timer_ev[i]));
    PLONK_HIP_TRY(ctx, hipEventRecord(ctx->timer_ev[0], ctx->stream));
    return PLONK_HIP_OK;
}

int plonk_hip_timer_stop(plonk_hip_ctx *ctx, float *ms) {
    PLONK_ENTER(ctx);
    if (!ms || !ctx->timer_ev[0] || !ctx->timer_ev[1]) return PLONK_HIP_ERR_INVALID_ARG;
    PLONK_HIP_TRY(ctx, hipEventRecord(ctx->timer_ev[1], ctx->stream));
    PLONK_HIP_TRY(ctx, hipEventSynchronize(ctx->timer_ev[1]));
    PLONK_HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->timer_ev[0], ctx->timer_ev[1]));
    return PLONK_HIP_OK;
}

int plonk_hip_profile(plonk_hip_ctx *ctx, int enable) {
    if (!ctx) return PLONK_HIP_ERR_INVALID_ARG;
    ctx->profile = enable != 0;
    return PLONK_HIP_OK;
}

int plonk_hip_msm_phase_ms(plonk_hip_ctx *ctx, float ms[PLONK_HIP_MSM_PHASES]) {
    PLONK_ENTER(ctx);
    if (!ms || !ctx->profile) return PLONK_HIP_ERR_INVALID_ARG;
    PLONK_HIP_TRY(ctx, hipEventSynchronize(ctx->ev[PH_COUNT]));
    for (int i = 0; i < PH_COUNT; i++) PLONK_HIP_TRY(ctx, hipEventElapsedTime(&ms[i], ctx->ev[i], ctx->ev[i + 1]));
    return PLONK_HIP_OK;
}

const char *plonk_hip_msm_phase_name(int i) {
    static const char *names[PH_COUNT] = {"digits", "sort", "bounds", "accumulate", "fixup", "reduce"};
    return (i >= 0 && i < PH_COUNT) ? names[i] : "";
}

int plonk_hip_msm_plan_for(size_t srs_n, int window_bits_override, int *window_bits, int *windows, size_t *buckets) {
    if (srs_n == 0) return PLONK_HIP_ERR_INVALID_ARG;
    MsmPlan pl = msm_choose_plan(srs_n, window_bits_override);
    if (window_bits) *window_bits = pl.c;
    if (windows) *windows = pl.nwin;
    if (buckets) *buckets = pl.nbuckets;
    return PLONK_HIP_OK;
}

int plonk_hip_ntt_plan(uint32_t log_n, int natural_order, int max_passes, uint32_t *first_stage,
                       uint32_t *end_stage) {
    if (!first_stage || !end_stage || max_passes < 1) return PLONK_HIP_ERR_INVALID_ARG;
    return ntt_plan_describe(log_n, natural_order != 0, max_passes, first_stage, end_stage);
}

int plonk_hip_msm_plan(plonk_hip_ctx *ctx, int *window_bits, int *windows, size_t *buckets) {
    if (!ctx) return PLONK_HIP_ERR_INVALID_ARG;
    if (ctx->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    if (window_bits) *window_bits = ctx->plan.c;
    if (windows) *windows = ctx->plan.nwin;
    if (buckets) *buckets = ctx->plan.nbuckets;
    return PLONK_HIP_OK;
}

// ---------------------------------------------------------------------------- polynomials
int plonk_hip_fr_scale_by_powers(plonk_hip_ctx *ctx, uint64_t *coeffs, size_t n, const uint64_t w[4]) {
    PLONK_ENTER(ctx);
    if ((!coeffs && n) || !w) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) return PLONK_HIP_OK;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, n * sizeof(Fr)));
    Fr *d = (Fr *)ctx->scratch_c.ptr;
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(d, coeffs, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    PLONK_TRY(scale_by_powers(ctx, d, n, 1, load_fr(w)));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(coeffs, d, n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PLONK_HIP_OK;
}

int plonk_hip_fr_eval(plonk_hip_ctx *ctx, const uint64_t *coeffs, size_t n, const uint64_t u[4], uint64_t out[4]) {
    PLONK_ENTER(ctx);
    if ((!coeffs && n) || !u || !out) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) {
        memset(out, 0, 32);  // the zero polynomial evaluates to zero
        return PLONK_HIP_OK;
    }
    size_t scr = suffix_scan_scratch_elems(n);
    PLONK_TRY(buf_reserve(ctx, ctx->scalars_stage, n * sizeof(Fr)));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, (n + scr) * sizeof(Fr)));
    Fr *d_f = (Fr *)ctx->scalars_stage.ptr;
    Fr *d_s = (Fr *)ctx->scratch_c.ptr;
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_f, coeffs, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    PLONK_TRY(suffix_scan(ctx, d_f, n, load_fr(u), d_s, d_s + n));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(out, d_s, sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PLONK_HIP_OK;
}

int plonk_hip_fr_div_vanishing(plonk_hip_ctx *ctx, const uint64_t *f, size_t n, size_t k, uint64_t *q) {
    PLONK_ENTER(ctx);
    if (!f || !q || k == 0) return PLONK_HIP_ERR_INVALID_ARG;
    if (n <= k) return PLONK_HIP_OK;
    PLONK_TRY(buf_reserve(ctx, ctx->scalars_stage, n * sizeof(Fr)));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, (n - k) * sizeof(Fr)));
    Fr *d_f = (Fr *)ctx->scalars_stage.ptr;
    Fr *d_q = (Fr *)ctx->scratch_c.ptr;
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_f, f, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    PLONK_TRY(div_vanishing(ctx, d_f, n, k, d_q));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(q, d_q, (n - k) * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PLONK_HIP_OK;
}

static int ntt_host(plonk_hip_ctx *ctx, uint64_t *data, uint32_t log_n, size_t batch, const uint64_t *shift,
                    int inverse) {
    PLONK_ENTER(ctx);
    if (!data || log_n > 30) return PLONK_HIP_ERR_INVALID_ARG;
    size_t total = ((size_t)1 << log_n) * batch;
    if (total == 0) return PLONK_HIP_OK;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, total * sizeof(Fr)));
    Fr *d = (Fr *)ctx->scratch_c.ptr;
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(d, data, total * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    Fr sh;
    if (shift) sh = load_fr(shift);
    PLONK_TRY(ntt_dev(ctx, d, log_n, batch, inverse != 0, shift ? &sh : nullptr));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(data, d, total * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PLONK_HIP_OK;
}

int plonk_hip_fr_ntt(plonk_hip_ctx *ctx, uint64_t *data, uint32_t log_n, size_t batch, int inverse) {
    return ntt_host(ctx, data, log_n, batch, nullptr, inverse);
}

int plonk_hip_fr_coset_ntt(plonk_hip_ctx *ctx, uint64_t *data, uint32_t log_n, size_t batch, const uint64_t shift[4],
                           int inverse) {
    if (!shift) return PLONK_HIP_ERR_INVALID_ARG;
    return ntt_host(ctx, data, log_n, batch, shift, inverse);
}

int plonk_hip_fr_ntt_mixed(plonk_hip_ctx *ctx, uint64_t *data, size_t n, size_t batch, int inverse) {
    PLONK_ENTER(ctx);
    if (!data && n * batch) return PLONK_HIP_ERR_INVALID_ARG;
    size_t total = n * batch;
    if (total == 0) return PLONK_HIP_OK;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, total * sizeof(Fr)));
    Fr *d = (Fr *)ctx->scratch_c.ptr;
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(d, data, total * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    PLONK_TRY(ntt_mixed_dev(ctx, d, n, batch, inverse != 0));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(data, d, total * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PLONK_HIP_OK;
}

int plonk_hip_fr_ntt_mixed_dev(plonk_hip_ctx *ctx, void *d_data, size_t n, size_t batch, int inverse) {
    PLONK_ENTER(ctx);
    if (!d_data) return PLONK_HIP_ERR_INVALID_ARG;
    return ntt_mixed_dev(ctx, (Fr *)d_data, n, batch, inverse != 0);
}

int plonk_hip_fr_ntt_dev(plonk_hip_ctx *ctx, void *d_data, uint32_t log_n, size_t batch, int inverse) {
    PLONK_ENTER(ctx);
    if (!d_data) return PLONK_HIP_ERR_INVALID_ARG;
    return ntt_dev(ctx, (Fr *)d_data, log_n, batch, inverse != 0, nullptr);
}

int plonk_hip_fr_coset_ntt_dev(plonk_hip_ctx *ctx, void *d_data, uint32_t log_n, size_t batch,
                               const uint64_t shift[4], int inverse) {
    PLONK_ENTER(ctx);
    if (!d_data || !shift) return PLONK_HIP_ERR_INVALID_ARG;
    Fr sh = load_fr(shift);
    return ntt_dev(ctx, (Fr *)d_data, log_n, batch, inverse != 0, &sh);
}

int plonk_hip_fr_quotient_coset_dev(plonk_hip_ctx *ctx, void *d_f_inout, uint32_t log_k, size_t batch,
                                    int *d_divisible) {
    PLONK_ENTER(ctx);
    if (!d_f_inout) return PLONK_HIP_ERR_INVALID_ARG;
    return quotient_coset_dev(ctx, (Fr *)d_f_inout, log_k, batch, d_divisible);
}

int plonk_hip_fr_quotient_coset(plonk_hip_ctx *ctx, const uint64_t *f, uint32_t log_k, size_t batch, uint64_t *q,
                                int *divisible) {
    PLONK_ENTER(ctx);
    if (!f || !q || log_k > 29) return PLONK_HIP_ERR_INVALID_ARG;
    if (batch == 0) return PLONK_HIP_OK;
    const size_t k = (size_t)1 << log_k, n = 2 * k;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, n * batch * sizeof(Fr)));
    PLONK_TRY(buf_reserve(ctx, ctx->out_stage, 64 * sizeof(G1Jac) + batch * sizeof(int)));
    Fr *d = (Fr *)ctx->scratch_c.ptr;
    int *d_flags = (int *)((char *)ctx->out_stage.ptr + 64 * sizeof(G1Jac));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(d, f, n * batch * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    PLONK_TRY(quotient_coset_dev(ctx, d, log_k, batch, d_flags));
    PLONK_HIP_TRY(ctx, hipMemcpy2DAsync(q, k * sizeof(Fr), d, n * sizeof(Fr), k * sizeof(Fr), batch,
                                        hipMemcpyDeviceToHost, ctx->stream));
    if (divisible) {
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(divisible, d_flags, batch * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    }
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PLONK_HIP_OK;
}


// ---------------------------------------------------------------------------- polynomial building blocks
// a zeroed int in device memory for the kernels' "zero denominator" report
static int fresh_flag(plonk_hip_ctx *ctx, int **d_flag) {
    PLONK_TRY(buf_reserve(ctx, ctx->out_stage, 64 * sizeof(G1Jac) + 4096));
    *d_flag = (int *)((char *)ctx->out_stage.ptr + 64 * sizeof(G1Jac));
    PLONK_HIP_TRY(ctx, hipMemsetAsync(*d_flag, 0, sizeof(int), ctx->stream));
    return PLONK_HIP_OK;
}
static int read_flag(plonk_hip_ctx *ctx, const int *d_flag, int *out) {
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(out, d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PLONK_HIP_OK;
}
static int upload(plonk_hip_ctx *ctx, DevBuf &buf, const uint64_t *src, size_t elems, size_t reserve_elems, Fr **d) {
    PLONK_TRY(buf_reserve(ctx, buf, (reserve_elems > elems ? reserve_elems : elems) * sizeof(Fr)));
    *d = (Fr *)buf.ptr;
    if (elems) PLONK_HIP_TRY(ctx, hipMemcpyAsync(*d, src, elems * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    return PLONK_HIP_OK;
}
static int download(plonk_hip_ctx *ctx, uint64_t *dst, const Fr *d, size_t elems) {
    if (elems) PLONK_HIP_TRY(ctx, hipMemcpyAsync(dst, d, elems * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PLONK_HIP_OK;
}

int plonk_hip_fr_eval_many(plonk_hip_ctx *ctx, const uint64_t *coeffs, size_t n, const uint64_t *points, size_t m,
                           uint64_t *out) {
    PLONK_ENTER(ctx);
    if ((!coeffs && n) || ((!points || !out) && m)) return PLONK_HIP_ERR_INVALID_ARG;
    if (m == 0) return PLONK_HIP_OK;
    Fr *d_f, *d_p, *d_o;
    PLONK_TRY(upload(ctx, ctx->scalars_stage, coeffs, n, 1, &d_f));
    PLONK_TRY(upload(ctx, ctx->scratch_c, points, m, 0, &d_p));
    PLONK_TRY(buf_reserve(ctx, ctx->stage_c, m * sizeof(Fr)));
    d_o = (Fr *)ctx->stage_c.ptr;
    PLONK_TRY(eval_points(ctx, d_f, n, d_p, m, d_o));
    return download(ctx, out, d_o, m);
}

int plonk_hip_fr_evaluate_over_omega(plonk_hip_ctx *ctx, const uint64_t *coeffs, size_t n, uint32_t log_k,
                                     uint64_t *out) {
    PLONK_ENTER(ctx);
    if ((!coeffs && n) || !out || log_k > 30) return PLONK_HIP_ERR_INVALID_ARG;
    const size_t k = (size_t)1 << log_k;
    Fr *d_f, *d_o;
    PLONK_TRY(upload(ctx, ctx->scalars_stage, coeffs, n, 1, &d_f));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, k * sizeof(Fr)));
    d_o = (Fr *)ctx->scratch_c.ptr;
    PLONK_TRY(fold_mod_vanishing(ctx, d_f, n, k, d_o));
    PLONK_TRY(ntt_dev(ctx, d_o, log_k, 1, false, nullptr));
    return download(ctx, out, d_o, k);
}

static int prefix_host(plonk_hip_ctx *ctx, uint64_t *data, size_t n, bool product) {
    PLONK_ENTER(ctx);
    if (!data && n) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) return PLONK_HIP_OK;
    Fr *d;
    PLONK_TRY(upload(ctx, ctx->scalars_stage, data, n, n + prefix_scan_scratch_elems(n), &d));
    PLONK_TRY(prefix_scan(ctx, d, n, product, d + n));
    return download(ctx, data, d, n);
}
int plonk_hip_fr_prefix_product(plonk_hip_ctx *ctx, uint64_t *data, size_t n) { return prefix_host(ctx, data, n, true); }
int plonk_hip_fr_prefix_sum(plonk_hip_ctx *ctx, uint64_t *data, size_t n) { return prefix_host(ctx, data, n, false); }

int plonk_hip_fr_batch_inverse(plonk_hip_ctx *ctx, uint64_t *data, size_t n) {
    PLONK_ENTER(ctx);
    if (!data && n) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) return PLONK_HIP_OK;
    Fr *d;
    int *d_flag, flag = 0;
    PLONK_TRY(fresh_flag(ctx, &d_flag));
    PLONK_TRY(upload(ctx, ctx->scalars_stage, data, n, 0, &d));
    PLONK_TRY(batch_inverse(ctx, d, n, d_flag));
    PLONK_TRY(read_flag(ctx, d_flag, &flag));
    if (flag) return PLONK_HIP_ERR_ZERO_DENOMINATOR;
    return download(ctx, data, d, n);
}

int plonk_hip_fr_linear_combination(plonk_hip_ctx *ctx, const uint64_t *x, size_t nx, const uint64_t *y, size_t ny,
                                    const uint64_t a[4], const uint64_t b[4], const uint64_t c[4], int broadcast,
                                    uint64_t *out) {
    PLONK_ENTER(ctx);
    const size_t n = nx > ny ? nx : ny;
    if (!a || !b || !c || (!x && nx) || (!y && ny) || (!out && n)) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) return PLONK_HIP_OK;
    Fr *d_x = nullptr, *d_y = nullptr;
    if (nx) PLONK_TRY(upload(ctx, ctx->scalars_stage, x, nx, n, &d_x));
    if (ny) PLONK_TRY(upload(ctx, ctx->scratch_c, y, ny, n, &d_y));
    Fr fa, fb, fc;
    memcpy(&fa, a, sizeof(Fr));
    memcpy(&fb, b, sizeof(Fr));
    memcpy(&fc, c, sizeof(Fr));
    Fr *d_out = nx >= ny ? d_x : d_y;  // the longer operand's buffer holds n elements
    PLONK_TRY(linear_combination(ctx, d_x, nx, d_y, ny, fa, fb, fc, broadcast != 0, d_out));
    return download(ctx, out, d_out, n);
}

int plonk_hip_fr_ratio_prefix_product(plonk_hip_ctx *ctx, const uint64_t *num, const uint64_t *den, size_t n,
                                      uint64_t *out) {
    PLONK_ENTER(ctx);
    if ((!num || !den || !out) && n) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) return PLONK_HIP_OK;
    Fr *d_num, *d_den;
    int *d_flag, flag = 0;
    PLONK_TRY(fresh_flag(ctx, &d_flag));
    PLONK_TRY(upload(ctx, ctx->scalars_stage, num, n, n + prefix_scan_scratch_elems(n), &d_num));
    PLONK_TRY(upload(ctx, ctx->scratch_c, den, n, 0, &d_den));
    PLONK_TRY(batch_inverse(ctx, d_den, n, d_flag));
    PLONK_TRY(pointwise_mul(ctx, d_num, d_den, d_num, n));
    PLONK_TRY(prefix_scan(ctx, d_num, n, true, d_num + n));
    PLONK_TRY(read_flag(ctx, d_flag, &flag));
    if (flag) return PLONK_HIP_ERR_ZERO_DENOMINATOR;
    return download(ctx, out, d_num, n);
}

int plonk_hip_fr_poly_mul(plonk_hip_ctx *ctx, const uint64_t *a, size_t na, const uint64_t *b, size_t nb,
                          uint64_t *out) {
    PLONK_ENTER(ctx);
    if ((!a && na) || (!b && nb)) return PLONK_HIP_ERR_INVALID_ARG;
    if (na == 0 || nb == 0) return PLONK_HIP_OK;
    if (!out) return PLONK_HIP_ERR_INVALID_ARG;
    Fr *d_a, *d_b;
    PLONK_TRY(upload(ctx, ctx->scalars_stage, a, na, 0, &d_a));
    PLONK_TRY(upload(ctx, ctx->scratch_c, b, nb, 0, &d_b));
    PLONK_TRY(buf_reserve(ctx, ctx->stage_c, (na + nb - 1) * sizeof(Fr)));
    Fr *d_o = (Fr *)ctx->stage_c.ptr;
    PLONK_TRY(poly_mul_dev(ctx, d_a, na, d_b, nb, d_o, na + nb - 1));
    return download(ctx, out, d_o, na + nb - 1);
}

int plonk_hip_fr_vanishing_from_roots(plonk_hip_ctx *ctx, const uint64_t *roots, size_t m, uint64_t *z) {
    PLONK_ENTER(ctx);
    if ((!roots && m) || !z) return PLONK_HIP_ERR_INVALID_ARG;
    Fr *d_r;
    PLONK_TRY(upload(ctx, ctx->scalars_stage, roots, m, 1, &d_r));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, (m + 1) * sizeof(Fr)));
    Fr *d_z = (Fr *)ctx->scratch_c.ptr;
    PLONK_TRY(vanishing_from_roots_dev(ctx, d_r, m, d_z));
    return download(ctx, z, d_z, m + 1);
}

int plonk_hip_fr_div_by_roots(plonk_hip_ctx *ctx, const uint64_t *f, size_t n, const uint64_t *roots, size_t m,
                              uint64_t *q) {
    PLONK_ENTER(ctx);
    if ((!f && n) || (!roots && m)) return PLONK_HIP_ERR_INVALID_ARG;
    if (n <= m) return PLONK_HIP_OK;
    if (!q) return PLONK_HIP_ERR_INVALID_ARG;
    Fr *d_f, *d_r;
    PLONK_TRY(upload(ctx, ctx->scalars_stage, f, n, 0, &d_f));
    PLONK_TRY(upload(ctx, ctx->scratch_c, roots, m, 1, &d_r));
    PLONK_TRY(buf_reserve(ctx, ctx->stage_c, (n - m) * sizeof(Fr)));
    Fr *d_q = (Fr *)ctx->stage_c.ptr;
    PLONK_TRY(div_by_roots_dev(ctx, d_f, n, d_r, (const Fr *)roots, m, d_q));
    return download(ctx, q, d_q, n - m);
}

int plonk_hip_fr_interpolate(plonk_hip_ctx *ctx, const uint64_t *xs, const uint64_t *ys, size_t n, uint64_t *coeffs) {
    PLONK_ENTER(ctx);
    if ((!xs || !ys || !coeffs) && n) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) return PLONK_HIP_OK;
    if (n > kMaxInterpolationPoints) return PLONK_HIP_ERR_INVALID_ARG;
    Fr *d_x, *d_y;
    int *d_flag, flag = 0;
    PLONK_TRY(fresh_flag(ctx, &d_flag));
    PLONK_TRY(upload(ctx, ctx->scalars_stage, xs, n, 0, &d_x));
    PLONK_TRY(upload(ctx, ctx->scratch_c, ys, n, 0, &d_y));
    PLONK_TRY(buf_reserve(ctx, ctx->stage_c, n * sizeof(Fr)));
    Fr *d_o = (Fr *)ctx->stage_c.ptr;
    PLONK_TRY(interpolate_dev(ctx, d_x, d_y, n, d_o, d_flag));
    PLONK_TRY(read_flag(ctx, d_flag, &flag));
    if (flag) return PLONK_HIP_ERR_ZERO_DENOMINATOR;
    return download(ctx, coeffs, d_o, n);
}

int plonk_hip_fr_interpolate_geometric(plonk_hip_ctx *ctx, const uint64_t q[4], const uint64_t *ys, size_t n,
                                       uint64_t *coeffs) {
    PLONK_ENTER(ctx);
    if (!q || ((!ys || !coeffs) && n)) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) return PLONK_HIP_OK;
    Fr *d_y;
    int *d_flag, flag = 0;
    PLONK_TRY(fresh_flag(ctx, &d_flag));
    PLONK_TRY(upload(ctx, ctx->scalars_stage, ys, n, 0, &d_y));
    PLONK_TRY(interpolate_geometric_dev(ctx, load_fr(q), d_y, n, d_y, d_flag));
    PLONK_TRY(read_flag(ctx, d_flag, &flag));
    if (flag) return PLONK_HIP_ERR_ZERO_DENOMINATOR;
    return download(ctx, coeffs, d_y, n);
}

int plonk_hip_fr_interpolate_geometric_dev(plonk_hip_ctx *ctx, const uint64_t q[4], void *d_values, size_t n) {
    PLONK_ENTER(ctx);
    if (!q || (!d_values && n)) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) return PLONK_HIP_OK;
    int *d_flag, flag = 0;
    PLONK_TRY(fresh_flag(ctx, &d_flag));
    PLONK_TRY(interpolate_geometric_dev(ctx, load_fr(q), (const Fr *)d_values, n, (Fr *)d_values, d_flag));
    PLONK_TRY(read_flag(ctx, d_flag, &flag));
    return flag ? PLONK_HIP_ERR_ZERO_DENOMINATOR : PLONK_HIP_OK;
}

// ---------------------------------------------------------------------------- device-resident variants
// The same operations on polynomials that already live in device memory (plonk_hip_malloc): nothing
// crosses PCIe but the scalars.  Stream-ordered like every _dev entry point; the ones that hand
// back host values wait for them.
int plonk_hip_memcpy_d2d(plonk_hip_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes) {
    PLONK_ENTER(ctx);
    if ((!dst_dev || !src_dev) && bytes) return PLONK_HIP_ERR_INVALID_ARG;
    if (bytes) PLONK_HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return PLONK_HIP_OK;
}

int plonk_hip_memset_zero(plonk_hip_ctx *ctx, void *dst_dev, size_t bytes) {
    PLONK_ENTER(ctx);
    if (!dst_dev && bytes) return PLONK_HIP_ERR_INVALID_ARG;
    if (bytes) PLONK_HIP_TRY(ctx, hipMemsetAsync(dst_dev, 0, bytes, ctx->stream));
    return PLONK_HIP_OK;
}

int plonk_hip_kzg_open_many_dev(plonk_hip_ctx *ctx, const void *d_coeffs, size_t n, const uint64_t *us, size_t m,
                                uint64_t *vs, uint64_t *proofs) {
    PLONK_ENTER(ctx);
    if (!d_coeffs || !us || !vs || !proofs) return PLONK_HIP_ERR_INVALID_ARG;
    if (ctx->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    if (n < 2) return PLONK_HIP_ERR_EMPTY_POLY;
    if (n - 1 > ctx->srs_n) return PLONK_HIP_ERR_SRS_TOO_SHORT;
    return open_many_core(ctx, (const Fr *)d_coeffs, n, us, m, vs, proofs);
}

int plonk_hip_kzg_commit_many_dev(plonk_hip_ctx *ctx, const void *const *d_coeffs, const size_t *ns, size_t batch,
                                  uint64_t *outs) {
    PLONK_ENTER(ctx);
    if (!d_coeffs || !ns || !outs || batch > 64) return PLONK_HIP_ERR_INVALID_ARG;
    if (ctx->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    for (size_t b = 0; b < batch; b++) {
        if (!d_coeffs[b]) return PLONK_HIP_ERR_INVALID_ARG;
        if (ns[b] == 0) return PLONK_HIP_ERR_EMPTY_POLY;
        if (ns[b] > ctx->srs_n) return PLONK_HIP_ERR_SRS_TOO_SHORT;
    }
    PLONK_TRY(buf_reserve(ctx, ctx->out_stage, 64 * sizeof(G1Jac) + 4096));
    G1Jac *d_out = (G1Jac *)ctx->out_stage.ptr;
    for (size_t b = 0; b < batch; b++)
        PLONK_TRY(msm_run_dev(ctx, (const Fr *)d_coeffs[b], ns[b], d_out + b, nullptr, batch > 1));
    PLONK_TRY(msm_join(ctx));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(outs, d_out, batch * sizeof(G1Jac), hipMemcpyDeviceToHost, ctx->stream));
    return stream_sync(ctx);
}

int plonk_hip_kzg_commit_open_many_dev(plonk_hip_ctx *ctx, const void *const *d_coeffs, const size_t *ns,
                                       const uint8_t *open, const uint64_t *us, size_t m, uint64_t *vs,
                                       uint64_t *outs) {
    PLONK_ENTER(ctx);
    if (!d_coeffs || !ns || !open || !us || !vs || !outs || m > 64) return PLONK_HIP_ERR_INVALID_ARG;
    if (ctx->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    size_t longest = 0;
    for (size_t k = 0; k < m; k++) {
        if (!d_coeffs[k]) return PLONK_HIP_ERR_INVALID_ARG;
        if (open[k]) {
            if (ns[k] < 2) return PLONK_HIP_ERR_EMPTY_POLY;
            if (ns[k] - 1 > ctx->srs_n) return PLONK_HIP_ERR_SRS_TOO_SHORT;
            if (ns[k] > longest) longest = ns[k];
        } else {
            if (ns[k] == 0) return PLONK_HIP_ERR_EMPTY_POLY;
            if (ns[k] > ctx->srs_n) return PLONK_HIP_ERR_SRS_TOO_SHORT;
        }
    }
    if (m == 0) return PLONK_HIP_OK;
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, (longest + suffix_scan_scratch_elems(longest) + 1) * sizeof(Fr)));
    PLONK_TRY(buf_reserve(ctx, ctx->out_stage, 64 * sizeof(G1Jac) + 4096));
    Fr *d_s = (Fr *)ctx->scratch_c.ptr;
    G1Jac *d_out = (G1Jac *)ctx->out_stage.ptr;
    memset(vs, 0, m * sizeof(Fr));
    for (size_t k = 0; k < m; k++) {
        const Fr *f = (const Fr *)d_coeffs[k];
        if (open[k]) {
            const size_t n = ns[k];
            PLONK_TRY(suffix_scan(ctx, f, n, load_fr(us + 4 * k), d_s, d_s + n));
            PLONK_TRY(msm_run_dev(ctx, d_s + 1, n - 1, d_out + k, nullptr, m > 1));
            PLONK_HIP_TRY(ctx, hipMemcpyAsync(vs + 4 * k, d_s, sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        } else {
            PLONK_TRY(msm_run_dev(ctx, f, ns[k], d_out + k, nullptr, m > 1));
        }
    }
    PLONK_TRY(msm_join(ctx));
    PLONK_HIP_TRY(ctx, hipMemcpyAsync(outs, d_out, m * sizeof(G1Jac), hipMemcpyDeviceToHost, ctx->stream));
    return stream_sync(ctx);
}

int plonk_hip_fr_scale_by_powers_dev(plonk_hip_ctx *ctx, void *d_coeffs, size_t n, const uint64_t w[4]) {
    PLONK_ENTER(ctx);
    if ((!d_coeffs && n) || !w) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) return PLONK_HIP_OK;
    return scale_by_powers(ctx, (Fr *)d_coeffs, n, 1, load_fr(w));
}

int plonk_hip_fr_poly_mul_dev(plonk_hip_ctx *ctx, const void *d_a, size_t na, const void *d_b, size_t nb, void *d_out) {
    PLONK_ENTER(ctx);
    if ((!d_a && na) || (!d_b && nb)) return PLONK_HIP_ERR_INVALID_ARG;
    if (na == 0 || nb == 0) return PLONK_HIP_OK;
    if (!d_out) return PLONK_HIP_ERR_INVALID_ARG;
    return poly_mul_dev(ctx, (const Fr *)d_a, na, (const Fr *)d_b, nb, (Fr *)d_out, na + nb - 1);
}

int plonk_hip_fr_linear_combination_dev(plonk_hip_ctx *ctx, const void *d_x, size_t nx, const void *d_y, size_t ny,
                                        const uint64_t a[4], const uint64_t b[4], const uint64_t c[4], int broadcast,
                                        void *d_out) {
    PLONK_ENTER(ctx);
    const size_t n = nx > ny ? nx : ny;
    if (!a || !b || !c || (!d_x && nx) || (!d_y && ny) || (!d_out && n)) return PLONK_HIP_ERR_INVALID_ARG;
    return linear_combination(ctx, (const Fr *)d_x, nx, (const Fr *)d_y, ny, load_fr(a), load_fr(b), load_fr(c),
                              broadcast != 0, (Fr *)d_out);
}

int plonk_hip_fr_div_by_roots_dev(plonk_hip_ctx *ctx, const void *d_f, size_t n, const void *d_roots, size_t m,
                                  void *d_q) {
    PLONK_ENTER(ctx);
    if ((!d_f && n) || (!d_roots && m)) return PLONK_HIP_ERR_INVALID_ARG;
    if (n <= m) return PLONK_HIP_OK;
    if (!d_q) return PLONK_HIP_ERR_INVALID_ARG;
    Fr few[24];
    if (m <= 24 && m) {  // the few-roots path divides by one linear factor after the other, roots on the host
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(few, d_roots, m * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return div_by_roots_dev(ctx, (const Fr *)d_f, n, (const Fr *)d_roots, few, m, (Fr *)d_q);
}

int plonk_hip_fr_div_vanishing_dev(plonk_hip_ctx *ctx, const void *d_f, size_t n, size_t k, void *d_q) {
    PLONK_ENTER(ctx);
    if (!d_f || k == 0) return PLONK_HIP_ERR_INVALID_ARG;
    if (n <= k) return PLONK_HIP_OK;
    if (!d_q) return PLONK_HIP_ERR_INVALID_ARG;
    return div_vanishing(ctx, (const Fr *)d_f, n, k, (Fr *)d_q);
}

int plonk_hip_fr_respects_permutation_dev(plonk_hip_ctx *ctx, const void *d_values, const uint32_t *perm, size_t n,
                                          int *holds) {
    PLONK_ENTER(ctx);
    if (((!d_values || !perm) && n) || !holds || n > 0xffffffffull) return PLONK_HIP_ERR_INVALID_ARG;
    int *d_flag, flag = 0;
    PLONK_TRY(fresh_flag(ctx, &d_flag));
    PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, n * sizeof(uint32_t)));
    uint32_t *d_perm = (uint32_t *)ctx->scratch_c.ptr;
    if (n) PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_perm, perm, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    PLONK_TRY(permutation_mismatch(ctx, (const Fr *)d_values, d_perm, n, d_flag));
    PLONK_TRY(read_flag(ctx, d_flag, &flag));
    *holds = flag ? 0 : 1;
    return PLONK_HIP_OK;
}

int plonk_hip_fr_is_zero_dev(plonk_hip_ctx *ctx, const void *d_x, size_t n, int *is_zero) {
    PLONK_ENTER(ctx);
    if ((!d_x && n) || !is_zero) return PLONK_HIP_ERR_INVALID_ARG;
    int *d_flag, flag = 0;
    PLONK_TRY(fresh_flag(ctx, &d_flag));
    PLONK_TRY(any_nonzero(ctx, (const Fr *)d_x, n, d_flag));
    PLONK_TRY(read_flag(ctx, d_flag, &flag));
    *is_zero = flag ? 0 : 1;
    return PLONK_HIP_OK;
}

int plonk_hip_fr_ratio_prefix_product_dev(plonk_hip_ctx *ctx, void *d_num, void *d_den, size_t n) {
    PLONK_ENTER(ctx);
    if ((!d_num || !d_den) && n) return PLONK_HIP_ERR_INVALID_ARG;
    if (n == 0) return PLONK_HIP_OK;
    int *d_flag, flag = 0;
    PLONK_TRY(fresh_flag(ctx, &d_flag));
    PLONK_TRY(buf_reserve(ctx, ctx->scalars_stage, prefix_scan_scratch_elems(n) * sizeof(Fr)));
    PLONK_TRY(batch_inverse(ctx, (Fr *)d_den, n, d_flag));
    PLONK_TRY(pointwise_mul(ctx, (const Fr *)d_num, (const Fr *)d_den, (Fr *)d_num, n));
    PLONK_TRY(prefix_scan(ctx, (Fr *)d_num, n, true, (Fr *)ctx->scalars_stage.ptr));
    PLONK_TRY(read_flag(ctx, d_flag, &flag));
    return flag ? PLONK_HIP_ERR_ZERO_DENOMINATOR : PLONK_HIP_OK;
}

// The same computation with no allocation anywhere: the running product's block totals go to this
// GPU's static scratch array, the zero-denominator report to the caller's device word, and nothing is
// read back, so the call only enqueues.
int plonk_hip_fr_ratio_prefix_product_dev_noalloc(plonk_hip_ctx *ctx, void *d_num, void *d_den, size_t n,
                                                  int *d_zero_flag) {
    PLONK_ENTER(ctx);
    if (((!d_num || !d_den) && n) || !d_zero_flag) return PLONK_HIP_ERR_INVALID_ARG;
    if (n > kStaticMaxElems || prefix_scan_scratch_elems(n) * sizeof(Fr) > kStaticScratchBytes) {
        ctx->last_error = "plonk_hip_fr_ratio_prefix_product_dev_noalloc: n exceeds 2^24 elements";
        return PLONK_HIP_ERR_INVALID_ARG;
    }
    if (!ctx->static_scratch) return PLONK_HIP_ERR_HIP;
    PLONK_HIP_TRY(ctx, hipMemsetAsync(d_zero_flag, 0, sizeof(int), ctx->stream));
    if (n == 0) return PLONK_HIP_OK;
    PLONK_TRY(batch_inverse(ctx, (Fr *)d_den, n, d_zero_flag));
    PLONK_TRY(pointwise_mul(ctx, (const Fr *)d_num, (const Fr *)d_den, (Fr *)d_num, n));
    {
        // the scan is the part that uses the GPU's shared array: behind the last use by any context, ahead of the next
        std::lock_guard<std::mutex> guard(g_static_mu);
        hipEvent_t done = g_static_done[ctx->device];
        PLONK_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, done, 0));
        int rc = prefix_scan(ctx, (Fr *)d_num, n, true, (Fr *)ctx->static_scratch);
        // recorded even after a failed launch: whatever part of the scan was queued still uses the array
        hipError_t e = hipEventRecord(done, ctx->stream);
        PLONK_TRY(rc);
        PLONK_HIP_TRY(ctx, e);
    }
    PLONK_HIP_TRY(ctx, hipGetLastError());
    return PLONK_HIP_OK;
}

}  // extern "C"
