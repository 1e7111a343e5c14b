// Several GPUs behind one handle, in one process: the SRS is split by point range over one
// sub-context per device, and kzg_commit / kzg_evaluate shard inside the library:
//     sum_i f_i [tau^i]G = sum over devices g of ( sum_{i in range_g} f_i [tau^i]G ).
// Every device gets its coefficient slice over its own PCIe link (one host thread per device, so the
// copies run side by side), runs the whole local MSM, and sends its 144-byte partial result to the
// first device by a peer copy (xGMI between the GPUs of a node); events order the streams and the
// first device adds the partials up.  Point addition is not a reduce-op, so this is a gather of opaque
// bytes, never an all-reduce.
//
// Opening at u needs s_j = f_j + u * s_(j+1) over the whole polynomial.  The recurrence is linear, so
// each device scans its own slice first with a zero carry-in, the host chains the slice totals
// (S_g = H_g + u^len_g * S_(g+1), a handful of field operations), and each device scans again with its
// true carry-in placed behind its slice as one more coefficient.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "context.hpp"
#include "poly.hpp"

using namespace plonk;

struct plonk_hip_multi {
    std::vector<plonk_hip_ctx *> sub;
    std::vector<hipEvent_t> done;  // per sub-context: its partial has been sent
    std::vector<size_t> first, count;  // SRS range of each sub-context
    size_t srs_n = 0;
    G1Jac *gather = nullptr;  // on the first device: one slot per sub-context, then the total
    // direct[g]: device g can write the first device's memory (peer access enabled, or the same device); otherwise
    // its partial is staged through slot g of `host_stage` (pinned host memory)
    std::vector<char> direct;
    G1Jac *host_stage = nullptr;
    std::string last_error;
    std::mutex mu;
};

namespace {

struct Bound {
    plonk_hip_ctx *ctx;
    std::lock_guard<std::recursive_mutex> guard;
    explicit Bound(plonk_hip_ctx *c) : ctx(c), guard(c->mu) {}
    int device() {
        if (hipSetDevice(ctx->device) != hipSuccess) {
            ctx->last_error = "hipSetDevice failed";
            return PLONK_HIP_ERR_HIP;
        }
        return PLONK_HIP_OK;
    }
};

// After a failure on one device the others may still have copies from the caller's buffers and kernels in
// flight: nothing returns to the caller before every stream has drained.
void drain_all(plonk_hip_multi *m) {
    for (plonk_hip_ctx *ctx : m->sub) {
        std::lock_guard<std::recursive_mutex> guard(ctx->mu);
        if (hipSetDevice(ctx->device) == hipSuccess) (void)hipStreamSynchronize(ctx->stream);
        ctx->npending = 0;
        (void)status_check(ctx);
    }
}

// fn(g) for every sub-context on a thread of its own; the first error wins (after all streams have drained)
template <class Fn>
int for_each_device(plonk_hip_multi *m, Fn fn) {
    const size_t p = m->sub.size();
    std::vector<int> rc(p, PLONK_HIP_OK);
    if (p == 1) {
        rc[0] = fn(0);
    } else {
        std::vector<std::thread> threads;
        threads.reserve(p);
        for (size_t g = 0; g < p; g++) threads.emplace_back([&, g] { rc[g] = fn(g); });
        for (auto &t : threads) t.join();
    }
    for (size_t g = 0; g < p; g++) {
        if (rc[g] != PLONK_HIP_OK) {
            m->last_error = "device " + std::to_string(m->sub[g]->device) + ": " + m->sub[g]->last_error;
            drain_all(m);
            return rc[g];
        }
    }
    return PLONK_HIP_OK;
}

void split_ranges(plonk_hip_multi *m, size_t n) {
    const size_t p = m->sub.size(), base = n / p, extra = n % p;
    m->first.assign(p, 0);
    m->count.assign(p, 0);
    for (size_t g = 0; g < p; g++) {
        m->count[g] = base + (g < extra ? 1 : 0);
        m->first[g] = g * base + (g < extra ? g : extra);
    }
    m->srs_n = n;
}

// Device g sends the result at d_partial to its slot on the first device and marks the event.
int send_partial(plonk_hip_multi *m, size_t g, const G1Jac *d_partial) {
    plonk_hip_ctx *ctx = m->sub[g];
    if (m->direct[g]) {
        PLONK_HIP_TRY(ctx, hipMemcpyPeerAsync(m->gather + g, m->sub[0]->device, d_partial, ctx->device, sizeof(G1Jac),
                                              ctx->stream));
    } else {
        // no peer access: as far as pinned host memory; the first device fetches it once the event has fired
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(m->host_stage + g, d_partial, sizeof(G1Jac), hipMemcpyDeviceToHost, ctx->stream));
    }
    PLONK_HIP_TRY(ctx, hipEventRecord(m->done[g], ctx->stream));
    return PLONK_HIP_OK;
}

// The first device waits for the partials of the devices in `active`, adds them up and hands the sum back.
int sum_partials(plonk_hip_multi *m, const std::vector<size_t> &active, uint64_t out[18]) {
    Bound b(m->sub[0]);
    PLONK_TRY(b.device());
    plonk_hip_ctx *c0 = m->sub[0];
    const size_t p = m->sub.size();
    // compact the active slots behind the per-device slots, in device order
    G1Jac *packed = m->gather + p, *total = m->gather + 2 * p;
    for (size_t k = 0; k < active.size(); k++) {
        const size_t g = active[k];
        PLONK_HIP_TRY(c0, hipStreamWaitEvent(c0->stream, m->done[g], 0));
        if (m->direct[g]) {
            PLONK_HIP_TRY(c0, hipMemcpyAsync(packed + k, m->gather + g, sizeof(G1Jac), hipMemcpyDeviceToDevice, c0->stream));
        } else {
            PLONK_HIP_TRY(c0, hipMemcpyAsync(packed + k, m->host_stage + g, sizeof(G1Jac), hipMemcpyHostToDevice, c0->stream));
        }
    }
    PLONK_TRY(g1_sum_dev(c0, packed, active.size(), total));
    PLONK_HIP_TRY(c0, hipMemcpyAsync(out, total, sizeof(G1Jac), hipMemcpyDeviceToHost, c0->stream));
    PLONK_HIP_TRY(c0, hipStreamSynchronize(c0->stream));
    // the first device's stream waited for every partial: all local sums are complete, look at what they reported
    int rc = PLONK_HIP_OK;
    for (size_t g : active) {
        std::lock_guard<std::recursive_mutex> guard(m->sub[g]->mu);
        const int r = status_check(m->sub[g]);
        if (r != PLONK_HIP_OK && rc == PLONK_HIP_OK) {
            rc = r;
            m->last_error = "device " + std::to_string(m->sub[g]->device) + ": " + m->sub[g]->last_error;
        }
    }
    return rc;
}

Fr load_fr(const uint64_t v[4]) {
    Fr r;
    memcpy(r.l, v, sizeof(r.l));
    return r;
}

Fr fr_pow(Fr base, size_t e) {
    Fr acc = fe_one<FrCfg>();
    while (e) {
        if (e & 1) acc = fe_mul(acc, base);
        base = fe_sqr(base);
        e >>= 1;
    }
    return acc;
}

}  // namespace

extern "C" {

int plonk_hip_multi_create(const int *devices, int n_devices, plonk_hip_multi **out) {
    if (!out) return PLONK_HIP_ERR_INVALID_ARG;
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return PLONK_HIP_ERR_INVALID_ARG;
    plonk_hip_multi *m = new (std::nothrow) plonk_hip_multi();
    if (!m) return PLONK_HIP_ERR_OUT_OF_MEMORY;
    int rc = PLONK_HIP_OK;
    for (int g = 0; g < n_devices && rc == PLONK_HIP_OK; g++) {
        plonk_hip_ctx *ctx = nullptr;
        rc = plonk_hip_ctx_create(devices[g], &ctx);
        if (rc != PLONK_HIP_OK) break;
        m->sub.push_back(ctx);
        hipEvent_t ev = nullptr;
        if (hipSetDevice(devices[g]) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
            rc = PLONK_HIP_ERR_HIP;
        m->done.push_back(ev);
    }
    if (rc == PLONK_HIP_OK) {
        if (hipSetDevice(devices[0]) != hipSuccess ||
            hipMalloc((void **)&m->gather, (3 * (size_t)n_devices + 1) * sizeof(G1Jac)) != hipSuccess ||
            hipHostMalloc((void **)&m->host_stage, (size_t)n_devices * sizeof(G1Jac), hipHostMallocPortable) != hipSuccess)
            rc = PLONK_HIP_ERR_HIP;
    }
    // Peer access towards the first device, so that the 144-byte partials travel GPU to GPU (xGMI inside a node)
    // instead of through a bounce buffer of the runtime; a device that cannot have it stages through pinned host
    // memory.  PLONK_HIP_MULTI_STAGED=1 forces the staged path (for tests on fewer GPUs than sub-contexts).
    if (rc == PLONK_HIP_OK) {
        const char *force = std::getenv("PLONK_HIP_MULTI_STAGED");
        const bool staged = force && force[0] == '1';
        m->direct.assign((size_t)n_devices, 0);
        for (int g = 0; g < n_devices; g++) {
            if (staged) continue;
            if (devices[g] == devices[0]) {
                m->direct[(size_t)g] = 1;
                continue;
            }
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[g], devices[0]) != hipSuccess || !can) continue;
            if (hipSetDevice(devices[g]) != hipSuccess) continue;
            const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) m->direct[(size_t)g] = 1;
            (void)hipGetLastError();  // "already enabled" is not an error to carry forward
        }
    }
    if (rc != PLONK_HIP_OK) {
        plonk_hip_multi_destroy(m);
        return rc;
    }
    *out = m;
    return PLONK_HIP_OK;
}

void plonk_hip_multi_destroy(plonk_hip_multi *m) {
    if (!m) return;
    for (size_t g = 0; g < m->sub.size(); g++) {
        (void)hipSetDevice(m->sub[g]->device);
        (void)hipStreamSynchronize(m->sub[g]->stream);
        if (g < m->done.size() && m->done[g]) (void)hipEventDestroy(m->done[g]);
    }
    if (m->gather && !m->sub.empty()) {
        (void)hipSetDevice(m->sub[0]->device);
        (void)hipFree(m->gather);
    }
    if (m->host_stage) (void)hipHostFree(m->host_stage);
    for (plonk_hip_ctx *ctx : m->sub) plonk_hip_ctx_destroy(ctx);
    delete m;
}

const char *plonk_hip_multi_last_error(plonk_hip_multi *m) { return m ? m->last_error.c_str() : ""; }

int plonk_hip_multi_device_count(plonk_hip_multi *m) { return m ? (int)m->sub.size() : 0; }

int plonk_hip_multi_set_window_bits(plonk_hip_multi *m, int c) {
    if (!m) return PLONK_HIP_ERR_INVALID_ARG;
    for (plonk_hip_ctx *ctx : m->sub) PLONK_TRY(plonk_hip_ctx_set_window_bits(ctx, c));
    return PLONK_HIP_OK;
}

int plonk_hip_multi_srs_size(plonk_hip_multi *m, size_t *n) {
    if (!m || !n) return PLONK_HIP_ERR_INVALID_ARG;
    *n = m->srs_n;
    return PLONK_HIP_OK;
}

int plonk_hip_multi_srs_range(plonk_hip_multi *m, int index, size_t *first, size_t *count) {
    if (!m || index < 0 || (size_t)index >= m->sub.size() || !first || !count) return PLONK_HIP_ERR_INVALID_ARG;
    if (m->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    *first = m->first[index];
    *count = m->count[index];
    return PLONK_HIP_OK;
}

int plonk_hip_multi_srs_load_projective(plonk_hip_multi *m, const uint64_t *pts, size_t n) {
    if (!m || !pts) return PLONK_HIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> guard(m->mu);
    if (n == 0) return PLONK_HIP_ERR_INVALID_ARG;
    m->srs_n = 0;
    split_ranges(m, n);
    int rc = for_each_device(m, [&](size_t g) -> int {
        if (m->count[g] == 0) return PLONK_HIP_OK;  // fewer points than devices: the last ones hold none
        Bound b(m->sub[g]);
        PLONK_TRY(b.device());
        PLONK_TRY(srs_load_projective(m->sub[g], pts + 18 * m->first[g], m->count[g]));
        PLONK_HIP_TRY(m->sub[g], hipStreamSynchronize(m->sub[g]->stream));
        return PLONK_HIP_OK;
    });
    if (rc != PLONK_HIP_OK) m->srs_n = 0;
    return rc;
}

int plonk_hip_multi_kzg_setup(plonk_hip_multi *m, const uint64_t tau[4], size_t degree, uint64_t *out_pts) {
    if (!m || !tau) return PLONK_HIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> guard(m->mu);
    const size_t n = degree + 1;
    m->srs_n = 0;
    split_ranges(m, n);
    int rc = for_each_device(m, [&](size_t g) -> int {
        if (m->count[g] == 0) return PLONK_HIP_OK;  // fewer points than devices: the last ones hold none
        Bound b(m->sub[g]);
        PLONK_TRY(b.device());
        PLONK_TRY(srs_setup(m->sub[g], tau, m->first[g], m->count[g], out_pts ? out_pts + 18 * m->first[g] : nullptr));
        PLONK_HIP_TRY(m->sub[g], hipStreamSynchronize(m->sub[g]->stream));
        return PLONK_HIP_OK;
    });
    if (rc != PLONK_HIP_OK) m->srs_n = 0;
    return rc;
}

int plonk_hip_multi_kzg_commit(plonk_hip_multi *m, const uint64_t *coeffs, size_t n, uint64_t out[18]) {
    if (!m || !out || (!coeffs && n)) return PLONK_HIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> guard(m->mu);
    if (m->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    if (n == 0) return PLONK_HIP_ERR_EMPTY_POLY;
    if (n > m->srs_n) return PLONK_HIP_ERR_SRS_TOO_SHORT;
    std::vector<size_t> active;
    for (size_t g = 0; g < m->sub.size(); g++)
        if (m->first[g] < n) active.push_back(g);
    PLONK_TRY(for_each_device(m, [&](size_t g) -> int {
        if (m->first[g] >= n) return PLONK_HIP_OK;  // the polynomial ends before this device's range
        const size_t len = std::min(m->count[g], n - m->first[g]);
        plonk_hip_ctx *ctx = m->sub[g];
        Bound b(ctx);
        PLONK_TRY(b.device());
        PLONK_TRY(buf_reserve(ctx, ctx->scalars_stage, len * sizeof(Fr)));
        PLONK_TRY(buf_reserve(ctx, ctx->out_stage, 64 * sizeof(G1Jac) + 4096));
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(ctx->scalars_stage.ptr, coeffs + 4 * m->first[g], len * sizeof(Fr),
                                          hipMemcpyHostToDevice, ctx->stream));
        PLONK_TRY(msm_run_dev(ctx, (const Fr *)ctx->scalars_stage.ptr, len, (G1Jac *)ctx->out_stage.ptr));
        return send_partial(m, g, (const G1Jac *)ctx->out_stage.ptr);
    }));
    return sum_partials(m, active, out);
}

int plonk_hip_multi_kzg_commit_batch(plonk_hip_multi *m, const uint64_t *const *coeffs, const size_t *ns, size_t batch,
                                     uint64_t *outs) {
    if (!m || !coeffs || !ns || !outs) return PLONK_HIP_ERR_INVALID_ARG;
    for (size_t k = 0; k < batch; k++) PLONK_TRY(plonk_hip_multi_kzg_commit(m, coeffs[k], ns[k], outs + 18 * k));
    return PLONK_HIP_OK;
}

int plonk_hip_multi_kzg_open(plonk_hip_multi *m, const uint64_t *coeffs, size_t n, const uint64_t u[4], uint64_t v[4],
                             uint64_t proof[18]) {
    if (!m || !coeffs || !u || !v || !proof) return PLONK_HIP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> guard(m->mu);
    if (m->srs_n == 0) return PLONK_HIP_ERR_NO_SRS;
    if (n < 2) return PLONK_HIP_ERR_EMPTY_POLY;
    if (n - 1 > m->srs_n) return PLONK_HIP_ERR_SRS_TOO_SHORT;
    const size_t p = m->sub.size(), nq = n - 1;  // quotient coefficient i = s_(i+1) meets SRS point i
    const Fr point = load_fr(u);
    std::vector<size_t> len(p, 0), active;
    for (size_t g = 0; g < p; g++) {
        if (m->first[g] < nq) {
            len[g] = std::min(m->count[g], nq - m->first[g]);
            active.push_back(g);
        }
    }
    // 1. every device: its slice f[first+1 .. first+1+len) plus one slot for the carry-in, scanned with
    //    a zero carry-in; the slice total H_g = s_local[0] comes back to the host
    std::vector<Fr> totals(p);
    PLONK_TRY(for_each_device(m, [&](size_t g) -> int {
        if (!len[g]) return PLONK_HIP_OK;
        plonk_hip_ctx *ctx = m->sub[g];
        Bound b(ctx);
        PLONK_TRY(b.device());
        const size_t scr = suffix_scan_scratch_elems(len[g] + 1);
        PLONK_TRY(buf_reserve(ctx, ctx->scalars_stage, (len[g] + 1) * sizeof(Fr)));
        PLONK_TRY(buf_reserve(ctx, ctx->scratch_c, (len[g] + 1 + scr) * sizeof(Fr)));
        PLONK_TRY(buf_reserve(ctx, ctx->out_stage, 64 * sizeof(G1Jac) + 4096));
        Fr *d_f = (Fr *)ctx->scalars_stage.ptr, *d_s = (Fr *)ctx->scratch_c.ptr;
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_f, coeffs + 4 * (m->first[g] + 1), len[g] * sizeof(Fr), hipMemcpyHostToDevice,
                                          ctx->stream));
        PLONK_TRY(suffix_scan(ctx, d_f, len[g], point, d_s, d_s + len[g] + 1));
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(&totals[g], d_s, sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return PLONK_HIP_OK;
    }));
    // 2. carries, from the top slice down: S_g = H_g + u^len_g * S_(g+1)
    std::vector<Fr> carry_in(p, fe_zero<FrCfg>());
    Fr above = fe_zero<FrCfg>();
    for (size_t k = active.size(); k-- > 0;) {
        const size_t g = active[k];
        carry_in[g] = above;
        above = fe_add(totals[g], fe_mul(fr_pow(point, len[g]), above));
    }
    // f(u) = f_0 + u * s_1, and s_1 is the true value at the start of the first slice
    const Fr value = fe_add(load_fr(coeffs), fe_mul(point, above));
    memcpy(v, value.l, sizeof(value.l));
    // 3. every device again, now with its carry-in as one more coefficient behind the slice; the MSM of
    //    the true quotient slice against the local SRS range; partial to the first device
    PLONK_TRY(for_each_device(m, [&](size_t g) -> int {
        if (!len[g]) return PLONK_HIP_OK;
        plonk_hip_ctx *ctx = m->sub[g];
        Bound b(ctx);
        PLONK_TRY(b.device());
        Fr *d_f = (Fr *)ctx->scalars_stage.ptr, *d_s = (Fr *)ctx->scratch_c.ptr;
        PLONK_HIP_TRY(ctx, hipMemcpyAsync(d_f + len[g], &carry_in[g], sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        PLONK_TRY(suffix_scan(ctx, d_f, len[g] + 1, point, d_s, d_s + len[g] + 1));
        PLONK_TRY(msm_run_dev(ctx, d_s, len[g], (G1Jac *)ctx->out_stage.ptr));
        PLONK_TRY(send_partial(m, g, (const G1Jac *)ctx->out_stage.ptr));
        // carry_in[g] must outlive the copy that reads it
        PLONK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return PLONK_HIP_OK;
    }));
    return sum_partials(m, active, proof);
}

}  // extern "C"
