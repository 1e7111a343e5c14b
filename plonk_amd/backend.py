"""Python view of one library context.

Mirrors the reference's `common::kzg` / `common::polynomials` surface on limb arrays:
numpy uint64, little-endian limbs, Montgomery form (Fr = 4 limbs, G1 projective = 18)."""
import ctypes

import numpy as np

from . import lib as _l

ERR_SRS_TOO_SHORT = -1
ERR_EMPTY_POLY = -2
ERR_ZERO_DENOMINATOR = -3
ERR_INVALID_POINT = -9


class PlonkHipError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("plonk_hip error %d: %s" % (code, text))
        self.code = code


class CommitError(PlonkHipError):
    """`CommitError::CommitFailed` of the reference (plonk/src/common/kzg.rs:11-15)."""


class ReferencePanic(PlonkHipError):
    """Inputs on which the reference panics (empty reduce, out-of-bounds index, inverse of 0)."""


def _u64(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_l.c_u64p)


def _fr(a, cols=4):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, cols)


def _host_check(rc):
    if rc != 0:
        text = _l.load_library().plonk_hip_strerror(rc).decode()
        raise (ReferencePanic if rc == ERR_INVALID_POINT else PlonkHipError)(rc, text)


def kzg_setup_tau_g2(tau):
    """gp.tau_g2 of `kzg_setup` (plonk/src/common/kzg.rs:38-39) for the trapdoor tau (4 Montgomery limbs): 96
    bytes, compressed.  Host code."""
    out = ctypes.create_string_buffer(96)
    _host_check(_l.load_library().plonk_hip_kzg_setup_tau_g2(_u64(_fr(tau)), out))
    return out.raw


def kzg_verify(tau_g2, com, u, v, proof):
    """`kzg_verify(gp, com_f, u, v, proof)` (plonk/src/common/kzg.rs:91-98).  tau_g2: the 96-byte compressed
    gp.tau_g2; com, proof: 18 limbs; u, v: 4 limbs (Montgomery).  Host code: needs no context and no GPU."""
    ok = ctypes.c_int(0)
    _host_check(_l.load_library().plonk_hip_kzg_verify(bytes(tau_g2), _u64(_fr(com, 18)), _u64(_fr(u)), _u64(_fr(v)),
                                                       _u64(_fr(proof, 18)), ctypes.byref(ok)))
    return bool(ok.value)


def kzg_verify_batch(tau_g2, coms, us, vs, proofs, rho=None):
    """All of a proof's kzg_verify calls as one pairing-product check (two Miller loops, one final
    exponentiation).  coms, proofs: (n, 18); us, vs: (n, 4); rho: (n, 2) 128-bit weights, drawn from the
    operating system's generator when not given (the first one is 1)."""
    coms, proofs, us, vs = _fr(coms, 18), _fr(proofs, 18), _fr(us), _fr(vs)
    n = coms.shape[0]
    assert proofs.shape[0] == n and us.shape[0] == n and vs.shape[0] == n
    if rho is None:
        import os
        rho = np.frombuffer(os.urandom(16 * max(n, 1)), dtype=np.uint64).reshape(-1, 2).copy()
        rho[0] = (1, 0)
    rho = _fr(rho, 2)
    assert rho.shape[0] >= n
    ok = ctypes.c_int(0)
    _host_check(_l.load_library().plonk_hip_kzg_verify_batch(bytes(tau_g2), n, _u64(coms), _u64(us), _u64(vs),
                                                             _u64(proofs), _u64(rho), ctypes.byref(ok)))
    return bool(ok.value)


class Backend:
    def __init__(self, device=0, window_bits=0):
        self._lib = _l.load_library()
        self._ctx = _l.c_ctx()
        rc = self._lib.plonk_hip_ctx_create(device, ctypes.byref(self._ctx))
        if rc != 0:
            raise PlonkHipError(rc, self._lib.plonk_hip_strerror(rc).decode())
        self.device = device
        if window_bits:
            self._check(self._lib.plonk_hip_ctx_set_window_bits(self._ctx, window_bits))

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if self._ctx:
            self._lib.plonk_hip_ctx_destroy(self._ctx)
            self._ctx = _l.c_ctx()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == 0:
            return
        text = self._lib.plonk_hip_strerror(rc).decode()
        detail = self._lib.plonk_hip_last_error(self._ctx).decode()
        if detail and rc in (-6, -7):
            text += " (" + detail + ")"
        if rc == ERR_SRS_TOO_SHORT:
            raise CommitError(rc, text)
        if rc in (ERR_EMPTY_POLY, ERR_ZERO_DENOMINATOR, ERR_INVALID_POINT):
            raise ReferencePanic(rc, text)
        raise PlonkHipError(rc, text)

    @property
    def handle(self):
        return self._ctx

    @property
    def stream(self):
        return self._lib.plonk_hip_ctx_stream(self._ctx)

    def sync(self):
        self._check(self._lib.plonk_hip_sync(self._ctx))

    def malloc(self, nbytes):
        p = ctypes.c_void_p()
        self._check(self._lib.plonk_hip_malloc(self._ctx, nbytes, ctypes.byref(p)))
        return p.value

    def free(self, dptr):
        self._check(self._lib.plonk_hip_free(self._ctx, dptr))

    def to_device(self, arr, dptr=None):
        arr = np.ascontiguousarray(arr)
        if dptr is None:
            dptr = self.malloc(arr.nbytes)
        self._check(self._lib.plonk_hip_memcpy_h2d(self._ctx, dptr, arr.ctypes.data, arr.nbytes))
        return dptr

    def from_device(self, dptr, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        self._check(self._lib.plonk_hip_memcpy_d2h(self._ctx, out.ctypes.data, dptr, out.nbytes))
        return out

    def set_window_bits(self, c):
        self._check(self._lib.plonk_hip_ctx_set_window_bits(self._ctx, c))

    def profile(self, enable=True):
        self._check(self._lib.plonk_hip_profile(self._ctx, 1 if enable else 0))

    def timer_start(self):
        self._check(self._lib.plonk_hip_timer_start(self._ctx))

    def timer_stop(self):
        """Milliseconds of GPU time since timer_start (HIP events on the library's stream); waits."""
        ms = ctypes.c_float()
        self._check(self._lib.plonk_hip_timer_stop(self._ctx, ctypes.byref(ms)))
        return float(ms.value)

    def msm_phase_ms(self):
        ms = (ctypes.c_float * 6)()
        self._check(self._lib.plonk_hip_msm_phase_ms(self._ctx, ms))
        return {self._lib.plonk_hip_msm_phase_name(i).decode(): float(ms[i]) for i in range(6)}

    def bench_op_rate(self, op):
        v = ctypes.c_double()
        self._check(self._lib.plonk_hip_bench_op_rate(self._ctx, op, ctypes.byref(v)))
        return v.value

    def sort_rank_mode(self):
        """"add" when the bucket sort ranks keys with one LDS add-with-return each (the library checked at
        context creation that this device serves the lanes of such an instruction in ascending order),
        "ballot" for the form that relies on nothing."""
        return "add" if self._lib.plonk_hip_bench_sort_rank_mode(self._ctx) else "ballot"

    def bench_batch_affine(self, accumulators_per_lane, steps, invert=True, want_points=False):
        """Additions per second of the batch-affine probe (plonk_hip_bench_batch_affine); with want_points also the
        first block's accumulators, (256 * accumulators_per_lane, 18) limbs."""
        v = ctypes.c_double()
        pts = np.empty((256 * accumulators_per_lane, 18), dtype=np.uint64) if want_points else None
        self._check(self._lib.plonk_hip_bench_batch_affine(self._ctx, accumulators_per_lane, steps, 1 if invert else 0,
                                                           ctypes.byref(v), _u64(pts) if want_points else None))
        return (v.value, pts) if want_points else v.value

    def debug_skip_sort(self, on):
        """Fault injection for the tests: commitments skip their bucket sort (the accumulation must then report
        keys out of order instead of returning a wrong point)."""
        self._check(self._lib.plonk_hip_bench_skip_sort(self._ctx, 1 if on else 0))

    def bench_msm_tuning(self, seg_len=0, pair_loads=0):
        """Measurement overrides of the accumulation (plonk_hip_bench_msm_tuning): segment length, and how the
        lanes fetch their pairs (1: one position at a time, 2: a group at a time through LDS); 0 restores the
        library's own choice of either."""
        self._check(self._lib.plonk_hip_bench_msm_tuning(self._ctx, seg_len, pair_loads))

    def bench_gather_rate(self, table_bytes):
        v = ctypes.c_double()
        self._check(self._lib.plonk_hip_bench_gather_rate(self._ctx, table_bytes, ctypes.byref(v)))
        return v.value

    def bench_mad_rate(self):
        """Measured peak of the integer multiply-add (mads per second, whole GPU)."""
        v = ctypes.c_double()
        self._check(self._lib.plonk_hip_bench_mad_rate(self._ctx, ctypes.byref(v)))
        return v.value

    def bench_mad_mix(self, extra=0, blocks_per_cu=8):
        v, clk = ctypes.c_double(), ctypes.c_double()
        self._check(self._lib.plonk_hip_bench_mad_mix(self._ctx, extra, blocks_per_cu, ctypes.byref(v),
                                                      ctypes.byref(clk)))
        return v.value, clk.value

    def bench_fq_mul_rate(self, variant=0, blocks_per_cu=8):
        v = ctypes.c_double()
        self._check(self._lib.plonk_hip_bench_fq_mul_rate(self._ctx, variant, blocks_per_cu, ctypes.byref(v)))
        return v.value

    def msm_plan(self):
        c, w, b = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        self._check(self._lib.plonk_hip_msm_plan(self._ctx, ctypes.byref(c), ctypes.byref(w), ctypes.byref(b)))
        return {"window_bits": c.value, "windows": w.value, "buckets": b.value}

    # ------------------------------------------------------------------ SRS
    def srs_load_projective(self, pts):
        """`GlobalParameters.tau_powers_g1` (kzg.rs:17-21): (n, 18) Jacobian limbs."""
        pts = _fr(pts, 18)
        self._check(self._lib.plonk_hip_srs_load_projective(self._ctx, _u64(pts), pts.shape[0]))

    def srs_load_compressed(self, encodings, check_subgroup=True):
        """The SRS as setup_global_params/json.rs:22-48 reads it: an iterable of 48-byte compressed
        G1 encodings (or one bytes object of n x 48).  Raises ReferencePanic on an encoding that
        `G1::deserialize_compressed` rejects."""
        raw = encodings if isinstance(encodings, (bytes, bytearray)) else b"".join(encodings)
        assert len(raw) % 48 == 0
        bad = ctypes.c_size_t()
        self._check(self._lib.plonk_hip_srs_load_compressed(self._ctx, bytes(raw), len(raw) // 48,
                                                            1 if check_subgroup else 0, ctypes.byref(bad)))

    def g1_decompress(self, encodings, check_subgroup=True):
        """Batch `G1::deserialize_compressed`: -> (n, 18) Jacobian limbs (Z = 1; identity (1, 1, 0))."""
        raw = encodings if isinstance(encodings, (bytes, bytearray)) else b"".join(encodings)
        assert len(raw) % 48 == 0
        n = len(raw) // 48
        out = np.empty((n, 18), dtype=np.uint64)
        bad = ctypes.c_size_t()
        rc = self._lib.plonk_hip_g1_decompress(self._ctx, bytes(raw), n, 1 if check_subgroup else 0, _u64(out),
                                               ctypes.byref(bad))
        self.last_bad_index = bad.value
        self._check(rc)
        return out

    def g1_compress(self, pts):
        """Batch `serialize_compressed` (utils.rs:70-72): (n, 18) limbs -> list of 48-byte strings."""
        pts = _fr(pts, 18)
        n = pts.shape[0]
        buf = ctypes.create_string_buffer(48 * n)
        self._check(self._lib.plonk_hip_g1_compress(self._ctx, _u64(pts), n, buf))
        raw = buf.raw  # one copy
        return [raw[48 * i:48 * i + 48] for i in range(n)]

    def srs_size(self):
        n = ctypes.c_size_t()
        self._check(self._lib.plonk_hip_srs_size(self._ctx, ctypes.byref(n)))
        return n.value

    def srs_read(self, first, count):
        out = np.empty((count, 18), dtype=np.uint64)
        self._check(self._lib.plonk_hip_srs_read(self._ctx, first, count, _u64(out)))
        return out

    def kzg_setup(self, tau, degree, want_points=True):
        """`kzg_setup(degree)` (kzg.rs:24-45) for a caller-supplied tau."""
        tau = _fr(tau).reshape(4)
        out = np.empty((degree + 1, 18), dtype=np.uint64) if want_points else None
        self._check(self._lib.plonk_hip_kzg_setup(self._ctx, _u64(tau), degree, _u64(out) if want_points else None))
        return out

    def kzg_setup_own_tau(self, degree, want_points=True):
        """`kzg_setup(degree)` with the reference's own trapdoor, tau = Fr::rand(&mut ark_std::test_rng())
        (kzg.rs:25-28): returns (tau_powers_g1 or None, tau_g2 as its 96-byte compressed encoding).  With
        degree = 21 this is plonk/data/srs.json."""
        from .ark_rng import TestRng
        tau = TestRng().fr()
        return self.kzg_setup(tau, degree, want_points), kzg_setup_tau_g2(tau)

    def kzg_setup_range(self, tau, first, count, want_points=False):
        """SRS slice [tau^first .. tau^(first+count-1)]G for point-range sharding."""
        tau = _fr(tau).reshape(4)
        out = np.empty((count, 18), dtype=np.uint64) if want_points else None
        self._check(self._lib.plonk_hip_kzg_setup_range(self._ctx, _u64(tau), first, count,
                                                        _u64(out) if want_points else None))
        return out

    def g1_sum_dev(self, d_pts, n, d_out):
        self._check(self._lib.plonk_hip_g1_sum_dev(self._ctx, d_pts, n, d_out))

    def msm_bucket_layout(self):
        """(number of buckets, bytes per bucket) of the installed SRS's window plan."""
        nb, each = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._check(self._lib.plonk_hip_msm_bucket_layout(self._ctx, ctypes.byref(nb), ctypes.byref(each)))
        return nb.value, each.value

    def msm_buckets_dev(self, d_coeffs, n, d_buckets):
        """Commitment of d_coeffs up to the bucket sums (opaque bytes at d_buckets)."""
        self._check(self._lib.plonk_hip_msm_buckets_dev(self._ctx, d_coeffs, n, d_buckets))

    def msm_reduce_bucket_slice_dev(self, d_parts, nparts, first, count, d_out):
        """Share of the commitment carried by buckets [first, first + count), summed over nparts
        bucket partials."""
        self._check(self._lib.plonk_hip_msm_reduce_bucket_slice_dev(self._ctx, d_parts, nparts, first, count, d_out))

    # ------------------------------------------------------------------ commit / open
    def kzg_commit(self, coeffs):
        """`kzg_commit(gp, f)` (kzg.rs:48-60) -> (18,) Jacobian limbs."""
        coeffs = _fr(coeffs)
        out = np.empty(18, dtype=np.uint64)
        self._check(self._lib.plonk_hip_kzg_commit(self._ctx, _u64(coeffs), coeffs.shape[0], _u64(out)))
        return out

    def kzg_commit_dev(self, d_coeffs, n, d_out):
        self._check(self._lib.plonk_hip_kzg_commit_dev(self._ctx, d_coeffs, n, d_out))

    def kzg_commit_batch(self, polys):
        polys = [_fr(p) for p in polys]
        ptrs = (_l.c_u64p * len(polys))(*[_u64(p) for p in polys])
        ns = (ctypes.c_size_t * len(polys))(*[p.shape[0] for p in polys])
        outs = np.empty((len(polys), 18), dtype=np.uint64)
        self._check(self._lib.plonk_hip_kzg_commit_batch(self._ctx, ptrs, ns, len(polys), _u64(outs)))
        return outs

    def kzg_evaluate(self, coeffs, u):
        """`kzg_evaluate(gp, f, u)` (kzg.rs:63-88) -> (v (4,), proof (18,))."""
        coeffs = _fr(coeffs)
        u = _fr(u).reshape(4)
        v = np.empty(4, dtype=np.uint64)
        proof = np.empty(18, dtype=np.uint64)
        self._check(self._lib.plonk_hip_kzg_open(self._ctx, _u64(coeffs), coeffs.shape[0], _u64(u), _u64(v),
                                                 _u64(proof)))
        return v, proof

    def kzg_evaluate_many(self, coeffs, us):
        coeffs = _fr(coeffs)
        us = _fr(us)
        vs = np.empty((us.shape[0], 4), dtype=np.uint64)
        proofs = np.empty((us.shape[0], 18), dtype=np.uint64)
        self._check(self._lib.plonk_hip_kzg_open_many(self._ctx, _u64(coeffs), coeffs.shape[0], _u64(us),
                                                      us.shape[0], _u64(vs), _u64(proofs)))
        return vs, proofs

    # ------------------------------------------------------------------ polynomials
    def fr_ntt(self, data, inverse=False, shift=None):
        """Evaluations over Omega (forward) / interpolation over Omega (inverse); natural order.
        data: (batch, 2^k, 4) or (2^k, 4).  Returns a new array."""
        d = np.array(data, dtype=np.uint64)
        shape = d.shape
        d = d.reshape(-1, shape[-2], 4) if d.ndim == 3 else d.reshape(1, -1, 4)
        n = d.shape[1]
        log_n = n.bit_length() - 1
        assert 1 << log_n == n
        if shift is None:
            rc = self._lib.plonk_hip_fr_ntt(self._ctx, _u64(d), log_n, d.shape[0], 1 if inverse else 0)
        else:
            s = _fr(shift).reshape(4)
            rc = self._lib.plonk_hip_fr_coset_ntt(self._ctx, _u64(d), log_n, d.shape[0], _u64(s), 1 if inverse else 0)
        self._check(rc)
        return d.reshape(shape)

    def fr_ntt_mixed(self, data, inverse=False):
        """Evaluations over / interpolation from construct_Omega(n), n = 2^k or 3 * 2^k; natural
        order.  data: (batch, n, 4) or (n, 4).  Returns a new array."""
        d = np.array(data, dtype=np.uint64)
        shape = d.shape
        d = d.reshape(-1, shape[-2], 4) if d.ndim == 3 else d.reshape(1, -1, 4)
        self._check(self._lib.plonk_hip_fr_ntt_mixed(self._ctx, _u64(d), d.shape[1], d.shape[0], 1 if inverse else 0))
        return d.reshape(shape)

    def fr_ntt_dev(self, dptr, log_n, batch, inverse=False):
        self._check(self._lib.plonk_hip_fr_ntt_dev(self._ctx, dptr, log_n, batch, 1 if inverse else 0))

    def interpolate_over_omega(self, ys):
        """`interpolate_polynomial(construct_Omega(n), ys)` (polynomials.rs:50-58)."""
        return self.fr_ntt(ys, inverse=True)

    def evaluate_over_omega(self, coeffs):
        return self.fr_ntt(coeffs, inverse=False)

    def compose_with_scaling(self, coeffs, w):
        """`compose_polynomials(f, [0, w])` (polynomials.rs:80-113): f(wX)."""
        c = np.array(_fr(coeffs))
        w = _fr(w).reshape(4)
        self._check(self._lib.plonk_hip_fr_scale_by_powers(self._ctx, _u64(c), c.shape[0], _u64(w)))
        return c

    def fr_eval(self, coeffs, u):
        coeffs = _fr(coeffs)
        u = _fr(u).reshape(4)
        out = np.empty(4, dtype=np.uint64)
        self._check(self._lib.plonk_hip_fr_eval(self._ctx, _u64(coeffs), coeffs.shape[0], _u64(u), _u64(out)))
        return out

    def compute_q_zero_test(self, k, f):
        """`compute_q_zero_test(k, f)` (protocols.rs:157-166): long-division quotient by X^k - 1."""
        f = _fr(f)
        n = f.shape[0]
        q = np.zeros((max(n - k, 0), 4), dtype=np.uint64)
        if n > k:
            self._check(self._lib.plonk_hip_fr_div_vanishing(self._ctx, _u64(f), n, k, _u64(q)))
        return q

    def fr_eval_many(self, coeffs, points):
        """f(points[i]) for every i: the `f.evaluate(&Omega[j])` calls of protocols.rs:246, 348, 464, 577."""
        coeffs, points = _fr(coeffs), _fr(points)
        out = np.empty((points.shape[0], 4), dtype=np.uint64)
        self._check(self._lib.plonk_hip_fr_eval_many(self._ctx, _u64(coeffs), coeffs.shape[0], _u64(points),
                                                     points.shape[0], _u64(out)))
        return out

    def fr_evaluate_over_omega(self, coeffs, log_k):
        """f(omega^j), j < 2^log_k, for a polynomial of any length."""
        coeffs = _fr(coeffs)
        out = np.empty((1 << log_k, 4), dtype=np.uint64)
        self._check(self._lib.plonk_hip_fr_evaluate_over_omega(self._ctx, _u64(coeffs), coeffs.shape[0], log_k,
                                                               _u64(out)))
        return out

    def fr_prefix_product(self, data):
        d = np.array(_fr(data), dtype=np.uint64)
        self._check(self._lib.plonk_hip_fr_prefix_product(self._ctx, _u64(d), d.shape[0]))
        return d

    def fr_prefix_sum(self, data):
        d = np.array(_fr(data), dtype=np.uint64)
        self._check(self._lib.plonk_hip_fr_prefix_sum(self._ctx, _u64(d), d.shape[0]))
        return d

    def fr_batch_inverse(self, data):
        """Element-wise inverse; raises ReferencePanic on a zero element (`inverse().unwrap()`)."""
        d = np.array(_fr(data), dtype=np.uint64)
        self._check(self._lib.plonk_hip_fr_batch_inverse(self._ctx, _u64(d), d.shape[0]))
        return d

    def fr_linear_combination(self, x, y, a, b, c, broadcast=False):
        """a*x + b*y + c (c on coefficient 0 only, or on every element with broadcast): x, y dense
        coefficient or value vectors of any lengths; a, b, c field elements (4 limbs)."""
        x, y = _fr(x), _fr(y)
        n = max(x.shape[0], y.shape[0])
        out = np.empty((n, 4), dtype=np.uint64)
        self._check(self._lib.plonk_hip_fr_linear_combination(
            self._ctx, _u64(x) if x.shape[0] else None, x.shape[0], _u64(y) if y.shape[0] else None, y.shape[0],
            _u64(_fr(a).reshape(4)), _u64(_fr(b).reshape(4)), _u64(_fr(c).reshape(4)), 1 if broadcast else 0,
            _u64(out) if n else None))
        return out

    # ------------------------------------------------------------------ device-resident polynomials
    def memcpy_d2d(self, dst, src, nbytes):
        self._check(self._lib.plonk_hip_memcpy_d2d(self._ctx, dst, src, nbytes))

    def memset_zero(self, dst, nbytes):
        self._check(self._lib.plonk_hip_memset_zero(self._ctx, dst, nbytes))

    def kzg_open_many_dev(self, d_coeffs, n, us):
        us = _fr(us)
        vs = np.empty((us.shape[0], 4), dtype=np.uint64)
        proofs = np.empty((us.shape[0], 18), dtype=np.uint64)
        self._check(self._lib.plonk_hip_kzg_open_many_dev(self._ctx, d_coeffs, n, _u64(us), us.shape[0], _u64(vs),
                                                          _u64(proofs)))
        return vs, proofs

    def kzg_commit_many_dev(self, polys):
        """polys: list of (device pointer, length) -> (len(polys), 18) commitments on the host."""
        ptrs = (ctypes.c_void_p * len(polys))(*[p for p, _ in polys])
        ns = (ctypes.c_size_t * len(polys))(*[n for _, n in polys])
        outs = np.empty((len(polys), 18), dtype=np.uint64)
        self._check(self._lib.plonk_hip_kzg_commit_many_dev(self._ctx, ptrs, ns, len(polys), _u64(outs)))
        return outs

    def kzg_commit_open_many_dev(self, tasks):
        """tasks: list of (device pointer, length, point or None): a commitment (None) or an opening at the
        point (4 limbs).  Returns (values (m, 4) -- zero for commitments --, group elements (m, 18))."""
        m = len(tasks)
        ptrs = (ctypes.c_void_p * m)(*[t[0] for t in tasks])
        ns = (ctypes.c_size_t * m)(*[t[1] for t in tasks])
        kinds = (ctypes.c_uint8 * m)(*[0 if t[2] is None else 1 for t in tasks])
        us = np.zeros((max(m, 1), 4), dtype=np.uint64)
        for k, t in enumerate(tasks):
            if t[2] is not None:
                us[k] = _fr(t[2]).reshape(4)
        vs = np.zeros((max(m, 1), 4), dtype=np.uint64)
        outs = np.zeros((max(m, 1), 18), dtype=np.uint64)
        self._check(self._lib.plonk_hip_kzg_commit_open_many_dev(self._ctx, ptrs, ns, kinds, _u64(us), m, _u64(vs),
                                                                 _u64(outs)))
        return vs[:m], outs[:m]

    def fr_scale_by_powers_dev(self, d_coeffs, n, w):
        self._check(self._lib.plonk_hip_fr_scale_by_powers_dev(self._ctx, d_coeffs, n, _u64(_fr(w).reshape(4))))

    def fr_poly_mul_dev(self, d_a, na, d_b, nb, d_out):
        self._check(self._lib.plonk_hip_fr_poly_mul_dev(self._ctx, d_a, na, d_b, nb, d_out))

    def fr_linear_combination_dev(self, d_x, nx, d_y, ny, a, b, c, broadcast, d_out):
        self._check(self._lib.plonk_hip_fr_linear_combination_dev(
            self._ctx, d_x, nx, d_y, ny, _u64(_fr(a).reshape(4)), _u64(_fr(b).reshape(4)), _u64(_fr(c).reshape(4)),
            1 if broadcast else 0, d_out))

    def fr_div_by_roots_dev(self, d_f, n, d_roots, m, d_q):
        self._check(self._lib.plonk_hip_fr_div_by_roots_dev(self._ctx, d_f, n, d_roots, m, d_q))

    def fr_div_vanishing_dev(self, d_f, n, k, d_q):
        self._check(self._lib.plonk_hip_fr_div_vanishing_dev(self._ctx, d_f, n, k, d_q))

    def fr_respects_permutation_dev(self, d_values, perm):
        """values[perm[i]] == values[i] for every i (perm: integer array, on the host)."""
        perm = np.ascontiguousarray(perm, dtype=np.uint32)
        flag = ctypes.c_int(0)
        self._check(self._lib.plonk_hip_fr_respects_permutation_dev(
            self._ctx, d_values, perm.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), perm.shape[0],
            ctypes.byref(flag)))
        return bool(flag.value)

    def fr_is_zero_dev(self, d_x, n):
        flag = ctypes.c_int(0)
        self._check(self._lib.plonk_hip_fr_is_zero_dev(self._ctx, d_x, n, ctypes.byref(flag)))
        return bool(flag.value)

    def fr_ratio_prefix_product_dev(self, d_num, d_den, n):
        self._check(self._lib.plonk_hip_fr_ratio_prefix_product_dev(self._ctx, d_num, d_den, n))

    def fr_ratio_prefix_product_dev_noalloc(self, d_num, d_den, n, d_zero_flag):
        """The same through the entry point that allocates nothing; d_zero_flag: a device int."""
        self._check(self._lib.plonk_hip_fr_ratio_prefix_product_dev_noalloc(self._ctx, d_num, d_den, n, d_zero_flag))

    def fr_ntt_mixed_dev(self, dptr, n, batch=1, inverse=False):
        self._check(self._lib.plonk_hip_fr_ntt_mixed_dev(self._ctx, dptr, n, batch, 1 if inverse else 0))

    def fr_ratio_prefix_product(self, num, den):
        """t_i = prod_{j<=i} num_j / den_j (protocols.rs:345-353, 571-588)."""
        num, den = _fr(num), _fr(den)
        assert num.shape == den.shape
        out = np.empty_like(num)
        self._check(self._lib.plonk_hip_fr_ratio_prefix_product(self._ctx, _u64(num), _u64(den), num.shape[0],
                                                                _u64(out)))
        return out

    def fr_poly_mul(self, a, b):
        """Dense product `&a * &b`, untrimmed (empty when either factor is empty)."""
        a, b = _fr(a), _fr(b)
        na, nb = a.shape[0], b.shape[0]
        out = np.zeros((na + nb - 1 if na and nb else 0, 4), dtype=np.uint64)
        if na and nb:
            self._check(self._lib.plonk_hip_fr_poly_mul(self._ctx, _u64(a), na, _u64(b), nb, _u64(out)))
        return out

    def construct_vanishing_polynomial_from_roots(self, roots):
        """prod (X - root_i) (utils.rs:51-62)."""
        roots = _fr(roots)
        z = np.empty((roots.shape[0] + 1, 4), dtype=np.uint64)
        self._check(self._lib.plonk_hip_fr_vanishing_from_roots(self._ctx, _u64(roots), roots.shape[0], _u64(z)))
        return z

    def compute_q_zero_test_from_roots(self, roots, f):
        """`compute_q_zero_test_from_roots(roots, f)` (protocols.rs:169-181), untrimmed."""
        roots, f = _fr(roots), _fr(f)
        n, m = f.shape[0], roots.shape[0]
        q = np.zeros((max(n - m, 0), 4), dtype=np.uint64)
        if n > m:
            self._check(self._lib.plonk_hip_fr_div_by_roots(self._ctx, _u64(f), n, _u64(roots), m, _u64(q)))
        return q

    def interpolate_polynomial(self, xs, ys):
        """`interpolate_polynomial(xs, ys)` (polynomials.rs:50-58) at arbitrary points, untrimmed."""
        xs, ys = _fr(xs), _fr(ys)
        assert xs.shape == ys.shape
        out = np.empty_like(xs)
        self._check(self._lib.plonk_hip_fr_interpolate(self._ctx, _u64(xs), _u64(ys), xs.shape[0], _u64(out)))
        return out

    def interpolate_geometric(self, q, ys):
        """`interpolate_polynomial([1, q, q^2, ...], ys)` in O(n log n), untrimmed; ReferencePanic when two of
        the points coincide."""
        ys = _fr(ys)
        out = np.empty_like(ys)
        self._check(self._lib.plonk_hip_fr_interpolate_geometric(self._ctx, _u64(_fr(q).reshape(4)), _u64(ys),
                                                                 ys.shape[0], _u64(out)))
        return out

    def interpolate_geometric_dev(self, q, d_values, n):
        self._check(self._lib.plonk_hip_fr_interpolate_geometric_dev(self._ctx, _u64(_fr(q).reshape(4)), d_values, n))

    def quotient_coset(self, f):
        """Batch of polynomials with 2k coefficients each -> (q (batch, k, 4), divisible flags)."""
        f = np.ascontiguousarray(f, dtype=np.uint64)
        f = f.reshape(-1, f.shape[-2], 4) if f.ndim == 3 else f.reshape(1, -1, 4)
        n = f.shape[1]
        k = n // 2
        log_k = k.bit_length() - 1
        assert 2 << log_k == n
        q = np.empty((f.shape[0], k, 4), dtype=np.uint64)
        flags = (ctypes.c_int * f.shape[0])()
        self._check(self._lib.plonk_hip_fr_quotient_coset(self._ctx, _u64(f), log_k, f.shape[0], _u64(q), flags))
        return q, [bool(x) for x in flags]

    def quotient_coset_dev(self, dptr, log_k, batch, d_flags=None):
        self._check(self._lib.plonk_hip_fr_quotient_coset_dev(self._ctx, dptr, log_k, batch, d_flags))


class MultiBackend:
    """Several GPUs behind one handle in one process (plonk_hip_multi_*): the SRS split by point range,
    `kzg_commit` / `kzg_evaluate` sharded inside the library.  devices: device ordinals, one sub-context
    each (an ordinal may repeat)."""

    def __init__(self, devices, window_bits=0):
        self._lib = _l.load_library()
        self._h = _l.c_ctx()
        arr = (ctypes.c_int * len(devices))(*devices)
        rc = self._lib.plonk_hip_multi_create(arr, len(devices), ctypes.byref(self._h))
        if rc != 0:
            raise PlonkHipError(rc, self._lib.plonk_hip_strerror(rc).decode())
        if window_bits:
            self._check(self._lib.plonk_hip_multi_set_window_bits(self._h, window_bits))

    def close(self):
        if self._h:
            self._lib.plonk_hip_multi_destroy(self._h)
            self._h = _l.c_ctx()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == 0:
            return
        text = self._lib.plonk_hip_strerror(rc).decode()
        detail = self._lib.plonk_hip_multi_last_error(self._h).decode()
        if detail and rc in (-6, -7):
            text += " (" + detail + ")"
        if rc == ERR_SRS_TOO_SHORT:
            raise CommitError(rc, text)
        if rc in (ERR_EMPTY_POLY, ERR_ZERO_DENOMINATOR, ERR_INVALID_POINT):
            raise ReferencePanic(rc, text)
        raise PlonkHipError(rc, text)

    def device_count(self):
        return self._lib.plonk_hip_multi_device_count(self._h)

    def srs_range(self, index):
        first, count = ctypes.c_size_t(), ctypes.c_size_t()
        self._check(self._lib.plonk_hip_multi_srs_range(self._h, index, ctypes.byref(first), ctypes.byref(count)))
        return first.value, count.value

    def srs_load_projective(self, pts):
        pts = _fr(pts, 18)
        self._check(self._lib.plonk_hip_multi_srs_load_projective(self._h, _u64(pts), pts.shape[0]))

    def kzg_setup(self, tau, degree, want_points=True):
        tau = _fr(tau).reshape(4)
        out = np.empty((degree + 1, 18), dtype=np.uint64) if want_points else None
        self._check(self._lib.plonk_hip_multi_kzg_setup(self._h, _u64(tau), degree, _u64(out) if want_points else None))
        return out

    def kzg_commit(self, coeffs):
        coeffs = _fr(coeffs)
        out = np.empty(18, dtype=np.uint64)
        self._check(self._lib.plonk_hip_multi_kzg_commit(self._h, _u64(coeffs), coeffs.shape[0], _u64(out)))
        return out

    def kzg_commit_batch(self, polys):
        polys = [_fr(p) for p in polys]
        ptrs = (_l.c_u64p * len(polys))(*[_u64(p) for p in polys])
        ns = (ctypes.c_size_t * len(polys))(*[p.shape[0] for p in polys])
        outs = np.empty((len(polys), 18), dtype=np.uint64)
        self._check(self._lib.plonk_hip_multi_kzg_commit_batch(self._h, ptrs, ns, len(polys), _u64(outs)))
        return outs

    def kzg_evaluate(self, coeffs, u):
        coeffs = _fr(coeffs)
        u = _fr(u).reshape(4)
        v = np.empty(4, dtype=np.uint64)
        proof = np.empty(18, dtype=np.uint64)
        self._check(self._lib.plonk_hip_multi_kzg_open(self._h, _u64(coeffs), coeffs.shape[0], _u64(u), _u64(v), _u64(proof)))
        return v, proof
