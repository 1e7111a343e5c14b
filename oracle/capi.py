"""ctypes view of oracle/_build/liboracle.so (TEST INFRASTRUCTURE ONLY -- see oracle.c).

Values cross this boundary as numpy uint64 arrays of little-endian limbs in Montgomery
form: Fr = 4 limbs, Fq = 6, G1 projective = 18 (X|Y|Z Jacobian), G1 affine = 12 (x|y,
identity = all zero).  Helpers convert to and from the Python ints used by oracle/pyref.py.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import pyref

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build():
    """Compile oracle.c (gcc, OpenMP) if the shared object is missing or stale."""
    src = os.path.join(_HERE, "oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_init()
        _lib.orc_g1_to_affine.restype = ctypes.c_size_t
    return _lib


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


# ----------------------------------------------------------------- int <-> limbs
def ints_to_limbs(vals, nlimbs):
    out = np.zeros((len(vals), nlimbs), dtype=np.uint64)
    mask = (1 << 64) - 1
    for i, v in enumerate(vals):
        for j in range(nlimbs):
            out[i, j] = (v >> (64 * j)) & mask
    return out


def limbs_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64)
    arr = arr.reshape(-1, arr.shape[-1])
    return [sum(int(x) << (64 * j) for j, x in enumerate(row)) for row in arr]


def fr_to_mont(vals):
    """Python ints (canonical) -> (n, 4) uint64 Montgomery limbs."""
    a = ints_to_limbs([v % pyref.R for v in vals], 4)
    out = np.empty_like(a)
    lib().orc_fr_to_mont(_p(a), _p(out), ctypes.c_size_t(len(vals)))
    return out


def fr_from_mont(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(arr)
    lib().orc_fr_from_mont(_p(arr), _p(out), ctypes.c_size_t(arr.shape[0]))
    return limbs_to_ints(out)


def fq_to_mont(vals):
    a = ints_to_limbs([v % pyref.Q for v in vals], 6)
    out = np.empty_like(a)
    lib().orc_fq_to_mont(_p(a), _p(out), ctypes.c_size_t(len(vals)))
    return out


def fq_from_mont(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 6)
    out = np.empty_like(arr)
    lib().orc_fq_from_mont(_p(arr), _p(out), ctypes.c_size_t(arr.shape[0]))
    return limbs_to_ints(out)


def g1_to_limbs(points):
    """List of pyref Jacobian triples -> (n, 18) Montgomery limbs."""
    flat = []
    for (x, y, z) in points:
        flat += [x, y, z]
    return fq_to_mont(flat).reshape(len(points), 18)


def g1_from_limbs(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 18)
    vals = fq_from_mont(arr.reshape(-1, 6))
    return [(vals[3 * i], vals[3 * i + 1], vals[3 * i + 2]) for i in range(arr.shape[0])]


# ----------------------------------------------------------------- wrapped routines
def _vec2(fn, a, b, w):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, w)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, w)
    out = np.empty_like(a)
    fn(_p(a), _p(b), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def fr_mul(a, b):
    return _vec2(lib().orc_fr_mul_vec, a, b, 4)


def fr_add(a, b):
    return _vec2(lib().orc_fr_add_vec, a, b, 4)


def fr_sub(a, b):
    return _vec2(lib().orc_fr_sub_vec, a, b, 4)


def fr_inv(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(a)
    lib().orc_fr_inv_vec(_p(a), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def fq_mul(a, b):
    return _vec2(lib().orc_fq_mul_vec, a, b, 6)


def fq_add(a, b):
    return _vec2(lib().orc_fq_add_vec, a, b, 6)


def fq_sub(a, b):
    return _vec2(lib().orc_fq_sub_vec, a, b, 6)


def g1_generator():
    out = np.empty(18, dtype=np.uint64)
    lib().orc_g1_generator(_p(out))
    return out


def g1_identity():
    out = np.empty(18, dtype=np.uint64)
    lib().orc_g1_identity(_p(out))
    return out


def g1_add(p, q):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    q = np.ascontiguousarray(q, dtype=np.uint64)
    out = np.empty(18, dtype=np.uint64)
    lib().orc_g1_add(_p(p), _p(q), _p(out))
    return out


def g1_double(p):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    out = np.empty(18, dtype=np.uint64)
    lib().orc_g1_double(_p(p), _p(out))
    return out


def g1_neg(p):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    out = np.empty(18, dtype=np.uint64)
    lib().orc_g1_neg(_p(p), _p(out))
    return out


def g1_mul(p, scalar_mont):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    s = np.ascontiguousarray(scalar_mont, dtype=np.uint64).reshape(4)
    out = np.empty(18, dtype=np.uint64)
    lib().orc_g1_mul(_p(p), _p(s), _p(out))
    return out


def g1_to_affine(pts):
    pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, 18)
    out = np.empty((pts.shape[0], 12), dtype=np.uint64)
    lib().orc_g1_to_affine(_p(pts), ctypes.c_size_t(pts.shape[0]), _p(out))
    return out


def g1_from_affine(aff):
    aff = np.ascontiguousarray(aff, dtype=np.uint64).reshape(-1, 12)
    out = np.empty((aff.shape[0], 18), dtype=np.uint64)
    lib().orc_g1_from_affine(_p(aff), ctypes.c_size_t(aff.shape[0]), _p(out))
    return out


def g1_eq(p, q):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    q = np.ascontiguousarray(q, dtype=np.uint64)
    return bool(lib().orc_g1_eq(_p(p), _p(q)))


def g1_compress(pts):
    """(n, 18) limbs -> list of 48-byte encodings."""
    pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, 18)
    out = np.empty((pts.shape[0], 48), dtype=np.uint8)
    lib().orc_g1_compress(_p(pts), ctypes.c_size_t(pts.shape[0]), out.ctypes.data_as(_u8p))
    return [bytes(row) for row in out]


def kzg_setup(tau_mont, degree):
    tau = np.ascontiguousarray(tau_mont, dtype=np.uint64).reshape(4)
    out = np.empty((degree + 1, 18), dtype=np.uint64)
    lib().orc_kzg_setup(_p(tau), ctypes.c_size_t(degree), _p(out))
    return out


def msm_naive(pts_proj, scalars_mont, threads=1):
    pts = np.ascontiguousarray(pts_proj, dtype=np.uint64).reshape(-1, 18)
    sc = np.ascontiguousarray(scalars_mont, dtype=np.uint64).reshape(-1, 4)
    n = sc.shape[0]
    assert pts.shape[0] >= n
    out = np.empty(18, dtype=np.uint64)
    lib().orc_msm_naive(_p(pts), _p(sc), ctypes.c_size_t(n), _p(out), ctypes.c_int(threads))
    return out


def msm_pippenger(pts_affine, scalars_mont, c=12, threads=1):
    pts = np.ascontiguousarray(pts_affine, dtype=np.uint64).reshape(-1, 12)
    sc = np.ascontiguousarray(scalars_mont, dtype=np.uint64).reshape(-1, 4)
    n = sc.shape[0]
    assert pts.shape[0] >= n
    out = np.empty(18, dtype=np.uint64)
    lib().orc_msm_pippenger(_p(pts), _p(sc), ctypes.c_size_t(n), _p(out), ctypes.c_int(c), ctypes.c_int(threads))
    return out


def fr_eval(coeffs, u):
    c = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    u = np.ascontiguousarray(u, dtype=np.uint64).reshape(4)
    out = np.empty(4, dtype=np.uint64)
    lib().orc_fr_eval(_p(c), ctypes.c_size_t(c.shape[0]), _p(u), _p(out))
    return out


def poly_divmod(a, b):
    """Long division; returns (q, rem) limb arrays with na-nb+1 and nb-1 rows (untrimmed)."""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    na, nb = a.shape[0], b.shape[0]
    if na < nb:
        return np.zeros((0, 4), dtype=np.uint64), a.copy()
    q = np.zeros((na - nb + 1, 4), dtype=np.uint64)
    rem = np.zeros((max(nb - 1, 1), 4), dtype=np.uint64)
    rc = lib().orc_poly_divmod(_p(a), ctypes.c_size_t(na), _p(b), ctypes.c_size_t(nb), _p(q), _p(rem))
    if rc != 0:
        raise ZeroDivisionError("leading coefficient of the divisor is zero")
    return q, rem[: nb - 1]


def poly_mul(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((a.shape[0] + b.shape[0] - 1, 4), dtype=np.uint64)
    lib().orc_poly_mul(_p(a), ctypes.c_size_t(a.shape[0]), _p(b), ctypes.c_size_t(b.shape[0]), _p(out))
    return out


def root_of_unity(logn):
    out = np.empty(4, dtype=np.uint64)
    lib().orc_fr_root_of_unity(ctypes.c_uint(logn), _p(out))
    return out


def fr_ntt(data, inverse=False):
    """Natural-order radix-2 transform of a (2^k, 4) array; returns a new array."""
    d = np.array(data, dtype=np.uint64).reshape(-1, 4)
    n = d.shape[0]
    logn = n.bit_length() - 1
    assert 1 << logn == n
    lib().orc_fr_ntt(_p(d), ctypes.c_uint(logn), ctypes.c_int(1 if inverse else 0))
    return d


def interpolate_lagrange(xs, ys, threads=1):
    xs = np.ascontiguousarray(xs, dtype=np.uint64).reshape(-1, 4)
    ys = np.ascontiguousarray(ys, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros_like(xs)
    rc = lib().orc_interpolate_lagrange(_p(xs), _p(ys), ctypes.c_size_t(xs.shape[0]), _p(out), ctypes.c_int(threads))
    if rc != 0:
        raise ZeroDivisionError("coincident interpolation points")
    return out


def num_threads():
    return int(lib().orc_num_threads())
