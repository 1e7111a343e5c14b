"""The gfx950 kernels inline plonk_amd/csrc/{field,g1,msm_core}.cuh.  Those headers are
__host__ __device__; here the host build is checked against the oracle so that arithmetic
errors are caught without a GPU."""
import ctypes
import random

import numpy as np
import pytest

from oracle import capi as c
from oracle import pyref as o
from tests.hostcheck import build as hc_build

u32p = ctypes.POINTER(ctypes.c_uint32)


@pytest.fixture(scope="module")
def hc():
    c.build()
    return hc_build.load()


def _p32(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


def _vec2(fn, a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    out = np.empty_like(a)
    fn(_p32(a), _p32(b), _p32(out), ctypes.c_size_t(a.shape[0]))
    return out


def _vec1(fn, a):
    a = np.ascontiguousarray(a)
    out = np.empty_like(a)
    fn(_p32(a), _p32(out), ctypes.c_size_t(a.shape[0]))
    return out


@pytest.mark.parametrize("field", ["fq", "fr"])
def test_field_ops(hc, field):
    rnd = random.Random(21)
    mod = o.Q if field == "fq" else o.R
    to_m = c.fq_to_mont if field == "fq" else c.fr_to_mont
    from_m = c.fq_from_mont if field == "fq" else c.fr_from_mont
    a = [rnd.randrange(mod) for _ in range(200)] + [0, 1, mod - 1, mod - 1, 0, 2]
    b = [rnd.randrange(mod) for _ in range(200)] + [mod - 1, 0, mod - 1, 1, 0, mod - 2]
    A, B = to_m(a), to_m(b)
    f = lambda name: getattr(hc, "hc_%s_%s" % (field, name))
    assert from_m(_vec2(f("mul"), A, B)) == [x * y % mod for x, y in zip(a, b)]
    assert from_m(_vec2(f("add"), A, B)) == [(x + y) % mod for x, y in zip(a, b)]
    assert from_m(_vec2(f("sub"), A, B)) == [(x - y) % mod for x, y in zip(a, b)]
    assert from_m(_vec1(f("dbl"), A)) == [2 * x % mod for x in a]
    # Montgomery conversions: canonical limbs in, Montgomery limbs out and back
    canon = c.ints_to_limbs(a, 6 if field == "fq" else 4)
    assert np.array_equal(_vec1(f("to_mont"), canon), A)
    assert np.array_equal(_vec1(f("from_mont"), A), canon)
    inv = from_m(_vec1(f("inv"), A[:12]))
    assert inv == [pow(x, -1, mod) for x in a[:12]]
    assert from_m(_vec1(f("inv"), to_m([0])))[0] == 0


def _xyzz_to_point(hc, x):
    """XYZZ limbs (48 u32 as 24 u64) -> pyref Jacobian triple via the header's own converter."""
    out = np.empty(18, dtype=np.uint64)
    hc.hc_xyzz_to_jac(_p32(np.ascontiguousarray(x)), _p32(out))
    return c.g1_from_limbs(out)[0]


def test_group_law(hc):
    rnd = random.Random(22)
    g = c.g1_generator()
    pts = [c.g1_mul(g, c.fr_to_mont([rnd.randrange(1, o.R)])[0]) for _ in range(6)]
    aff = c.g1_to_affine(np.stack(pts))
    ref = c.g1_from_limbs(np.stack(pts))
    ident = np.zeros(24, dtype=np.uint64)
    assert o.g1_is_identity(_xyzz_to_point(hc, ident))

    def madd(acc, a, neg):
        out = np.empty(24, dtype=np.uint64)
        hc.hc_xyzz_madd(_p32(np.ascontiguousarray(acc)), _p32(np.ascontiguousarray(a)), int(neg), _p32(out))
        return out

    def add(a, b):
        out = np.empty(24, dtype=np.uint64)
        hc.hc_xyzz_add(_p32(np.ascontiguousarray(a)), _p32(np.ascontiguousarray(b)), _p32(out))
        return out

    def dbl(a):
        out = np.empty(24, dtype=np.uint64)
        hc.hc_xyzz_dbl(_p32(np.ascontiguousarray(a)), _p32(out))
        return out

    # identity + P, then generic adds, P + P, P - P, identity operand
    acc = madd(ident, aff[0], False)
    assert o.g1_eq(_xyzz_to_point(hc, acc), ref[0])
    acc = madd(acc, aff[1], False)
    want = o.g1_add(ref[0], ref[1])
    assert o.g1_eq(_xyzz_to_point(hc, acc), want)
    acc = madd(acc, aff[2], True)
    want = o.g1_sub(want, ref[2])
    assert o.g1_eq(_xyzz_to_point(hc, acc), want)
    assert o.g1_eq(_xyzz_to_point(hc, madd(acc, np.zeros(12, dtype=np.uint64), False)), want)
    one = madd(ident, aff[3], False)
    assert o.g1_eq(_xyzz_to_point(hc, madd(one, aff[3], False)), o.g1_double(ref[3]))
    assert o.g1_is_identity(_xyzz_to_point(hc, madd(one, aff[3], True)))
    # doubling of a non-trivial XYZZ value, then P + P through the full add
    assert o.g1_eq(_xyzz_to_point(hc, dbl(acc)), o.g1_double(want))
    assert o.g1_eq(_xyzz_to_point(hc, add(acc, acc)), o.g1_double(want))
    other = madd(madd(ident, aff[4], False), aff[5], False)
    w2 = o.g1_add(ref[4], ref[5])
    assert o.g1_eq(_xyzz_to_point(hc, add(acc, other)), o.g1_add(want, w2))
    assert o.g1_eq(_xyzz_to_point(hc, add(acc, ident)), want)
    assert o.g1_eq(_xyzz_to_point(hc, add(ident, other)), w2)
    # a + (-a): negate through a Jacobian round trip
    neg = c.g1_to_limbs([o.g1_neg(want)])[0]
    nx = np.empty(24, dtype=np.uint64)
    hc.hc_jac_to_xyzz(_p32(np.ascontiguousarray(neg)), _p32(nx))
    assert o.g1_is_identity(_xyzz_to_point(hc, add(acc, nx)))
    # Jacobian doubling used by the SRS window tables
    jd = np.empty(18, dtype=np.uint64)
    hc.hc_jac_dbl(_p32(np.ascontiguousarray(pts[0])), _p32(jd))
    assert o.g1_eq(c.g1_from_limbs(jd)[0], o.g1_double(ref[0]))
    hc.hc_jac_dbl(_p32(np.ascontiguousarray(c.g1_identity())), _p32(jd))
    assert o.g1_is_identity(c.g1_from_limbs(jd)[0])


@pytest.mark.parametrize("cbits", [2, 3, 7, 8, 13, 15, 16, 17, 20, 24])
def test_signed_digits_recompose(hc, cbits):
    rnd = random.Random(23 + cbits)
    nwin = hc.hc_num_windows(cbits)
    assert nwin == -(-255 // cbits)
    half_r = (o.R - 1) // 2
    vals = [0, 1, o.R - 1, o.R - 2, (1 << 255) % o.R, (1 << 254) - 1, half_r, half_r + 1, half_r - 1]
    vals += [rnd.randrange(o.R) for _ in range(50)]
    half = 1 << (cbits - 1)
    # worst case for carries: every window at its maximum
    vals.append(min(o.R - 1, (1 << 255) - 1))
    # ... and of the folded scalar: the largest value below (r + 1) / 2 with all low windows full
    vals.append(half_r - (half_r % (1 << (cbits * (nwin - 2)))) - 1)
    for v in vals:
        sm = np.ascontiguousarray(c.fr_to_mont([v])[0])
        out = np.zeros(nwin, dtype=np.int32)
        carry = ctypes.c_uint32(7)
        opposite = hc.hc_scalar_digits(_p32(sm), cbits, nwin, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                       ctypes.byref(carry))
        assert opposite == (1 if 255 % cbits == 0 and v > half_r else 0)   # only widths that divide 255 fold
        assert carry.value == 0                       # the top window absorbs everything
        assert all(-half < int(d) <= half for d in out)
        # the digits are those of the smaller of v and r - v
        assert sum(int(d) << (cbits * w) for w, d in enumerate(out)) == (o.R - v if opposite else v)


def test_28_bit_limb_form_matches_fq(hc):
    rnd = random.Random(31)
    a = [rnd.randrange(o.Q) for _ in range(300)] + [0, 1, o.Q - 1, o.Q - 1, 0, 2, o.Q - 2]
    b = [rnd.randrange(o.Q) for _ in range(300)] + [o.Q - 1, 0, o.Q - 1, 1, 0, o.Q - 2, o.Q - 1]
    A, B = c.fq_to_mont(a), c.fq_to_mont(b)
    assert np.array_equal(_vec1(hc.hc_q28_roundtrip, A), A)
    assert c.fq_from_mont(_vec2(hc.hc_q28_mul, A, B)) == [x * y % o.Q for x, y in zip(a, b)]
    assert c.fq_from_mont(_vec2(hc.hc_q28_add, A, B)) == [(x + y) % o.Q for x, y in zip(a, b)]
    assert c.fq_from_mont(_vec2(hc.hc_q28_sub, A, B)) == [(x - y) % o.Q for x, y in zip(a, b)]
    assert c.fq_from_mont(_vec1(hc.hc_q28_sqr, A)) == [x * x % o.Q for x in a]
    # canonical limbs: value * 2^392 mod q sliced into 28-bit limbs
    limbs = np.zeros((len(a), 14), dtype=np.uint32)
    hc.hc_q28_limbs(_p32(np.ascontiguousarray(A)), limbs.ctypes.data_as(u32p), ctypes.c_size_t(len(a)))
    for x, row in zip(a, limbs):
        want = x * pow(2, 392, o.Q) % o.Q
        assert [int(v) for v in row] == [(want >> (28 * i)) & 0xFFFFFFF for i in range(14)]


def test_28_bit_bucket_accumulation(hc):
    rnd = random.Random(32)
    g = c.g1_generator()
    pts = [c.g1_mul(g, c.fr_to_mont([rnd.randrange(1, o.R)])[0]) for _ in range(5)]
    ref = c.g1_from_limbs(np.stack(pts))
    aff = c.g1_to_affine(np.stack(pts))
    ident_aff = np.zeros(12, dtype=np.uint64)

    def fold_with(fn, seq):
        a = np.ascontiguousarray(np.stack([x for x, _ in seq]))
        neg = (ctypes.c_int * len(seq))(*[int(s) for _, s in seq])
        out = np.empty(24, dtype=np.uint64)
        fn(_p32(a), neg, ctypes.c_size_t(len(seq)), _p32(out))
        return _xyzz_to_point(hc, out)

    def fold(seq):
        strict = fold_with(hc.hc_xyzz28_fold, seq)
        lazy = fold_with(hc.hc_xyzz28_fold_lazy, seq)  # the accumulation loop's lazy arithmetic
        tree = fold_with(hc.hc_xyzz28_tree_lazy, seq)  # the tree sums' lazy full additions
        assert o.g1_to_affine(strict) == o.g1_to_affine(lazy) == o.g1_to_affine(tree)
        # the same tree through the additions staged over four waves, then 5 staged doublings
        for dbl in (0, 5):
            a = np.ascontiguousarray(np.stack([x for x, _ in seq]))
            neg = (ctypes.c_int * len(seq))(*[int(sg) for _, sg in seq])
            out = np.empty(24, dtype=np.uint64)
            hc.hc_xyzz28_tree_coop(_p32(a), neg, ctypes.c_size_t(len(seq)), ctypes.c_int(dbl), _p32(out))
            want = strict
            for _ in range(dbl):
                want = o.g1_double(want)
            assert o.g1_to_affine(_xyzz_to_point(hc, out)) == o.g1_to_affine(want)
        return lazy

    rt = np.empty(12, dtype=np.uint64)
    hc.hc_affine28_roundtrip(_p32(np.ascontiguousarray(aff[0])), _p32(rt))
    assert np.array_equal(rt, aff[0])
    assert o.g1_eq(fold([(aff[0], 0)]), ref[0])
    assert o.g1_eq(fold([(aff[0], 0), (aff[1], 0), (aff[2], 1)]), o.g1_sub(o.g1_add(ref[0], ref[1]), ref[2]))
    assert o.g1_eq(fold([(aff[3], 0), (aff[3], 0)]), o.g1_double(ref[3]))          # doubling branch
    assert o.g1_is_identity(fold([(aff[3], 0), (aff[3], 1)]))                      # P - P
    assert o.g1_eq(fold([(aff[3], 0), (aff[3], 1), (aff[4], 1)]), o.g1_neg(ref[4]))  # restart from identity
    assert o.g1_eq(fold([(ident_aff, 0), (aff[1], 0), (ident_aff, 1)]), ref[1])    # identity operands
    want = o.G1_IDENTITY
    seq = []
    for i in range(400):
        j, sgn = rnd.randrange(5), rnd.randrange(2)
        seq.append((aff[j], sgn))
        want = o.g1_add(want, o.g1_neg(ref[j]) if sgn else ref[j])
    assert o.g1_eq(fold(seq), want)


def test_29_bit_limb_form_of_fr(hc):
    rnd = random.Random(33)
    xs = [rnd.randrange(o.R) for _ in range(200)] + [0, 1, o.R - 1, o.R - 1]
    ws = [rnd.randrange(o.R) for _ in range(200)] + [o.R - 1, 0, o.R - 1, 1]
    X, W = c.fr_to_mont(xs), c.fr_to_mont(ws)
    assert np.array_equal(_vec1(hc.hc_fr29_roundtrip, X), X)
    # any 256-bit pattern survives the re-slicing
    raw = np.frombuffer(rnd.randbytes(32 * 50), dtype=np.uint64).reshape(50, 4).copy()
    assert np.array_equal(_vec1(hc.hc_fr29_roundtrip, raw), raw)
    assert c.fr_from_mont(_vec2(hc.hc_fr29_mul, X, W)) == [x * w % o.R for x, w in zip(xs, ws)]
    # lazy butterfly chain
    for steps in (1, 4, 5, 9, 10, 20):
        a, b, w = rnd.randrange(o.R), rnd.randrange(o.R), rnd.randrange(o.R)
        AB = np.ascontiguousarray(c.fr_to_mont([a, b]))
        Wm = np.ascontiguousarray(c.fr_to_mont([w]))
        s_out = np.empty(4, dtype=np.uint64)
        d_out = np.empty(4, dtype=np.uint64)
        hc.hc_fr29_chain(_p32(AB), _p32(Wm), steps, _p32(s_out), _p32(d_out))
        for _ in range(steps):
            m = b * w % o.R
            a, b = (a + m) % o.R, (a - m) % o.R
        assert c.fr_from_mont(s_out)[0] == a and c.fr_from_mont(d_out)[0] == b
