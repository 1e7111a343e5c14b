"""Pins the C oracle (oracle/oracle.c) against oracle/pyref.py and the golden vectors."""
import random

import numpy as np
import pytest

from oracle import capi as c
from oracle import pyref as o


@pytest.fixture(scope="module", autouse=True)
def _built():
    c.build()


def test_field_arithmetic():
    rnd = random.Random(1)
    for mod, to_m, from_m, mul, add, sub in (
        (o.R, c.fr_to_mont, c.fr_from_mont, c.fr_mul, c.fr_add, c.fr_sub),
        (o.Q, c.fq_to_mont, c.fq_from_mont, c.fq_mul, c.fq_add, c.fq_sub),
    ):
        a = [rnd.randrange(mod) for _ in range(64)] + [0, 1, mod - 1, mod - 1]
        b = [rnd.randrange(mod) for _ in range(64)] + [mod - 1, 0, mod - 1, 1]
        A, B = to_m(a), to_m(b)
        assert from_m(A) == a
        assert from_m(mul(A, B)) == [x * y % mod for x, y in zip(a, b)]
        assert from_m(add(A, B)) == [(x + y) % mod for x, y in zip(a, b)]
        assert from_m(sub(A, B)) == [(x - y) % mod for x, y in zip(a, b)]
    assert c.limbs_to_ints(c.fr_to_mont([1]))[0] == pow(2, 256, o.R)
    assert c.limbs_to_ints(c.fq_to_mont([1]))[0] == pow(2, 384, o.Q)
    a = [rnd.randrange(1, o.R) for _ in range(8)]
    assert c.fr_from_mont(c.fr_inv(c.fr_to_mont(a))) == [pow(x, -1, o.R) for x in a]


def test_g1_group_law():
    rnd = random.Random(2)
    g = c.g1_generator()
    assert c.g1_from_limbs(g)[0] == o.G1_GENERATOR
    k1, k2 = rnd.randrange(o.R), rnd.randrange(o.R)
    P, S = c.g1_mul(g, c.fr_to_mont([k1])[0]), c.g1_mul(g, c.fr_to_mont([k2])[0])
    Pp, Sp = o.g1_mul(o.G1_GENERATOR, k1), o.g1_mul(o.G1_GENERATOR, k2)
    assert c.g1_from_limbs(P)[0] == Pp  # same ladder, same Jacobian representative
    assert c.g1_compress(c.g1_add(P, S))[0] == o.g1_compress(o.g1_add(Pp, Sp))
    assert c.g1_compress(c.g1_add(P, P))[0] == o.g1_compress(o.g1_double(Pp))
    assert c.g1_compress(c.g1_double(P))[0] == o.g1_compress(o.g1_double(Pp))
    ident = o.g1_compress(o.G1_IDENTITY)
    assert c.g1_compress(c.g1_add(P, c.g1_neg(P)))[0] == ident
    assert c.g1_compress(c.g1_identity())[0] == ident
    assert c.g1_compress(c.g1_mul(g, c.fr_to_mont([0])[0]))[0] == ident
    assert c.g1_eq(c.g1_add(P, c.g1_identity()), P)
    aff = c.g1_to_affine(np.stack([P, c.g1_identity()]))
    assert not aff[1].any()
    assert c.g1_eq(c.g1_from_affine(aff)[0], P)


def test_naive_msm_reproduces_verification_key(golden):
    # two MSM known answers from the reference: com_S (3 points), com_W (12 points)
    setup = o.load_setup_json(golden["srs"])
    pts = c.g1_to_limbs(setup.gp.tau_powers_g1)
    aff = c.g1_to_affine(pts)
    for name in "SW":
        sc = c.fr_to_mont([int(x) for x in golden["proving_key"][name]])
        want = golden["verification_key"]["com_" + name]
        assert c.g1_compress(c.msm_naive(pts, sc))[0].hex() == want
        assert c.g1_compress(c.msm_naive(pts, sc, threads=3))[0].hex() == want
        for win in (2, 5, 12, 16):
            assert c.g1_compress(c.msm_pippenger(aff, sc, c=win, threads=2))[0].hex() == want


def test_pippenger_equals_naive_and_commit_kat():
    rnd = random.Random(3)
    n = 200
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    srs = c.kzg_setup(tau, n - 1)
    assert c.g1_eq(srs[1], c.g1_mul(c.g1_generator(), tau))
    sc = c.fr_to_mont([rnd.randrange(o.R) for _ in range(n - 4)] + [0, 1, o.R - 1, o.R - 2])
    naive = c.msm_naive(srs, sc, threads=4)
    aff = c.g1_to_affine(srs)
    for win in (3, 8, 13, 15, 16, 17):
        assert c.g1_eq(naive, c.msm_pippenger(aff, sc, c=win, threads=4))
    # plonk/tests/kzg_tests.rs:36-69: commit == G * f(tau)
    assert c.g1_eq(naive, c.g1_mul(c.g1_generator(), c.fr_eval(sc, tau)))


def test_transform_and_polynomials():
    rnd = random.Random(4)
    for logn in (0, 1, 2, 3, 6):
        n = 1 << logn
        co = [rnd.randrange(o.R) for _ in range(n)]
        ev = c.fr_from_mont(c.fr_ntt(c.fr_to_mont(co)))
        if n > 1:
            assert ev == o.dft_naive(co, n)
        assert c.fr_from_mont(c.fr_ntt(c.fr_to_mont(ev), inverse=True)) == co
    a = [rnd.randrange(o.R) for _ in range(40)]
    b = [rnd.randrange(o.R) for _ in range(7)]
    q, r = c.poly_divmod(c.fr_to_mont(a), c.fr_to_mont(b))
    qq, rr = o.poly_divmod(a, b)
    assert o.poly_trim(c.fr_from_mont(q)) == qq and o.poly_trim(c.fr_from_mont(r)) == rr
    assert o.poly_trim(c.fr_from_mont(c.poly_mul(c.fr_to_mont(a), c.fr_to_mont(b)))) == o.poly_mul(a, b)
    xs = [rnd.randrange(o.R) for _ in range(20)]
    ys = [rnd.randrange(o.R) for _ in range(20)]
    got = c.fr_from_mont(c.interpolate_lagrange(c.fr_to_mont(xs), c.fr_to_mont(ys), threads=3))
    assert o.poly_trim(got) == o.interpolate_polynomial(xs, ys)
    u = rnd.randrange(o.R)
    assert c.fr_from_mont(c.fr_eval(c.fr_to_mont(a), c.fr_to_mont([u])[0]))[0] == o.poly_eval(a, u)
