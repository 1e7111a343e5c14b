"""The literal known answers of the reference's own unit tests (plonk/tests/polynomials_tests.rs,
utils_tests.rs, kzg_tests.rs), run through the C ABI wherever the library has the operation: small integers in,
small integers out, no oracle in between except as the second opinion on the random cases."""
import random

import numpy as np
import pytest

import plonk_amd
from oracle import capi as c
from oracle import pyref as o
from plonk_amd import backend as B
from plonk_amd.prover import trim
from tests.gpu_util import backend, compress  # noqa: F401

pytestmark = pytest.mark.gpu


def limbs(values):
    return c.fr_to_mont([v % o.R for v in values])


def ints(arr):
    return c.fr_from_mont(np.ascontiguousarray(arr).reshape(-1, 4))


def test_polynomial_known_answers(backend):
    # polynomials_tests.rs:33-43: (1 + x)(1 + x) = 1 + 2x + x^2
    assert ints(backend.fr_poly_mul(limbs([1, 1]), limbs([1, 1]))) == [1, 2, 1]
    # :45-55: (-1 + x^2) / (-1 + x) = 1 + x
    assert ints(trim(backend.compute_q_zero_test_from_roots(limbs([1]), limbs([-1, 0, 1])))) == [1, 1]
    # :57-92: (4 - 5x + x^9) = (1 + x^2)(-x + x^3 - x^5 + x^7) + (4 - 4x); 1 + x^2 = (x - i)(x + i), i of order 4
    i4 = o.construct_Omega(4)[1]
    a = limbs([4, -5, 0, 0, 0, 0, 0, 0, 0, 1])
    roots = limbs([i4, -i4])
    assert ints(backend.construct_vanishing_polynomial_from_roots(roots)) == [1, 0, 1]
    q = trim(backend.compute_q_zero_test_from_roots(roots, a))
    assert ints(q) == [0, o.R - 1, 0, 1, 0, o.R - 1, 0, 1]
    r = trim(backend.fr_linear_combination(a, backend.fr_poly_mul(q, limbs([1, 0, 1])), limbs([1])[0], limbs([-1])[0],
                                           limbs([0])[0]))
    assert ints(r) == [4, o.R - 4]
    # :152-158: (x + x^2)(5) = 30
    assert ints(backend.fr_eval(limbs([0, 1, 1]), limbs([5])[0])) == [30]
    # :200-215: (1 + x)^2, (1 + x)^3
    sq = backend.fr_poly_mul(limbs([1, 1]), limbs([1, 1]))
    assert ints(sq) == [1, 2, 1] and ints(backend.fr_poly_mul(sq, limbs([1, 1]))) == [1, 3, 3, 1]
    # :160-177: through (0, 0), (1, 1), (2, 8): -2x + 3x^2
    assert ints(trim(backend.interpolate_polynomial(limbs([0, 1, 2]), limbs([0, 1, 8])))) == [0, o.R - 2, 3]
    # utils_tests.rs:24-52: Z of the subgroup of size 4, Z from the roots 2 and 3
    omega4 = limbs(o.construct_Omega(4))
    assert ints(backend.construct_vanishing_polynomial_from_roots(omega4)) == [o.R - 1, 0, 0, 0, 1]
    assert ints(backend.construct_vanishing_polynomial_from_roots(limbs([2, 3]))) == [6, o.R - 5, 1]


def test_division_by_monic_polynomials_with_remainder(backend):
    # polynomials_tests.rs:94-118 for divisors given by their roots: a = q b + r with deg r < deg b
    rnd = random.Random(94)
    for _ in range(20):
        deg_a, deg_b = rnd.randrange(0, 51), rnd.randrange(1, 51)
        a = [rnd.randrange(o.R) for _ in range(deg_a)] + [rnd.randrange(1, o.R)]
        roots = [rnd.randrange(o.R) for _ in range(deg_b)]
        b = o.construct_vanishing_polynomial_from_roots(roots)
        q = trim(backend.compute_q_zero_test_from_roots(limbs(roots), limbs(a)))
        want_q, want_r = o.poly_divmod(a, b)
        assert ints(q) == o.poly_trim(want_q)
        qb = backend.fr_poly_mul(q, limbs(b)) if q.shape[0] else np.zeros((0, 4), dtype=np.uint64)
        r = trim(backend.fr_linear_combination(limbs(a), qb, limbs([1])[0], limbs([-1])[0], limbs([0])[0]))
        assert ints(r) == o.poly_trim(want_r) and len(ints(r)) <= deg_b


def test_omega_has_order_k(backend):
    # utils_tests.rs:12-22, with the library's own powers: evaluating X^k over Omega gives ones, and X^(k/2) does not
    k = 8
    xk = np.zeros((k + 1, 4), dtype=np.uint64)
    xk[k] = limbs([1])[0]
    assert ints(backend.fr_evaluate_over_omega(xk, 3)) == [1] * k
    half = np.zeros((k // 2 + 1, 4), dtype=np.uint64)
    half[k // 2] = limbs([1])[0]
    assert ints(backend.fr_evaluate_over_omega(half, 3)) == [1, o.R - 1] * (k // 2)


def test_srs_powers_chain_through_the_pairing(backend, golden):
    # kzg_tests.rs:14-34: e(tau^(i+1) G, H) = e(tau^i G, tau H): the opening check of `tau^(i+1) G` at u = 0 with
    # value 0 and proof `tau^i G` (kzg.rs:91-98), folded over the whole SRS of plonk/data/srs.json
    setup = o.load_setup_json(golden["srs"])
    pts = c.g1_to_limbs(setup.gp.tau_powers_g1)
    tau_g2 = np.frombuffer(bytes.fromhex(golden["srs"]["gp"]["tau_g2"]), dtype=np.uint8)
    n = pts.shape[0]
    zeros = np.zeros((n - 1, 4), dtype=np.uint64)
    assert B.kzg_verify_batch(tau_g2, pts[1:], zeros, zeros, pts[:-1])
    assert all(B.kzg_verify(tau_g2, pts[i + 1], zeros[0], zeros[0], pts[i]) for i in (0, 7, n - 2))
    assert not B.kzg_verify(tau_g2, pts[5], zeros[0], zeros[0], pts[5])
