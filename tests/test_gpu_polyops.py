"""GPU parity of the polynomial building blocks behind the t / t1 / quotient builders
(protocols.rs:157-181, 235-275, 336-385, 453-493, 559-638; polynomials.rs:14-58; utils.rs:51-62)
against the CPU oracle, through the C ABI."""
import random

import numpy as np
import pytest

import plonk_amd
from oracle import capi as c
from oracle import pyref as o
from tests.gpu_util import backend, rand_fr  # noqa: F401

pytestmark = pytest.mark.gpu


def M(vals):
    return c.fr_to_mont(vals)


def U(arr):
    return c.fr_from_mont(arr)


@pytest.mark.parametrize("n", [1, 2, 7, 2048, 2049, 5000, 300000])
def test_prefix_product_and_sum(backend, n):
    rnd = random.Random(700 + n)
    vals = rand_fr(rnd, n)
    prod, acc = [], 1
    for v in vals:
        acc = acc * v % o.R
        prod.append(acc)
    assert U(backend.fr_prefix_product(M(vals))) == prod
    sums, acc = [], 0
    for v in vals:
        acc = (acc + v) % o.R
        sums.append(acc)
    assert U(backend.fr_prefix_sum(M(vals))) == sums


@pytest.mark.parametrize("n", [1, 3, 4, 5, 1000, 65536, 70001])
def test_batch_inverse(backend, n):
    rnd = random.Random(710 + n)
    vals = [rnd.randrange(1, o.R) for _ in range(n)]
    got = U(backend.fr_batch_inverse(M(vals)))
    if n <= 1000:
        assert got == [o.fr_inv(v) for v in vals]
    else:
        # x * x^-1 == 1 for all, exact inverses for a sample
        assert U(c.fr_mul(M(vals), M(got))) == [1] * n
        for i in rnd.sample(range(n), 20):
            assert got[i] == o.fr_inv(vals[i])


def test_batch_inverse_of_zero_is_the_reference_panic(backend):
    rnd = random.Random(720)
    for n, pos in ((1, 0), (10, 9), (70000, 12345)):
        vals = [rnd.randrange(1, o.R) for _ in range(n)]
        vals[pos] = 0  # protocols.rs:349 / :583: `.inverse().unwrap()` panics
        with pytest.raises(plonk_amd.ReferencePanic):
            backend.fr_batch_inverse(M(vals))
    with pytest.raises(plonk_amd.ReferencePanic):
        backend.fr_ratio_prefix_product(M([1, 2, 3]), M([5, 0, 7]))


def test_ratio_prefix_product_is_the_rational_product_check_t(backend):
    # protocols.rs:345-353 on the example-sized domain and on a larger one
    rnd = random.Random(730)
    for k in (12, 4096):
        num, den = rand_fr(rnd, k), [rnd.randrange(1, o.R) for _ in range(k)]
        want, acc = [], 1
        for a, b in zip(num, den):
            acc = acc * a % o.R * o.fr_inv(b) % o.R
            want.append(acc)
        assert U(backend.fr_ratio_prefix_product(M(num), M(den))) == want


def test_eval_many_on_omega_12(backend):
    # the example circuit's Omega has 12 elements (not a power of two): protocols.rs:246
    rnd = random.Random(740)
    Omega = o.construct_Omega(12)
    f = rand_fr(rnd, 12)
    got = U(backend.fr_eval_many(M(f), M(Omega)))
    assert got == [o.poly_eval(f, w) for w in Omega]
    pts = rand_fr(rnd, 300)
    f = rand_fr(rnd, 777)
    assert U(backend.fr_eval_many(M(f), M(pts))) == [o.poly_eval(f, x) for x in pts]


@pytest.mark.parametrize("log_k,n", [(3, 5), (3, 8), (4, 40), (10, 3000), (12, 4096)])
def test_evaluate_over_omega(backend, log_k, n):
    rnd = random.Random(750 + n)
    k = 1 << log_k
    f = rand_fr(rnd, n)
    got = U(backend.fr_evaluate_over_omega(M(f), log_k))
    Omega = o.construct_Omega(k)
    idx = range(k) if k <= 64 else rnd.sample(range(k), 16)
    for j in idx:
        assert got[j] == o.poly_eval(f, Omega[j])


@pytest.mark.parametrize("na,nb", [(1, 1), (2, 12), (12, 12), (33, 33), (40, 100), (1000, 1), (1500, 2500), (4096, 4096)])
def test_poly_mul(backend, na, nb):
    rnd = random.Random(760 + na + nb)
    a, b = rand_fr(rnd, na), rand_fr(rnd, nb)
    got = U(backend.fr_poly_mul(M(a), M(b)))
    assert got == U(c.poly_mul(M(a), M(b)))
    assert backend.fr_poly_mul(M(a), np.zeros((0, 4), dtype=np.uint64)).shape[0] == 0


def test_poly_mul_builds_t1_of_the_product_check(backend):
    # t1(X) = t(wX) - t(X) f(wX)   (protocols.rs:257-272) with device products and scalings
    rnd = random.Random(770)
    k = 12
    Omega = o.construct_Omega(k)
    f = rand_fr(rnd, k)
    t, t1 = o.compute_t_and_t1_product_check(Omega, f)
    w = M([Omega[1]])[0]
    t_w = backend.compose_with_scaling(M(t), w)
    f_w = backend.compose_with_scaling(M(f), w)
    prod = U(backend.fr_poly_mul(M(t), f_w))
    got = o.poly_sub(U(t_w), prod)
    assert o.poly_trim(got) == t1
    # and t itself from device evaluations + running product + interpolation
    evals = backend.fr_eval_many(M(f), M(Omega))
    t_vals = backend.fr_prefix_product(evals)
    assert o.poly_trim(U(backend.interpolate_polynomial(M(Omega), t_vals))) == t


@pytest.mark.parametrize("m", [0, 1, 2, 3, 255, 256, 257, 1000, 5000])
def test_vanishing_from_roots(backend, m):
    rnd = random.Random(780 + m)
    roots = rand_fr(rnd, m)
    z = U(backend.construct_vanishing_polynomial_from_roots(M(roots)) if m else
          backend.construct_vanishing_polynomial_from_roots(np.zeros((0, 4), dtype=np.uint64)))
    assert len(z) == m + 1 and z[-1] == 1
    if m <= 300:
        assert z == o.construct_vanishing_polynomial_from_roots(roots)  # utils.rs:51-62
    else:
        for x in roots[:5] + roots[-5:]:
            assert o.poly_eval(z, x) == 0
        x = rnd.randrange(o.R)
        want = 1
        for rho in roots:
            want = want * (x - rho) % o.R
        assert o.poly_eval(z, x) == want


def test_vanishing_from_roots_of_unity_is_x_k_minus_1(backend):
    # plonk/tests/polynomials_tests.rs:120-150: prod (X - g^i) = X^1024 - 1
    Omega = o.construct_Omega(1024)
    z = U(backend.construct_vanishing_polynomial_from_roots(M(Omega)))
    assert z == [o.R - 1] + [0] * 1023 + [1]


@pytest.mark.parametrize("n,m", [(12, 2), (23, 3), (100, 24), (100, 25), (3000, 700), (3000, 2999), (5, 7), (8192, 4096)])
def test_div_by_roots_is_long_division(backend, n, m):
    rnd = random.Random(790 + n + m)
    roots, f = rand_fr(rnd, m), rand_fr(rnd, n)  # not divisible: remainder dropped (protocols.rs:177-180)
    got = U(backend.compute_q_zero_test_from_roots(M(roots), M(f)))
    if n <= m:
        assert got == []
        return
    zpoly = [1]
    for rho in roots:
        zpoly = U(c.poly_mul(M(zpoly), M([(-rho) % o.R, 1])))
    q, _ = c.poly_divmod(M(f), M(zpoly))
    assert got == U(q)
    assert len(got) == n - m


def test_div_by_roots_example_circuit_shapes(backend):
    # roots = Omega_inputs (2 roots) and Omega_gates (3 roots) of the example circuit
    rnd = random.Random(800)
    Omega = o.construct_Omega(12)
    for roots in ([Omega[11], Omega[10]], [Omega[0], Omega[3], Omega[6]]):
        qs = rand_fr(rnd, 20)
        f = o.poly_mul(qs, o.construct_vanishing_polynomial_from_roots(roots))
        got = U(backend.compute_q_zero_test_from_roots(M(roots), M(f)))
        assert o.poly_trim(got) == o.compute_q_zero_test_from_roots(roots, f) == o.poly_trim(qs)


@pytest.mark.parametrize("n", [1, 2, 3, 12, 64, 65, 300, 1500])
def test_interpolate_at_arbitrary_points(backend, n):
    rnd = random.Random(810 + n)
    xs = rand_fr(rnd, n)  # distinct with overwhelming probability
    assert len(set(xs)) == n
    ys = rand_fr(rnd, n)
    got = U(backend.interpolate_polynomial(M(xs), M(ys)))
    if n <= 300:
        assert got == U(c.interpolate_lagrange(M(xs), M(ys)))  # polynomials.rs:14-58 restated
    for i in (rnd.sample(range(n), min(n, 8))):
        assert o.poly_eval(got, xs[i]) == ys[i]


def test_interpolate_reference_known_answer(backend):
    # plonk/tests/polynomials_tests.rs:160-177: (0,0), (1,1), (2,8) -> -2x + 3x^2
    xs, ys = [0, 1, 2], [0, 1, 8]
    got = o.poly_trim(U(backend.interpolate_polynomial(M(xs), M(ys))))
    assert got == [0, o.R - 2, 3] == o.interpolate_polynomial(xs, ys)
    # coincident points: polynomials.rs:42 `.inverse().unwrap()` panics
    with pytest.raises(plonk_amd.ReferencePanic):
        backend.interpolate_polynomial(M([1, 2, 1]), M([2, 3, 5]))


def _rand_mont(rng, n):
    a = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)  # below 2^254 < r: a valid Montgomery representative
    return a


def _product_of(vals):
    """Product of all rows (Montgomery form) by a pairwise tree on the oracle's vector multiply."""
    one = c.fr_to_mont([1])
    while len(vals) > 1:
        if len(vals) & 1:
            vals = np.concatenate([vals, one])
        vals = c.fr_mul(vals[0::2].copy(), vals[1::2].copy())
    return vals[0]


def test_quotient_by_2_20_gate_roots(backend):
    """`compute_q_zero_test_from_roots` (protocols.rs:169-181) at the size of a 2^20-gate circuit:
    2^20 roots, dividend of 2^22 coefficients.  Size-independent checks: Z(x) = prod (x - rho) at a
    random x, exact recovery of a known quotient, remainder dropped."""
    rng = np.random.default_rng(2020)
    m, nq = 1 << 20, 3 << 20
    roots = _rand_mont(rng, m)
    z = backend.construct_vanishing_polynomial_from_roots(roots)
    assert z.shape == (m + 1, 4) and np.array_equal(z[m], c.fr_to_mont([1])[0])
    x = _rand_mont(rng, 1)
    want = _product_of(c.fr_sub(np.repeat(x, m, axis=0), roots))
    assert np.array_equal(c.fr_eval(z, x[0]), want)
    q_star = _rand_mont(rng, nq)
    f = backend.fr_poly_mul(q_star, z)
    assert f.shape[0] == nq + m
    assert np.array_equal(c.fr_eval(f, x[0]), c.fr_mul(c.fr_eval(q_star, x[0]).reshape(1, 4), want.reshape(1, 4))[0])
    got = backend.compute_q_zero_test_from_roots(roots, f)
    assert np.array_equal(got, q_star)
    f[:m] = c.fr_add(f[:m].copy(), _rand_mont(rng, m))  # a remainder of degree < m
    assert np.array_equal(backend.compute_q_zero_test_from_roots(roots, f), q_star)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 33, 100, 1000])
def test_interpolation_on_a_geometric_progression(backend, n):
    """`interpolate_polynomial([1, q, q^2, ...], ys)` (polynomials.rs:50-58) in O(n log n): the same coefficients as
    the restatement's Lagrange form, for q a root of unity of large order (the gate rows, q = omega^3) and for a
    random q."""
    rnd = random.Random(700 + n)
    for q in (pow(7, (o.R - 1) // (3 << 20), o.R), rnd.randrange(2, o.R)):
        ys = rand_fr(rnd, n)
        xs = [pow(q, i, o.R) for i in range(n)]
        got = backend.interpolate_geometric(c.fr_to_mont([q])[0], c.fr_to_mont(ys))
        want = o.interpolate_polynomial(xs, ys)
        assert c.fr_from_mont(got) == want + [0] * (n - len(want))
        # the general entry point agrees
        assert np.array_equal(got, backend.interpolate_polynomial(c.fr_to_mont(xs), c.fr_to_mont(ys)))


def test_interpolation_on_a_geometric_progression_large_and_degenerate(backend):
    rnd = random.Random(800)
    # beyond the 2^16 points of the quadratic kernel: checked at sampled points by the oracle's Horner evaluation
    n = (1 << 17) + 12345
    q = pow(7, (o.R - 1) // (3 << 18), o.R)
    ys = c.fr_to_mont(rand_fr(rnd, n))
    coeffs = backend.interpolate_geometric(c.fr_to_mont([q])[0], ys)
    for i in [0, 1, 2, n // 2, n - 2, n - 1] + [rnd.randrange(n) for _ in range(10)]:
        assert np.array_equal(c.fr_eval(coeffs, c.fr_to_mont([pow(q, i, o.R)])[0]), ys[i]), i
    # points that coincide: q of order 8 and nine points; q = 1; q = 0 with three points
    with pytest.raises(plonk_amd.ReferencePanic):
        backend.interpolate_geometric(c.fr_to_mont([pow(7, (o.R - 1) // 8, o.R)])[0], c.fr_to_mont(rand_fr(rnd, 9)))
    with pytest.raises(plonk_amd.ReferencePanic):
        backend.interpolate_geometric(c.fr_to_mont([1])[0], c.fr_to_mont(rand_fr(rnd, 2)))
    with pytest.raises(plonk_amd.ReferencePanic):
        backend.interpolate_geometric(c.fr_to_mont([0])[0], c.fr_to_mont(rand_fr(rnd, 3)))
    # eight points on the subgroup of order 8 are fine: the interpolant is the inverse transform
    w8 = pow(7, (o.R - 1) // 8, o.R)
    ys8 = c.fr_to_mont(rand_fr(rnd, 8))
    assert np.array_equal(backend.interpolate_geometric(c.fr_to_mont([w8])[0], ys8), backend.fr_ntt(ys8, inverse=True))
    # two points 1, 0
    two = rand_fr(rnd, 2)
    got = c.fr_from_mont(backend.interpolate_geometric(c.fr_to_mont([0])[0], c.fr_to_mont(two)))
    assert got == [two[1], (two[0] - two[1]) % o.R]
