"""Inputs of the reference's gadget tests (plonk/tests/protocols_tests.rs:24-644), regenerated draw by draw with
the restated `ark_std::test_rng()` (oracle/arkrng.py): the SRS of `kzg_setup(degree)` (its own tau), Omega of
size 8, the polynomials of degree 10 and the verifier's point r -- as Python integers.  Used by the CPU tests of
the oracle and by the GPU parity tests, which feed the same inputs to both sides."""
from oracle import arkrng as a
from oracle import pyref as o

K = 8


def _inv(x):
    return o.fr_inv(x)


def equality(success):
    """:24-80."""
    rng = a.test_rng()
    f = a.random_polynomial(rng, 10)
    g = list(f) if success else a.random_polynomial(rng, 10)
    return dict(degree=10, f=f, g=g, r=a.fr_rand(rng))


def zero_test(success):
    """:82-167."""
    rng = a.test_rng()
    Omega = o.construct_Omega(K)
    if success:
        xs = Omega + [a.fr_rand(rng) for _ in range(3)]
        ys = [0] * 8 + [a.fr_rand(rng) for _ in range(3)]
        f = o.interpolate_polynomial(xs, ys)
    else:
        f = a.random_polynomial(rng, 10)
    return dict(degree=10, Omega=Omega, f=f, r=a.fr_rand(rng))


def zero_on_roots(success):
    """:169-251."""
    rng = a.test_rng()
    roots = [2, 3]
    if success:
        xs = roots + [a.fr_rand(rng) for _ in range(9)]
        ys = [0, 0] + [a.fr_rand(rng) for _ in range(9)]
        f = o.interpolate_polynomial(xs, ys)
    else:
        f = a.random_polynomial(rng, 10)
    return dict(degree=10, roots=roots, f=f, r=a.fr_rand(rng))


def product_check(success):
    """:253-376."""
    rng = a.test_rng()
    Omega = o.construct_Omega(K)
    if success:
        xs = Omega + [a.fr_rand(rng) for _ in range(3)]
        ys = [a.fr_rand(rng) for _ in range(7)]
        prod = 1
        for y in ys:
            prod = prod * y % o.R
        ys.append(_inv(prod))
        ys += [a.fr_rand(rng) for _ in range(3)]
        f = o.interpolate_polynomial(xs, ys)
    else:
        f = a.random_polynomial(rng, 10)
    return dict(degree=12, Omega=Omega, f=f, r=a.fr_rand(rng))


def sum_check(success):
    """:378-501."""
    rng = a.test_rng()
    Omega = o.construct_Omega(K)
    if success:
        xs = Omega + [a.fr_rand(rng) for _ in range(3)]
        ys = [a.fr_rand(rng) for _ in range(7)]
        ys.append(-sum(ys) % o.R)
        ys += [a.fr_rand(rng) for _ in range(3)]
        f = o.interpolate_polynomial(xs, ys)
    else:
        f = a.random_polynomial(rng, 10)
    return dict(degree=10, Omega=Omega, f=f, r=a.fr_rand(rng))


def rational_check(success):
    """:503-644.  (In the success case the reference interpolates g from f's values, :539-540: g = f.)"""
    rng = a.test_rng()
    Omega = o.construct_Omega(K)
    if success:
        xs = Omega + [a.fr_rand(rng) for _ in range(3)]
        f_ys = [a.fr_rand(rng) for _ in range(8)]
        g_ys = [a.fr_rand(rng) for _ in range(7)]  # drawn, then unused: g is interpolated from f_ys
        f_ys += [a.fr_rand(rng) for _ in range(3)]
        g_ys += [a.fr_rand(rng) for _ in range(3)]
        f = o.interpolate_polynomial(xs, f_ys)
        g = o.interpolate_polynomial(xs, f_ys)
    else:
        f = a.random_polynomial(rng, 10)
        g = a.random_polynomial(rng, 10)
    return dict(degree=12, Omega=Omega, f=f, g=g, r=a.fr_rand(rng))
