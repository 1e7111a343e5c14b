"""Inputs of the reference's gadget tests (plonk/tests/protocols_tests.rs:24-644), regenerated draw by draw with
the restated `ark_std::test_rng()` (oracle/arkrng.py): the SRS of `kzg_setup(degree)` (its own tau), Omega of
size 8, the polynomials of degree 10 and the verifier's point r -- as Python integers.  Used by the CPU tests of
the oracle and by the GPU parity tests, which feed the same inputs to both sides.
The permutation cases (:646-943) shuffle with rand's `SliceRandom::shuffle`, restated but unpinned
(oracle/arkrng.py: shuffle): whichever permutation it yields, the cases keep the property the reference tests."""
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


def permutation_check(success):
    """:646-776: g's values on Omega are a permutation of f's; proved by the product check of rational functions."""
    rng = a.test_rng()
    Omega = o.construct_Omega(K)
    if success:
        xs = Omega + [a.fr_rand(rng) for _ in range(3)]
        f_ys = [a.fr_rand(rng) for _ in range(8)]
        g_ys = list(f_ys)
        a.shuffle(rng, g_ys)
        f_ys += [a.fr_rand(rng) for _ in range(3)]
        g_ys += [a.fr_rand(rng) for _ in range(3)]
        f = o.interpolate_polynomial(xs, f_ys)
        g = o.interpolate_polynomial(xs, g_ys)
    else:
        f = a.random_polynomial(rng, 10)
        g = a.random_polynomial(rng, 10)
    return dict(degree=12, Omega=Omega, f=f, g=g, r=a.fr_rand(rng))


def prescribed_permutation_check(success):
    """:778-943: f(x) = g(W(x)) on Omega for a permutation W of Omega."""
    rng = a.test_rng()
    Omega = o.construct_Omega(K)
    if success:
        xs = list(Omega)
        W_ys = list(xs)
        a.shuffle(rng, W_ys)
        xs += [a.fr_rand(rng) for _ in range(3)]
        f_ys = [a.fr_rand(rng) for _ in range(8)]
        g_ys = list(f_ys)
        W_ys += [a.fr_rand(rng) for _ in range(3)]
        f_ys += [a.fr_rand(rng) for _ in range(3)]
        g_ys += [a.fr_rand(rng) for _ in range(3)]
        W = o.interpolate_polynomial(xs, W_ys)
        f = o.interpolate_polynomial(xs, f_ys)
        g = o.interpolate_polynomial(W_ys, g_ys)
    else:
        f = a.random_polynomial(rng, 10)
        g = a.random_polynomial(rng, 10)
        W = a.random_polynomial(rng, 10)
    r, s = a.fr_rand(rng), a.fr_rand(rng)
    return dict(degree=12, Omega=Omega, f=f, g=g, W=W, r=r, s=s, rp=a.fr_rand(rng))


def t_s_zero_test(success):
    """:945-1149: the example circuit's trace T and selector S (d = 12, three gates), or random T and S."""
    rng = a.test_rng()
    d, number_gates = 12, 3
    Omega = o.construct_Omega(d)
    Omega_gates = [Omega[3 * l] for l in range(number_gates)]
    if success:
        xs = [Omega[d - 1], Omega[d - 2], Omega[d - 3]] + Omega[0:9]
        ys = [5, 6, 1, 5, 6, 11, 6, 1, 7, 11, 7, 77]
        T = o.interpolate_polynomial(xs, ys)
        S = o.interpolate_polynomial(Omega_gates, [1, 1, 0])
    else:
        T = a.random_polynomial(rng, 11)
        S = a.random_polynomial(rng, 2)
    return dict(degree=21, Omega=Omega, Omega_gates=Omega_gates, T=T, S=S, r=a.fr_rand(rng))
