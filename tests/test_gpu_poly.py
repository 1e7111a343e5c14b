"""GPU parity: Fr transforms over Omega, coset transforms, quotient by Z_Omega, evaluation and
coefficient scaling, through the C ABI, against the oracle.  Bit-exact."""
import random

import numpy as np
import pytest

import plonk_amd

from oracle import capi as c
from oracle import pyref as o
from tests.gpu_util import backend, rand_fr  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 5, 9, 10, 11, 12, 13, 16])
def test_ntt_matches_oracle(backend, log_n):
    rnd = random.Random(50 + log_n)
    n = 1 << log_n
    batch = 3
    co = c.fr_to_mont(rand_fr(rnd, n * batch)).reshape(batch, n, 4)
    ev = backend.fr_ntt(co)
    for b in range(batch):
        assert np.array_equal(ev[b], c.fr_ntt(co[b]))
    assert np.array_equal(backend.fr_ntt(ev, inverse=True), co)
    if 1 <= log_n <= 5:
        # the definition: evaluations over construct_Omega(n) / interpolate_polynomial
        ints = c.fr_from_mont(co[0])
        Om = o.construct_Omega(n)
        assert c.fr_from_mont(ev[0]) == [o.poly_eval(ints, x) for x in Om]
        assert o.poly_trim(c.fr_from_mont(backend.interpolate_over_omega(ev[0]))) == o.interpolate_polynomial(
            Om, c.fr_from_mont(ev[0]))


@pytest.mark.parametrize("log_n", [3, 10, 11, 13, 20])
def test_ntt_worst_case_growth_of_the_lazy_sums(backend, log_n):
    # a constant vector of r - 1 makes every butterfly sum as large as it can get
    n = 1 << log_n
    co = c.fr_to_mont([o.R - 1]).reshape(1, 4).repeat(n, axis=0).reshape(1, n, 4).copy()
    ev = backend.fr_ntt(co)
    want = np.zeros((n, 4), dtype=np.uint64)
    want[0] = c.fr_to_mont([(n * (o.R - 1)) % o.R])[0]
    assert np.array_equal(ev[0], want)
    assert np.array_equal(backend.fr_ntt(ev, inverse=True), co)
    assert np.array_equal(backend.fr_ntt(co, inverse=True)[0, 0], c.fr_to_mont([o.R - 1])[0])
    # alternating r - 1, 0: the largest differences
    alt = co.copy()
    alt[0, 1::2] = 0
    ev = backend.fr_ntt(alt)
    if log_n <= 13:
        assert np.array_equal(ev[0], c.fr_ntt(alt[0]))
    assert np.array_equal(backend.fr_ntt(ev, inverse=True), alt)


def test_coset_ntt(backend):
    rnd = random.Random(60)
    n = 256
    co = rand_fr(rnd, n)
    shift = rnd.randrange(2, o.R)
    ev = backend.fr_ntt(c.fr_to_mont(co), shift=c.fr_to_mont([shift])[0])
    w = o.root_of_unity(n)
    assert c.fr_from_mont(ev) == [o.poly_eval(co, shift * pow(w, j, o.R) % o.R) for j in range(n)]
    back = backend.fr_ntt(ev, inverse=True, shift=c.fr_to_mont([shift])[0])
    assert c.fr_from_mont(back) == co


@pytest.mark.parametrize("log_n", [21, 22])
def test_ntt_beyond_the_two_pass_natural_order_path(backend, log_n):
    # above 2^20 the butterflies run on the plain layout and a separate permutation follows
    rnd = random.Random(70 + log_n)
    n = 1 << log_n
    data = np.frombuffer(rnd.randbytes(n * 32), dtype=np.uint64).reshape(1, n, 4).copy()
    data[:, :, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    ev = backend.fr_ntt(data)
    assert np.array_equal(ev[0], c.fr_ntt(data[0]))
    assert np.array_equal(backend.fr_ntt(ev, inverse=True), data)


def test_ntt_2_20_round_trip_and_oracle(backend):
    rnd = random.Random(61)
    n = 1 << 20
    batch = 2
    data = np.frombuffer(rnd.randbytes(batch * n * 32), dtype=np.uint64).reshape(batch, n, 4).copy()
    data[:, :, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    ev = backend.fr_ntt(data)
    assert np.array_equal(ev[0], c.fr_ntt(data[0]))
    assert np.array_equal(backend.fr_ntt(ev, inverse=True), data)


def test_eval_and_scaling(backend):
    rnd = random.Random(62)
    for n in (1, 2, 100, 257, 70000):
        f = rand_fr(rnd, n)
        u = rnd.randrange(o.R)
        got = backend.fr_eval(c.fr_to_mont(f), c.fr_to_mont([u])[0])
        assert c.fr_from_mont(got)[0] == o.poly_eval(f, u)
    f = rand_fr(rnd, 3000)
    w = rnd.randrange(o.R)
    got = c.fr_from_mont(backend.compose_with_scaling(c.fr_to_mont(f), c.fr_to_mont([w])[0]))
    assert got == [x * pow(w, i, o.R) % o.R for i, x in enumerate(f)]
    small = rand_fr(rnd, 12)
    assert o.poly_trim(c.fr_from_mont(backend.compose_with_scaling(c.fr_to_mont(small), c.fr_to_mont([w])[0]))) == \
        o.compose_polynomials(small, [0, w])  # polynomials.rs:80-113


def test_div_vanishing_is_long_division(backend):
    rnd = random.Random(63)
    for n, k in ((25, 12), (13, 12), (100, 7), (2048, 1024), (5000, 1)):
        f = rand_fr(rnd, n)  # not divisible: remainder is dropped, protocols.rs:162-165
        q = c.fr_from_mont(backend.compute_q_zero_test(k, c.fr_to_mont(f)))
        assert o.poly_trim(q) == o.compute_q_zero_test(k, f)
    assert backend.compute_q_zero_test(12, c.fr_to_mont(rand_fr(rnd, 5))).shape[0] == 0
    # divisible case: exact
    qs = rand_fr(rnd, 40)
    f = o.poly_mul(qs, o.construct_vanishing_polynomial(16))
    assert o.poly_trim(c.fr_from_mont(backend.compute_q_zero_test(16, c.fr_to_mont(f)))) == o.poly_trim(qs)


@pytest.mark.parametrize("log_k", [1, 4, 10, 12])
def test_quotient_coset(backend, log_k):
    rnd = random.Random(64 + log_k)
    k = 1 << log_k
    batch = 3
    qs = [rand_fr(rnd, k) for _ in range(batch)]
    fs = [[(-x) % o.R for x in q] + q for q in qs]  # q * (X^k - 1)
    fs[2][0] = (fs[2][0] + 1) % o.R  # not divisible
    f = c.fr_to_mont([x for row in fs for x in row]).reshape(batch, 2 * k, 4)
    q, ok = backend.quotient_coset(f)
    assert ok == [True, True, False]
    for b in range(2):
        assert c.fr_from_mont(q[b]) == qs[b]
        assert o.poly_trim(qs[b]) == o.compute_q_zero_test(k, fs[b])


@pytest.mark.parametrize("n", [3, 6, 12, 24, 96, 3 << 10, 1 << 6])
def test_ntt_over_domains_of_size_3_times_a_power_of_two(backend, n):
    # the reference's domains hold three points per gate (setup_global_params/mod.rs:57); the
    # example circuit's Omega has 12 elements (utils.rs:11-38 with k = 12)
    rnd = random.Random(80 + n)
    batch = 2
    co = [rand_fr(rnd, n) for _ in range(batch)]
    data = c.fr_to_mont([x for row in co for x in row]).reshape(batch, n, 4)
    ev = backend.fr_ntt_mixed(data)
    Om = o.construct_Omega(n)
    idx = range(n) if n <= 24 else rnd.sample(range(n), 12)
    for b in range(batch):
        got = c.fr_from_mont(ev[b])
        for j in idx:
            assert got[j] == o.poly_eval(co[b], Om[j])
    assert np.array_equal(backend.fr_ntt_mixed(ev, inverse=True), data)
    if n == 12:
        ys = rand_fr(rnd, n)
        coeffs = c.fr_from_mont(backend.fr_ntt_mixed(c.fr_to_mont(ys), inverse=True))
        assert o.poly_trim(coeffs) == o.interpolate_polynomial(Om, ys)  # polynomials.rs:50-58


def test_ntt_mixed_rejects_other_sizes(backend):
    with pytest.raises(plonk_amd.PlonkHipError):
        backend.fr_ntt_mixed(np.zeros((10, 4), dtype=np.uint64))
