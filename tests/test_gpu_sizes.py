"""Every transform size from one point to 2^22, coset transforms at 2^10 / 2^16 / 2^20 and the mixed-radix
transform up to 3 * 2^18, against the oracle: the whole output against the oracle's radix-2 transform
(oracle/oracle.c, itself pinned on the O(n^2) DFT of oracle/pyref.py) where the size is a power of two, and
value j against f(shift * omega^j) by the oracle's Horner evaluation at sampled j (the definition the reference's
`f.evaluate(&Omega[j])` loops use, plonk/src/common/protocols.rs:243-250; omega of plonk/src/common/utils.rs:11-38)
for cosets and 3 * 2^k.  The inverse must give the input back exactly (polynomials.rs:50-58: interpolation over
Omega is the inverse of evaluation).  Bit-exact."""
import numpy as np
import pytest

from oracle import capi as c
from oracle import pyref as o
from tests.gpu_util import backend  # noqa: F401

pytestmark = pytest.mark.gpu


def rand_limbs(rng, n):
    a = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    return a


def root(n):
    assert (o.R - 1) % n == 0
    return pow(7, (o.R - 1) // n, o.R)


@pytest.mark.parametrize("log_n", list(range(0, 23)))
def test_every_power_of_two_size(backend, log_n):
    rng = np.random.default_rng(100 + log_n)
    n = 1 << log_n
    f = rand_limbs(rng, n)
    vals = backend.fr_ntt(f)
    assert np.array_equal(vals, c.fr_ntt(f))
    assert np.array_equal(backend.fr_ntt(vals, inverse=True), f)
    if log_n <= 14:  # a batch of three, the middle one all zero
        batch = np.stack([f, np.zeros_like(f), rand_limbs(rng, n)])
        got = backend.fr_ntt(batch)
        assert np.array_equal(got[0], vals) and not got[1].any() and np.array_equal(got[2], c.fr_ntt(batch[2]))


@pytest.mark.parametrize("log_n", [10, 16, 20])
def test_coset_transforms(backend, log_n):
    rng = np.random.default_rng(200 + log_n)
    n = 1 << log_n
    w = root(n)
    f = rand_limbs(rng, n)
    for shift in (7, int(rng.integers(2, 1 << 62)) * 0x9E3779B97F4A7C15 % o.R):
        sh = c.fr_to_mont([shift])[0]
        vals = backend.fr_ntt(f, shift=sh)
        for j in [0, 1, n // 2, n - 1] + [int(x) for x in rng.integers(0, n, 12)]:
            x = c.fr_to_mont([shift * pow(w, j, o.R) % o.R])[0]
            assert np.array_equal(vals[j], c.fr_eval(f, x)), (shift, j)
        assert np.array_equal(backend.fr_ntt(vals, inverse=True, shift=sh), f)


@pytest.mark.parametrize("k", list(range(0, 19)))
def test_three_times_a_power_of_two(backend, k):
    rng = np.random.default_rng(300 + k)
    n = 3 << k
    w = root(n)
    f = rand_limbs(rng, n)
    vals = backend.fr_ntt_mixed(f)
    for j in [0, 1, 2, n // 3, n - 1] + [int(x) for x in rng.integers(0, n, 8)]:
        x = c.fr_to_mont([pow(w, j, o.R)])[0]
        assert np.array_equal(vals[j], c.fr_eval(f, x)), j
    assert np.array_equal(backend.fr_ntt_mixed(vals, inverse=True), f)
