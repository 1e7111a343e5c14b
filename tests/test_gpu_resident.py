"""The device-resident entry points (`*_dev`) against their host-buffer forms, which the other GPU tests
pin on the oracle: same inputs, bit-identical outputs, same error codes."""
import numpy as np
import pytest

import plonk_amd
from tests.gpu_util import backend  # noqa: F401

pytestmark = pytest.mark.gpu


def rand(rng, n):
    a = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    return a


class Dev:
    def __init__(self, be):
        self.be, self.ptrs = be, []

    def put(self, a):
        p = self.be.to_device(np.ascontiguousarray(a)) if len(a) else self.be.malloc(32)
        self.ptrs.append(p)
        return p

    def new(self, n):
        p = self.be.malloc(max(n, 1) * 32)
        self.ptrs.append(p)
        return p

    def get(self, p, n):
        self.be.sync()
        return self.be.from_device(p, (n, 4))

    def close(self):
        self.be.sync()
        for p in self.ptrs:
            self.be.free(p)


@pytest.fixture()
def dev(backend):
    d = Dev(backend)
    yield d
    d.close()


@pytest.mark.parametrize("na,nb", [(1, 1), (5, 3), (300, 1000), (4096, 4097), (70000, 3)])
def test_poly_mul_dev(backend, dev, na, nb):
    rng = np.random.default_rng(na + nb)
    a, b = rand(rng, na), rand(rng, nb)
    out = dev.new(na + nb - 1)
    backend.fr_poly_mul_dev(dev.put(a), na, dev.put(b), nb, out)
    assert np.array_equal(dev.get(out, na + nb - 1), backend.fr_poly_mul(a, b))


@pytest.mark.parametrize("nx,ny,broadcast", [(10, 10, False), (10, 3, False), (3, 10, True), (5000, 4999, False), (7, 0, False)])
def test_linear_combination_dev_and_in_place(backend, dev, nx, ny, broadcast):
    rng = np.random.default_rng(nx * 31 + ny)
    x, y = rand(rng, nx), rand(rng, ny)
    a, b, c = rand(rng, 3)
    want = backend.fr_linear_combination(x, y, a, b, c, broadcast)
    n = max(nx, ny)
    dx, dy = dev.put(x), dev.put(y)
    out = dev.new(n)
    backend.fr_linear_combination_dev(dx, nx, dy if ny else None, ny, a, b, c, broadcast, out)
    assert np.array_equal(dev.get(out, n), want)
    if nx >= ny:  # result written over the longer operand
        backend.fr_linear_combination_dev(dx, nx, dy if ny else None, ny, a, b, c, broadcast, dx)
        assert np.array_equal(dev.get(dx, n), want)


@pytest.mark.parametrize("n,m", [(12, 2), (100, 24), (100, 25), (3000, 700), (5, 7)])
def test_div_by_roots_dev(backend, dev, n, m):
    rng = np.random.default_rng(n * 7 + m)
    f, roots = rand(rng, n), rand(rng, m)
    want = backend.compute_q_zero_test_from_roots(roots, f)
    q = dev.new(max(n - m, 1))
    backend.fr_div_by_roots_dev(dev.put(f), n, dev.put(roots), m, q)
    if n > m:
        assert np.array_equal(dev.get(q, n - m), want)


@pytest.mark.parametrize("n,k", [(24, 12), (25, 12), (1000, 7), (100, 100), (100000, 4096)])
def test_div_vanishing_dev(backend, dev, n, k):
    rng = np.random.default_rng(n + k)
    f = rand(rng, n)
    want = backend.compute_q_zero_test(k, f)
    q = dev.new(max(n - k, 1))
    backend.fr_div_vanishing_dev(dev.put(f), n, k, q)
    if n > k:
        assert np.array_equal(dev.get(q, n - k), want)


def test_scale_by_powers_and_ratio_prefix_product_dev(backend, dev):
    rng = np.random.default_rng(5)
    n = 5000
    f, num, den, w = rand(rng, n), rand(rng, n), rand(rng, n), rand(rng, 1)[0]
    p = dev.put(f)
    backend.fr_scale_by_powers_dev(p, n, w)
    assert np.array_equal(dev.get(p, n), backend.compose_with_scaling(f, w))
    dn, dd = dev.put(num), dev.put(den)
    backend.fr_ratio_prefix_product_dev(dn, dd, n)
    assert np.array_equal(dev.get(dn, n), backend.fr_ratio_prefix_product(num, den))
    den[77] = 0
    with pytest.raises(plonk_amd.ReferencePanic):  # inverse of zero: protocols.rs:583
        backend.fr_ratio_prefix_product_dev(dev.put(num), dev.put(den), n)


def test_commit_and_open_many_dev(backend, dev):
    rng = np.random.default_rng(6)
    n = 3000
    backend.kzg_setup(rand(rng, 1)[0], n - 1, want_points=False)
    f, g, us = rand(rng, n), rand(rng, 17), rand(rng, 4)
    coms = backend.kzg_commit_many_dev([(dev.put(f), n), (dev.put(g), 17)])
    assert np.array_equal(coms, backend.kzg_commit_batch([f, g]))
    vs, proofs = backend.kzg_open_many_dev(dev.put(f), n, us)
    want_vs, want_proofs = backend.kzg_evaluate_many(f, us)
    assert np.array_equal(vs, want_vs) and np.array_equal(proofs, want_proofs)
    with pytest.raises(plonk_amd.CommitError):  # longer than the SRS: kzg.rs:49-50
        backend.kzg_commit_many_dev([(dev.put(rand(rng, n + 1)), n + 1)])
    with pytest.raises(plonk_amd.ReferencePanic):  # constant polynomial: kzg.rs:85
        backend.kzg_open_many_dev(dev.put(g), 1, us[:1])


def test_zero_and_permutation_checks_dev(backend, dev):
    rng = np.random.default_rng(7)
    n = 10000
    x = rand(rng, n)
    z = dev.new(n)
    backend.memset_zero(z, n * 32)
    assert backend.fr_is_zero_dev(z, n) and not backend.fr_is_zero_dev(dev.put(x), n)
    assert backend.fr_is_zero_dev(z, 0)
    perm = rng.permutation(n)
    # values constant on the cycles of perm: copy each cycle leader's value around its cycle
    vals = x.copy()
    seen = np.zeros(n, dtype=bool)
    for i in range(n):
        j = i
        while not seen[j]:
            seen[j] = True
            vals[j] = x[i]
            j = perm[j]
    p = dev.put(vals)
    assert backend.fr_respects_permutation_dev(p, perm)
    vals[perm[3]] = x[(3 + 1) % n] if not np.array_equal(x[(3 + 1) % n], vals[perm[3]]) else x[5]
    assert not backend.fr_respects_permutation_dev(dev.put(vals), perm) or perm[3] == 3
    copy = dev.new(n)
    backend.memcpy_d2d(copy, p, n * 32)
    assert np.array_equal(dev.get(copy, n), dev.get(p, n))


def test_commit_open_task_list_dev(backend, dev):
    """A mixed list of commitments and openings in one call (tails overlapped with the next accumulation)
    against the one-at-a-time entry points."""
    rng = np.random.default_rng(8)
    n = 5000
    backend.kzg_setup(rand(rng, 1)[0], n - 1, want_points=False)
    polys = [rand(rng, m) for m in (n, 17, n - 1, 2, 300, n)]
    points = [None, rand(rng, 1)[0], rand(rng, 1)[0], None, rand(rng, 1)[0], rand(rng, 1)[0]]
    ptrs = [dev.put(p) for p in polys]
    vals, outs = backend.kzg_commit_open_many_dev([(ptr, p.shape[0], u) for ptr, p, u in zip(ptrs, polys, points)])
    for k, (p, u) in enumerate(zip(polys, points)):
        if u is None:
            assert np.array_equal(outs[k], backend.kzg_commit(p)) and not vals[k].any()
        else:
            v, proof = backend.kzg_evaluate(p, u)
            assert np.array_equal(vals[k], v) and np.array_equal(outs[k], proof)
    # the same list again: slots and events are reused
    vals2, outs2 = backend.kzg_commit_open_many_dev([(ptr, p.shape[0], u) for ptr, p, u in zip(ptrs, polys, points)])
    assert np.array_equal(vals, vals2) and np.array_equal(outs, outs2)
    with pytest.raises(plonk_amd.ReferencePanic):  # opening a constant: kzg.rs:85
        backend.kzg_commit_open_many_dev([(ptrs[0], n, None), (ptrs[1], 1, points[1])])
