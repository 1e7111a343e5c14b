"""The entry point that allocates nothing (plonk_hip_fr_ratio_prefix_product_dev_noalloc) against the
entry point it is a variant of, through the C ABI: bit-identical outputs on two calls in a row, the
second with a smaller input, so that whatever the first call left in the shared static scratch array
is seen not to leak into the second."""
import numpy as np
import pytest

import plonk_amd
from tests.gpu_util import backend  # noqa: F401

pytestmark = pytest.mark.gpu


def rand(rng, n):
    a = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    return a


def run_both(be, num, den):
    """(result of the existing entry point or None when it reports a zero denominator,
    result of the no-allocation entry point, its flag)"""
    n = num.shape[0]
    ptrs = [be.to_device(num), be.to_device(den), be.to_device(num), be.to_device(den), be.malloc(256)]
    try:
        try:
            be.fr_ratio_prefix_product_dev(ptrs[0], ptrs[1], n)
            be.sync()
            want = be.from_device(ptrs[0], (n, 4))
        except plonk_amd.ReferencePanic:
            want = None
        be.to_device(np.full(1, 7, dtype=np.int32), ptrs[4])  # the call itself must clear the flag
        be.fr_ratio_prefix_product_dev_noalloc(ptrs[2], ptrs[3], n, ptrs[4])
        be.sync()
        got = be.from_device(ptrs[2], (n, 4))
        flag = int(be.from_device(ptrs[4], (1,), dtype=np.int32)[0])
    finally:
        be.sync()
        for p in ptrs:
            be.free(p)
    return want, got, flag


def test_two_calls_in_a_row_second_one_smaller(backend):
    rng = np.random.default_rng(2024)
    # several levels of block totals, then a single block
    for n in ((1 << 22) + 12345, 70001, 1500):
        num, den = rand(rng, n), rand(rng, n)
        den[den.sum(axis=1) == 0] = 1
        want, got, flag = run_both(backend, num, den)
        assert flag == 0
        assert want is not None and np.array_equal(got, want), n


def test_zero_denominator_goes_to_the_flag(backend):
    rng = np.random.default_rng(7)
    n = 5000
    num, den = rand(rng, n), rand(rng, n)
    den[4321] = 0
    want, _, flag = run_both(backend, num, den)
    assert want is None and flag == 1  # the existing entry point panics like protocols.rs:583
    # and the next call on the same GPU starts from a clean flag and clean scratch
    den[4321] = 1
    want, got, flag = run_both(backend, num, den)
    assert flag == 0 and np.array_equal(got, want)


def test_size_limit_is_an_error_not_an_allocation(backend):
    flag = backend.malloc(256)
    try:
        with pytest.raises(plonk_amd.PlonkHipError) as e:
            # pointers are not touched: the size check comes first
            backend.fr_ratio_prefix_product_dev_noalloc(flag, flag, (1 << 24) + 1, flag)
        assert e.value.code == -4
    finally:
        backend.free(flag)


def test_two_contexts_on_one_gpu_share_the_array_in_turn(backend):
    """Two contexts on device 0 (each with a stream of its own) queue the call back to back without waiting in
    between: the library orders their uses of the GPU's one scratch array, so both results are right."""
    rng = np.random.default_rng(11)
    other = plonk_amd.Backend(device=0)
    n = (1 << 21) + 777
    inputs = []
    try:
        for be in (backend, other, backend, other):
            num, den = rand(rng, n), rand(rng, n)
            den[den.sum(axis=1) == 0] = 1
            inputs.append((be, num, den, be.to_device(num), be.to_device(den), be.malloc(256)))
        for be, _, _, d_num, d_den, d_flag in inputs:  # four launches queued on two streams, no host wait
            be.fr_ratio_prefix_product_dev_noalloc(d_num, d_den, n, d_flag)
        backend.sync()
        other.sync()
        for be, num, den, d_num, d_den, d_flag in inputs:
            got = be.from_device(d_num, (n, 4))
            d_a, d_b = be.to_device(num), be.to_device(den)
            be.fr_ratio_prefix_product_dev(d_a, d_b, n)
            be.sync()
            assert np.array_equal(got, be.from_device(d_a, (n, 4)))
            assert int(be.from_device(d_flag, (1,), dtype=np.int32)[0]) == 0
            be.free(d_a)
            be.free(d_b)
    finally:
        backend.sync()
        other.sync()
        for be, _, _, d_num, d_den, d_flag in inputs:
            for p in (d_num, d_den, d_flag):
                be.free(p)
        other.close()
