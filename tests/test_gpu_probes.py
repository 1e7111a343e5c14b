"""The batch-affine design probe (plonk_hip_bench_batch_affine) computes real curve additions: with the shared
inversion switched on, the accumulators of its first block equal the sums the oracle forms from the same SRS
points (P + Q by the group law of ark-ec as restated in oracle/oracle.c).  The timing it reports is only worth
reading if the arithmetic is right."""
import numpy as np
import pytest

from oracle import capi as c
from tests.gpu_util import backend, compress  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,steps", [(1, 2), (2, 3), (4, 2)])
def test_batch_affine_probe_adds_points_correctly(backend, k, steps):
    n = 4096
    tau = c.fr_to_mont([0x5EED5EED5EED + k])[0]
    srs = backend.kzg_setup(tau, n - 1)
    rate, got = backend.bench_batch_affine(k, steps, invert=True, want_points=True)
    assert rate > 0
    for lane in (0, 1, 100, 255):
        for j in range(k):
            g = lane * k + j
            want = srs[g % n]
            for it in range(1, steps + 1):
                want = c.g1_add(want, srs[(g + it * 1021) % n])
            assert compress(got[lane * k + j]) == compress(want), (lane, j)
