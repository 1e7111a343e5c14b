"""Helpers shared by the GPU parity tests: one Backend per session, conversions between the
oracle's Python ints and the limb arrays that cross the C ABI."""
import numpy as np
import pytest

from oracle import capi as c
from oracle import pyref as o


def compress(pt):
    """(18,) Jacobian limbs -> 48-byte compressed encoding (via the oracle: the checker)."""
    return c.g1_compress(np.ascontiguousarray(pt, dtype=np.uint64).reshape(1, 18))[0]


def same_point(a, b):
    return compress(a) == compress(b)


def rand_fr(rnd, n):
    return [rnd.randrange(o.R) for _ in range(n)]


@pytest.fixture(scope="session")
def backend():
    import plonk_amd

    c.build()
    be = plonk_amd.Backend(device=0)
    yield be
    be.close()
