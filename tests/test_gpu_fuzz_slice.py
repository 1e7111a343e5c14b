"""A bounded, seeded slice of scripts/fuzz_msm.py under the GPU suite: random sizes up to 2^17, every window
width the plan accepts (and the automatic one), uniform scalars and scalars from a small set of special values
(0, 1, r - 1, (r +- 1) / 2, 2^254, repeated), single commitments, batches, and openings.  Checked against the
ORACLE here (the script checks against the library's own evaluation): commit(f) == [f(tau)]G with f(tau) by
oracle/oracle.c (plonk/tests/kzg_tests.rs:36-69), v == f(u) and the proof by the pairing equation (:95-197)."""
import random

import numpy as np
import pytest

import plonk_amd
from oracle import capi as c
from oracle import pyref as o
from tests.gpu_util import compress

pytestmark = pytest.mark.gpu

SPECIAL = [0, 1, 2, o.R - 1, o.R - 2, (o.R - 1) // 2, (o.R + 1) // 2, 1 << 254, (1 << 254) - 1, (1 << 255) % o.R]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_commitments_and_openings(seed):
    c.build()
    rnd = random.Random(1000 + seed)
    g = c.g1_generator()
    for _ in range(12):
        n = max(1, int(2 ** rnd.uniform(0, 17)))
        width = rnd.choice([0, 0, 0] + list(range(2, 21)))
        be = plonk_amd.Backend(0, window_bits=width)
        try:
            tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
            be.kzg_setup(tau, n - 1, want_points=False)
            tau_g2 = plonk_amd.kzg_setup_tau_g2(tau)
            for _ in range(3):
                m = rnd.randint(1, n)
                kind = rnd.random()
                if kind < 0.5:
                    f = np.frombuffer(rnd.randbytes(m * 32), dtype=np.uint64).reshape(m, 4).copy()
                    f[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
                elif kind < 0.8:
                    f = c.fr_to_mont([rnd.choice(SPECIAL) for _ in range(m)])
                else:
                    f = c.fr_to_mont([rnd.choice(SPECIAL[:4])] * m)
                want = compress(c.g1_mul(g, c.fr_eval(f, tau)))
                got = be.kzg_commit(f)
                assert compress(got) == want, (n, width, m, kind)
                if rnd.random() < 0.4:
                    both = be.kzg_commit_batch([f, f[: max(1, m // 2)]])
                    assert compress(both[0]) == want, (n, width, m, kind)
                    assert compress(both[1]) == compress(c.g1_mul(g, c.fr_eval(f[: max(1, m // 2)], tau)))
                if m >= 2 and rnd.random() < 0.6:
                    u = c.fr_to_mont([rnd.choice(SPECIAL + [rnd.randrange(o.R)])])[0]
                    v, proof = be.kzg_evaluate(f, u)
                    assert np.array_equal(np.asarray(v).reshape(4), c.fr_eval(f, u)), (n, width, m)
                    assert plonk_amd.kzg_verify(tau_g2, got, u, v, proof), (n, width, m)
        finally:
            be.close()
