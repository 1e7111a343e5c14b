"""Wall time of the polynomial entry points at the sizes of a 2^20-gate circuit (host buffers in and
out, so host copies are included; development aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonk_amd


def rand(rng, n):
    a = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    return a


def timed(name, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        out = fn()
        best = min(best, time.perf_counter() - t)
    print("%-58s %8.2f ms" % (name, best * 1e3), flush=True)
    return out


be = plonk_amd.Backend(0)
rng = np.random.default_rng(5)
m = 1 << 20
roots = rand(rng, m)
z = timed("vanishing polynomial from 2^20 roots", lambda: be.construct_vanishing_polynomial_from_roots(roots))
q = rand(rng, 3 << 20)
f = timed("product (3*2^20) x (2^20 + 1) coefficients", lambda: be.fr_poly_mul(q, z))
timed("quotient of 2^22 coefficients by 2^20 roots", lambda: be.compute_q_zero_test_from_roots(roots, f))
a, b = rand(rng, 1 << 22), rand(rng, 1 << 22)
timed("ratio prefix product over 2^22 values", lambda: be.fr_ratio_prefix_product(a, b))
timed("batch inverse of 2^22 values", lambda: be.fr_batch_inverse(a))
timed("evaluation of 2^22 coefficients over Omega (2^22)", lambda: be.fr_evaluate_over_omega(a, 22))
timed("quotient by X^k - 1, k = 2^21 (2^22 coefficients)", lambda: be.compute_q_zero_test(1 << 21, a))
g = rand(rng, 3 << 19)
timed("transform of 3*2^19 values (mixed radix)", lambda: be.fr_ntt_mixed(g.reshape(1, -1, 4)))
