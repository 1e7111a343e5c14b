"""Times the Fr NTT / coset quotient with data resident (development aid).
usage: python scripts/ntt_time.py LOG_N BATCH"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonk_amd

log_n = int(sys.argv[1])
batch = int(sys.argv[2])
be = plonk_amd.Backend(0)
WARM, REPS = 100, 200
n = 1 << log_n
data = np.frombuffer(np.random.default_rng(1).bytes(n * batch * 32), dtype=np.uint64).reshape(-1, 4).copy()
data[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
d = be.to_device(data)
for inverse in (False, True):
    # the shader clock takes tens of milliseconds of continuous work to come up from idle: warm up first
    for _ in range(WARM):
        be.fr_ntt_dev(d, log_n, batch, inverse)
    be.sync()
    t = time.perf_counter()
    for _ in range(REPS):
        be.fr_ntt_dev(d, log_n, batch, inverse)
    be.sync()
    ms = (time.perf_counter() - t) / REPS * 1e3
    print("ntt log_n=%d batch=%d inverse=%d: %.3f ms (%.4f ms per transform, %.1f GB/s algorithmic)" % (
        log_n, batch, inverse, ms, ms / batch, n * batch * 128 / ms / 1e6))
be.free(d)
if len(sys.argv) > 3:
    # coset quotient of batch polynomials of 2^(log_n+1) coefficients by X^(2^log_n) - 1
    q = be.to_device(np.zeros((batch * 2 * n, 4), dtype=np.uint64))
    for _ in range(WARM // 4):
        be.quotient_coset_dev(q, log_n, batch)
    be.sync()
    t = time.perf_counter()
    for _ in range(REPS // 4):
        be.quotient_coset_dev(q, log_n, batch)
    be.sync()
    print("coset quotient log_k=%d batch=%d: %.3f ms" % (log_n, batch, (time.perf_counter() - t) / (REPS // 4) * 1e3))
    be.free(q)
