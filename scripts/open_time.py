"""Times kzg_commit / kzg_evaluate through the host-pointer entry points (development aid).
usage: python scripts/open_time.py LOG_N"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonk_amd

logn = int(sys.argv[1])
n = 1 << logn
be = plonk_amd.Backend(0)
be.kzg_setup(np.array([5, 6, 7, 8], dtype=np.uint64), n - 1, want_points=False)
f = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
f[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
u = np.array([11, 12, 13, 14], dtype=np.uint64)
for name, fn in (("commit", lambda: be.kzg_commit(f)), ("open", lambda: be.kzg_evaluate(f, u)),
                 ("eval", lambda: be.fr_eval(f, u))):
    fn()
    t = time.perf_counter()
    for _ in range(3):
        fn()
    print("%s 2^%d: %.2f ms (host buffers)" % (name, logn, (time.perf_counter() - t) / 3 * 1e3), flush=True)
