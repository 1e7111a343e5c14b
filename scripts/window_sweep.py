"""Single and batch-of-8 commitment time per window width (development aid).
usage: python scripts/window_sweep.py LOG_N C [C ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonk_amd

logn = int(sys.argv[1])
n = 1 << logn
sc = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
for c in [int(x) for x in sys.argv[2:]]:
    be = plonk_amd.Backend(0, window_bits=c)
    be.kzg_setup(np.array([3, 1, 4, 1], dtype=np.uint64), n - 1, want_points=False)
    d = be.to_device(sc)
    out = be.malloc(144)
    best = (1e9, 1e9)
    for rep in range(4):
        be.kzg_commit_dev(d, n, out)
        be.sync()
        t = time.perf_counter()
        for _ in range(8):
            be.kzg_commit_dev(d, n, out)
        be.sync()
        single = (time.perf_counter() - t) / 8 * 1e3
        t = time.perf_counter()
        be.kzg_commit_many_dev([(d, n)] * 8)
        batch = (time.perf_counter() - t) / 8 * 1e3
        best = (min(best[0], single), min(best[1], batch))
    print("log_n %d c %d windows %d: single %.3f ms, in a batch of 8 %.3f ms" % (
        logn, c, be.msm_plan()["windows"], best[0], best[1]), flush=True)
    be.free(d)
    be.free(out)
    be.close()
