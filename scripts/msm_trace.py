"""One context, a few commitments of 2^LOG_N points (development aid for kernel traces).
usage: python scripts/msm_trace.py LOG_N [REPS]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonk_amd

logn = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1 << logn
be = plonk_amd.Backend(0)
be.kzg_setup(np.array([3, 1, 4, 1], dtype=np.uint64), n - 1, want_points=False)
sc = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
d = be.to_device(sc)
out = be.malloc(144)
for _ in range(reps):
    be.kzg_commit_dev(d, n, out)
be.sync()
