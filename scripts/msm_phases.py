"""Per-phase GPU timing of the MSM for one size / window width (development aid).
usage: python scripts/msm_phases.py LOG_N [WINDOW_BITS] [ITERS]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonk_amd

logn = int(sys.argv[1])
cbits = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
be = plonk_amd.Backend(0)
n = 1 << logn
be.set_window_bits(cbits)
tau = np.array([5, 6, 7, 8], dtype=np.uint64)
t = time.time()
be.kzg_setup(tau, n - 1, want_points=False)
ts = time.time() - t
sc = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
d_sc = be.to_device(sc)
d_out = be.malloc(144)
be.profile(True)
for it in range(iters):
    t = time.time()
    be.kzg_commit_dev(d_sc, n, d_out)
    be.sync()
    dt = time.time() - t
ph = be.msm_phase_ms()
print(logn, be.msm_plan(), "setup %.2fs" % ts, "msm %.2f ms" % (dt * 1e3), "Mpts/s %.1f" % (n / dt / 1e6),
      {k: round(v, 3) for k, v in ph.items()}, flush=True)
