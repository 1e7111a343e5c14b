"""Samples board power and shader clock (rocm-smi) while one workload runs in a loop (development aid).
usage: python scripts/power_probe.py ntt|msm|mad"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonk_amd

what = sys.argv[1]
be = plonk_amd.Backend(0)
stop = False


def sample():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True).stdout
        keep = [l.strip() for l in out.splitlines() if "Power" in l or "sclk" in l or "mclk" in l]
        print(" | ".join(keep), flush=True)
        time.sleep(0.3)


if what == "ntt":
    n, batch = 1 << 20, 16
    data = np.frombuffer(np.random.default_rng(1).bytes(n * batch * 32), dtype=np.uint64).reshape(-1, 4).copy()
    data[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    d = be.to_device(data)
    run = lambda: be.fr_ntt_dev(d, 20, batch, False)
elif what == "msm":
    n = 1 << 22
    sc = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    be.kzg_setup(np.array([3, 1, 4, 1], dtype=np.uint64), n - 1, want_points=False)
    d = be.to_device(sc)
    out = be.malloc(144)
    run = lambda: be.kzg_commit_dev(d, n, out)
else:
    run = lambda: be.bench_mad_rate()
run()
be.sync()
th = threading.Thread(target=sample)
th.start()
t0 = time.perf_counter()
count = 0
while time.perf_counter() - t0 < 4.0:
    for _ in range(20):
        run()
    be.sync()
    count += 20
el = time.perf_counter() - t0
stop = True
th.join()
print(what, "%.3f ms per call" % (el / count * 1e3))
