"""Additions per second of the batch-affine probe (K accumulators per lane, one inversion per block of 256 lanes,
with and without the inversion) next to the rate of the accumulation kernel in production (development aid).
usage: python scripts/batch_affine_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonk_amd

be = plonk_amd.Backend(0)
n = 1 << 22
be.kzg_setup(np.array([3, 1, 4, 1], dtype=np.uint64), n - 1, want_points=False)
sc = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
d, out = be.to_device(sc), be.malloc(144)
be.profile(True)
for _ in range(8):
    be.kzg_commit_dev(d, n, out)
    acc_ms = be.msm_phase_ms()["accumulate"]
plan = be.msm_plan()
print("k_accumulate (mixed XYZZ additions): %.2f G additions/s (2^22 points x %d windows in %.3f ms)" % (
    n * plan["windows"] / acc_ms / 1e6, plan["windows"], acc_ms), flush=True)
be.profile(False)
for k in (1, 2, 4):
    for invert in (True, False):
        for _ in range(2):
            rate = be.bench_batch_affine(k, 24, invert=invert)
        print("batch affine, %d accumulators per lane, %s: %.3f G additions/s" % (
            k, "one inversion per 256 lanes and step" if invert else "inversion skipped (free)", rate / 1e9), flush=True)
