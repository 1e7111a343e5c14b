#!/usr/bin/env python3
"""Commitment time on scalar sets that skew the bucket method (all equal, all r - 1, one in sixteen zero,
only 0 and 1, all zero) next to uniform scalars, with the phase split of each, at 2^20 and 2^24 points.

    python scripts/adversarial_time.py [log_n ...]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

import bench
import plonk_amd


def main():
    logs = [int(a) for a in sys.argv[1:]] or [20, 24]
    tau = np.array([0x0123456789ABCDEF, 0x0FEDCBA987654321, 0x1122334455667788, 0x0102030405060708],
                   dtype=np.uint64)
    for log_n in logs:
        n = 1 << log_n
        be = plonk_amd.Backend(device=0)
        be.kzg_setup(tau, n - 1, want_points=False)
        rec = bench.time_adversarial(be, log_n, tau, 5)
        # phase split of every set (events between the kernels)
        d_sc = be.malloc(n * 32)
        d_out = be.malloc(144)
        be.profile(True)
        sets = dict(bench.adversarial_scalar_sets(n, 77 + log_n))
        sets["uniform"] = bench.synthetic_scalars(n, 4242)
        for name, sc in sets.items():
            be.to_device(sc, d_sc)
            for _ in range(3):
                be.kzg_commit_dev(d_sc, n, d_out)
            ph = {k: round(v, 3) for k, v in be.msm_phase_ms().items()}
            if name == "uniform":
                rec["uniform_phase_ms"] = ph
            else:
                rec[name]["phase_ms"] = ph
        be.free(d_sc)
        be.free(d_out)
        be.close()
        print(json.dumps({"log_n": log_n, **rec}), flush=True)


if __name__ == "__main__":
    main()
