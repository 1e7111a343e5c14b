"""Accumulation and fix-up time against the segment length (development aid): the library's own choice next to the
exact fits (the shortest segment length with at most R * 131072 segments, 131072 = the lanes the accumulation
keeps resident) and their neighbours, interleaved in one process so that the box's clock drift shows up as a
difference between the first and the last line.
usage: python scripts/seg_len_sweep.py LOG_N WINDOW_BITS [uniform|equal|minus_one|bits] [R ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import plonk_amd

logn, c = int(sys.argv[1]), int(sys.argv[2])
rest = sys.argv[3:]
kind = rest.pop(0) if rest and not rest[0].isdigit() else "uniform"
rounds = [int(x) for x in rest]
n = 1 << logn
be = plonk_amd.Backend(0, window_bits=c)
be.kzg_setup(np.array([3, 1, 4, 1], dtype=np.uint64), n - 1, want_points=False)
if kind == "uniform":
    sc = bench.synthetic_scalars(n, 1)
else:
    sets = bench.adversarial_scalar_sets(n, 5)
    sc = sets[{"equal": "all_equal", "minus_one": "all_r_minus_1", "bits": "zeros_and_ones"}[kind]]
d, out = be.to_device(sc), be.malloc(144)
total = n * be.msm_plan()["windows"]
be.profile(True)


def run(seg_len, label, pair_loads=0):
    be.bench_msm_tuning(seg_len, pair_loads)
    runs = []
    for _ in range(24):
        be.kzg_commit_dev(d, n, out)
        runs.append(be.msm_phase_ms())
    ph = {k: float(np.median([r[k] for r in runs[10:]])) for k in runs[0]}
    print("%-28s accumulate %.3f fixup %.3f reduce %.3f  sum %.3f  all %.3f" % (
        label, ph["accumulate"], ph["fixup"], ph["reduce"], ph["accumulate"] + ph["fixup"] + ph["reduce"],
        sum(ph.values())), flush=True)


for rep in range(2):
    run(0, "library's choice")
    run(0, "choice, pairs direct", 1)
    run(0, "choice, pairs by LDS", 2)
    for r in rounds:
        s = -(-total // (r * 131072))
        for cand in (s - 1, s, s + 1):
            if cand >= 1:
                run(cand, "R=%d s=%d (%.3f) direct" % (r, cand, -(-total // cand) / 131072), 1)
                run(cand, "R=%d s=%d (%.3f) by LDS" % (r, cand, -(-total // cand) / 131072), 2)
    run(0, "library's choice")
