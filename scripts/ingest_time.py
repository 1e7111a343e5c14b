"""Times SRS ingest from compressed encodings and batch compression (development aid).
usage: python scripts/ingest_time.py LOG_N"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonk_amd

logn = int(sys.argv[1])
n = 1 << logn
be = plonk_amd.Backend(0)
pts = be.kzg_setup(np.array([5, 6, 7, 8], dtype=np.uint64), n - 1)
t = time.perf_counter()
enc = b"".join(be.g1_compress(pts))
print("compress 2^%d points: %.1f ms" % (logn, (time.perf_counter() - t) * 1e3), flush=True)
for check in (False, True):
    t = time.perf_counter()
    be.srs_load_compressed(enc, check_subgroup=check)
    print("srs_load_compressed 2^%d points, subgroup check %s: %.1f ms (window tables included)" % (
        logn, check, (time.perf_counter() - t) * 1e3), flush=True)
t = time.perf_counter()
be.srs_load_projective(pts)
print("srs_load_projective 2^%d points: %.1f ms" % (logn, (time.perf_counter() - t) * 1e3), flush=True)
