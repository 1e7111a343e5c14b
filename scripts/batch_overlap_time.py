"""Eight commitments one after the other against the same eight in one call (tails overlapped with the next
accumulation), per commitment in ms (development aid)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, plonk_amd
be = plonk_amd.Backend(0)
for logn in (12, 14, 16, 18, 20, 21, 22):
    n = 1 << logn
    be.kzg_setup(np.array([3, 1, 4, 1], dtype=np.uint64), n - 1, want_points=False)
    sc = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy(); sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    d = be.to_device(sc); out = be.malloc(144)
    res = []
    for rep in range(8):  # the later repetitions run on a warm clock
        be.kzg_commit_dev(d, n, out); be.sync()
        t = time.perf_counter()
        for _ in range(8): be.kzg_commit_dev(d, n, out)
        be.sync(); single = (time.perf_counter() - t) / 8 * 1e3
        t = time.perf_counter(); be.kzg_commit_many_dev([(d, n)] * 8); batch = (time.perf_counter() - t) / 8 * 1e3
        res.append((round(single, 3), round(batch, 3)))
    print(logn, res, flush=True)
    be.free(d); be.free(out)
