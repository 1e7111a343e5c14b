import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, plonk_amd
be = plonk_amd.Backend(0)
for logn in (22, 24):
    n = 1 << logn
    be.kzg_setup(np.array([3, 1, 4, 1], dtype=np.uint64), n - 1, want_points=False)
    sc = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy(); sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    d = be.to_device(sc); out = be.malloc(144)
    be.kzg_commit_dev(d, n, out); be.sync()
    t = time.perf_counter(); be.kzg_commit_dev(d, n, out); be.sync(); res = (time.perf_counter() - t) * 1e3
    want = be.from_device(out, (18,))
    got = be.kzg_commit(sc)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); got = be.kzg_commit(sc); ts.append((time.perf_counter() - t) * 1e3)
    enc = be.g1_compress(np.stack([want, got]))
    print(logn, "resident %.2f ms, host buffers %s ms, equal %s" % (res, [round(x, 2) for x in ts], enc[0] == enc[1]), flush=True)
    be.free(d); be.free(out)
