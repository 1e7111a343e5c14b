import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, plonk_amd
logn = int(sys.argv[1]); n = 1 << logn
sc = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy(); sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
for c in [int(x) for x in sys.argv[2:]]:
    be = plonk_amd.Backend(0, window_bits=c)
    be.kzg_setup(np.array([3, 1, 4, 1], dtype=np.uint64), n - 1, want_points=False)
    d = be.to_device(sc); out = be.malloc(144)
    be.profile(True)
    runs = []
    for _ in range(40):  # the first ones run on a rising clock: medians over the later half
        be.kzg_commit_dev(d, n, out); runs.append(be.msm_phase_ms())
    ph = {k: float(np.median([r[k] for r in runs[20:]])) for k in runs[0]}
    print(logn, c, be.msm_plan(), {k: round(v, 3) for k, v in ph.items()}, round(sum(ph.values()), 3), flush=True)
    be.free(d); be.free(out); be.close()
