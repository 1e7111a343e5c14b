import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, plonk_amd
be = plonk_amd.Backend(0)
logn = int(sys.argv[1]); n = 1 << logn
be.kzg_setup(np.array([3, 1, 4, 1], dtype=np.uint64), n - 1, want_points=False)
sc = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy(); sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
d = be.to_device(sc); out = be.malloc(144)
for _ in range(3):
    be.kzg_commit_dev(d, n, out); be.sync()
    be.kzg_commit_many_dev([(d, n)] * 8)
