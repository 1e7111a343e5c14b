"""Does per-phase event timing (plonk_hip_profile) slow the 2^24 commitment down?  Ten commitments with and
without it, twice (development aid; measured: 36.47 / 36.40 / 36.42 / 36.46 ms -- it does not)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, plonk_amd
n = 1 << 24
sc = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy(); sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
be = plonk_amd.Backend(0)
be.kzg_setup(np.array([3, 1, 4, 1], dtype=np.uint64), n - 1, want_points=False)
d = be.to_device(sc); out = be.malloc(144)
for mode in ("plain", "profile", "plain", "profile"):
    be.profile(mode == "profile")
    for _ in range(2): be.kzg_commit_dev(d, n, out)
    be.sync()
    t = time.perf_counter()
    for _ in range(10):
        be.kzg_commit_dev(d, n, out)
        if mode == "profile": ph = be.msm_phase_ms()
    be.sync()
    print(mode, round((time.perf_counter() - t) / 10 * 1e3, 3), flush=True)
