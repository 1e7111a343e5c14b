"""Random 128-byte-line gather bandwidth for a few table sizes (development aid; run under
`rocprofv3 --pmc FETCH_SIZE` to calibrate that counter for this access pattern: each k_gather_rate
launch reads exactly 6144 blocks x 128 lanes x 64 lines x 128 B = 6.44 GB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd

be = plonk_amd.Backend(0)
for gib in [float(x) for x in sys.argv[1:]] or [0.125, 2.0, 24.0]:
    r = be.bench_gather_rate(int(gib * (1 << 30)))
    rc = be.bench_gather_rate(int(gib * (1 << 30)) | 1)
    print("table %.3f GiB: %.1f GB/s lane-per-line, %.1f GB/s eight lanes per line through LDS" % (
        gib, r / 1e9, rc / 1e9), flush=True)
