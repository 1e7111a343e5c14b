"""Register-only rate probes (development aid): integer mad peak and field multiplication forms."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd

be = plonk_amd.Backend(0)
peak = be.bench_mad_rate()
print("v_mad_u64_u32 peak: %.2f Tmad/s" % (peak / 1e12))
for extra in (0, 1, 2, 4):
    for bpc in (1, 2, 8):
        r, clk = be.bench_mad_mix(extra, bpc)
        print("mad + %d add  blocks/CU=%d: %.2f Tmad/s, clock %.2f GHz -> %.2f cycles per wave-mad group" % (
            extra, bpc, r / 1e12, clk, clk * 1e9 / (r / 64 / 1024)))
names = {0: "fe_mul (device form, 12x32)", 1: "fe_mul_cios (12x32)", 2: "q28_mul (14x28)", 3: "q28_sqr+q28_mul"}
mads = {0: 300, 1: 300, 2: 392, 3: (392 + 301) / 2}
for variant in (1, 2):
    for bpc in (2, 4, 8):
        r = be.bench_fq_mul_rate(variant, bpc)
        print("%-32s blocks/CU=%d  %.2f Gmul/s  (%.2f Tmad/s executed, %.2f Tmad/s at 300/mul, %.1f%% of peak)" % (
            names[variant], bpc, r / 1e9, r * mads[variant] / 1e12, r * 300 / 1e12, 100 * r * 300 / peak))
