"""Issue cost of the VALU instructions the field arithmetic is made of (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd

names = ["v_lshrrev_b64", "v_lshl_add_u64", "v_alignbit_b32", "v_and_b32", "v_add3_u32", "v_mad_i64_i32",
         "v_ashrrev_i64", "v_mul_lo_u32", "v_mad_u64_u32", "v_sub_u32"]
be = plonk_amd.Backend(0)
_, ghz = be.bench_mad_mix(0, 8)
for op, name in enumerate(names):
    r = be.bench_op_rate(op)
    # 1024 SIMDs x 64 lanes
    cyc = 1024 * 64 * ghz * 1e9 / r
    print("%-16s %8.2f Tlane-op/s  %.2f cycles per wave-instruction at %.2f GHz" % (name, r / 1e12, cyc, ghz), flush=True)

# one wave per SIMD: does a lone wave reach the issue rate, and what does dependence cost?
for op, name in ((108, "v_mad_u64_u32, 8 chains, lone wave"), (111, "v_mad_u64_u32, 2 chains, lone wave"),
                 (110, "v_mad_u64_u32, 1 chain, lone wave"), (10, "v_mad_u64_u32, 1 chain, 8 waves"),
                 (101, "v_lshl_add_u64, 8 chains, lone wave"), (103, "v_and_b32, 8 chains, lone wave")):
    r = be.bench_op_rate(op)
    cyc = 1024 * 64 * ghz * 1e9 / r
    print("%-38s %8.2f Tlane-op/s  %.2f cycles per wave-instruction" % (name, r / 1e12, cyc), flush=True)
