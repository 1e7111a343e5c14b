"""Per-kernel summary of a rocprofv3 --kernel-trace CSV directory (development aid).
usage: python scripts/trace_summary.py DIR [name-filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(list)
for path in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("void ", "").replace("plonk::", "").split("(")[0]
        if flt and flt not in name:
            continue
        key = (name, r["Grid_Size_X"], r["Workgroup_Size_X"])
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-46s grid=%-9s wg=%-4s n=%-4d avg=%9.1f med=%9.1f min=%9.1f us" % (
        key[0][:46], key[1], key[2], len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0]))
