"""Averages rocprofv3 counter_collection.csv files per kernel and counter.
usage: python scripts/summarize_pmc.py OUT.csv DIR [DIR ...]   (each DIR searched for *_counter_collection.csv)"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("plonk::", "")
    if "rocprim" in name:
        m = re.search(r"(onesweep_histograms|onesweep_iteration|radix_sort_\w+|scan\w*|transform\w*)", name)
        name = "rocprim::" + (m.group(1) if m else "kernel")
    return name[:60]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    key = (row["Dispatch_Id"], row["Counter_Name"])
                    per_dispatch[key] += float(row["Counter_Value"])
                    names[row["Dispatch_Id"]] = short(row["Kernel_Name"])
                    meta[short(row["Kernel_Name"])] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"],
                                                      row["LDS_Block_Size"], row["Scratch_Size"], row["Workgroup_Size"])
            for (disp, counter), v in per_dispatch.items():
                acc[names[disp]][counter].append(v)
    counters = sorted({c for k in acc.values() for c in k})
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "vgpr", "agpr", "sgpr", "lds_bytes", "scratch_bytes", "workgroup"]
                   + [c + "_avg_per_dispatch" for c in counters])
        for k in sorted(acc):
            n = max(len(v) for v in acc[k].values())
            m = meta[k]
            w.writerow([k, n, *m] + ["%.0f" % (sum(acc[k][c]) / len(acc[k][c])) if c in acc[k] else "" for c in counters])


main()
