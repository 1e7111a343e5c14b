"""HBM bytes per launch / per step from the FETCH_SIZE and WRITE_SIZE passes of scripts/profile.sh, as the
JSON record bench.py reads from profiles/traffic.json.
usage: python scripts/pmc_traffic.py TAG COMMIT   (reads gpurun_out/prof_TAG_{msm,ntt,quo}_{fetch,write})

gfx950: FETCH_SIZE (KB) counts 128-byte requests as 64 bytes, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE (KB)
is exact for wide streaming stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(directory, counter):
    out = defaultdict(float)
    names = {}
    for path in glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                out[row["Dispatch_Id"]] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = row["Kernel_Name"]
    return out, names


def kernel_bytes(tag, leg, needle):
    """(launches, total bytes) of the kernels whose name contains `needle`."""
    base = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, leg))
    fetch, names = per_dispatch(base + "_fetch", "FETCH_SIZE")
    write, wnames = per_dispatch(base + "_write", "WRITE_SIZE")
    rd = [v for k, v in fetch.items() if needle in names[k]]
    wr = [v for k, v in write.items() if needle in wnames[k]]
    if not rd or len(rd) != len(wr):
        raise SystemExit("no matching launches of %s in %s (%d reads, %d writes)" % (needle, base, len(rd), len(wr)))
    return len(rd), (2 * sum(rd) + sum(wr)) * 1024


def main():
    tag, commit = sys.argv[1], sys.argv[2]
    entries = {}
    n, total = kernel_bytes(tag, "msm", "k_accumulate")
    entries["msm_2^24_x1"] = {"k_accumulate": {
        "bytes": int(total / n), "what": "average over %d launches of one 2^24-point commitment" % n}}
    # bench.py --workload ntt / quotient --steps 3 --warmup 1: four steps, two passes per transform
    for leg, key, steps in (("ntt", "ntt_16x2^20", 4), ("quo", "quotient_16x2^20", 4)):
        n, total = kernel_bytes(tag, leg, "k_ntt_pass")
        entries[key] = {"k_ntt_pass": {
            "bytes": int(total / steps),
            "what": "all %d pass launches of %d steps, per step (16 polynomials)" % (n, steps)}}
    json.dump({"commit": commit, "tag": tag, "entries": entries}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
