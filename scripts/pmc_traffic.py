"""HBM bytes per launch / per step from the counter passes of scripts/profile.sh, as the JSON record
bench.py reads from profiles/traffic.json.
usage: python scripts/pmc_traffic.py TAG COMMIT   (reads gpurun_out/prof_TAG_{msm,ntt,quo}_{fetch,write,rdsz})

Reads: gfx950's FETCH_SIZE (KB) tallies a 128-byte request as 64 bytes, so for wide streaming reads the
bytes are 2 x FETCH_SIZE; kernels that read 64-byte rows (the strided transform passes) are counted
exactly by FETCH_SIZE itself.  The size-resolved request counters settle it per kernel:
    read bytes = 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B.
Both figures are recorded; `bytes` uses the size-resolved one.  Writes: WRITE_SIZE (KB), exact for wide
streaming stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_dispatch(directory, counters):
    """{dispatch: {counter: value}}, {dispatch: (kernel name, grid size)}"""
    out = defaultdict(lambda: defaultdict(float))
    meta = {}
    for path in glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] not in counters:
                    continue
                out[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
                meta[row["Dispatch_Id"]] = (row["Kernel_Name"], int(row["Grid_Size"]))
    return out, meta


def kernel_bytes(tag, leg, needle, min_grid=0):
    """Launches of the kernels whose name contains `needle` (grid of at least min_grid work items): count, total
    bytes by the size-resolved read counters + WRITE_SIZE, total bytes by 2 x FETCH_SIZE + WRITE_SIZE."""
    base = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, leg))

    def pick(suffix, counters):
        vals, meta = per_dispatch(base + suffix, counters)
        return [vals[k] for k in sorted(vals, key=int) if needle in meta[k][0] and meta[k][1] >= min_grid]

    fetch = pick("_fetch", {"FETCH_SIZE"})
    write = pick("_write", {"WRITE_SIZE"})
    sized = pick("_rdsz", {"TCC_EA0_RDREQ_128B", "TCC_EA0_RDREQ_64B", "TCC_EA0_RDREQ_32B"})
    if not fetch or not write or not sized:
        raise SystemExit("no launches of %s in %s" % (needle, base))
    # every counter pass is a run of its own and a clock warm-up that lasts a fixed time launches a different
    # number of kernels in each: per-launch averages pass by pass
    wr = sum(v["WRITE_SIZE"] for v in write) * 1024 / len(write)
    rd_sized = sum(128 * v["TCC_EA0_RDREQ_128B"] + 64 * v["TCC_EA0_RDREQ_64B"] + 32 * v["TCC_EA0_RDREQ_32B"]
                   for v in sized) / len(sized)
    rd_fetch2 = sum(v["FETCH_SIZE"] for v in fetch) * 2 * 1024 / len(fetch)
    return min(len(write), len(fetch), len(sized)), rd_sized + wr, rd_fetch2 + wr, rd_sized, wr


def main():
    tag, commit = sys.argv[1], sys.argv[2]
    entries = {}
    n, total, total_f2, rd, wr = kernel_bytes(tag, "msm", "k_accumulate", min_grid=1 << 16)
    entries["msm_2^24_x1"] = {"k_accumulate": {
        "bytes": int(total), "bytes_by_2x_fetch_size": int(total_f2), "read_bytes": int(rd),
        "write_bytes": int(wr), "what": "average over %d launches of one 2^24-point commitment" % n}}
    # bench.py --workload ntt / quotient: a step is two pass launches (one transform of 2^20 points per polynomial)
    # or four (two transforms of 2^21 points); the clock warm-up ahead of the timed steps runs the same step
    for leg, key, per_step in (("ntt", "ntt_16x2^20", 2), ("quo", "quotient_16x2^20", 4)):
        n, total, total_f2, rd, wr = kernel_bytes(tag, leg, "k_ntt_pass")
        entries[key] = {"k_ntt_pass": {
            "bytes": int(total * per_step), "bytes_by_2x_fetch_size": int(total_f2 * per_step),
            "read_bytes": int(rd * per_step), "write_bytes": int(wr * per_step),
            "what": "average over at least %d pass launches, times the %d launches of a step (16 polynomials)" % (n, per_step)}}
    json.dump({"commit": commit, "tag": tag, "entries": entries}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
