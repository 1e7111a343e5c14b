#!/bin/bash
# Kernel statistics of a 2^LOG-point commitment with the sort's first pass made from the scalars / from pairs.
# usage (on the GPU box): bash scripts/front_end_profile.sh LOG TAG
set -e
LOG=${1:-24}; TAG=${2:-fe}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for m in scalars pairs; do
  rm -rf /tmp/prof_$m
  PLONK_HIP_FRONT_END=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -o p --output-format csv -- python3 $ROOT/scripts/msm_phases.py $LOG 0 5 > /tmp/p_$m.log 2>&1
  grep "Mpts" /tmp/p_$m.log | sed "s/^/$m /"
  f=$(find /tmp/prof_$m -name "*kernel_stats.csv" | head -1)
  cp $f $ROOT/gpurun_out/${TAG}_${m}_kernel_stats.csv
  python3 - $f <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0]
    if any(k in n for k in ('radix', 'digits', 'bounds')):
        print("   ", n, r['Calls'], "avg %.1f min %.1f max %.1f us" % (float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
P
done
