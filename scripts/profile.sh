#!/bin/bash
# Kernel-trace and counter passes over bench.py (run on the GPU box from the repo root).
# usage: bash scripts/profile.sh TAG     -> gpurun_out/prof_TAG_{stats,sq,fetch,write}/
set -u
TAG=${1:-x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
BENCH="$ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
# counter passes: the 2^24 commitment only, so that per-kernel averages are per 2^24 launch
PMC="$BENCH --no-extras"
if [ "${2:-}" = "pmc-only" ]; then SKIP_STATS=1; else SKIP_STATS=0; fi
[ $SKIP_STATS = 1 ] || timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -- python3 $BENCH > $OUT/prof_${TAG}_stats.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/prof_${TAG}_sq -- python3 $PMC > $OUT/prof_${TAG}_sq.log 2>&1 || exit 2
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 $PMC > $OUT/prof_${TAG}_fetch.log 2>&1 || exit 3
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -- python3 $PMC > $OUT/prof_${TAG}_write.log 2>&1 || exit 4
echo done
