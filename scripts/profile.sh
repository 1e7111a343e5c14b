#!/bin/bash
# Kernel-trace and counter passes over bench.py (run on the GPU box from the repo root).
# usage: bash scripts/profile.sh TAG [COMMIT]   -> gpurun_out/prof_TAG_*/ and gpurun_out/traffic_TAG.json
# Every counter pass is a run of its own (a pass holds at most 8 SQ and 4 TCC counters; FETCH_SIZE
# and WRITE_SIZE do not fit together) and never shares a run with a trace.
set -u
TAG=${1:-x}
COMMIT=${2:-unknown}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
MSM="$ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
NTT="$ROOT/bench.py --workload ntt --steps 20 --warmup 2"
QUO="$ROOT/bench.py --workload quotient --steps 10 --warmup 2"
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS GRBM_GUI_ACTIVE GRBM_COUNT"

# read requests by size: FETCH_SIZE tallies a 128-byte request as 64 bytes, which suits wide streaming reads
# (double it) but not kernels that read 64-byte rows; the size-resolved counts give the bytes either way
RDSZ="TCC_EA0_RDREQ_128B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_32B TCC_EA0_RDREQ"

pass() {  # name, counters, command...
    local name=$1 counters=$2
    shift 2
    timeout -s KILL 240 rocprofv3 --pmc $counters --output-format csv -d $OUT/prof_${TAG}_$name -- python3 "$@" \
        > $OUT/prof_${TAG}_$name.log 2>&1 || { echo "pass $name failed"; exit 1; }
}

# the whole default bench (extras included) as a kernel trace with per-kernel statistics
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats -- \
    python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_${TAG}_stats.log 2>&1 || exit 1
# the headline command alone (2^24 commitments only): its per-kernel averages are per 2^24 launch
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats_msm -- \
    python3 $MSM > $OUT/prof_${TAG}_stats_msm.log 2>&1 || exit 1
# the transform and quotient workloads alone: their pass kernels' averages are per launch of THAT workload
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats_ntt -- \
    python3 $NTT > $OUT/prof_${TAG}_stats_ntt.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_stats_quo -- \
    python3 $QUO > $OUT/prof_${TAG}_stats_quo.log 2>&1 || exit 1
pass msm_sq1 "$SQ1" $MSM
pass msm_sq2 "$SQ2" $MSM
pass msm_fetch FETCH_SIZE $MSM
pass msm_write WRITE_SIZE $MSM
pass msm_rdsz "$RDSZ" $MSM
pass ntt_sq1 "$SQ1" $NTT
pass ntt_sq2 "$SQ2" $NTT
pass ntt_fetch FETCH_SIZE $NTT
pass ntt_write WRITE_SIZE $NTT
pass ntt_rdsz "$RDSZ" $NTT
pass quo_fetch FETCH_SIZE $QUO
pass quo_write WRITE_SIZE $QUO
pass quo_rdsz "$RDSZ" $QUO
cd $ROOT
python3 scripts/pmc_traffic.py $TAG $COMMIT > $OUT/traffic_$TAG.json || exit 2
python3 scripts/summarize_pmc.py $OUT/pmc_${TAG}_msm.csv $OUT/prof_${TAG}_msm_sq1 $OUT/prof_${TAG}_msm_sq2 \
    $OUT/prof_${TAG}_msm_fetch $OUT/prof_${TAG}_msm_write $OUT/prof_${TAG}_msm_rdsz
python3 scripts/summarize_pmc.py $OUT/pmc_${TAG}_ntt.csv $OUT/prof_${TAG}_ntt_sq1 $OUT/prof_${TAG}_ntt_sq2 \
    $OUT/prof_${TAG}_ntt_fetch $OUT/prof_${TAG}_ntt_write $OUT/prof_${TAG}_ntt_rdsz
python3 scripts/summarize_pmc.py $OUT/pmc_${TAG}_quotient.csv $OUT/prof_${TAG}_quo_fetch $OUT/prof_${TAG}_quo_write $OUT/prof_${TAG}_quo_rdsz
echo done
