set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp; cd /tmp
for W in ntt quotient; do
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_128B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_32B TCC_EA0_RDREQ --output-format csv -d $OUT/rdsz_$W -- python3 $ROOT/bench.py --workload $W --steps 1 --warmup 1 > $OUT/rdsz_$W.log 2>&1 || exit 1
done
cd $ROOT
python3 - <<'PY'
import csv,glob
from collections import defaultdict
for w in ("ntt","quotient"):
    d=defaultdict(lambda: defaultdict(float)); names={}
    for path in glob.glob('gpurun_out/rdsz_%s/**/*_counter_collection.csv'%w,recursive=True):
        for r in csv.DictReader(open(path)):
            d[int(r['Dispatch_Id'])][r['Counter_Name']]+=float(r['Counter_Value']); names[int(r['Dispatch_Id'])]=r['Kernel_Name'][:34]
    for k in sorted(d):
        if 'ntt_pass' in names[k]:
            v=d[k]; print(w, names[k][12:30], "read MB", round((128*v['TCC_EA0_RDREQ_128B']+64*v['TCC_EA0_RDREQ_64B']+32*v['TCC_EA0_RDREQ_32B'])/1e6), {c[9:]:int(x) for c,x in v.items()})
PY
