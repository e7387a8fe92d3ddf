#!/bin/bash
# kernel stats of the BA part of bench.py only (quick look while tuning): gpurun_out/ks.csv
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_prof -o r -- python3 $ROOT/bench.py --no-cpu-baseline --no-matcher --no-frontend --no-replay > $OUT/ks.log 2>&1
cp $OUT/ks_prof/r_kernel_stats.csv $OUT/ks.csv; rm -rf $OUT/ks_prof
python3 - <<PY
import csv,re
for r in csv.DictReader(open("$OUT/ks.csv")):
    n=r['Name']
    if 'svi::' in n:
        m=re.search(r'(k_\w+)(<[^>]*>)?',n)
        print("%-34s calls %6s avg %9.1f us  min %8.1f max %8.1f" % (m.group(0), r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
