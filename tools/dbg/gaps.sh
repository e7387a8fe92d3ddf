#!/bin/bash
# idle time between consecutive kernels of the LM loop (rocprofv3 kernel trace of a short bench run): gpurun_out/gaps.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/gap_prof -o r -- python3 $ROOT/bench.py --no-cpu-baseline --no-matcher --no-frontend --no-replay --steps 20 --warmup 5 > $OUT/gap.log 2>&1
python3 - <<PY
import csv, re, collections
rows=[r for r in csv.DictReader(open("$OUT/gap_prof/r_kernel_trace.csv"))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
gaps=collections.defaultdict(list)
prev=None
for r in rows:
    n=re.search(r'(k_\w+)',r['Kernel_Name']); n=n.group(1) if n else r['Kernel_Name'][:30]
    if prev is not None:
        g=int(r['Start_Timestamp'])-prev[1]
        if g<200000: gaps[(prev[0],n)].append(g)
    prev=(n,int(r['End_Timestamp']))
tot=0
for k,v in sorted(gaps.items(), key=lambda kv:-sum(kv[1])):
    if len(v)>=20: print("%-22s -> %-22s n %5d avg gap %7.2f us" % (k[0],k[1],len(v),sum(v)/len(v)/1e3))
PY
rm -rf $OUT/gap_prof
