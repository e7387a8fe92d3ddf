#!/bin/bash
# kernel timeline of one LM iteration in the middle of a run (rocprofv3 kernel trace): start offset and duration of every kernel
# relative to the iteration's k_schur launch -> gpurun_out/timeline.txt      usage: tools/dbg/timeline.sh [env assignments...]
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tl_prof
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_prof -o r -- python3 $ROOT/tools/dbg/stage_bench.py > $OUT/tl.log 2>&1
python3 - <<PY
import csv, re
rows=[r for r in csv.DictReader(open("$OUT/tl_prof/r_kernel_trace.csv"))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def nm(r):
    n=re.search(r'(k_\w+)',r['Kernel_Name']); return n.group(1) if n else r['Kernel_Name'][:30]
idx=[i for i,r in enumerate(rows) if nm(r)=='k_schur']
i0=idx[len(idx)*2//3]; i1=idx[len(idx)*2//3+1]
t0=int(rows[i0]['Start_Timestamp'])
# start a little earlier: the sweep of this iteration
j=i0
while j>0 and int(rows[j]['Start_Timestamp'])>t0-120000: j-=1
out=[]
for r in rows[j:i1]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    out.append("%9.1f %9.1f %7.1f  q%-3s %s grid %s" % (s,e,e-s,r.get('Queue_Id','?'),nm(r),r.get('Grid_Size','?')))
open("$OUT/timeline.txt","w").write("\n".join(out)+"\n")
print("\n".join(out))
PY
rm -rf $OUT/tl_prof
