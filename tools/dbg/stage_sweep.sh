#!/bin/bash
# iterations/s of the overlapped staged trial at config 4 for a few stage boundary sets / reserved CUs (tools/dbg/stage_bench.py)
OUT=gpurun_out/stage_sweep.log; : > $OUT
for st in "4" "2,4" "4,8" "2,4,8"; do
  for rs in 1 2; do
    echo "== stages $st reserve_per_se $rs" >> $OUT
    SVI_SCHUR_STAGES=$st SVI_SCHUR_RESERVE_PER_SE=$rs timeout -k 10 120 python tools/dbg/stage_bench.py 2>&1 | grep "it/s\|failures" | tail -2 >> $OUT || exit 1
  done
done
cat $OUT
