#!/bin/bash
# Runs on the GPU box: the judged numbers of a round.  Output under gpurun_out/final/ (copy into profiles/ afterwards).
#   bench.json            python bench.py (default flags)
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command (BA part only: no cpu baseline)
#   pmc/                  FETCH_SIZE / WRITE_SIZE passes (tools/pmc_passes.sh)
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/final; mkdir -p $OUT
python3 bench.py > $OUT/bench.log 2>&1
grep '^{"metric"' $OUT/bench.log > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r -- python3 $ROOT/bench.py --no-cpu-baseline --no-matcher --no-frontend > $OUT/prof.log 2>&1
cp $OUT/prof/r_kernel_stats.csv $OUT/kernel_stats.csv
grep '^{"metric"' $OUT/prof.log > $OUT/bench_under_rocprof.json
cd $ROOT && bash tools/pmc_passes.sh > $OUT/pmc.log 2>&1
echo done
