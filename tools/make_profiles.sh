#!/bin/bash
# Runs on the GPU box: the judged numbers of a round.  Output under gpurun_out/final/ (copy into profiles/ afterwards, see
# profiles/README.md).
#   bench.json                      python bench.py (default flags)
#   kernel_stats_inloop.csv         rocprofv3 --kernel-trace --stats of bench.py --no-cpu-baseline --no-replay: every launch of the
#                                   Jacobian sweep in it sits INSIDE an LM loop (no back-to-back replays), and the matcher, the
#                                   tracking cascades, BRIEF, StereoPosit, the landmark refinement and the config-5 stream are in it
#   kernel_stats_inloop_c4.csv      the same restricted to CONFIG 4 (bench.py --no-matcher --no-frontend --no-replay --no-cpu-baseline: no
#                                   config 3, no config 5): the average of k_linearize_lm + k_linearize_pose in it, divided into the
#                                   algorithmic bytes of the sweep, is roofline.frac of the bench line under the profiler
#   kernel_stats_replay.csv         the same command with the sweep replays (roofline.frac_replay / frac_cold come from those)
#   pmc/                            FETCH_SIZE / WRITE_SIZE passes (tools/pmc_passes.sh)
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/final; mkdir -p $OUT
python3 bench.py > $OUT/bench.log 2>&1
grep '^{"metric"' $OUT/bench.log > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_inloop -o r -- python3 $ROOT/bench.py --no-cpu-baseline --no-replay > $OUT/prof_inloop.log 2>&1
cp $OUT/prof_inloop/r_kernel_stats.csv $OUT/kernel_stats_inloop.csv
grep '^{"metric"' $OUT/prof_inloop.log > $OUT/bench_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c4 -o r -- python3 $ROOT/bench.py --no-cpu-baseline --no-matcher --no-frontend --no-replay > $OUT/prof_c4.log 2>&1
cp $OUT/prof_c4/r_kernel_stats.csv $OUT/kernel_stats_inloop_c4.csv
grep '^{"metric"' $OUT/prof_c4.log > $OUT/bench_under_rocprof_c4.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_replay -o r -- python3 $ROOT/bench.py --no-cpu-baseline --no-matcher --no-frontend > $OUT/prof_replay.log 2>&1
cp $OUT/prof_replay/r_kernel_stats.csv $OUT/kernel_stats_replay.csv
rm -rf $OUT/prof_inloop $OUT/prof_replay $OUT/prof_c4
cd $ROOT && bash tools/pmc_passes.sh > $OUT/pmc.log 2>&1
echo done
