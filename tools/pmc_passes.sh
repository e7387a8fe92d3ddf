#!/bin/bash
# Runs on the GPU box (through gpurun): HBM traffic counters for the hot kernels, one counter per pass,
# --kernel-trace only (no sys/runtime/hip trace together with --pmc).  Output: gpurun_out/pmc/*.csv
# The bench run is restricted to CONFIG 4 (--no-matcher also drops config 3, --no-frontend config 5): every launch counted is one of
# the headline workload.
# Usage: bash tools/pmc_passes.sh   (from the repo root)
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
[ -x tools/pmc_calib ] || /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/pmc_calib.hip -o tools/pmc_calib
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/calib_$C -o calib_$C -- $ROOT/tools/pmc_calib > $OUT/calib_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/bench_$C -o bench_$C -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-matcher --no-frontend --no-replay > $OUT/bench_$C.log 2>&1
  find $OUT/calib_$C $OUT/bench_$C -name '*counter_collection.csv' -exec cp {} $OUT/ \;
  echo "pass $C done"
done
ls -la $OUT/*.csv
