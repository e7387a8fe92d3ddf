#!/bin/bash
# A/B timing of library builds on the GPU box: tools/dbg/ab.sh lib_v1 lib_v2 ...  (BA kernel stats of bench.py per build)
for v in "$@"; do
  export SVI_HOT_LIB=$(pwd)/svi_mapper_amd/$v/libsvi_hot.so
  bash tools/dbg/kstats.sh > gpurun_out/ks_$v.txt 2>&1
  cp gpurun_out/ks.log gpurun_out/ks_$v.log
done
