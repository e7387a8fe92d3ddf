#!/bin/bash
# SQ counters of the Jacobian-sweep kernels (two passes, --kernel-trace only)
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_sweep; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --kernel-include-regex "k_linearize|k_backsub" --output-format csv -d $OUT/p1 -o p1 -- python3 $ROOT/tools/time_sweep.py 10 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA --kernel-trace --kernel-include-regex "k_linearize|k_backsub" --output-format csv -d $OUT/p2 -o p2 -- python3 $ROOT/tools/time_sweep.py 10 > $OUT/p2.log 2>&1
find $OUT -name '*counter_collection.csv' -exec cp {} $OUT/ \;
ls $OUT
