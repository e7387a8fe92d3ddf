#!/bin/bash
# here: builds svi_mapper_amd/lib_ts (the library with -DPOTRF_TS);  on the GPU box: tools/dbg/potrf_ts.sh run
if [ "$1" = run ]; then
  SVI_HOT_LIB=$(pwd)/svi_mapper_amd/lib_ts/libsvi_hot.so python3 bench.py --no-cpu-baseline --no-matcher --no-frontend --no-replay > gpurun_out/ts.log 2>&1
  grep "^TS" gpurun_out/ts.log
else
  cd svi_mapper_amd && rm -rf lib_ts && cp -r lib lib_ts && rm -f lib_ts/obj/ba_chol.hip.o && cd csrc && make -s OUTDIR=$(pwd)/../lib_ts EXTRA=-DPOTRF_TS 2>&1 | grep -i " error"
fi
