#!/bin/bash
# Host-side AddressSanitizer pass over the C-ABI's host code (ba_host.cpp: structure analysis, work lists, LM driver;
# ba_g2o_io.cpp; capi_common.cpp).  Device code is not instrumented (GPU ASAN is not available on this pool).
#   1. here (no GPU needed):   bash tools/asan_host.sh build
#   2. on the GPU box:         gpurun -- 'bash tools/asan_host.sh run'     (BA parity tests + the random-graph sweep;
#      tests that initialise torch.cuda cannot run under the preloaded ASAN runtime: its dlopen hook breaks torch's own)
#   3. here:                   bash tools/asan_host.sh restore             (the normal library again - do not skip)
set -e
cd "$(dirname "$0")/../svi_mapper_amd/csrc"
RT=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.asan-x86_64.so)
HOST_OBJS="../lib/obj/ba_host.cpp.o ../lib/obj/ba_g2o_io.cpp.o ../lib/obj/capi_common.cpp.o"
case "$1" in
build)
    touch ba_host.cpp ba_g2o_io.cpp capi_common.cpp
    make EXTRA="-Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -g" $(for o in $HOST_OBJS; do realpath -m $o; done)
    make ;;
run)
    cd ../..
    mkdir -p gpurun_out
    export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0
    LD_PRELOAD=$RT python3 -m pytest tests/test_ba_gpu.py -m gpu -q 2>&1 | tee gpurun_out/asan_tests.log | tail -3
    LD_PRELOAD=$RT python3 tests/fuzz_ba_gpu.py 2000 80 > gpurun_out/asan_fuzz.log 2>&1 || true
    grep -c "ERROR: AddressSanitizer: [a-z-]*-" gpurun_out/asan_tests.log gpurun_out/asan_fuzz.log || true
    tail -1 gpurun_out/asan_fuzz.log ;;
restore)
    touch ba_host.cpp ba_g2o_io.cpp capi_common.cpp
    make
    if nm -D ../lib/libsvi_hot.so | grep -q __asan; then echo "still instrumented"; exit 1; fi ;;
*) echo "usage: $0 build|run|restore"; exit 2 ;;
esac
