#!/bin/bash
# Host-side sanitizer pass (AddressSanitizer + UBSan + leak check) over the C ABI's host code: ba_host.cpp (graph store,
# construction rules, structure analysis, work lists), ba_g2o_io.cpp, capi_common.cpp.  Needs no GPU: see
# tools/host_san/Makefile for the set-up and for what went wrong with round 1's in-library variant (the ROCm ASan runtime's
# amdgpu allocator hook faulting in HSA's teardown at exit()).  Exit code = the program's: a sanitizer report is a failure.
set -e
cd "$(dirname "$0")/host_san"
make -s -j4 run "$@"
