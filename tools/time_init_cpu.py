"""Times svi_ba_initialize (the host-side structure analysis of ba_host.cpp) WITHOUT a GPU: the host sources built with g++
-O3 against the malloc-backed HIP stand-in of tools/host_san (kernels are no-ops).  Feeds the cached config-4 (or config-3)
graph through the C ABI and prints the section times of build_structure (SVI_DEBUG_PLAN=1).
    python tools/time_init_cpu.py [c4|c3] [reps]"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = "/tmp/svi_stub"


def build():
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, "libsvi_stub.so")
    src = [os.path.join(ROOT, "svi_mapper_amd", "csrc", f) for f in ("ba_host.cpp", "ba_structure.cpp", "ba_g2o_io.cpp", "capi_common.cpp")] + \
          [os.path.join(ROOT, "tools", "host_san", f) for f in ("hip_host_stub.cpp", "kernel_stubs.cpp")]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in src + [os.path.join(ROOT, "svi_mapper_amd", "csrc", "ba_host.h")]):
        subprocess.check_call(["g++", "-std=c++17", "-O3", "-march=native", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "svi_mapper_amd", "csrc")] + src + ["-o", lib])
    return lib


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c4"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = C.CDLL(build())
    from svi_mapper_amd import _capi, synth
    from svi_mapper_amd.optimizer import BundleAdjuster
    for name, (res, args) in _capi.SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    _capi._lib = lib       # the harness classes bind to the stub library in this process
    if which == "c4":
        import bench
        prob = bench.cached_problem(1)
    else:
        prob = synth.make_c3()
    cam = prob["cam"]
    ba = BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    t0 = time.perf_counter()
    synth.build_ba_graph(ba, prob)
    print("graph construction through the C ABI: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
    for r in range(reps):
        if r > 0:     # an edit of the graph: the whole structure analysis runs again (an unchanged graph only re-uploads the estimates)
            ba.add_landmark(900000 + r, [0.0, 0.0, 5.0])
        t0 = time.perf_counter()
        ba.initialize()
        print("svi_ba_initialize: %.2f ms" % (1e3 * (time.perf_counter() - t0)))
    st = ba.stats()
    print("levels", st.chol_steps, "tiles", st.chol_tiles_nnz, "jobs", st.n_schur_tiles)


if __name__ == "__main__":
    main()
