"""The C++ facade header compiles against the C ABI with a plain g++ and links to the library."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "svi_g2o_optimizer.hpp"
#include <cstdio>
int main() {
    std::printf("%d %s\n", svi_version(), svi_status_string(0));
    try { svi::BundleAdjusterGPU ba(718.856, 718.856, 607.1928, 185.2157, 0.54); }
    catch (const std::exception& e) { std::printf("no device: %s\n", e.what()); }
    return 0;
}
'''


def test_facade_compiles_and_links(svi):
    from svi_mapper_amd import _capi
    libdir = os.path.dirname(_capi.LIB_PATH)
    hip = "/opt/rocm/lib"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.cpp")
        open(c, "w").write(SRC)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), c, "-o", exe, "-L", libdir, "-lsvi_hot",
                               "-L", hip, "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath," + hip])
        out = subprocess.check_output([exe]).decode()
    assert out.startswith("100 ok")
