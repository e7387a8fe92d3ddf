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


REPLAY = r'''
// A maintainer-side program: plain C++, the stock HIP runtime, no Python.  Replays a saved graph through the facade
// (what Cg2oOptimizer::optimize does after the graph is built) and writes the optimised graph back.
#include "svi_g2o_optimizer.hpp"
#include <cstdio>
int main(int argc, char** argv) {
    if (argc < 3) return 2;
    try {
        svi::BundleAdjusterGPU ba(718.856, 718.856, 607.1928, 185.2157, 0.54);
        ba.load(argv[1]);
        uint64_t executed = 0;
        const uint64_t nominal = ba.optimizeUnLimited(&executed);
        ba.save(argv[2]);
        std::printf("%llu %llu %.17g\n", (unsigned long long)nominal, (unsigned long long)executed, ba.chi2());
    } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
    return 0;
}
'''


import pytest  # noqa: E402


@pytest.mark.gpu
def test_cpp_program_replays_a_graph(svi, tmp_path):
    """g++-built program against libsvi_hot.so + /opt/rocm libamdhip64: same result as the Python-driven library"""
    import numpy as np
    from svi_mapper_amd import _capi, synth
    prob = synth.make_ba_problem(12, 300, 2200, seed=7)
    cam = prob["cam"]
    a = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    synth.build_ba_graph(a, prob)
    f_in, f_out = tmp_path / "in.g2o", tmp_path / "out.g2o"
    a.save_g2o(f_in)
    libdir = os.path.dirname(_capi.LIB_PATH)
    hip = "/opt/rocm/lib"
    c = tmp_path / "replay.cpp"
    c.write_text(REPLAY)
    exe = tmp_path / "replay"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe), "-L", libdir, "-lsvi_hot",
                           "-L", hip, "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath," + hip])
    out = subprocess.check_output([str(exe), str(f_in), str(f_out)]).decode().split()
    # the same graph through the Python harness (loaded from the same text file: identical inputs)
    b = svi.BundleAdjuster(1, 1, 0, 0, cam["baseline_m"])
    b.load_g2o(f_in)
    b.initialize()
    nominal, executed = b.optimize_until()
    assert (int(out[0]), int(out[1])) == (nominal, executed)
    assert abs(float(out[2]) - b.chi2()[0]) <= 1e-9 * b.chi2()[0]
    r = svi.BundleAdjuster(1, 1, 0, 0, cam["baseline_m"])
    r.load_g2o(f_out)
    assert np.abs(r.get_landmarks()[1] - b.get_landmarks()[1]).max() < 1e-6
