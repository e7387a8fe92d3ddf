"""The C++ adapter headers go through a compiler: include/svi_cv_matcher.hpp against minimal stand-in declarations of the
OpenCV names it uses (tests/stubs/opencv2 - OpenCV is not in this image; a syntax / override-signature check that pins no
behaviour), svi_g2o_optimizer.hpp and svi_fundamental_matcher.hpp as they are.  The program also exercises the call shape of
the reference, m_pMatcher->match( query, pool, matches ) (CTriangulator.cpp:93), which OpenCV implements as
clone( true ) -> add -> match: the clone must share the GPU handle, not create one per call."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "svi_cv_matcher.hpp"
#include "svi_g2o_optimizer.hpp"
#include "svi_fundamental_matcher.hpp"
#include <cstdio>
#include <type_traits>
static_assert(std::is_base_of<cv::DescriptorMatcher, svi::HammingMatcherGPU>::value, "plug-in shape");
static_assert(!std::is_abstract<svi::HammingMatcherGPU>::value, "every pure virtual overridden");
int main() {
    try {
        std::shared_ptr<cv::DescriptorMatcher> m = std::make_shared<svi::HammingMatcherGPU>(0);   // CTriangulator.cpp:12
        cv::Mat q, pool;
        std::vector<cv::DMatch> matches;
        for (int i = 0; i < 3; ++i) m->match(q, pool, matches);                                     // CTriangulator.cpp:93
        cv::Ptr<cv::DescriptorMatcher> c = m->clone(true);
        std::printf("cloned %d\n", c->empty() ? 1 : 0);
    } catch (const std::exception& e) { std::printf("no device: %s\n", e.what()); }
    return 0;
}
'''


def test_adapter_headers_compile(svi):
    from svi_mapper_amd import _capi
    libdir = os.path.dirname(_capi.LIB_PATH)
    hip = "/opt/rocm/lib"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "a.cpp")
        open(c, "w").write(SRC)
        exe = os.path.join(d, "a")
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror=overloaded-virtual", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
                               "-I", os.path.join(ROOT, "tests", "stubs"), "-I", os.path.join(ROOT, "include"), c, "-o", exe, "-L", libdir,
                               "-lsvi_hot", "-L", hip, "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath," + hip])
        out = subprocess.check_output([exe]).decode()
    # without a GPU the constructor throws (the library has no CPU path); with one the clone shares the handle
    assert out.startswith("no device") or out.startswith("cloned 1")
