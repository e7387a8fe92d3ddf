"""AddressSanitizer + UBSan + leak check over the host side of the BA C ABI (graph store, the reference's construction rules,
build_structure for 1-3 ranks / both tile sizes / both elimination orders, .g2o writer and reader, pruning, write-back) -
g++ build with a malloc-backed HIP stand-in and no-op kernels, no GPU runtime in the process (tools/host_san/Makefile says
why).  Any sanitizer report aborts the program (-fno-sanitize-recover=all): a non-zero exit code fails the test, nothing
is filtered."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_ubsan():
    out = os.path.join(tempfile.gettempdir(), "svi_host_san_%d" % os.getuid())
    r = subprocess.run(["make", "-s", "-j4", "-C", os.path.join(ROOT, "tools", "host_san"), "OUT=" + out, "run"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "host_san: ok" in r.stdout and "ERROR: AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout
