import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# a fresh checkout has no built library yet (built artefacts are not tracked): same step as __graft_entry__.build(),
# done here because several test modules import the package - which loads the library - at collection time
if not os.path.exists(os.path.join(ROOT, "svi_mapper_amd", "lib", "libsvi_hot.so")):
    import subprocess
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "svi_mapper_amd", "csrc")])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.load()
    return orc


@pytest.fixture(scope="session")
def svi():
    import svi_mapper_amd
    return svi_mapper_amd
