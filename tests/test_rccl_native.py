"""The native all-reduce hook (csrc/rccl_hook.cpp: ncclAllReduce on the handle's stream) under a torchless C++ multi-process
driver (tests/rccl_driver.cpp): one process per GPU, no Python on the data path.  RCCL refuses two ranks on one GPU, so on a
one-GPU box the communicator has one rank (every sum is the identity: the run must equal the run without a communicator); on a
node with more GPUs the same test shards the graph over two of them and compares with the unsharded solve."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _build(tmp_path):
    from svi_mapper_amd import _capi
    libdir = os.path.dirname(_capi.LIB_PATH)
    hip = "/opt/rocm/lib"
    exe = tmp_path / "rccl_driver"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "rccl_driver.cpp"), "-o", str(exe),
                           "-L", libdir, "-lsvi_hot", "-L", hip, "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath," + hip])
    return str(exe)


def _run(exe, n_ranks, tmp_path, tag, shards=0):
    out = subprocess.check_output([exe, str(n_ranks), "40", "3000", str(tmp_path / ("id_" + tag))] + ([str(shards)] if shards else []), timeout=600).decode()
    rows = [l.split() for l in out.splitlines() if l.startswith("rank ")]
    return [(int(r[2]), int(r[3]), float(r[4]), float(r[5]), float(r[6]), int(r[8])) for r in rows]


def test_native_hook_under_a_cpp_driver(svi, tmp_path):
    import torch
    exe = _build(tmp_path)
    plain = _run(exe, 0, tmp_path, "a")[0]
    one = _run(exe, 1, tmp_path, "b")[0]
    assert plain[0] == one[0] and plain[1] == one[1] and plain[1] >= 3
    assert abs(plain[2] - one[2]) <= 1e-9 * plain[2] and abs(plain[4] - one[4]) <= 1e-9 * abs(plain[4])
    assert one[5] == 3000
    # shard 0 of 2 through the 1-rank communicator: every collective of the sharded path goes through ncclAllReduce
    half = _run(exe, 1, tmp_path, "s", shards=2)[0]
    assert 0 < half[5] < 3000 and half[1] >= 1 and half[2] > 0 and half[2] == half[2]
    if torch.cuda.device_count() >= 2:     # a real two-GPU exchange over xGMI
        two = _run(exe, 2, tmp_path, "c")
        assert len(two) == 2 and sum(r[5] for r in two) == 3000 and min(r[5] for r in two) > 0
        for r in two:
            assert r[0] == plain[0] and r[1] == plain[1]
            assert abs(r[2] - plain[2]) <= 1e-6 * plain[2] and abs(r[4] - plain[4]) <= 1e-6 * abs(plain[4])
        assert two[0][2:5] == two[1][2:5]   # bit-identical between the ranks


def test_native_hook_from_python(svi):
    """the same hook installed by the Python harness (what bench.py --hook native does), 1-rank communicator"""
    import numpy as np
    from svi_mapper_amd import dist as sdist
    from svi_mapper_amd import synth
    prob = synth.make_ba_problem(20, 1000, 7000, seed=11)
    cam = prob["cam"]
    ref = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    synth.build_ba_graph(ref, prob)
    ref.initialize()
    want = [ref.optimize(n) for n in (1, 6)]
    comm = sdist.NativeRccl(0, 1, 0)
    ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], rank=0, n_ranks=1)
    synth.build_ba_graph(ba, prob)
    ba.set_allreduce_native(comm)
    ba.initialize()
    assert [ba.optimize(n) for n in (1, 6)] == want
    assert np.array_equal(ba.get_poses()[1], ref.get_poses()[1])
    # a shard of two with the native 1-rank communicator: the collective path (4 scalars, reduced system, landmark gather) runs
    sh = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], rank=0, n_ranks=2)
    synth.build_ba_graph(sh, prob)
    sh.set_allreduce_native(comm)
    sh.initialize()
    assert sh.optimize(3) == 3 and np.isfinite(sh.chi2()[0])
    sh.get_landmarks()
    sh.close()
    ba.close()
    comm.close()
