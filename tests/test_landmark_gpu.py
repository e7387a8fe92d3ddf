"""GPU parity of the batched landmark refinement (svi_mapper_amd/csrc/landmark.hip) against oracle/oracle_landmark.c:
one thread per landmark walks its measurements in the reference's order, so positions, statuses, error averages and
iteration counts are BIT-identical.  PARITY UNPINNED with respect to the reference itself."""
import numpy as np
import pytest

import landmark_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("the -m gpu tests need a visible GPU (torch.cuda.is_available() is False)")
    return torch


@pytest.fixture(scope="module")
def opt(svi, torch):
    from svi_mapper_amd import temporal
    return temporal.LandmarkOptimizer()


def run(opt, torch, c):
    d = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")  # noqa: E731
    out = opt.optimize(d(c["PL"]), d(c["PR"]), d(c["seg"]), d(c["frame"]), d(c["uvl"]), d(c["uvr"]), d(c["xyz0"]))
    return [t.cpu().numpy() for t in out]


@pytest.mark.parametrize("n,seed,noise", [(1, 3, 0.3), (65, 4, 0.0), (3000, 5, 0.3), (20000, 6, 0.6)])
def test_bit_exact(oracle, torch, opt, n, seed, noise):
    c = landmark_case.make(n, seed, noise=noise)
    want = oracle.landmarks_optimize(oracle.landmark_params(), c["PL"], c["PR"], c["seg"], c["frame"], c["uvl"], c["uvr"], c["xyz0"])
    got = run(opt, torch, c)
    for a, b, name in zip(got, want, ("xyz", "status", "error_average", "iterations")):
        assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8)), name
    if n >= 3000:
        assert set(want[1]) >= {0, 1, 3}


def test_parameters_and_empty(oracle, torch, opt, svi):
    c = landmark_case.make(500, 7)
    for kw in (dict(cap_iterations=3), dict(min_measurements=0, kernel_max_error_l2=2.0), dict(min_inlier_ratio=0.95, max_error_average_l2=0.5)):
        prm = oracle.landmark_params(**kw)
        for k, v in kw.items():
            setattr(opt.params, k, v)
        want = oracle.landmarks_optimize(prm, c["PL"], c["PR"], c["seg"], c["frame"], c["uvl"], c["uvr"], c["xyz0"])
        got = run(opt, torch, c)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        opt._lib.svi_landmark_params_default(__import__("ctypes").byref(opt.params))
    assert 4 in oracle.landmarks_optimize(oracle.landmark_params(cap_iterations=3), c["PL"], c["PR"], c["seg"], c["frame"], c["uvl"], c["uvr"],
                                          c["xyz0"])[1]
    e = dict(PL=c["PL"], PR=c["PR"], seg=np.zeros(1, np.int32), frame=np.zeros(0, np.int32), uvl=np.zeros((0, 2), np.float32),
             uvr=np.zeros((0, 2), np.float32), xyz0=np.zeros((0, 3)))
    assert run(opt, torch, e)[0].shape == (0, 3)
    lib = svi.load_library()
    assert lib.svi_landmarks_optimize_dev(None, None, None, None, 0, None, None, None, None, None, 0, None, None, None, None) == 1
