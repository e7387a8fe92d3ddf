"""CPU tests of the CSolverStereoPosit restatement (oracle/oracle_posit.c): recovers a known pose, the reference's
failure branches, and the pivoted LDLT against numpy.  PARITY UNPINNED (no reference fixtures)."""
import numpy as np

import posit_case
import track_scene as ts


def test_recovers_the_true_pose(oracle):
    c = posit_case.make(400, 1, noise=0.0, outliers=0.0)
    prm = oracle.posit_params(ts.P_LEFT, ts.P_RIGHT)
    r = oracle.stereo_posit(prm, c["T_last"], c["t_imu"], c["T_est"], c["xyz"], c["uvl"], c["uvr"])
    assert r["status"] == 0 and r["iterations"] < 20 and r["n"] == 400
    assert np.abs(r["T"] - c["T_true"]).max() < 5e-6           # float32 pixels limit the fit
    R = r["T"][:9].reshape(3, 3)
    assert np.abs(R.T @ R - np.eye(3)).max() < 1e-9            # the first-order re-orthogonalisation holds R on SO(3)
    c = posit_case.make(300, 2)                                # noisy pixels + 10 % gross outliers: down-weighted
    r = oracle.stereo_posit(prm, c["T_last"], c["t_imu"], c["T_est"], c["xyz"], c["uvl"], c["uvr"])
    assert r["status"] == 0 and np.abs(r["T"] - c["T_true"]).max() < 2e-3 and 200 < r["inliers"] < 300


def test_failure_branches(oracle):
    prm = oracle.posit_params(ts.P_LEFT, ts.P_RIGHT)
    c = posit_case.make(25, 3)
    r = oracle.stereo_posit(prm, c["T_last"], c["t_imu"], c["T_est"], c["xyz"], c["uvl"], c["uvr"])
    assert r["status"] == 1 and np.array_equal(r["T"], c["T_est"])              # needs MORE than 25 points (:19)
    c = posit_case.make(26, 3)
    assert oracle.stereo_posit(prm, c["T_last"], c["t_imu"], c["T_est"], c["xyz"], c["uvl"], c["uvr"])["status"] == 0
    active = np.ones(26, np.uint8); active[:3] = 0
    assert oracle.stereo_posit(prm, c["T_last"], c["t_imu"], c["T_est"], c["xyz"], c["uvl"], c["uvr"], active)["status"] == 1
    c = posit_case.make(300, 4)
    one = oracle.posit_params(ts.P_LEFT, ts.P_RIGHT, max_iterations=2)
    assert oracle.stereo_posit(one, c["T_last"], c["t_imu"], c["T_est"], c["xyz"], c["uvl"], c["uvr"])["status"] == 2
    # inconsistent with the prior: the IMU says we moved 3 m sideways
    r = oracle.stereo_posit(prm, c["T_last"], np.array([3.0, 0, 0]), c["T_est"], c["xyz"], c["uvl"], c["uvr"])
    assert r["status"] == 4 and r["risk"] > 2.0
    # garbage measurements: large average error and too few inliers
    g = posit_case.make(60, 5, outliers=1.0)
    r = oracle.stereo_posit(prm, g["T_last"], g["t_imu"], g["T_est"], g["xyz"], g["uvl"], g["uvr"])
    assert r["status"] in (2, 3)
    # tiny motion is not integrated (:137-141): translation snaps back to the last pose
    s = posit_case.make(300, 6, motion=(0.001, 0.0, 0.0, 0.005, 0.0, 0.01))
    r = oracle.stereo_posit(prm, s["T_last"], s["t_imu"], s["T_est"], s["xyz"], s["uvl"], s["uvr"])
    assert r["status"] == 0 and np.array_equal(r["T"][9:], s["T_last"][9:])


def test_gauss_newton_step_against_numpy(oracle):
    """one iteration from the estimate: dx solves (sum w J'J) dx = -sum w J'e with numerically differentiated J"""
    c = posit_case.make(120, 7, outliers=0.0)
    prm = oracle.posit_params(ts.P_LEFT, ts.P_RIGHT, max_iterations=1)
    r = oracle.stereo_posit(prm, c["T_last"], c["t_imu"], c["T_est"], c["xyz"], c["uvl"], c["uvr"])

    def residual(T):
        p = c["xyz"] @ T[:9].reshape(3, 3).T + T[9:]
        ok = p[:, 2] > 0
        hl = np.c_[p, np.ones(len(p))] @ ts.P_LEFT.T
        hr = np.c_[p, np.ones(len(p))] @ ts.P_RIGHT.T
        e = np.c_[hl[:, 0] / hl[:, 2] - c["uvl"][:, 0], hl[:, 1] / hl[:, 2] - c["uvl"][:, 1], hr[:, 0] / hr[:, 2] - c["uvr"][:, 0],
                  hr[:, 1] / hr[:, 2] - c["uvr"][:, 1]]
        e[~ok] = 0
        return e

    def oplus(T, d):  # fromVector(d) * T
        w = np.sqrt(1 - d[3:] @ d[3:])
        x, y, z = d[3:]
        dR = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                       [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        return np.concatenate([(dR @ T[:9].reshape(3, 3)).ravel(), dR @ T[9:] + d[:3]])

    e0 = residual(c["T_est"])
    J = np.zeros((len(e0), 4, 6))
    h = 1e-6
    for k in range(6):
        d = np.zeros(6); d[k] = h
        J[:, :, k] = (residual(oplus(c["T_est"], d)) - residual(oplus(c["T_est"], -d))) / (2 * h)
    e2 = (e0 ** 2).sum(1)
    w = np.where(e2 > 10, 10 / np.maximum(e2, 1e-300), 1.0)
    H = np.einsum("n,nik,nil->kl", w, J, J)
    b = np.einsum("n,nik,ni->k", w, J, e0)
    dx = np.linalg.solve(H, -b)
    T1 = oplus(c["T_est"], dx)
    # the reference's analytic Jacobian treats the rotation part as 2[p]x (a quaternion-vector step), so one step agrees
    # with the numerically differentiated Gauss-Newton step to first order
    assert np.abs(r["T"] - T1).max() < 5e-4
