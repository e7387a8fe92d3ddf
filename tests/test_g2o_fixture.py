"""tests/golden/handwritten_vi.g2o (written by tests/golden/make_handwritten_g2o.py without the product's or the oracle's
code): what the oracle makes of it, checked against numpy on the literals of the file.  The GPU side is
tests/test_g2o_fixture_gpu.py."""
import os

import numpy as np

import g2o_text

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "handwritten_vi.g2o")


def test_fixture_holds_every_tag_of_the_reference():
    g = g2o_text.read(FIXTURE)
    assert len(g["poses"]) == 4 and len(g["lms"]) == 13 and g["fixed"] == [1000000, 500]
    assert len(g["se3"]) == 3 and len(g["accel"]) == 4 and len(g["lmlm"]) == 1
    assert {t for t, *_ in g["proj"]} == {0, 1, 2}
    off = g["offsets"][3]
    assert np.abs(off[:9].reshape(3, 3) - np.eye(3)).max() > 0.5          # a real rotation, not the identity
    assert all(pid == 3 and abs(np.linalg.norm(a) - 1) < 1e-4 and abs(a[2] + 1) > 1e-3 for _, pid, a, _ in g["accel"])
    assert any(info[1] != 0 for *_, info in g["accel"]) and any(info[1] != 0 for *_, info in g["proj"])


def test_oracle_on_the_fixture_matches_numpy(oracle):
    g = g2o_text.read(FIXTURE)
    fx, fy, cx, cy = g["cams"][1]
    o = g2o_text.build(oracle.OracleBA(fx, fy, cx, cy, 49.6325 / fx), g)
    o.initialize()
    poses = dict(g["poses"])
    lms = dict(g["lms"])
    # gravity edges: e = R R_off a - (0,0,-1)  (edge_se3_linear_acceleration.cpp:106-116)
    ae = o.aux_jacobians()[3]
    R_off = g["offsets"][3][:9].reshape(3, 3)
    chi = 0.0
    for k, (p, _, a, info) in enumerate(g["accel"]):
        e = poses[p][:9].reshape(3, 3) @ R_off @ a + np.array([0, 0, 1.0])
        np.testing.assert_allclose(ae[k], e, atol=1e-14)
        O = np.array([[info[0], info[1], info[2]], [info[1], info[3], info[4]], [info[2], info[4], info[5]]])
        chi += e @ O @ e
    # projection edges: Z = R'(p - t); XYZ: Z - z; depth: (u, v, Z_z) - z; disparity: (u, v, 1/Z_z) - z
    eo = o.edge_jacobians()[0]
    for k, (t, p, l, _, z, info) in enumerate(g["proj"]):
        T = poses[p]
        Z = T[:9].reshape(3, 3).T @ (lms[l] - T[9:])
        u, v = fx * Z[0] / Z[2] + cx, fy * Z[1] / Z[2] + cy
        e = (Z, np.array([u, v, Z[2]]), np.array([u, v, 1.0 / Z[2]]))[t] - z
        np.testing.assert_allclose(eo[k], e, rtol=1e-12, atol=1e-12)
        O = np.array([[info[0], info[1], info[2]], [info[1], info[3], info[4]], [info[2], info[4], info[5]]])
        chi += e @ O @ e
    for i, j, z, info in g["lmlm"]:
        e = lms[j] - lms[i] - z
        chi += e @ (np.diag(info[[0, 3, 5]])) @ e
    # odometry edges are written with zero initial error up to the rounding of the literals: a small remainder
    plain, robust = o.chi2()
    assert abs(plain - chi) < 1e-3 * chi + 0.5
    assert robust < plain
