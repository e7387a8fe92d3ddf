"""Writes tests/golden/handwritten_vi.g2o: a small graph in g2o's text format with every tag the reference's graphs
contain (Cg2oOptimizer.cpp:495-497 saves with g2o::SparseOptimizer::save), including the reference's own
EDGE_SE3_LINEAR_ACCELERATION (edge_se3_linear_acceleration.cpp:35-103: parameter id, 3 measurement values, upper triangle
of the information) with a != 0 and a non-identity PARAMS_SE3OFFSET 3.

Independent of the product AND of the oracle on purpose: plain numpy and string formatting, no writer of either is
involved; tests/test_g2o_fixture*.py parse the file with their own few lines and build the oracle's graph from the
literals through its API.  Values are rounded to a few decimals so the file reads like the hand-written fixture it is."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def quat_R(x, y, z, w):
    n = np.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def main():
    r = np.random.default_rng(20260505)
    f, cx, cy, fb = 450.5097, 375.9432, 222.338, 49.6325     # vi_sensor rectified K, f * baseline
    out = []
    out.append("PARAMS_SE3OFFSET 0 0 0 0 0 0 0 1")
    out.append("PARAMS_CAMERACALIB 1 0 0 0 0 0 0 1 %.4f %.4f %.4f %.4f" % (f, f, cx, cy))
    out.append("PARAMS_CAMERACALIB 2 0 0 0 0 0 0 1 %.4f %.4f %.4f %.4f" % (f, f, cx, cy))
    # IMU -> LEFT: mostly a half turn about the optical axis, as on the vi_sensor, plus a visible tilt
    q_off = np.array([0.02, -0.035, 0.9985, 0.04])
    q_off /= np.linalg.norm(q_off)
    q_off = np.round(q_off, 6)
    out.append("PARAMS_SE3OFFSET 3 0.0665 0.0053 0.0103 %.6f %.6f %.6f %.6f" % tuple(q_off))
    R_off = quat_R(*q_off)
    # four poses (LEFT -> WORLD), the first fixed
    n_p = 4
    pose_q = np.round(np.array([[0, 0, 0, 1], [0.01, 0.05, -0.008, 0.9987], [-0.015, 0.11, 0.012, 0.9937],
                                [0.02, 0.16, -0.01, 0.9869]]), 6)
    pose_t = np.array([[0, 0, 0], [0.05, -0.01, 0.7], [0.16, 0.02, 1.38], [0.33, 0.01, 2.02]])
    pose_R = [quat_R(*q) for q in pose_q]
    # estimates = truth + a perturbation, written with few decimals
    est_q = pose_q.copy()
    est_t = pose_t.copy()
    est_q[1:, :3] += np.round(r.normal(0, 0.004, (3, 3)), 4)
    est_t[1:] += np.round(r.normal(0, 0.03, (3, 3)), 3)
    lm_true = np.stack([r.uniform(-1.5, 1.5, 12), r.uniform(-0.8, 0.8, 12), r.uniform(2.6, 9.0, 12)], 1)
    lm_true[0, 2], lm_true[1, 2], lm_true[2, 2] = 2.7, 2.9, 3.3      # close enough for EDGE_SE3_TRACKXYZ from the first poses
    lm_est = np.round(lm_true + r.normal(0, 0.04, lm_true.shape), 3)
    for l in range(12):
        out.append("VERTEX_TRACKXYZ %d %.3f %.3f %.3f" % (l, *lm_est[l]))
    out.append("VERTEX_TRACKXYZ 500 %.3f %.3f %.3f" % tuple(np.round(lm_true[3] + [0.02, -0.01, 0.03], 3)))
    for k in range(n_p):
        q = est_q[k] / np.linalg.norm(est_q[k])
        out.append("VERTEX_SE3:QUAT %d %.3f %.3f %.3f %.6f %.6f %.6f %.6f" % (1000000 + k, *est_t[k], *q))
    out.append("FIX 1000000 500")
    # gravity edges: a = (R_k R_off)' (0,0,-1) + noise, normalised; information not the identity for one of them
    for k in range(n_p):
        a = (pose_R[k] @ R_off).T @ np.array([0, 0, -1.0]) + r.normal(0, 0.02, 3)
        a = np.round(a / np.linalg.norm(a), 5)
        info = "1 0 0 1 0 1" if k != 2 else "2.5 0.1 0 1.5 -0.2 3"
        out.append("EDGE_SE3_LINEAR_ACCELERATION %d 3 %.5f %.5f %.5f %s" % (1000000 + k, *a, info))
    # odometry: measured relative pose = estimate-relative (the reference's rule), 1e5 * diag(s,s,s,1,1,1)
    for k in range(1, n_p):
        Ri, Rj = quat_R(*est_q[k - 1]), quat_R(*est_q[k])
        Rz = Ri.T @ Rj
        tz = Ri.T @ (est_t[k] - est_t[k - 1])
        # rotation -> quaternion (w >= 0)
        w = np.sqrt(max(0.0, 1 + np.trace(Rz))) / 2
        qz = np.array([(Rz[2, 1] - Rz[1, 2]) / (4 * w), (Rz[0, 2] - Rz[2, 0]) / (4 * w), (Rz[1, 0] - Rz[0, 1]) / (4 * w), w])
        s = 1e5 / (1 + tz @ tz)
        info = []
        for a_ in range(6):
            for b_ in range(a_, 6):
                info.append(("%.3f" % (s if a_ < 3 else 1e5)) if a_ == b_ else "0")
        out.append("EDGE_SE3:QUAT %d %d %.4f %.4f %.4f %.6f %.6f %.6f %.6f %s" % (1000000 + k - 1, 1000000 + k, *np.round(tz, 4),
                                                                          *np.round(qz, 6), " ".join(info)))
    # projection edges by the reference's type rule on |p|^2 (10 / 50), information from w = 1/z
    for l in range(12):
        for k in range(n_p):
            if (l + k) % 5 == 4:
                continue                                    # not every landmark in every key frame
            p = pose_R[k].T @ (lm_true[l] - pose_t[k])
            if p[2] < 0.5:
                continue
            u = f * p[0] / p[2] + cx + r.normal(0, 0.4)
            v = f * p[1] / p[2] + cy + r.normal(0, 0.4)
            d = np.rint(fb / p[2] + r.normal(0, 0.3))
            z = fb / d
            pm = np.array([z * (u - cx) / f, z * (v - cy) / f, z])
            wgt = 1.0 / z
            l2 = pm @ pm
            if l2 < 10:
                out.append("EDGE_SE3_TRACKXYZ %d %d 0 %.4f %.4f %.4f %.3f 0 0 %.3f 0 %.3f" % (1000000 + k, l, *pm, 1000 * wgt, 1000 * wgt,
                                                                                      1000 * wgt))
            elif l2 < 50:
                out.append("EDGE_PROJECT_DEPTH %d %d 1 %.3f %.3f %.4f %.5f 0 0 %.5f 0 %.4f" % (1000000 + k, l, u, v, z, wgt, wgt, 100 * wgt))
            else:
                out.append("EDGE_PROJECT_DISPARITY %d %d 1 %.3f %.3f %.6f %.5f 0 0 %.5f 0 %.3f" % (1000000 + k, l, u, v, d / fb, wgt, wgt,
                                                                                           1000 * wgt))
    # one full (non-diagonal) information matrix, as a .g2o file may carry
    out.append("EDGE_SE3_TRACKXYZ 1000003 1 0 -1.85 -0.21 0.36 40 3 -2 35 1.5 50")
    # landmark closure against the fixed landmark 500 (Cg2oOptimizer.cpp:448-458)
    out.append("EDGE_POINTXYZ 500 3 0 0 0 1000 0 0 1000 0 1000")
    with open(os.path.join(HERE, "handwritten_vi.g2o"), "w") as fo:
        fo.write("\n".join(out) + "\n")
    print("wrote", len(out), "lines")


if __name__ == "__main__":
    main()
