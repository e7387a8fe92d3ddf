"""CPU tests of the CLandmark::optimize restatement (oracle/oracle_landmark.c).  PARITY UNPINNED."""
import numpy as np

import landmark_case


def test_refinement_improves_and_classifies(oracle):
    c = landmark_case.make(600, 1)
    prm = oracle.landmark_params()
    out, st, err, its = oracle.landmarks_optimize(prm, c["PL"], c["PR"], c["seg"], c["frame"], c["uvl"], c["uvr"], c["xyz0"])
    cnt = np.diff(c["seg"])
    assert np.all(st[cnt <= 5] == 0) and np.all(st[cnt > 5] != 0)                 # refined iff MORE than 5 measurements
    assert np.array_equal(out[st == 0], c["xyz0"][st == 0]) and np.all(its[st == 0] == 0)
    assert set(st) >= {0, 1, 3} and (st == 1).sum() > 300
    kept = (st == 1) | (st == 2)
    assert np.array_equal(out[~kept], c["xyz0"][~kept])                           # failures keep the initial guess
    e0 = np.linalg.norm(c["xyz0"] - c["xyz_true"], axis=1)
    e1 = np.linalg.norm(out - c["xyz_true"], axis=1)
    good = st == 1
    assert np.median(e1[good]) < 0.35 * np.median(e0[good])
    assert np.all(err[st == 1] < 9.0) and np.all(err[st == 2] >= 9.0)
    assert its[kept].max() < 200


def test_qr_step_against_numpy(oracle):
    """a single iteration equals the numpy least-squares solution of H(:, :3) dx = -b"""
    c = landmark_case.make(40, 2)
    one = oracle.landmark_params(cap_iterations=1, convergence_delta=1e300)       # accept after the first update
    out, st, err, its = oracle.landmarks_optimize(one, c["PL"], c["PR"], c["seg"], c["frame"], c["uvl"], c["uvr"], c["xyz0"])
    for l in np.nonzero(np.diff(c["seg"]) > 5)[0][:10]:
        X = np.append(c["xyz0"][l], 1.0)
        H, b, inl = np.zeros((4, 4)), np.zeros(4), 0
        for q in range(c["seg"][l], c["seg"][l + 1]):
            PL, PR = c["PL"][c["frame"][q]].reshape(3, 4), c["PR"][c["frame"][q]].reshape(3, 4)
            J, e = np.zeros((4, 4)), np.zeros(4)
            for k, (P, uv) in enumerate(((PL, c["uvl"][q]), (PR, c["uvr"][q]))):
                a = P @ X
                e[2 * k:2 * k + 2] = a[:2] / a[2] - uv
                D = np.array([[1 / a[2], 0, -a[0] / a[2] ** 2], [0, 1 / a[2], -a[1] / a[2] ** 2]])
                J[2 * k:2 * k + 2] = D @ P
            e2 = e @ e
            w = 10 / e2 if e2 > 10 else 1.0
            inl += e2 <= 10
            H += w * J.T @ J
            b += w * J.T @ e
        dx = np.linalg.lstsq(H[:, :3], -b, rcond=None)[0]
        if st[l] in (1, 2):
            assert np.allclose(out[l], c["xyz0"][l] + dx, rtol=1e-9, atol=1e-9)
