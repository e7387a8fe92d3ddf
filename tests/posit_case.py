"""Synthetic CSolverStereoPosit inputs shared by the CPU and GPU tests."""
import numpy as np

import track_scene as ts


def make(n, seed, noise=0.3, outliers=0.1, motion=(0.01, -0.02, 0.005, 0.1, -0.05, 0.3)):
    r = np.random.default_rng(seed)
    T_true = ts.pack(ts.rot(*motion[:3]), motion[3:])
    z = np.exp(r.uniform(np.log(3), np.log(60), n))
    u, v = r.uniform(50, 1190, n), r.uniform(30, 340, n)
    pc = np.stack([(u - ts.CX) / ts.FX * z, (v - ts.CY) / ts.FX * z, z], 1)
    Ti = ts.inv12(T_true)
    xw = pc @ Ti[:9].reshape(3, 3).T + Ti[9:]
    uvl = np.stack([u, v], 1) + r.normal(0, noise, (n, 2))
    uvr = uvl.copy()
    uvr[:, 0] -= -ts.DUR / z
    uvl, uvr = uvl.astype(np.float32), uvr.astype(np.float32)
    bad = r.random(n) < outliers
    uvl[bad] += r.normal(0, 30, (int(bad.sum()), 2)).astype(np.float32)
    behind = r.random(n) < 0.02          # points that end up behind the camera are skipped (:41)
    xw[behind] = (np.array([0.0, 0.0, -5.0]) @ Ti[:9].reshape(3, 3).T + Ti[9:])
    T_est = ts.pack(np.eye(3), [0, 0, 0])
    return dict(T_true=T_true, T_est=T_est, T_last=T_est, t_imu=np.zeros(3), xyz=xw, uvl=uvl, uvr=uvr)
