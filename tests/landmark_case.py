"""Synthetic CLandmark::optimize inputs: a short trajectory of frames, landmarks measured in runs of consecutive frames
with pixel noise, some gross outliers, perturbed initial positions."""
import numpy as np

import track_scene as ts


def make(n, seed, n_frames=40, noise=0.3):
    r = np.random.default_rng(seed)
    PL, PR, Tw2c = [], [], []
    for f in range(n_frames):
        T = ts.inv12(ts.pack(ts.rot(0.0, 0.004 * f, 0.0), [0.02 * f, 0.0, 0.6 * f]))   # WORLD -> LEFT of frame f
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = T[:9].reshape(3, 3), T[9:]
        PL.append((ts.P_LEFT @ M).ravel())
        PR.append((ts.P_RIGHT @ M).ravel())
        Tw2c.append(T)
    PL, PR = np.array(PL), np.array(PR)
    seg, frames, uvl, uvr, xyz_true = [0], [], [], [], []
    for i in range(n):
        f0 = r.integers(0, n_frames - 2)
        cnt = int(r.integers(1, min(30, n_frames - f0)))        # 1..29 measurements: some are skipped (<= 5)
        T = Tw2c[f0]
        z = np.exp(r.uniform(np.log(4), np.log(50)))
        pc = np.array([(r.uniform(100, 1100) - ts.CX) / ts.FX * z, (r.uniform(50, 320) - ts.CY) / ts.FX * z, z])
        Ti = ts.inv12(T)
        pw = Ti[:9].reshape(3, 3) @ pc + Ti[9:]
        xyz_true.append(pw)
        gross = r.random() < 0.06
        for f in range(f0, f0 + cnt):
            hl = PL[f].reshape(3, 4) @ np.append(pw, 1.0)
            hr = PR[f].reshape(3, 4) @ np.append(pw, 1.0)
            ul = hl[:2] / hl[2] + r.normal(0, noise, 2)
            ur = np.array([hr[0] / hr[2] + r.normal(0, noise), ul[1]])
            if r.random() < (0.6 if gross else 0.03):
                ul += r.normal(0, 25, 2)
            frames.append(f); uvl.append(ul); uvr.append(ur)
        seg.append(len(frames))
    xyz_true = np.array(xyz_true)
    xyz0 = xyz_true + r.normal(0, 0.15, xyz_true.shape) * (1 + np.linalg.norm(xyz_true, axis=1, keepdims=True) / 20)
    return dict(PL=PL, PR=PR, seg=np.array(seg, np.int32), frame=np.array(frames, np.int32), uvl=np.array(uvl, np.float32).reshape(-1, 2),
                uvr=np.array(uvr, np.float32).reshape(-1, 2), xyz0=xyz0, xyz_true=xyz_true)
