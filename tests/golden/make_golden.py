"""Generates the golden vectors under tests/golden/ from OUR CPU oracle (oracle/).

The reference ships no tests or fixtures and its arithmetic (OpenCV, g2o, CHOLMOD) cannot be built
or imported here, so these vectors pin the oracle against regressions and give the
GPU tests fixed inputs; they do NOT pin the oracle to g2o ("parity unpinned").

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import oracle as orc  # noqa: E402
from svi_mapper_amd import synth  # noqa: E402


def track_golden():
    """tests/golden/track_small.npz: the tracking schedule of a 96-landmark synthetic frame (tests/track_scene.py)"""
    sys.path.insert(0, os.path.dirname(HERE))
    import track_scene as ts
    sc = ts.Scene(n=96, seed=3)
    cam = orc.track_camera(ts.P_LEFT, ts.P_RIGHT, ts.K_INV, ts.W, ts.H)
    rec, seg = orc.track_plan(cam, sc.T_est_w2l, sc.dp_T, sc.motion_scaling, sc.xyz_world, sc.kp_size, sc.last_disparity, sc.uv_reference,
                              sc.dp_index)
    s0, roi0 = orc.track_epipolar_samples(cam, rec, sc.kp_size, seg, 0)
    s2, roi2 = orc.track_epipolar_samples(cam, rec, sc.kp_size, seg, 2)
    om = orc.OracleFundamentalMatcher(cam, sc.stereo_dict())
    out = {}
    for name, res in (("s1", om.stage1(rec, sc.kp_size, sc.extract_one, sc.last_left, sc.last_right)),
                      ("s2", om.stage2(rec, sc.kp_size, sc.detect_one, sc.extract_one, sc.last_left, sc.last_right)),
                      ("s3", om.epipolar(rec, sc.kp_size, sc.extract_one, sc.last_left, sc.ref_desc))):
        out[name + "_status"] = np.array([d["status"] for d in res], np.int32)
        out[name + "_uv_left"] = np.array([d.get("uv_left", (0, 0)) for d in res], np.float32)
        out[name + "_uv_right"] = np.array([d.get("uv_right", (0, 0)) for d in res], np.float32)
        out[name + "_xyz"] = np.array([d.get("xyz", (0, 0, 0)) for d in res], np.float64)
    np.savez_compressed(os.path.join(HERE, "track_small.npz"), records=rec.view(np.uint8).reshape(len(rec), -1), seg=seg, samples0=s0, roi0=roi0,
                        samples2=s2, roi2=roi2, **out)


def main():
    track_golden()
    s = synth.make_descriptor_pair(257, 1023, seed=0xC2)
    gt = s["gate"]
    i, d = orc.match_hamming256(s["q"], s["t"], gt, s["cutoff"])
    iu, du = orc.match_hamming256(s["q"], s["t"])
    np.savez_compressed(os.path.join(HERE, "hamming_c2_small.npz"), q=s["q"], t=s["t"], q_uv=gt["q_uv"], t_uv=gt["t_uv"],
                        q_umin=gt["q_umin"], q_umax=gt["q_umax"], v_tol=np.float32(gt["v_tol"]), cutoff=np.int32(s["cutoff"]),
                        idx=i, dist=d, idx_ungated=iu, dist_ungated=du)
    cam = synth.kitti_camera()
    rng = np.random.default_rng(7)
    uvL = np.stack([rng.uniform(100, 1200, 256), rng.integers(28, 340, 256)], 1).astype(np.float32)
    disp = rng.uniform(-0.5, 150, 256).astype(np.float32)
    disp[:6] = [0.0, 0.009, 0.01, 0.0100001, 1.0, 386.1448]
    uvR = uvL.copy()
    uvR[:, 0] = uvL[:, 0] - disp
    xyz, ok = orc.triangulate_rectified(cam["fx"], cam["cx"], cam["cy"], cam["duR_flipped"], uvL, uvR)
    np.savez_compressed(os.path.join(HERE, "triangulate.npz"), uvL=uvL, uvR=uvR, xyz=xyz, ok=ok)

    n_kf, n_lm, n_edges, seed = 6, 60, 330, 42
    prob = synth.make_ba_problem(n_kf, n_lm, n_edges, seed=seed)
    ba = orc.OracleBA(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    stored = synth.build_ba_graph(ba, prob)
    ty, pid, lid, z, info = ba.get_edges()
    ba.initialize()
    e, Jp, Jl = ba.edge_jacobians()
    chi0 = ba.chi2()
    nom, exe = ba.optimize_until()
    _, T = ba.get_poses()
    _, pl = ba.get_landmarks()
    np.savez_compressed(os.path.join(HERE, "ba_tiny.npz"), n_kf=n_kf, n_lm=n_lm, n_edges=n_edges, seed=seed, stored=stored,
                        edge_type=ty, edge_pose=pid, edge_lm=lid, edge_z=z, edge_info=info, err0=e, Jp0=Jp, Jl0=Jl,
                        chi0=np.array(chi0), nominal=nom, executed=exe, trace=ba.trace(), poses=T, landmarks=pl)
    print("golden vectors written:", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
