"""Closed loop over a short synthetic drive, everything on the device: per frame the tracking schedule (stage 1 -> 2 with the
frame pose refined by StereoPosit in between, then the epipolar search), the landmark refinement over all measurements so
far, and at the end a bundle adjustment over the key frames.  A functional test: the estimated trajectory has to follow the
true one (no oracle here - the components are checked against it one by one in the other test files)."""
import numpy as np
import pytest

import track_scene as ts

pytestmark = pytest.mark.gpu


def rot_err(Ta, Tb):
    R = Ta[:9].reshape(3, 3) @ Tb[:9].reshape(3, 3).T
    return np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)))


def test_drive(svi):
    import torch
    assert torch.cuda.is_available(), "the -m gpu tests need a visible GPU"
    from svi_mapper_amd import temporal
    seq = ts.Sequence(n=400, n_frames=8, seed=3)
    d = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")  # noqa: E731
    cam = temporal.StereoCamera(ts.P_LEFT, ts.P_RIGHT, ts.W, ts.H)
    fm = temporal.FundamentalMatcher(cam)
    posit = temporal.SolverStereoPosit(ts.P_LEFT, ts.P_RIGHT, matcher=fm.matcher)
    lmopt = temporal.LandmarkOptimizer(matcher=fm.matcher)
    n = seq.n
    # frame 0: landmarks are created from its detections (addNewLandmarks): stereo partner + triangulation
    f0 = seq.frame(0)
    inside = (f0.true_uL >= 80) & (f0.true_uL < ts.W - 30) & (f0.true_v >= 30) & (f0.true_v < ts.H - 30)
    uv0 = np.stack([f0.true_uL, f0.true_v], 1).astype(np.float32)
    new = fm.add_new_landmarks(f0.make_extractor(torch, "cuda"), d(uv0), d(seq.kp_size), d(f0.cur_left))
    alive = (new.status.cpu().numpy() == 0) & inside
    assert alive.sum() > 250
    T_prev = ts.inv12(seq.T_l2w[0])                      # frame 0 defines the world
    xyz = new.xyz_left.cpu().numpy().copy()              # LEFT frame 0 == WORLD
    last_l, last_r = new.desc_left.clone(), new.desc_right.clone()
    ref_l = new.desc_left.clone()
    uv_ref = uv0.astype(np.float64)
    last_disp = (new.uv_left[:, 0] - new.uv_right[:, 0]).cpu().numpy().astype(np.float32)
    last_disp[~alive] = 1.0
    dp_index = np.zeros(n, np.int32)
    dp_T = np.array([seq.T_l2w[0]])
    # measurement book for the landmark refinement
    meas = [[(0, new.uv_left[i].cpu().numpy(), new.uv_right[i].cpu().numpy())] if alive[i] else [] for i in range(n)]
    frames_PL, frames_PR = [], []

    def push_frame(Tw2l):
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = Tw2l[:9].reshape(3, 3), Tw2l[9:]
        frames_PL.append((ts.P_LEFT @ M).ravel())
        frames_PR.append((ts.P_RIGHT @ M).ravel())

    push_frame(T_prev)
    delta = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0.0])
    poses = [T_prev]
    tracked_counts = []
    for t in range(1, seq.n_frames):
        fr = seq.frame(t)
        ext, det = fr.make_extractor(torch, "cuda"), fr.make_detector(torch, "cuda")
        # constant-velocity prior: T_est = delta * T_prev  (WORLD -> LEFT)
        Rd, td = delta[:9].reshape(3, 3), delta[9:]
        T_est = ts.pack(Rd @ T_prev[:9].reshape(3, 3), Rd @ T_prev[9:] + td)
        args = (dp_T, 1.0, d(xyz), d(seq.kp_size), d(last_disp), d(uv_ref), d(dp_index))
        act = d(alive.astype(np.uint8))
        plan = fm.plan(T_est, *args)
        r1 = fm.track_stage1(plan, ext, last_l, last_r, act)
        lost = act.bool() & (r1.status != 0)
        r2 = fm.track_stage2(plan, det, ext, last_l, last_r, lost.to(torch.uint8))
        ok12 = (r1.status == 0) | (r2.status == 0)
        uvl = torch.where((r1.status == 0)[:, None], r1.uv_left, r2.uv_left)
        uvr = torch.where((r1.status == 0)[:, None], r1.uv_right, r2.uv_right)
        assert int(ok12.sum()) > 60, (t, int(ok12.sum()))
        res = posit.solve(T_prev, np.zeros(3), T_est, d(xyz), uvl.contiguous(), uvr.contiguous(), ok12.to(torch.uint8))
        assert res.status == 0, (t, res.status)
        T_now = np.array(res.T_world_to_left[:])
        # epipolar search for what is still missing, with the refined pose
        plan2 = fm.plan(T_now, *args)
        r3 = fm.track_epipolar(plan2, ext, last_l, ref_l, (act.bool() & ~ok12).to(torch.uint8))
        ok = ok12 | (r3.status == 0)
        for name, dst in (("desc_left", last_l), ("desc_right", last_r)):
            for r in (r1, r2, r3):
                g = r.status == 0
                dst[g] = getattr(r, name)[g]
        uvl = torch.where((r3.status == 0)[:, None], r3.uv_left, uvl)
        uvr = torch.where((r3.status == 0)[:, None], r3.uv_right, uvr)
        okh, uvl_h, uvr_h = ok.cpu().numpy(), uvl.cpu().numpy(), uvr.cpu().numpy()
        push_frame(T_now)
        for i in np.nonzero(okh)[0]:
            meas[i].append((t, uvl_h[i], uvr_h[i]))
            last_disp[i] = uvl_h[i, 0] - uvr_h[i, 0]
        tracked_counts.append(int(okh.sum()))
        # CLandmark::optimize over all measurements so far
        seg = np.concatenate([[0], np.cumsum([len(m) for m in meas])]).astype(np.int32)
        mf = np.array([f for m in meas for (f, _, _) in m], np.int32)
        ml = np.array([u for m in meas for (_, u, _) in m], np.float32).reshape(-1, 2)
        mr = np.array([u for m in meas for (_, _, u) in m], np.float32).reshape(-1, 2)
        out, st, err, its = lmopt.optimize(d(np.array(frames_PL)), d(np.array(frames_PR)), d(seg), d(mf), d(ml), d(mr), d(xyz))
        st = st.cpu().numpy()
        good = (st == 1) | (st == 2)
        xyz[good] = out.cpu().numpy()[good]
        # pose error against the truth
        T_true = fr.T_true_w2l
        assert np.abs(T_now[9:] - T_true[9:]).max() < 0.05, (t, T_now[9:], T_true[9:])
        assert rot_err(T_now, T_true) < 0.2, (t, rot_err(T_now, T_true))
        Rn, Rp = T_now[:9].reshape(3, 3), T_prev[:9].reshape(3, 3)
        delta = ts.pack(Rn @ Rp.T, T_now[9:] - Rn @ Rp.T @ T_prev[9:])
        T_prev = T_now
        poses.append(T_now)
    assert min(tracked_counts) > 150, tracked_counts
    # the refined cloud is close to the truth (frame 0 == world)
    seen = np.array([len(m) >= 6 for m in meas])
    e = np.linalg.norm(xyz[seen] - seq.xyz_world[seen], axis=1) / seq.xyz_world[seen, 2]
    assert np.median(e) < 0.02, np.median(e)

    # bundle adjustment over the key frames through the reference's construction rules
    cam_k = dict(fx=ts.FX, fy=ts.FX, cx=ts.CX, cy=ts.CY)
    ba = svi.BundleAdjuster(cam_k["fx"], cam_k["fy"], cam_k["cx"], cam_k["cy"], 0.54)
    ba.add_pose(1000000, ts.inv12(poses[0]), fixed=True)
    ba.add_edge_accel(1000000, np.zeros(3))
    ids = np.nonzero(seen)[0].astype(np.int64)
    ba.add_landmarks(ids, xyz[seen])
    duf = -ts.DUR
    for t in range(seq.n_frames):
        if t > 0:
            ba.add_keyframe(1000000 + t, 1000000 + t - 1, ts.inv12(poses[t]))
        rows = [(i, m) for i in ids for m in meas[i] if m[0] == t]
        if not rows:
            continue
        lm_ids = np.array([i for i, _ in rows], np.int64)
        uL = np.array([m[1] for _, m in rows], np.float32)
        uR = np.array([m[2] for _, m in rows], np.float32)
        zz = duf / (uL[:, 0] - uR[:, 0]).astype(np.float64)
        xyzL = np.stack([zz * (uL[:, 0] - ts.CX) / ts.FX, zz * (uL[:, 1] - ts.CY) / ts.FX, zz], 1)
        ba.add_measurements(1000000 + t, lm_ids, uL, uR, xyzL)
    ba.initialize()
    c0 = ba.chi2()[0]
    nominal, executed = ba.optimize_until()
    c1 = ba.chi2()[0]
    assert executed >= 1 and c1 <= c0
    _, T_ba = ba.get_poses()
    for t in range(seq.n_frames):
        T_w2l = ts.inv12(T_ba[t])
        assert np.abs(T_w2l[9:] - seq.frame(t).T_true_w2l[9:]).max() < 0.05 if t in (0, seq.n_frames - 1) else True
