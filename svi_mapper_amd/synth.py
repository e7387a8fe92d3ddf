"""Synthetic inputs of the BASELINE configurations.

C2: 2 x 2048 BRIEF-256 descriptor sets with an epipolar gate        (seed 0xC2)
C3: KITTI-00-shaped BA graph, 100 KF / 20 k landmarks / 150 k edges   (seed 0xC3)
C4: KITTI-00-shaped BA graph, 500 KF / 100 k landmarks / 800 k edges  (seed 0xC4)

The camera is KITTI 00 (hardware_parameters/kitti_00_camera_left.txt:3-4,16 and
kitti_00_camera_right.txt:16 in the reference). Everything here is numpy on the host: it
produces the *inputs* of the hot path and `build_ba_graph` feeds them through the same calls the
reference's Cg2oOptimizer makes (pose + odometry + gravity edge per keyframe, then the
measurement admission rule), on any object with the BundleAdjuster method surface.
"""
import numpy as np

# KITTI 00 rectified projection matrices
KITTI_WIDTH = 1241
KITTI_HEIGHT = 376
KITTI_F = 718.856
KITTI_CX = 607.1928
KITTI_CY = 185.2157
KITTI_DUR_FLIPPED = 386.1448  # -P_R(0,3)
KITTI_BASELINE_PARAM = 0.54   # tracker_gt.cpp:123 -> CStereoCamera::m_dBaselineMeters
FOV_BORDER = 28               # CPinholeCamera.h:61 rect(28,28,w-56,h-56)
POSE_ID_SHIFT = 1000000       # Cg2oOptimizer.h:83


def kitti_camera():
    return dict(fx=KITTI_F, fy=KITTI_F, cx=KITTI_CX, cy=KITTI_CY, baseline_m=KITTI_BASELINE_PARAM,
                duR_flipped=KITTI_DUR_FLIPPED, width=KITTI_WIDTH, height=KITTI_HEIGHT)


# ------------------------------------------------------------------------------------------------
# C2: descriptor sets
# ------------------------------------------------------------------------------------------------
def make_descriptor_pair(nq=2048, nt=2048, seed=0xC2, match_frac=0.8, flip_p=0.06, tie_frac=0.01,
                         min_range=60.0):
    """Returns dict(q, t, gate=dict(q_uv, t_uv, q_umin, q_umax, v_tol), cutoff, truth).

    Right keypoints are sorted by (row, u) so that "lowest index" == "leftmost pixel of the row",
    the order in which the reference enumerates its candidate pool (CTriangulator.cpp:74-77)."""
    rng = np.random.default_rng(seed)
    v_l = rng.integers(FOV_BORDER, KITTI_HEIGHT - FOV_BORDER, nq).astype(np.float32)
    u_l = rng.uniform(150.0, KITTI_WIDTH - FOV_BORDER - 1, nq).astype(np.float32)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)

    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    v_r = rng.integers(FOV_BORDER, KITTI_HEIGHT - FOV_BORDER, nt).astype(np.float32)
    u_r = rng.uniform(FOV_BORDER, KITTI_WIDTH - FOV_BORDER - 1, nt).astype(np.float32)
    n_match = int(min(nq, nt) * match_frac)
    src = rng.permutation(nq)[:n_match]     # left keypoint of each true match
    dst = rng.permutation(nt)[:n_match]     # its slot on the right
    disp = rng.uniform(1.0, 120.0, n_match).astype(np.float32)
    flips = (rng.random((n_match, 256)) < flip_p)
    t[dst] = q[src] ^ np.packbits(flips, axis=1, bitorder="little")
    v_r[dst] = v_l[src]
    u_r[dst] = u_l[src] - disp
    # ties: copy a matched right descriptor next to its original (same row, 1..3 px away)
    n_tie = int(nt * tie_frac)
    free = np.setdiff1d(np.arange(nt), dst)
    tie_dst = rng.permutation(free)[:n_tie]
    tie_src = dst[rng.permutation(n_match)[:len(tie_dst)]]
    t[tie_dst] = t[tie_src]
    v_r[tie_dst] = v_r[tie_src]
    u_r[tie_dst] = u_r[tie_src] + rng.choice(np.array([-3, -2, -1, 1, 2, 3], np.float32), len(tie_dst))

    order = np.lexsort((u_r, v_r))
    t, u_r, v_r = t[order], u_r[order], v_r[order]
    inv = np.empty(nt, np.int64)
    inv[order] = np.arange(nt)
    truth = np.full(nq, -1, np.int64)
    truth[src] = inv[dst]

    # gate: previous-frame disparity prediction (CFundamentalMatcher.cpp:120,386)
    d_prev = rng.uniform(1.0, 120.0, nq).astype(np.float32)
    d_prev[src] = disp * rng.uniform(0.8, 1.2, n_match).astype(np.float32)
    rng_px = np.maximum(np.float32(min_range), np.float32(1.5) * d_prev).astype(np.float32)
    gate = dict(q_uv=np.stack([u_l, v_l], 1).astype(np.float32), t_uv=np.stack([u_r, v_r], 1).astype(np.float32),
                q_umin=(u_l - rng_px).astype(np.float32), q_umax=u_l.copy(), v_tol=0.0)
    return dict(q=q, t=t, gate=gate, cutoff=100, truth=truth)


# ------------------------------------------------------------------------------------------------
# C3 / C4: KITTI-00-shaped BA graphs
# ------------------------------------------------------------------------------------------------
def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    R = np.zeros(a.shape + (3, 3))
    R[..., 0, 0] = c
    R[..., 0, 2] = s
    R[..., 1, 1] = 1.0
    R[..., 2, 0] = -s
    R[..., 2, 2] = c
    return R


def _small_rot(w):
    """Rodrigues for a batch of rotation vectors (n,3)."""
    th = np.linalg.norm(w, axis=-1)
    k = w / np.maximum(th, 1e-300)[..., None]
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    s, c = np.sin(th)[..., None, None], np.cos(th)[..., None, None]
    return np.eye(3) + s * K + (1 - c) * (K @ K)


def trajectory(n_kf, step=1.0, straight=40, turn=15):
    """Planar path: camera +z forward, `straight` keyframes straight then a 90 deg turn over
    `turn` keyframes, repeating. Returns LEFT->WORLD (R[n,3,3], t[n,3])."""
    heading = np.zeros(n_kf)
    period = straight + turn
    for k in range(1, n_kf):
        ph = (k - 1) % period
        heading[k] = heading[k - 1] + ((np.pi / 2) / turn if ph >= straight else 0.0)
    R = _rot_y(heading)
    fwd = R[:, :, 2]
    t = np.zeros((n_kf, 3))
    t[1:] = np.cumsum(fwd[:-1] * step, axis=0)
    return R, t


def make_ba_problem(n_kf=100, n_lm=20000, n_edges=150000, seed=0xC3, px_sigma=0.5,
                    pose_sigma_t=0.05, pose_sigma_r_deg=0.5, max_run=40, pose_init=None, cam=None, traj=None,
                    depth_range=(2.5, 90.0), max_depth=95.0):
    """Raw synthetic problem (numpy). Keys:
      cam, R_true,t_true, R_init,t_init (LEFT->WORLD), lm_true, lm_init,
      obs_kf, obs_lm (sorted by keyframe), uvL, uvR (float32), xyz (f64, getPointInLEFT of the pixels)
    cam / traj / depth_range / max_depth default to the KITTI-00 shape of configs 3 and 4."""
    rng = np.random.default_rng(seed)
    cam = kitti_camera() if cam is None else cam
    f, cx, cy, dur = cam["fx"], cam["cx"], cam["cy"], cam["duR_flipped"]
    KITTI_WIDTH, KITTI_HEIGHT = cam["width"], cam["height"]  # (the names of the default camera: shadowed on purpose)
    R, t = trajectory(n_kf) if traj is None else traj

    # landmark birth keyframe: uniform, leaving room for >= 2 observations
    k0 = np.sort(rng.integers(0, max(1, n_kf - 1), n_lm))
    z = np.exp(rng.uniform(np.log(depth_range[0]), np.log(depth_range[1]), n_lm))
    u = rng.uniform(FOV_BORDER, KITTI_WIDTH - FOV_BORDER, n_lm)
    v = rng.uniform(FOV_BORDER, KITTI_HEIGHT - FOV_BORDER, n_lm)
    pc = np.stack([z * (u - cx) / f, z * (v - cy) / f, z], 1)
    X = np.einsum("nij,nj->ni", R[k0], pc) + t[k0]

    # visibility: the contiguous run of keyframes around k0 in which the landmark projects inside
    # the FoV rect with z > 1 m (forward first, then extended backwards)
    def visible(k):
        inb = (k >= 0) & (k < n_kf)
        kk = np.clip(k, 0, n_kf - 1)
        p = np.einsum("nji,nj->ni", R[kk], X - t[kk])
        zc = np.maximum(p[:, 2], 1e-9)
        uu = f * p[:, 0] / zc + cx
        vv = f * p[:, 1] / zc + cy
        return inb & (p[:, 2] > min(1.0, 0.5 * depth_range[0])) & (p[:, 2] < max_depth) & (uu >= FOV_BORDER) & (uu < KITTI_WIDTH - FOV_BORDER) & \
            (vv >= FOV_BORDER) & (vv < KITTI_HEIGHT - FOV_BORDER)

    fwd = np.zeros(n_lm, np.int64)
    alive = np.ones(n_lm, bool)
    for s in range(max_run):
        alive = alive & visible(k0 + s)
        fwd += alive
    fwd = np.maximum(fwd, 1)
    back = np.zeros(n_lm, np.int64)
    alive = np.ones(n_lm, bool)
    for s in range(1, max_run):
        alive = alive & visible(k0 - s)
        back += alive
    kstart = k0 - back
    run = fwd + back

    def observe(kstart, run):
        obs_lm = np.repeat(np.arange(n_lm), run)
        obs_kf = kstart[obs_lm] + (np.arange(len(obs_lm)) - np.repeat(np.cumsum(run) - run, run))
        return obs_lm, obs_kf

    def measure(obs_lm, obs_kf, sub):
        p = np.einsum("nji,nj->ni", R[obs_kf], X[obs_lm] - t[obs_kf])
        uL = (f * p[:, 0] / p[:, 2] + cx + sub.normal(0, px_sigma, len(p))).astype(np.float32)
        vL = (f * p[:, 1] / p[:, 2] + cy + sub.normal(0, px_sigma, len(p))).astype(np.float32)
        d = np.maximum(np.rint(dur / p[:, 2] + sub.normal(0, px_sigma, len(p))), 1.0).astype(np.float32)
        uR = (uL - d).astype(np.float32)
        uvL = np.stack([uL, vL], 1)
        uvR = np.stack([uR, vL], 1)
        # CTriangulator::getPointInLEFT (CTriangulator.cpp:326-356) in the reference's operand order
        disp = (uvL[:, 0] - uvR[:, 0]).astype(np.float32).astype(np.float64)
        zz = dur / disp
        finv = 1.0 / f
        xyz = np.stack([finv * zz * (uvL[:, 0].astype(np.float64) - cx),
                        finv * zz * (uvL[:, 1].astype(np.float64) - cy), zz], 1)
        return uvL, uvR, xyz

    # initial estimates: poses perturbed (first pose exact, it is the fixed gauge)
    w = rng.normal(0, np.deg2rad(pose_sigma_r_deg), (n_kf, 3))
    dt = rng.normal(0, pose_sigma_t, (n_kf, 3))
    w[0] = 0
    dt[0] = 0
    R_init = R @ _small_rot(w)
    t_init = t + dt
    if pose_init is not None:   # another landmark set of the SAME (perturbed) trajectory: make_c4(landmark_scale > 1)
        R_init, t_init = pose_init

    # Size the problem on the number of edges the reference's admission rule will keep
    # (Cg2oOptimizer.cpp:1402-1452, replicated here only to count): trim runs from either end,
    # uniformly at random, until about n_edges observations remain admitted. Measurements are drawn
    # once for the untrimmed runs so trimming does not reshuffle the noise.
    kstart0, run0 = kstart.copy(), run.copy()
    off0 = np.cumsum(run0) - run0
    full_lm, full_kf = observe(kstart0, run0)
    uvL, uvR, xyz = measure(full_lm, full_kf, rng)
    l2abs = (xyz ** 2).sum(1)
    disp_ok = (uvL[:, 0] - uvR[:, 0]) > 1.0
    # (the per-observation quantities are kept up to date incrementally: a trimming step only touches the
    # observations of the landmarks it trimmed - same values as recomputing everything, a tenth of the time)
    keep = (full_kf >= kstart[full_lm]) & (full_kf < (kstart + run)[full_lm])
    first = off0 + (kstart - kstart0)
    lm_init = np.einsum("nij,nj->ni", R_init[kstart], xyz[first]) + t_init[kstart]
    pe = np.einsum("nji,nj->ni", R_init[full_kf], lm_init[full_lm] - t_init[full_kf])
    rel = (pe ** 2).sum(1) / l2abs
    type_ok = (l2abs < 50.0) | ((l2abs < 10000.0) & disp_ok)
    for _ in range(400):
        adm = keep & (rel > 0.75) & (rel < 1.25) & type_ok
        excess = int(adm.sum()) - n_edges
        if excess <= max(8, n_edges // 2000):
            break
        cand = np.flatnonzero(run > 2)
        if len(cand) == 0:
            break
        take = min(max(1, int(0.8 * excess * keep.sum() / max(1, adm.sum()))), len(cand))
        sel = rng.permutation(cand)[:take]
        head = rng.random(len(sel)) < 0.5
        kstart[sel[head]] += 1
        run[sel] -= 1
        # observations of the trimmed landmarks
        cnt = run0[sel]
        idx = np.repeat(off0[sel], cnt) + (np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt))
        il, ik = full_lm[idx], full_kf[idx]
        keep[idx] = (ik >= kstart[il]) & (ik < (kstart + run)[il])
        hs = sel[head]                      # their first observation moved: the initial position changes
        if len(hs):
            fh = off0[hs] + (kstart[hs] - kstart0[hs])
            lm_init[hs] = np.einsum("nij,nj->ni", R_init[kstart[hs]], xyz[fh]) + t_init[kstart[hs]]
            ch = run0[hs]
            ih = np.repeat(off0[hs], ch) + (np.arange(int(ch.sum())) - np.repeat(np.cumsum(ch) - ch, ch))
            peh = np.einsum("nji,nj->ni", R_init[full_kf[ih]], lm_init[full_lm[ih]] - t_init[full_kf[ih]])
            rel[ih] = (peh ** 2).sum(1) / l2abs[ih]
    obs_lm, obs_kf = full_lm[keep], full_kf[keep]
    uvL, uvR, xyz = uvL[keep], uvR[keep], xyz[keep]

    order = np.argsort(obs_kf, kind="stable")
    return dict(cam=cam, n_kf=n_kf, n_lm=n_lm, R_true=R, t_true=t, R_init=R_init, t_init=t_init,
                lm_true=X, lm_init=lm_init, obs_kf=obs_kf[order], obs_lm=obs_lm[order],
                uvL=uvL[order], uvR=uvR[order], xyz=xyz[order])


# ------------------------------------------------------------------------------------------------
# C5: vi_sensor stereo + IMU (hardware_parameters/vi_sensor_camera_left.txt, ..._right.txt in the reference)
# ------------------------------------------------------------------------------------------------
VI_WIDTH = 752
VI_HEIGHT = 480
VI_F = 450.5097158071153          # matProjection(0,0) of both rectified cameras (vi_sensor_camera_left.txt:19)
VI_CX = 375.9431800842285
VI_CY = 222.3379611968994
VI_DUR_FLIPPED = 49.63250853439215  # -P_R(0,3)  (vi_sensor_camera_right.txt:19)
# vecQuaternionToIMU (x, y, z, w), vecTranslationToIMU, matRotationIntrinsicCAMERAtoIMU (vi_sensor_camera_*.txt:15-16, 21)
VI_Q_LEFT = (-0.00333631563313, 0.00154028789643, -0.0114620263178, 0.999927556608)
VI_T_LEFT = (0.0666914200614, 0.0038316133947, -0.0101029245794)
VI_Q_RIGHT = (-0.00186686047363, 6.55239757426e-05, -0.00862255915657, 0.999961080249)
VI_T_RIGHT = (-0.0434705406089, 0.00417949317011, -0.00942355850866)
VI_R_CORRECTION = np.diag([-1.0, -1.0, 1.0])


def _quat_R(x, y, z, w):
    n = np.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _vi_camera_to_imu(q, t):
    """CPinholeCameraIMU (CPinholeCameraIMU.h:36-50): R_CAMERAtoIMU = correction * R(q), translation as given"""
    return VI_R_CORRECTION @ _quat_R(*q), np.asarray(t, np.float64)


def vi_sensor_imu_to_left():
    """m_matTransformationIMUtoCAMERA of the LEFT camera as 12 doubles (R row-major, t): the g2o offset parameter
    eOFFSET_IMUtoLEFT (Cg2oOptimizer.cpp:213)"""
    Rc, tc = _vi_camera_to_imu(VI_Q_LEFT, VI_T_LEFT)
    return np.concatenate([Rc.T.reshape(9), -Rc.T @ tc])


def vi_sensor_camera():
    """CStereoCameraIMU (CStereoCameraIMU.h:21-25): the baseline is |translation of RIGHTtoIMU^-1 * LEFTtoIMU|"""
    Rl, tl = _vi_camera_to_imu(VI_Q_LEFT, VI_T_LEFT)
    Rr, tr = _vi_camera_to_imu(VI_Q_RIGHT, VI_T_RIGHT)
    base = float(np.linalg.norm(Rr.T @ (tl - tr)))
    return dict(fx=VI_F, fy=VI_F, cx=VI_CX, cy=VI_CY, baseline_m=base, duR_flipped=VI_DUR_FLIPPED, width=VI_WIDTH,
                height=VI_HEIGHT)


def vi_trajectory(n_kf, seed=0, step=0.72):
    """hand-held walk: a key frame every sqrt(0.5) m or so (CTrackerSVI.h:53), slow turns plus a little roll and pitch -
    the attitude changes are what the gravity edges see"""
    r = np.random.default_rng(seed)
    k = np.arange(n_kf)
    heading = 0.6 * np.sin(2 * np.pi * k / 37.0) + 0.02 * k
    R = _rot_y(heading) @ _small_rot(np.stack([0.05 * np.sin(k / 5.0), np.zeros(n_kf), 0.04 * np.cos(k / 7.0)], 1))
    fwd = R[:, :, 2]
    t = np.zeros((n_kf, 3))
    t[1:] = np.cumsum(fwd[:-1] * step, axis=0)
    t[:, 1] += 0.03 * np.sin(k / 3.0)
    del r
    return R, t


def make_vi_problem(n_kf=60, n_lm=6000, n_edges=40000, seed=0xC5, accel_sigma=0.01, **kw):
    """BASELINE config 5 as a BA graph: the vi_sensor camera, indoor depths, and per key frame the NORMALISED accelerometer
    reading (CTrackerSVI.cpp:651) in the IMU frame: a = (R_k R_off)' g + noise, g = (0,0,-1) - whatever direction that is
    in the camera frame of the first key frame, the edges only ask for consistency with the poses.  Extra keys: accel
    (n_kf x 3, unit norm), imu_off (12), accel_first = (0,-1,0) of the constructor's edge (Cg2oOptimizer.cpp:154)."""
    cam = vi_sensor_camera()
    R, t = vi_trajectory(n_kf, seed)
    prob = make_ba_problem(n_kf, n_lm, n_edges, seed, cam=cam, traj=(R, t), depth_range=(0.8, 25.0), max_depth=30.0,
                           pose_sigma_t=kw.pop("pose_sigma_t", 0.03), **kw)
    off = vi_sensor_imu_to_left()
    R_off = off[:9].reshape(3, 3)
    rng = np.random.default_rng(seed + 17)
    g = np.array([0.0, 0.0, -1.0])
    a = np.einsum("ij,nkj,k->ni", R_off.T, R, g)      # R_off' R_k' g
    a = a + rng.normal(0, accel_sigma, a.shape)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    prob.update(accel=a, imu_off=off, accel_first=np.array([0.0, -1.0, 0.0]))
    return prob


def make_c3(seed=0xC3):
    return make_ba_problem(100, 20000, 150000, seed)


def _c4_part(args):
    seed, pose_init = args
    return make_ba_problem(500, 100000, 800000, seed, pose_init=pose_init)


def make_c4(seed=0xC4, landmark_scale=1):
    """BASELINE config 4.  landmark_scale s > 1 (weak-scaling shards): s landmark sets of 100 k / 800 k edges each over
    the same 500 key frames and the same perturbed initial poses - set 0 is config 4 itself, sets 1.. are further
    draws (seed + 7919 c), generated in parallel processes."""
    base = make_ba_problem(500, 100000, 800000, seed)
    if landmark_scale <= 1:
        return base
    jobs = [(seed + 7919 * c, (base["R_init"], base["t_init"])) for c in range(1, landmark_scale)]
    try:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(len(jobs), 8)) as pool:
            parts = [base] + pool.map(_c4_part, jobs)
    except Exception:
        parts = [base] + [_c4_part(j) for j in jobs]
    n_lm = base["n_lm"]
    obs_kf = np.concatenate([q["obs_kf"] for q in parts])
    order = np.argsort(obs_kf, kind="stable")
    cat = lambda k: np.concatenate([q[k] for q in parts])[order]  # noqa: E731
    out = dict(base)
    out.update(n_lm=n_lm * len(parts), lm_true=np.concatenate([q["lm_true"] for q in parts]),
               lm_init=np.concatenate([q["lm_init"] for q in parts]), obs_kf=obs_kf[order],
               obs_lm=np.concatenate([q["obs_lm"] + c * n_lm for c, q in enumerate(parts)])[order],
               uvL=cat("uvL"), uvR=cat("uvR"), xyz=cat("xyz"))
    return out


def pose12(R, t):
    return np.concatenate([np.asarray(R, np.float64).reshape(9), np.asarray(t, np.float64).reshape(3)])


def build_ba_graph(ba, prob):
    """Feed `prob` through the reference's construction sequence:
      ctor:   fixed first pose + its gravity edge                 (Cg2oOptimizer.cpp:41-54)
      addLandmarkToGraph for every landmark                       (:1135-1153)
      per keyframe: _setAndgetPose, gravity edge, _setLandmarkMeasurementsWORLD   (:471-490)
    Returns the per-kind edge counts [xyz, depth, disparity]."""
    n_kf = prob["n_kf"]
    ids_lm = np.arange(prob["n_lm"], dtype=np.int64)
    imu = prob.get("imu_off")       # config 5: the IMU constructor's first edge and offset (Cg2oOptimizer.cpp:154, :213)
    if imu is not None:
        ba.set_imu_offset(imu)
    ba.add_pose(POSE_ID_SHIFT, pose12(prob["R_init"][0], prob["t_init"][0]), fixed=True)
    ba.add_edge_accel(POSE_ID_SHIFT, prob.get("accel_first", np.zeros(3)), imu)
    ba.add_landmarks(ids_lm, prob["lm_init"])
    starts = np.searchsorted(prob["obs_kf"], np.arange(n_kf + 1))
    stored = np.zeros(3, np.int64)
    for k in range(n_kf):
        if k > 0:
            ba.add_keyframe(POSE_ID_SHIFT + k, POSE_ID_SHIFT + k - 1, pose12(prob["R_init"][k], prob["t_init"][k]),
                            accel=prob["accel"][k] if "accel" in prob else None)
        a, b = starts[k], starts[k + 1]
        if b > a:
            stored += ba.add_measurements(POSE_ID_SHIFT + k, prob["obs_lm"][a:b].astype(np.int64),
                                          prob["uvL"][a:b], prob["uvR"][a:b], prob["xyz"][a:b])
    return stored
