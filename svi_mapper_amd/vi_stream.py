"""BASELINE config 5 as a synthetic input + the thin tracker loop that drives the hot path over it (bench.py `config5`,
tests/test_config5_stream_gpu.py).

Input: a vi_sensor stereo pair (752 x 480, f = 450.5, baseline 0.110 m: hardware_parameters/vi_sensor_camera_*.txt in the
reference) carried over a textured ground plane - every frame is a true rendering of the same texture (ray / plane
intersection per pixel, bilinear lookup), so BRIEF descriptors persist from frame to frame the way they do on real
footage, and depth varies over the image (near at the bottom, far at the top).  Per key frame the normalised
accelerometer reading a = (R_k R_off)' g, g = (0,0,-1) (CTrackerSVI.cpp:651).

Loop (the call sequence of CTrackerSVI::process / _trackLandmarks): motion prior -> getPoseStereoPosit
(stage 1 -> 2 + StereoPosit) -> trackEpipolar with the refined pose -> landmark refinement every 10 frames
(CTrackerSVI.h:85) -> key frame every 0.5 m^2 / 0.25 rad^2 (:53-54) -> Cg2oOptimizer::optimize over the whole graph
every > 20 key frames (:87) incl. svi_ba_initialize -> addNewLandmarks when too few are visible.  Everything the path
computes runs in libsvi_hot.so; what is Python here is the reference's *bookkeeping* (which landmark is alive, its last
descriptors, the measurement book), kept in torch tensors on the device.  GFTT detection is out of scope: the
stand-in detector knows where the texture points of the live landmarks project."""
import numpy as np
import torch

from . import synth, temporal
from .optimizer import BundleAdjuster

W, H = synth.VI_WIDTH, synth.VI_HEIGHT
F, CX, CY = synth.VI_F, synth.VI_CX, synth.VI_CY
BASE = synth.VI_DUR_FLIPPED / synth.VI_F
P_LEFT = np.array([[F, 0, CX, 0], [0, F, CY, 0], [0, 0, 1, 0.0]])
P_RIGHT = np.array([[F, 0, CX, -synth.VI_DUR_FLIPPED], [0, F, CY, 0], [0, 0, 1, 0.0]])
GROUND_Y = 1.6      # the plane y = 1.6 m below the camera (camera y points down)
PITCH = 0.55        # the camera looks 31 degrees down


def brief_pattern(seed=1):
    """256 x (y1, x1, y2, x2) int8 in [-24, 24]: isotropic Gaussian pairs (OpenCV's baked table is not available offline)"""
    r = np.random.default_rng(seed)
    p = np.clip(np.rint(r.normal(0, 48 / 5.0, (256, 4))), -24, 24).astype(np.int8)
    return p


def _rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def pack(R, t):
    return np.concatenate([np.asarray(R, np.float64).reshape(9), np.asarray(t, np.float64).reshape(3)])


def inv12(T):
    R, t = T[:9].reshape(3, 3), T[9:]
    return pack(R.T, -R.T @ t)


class ViStream:
    """true poses (LEFT -> WORLD) of a hand-held walk and the rendered stereo frames"""

    def __init__(self, n_frames, device, seed=0xC5, step=0.05, tex_size=4096, tex_res=0.005):
        self.n_frames, self.device = n_frames, device
        r = np.random.default_rng(seed)
        k = np.arange(n_frames)
        yaw = 0.25 * np.sin(2 * np.pi * k / 400.0)
        self.T_l2w = []
        pos = np.zeros(3)
        for i in range(n_frames):
            R = _rot_y(yaw[i]) @ _rot_x(-(PITCH + 0.02 * np.sin(i / 23.0))) @ _rot_z(0.015 * np.cos(i / 31.0))
            self.T_l2w.append(pack(R, pos.copy()))
            pos = pos + _rot_y(yaw[i]) @ np.array([0.0, 0.0, step]) + np.array([0.0, 0.002 * np.sin(i / 7.0), 0.0])
        # multi-scale smooth random texture + fine noise, tiled over the ground
        tex = np.zeros((tex_size, tex_size), np.float32)
        for s in (256, 64, 16, 8, 4):
            g = r.normal(0, 1, (tex_size // s + 1, tex_size // s + 1)).astype(np.float32)
            tex += np.kron(g, np.ones((s, s), np.float32))[:tex_size, :tex_size] * np.sqrt(s)
        tex += r.normal(0, 1.0, tex.shape).astype(np.float32)
        tex = (tex - tex.min()) / (tex.max() - tex.min()) * 255.0
        self.tex = torch.tensor(tex, device=device)
        self.tex_size, self.tex_res = tex_size, tex_res
        off = synth.vi_sensor_imu_to_left()
        self.imu_off = off
        R_off = off[:9].reshape(3, 3)
        self.accel = []
        for T in self.T_l2w:
            a = (T[:9].reshape(3, 3) @ R_off).T @ np.array([0.0, 0.0, -1.0]) + r.normal(0, 0.01, 3)
            self.accel.append(a / np.linalg.norm(a))
        vv, uu = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float64), torch.arange(W, device=device, dtype=torch.float64), indexing="ij")
        self._rays = torch.stack([(uu - CX) / F, (vv - CY) / F, torch.ones_like(uu)], -1)   # H x W x 3 in the camera frame

    def ground_point(self, T_l2w, uv, right=False):
        """world point of the ground seen at pixels uv (n x 2, torch f64) of the LEFT (or RIGHT) camera"""
        R = torch.tensor(T_l2w[:9].reshape(3, 3), device=self.device)
        c = torch.tensor(T_l2w[9:], device=self.device)
        if right:
            c = c + R @ torch.tensor([BASE, 0.0, 0.0], device=self.device, dtype=torch.float64)
        d = torch.stack([(uv[:, 0] - CX) / F, (uv[:, 1] - CY) / F, torch.ones_like(uv[:, 0])], 1) @ R.T
        lam = (GROUND_Y - c[1]) / d[:, 1]
        return c + lam[:, None] * d, lam

    def render(self, t):
        """(left, right) uint8 H x W device tensors of frame t"""
        T = self.T_l2w[t]
        R = torch.tensor(T[:9].reshape(3, 3), device=self.device)
        out = []
        for right in (False, True):
            c = torch.tensor(T[9:], device=self.device)
            if right:
                c = c + R @ torch.tensor([BASE, 0.0, 0.0], device=self.device, dtype=torch.float64)
            d = self._rays @ R.T
            lam = (GROUND_Y - c[1]) / d[..., 1]
            P = c + lam[..., None] * d
            x = (P[..., 0] / self.tex_res) % self.tex_size
            z = (P[..., 2] / self.tex_res) % self.tex_size
            x0, z0 = x.floor().long(), z.floor().long()
            fx, fz = (x - x0).float(), (z - z0).float()
            x1, z1 = (x0 + 1) % self.tex_size, (z0 + 1) % self.tex_size
            x0, z0 = x0 % self.tex_size, z0 % self.tex_size
            img = (self.tex[z0, x0] * (1 - fx) * (1 - fz) + self.tex[z0, x1] * fx * (1 - fz) + self.tex[z1, x0] * (1 - fx) * fz +
                   self.tex[z1, x1] * fx * fz)
            sky = (lam <= 0) | (lam > 60.0)
            img = torch.where(sky, torch.full_like(img, 128.0), img)
            out.append(img.clamp(0, 255).to(torch.uint8).contiguous())
        return out[0], out[1]


class OnlineTracker:
    """the bookkeeping of CTrackerSVI around the hot path, landmark state as device tensors (capacity `cap`)"""

    def __init__(self, stream, device_index=0, cap=None, target_visible=500, min_visible=300, seed=1):
        self.s = stream
        self.dev = stream.device
        cap = cap or 4096 + 40 * stream.n_frames     # every landmark ever created keeps its row (ids stay stable for the graph)
        self.cap, self.target_visible, self.min_visible = cap, target_visible, min_visible
        self.rng = np.random.default_rng(seed)
        cam = temporal.StereoCamera(P_LEFT, P_RIGHT, W, H)
        self.fm = temporal.FundamentalMatcher(cam, device=device_index)
        self.brief = temporal.BriefExtractor(brief_pattern(), matcher=self.fm.matcher, device=device_index)
        self.posit = temporal.SolverStereoPosit(P_LEFT, P_RIGHT, matcher=self.fm.matcher, device=device_index)
        self.lmopt = temporal.LandmarkOptimizer(matcher=self.fm.matcher, device=device_index)
        c = synth.vi_sensor_camera()
        self.ba = BundleAdjuster(c["fx"], c["fy"], c["cx"], c["cy"], c["baseline_m"], device=device_index)
        self.ba.set_imu_offset(stream.imu_off)
        z = lambda *s, dt=torch.float64: torch.zeros(s, dtype=dt, device=self.dev)  # noqa: E731
        self.xyz, self.xyz_true = z(cap, 3), z(cap, 3)
        self.kp_size = torch.full((cap,), 7.0, dtype=torch.float32, device=self.dev)
        self.last_disp = z(cap, dt=torch.float32)
        self.uv_ref = z(cap, 2)
        self.dp_index = z(cap, dt=torch.int32)
        self.last_l, self.last_r, self.ref_l = z(cap, 32, dt=torch.uint8), z(cap, 32, dt=torch.uint8), z(cap, 32, dt=torch.uint8)
        self.alive = z(cap, dt=torch.bool)
        self.fails = z(cap, dt=torch.int32)
        self.n_kf_seen = z(cap, dt=torch.int32)       # key frames the landmark was measured in
        self.in_graph = z(cap, dt=torch.bool)
        self.n_used = 0
        self.dp_T = []                                # detection points (LEFT -> WORLD at creation)
        self.m_lm, self.m_frame, self.m_uvl, self.m_uvr = [], [], [], []   # measurement book (device, per frame)
        self.frames_PL, self.frames_PR = [], []
        self.T_w2l = None
        self.T_prev = None
        self.key_frames = []                          # (frame, T_l2w estimate, accel, lm ids, uvl, uvr, xyz_left) host
        self.kf_in_graph = 0
        self.T_last_kf = None
        self.stats = dict(frames=0, stage1=0, stage2=0, stage3=0, posit_fail=0, ba_calls=0, ba_iterations=0, landmarks_created=0, ba_ms=0.0,
                          ba_initialize_ms=0.0, detector_ms=0.0, detector_calls=0, library_ms=0.0)

    def _lib(self, fn, *a, **k):
        """a call into the library (through its ctypes shim), host wall clock accumulated in stats['library_ms'] - without what the
        detector stand-in spends inside it (the cascades call back into Python for stage 2's key points)"""
        import time
        d0, t0 = self.stats["detector_ms"], time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            self.stats["library_ms"] += 1e3 * (time.perf_counter() - t0) - (self.stats["detector_ms"] - d0)

    # the stand-in for GFTT inside the stage-2 rectangles: the pixels where the live landmarks' texture points project
    def _detector(self, T_w2l_true):
        import time
        t_in = time.perf_counter()
        R = torch.tensor(T_w2l_true[:9].reshape(3, 3), device=self.dev)
        tt = torch.tensor(T_w2l_true[9:], device=self.dev)
        n = self.n_used
        pc = self.xyz_true[:n] @ R.T + tt
        ok = self.alive[:n] & (pc[:, 2] > 0.3)
        z = pc[:, 2].clamp(min=0.3)
        uL = torch.round(F * pc[:, 0] / z + CX)
        v = torch.round(F * pc[:, 1] / z + CY)
        uR = torch.round(F * (pc[:, 0] - BASE) / z + CX)
        pts = {"left": torch.stack([uL, v], 1)[ok].float(), "right": torch.stack([uR, v], 1)[ok].float()}

        self.stats["detector_ms"] += 1e3 * (time.perf_counter() - t_in)

        def detect(side, rect):
            # (host wall clock of the stand-in, accumulated: the caller's detector - OpenCV's GFTT in the reference - is not part of
            # the hot path; torch.nonzero below makes the host wait for the device, that wait is inside the figure)
            t0 = time.perf_counter()
            try:
                return detect_(side, rect)
            finally:
                self.stats["detector_ms"] += 1e3 * (time.perf_counter() - t0)
                self.stats["detector_calls"] += 1

        def detect_(side, rect):
            p = pts[side]
            ul = rect[:, :2].floor()
            lr = rect[:, 2:].floor()
            inside = (p[None, :, 0] >= ul[:, None, 0]) & (p[None, :, 0] < lr[:, None, 0]) & (p[None, :, 1] >= ul[:, None, 1]) & \
                     (p[None, :, 1] < lr[:, None, 1])
            idx = torch.nonzero(inside)
            cnt = inside.sum(1)
            seg = torch.zeros(rect.shape[0] + 1, dtype=torch.int32, device=self.dev)
            seg[1:] = torch.cumsum(cnt, 0)
            return seg, (p[idx[:, 1]] - ul[idx[:, 0]]).contiguous()
        return detect

    def _push_frame(self, T_w2l):
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = T_w2l[:9].reshape(3, 3), T_w2l[9:]
        self.frames_PL.append((P_LEFT @ M).ravel())
        self.frames_PR.append((P_RIGHT @ M).ravel())

    def _add_landmarks(self, t, T_w2l, want):
        """addNewLandmarks: `want` fresh key points (jittered grid over the ground part of the LEFT image), BRIEF, stereo partner"""
        if want <= 0 or self.n_used + want > self.cap:
            return 0
        T_l2w_true = self.s.T_l2w[t]
        g = int(np.ceil(np.sqrt(want * 1.6)))
        uu = np.linspace(60, W - 60, g)[None, :].repeat(g, 0) + self.rng.uniform(-12, 12, (g, g))
        vv = np.linspace(40, H - 40, g)[:, None].repeat(g, 1) + self.rng.uniform(-8, 8, (g, g))
        uv = np.rint(np.stack([uu.ravel(), vv.ravel()], 1))
        uv = uv[self.rng.permutation(len(uv))[:want]]
        uv_d = torch.tensor(uv, device=self.dev)
        P, lam = self.s.ground_point(T_l2w_true, uv_d)
        good = (lam > 0.5) & (lam < 7.0)      # disparity >= 7 px: farther ground is too coarsely triangulated with this baseline
        uv_d, P = uv_d[good], P[good]
        n = uv_d.shape[0]
        if n == 0:
            return 0
        kp = torch.full((n,), 7.0, dtype=torch.float32, device=self.dev)
        roi = torch.tensor([[0.0, 0.0, float(W), float(H)]], dtype=torch.float32, device=self.dev)
        seg = torch.tensor([0, n], dtype=torch.int32, device=self.dev)
        seg_o, kp_o, desc = self.brief("left", roi, seg, uv_d.float().contiguous())
        if kp_o.shape[0] != n:      # key points near the border were dropped: keep the rest (order is preserved)
            keep = (uv_d[:, 0] >= 28) & (uv_d[:, 0] < W - 28) & (uv_d[:, 1] >= 28) & (uv_d[:, 1] < H - 28)
            uv_d, P, kp = uv_d[keep], P[keep], kp[keep]
            n = uv_d.shape[0]
            if kp_o.shape[0] != n:
                return 0
        res = self._lib(self.fm.add_new_landmarks, self.brief, uv_d.float().contiguous(), kp, desc.contiguous())
        ok = res.status == 0
        m = int(ok.sum())
        if m == 0:
            return 0
        a, b = self.n_used, self.n_used + m
        T_l2w = inv12(T_w2l)
        R = torch.tensor(T_l2w[:9].reshape(3, 3), device=self.dev)
        tt = torch.tensor(T_l2w[9:], device=self.dev)
        self.xyz[a:b] = res.xyz_left[ok] @ R.T + tt                       # LEFT -> WORLD with the ESTIMATED pose
        self.xyz_true[a:b] = P[ok]
        self.last_disp[a:b] = (res.uv_left[ok, 0] - res.uv_right[ok, 0])
        self.uv_ref[a:b] = res.uv_left[ok].double()
        self.dp_index[a:b] = len(self.dp_T)
        self.last_l[a:b], self.last_r[a:b], self.ref_l[a:b] = res.desc_left[ok], res.desc_right[ok], res.desc_left[ok]
        self.alive[a:b] = True
        self.fails[a:b] = 0
        self.dp_T.append(T_l2w)
        ids = torch.arange(a, b, device=self.dev, dtype=torch.int32)
        self.m_lm.append(ids)
        self.m_frame.append(torch.full((m,), len(self.frames_PL) - 1, dtype=torch.int32, device=self.dev))
        self.m_uvl.append(res.uv_left[ok])
        self.m_uvr.append(res.uv_right[ok])
        self.n_used = b
        self.stats["landmarks_created"] += m
        return m

    def start(self, images0):
        """frame 0: the world is its LEFT camera pose (taken from the stream's truth), landmarks from its detections"""
        T_w2l = inv12(self.s.T_l2w[0])
        self.brief.set_image("left", images0[0])
        self.brief.set_image("right", images0[1])
        self.T_w2l = self.T_prev = T_w2l
        self._push_frame(T_w2l)
        self._add_landmarks(0, T_w2l, self.target_visible + 150)
        self._key_frame(0, visible=None)
        self.stats["frames"] = 1

    def _key_frame(self, t, visible):
        """CKeyFrame: the measurements of the visible landmarks (getMeasurementsForVisibleLandmarks)"""
        n = self.n_used
        if visible is None:      # frame 0: what addNewLandmarks just measured
            ids, uvl, uvr = self.m_lm[-1], self.m_uvl[-1], self.m_uvr[-1]
        else:
            ids, uvl, uvr = visible
        disp = (uvl[:, 0] - uvr[:, 0]).double()
        zz = synth.VI_DUR_FLIPPED / disp
        xyzL = torch.stack([zz * (uvl[:, 0].double() - CX) / F, zz * (uvl[:, 1].double() - CY) / F, zz], 1)
        self.n_kf_seen[ids.long()] += 1
        self.key_frames.append((t, inv12(self.T_w2l), self.s.accel[t], ids.cpu().numpy().astype(np.int64), uvl.cpu().numpy(), uvr.cpu().numpy(),
                                xyzL.cpu().numpy()))
        self.T_last_kf = self.T_w2l
        del n

    def _optimize(self):
        """Cg2oOptimizer::optimize (:232-522) over the whole graph: new landmarks (seen in >= 2 key frames, CFundamentalMatcher.cpp:
        215), new key frames with odometry + gravity edges and admitted measurements, initializeOptimization, _optimizeUnLimited,
        write-back of landmarks and key-frame poses"""
        import time
        t0 = time.perf_counter()
        ba = self.ba
        new_lm = torch.nonzero((self.n_kf_seen[:self.n_used] >= 2) & ~self.in_graph[:self.n_used]).flatten()
        if new_lm.numel():
            ba.add_landmarks(new_lm.cpu().numpy().astype(np.int64), self.xyz[new_lm].cpu().numpy())
            self.in_graph[new_lm] = True
        for k in range(self.kf_in_graph, len(self.key_frames)):
            t, T_l2w, accel, ids, uvl, uvr, xyzL = self.key_frames[k]
            if k == 0:
                ba.add_pose(synth.POSE_ID_SHIFT, T_l2w, fixed=True)
                ba.add_edge_accel(synth.POSE_ID_SHIFT, [0.0, -1.0, 0.0], self.s.imu_off)      # Cg2oOptimizer.cpp:154
            else:
                ba.add_keyframe(synth.POSE_ID_SHIFT + k, synth.POSE_ID_SHIFT + k - 1, T_l2w, accel=accel)
            ba.add_measurements(synth.POSE_ID_SHIFT + k, ids, uvl, uvr, xyzL)
        self.kf_in_graph = len(self.key_frames)
        t1 = time.perf_counter()
        ba.initialize()
        t2 = time.perf_counter()
        nominal, executed = ba.optimize_until()
        out = ba.apply_optimization()
        kept = out["lm_kept"].astype(bool)
        ids = torch.tensor(out["lm_ids"][kept], device=self.dev)
        self.xyz[ids] = torch.tensor(out["lm_xyz"][kept], device=self.dev)
        # the tracker re-anchors on the optimised last key frame (CTrackerSVI.cpp:710-712)
        T_l2w_opt = out["kf_T"][-1]
        T_l2w_old = self.key_frames[-1][1]
        corr_R = T_l2w_opt[:9].reshape(3, 3) @ T_l2w_old[:9].reshape(3, 3).T
        corr_t = T_l2w_opt[9:] - corr_R @ T_l2w_old[9:]
        for name in ("T_w2l", "T_prev"):
            T = inv12(getattr(self, name))
            setattr(self, name, inv12(pack(corr_R @ T[:9].reshape(3, 3), corr_R @ T[9:] + corr_t)))
        for k in range(len(self.key_frames)):
            kf = self.key_frames[k]
            self.key_frames[k] = (kf[0], out["kf_T"][k]) + kf[2:]
        if out["erased"]:
            ba.initialize()
        t3 = time.perf_counter()
        self.stats["ba_calls"] += 1
        self.stats["ba_iterations"] += int(executed)
        self.stats["ba_ms"] += 1e3 * (t3 - t0)
        self.stats["ba_initialize_ms"] += 1e3 * (t2 - t1)

    def step(self, t, images):
        """one frame of CTrackerSVI::_trackLandmarks"""
        dev = self.dev
        self._lib(self.brief.set_image, "left", images[0])
        self._lib(self.brief.set_image, "right", images[1])
        n = self.n_used
        # the motion prior of CTrackerSVI::process (:330-470): rotation increment from the gyroscope (here: the stream's true
        # relative rotation plus a little noise), translation increment carried over from the last frame pair
        d = self._delta(self.T_w2l, self.T_prev)
        g = self._delta(inv12(self.s.T_l2w[t]), inv12(self.s.T_l2w[t - 1]))
        w = self.rng.normal(0, 2e-4, 3)
        Rn = np.array([[1, -w[2], w[1]], [w[2], 1, -w[0]], [-w[1], w[0], 1]])
        u_, _, vt_ = np.linalg.svd(Rn @ g[:9].reshape(3, 3))
        Rd, td = u_ @ vt_, d[9:]
        T_est = pack(Rd @ self.T_w2l[:9].reshape(3, 3), Rd @ self.T_w2l[9:] + td)
        det = self._detector(inv12(self.s.T_l2w[t]))
        dp_T = np.array(self.dp_T)
        args = (dp_T, 1.0, self.xyz[:n], self.kp_size[:n], self.last_disp[:n], self.uv_ref[:n], self.dp_index[:n])
        act = self.alive[:n].to(torch.uint8)
        plan = self._lib(self.fm.plan, T_est, *args)
        r12, pose = self._lib(self.fm.pose_stereo_posit, plan, det, self.brief, self.last_l[:n], self.last_r[:n], self.posit, self.T_w2l, np.zeros(3), T_est, act)
        if pose.status != 0:
            self.stats["posit_fail"] += 1
            T_now = T_est
        else:
            T_now = np.array(pose.T_world_to_left[:])
        ok12 = r12.status == 0
        # epipolar search with the refined pose for what the pose stage did not find
        plan2 = self._lib(self.fm.plan, T_now, *args)
        r3 = self._lib(self.fm.track_epipolar, plan2, self.brief, self.last_l[:n], self.ref_l[:n], (self.alive[:n] & ~ok12).to(torch.uint8), detector=det,
                       last_desc_right=self.last_r[:n])
        ok3 = r3.status == 0
        ok = ok12 | ok3
        uvl = torch.where(ok3[:, None], r3.uv_left, r12.uv_left)
        uvr = torch.where(ok3[:, None], r3.uv_right, r12.uv_right)
        self.last_l[:n] = torch.where(ok3[:, None], r3.desc_left, torch.where(ok12[:, None], r12.desc_left, self.last_l[:n]))
        self.last_r[:n] = torch.where(ok3[:, None], r3.desc_right, torch.where(ok12[:, None], r12.desc_right, self.last_r[:n]))
        self.last_disp[:n] = torch.where(ok, uvl[:, 0] - uvr[:, 0], self.last_disp[:n])
        tried = self.alive[:n]
        self.fails[:n] = torch.where(ok, torch.zeros_like(self.fails[:n]), self.fails[:n] + tried.int())
        self.alive[:n] &= self.fails[:n] < temporal.FundamentalMatcher.max_failed_subsequent_trackings      # CFundamentalMatcher.h:83
        self.T_prev, self.T_w2l = self.T_w2l, T_now
        self._push_frame(T_now)
        ids = torch.nonzero(ok).flatten().to(torch.int32)
        self.m_lm.append(ids)
        self.m_frame.append(torch.full((ids.numel(),), len(self.frames_PL) - 1, dtype=torch.int32, device=dev))
        self.m_uvl.append(uvl[ok])
        self.m_uvr.append(uvr[ok])
        st = self.stats
        st["frames"] += 1
        stage = r12.stage
        st["stage1"] += int((ok12 & (stage == 1)).sum())
        st["stage2"] += int((ok12 & (stage == 2)).sum()) + int((ok3 & (r3.stage == 2)).sum())
        st["stage3"] += int((ok3 & (r3.stage == 3)).sum())
        n_visible = int(ok.sum())
        if st["frames"] % 10 == 0:                      # m_uLandmarkOptimizationEveryNFrames
            self._refine_landmarks()
        # key frame?  (CTrackerSVI.h:53-54: 0.5 m^2 / 0.25 rad^2 since the last one, enough landmarks :84)
        dk = self._delta(T_now, self.T_last_kf)
        ang = np.arccos(np.clip((np.trace(dk[:9].reshape(3, 3)) - 1) / 2, -1, 1))
        if (dk[9:] @ dk[9:] > 0.5 or ang * ang > 0.25) and n_visible > 50:
            self._key_frame(t, (ids, uvl[ok], uvr[ok]))
            if len(self.key_frames) - self.kf_in_graph > 20:   # m_uIDDeltaKeyFrameForOptimization
                self._optimize()
        if n_visible < self.min_visible:
            self._add_landmarks(t, T_now, self.target_visible - n_visible + 100)
        return n_visible

    @staticmethod
    def _delta(T_now, T_prev):
        Rn, Rp = T_now[:9].reshape(3, 3), T_prev[:9].reshape(3, 3)
        Rd = Rn @ Rp.T
        return pack(Rd, T_now[9:] - Rd @ T_prev[9:])

    def _refine_landmarks(self):
        """CFundamentalMatcher::optimizeActiveLandmarks: CLandmark::optimize over every live landmark's measurement book"""
        lm = torch.cat(self.m_lm).long()
        order = torch.argsort(lm, stable=True)
        n = self.n_used
        cnt = torch.bincount(lm, minlength=n)
        seg = torch.zeros(n + 1, dtype=torch.int32, device=self.dev)
        seg[1:] = torch.cumsum(cnt, 0)
        fr = torch.cat(self.m_frame)[order].contiguous()
        ul = torch.cat(self.m_uvl)[order].contiguous()
        ur = torch.cat(self.m_uvr)[order].contiguous()
        PL = torch.tensor(np.array(self.frames_PL), device=self.dev)
        PR = torch.tensor(np.array(self.frames_PR), device=self.dev)
        out, st, err, its = self._lib(self.lmopt.optimize, PL, PR, seg, fr, ul, ur, self.xyz[:n].contiguous())
        good = ((st == 1) | (st == 2)) & self.alive[:n] & ~self.in_graph[:n]
        self.xyz[:n] = torch.where(good[:, None], out, self.xyz[:n])

    def pose_error(self, t):
        """(translation error m, rotation error deg) of the current estimate against the stream's truth for frame t"""
        T_true = inv12(self.s.T_l2w[t])
        T = self.T_w2l
        R = T[:9].reshape(3, 3) @ T_true[:9].reshape(3, 3).T
        return float(np.linalg.norm(inv12(T)[9:] - self.s.T_l2w[t][9:])), float(np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))))
