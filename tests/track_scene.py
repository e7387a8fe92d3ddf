"""Synthetic frame for the temporal-tracking tests: KITTI-00 camera, landmarks seen from a few detection points, a
slightly wrong pose estimate for the new frame, and a *synthetic image*: a pure function (side, pixel) -> BRIEF-like
256-bit descriptor that equals the landmark's current descriptor at its true pixel, degrades with distance from
it and is hash noise elsewhere.  The extractor / detector built on it stand in for OpenCV's (absent here) on BOTH
sides of the parity tests - host numpy for the oracle cascade, the same numpy behind device tensors for the product.
"""
import numpy as np

W, H = 1241, 376
FX, CX, CY = 718.856, 607.1928, 185.2157
DUR = -386.1448
P_LEFT = np.array([[FX, 0, CX, 0], [0, FX, CY, 0], [0, 0, 1, 0]], np.float64)
P_RIGHT = np.array([[FX, 0, CX, DUR], [0, FX, CY, 0], [0, 0, 1, 0]], np.float64)
K_INV = np.array([[1.0 / FX, 0, -CX / FX], [0, 1.0 / FX, -CY / FX], [0, 0, 1.0]])
BORDER = 28  # OpenCV's BRIEF drops key points closer than 28 px to the image (ROI) border


def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def inv12(T):
    R, t = T[:9].reshape(3, 3), T[9:]
    return np.concatenate([R.T.ravel(), -R.T @ t])


def pack(R, t):
    return np.concatenate([np.asarray(R, np.float64).ravel(), np.asarray(t, np.float64).ravel()])


def flip_bits(desc, nbits, seed):
    """desc (32,) uint8 with `nbits` distinct bits flipped, chosen by seed"""
    r = np.random.default_rng(seed)
    bits = r.choice(256, size=int(min(max(nbits, 0), 256)), replace=False)
    out = desc.copy()
    for b in bits:
        out[b >> 3] ^= np.uint8(1 << (b & 7))
    return out


def _hash_desc(side, px, py):
    """(N,) int pixel coordinates -> (N,32) uint8 hash noise"""
    px = np.asarray(px, np.uint64)
    py = np.asarray(py, np.uint64)
    out = np.empty((len(px), 8), np.uint32)
    for w in range(8):
        x = (px * np.uint64(73856093)) ^ (py * np.uint64(19349663)) ^ np.uint64((side + 1) * 83492791) ^ np.uint64((w + 1) * 2654435761)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xff51afd7ed558ccd)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xc4ceb9fe1a85ec53)
        x ^= x >> np.uint64(33)
        out[:, w] = (x & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return out.view(np.uint8).reshape(-1, 32)


class Scene:
    def __init__(self, n=300, n_dp=5, seed=7, pose_error=0.008, kp_sizes=(7.0, 7.0, 7.0, 7.0, 7.0, 3.5), motion_scaling=1.0):
        r = np.random.default_rng(seed)
        self.n, self.motion_scaling = n, float(motion_scaling)
        # detection points: a short forward trajectory; the new frame continues it
        self.dp_T = []
        for d in range(n_dp):
            self.dp_T.append(pack(rot(0.0, 0.01 * d, 0.0), [0.02 * d, 0.0, 0.9 * d]))
        self.dp_T = np.array(self.dp_T)
        T_true_l2w = pack(rot(0.002, 0.01 * n_dp + 0.004, -0.001), [0.02 * n_dp + 0.01, -0.005, 0.9 * n_dp + 0.1])
        self.T_true_w2l = inv12(T_true_l2w)
        T_est_l2w = pack(rot(0.002 + pose_error * 0.1, 0.01 * n_dp + 0.004 + pose_error * 0.2, -0.001),
                         [0.02 * n_dp + 0.01 + pose_error, -0.005, 0.9 * n_dp + 0.1 - pose_error])
        self.T_est_w2l = inv12(T_est_l2w)
        self.dp_index = r.integers(0, n_dp, n).astype(np.int32)
        if n_dp > 2:  # a detection point (almost) on top of the estimate: degenerate epipolar geometry
            self.dp_T[n_dp - 1] = T_est_l2w
        # landmarks: placed in the frustum of the TRUE new frame (so most are visible), some far outside
        z = np.exp(r.uniform(np.log(3.0), np.log(60.0), n))
        u = r.uniform(-80, W + 80, n)
        v = r.uniform(-40, H + 40, n)
        p_cam = np.stack([(u - CX) / FX * z, (v - CY) / FX * z, z], 1)
        behind = r.random(n) < 0.03
        p_cam[behind, 2] *= -1
        Rl2w, tl2w = T_true_l2w[:9].reshape(3, 3), T_true_l2w[9:]
        self.xyz_world = p_cam @ Rl2w.T + tl2w
        self.kp_size = r.choice(np.asarray(kp_sizes, np.float32), n).astype(np.float32)
        # reference pixel = projection into the landmark's detection frame, as float32 pixels
        uvr = np.zeros((n, 2))
        for i in range(n):
            Tw2dp = inv12(self.dp_T[self.dp_index[i]])
            pc = Tw2dp[:9].reshape(3, 3) @ self.xyz_world[i] + Tw2dp[9:]
            if abs(pc[2]) < 1e-6:
                pc[2] = 1e-6
            uvr[i] = [np.float32(FX * pc[0] / pc[2] + CX), np.float32(FX * pc[1] / pc[2] + CY)]
        wild = r.random(n) < 0.12  # broken tracks: the epipolar line has nothing to do with the landmark
        uvr[wild] = np.stack([r.uniform(0, W, wild.sum()), r.uniform(0, H, wild.sum())], 1).astype(np.float32)
        self.uv_reference = uvr
        # true pixels in the new frame
        Rt, tt = self.T_true_w2l[:9].reshape(3, 3), self.T_true_w2l[9:]
        pc = self.xyz_world @ Rt.T + tt
        zt = np.where(np.abs(pc[:, 2]) < 1e-6, 1e-6, pc[:, 2])
        self.true_uL = np.rint(FX * pc[:, 0] / zt + CX).astype(np.int64)
        self.true_v = np.rint(FX * pc[:, 1] / zt + CY).astype(np.int64)
        disp = np.maximum(np.rint(-DUR / np.maximum(zt, 0.5)), 1).astype(np.int64)
        self.true_uR = self.true_uL - disp
        self.last_disparity = (disp + r.integers(-2, 3, n)).clip(1).astype(np.float32)
        # descriptors: reference (first sighting), last (previous frame), current (this frame)
        self.ref_desc = r.integers(0, 256, (n, 32), dtype=np.uint8)
        self.last_left = np.stack([flip_bits(self.ref_desc[i], r.integers(0, 14), 1000 + i) for i in range(n)])
        self.last_right = np.stack([flip_bits(self.last_left[i], r.integers(0, 10), 2000 + i) for i in range(n)])
        far = r.random(n) < 0.08  # drifted appearance: far from the reference descriptor
        self.cur_left = np.stack([flip_bits(self.last_left[i], r.integers(0, 12) if not far[i] else 30, 3000 + i) for i in range(n)])
        for i in np.nonzero(far)[0]:
            self.last_left[i] = flip_bits(self.ref_desc[i], 95, 5000 + i)
            self.cur_left[i] = flip_bits(self.last_left[i], 8, 6000 + i)
        self.cur_right = np.stack([flip_bits(self.cur_left[i], r.integers(0, 30), 4000 + i) for i in range(n)])
        # feature maps of the synthetic images
        self.feat = [np.full((H, W), -1, np.int32), np.full((H, W), -1, np.int32)]
        self.fdist = [np.zeros((H, W), np.int32), np.zeros((H, W), np.int32)]
        for i in range(n):
            for side, uu in ((0, self.true_uL[i]), (1, self.true_uR[i])):
                vv = self.true_v[i]
                for dy in range(-2, 3):
                    for dx in range(-2, 3):
                        x, y = uu + dx, vv + dy
                        if 0 <= x < W and 0 <= y < H:
                            self.feat[side][y, x] = i
                            self.fdist[side][y, x] = max(abs(dx), abs(dy))
        # detector output: true feature pixels + distractors
        self.corners = []
        for side in (0, 1):
            uu = self.true_uL if side == 0 else self.true_uR
            ok = (uu >= 0) & (uu < W) & (self.true_v >= 0) & (self.true_v < H)
            pts = np.stack([uu[ok], self.true_v[ok]], 1)
            extra = np.stack([r.integers(0, W, 1500), r.integers(0, H, 1500)], 1)
            allp = np.unique(np.concatenate([pts, extra]), axis=0)
            order = np.lexsort((allp[:, 0], allp[:, 1]))
            self.corners.append(allp[order].astype(np.float32))

    # ---- the synthetic image --------------------------------------------------------------------
    def describe(self, side, px, py):
        px = np.asarray(px, np.int64)
        py = np.asarray(py, np.int64)
        out = _hash_desc(side, px, py)
        inside = (px >= 0) & (px < W) & (py >= 0) & (py < H)
        ids = np.full(len(px), -1, np.int32)
        ids[inside] = self.feat[side][py[inside], px[inside]]
        for k in np.nonzero(ids >= 0)[0]:
            i = ids[k]
            base = self.cur_left[i] if side == 0 else self.cur_right[i]
            out[k] = flip_bits(base, 9 * int(self.fdist[side][py[k], px[k]]), 7000 + i)
        return out

    def extract_one(self, side, roi, kp_uv):
        """OpenCV-like extractor on ONE ROI: drops key points within BORDER of the ROI border, describes the rest"""
        s = 0 if side == "left" else 1
        kp_uv = np.asarray(kp_uv, np.float32).reshape(-1, 2)
        roi = np.asarray(roi, np.float32)
        w, h = np.floor(roi[2]), np.floor(roi[3])       # cv::Rect truncates
        x, y = kp_uv[:, 0], kp_uv[:, 1]
        keep = (x >= BORDER) & (x < w - BORDER) & (y >= BORDER) & (y < h - BORDER)
        kp = kp_uv[keep]
        u0, v0 = np.floor(roi[0]), np.floor(roi[1])
        px = np.floor(kp[:, 0] + u0 + np.float32(0.5)).astype(np.int64)
        py = np.floor(kp[:, 1] + v0 + np.float32(0.5)).astype(np.int64)
        return np.ascontiguousarray(kp), self.describe(s, px, py)

    def detect_one(self, side, rect):
        s = 0 if side == "left" else 1
        c = self.corners[s]
        rect = np.asarray(rect, np.float32)
        ul = np.floor(rect[:2])
        m = (c[:, 0] >= ul[0]) & (c[:, 0] < np.floor(rect[2])) & (c[:, 1] >= ul[1]) & (c[:, 1] < np.floor(rect[3]))
        return (c[m] - ul).astype(np.float32)

    # ---- batched adapters (device tensors in, device tensors out) -------------------------------------
    def make_extractor(self, torch, device):
        def extractor(side, roi, seg, kp_uv):
            roi_h, seg_h, kp_h = roi.cpu().numpy(), seg.cpu().numpy(), kp_uv.cpu().numpy()
            kps, descs, new_seg = [], [], [0]
            for i in range(len(roi_h)):
                k, d = self.extract_one(side, roi_h[i], kp_h[seg_h[i]:seg_h[i + 1]])
                kps.append(k)
                descs.append(d)
                new_seg.append(new_seg[-1] + len(k))
            kp_all = np.concatenate(kps).astype(np.float32) if kps else np.zeros((0, 2), np.float32)
            d_all = np.concatenate(descs).astype(np.uint8) if descs else np.zeros((0, 32), np.uint8)
            return (torch.tensor(np.asarray(new_seg, np.int32), device=device), torch.tensor(kp_all.reshape(-1, 2), device=device),
                    torch.tensor(d_all.reshape(-1, 32), device=device))
        return extractor

    def make_detector(self, torch, device):
        def detector(side, rect):
            rect_h = rect.cpu().numpy()
            pts, seg = [], [0]
            for i in range(len(rect_h)):
                p = self.detect_one(side, rect_h[i])
                pts.append(p)
                seg.append(seg[-1] + len(p))
            allp = np.concatenate(pts).astype(np.float32) if pts else np.zeros((0, 2), np.float32)
            return torch.tensor(np.asarray(seg, np.int32), device=device), torch.tensor(allp.reshape(-1, 2), device=device)
        return detector

    def stereo_dict(self):
        duf = -DUR
        return dict(f=FX, cx=CX, cy=CY, duR_flipped=duf, min_disparity=0.01, depth_min=duf / W, depth_max=duf / 0.01, width=W)


class Sequence:
    """A short drive past a fixed landmark cloud: frame(t) is a Scene-like object (same synthetic-image interface) for
    the true pose of frame t, with per-frame appearance noise on the landmark descriptors."""

    def __init__(self, n=400, n_frames=8, seed=3, step=0.7):
        r = np.random.default_rng(seed)
        self.n, self.n_frames = n, n_frames
        self.T_l2w = [pack(rot(0.001 * t, 0.006 * t, -0.0005 * t), [0.015 * t, -0.002 * t, step * t]) for t in range(n_frames)]
        # landmarks in front of frame 0, deep enough to stay visible for a while
        z = np.exp(r.uniform(np.log(6.0), np.log(60.0), n)) + step * n_frames * 0.5
        u, v = r.uniform(60, W - 60, n), r.uniform(50, H - 50, n)
        self.xyz_world = np.stack([(u - CX) / FX * z, (v - CY) / FX * z, z], 1)
        self.base_desc = r.integers(0, 256, (n, 32), dtype=np.uint8)
        self.kp_size = np.full(n, 7.0, np.float32)
        self._r = r

    def frame(self, t):
        f = Scene.__new__(Scene)
        f.n, f.motion_scaling = self.n, 1.0
        T = inv12(self.T_l2w[t])
        f.T_true_w2l = T
        pc = self.xyz_world @ T[:9].reshape(3, 3).T + T[9:]
        zt = np.maximum(pc[:, 2], 0.5)
        f.true_uL = np.rint(FX * pc[:, 0] / zt + CX).astype(np.int64)
        f.true_v = np.rint(FX * pc[:, 1] / zt + CY).astype(np.int64)
        disp = np.maximum(np.rint(-DUR / zt), 1).astype(np.int64)
        f.true_uR = f.true_uL - disp
        f.true_disparity = disp.astype(np.float32)
        f.cur_left = np.stack([flip_bits(self.base_desc[i], 3 + (7 * t + i) % 5, 10000 * t + i) for i in range(self.n)])
        f.cur_right = np.stack([flip_bits(f.cur_left[i], (3 * t + i) % 6, 20000 * t + i) for i in range(self.n)])
        f.feat = [np.full((H, W), -1, np.int32), np.full((H, W), -1, np.int32)]
        f.fdist = [np.zeros((H, W), np.int32), np.zeros((H, W), np.int32)]
        for i in range(self.n):
            if pc[i, 2] <= 1.0:
                continue
            for side, uu in ((0, f.true_uL[i]), (1, f.true_uR[i])):
                for dy in range(-2, 3):
                    for dx in range(-2, 3):
                        x, y = uu + dx, f.true_v[i] + dy
                        if 0 <= x < W and 0 <= y < H:
                            f.feat[side][y, x] = i
                            f.fdist[side][y, x] = max(abs(dx), abs(dy))
        rr = np.random.default_rng(500 + t)
        f.corners = []
        for side in (0, 1):
            uu = f.true_uL if side == 0 else f.true_uR
            ok = (uu >= 0) & (uu < W) & (f.true_v >= 0) & (f.true_v < H) & (pc[:, 2] > 1.0)
            pts = np.stack([uu[ok], f.true_v[ok]], 1)
            extra = np.stack([rr.integers(0, W, 800), rr.integers(0, H, 800)], 1)
            allp = np.unique(np.concatenate([pts, extra]), axis=0)
            f.corners.append(allp[np.lexsort((allp[:, 0], allp[:, 1]))].astype(np.float32))
        return f
