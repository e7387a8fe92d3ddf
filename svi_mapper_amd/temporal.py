"""Host-side mirror of the reference's temporal tracking schedule over the C ABI.

Reference shapes mirrored here (src/core/CFundamentalMatcher.{h,cpp})
  getPoseStereoPosit stage 1 / stage 2        :368-733      -> FundamentalMatcher.track_stage1 / track_stage2
  trackEpipolar (stage 3)                     :794-1030     -> FundamentalMatcher.track_epipolar
  trackManual (stages 1 -> 2 -> 3)            :1366-2019    -> FundamentalMatcher.track_manual
  addNewLandmarks                             :109-175      -> FundamentalMatcher.add_new_landmarks
  _getMatchSampleRecursiveU/V                 :2142-2334    -> FundamentalMatcher.epipolar_samples
  _getMatch                                   :2336-2397    -> FundamentalMatcher.get_match
  _addMeasurementToLandmarkLEFT               :2400-2450    -> the stereo half of track_epipolar
  CTriangulator::getPointTriangulatedInRIGHT / InLEFT  src/core/CTriangulator.cpp:185-324
                                                            -> stereo_range / stereo_candidates / stereo_verify

The reference walks one landmark at a time and uses exceptions for control flow; here every step runs for all
landmarks of a frame in one launch and the exceptions become per-landmark status codes (SVI_TRK_MATCH_*).
BRIEF extraction / GFTT detection are OpenCV's and stay with the caller: `extractor(side, roi, seg, kp_uv)` is
called with device tensors (roi n x 4 f32, seg n+1 i32, kp_uv total x 2 f32 in ROI coordinates) and returns
(seg', kp_uv', desc') - the key points it kept (OpenCV drops those too close to the ROI border) and their
32-byte descriptors.  All tensors are torch CUDA tensors; only data_ptr() crosses the boundary and torch itself is
used for boolean masks, index lists / gathers and three exactly rounded float32 expressions (4s, 8s+1, kp + 4s).
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from ._capi import LandmarkParams, PositParams, PositResult
from ._capi import (MATCH_OK, MATCH_SKIPPED, TRACK_RECORD_FIELDS, TRACK_RECORD_SIZE, TRK_EPI_NO_MOTION, TRK_EPI_OK, TRK_FOV_LEFT,
                    TRK_FOV_RIGHT, TrackCamera, TrackStereoParams, check)
from .matcher import HammingMatcher

RECORD_DTYPE = np.dtype(TRACK_RECORD_FIELDS)


def _p(t):
    return None if t is None else t.data_ptr()


class StereoCamera:
    """The numbers CStereoCamera / CPinholeCamera hold that the schedule reads (src/vision/CPinholeCamera.h:20-61,
    src/vision/CStereoCamera.h, src/core/CTriangulator.cpp:14-21)."""

    def __init__(self, P_left, P_right, width, height):
        self.P_left = np.asarray(P_left, np.float64).reshape(3, 4)
        self.P_right = np.asarray(P_right, np.float64).reshape(3, 4)
        self.width, self.height = float(width), float(height)
        K = self.P_left[:, :3]
        if K[0, 1] == 0 and K[1, 0] == 0 and K[2, 0] == 0 and K[2, 1] == 0 and K[2, 2] == 1:
            fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
            self.K_inv = np.array([[1.0 / fx, 0, -cx / fx], [0, 1.0 / fy, -cy / fy], [0, 0, 1.0]])
        else:
            self.K_inv = np.linalg.inv(K)
        self.f, self.cx, self.cy = K[0, 0], K[0, 2], K[1, 2]      # m_dFx, m_dPu, m_dPv  (CTriangulator.cpp:14-17)
        self.duR_flipped = -self.P_right[0, 3]                      # m_dDuRFlipped
        self.min_disparity = 0.01                                   # CTriangulator.h:21
        self.depth_min = self.duR_flipped / self.width              # CTriangulator.cpp:20
        self.depth_max = self.duR_flipped / self.min_disparity      # CTriangulator.cpp:21

    def c_struct(self):
        c = TrackCamera()
        c.P_left[:] = self.P_left.ravel().tolist()
        c.P_right[:] = self.P_right.ravel().tolist()
        c.K_inv[:] = np.asarray(self.K_inv, np.float64).ravel().tolist()
        c.width, c.height = self.width, self.height
        return c


class TrackPlan:
    """Device records of one svi_track_plan_dev call."""

    def __init__(self, records, seg, total, kp_size):
        self.records, self.seg, self.total, self.kp_size = records, seg, int(total), kp_size
        self.n = records.shape[0]

    def host(self):
        """numpy structured view of the records (synchronises)."""
        return self.records.cpu().numpy().view(RECORD_DTYPE).reshape(-1)

    def status(self):
        """device int32 view of the status words"""
        return self.records.view(torch.int32)[:, TRACK_RECORD_SIZE // 4 - 1]


class StageResult:
    """Outcome of one stage for n landmarks: status (SVI_TRK_MATCH_*, SKIPPED where the stage did not run),
    the measurement (uv_left, uv_right, xyz_left) and the descriptors found (desc_left, desc_right)."""

    def __init__(self, n, device):
        self.status = torch.full((n,), MATCH_SKIPPED, dtype=torch.int32, device=device)
        self.uv_left = torch.zeros((n, 2), dtype=torch.float32, device=device)
        self.uv_right = torch.zeros((n, 2), dtype=torch.float32, device=device)
        self.xyz_left = torch.zeros((n, 3), dtype=torch.float64, device=device)
        self.desc_left = torch.zeros((n, 32), dtype=torch.uint8, device=device)
        self.desc_right = torch.zeros((n, 32), dtype=torch.uint8, device=device)

    def ok(self):
        return self.status == MATCH_OK


class FundamentalMatcher:
    """CFundamentalMatcher's landmark schedule on the MI355X (constants: CFundamentalMatcher.cpp:22-25, .h:83-95)."""

    cutoff_stage1 = 25
    cutoff_stage2 = 50
    cutoff_stage3 = 50
    cutoff_original = 100          # 2 * stage 3
    recursion_limit = 2
    recursion_step = 2
    max_failed_subsequent_trackings = 5

    def __init__(self, camera, device=0, matcher=None):
        self.camera = camera
        self.matcher = matcher or HammingMatcher(device)
        self._lib = _capi.load_library()
        self._h = self.matcher._h
        self._cam = camera.c_struct()
        self.device = torch.device("cuda", device)
        # the library launches on the matcher's own stream: order torch's work on the current stream around it
        self._ext = torch.cuda.ExternalStream(self.matcher.stream, device=self.device)
        self._dummy = torch.zeros(64, dtype=torch.uint8, device=self.device)

    # ------------------------------------------------------------------------------------------
    def _enter(self):
        self._ext.wait_stream(torch.cuda.current_stream(self.device))

    def _leave(self):
        torch.cuda.current_stream(self.device).wait_stream(self._ext)

    def _empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _q(self, t):
        """device pointer of a pool-like tensor; an EMPTY tensor has a null data_ptr in torch, the C ABI wants an address
        (it reads nothing from it: the segment table says the pools are empty)"""
        if t is None:
            return None
        return t.data_ptr() if t.numel() else self._dummy.data_ptr()

    # ---- plan ----------------------------------------------------------------------------------
    def plan(self, T_world_to_left, dp_T_left_to_world, motion_scaling, xyz_world, kp_size, last_disparity, uv_reference, dp_index):
        """Projection, FoV gate, stage-1/2 rectangles and the clipped epipolar segment of every landmark."""
        T = np.ascontiguousarray(T_world_to_left, np.float64).reshape(12)
        dp = np.ascontiguousarray(dp_T_left_to_world, np.float64).reshape(-1, 12)
        n = xyz_world.shape[0]
        for t, dt in ((xyz_world, torch.float64), (kp_size, torch.float32), (last_disparity, torch.float32),
                      (uv_reference, torch.float64), (dp_index, torch.int32)):
            if t.dtype != dt or not t.is_cuda or not t.is_contiguous():
                raise ValueError("plan: inputs must be contiguous CUDA tensors of the documented dtypes")
        records = self._empty((n, TRACK_RECORD_SIZE), torch.uint8)
        seg = self._empty((n + 1,), torch.int32)
        total = C.c_int64(0)
        self._enter()
        check(self._lib.svi_track_plan_dev(self._h, C.byref(self._cam), T.ctypes.data_as(_capi.f64p), dp.ctypes.data_as(_capi.f64p) if len(dp) else None,
                                           len(dp), float(motion_scaling), _p(xyz_world), _p(kp_size), _p(last_disparity), _p(uv_reference),
                                           _p(dp_index), n, _p(records), _p(seg), C.byref(total)), "svi_track_plan_dev")
        self._leave()
        return TrackPlan(records, seg, total.value, kp_size)

    # ---- stage 3 sampling ------------------------------------------------------------------------
    def epipolar_samples(self, plan, depth, sel=None):
        """(seg, sample_uv, roi) of recursion depth `depth` for all landmarks (sel None) or the subset sel."""
        if sel is None:
            seg, n_sel, total = plan.seg, plan.n, plan.total
        else:
            sel = sel.to(torch.int32).contiguous()
            n_sel = sel.numel()
            cnt = (plan.seg[1:] - plan.seg[:-1])[sel.long()]
            seg = torch.zeros(n_sel + 1, dtype=torch.int32, device=self.device)
            seg[1:] = torch.cumsum(cnt, 0)
            total = int(seg[-1].item()) if n_sel else 0
        sample_uv = self._empty((total, 2), torch.float32)
        roi = self._empty((n_sel, 4), torch.float32)
        self._enter()
        check(self._lib.svi_track_epipolar_samples_dev(self._h, C.byref(self._cam), _p(plan.records), _p(plan.kp_size), _p(sel), n_sel, _p(seg),
                                                       int(depth), self._q(sample_uv), self._q(roi)), "svi_track_epipolar_samples_dev")
        self._leave()
        return seg, sample_uv, roi

    # ---- _getMatch -----------------------------------------------------------------------------
    def get_match(self, ref, original, seg, pool, cutoff, cutoff_original=None, active=None):
        """(idx, dist, status): idx is the index inside the landmark's segment (cv::DMatch::trainIdx) or -1."""
        nq = ref.shape[0]
        idx = self._empty((nq,), torch.int32)
        dist = self._empty((nq,), torch.int32)
        status = self._empty((nq,), torch.int32)
        self._enter()
        check(self._lib.svi_match_ragged_dev(self._h, _p(ref), _p(original), _p(active), nq, _p(seg), self._q(pool), int(cutoff),
                                             int(cutoff_original if cutoff_original is not None else 257), _p(idx), _p(dist), _p(status)),
              "svi_match_ragged_dev")
        self._leave()
        return idx, dist, status

    # ---- stereo search -----------------------------------------------------------------------------
    def handover(self, mode, plan, sel=None, seg=None, pool_uv=None, idx=None, roi=None):
        n_sel = plan.n if sel is None else sel.numel()
        uv_ref = self._empty((n_sel, 2), torch.float32)
        topleft = self._empty((n_sel, 2), torch.float32)
        ok = self._empty((n_sel,), torch.uint8)
        self._enter()
        check(self._lib.svi_track_handover_dev(self._h, int(mode), _p(plan.records), _p(plan.kp_size), _p(sel), n_sel, _p(seg), self._q(pool_uv),
                                               _p(idx), self._q(roi), _p(uv_ref), _p(topleft), _p(ok)), "svi_track_handover_dev")
        self._leave()
        return uv_ref, topleft, ok

    def stereo_range(self, search_in_left, uv_ref, topleft, kp_size, search_range=None, active=None):
        n = uv_ref.shape[0]
        seg = self._empty((n + 1,), torch.int32)
        status = self._empty((n,), torch.int32)
        roi = self._empty((n, 4), torch.float32)
        total = C.c_int64(0)
        self._enter()
        check(self._lib.svi_track_stereo_range_dev(self._h, self.camera.width, int(search_in_left), _p(uv_ref), _p(topleft), _p(kp_size),
                                                   _p(search_range), _p(active), n, _p(seg), _p(status), _p(roi), C.byref(total)),
              "svi_track_stereo_range_dev")
        self._leave()
        return seg, status, roi, total.value

    def stereo_candidates(self, search_in_left, kp_size, seg, total):
        pool_uv = self._empty((total, 2), torch.float32)
        self._enter()
        check(self._lib.svi_track_stereo_candidates_dev(self._h, int(search_in_left), _p(kp_size), kp_size.shape[0], _p(seg), self._q(pool_uv)),
              "svi_track_stereo_candidates_dev")
        self._leave()
        return pool_uv

    def stereo_params(self, search_in_left, cutoff_other, other_inclusive):
        c = self.camera
        return TrackStereoParams(c.f, c.cx, c.cy, c.duR_flipped, c.min_disparity, c.depth_min, c.depth_max, 100, int(cutoff_other),
                                 int(other_inclusive), int(search_in_left))

    def stereo_verify(self, params, ref, last_other, uv_ref, topleft, seg, pool, pool_uv, active=None):
        nq = ref.shape[0]
        idx = self._empty((nq,), torch.int32)
        dist = self._empty((nq,), torch.int32)
        status = self._empty((nq,), torch.int32)
        uv_other = self._empty((nq, 2), torch.float32)
        xyz = self._empty((nq, 3), torch.float64)
        self._enter()
        check(self._lib.svi_track_stereo_verify_dev(self._h, C.byref(params), _p(ref), _p(last_other), _p(active), _p(uv_ref), _p(topleft), nq,
                                                    _p(seg), self._q(pool), self._q(pool_uv), _p(idx), _p(dist), _p(status), _p(uv_other), _p(xyz)),
              "svi_track_stereo_verify_dev")
        self._leave()
        return idx, dist, status, uv_other, xyz

    # ---- the stereo half shared by all stages -----------------------------------------------------
    def _stereo(self, extractor, search_in_left, kp_size, search_range, ref_desc, last_other, uv_ref, topleft, active, cutoff_other,
                other_inclusive):
        seg, st_range, roi, total = self.stereo_range(search_in_left, uv_ref, topleft, kp_size, search_range, active)
        pool_uv = self.stereo_candidates(search_in_left, kp_size, seg, total)
        seg2, pool_uv2, pool = extractor("left" if search_in_left else "right", roi, seg, pool_uv)
        run = (st_range == MATCH_OK).to(torch.uint8)
        prm = self.stereo_params(search_in_left, cutoff_other, other_inclusive)
        idx, dist, status, uv_other, xyz = self.stereo_verify(prm, ref_desc, last_other, uv_ref, topleft, seg2, pool, pool_uv2, run)
        status = torch.where(st_range != MATCH_OK, st_range, status)
        won = (seg2[:-1] + idx.clamp(min=0)).long()
        desc_other = pool[won.clamp(max=max(pool.shape[0] - 1, 0))] if pool.shape[0] else torch.zeros((ref_desc.shape[0], 32), dtype=torch.uint8, device=self.device)
        return status, uv_other, xyz, desc_other

    @staticmethod
    def _store(res, rows, status, uv_l, uv_r, xyz, d_l, d_r):
        res.status[rows] = status
        good = status == MATCH_OK
        g = rows[good]
        res.uv_left[g] = uv_l[good]
        res.uv_right[g] = uv_r[good]
        res.xyz_left[g] = xyz[good]
        res.desc_left[g] = d_l[good]
        res.desc_right[g] = d_r[good]

    # ---- stage 1 (:391-486) ------------------------------------------------------------------------
    def track_stage1(self, plan, extractor, last_desc_left, last_desc_right, active=None):
        """Descriptor at the projected pixel, LEFT first then RIGHT, followed by the stereo search in the other
        image.  Returns a StageResult over all plan.n landmarks; rows that fail keep their last failure code."""
        n = plan.n
        res = StageResult(n, self.device)
        st = plan.status()
        both = ((st & TRK_FOV_LEFT) != 0) & ((st & TRK_FOV_RIGHT) != 0)              # :389
        run = both if active is None else both & active.bool()
        todo = torch.nonzero(run).flatten()
        rec32 = plan.records.view(torch.float32)
        for side in (0, 1):
            if todo.numel() == 0:
                break
            sel = todo.to(torch.int32)
            kp = plan.kp_size[todo]
            # one key point at (4s, 4s) of the (8s+1)^2 ROI around the projection (:395-400 / :449-454)
            uv_ref, topleft, _ = self.handover(side, plan, sel)
            col = (RECORD_DTYPE.fields["s1_roi_left" if side == 0 else "s1_roi_right"][1]) // 4
            roi_xy = rec32[todo][:, col:col + 2]
            side_len = 8 * kp + 1
            roi = torch.stack([roi_xy[:, 0], roi_xy[:, 1], side_len, side_len], 1).contiguous()
            seg1 = torch.arange(todo.numel() + 1, dtype=torch.int32, device=self.device)
            kp_uv = torch.stack([4 * kp, 4 * kp], 1).contiguous()
            seg_e, kp_e, desc_e = extractor("left" if side == 0 else "right", roi, seg1, kp_uv)
            last_here = (last_desc_left if side == 0 else last_desc_right)[todo].contiguous()
            last_there = (last_desc_right if side == 0 else last_desc_left)[todo].contiguous()
            idx, dist, status = self.get_match(last_here, None, seg_e, desc_e, self.cutoff_stage1)     # :404 / :453
            found = status == MATCH_OK
            pos = (seg_e[:-1] + idx.clamp(min=0)).long().clamp(max=max(desc_e.shape[0] - 1, 0))
            desc_here = desc_e[pos] if desc_e.shape[0] else torch.zeros((todo.numel(), 32), dtype=torch.uint8, device=self.device)
            rng = rec32[todo][:, RECORD_DTYPE.fields["search_range"][1] // 4].contiguous()
            s_status, uv_other, xyz, desc_other = self._stereo(extractor, side, kp.contiguous(), rng, desc_here.contiguous(), last_there, uv_ref,
                                                               topleft, found.to(torch.uint8), self.cutoff_stage1, 1)     # :423 / :473
            status = torch.where(found, s_status, status)
            # the measurement keeps the PROJECTED pixel of the image the descriptor was found in (:427 / :477)
            uvL = plan.records.view(torch.float32)[todo][:, RECORD_DTYPE.fields["uv_left"][1] // 4:][:, :2]
            uvR = plan.records.view(torch.float32)[todo][:, RECORD_DTYPE.fields["uv_right"][1] // 4:][:, :2]
            if side == 0:
                self._store(res, todo, status, uvL, uv_other, xyz, desc_here, desc_other)
            else:
                self._store(res, todo, status, uv_other, uvR, xyz, desc_other, desc_here)
            todo = todo[status != MATCH_OK]
        return res

    # ---- stage 2 (:489-709, :1042-1290) --------------------------------------------------------------
    def track_stage2(self, plan, detector, extractor, last_desc_left, last_desc_right, active=None):
        """Regional search: `detector(side, rect)` (rect n x 4 f32: the search rectangle corners) returns the ragged
        key points it found (seg, kp_uv in search-rectangle coordinates); they are shifted by (4s,4s) and described
        inside the grown rectangle (:533-534), matched against the last descriptor (cut-off 50) and verified in
        the other image."""
        n = plan.n
        res = StageResult(n, self.device)
        st = plan.status()
        both = ((st & TRK_FOV_LEFT) != 0) & ((st & TRK_FOV_RIGHT) != 0)
        run = both if active is None else both & active.bool()
        todo = torch.nonzero(run).flatten()
        rec32 = plan.records.view(torch.float32)
        for side in (0, 1):
            if todo.numel() == 0:
                break
            sel = todo.to(torch.int32)
            kp = plan.kp_size[todo].contiguous()
            name = "left" if side == 0 else "right"
            c_rect = RECORD_DTYPE.fields["s2_" + name][1] // 4
            c_ext = RECORD_DTYPE.fields["s2_ext_" + name][1] // 4
            rect = rec32[todo][:, c_rect:c_rect + 4].contiguous()
            corners = torch.round(rec32[todo][:, c_ext:c_ext + 4])   # cv::Rect( Point2f, Point2f ): corners are cvRound()ed
            ext = torch.stack([corners[:, 0], corners[:, 1], corners[:, 2] - corners[:, 0], corners[:, 3] - corners[:, 1]], 1).contiguous()
            seg_d, kp_d = detector(name, rect)
            owner = torch.repeat_interleave(torch.arange(todo.numel(), device=self.device), (seg_d[1:] - seg_d[:-1]).long())
            kp_shift = (kp_d + (4 * kp)[owner][:, None]).contiguous()                                   # :533
            seg_e, kp_e, desc_e = extractor(name, ext, seg_d, kp_shift)
            last_here = (last_desc_left if side == 0 else last_desc_right)[todo].contiguous()
            last_there = (last_desc_right if side == 0 else last_desc_left)[todo].contiguous()
            idx, dist, status = self.get_match(last_here, None, seg_e, desc_e, self.cutoff_stage2)     # :540-545
            uv_ref, topleft, ok = self.handover(2 + side, plan, sel, seg_e, kp_e, idx)
            found = (status == MATCH_OK) & ok.bool()
            status = torch.where((status == MATCH_OK) & ~ok.bool(), torch.full_like(status, _capi.MATCH_RANGE), status)   # "out of tracking range"
            pos = (seg_e[:-1] + idx.clamp(min=0)).long().clamp(max=max(desc_e.shape[0] - 1, 0))
            desc_here = desc_e[pos] if desc_e.shape[0] else torch.zeros((todo.numel(), 32), dtype=torch.uint8, device=self.device)
            rng = rec32[todo][:, RECORD_DTYPE.fields["search_range"][1] // 4].contiguous()
            s_status, uv_other, xyz, desc_other = self._stereo(extractor, side, kp, rng, desc_here.contiguous(), last_there, uv_ref, topleft,
                                                               found.to(torch.uint8), self.cutoff_stage2, 0)               # :573 / :691
            status = torch.where(found, s_status, status)
            if side == 0:
                self._store(res, todo, status, uv_ref, uv_other, xyz, desc_here, desc_other)
            else:
                self._store(res, todo, status, uv_other, uv_ref, xyz, desc_other, desc_here)
            todo = todo[status != MATCH_OK]
        return res

    # ---- stage 3 (:847-1030) ---------------------------------------------------------------------------
    def track_epipolar(self, plan, extractor, last_desc_left, ref_desc_left, active=None):
        """Sampling along the clipped epipolar line (depth 0, then 2 for the landmarks that found nothing), _getMatch
        with the relative (50) and original (100) cut-offs, then _addMeasurementToLandmarkLEFT: the stereo search in
        RIGHT without a descriptor check.  Landmarks without SVI_TRK_EPI_OK report SKIPPED."""
        n = plan.n
        res = StageResult(n, self.device)
        st = plan.status()
        run = (st & TRK_EPI_OK) != 0
        if active is not None:
            run = run & active.bool()
        todo = torch.nonzero(run).flatten()
        depth = 0
        rec32 = plan.records.view(torch.float32)
        while todo.numel() > 0:
            sel = todo.to(torch.int32)
            seg, sample_uv, roi = self.epipolar_samples(plan, depth, sel)
            seg_e, kp_e, desc_e = extractor("left", roi, seg, sample_uv)
            ref = last_desc_left[todo].contiguous()
            orig = ref_desc_left[todo].contiguous()
            idx, dist, status = self.get_match(ref, orig, seg_e, desc_e, self.cutoff_stage3, self.cutoff_original)
            found = status == MATCH_OK
            uv_ref, topleft, ok = self.handover(4, plan, sel, seg_e, kp_e, idx, roi)
            pos = (seg_e[:-1] + idx.clamp(min=0)).long().clamp(max=max(desc_e.shape[0] - 1, 0))
            desc_here = desc_e[pos] if desc_e.shape[0] else torch.zeros((todo.numel(), 32), dtype=torch.uint8, device=self.device)
            kp = plan.kp_size[todo].contiguous()
            rng = rec32[todo][:, RECORD_DTYPE.fields["search_range"][1] // 4].contiguous()
            s_status, uv_other, xyz, desc_other = self._stereo(extractor, 0, kp, rng, desc_here.contiguous(), None, uv_ref, topleft,
                                                               found.to(torch.uint8), -1, 0)
            final = torch.where(found, s_status, status)
            self._store(res, todo, final, uv_ref, uv_other, xyz, desc_here, desc_other)
            # only an internal "no match" recurses (:2221-2233); a stereo failure after a match is final
            if depth >= self.recursion_limit:
                break
            todo = todo[~found]
            depth += self.recursion_step
        return res

    # ---- trackManual (:1366-2019): stage 1 -> stage 2 -> epipolar, each only for what the previous one lost ----------
    def track_manual(self, plan, detector, extractor, last_desc_left, last_desc_right, ref_desc_left, active=None):
        """One StageResult; `stage` (int8: 1, 2, 3, 0 = none) tells which stage produced each measurement.  A landmark
        outside the field of view of either camera is not tracked at all (:1415, :2008-2012)."""
        r1 = self.track_stage1(plan, extractor, last_desc_left, last_desc_right, active)
        tried = r1.status != MATCH_SKIPPED
        lost1 = tried & (r1.status != MATCH_OK)
        r2 = self.track_stage2(plan, detector, extractor, last_desc_left, last_desc_right, lost1.to(torch.uint8))
        lost2 = lost1 & (r2.status != MATCH_OK)
        r3 = self.track_epipolar(plan, extractor, last_desc_left, ref_desc_left, lost2.to(torch.uint8))
        out = StageResult(plan.n, self.device)
        out.stage = torch.zeros(plan.n, dtype=torch.int8, device=self.device)
        out.status = torch.where(tried, r1.status, out.status)
        for k, r in ((1, r1), (2, r2), (3, r3)):
            ran = r.status != MATCH_SKIPPED
            out.status = torch.where(ran, r.status, out.status)
            good = r.status == MATCH_OK
            for name in ("uv_left", "uv_right", "xyz_left", "desc_left", "desc_right"):
                getattr(out, name)[good] = getattr(r, name)[good]
            out.stage[good] = k
        return out

    # ---- addNewLandmarks (:109-175) ------------------------------------------------------------------------------------
    def add_new_landmarks(self, extractor, uv_left, kp_size, desc_left):
        """Stereo partner + triangulation of freshly detected key points: getPointTriangulatedInRIGHTFull with the search
        window of CTriangulator::fMinimumSearchRangePixels = 60 (no depth gate, no second descriptor check).
        uv_left n x 2 f32, kp_size n f32, desc_left n x 32 u8 (the detector / extractor output on the LEFT image)."""
        n = uv_left.shape[0]
        res = StageResult(n, self.device)
        half = 4 * kp_size
        topleft = torch.stack([torch.clamp_min(uv_left[:, 0] - 60.0 - half, 0.0), uv_left[:, 1] - half], 1).contiguous()   # :119-120
        prm = self.stereo_params(0, -1, 0)
        prm.depth_min, prm.depth_max = -1.0e300, 1.0e300
        seg, st_range, roi, total = self.stereo_range(0, uv_left.contiguous(), topleft, kp_size.contiguous())
        pool_uv = self.stereo_candidates(0, kp_size.contiguous(), seg, total)
        seg2, pool_uv2, pool = extractor("right", roi, seg, pool_uv)
        run = (st_range == MATCH_OK).to(torch.uint8)
        idx, dist, status, uv_other, xyz = self.stereo_verify(prm, desc_left.contiguous(), None, uv_left.contiguous(), topleft, seg2, pool, pool_uv2, run)
        status = torch.where(st_range != MATCH_OK, st_range, status)
        won = (seg2[:-1] + idx.clamp(min=0)).long()
        desc_other = pool[won.clamp(max=max(pool.shape[0] - 1, 0))] if pool.shape[0] else torch.zeros((n, 32), dtype=torch.uint8, device=self.device)
        self._store(res, torch.arange(n, device=self.device), status, uv_left, uv_other, xyz, desc_left, desc_other)
        return res


class PoseOptimizationError(Exception):
    """CExceptionPoseOptimization (src/exceptions/CExceptionPoseOptimization.h)"""

    def __init__(self, message, result):
        super().__init__(message)
        self.result = result


class SolverStereoPosit:
    """CSolverStereoPosit (src/optimization/CSolverStereoPosit.{h,cpp}): frame pose from the tracked measurements,
    the whole re-weighted Gauss-Newton loop in one launch."""

    _messages = {1: "insufficient number of points", 2: "system did not converge", 3: "insufficient accuracy",
                 4: "inconsistent with prior (HIGH RISK)"}

    def __init__(self, P_left, P_right, matcher=None, device=0):
        self.matcher = matcher or HammingMatcher(device)
        self._lib = _capi.load_library()
        self.params = PositParams()
        self._lib.svi_posit_params_default(C.byref(self.params))
        self.params.P_left[:] = np.asarray(P_left, np.float64).ravel().tolist()
        self.params.P_right[:] = np.asarray(P_right, np.float64).ravel().tolist()
        self.device = torch.device("cuda", device)
        self._ext = torch.cuda.ExternalStream(self.matcher.stream, device=self.device)

    def solve(self, T_world_to_left_last, t_imu, T_world_to_left_estimate, xyz_world, uv_left, uv_right, active=None):
        """-> PositResult (status 0) ; device tensors xyz_world n x 3 f64, uv_* n x 2 f32, active u8 or None"""
        Tl = np.ascontiguousarray(T_world_to_left_last, np.float64).reshape(12)
        Te = np.ascontiguousarray(T_world_to_left_estimate, np.float64).reshape(12)
        ti = np.ascontiguousarray(t_imu, np.float64).reshape(3)
        res = PositResult()
        self._ext.wait_stream(torch.cuda.current_stream(self.device))
        check(self._lib.svi_stereo_posit_dev(self.matcher._h, C.byref(self.params), Tl.ctypes.data_as(_capi.f64p), ti.ctypes.data_as(_capi.f64p),
                                             Te.ctypes.data_as(_capi.f64p), _p(xyz_world), _p(uv_left), _p(uv_right), _p(active),
                                             int(xyz_world.shape[0]), C.byref(res)), "svi_stereo_posit_dev")
        return res

    def get_transformation_world_to_left(self, T_last, t_imu, T_estimate, xyz_world, uv_left, uv_right, active=None):
        """getTransformationWORLDtoLEFT: the refined 12-vector, or CExceptionPoseOptimization as the reference throws"""
        r = self.solve(T_last, t_imu, T_estimate, xyz_world, uv_left, uv_right, active)
        if r.status != 0:
            raise PoseOptimizationError(self._messages.get(r.status, "failed"), r)
        return np.array(r.T_world_to_left[:])


class LandmarkOptimizer:
    """CLandmark::optimize for all active landmarks of a frame (src/types/CLandmark.cpp:281-296, 447-581)."""

    SKIPPED, OPTIMAL, CONVERGED, REJECTED, NOT_CONVERGED = range(5)

    def __init__(self, matcher=None, device=0):
        self.matcher = matcher or HammingMatcher(device)
        self._lib = _capi.load_library()
        self.params = LandmarkParams()
        self._lib.svi_landmark_params_default(C.byref(self.params))
        self.device = torch.device("cuda", device)
        self._ext = torch.cuda.ExternalStream(self.matcher.stream, device=self.device)

    def optimize(self, frame_P_left, frame_P_right, meas_seg, meas_frame, meas_uv_left, meas_uv_right, xyz):
        """-> (xyz_optimized, status, error_average, iterations); bIsOptimal = status in (SKIPPED, OPTIMAL)"""
        n = xyz.shape[0]
        out = torch.empty_like(xyz)
        status = torch.empty((n,), dtype=torch.int32, device=self.device)
        err = torch.empty((n,), dtype=torch.float64, device=self.device)
        its = torch.empty((n,), dtype=torch.int32, device=self.device)
        self._ext.wait_stream(torch.cuda.current_stream(self.device))
        check(self._lib.svi_landmarks_optimize_dev(self.matcher._h, C.byref(self.params), _p(frame_P_left), _p(frame_P_right),
                                                   int(frame_P_left.shape[0]), _p(meas_seg), _p(meas_frame), _p(meas_uv_left), _p(meas_uv_right),
                                                   _p(xyz), n, _p(out), _p(status), _p(err), _p(its)), "svi_landmarks_optimize_dev")
        torch.cuda.current_stream(self.device).wait_stream(self._ext)
        return out, status, err, its


class BriefExtractor:
    """cv::xfeatures2d::BriefDescriptorExtractor (32 bytes) on the MI355X with a caller-supplied test-pair table
    (SURVEY.md §8f-4); an instance is the `extractor` callable the tracking cascades expect, so a frame is tracked
    without leaving the device.  roi rows are the float arguments of cv::Rect( x, y, w, h ): truncated like the
    implicit float -> int conversion does."""

    def __init__(self, pattern, matcher=None, device=0):
        self.matcher = matcher or HammingMatcher(device)
        self._lib = _capi.load_library()
        pat = np.ascontiguousarray(pattern, np.int8).reshape(1024)
        h = C.c_void_p()
        check(self._lib.svi_brief_create(self.matcher._h, pat.ctypes.data_as(C.c_void_p), C.byref(h)), "svi_brief_create")
        self._h = h
        self.device = torch.device("cuda", device)
        self._ext = torch.cuda.ExternalStream(self.matcher.stream, device=self.device)
        self._shape = [None, None]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.svi_brief_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_image(self, side, image):
        """image: H x W uint8 CUDA tensor (rows may be strided)"""
        s = 0 if side in (0, "left") else 1
        if image.dtype != torch.uint8 or not image.is_cuda or image.dim() != 2 or image.stride(1) != 1:
            raise ValueError("image must be a 2-D uint8 CUDA tensor with unit column stride")
        self._ext.wait_stream(torch.cuda.current_stream(self.device))
        check(self._lib.svi_brief_set_image_dev(self._h, s, _p(image), image.shape[1], image.shape[0], image.stride(0)), "svi_brief_set_image_dev")
        torch.cuda.current_stream(self.device).wait_stream(self._ext)
        self._shape[s] = (image.shape[0], image.shape[1])

    def integral(self, side):
        s = 0 if side in (0, "left") else 1
        hh, ww = self._shape[s]
        out = torch.empty((hh + 1, ww + 1), dtype=torch.int32, device=self.device)
        self._ext.wait_stream(torch.cuda.current_stream(self.device))
        check(self._lib.svi_brief_integral_dev(self._h, s, _p(out)), "svi_brief_integral_dev")
        torch.cuda.current_stream(self.device).wait_stream(self._ext)
        return out

    def __call__(self, side, roi, seg, kp_uv):
        s = 0 if side in (0, "left") else 1
        n = roi.shape[0]
        total_in = int(kp_uv.shape[0])
        roi_i = roi.to(torch.int32).contiguous() if roi.dtype != torch.int32 else roi.contiguous()
        seg = seg.contiguous()
        kp_uv = kp_uv.contiguous()
        seg_out = torch.empty((n + 1,), dtype=torch.int32, device=self.device)
        kp_out = torch.empty((max(total_in, 1), 2), dtype=torch.float32, device=self.device)
        desc = torch.empty((max(total_in, 1), 32), dtype=torch.uint8, device=self.device)
        total = C.c_int64(0)
        self._ext.wait_stream(torch.cuda.current_stream(self.device))
        check(self._lib.svi_brief_compute_dev(self._h, s, _p(roi_i), _p(seg), _p(kp_uv), n, total_in, _p(seg_out), _p(kp_out), _p(desc),
                                              C.byref(total)), "svi_brief_compute_dev")
        torch.cuda.current_stream(self.device).wait_stream(self._ext)
        return seg_out, kp_out[:total.value], desc[:total.value]
