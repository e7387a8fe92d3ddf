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
The cascades themselves (stage 1 -> 2 -> 3, masks, hand-overs) run behind the C ABI (csrc/track_cascade.hip:
svi_track_stage1 / _stage2 / _epipolar / _manual / _pose_stereo_posit / _add_new_landmarks); this file only marshals
tensors and adapts Python callables to the library's extractor / detector callbacks.
GFTT detection is OpenCV's and stays with the caller; BRIEF extraction is either the built-in BriefExtractor (no callback:
the frame is tracked without leaving the device) or a caller's `extractor(side, roi, seg, kp_uv)`, called with device tensors
(roi n x 4 f32, seg n+1 i32, kp_uv total x 2 f32 in ROI coordinates) and returning (seg', kp_uv', desc') - the key points it
kept (OpenCV drops those too close to the ROI border) and their 32-byte descriptors.  `detector(side, rect)` gets the n
search rectangles (corners) and returns (seg, kp_uv in rectangle coordinates).
"""
import ctypes as C

import numpy as np
import torch

from . import _capi, _dlpack
from ._capi import LandmarkParams, PositParams, PositResult
from ._capi import (MATCH_OK, MATCH_SKIPPED, TRACK_RECORD_FIELDS, TRACK_RECORD_SIZE, TrackCamera, TrackLandmarks, TrackResult,
                    TrackStereoParams, check)
from .matcher import HammingMatcher

RECORD_DTYPE = np.dtype(TRACK_RECORD_FIELDS)


def _p(t):
    return None if t is None else t.data_ptr()


def _alias(ptr, shape, dtype, device):
    """torch view of library-owned device memory, no copy (a DLPack capsule that states the device: _dlpack.py)"""
    return _dlpack.alias(ptr, shape, dtype, device)


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

        self.stage = torch.zeros((n,), dtype=torch.int8, device=device)

    def ok(self):
        return self.status == MATCH_OK

    def c_struct(self):
        return TrackResult(_p(self.status), _p(self.stage), _p(self.uv_left), _p(self.uv_right), _p(self.xyz_left), _p(self.desc_left),
                           _p(self.desc_right))


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
        h = C.c_void_p()
        check(self._lib.svi_tracker_create(self._h, C.byref(self._cam), C.byref(h)), "svi_tracker_create")
        self._trk = h
        self._bound = (None, None)   # the (extractor, detector) the tracker handle is bound to, with their ctypes thunks kept alive
        self._thunks = ()
        self._frame = None           # tensors of the planned frame (kept alive while the tracker holds their pointers)
        self.detector_capacity = 1 << 20

    def close(self):
        if getattr(self, "_trk", None):
            self._lib.svi_tracker_destroy(self._trk)
            self._trk = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- callbacks: Python callables behind svi_extract_fn / svi_detect_fn -------------------------------------------
    def _bind(self, extractor, detector):
        if self._bound == (extractor, detector):
            return
        dev = self.device
        thunks = []
        if isinstance(extractor, BriefExtractor):
            if extractor.matcher is not self.matcher:
                raise ValueError("the BriefExtractor must be built on this matcher (same stream)")
            check(self._lib.svi_tracker_set_brief(self._trk, extractor._h), "svi_tracker_set_brief")
        elif extractor is not None:
            def extract_cb(user, side, roi, seg, kp_uv, n, total_in, seg_out, kp_out, desc_out, total_out, stream):
                try:
                    with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                        seg_t = _alias(seg, (n + 1,), torch.int32, dev)
                        total = int(seg_t[-1].item()) if n else 0
                        s2, k2, d2 = extractor("left" if side == 0 else "right", _alias(roi, (n, 4), torch.float32, dev), seg_t,
                                               _alias(kp_uv, (total, 2), torch.float32, dev))
                        kept = int(k2.shape[0])
                        if kept > total_in:
                            return 2
                        _alias(seg_out, (n + 1,), torch.int32, dev).copy_(s2.to(torch.int32))
                        if kept:
                            _alias(kp_out, (kept, 2), torch.float32, dev).copy_(k2)
                            _alias(desc_out, (kept, 32), torch.uint8, dev).copy_(d2)
                        torch.cuda.current_stream(dev).synchronize()
                    total_out[0] = kept
                    return 0
                except Exception:  # noqa: BLE001  (an exception must not unwind through the C frames)
                    import traceback
                    traceback.print_exc()
                    return 1
            thunks.append(_capi.EXTRACT_FN(extract_cb))
            check(self._lib.svi_tracker_set_extractor(self._trk, thunks[-1], None), "svi_tracker_set_extractor")
        if detector is not None:
            def detect_cb(user, side, rect, active, n, cap, seg_out, kp_out, total_out, stream):
                try:
                    with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                        seg_d, kp_d = detector("left" if side == 0 else "right", _alias(rect, (n, 4), torch.float32, dev))
                        act = _alias(active, (n,), torch.uint8, dev).bool()
                        cnt = (seg_d[1:] - seg_d[:-1]).long()
                        if bool((cnt[~act] != 0).any()):    # rows that do not run must stay empty
                            keep = torch.repeat_interleave(act, cnt)
                            kp_d = kp_d[keep]
                            cnt = torch.where(act, cnt, torch.zeros_like(cnt))
                            seg_d = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(cnt, 0)]).to(torch.int32)
                        total = int(kp_d.shape[0])
                        if total > cap:
                            return 2
                        _alias(seg_out, (n + 1,), torch.int32, dev).copy_(seg_d.to(torch.int32))
                        if total:
                            _alias(kp_out, (total, 2), torch.float32, dev).copy_(kp_d)
                        torch.cuda.current_stream(dev).synchronize()
                    total_out[0] = total
                    return 0
                except Exception:  # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    return 1
            thunks.append(_capi.DETECT_FN(detect_cb))
            check(self._lib.svi_tracker_set_detector(self._trk, thunks[-1], None, int(self.detector_capacity)), "svi_tracker_set_detector")
        else:
            check(self._lib.svi_tracker_set_detector(self._trk, _capi.DETECT_FN(0), None, 0), "svi_tracker_set_detector")
        self._bound = (extractor, detector)
        self._thunks = tuple(thunks)

    def _descriptors(self, last_left, last_right, ref_left):
        keep = []
        ptrs = []
        for t in (last_left, last_right, ref_left):
            if t is None:
                ptrs.append(None)
                continue
            t = t.contiguous()
            if t.dtype != torch.uint8 or not t.is_cuda:
                raise ValueError("descriptors must be uint8 CUDA tensors")
            keep.append(t)
            ptrs.append(self._q(t))
        self._desc_keep = keep
        check(self._lib.svi_tracker_set_descriptors(self._trk, *ptrs), "svi_tracker_set_descriptors")

    def _run(self, fn, name, plan, active):
        if self._frame is None or plan is not self._frame:
            raise ValueError("the cascades run on the most recently planned frame")
        res = StageResult(plan.n, self.device)
        if active is not None:
            active = active.to(torch.uint8).contiguous()
        rs = res.c_struct()
        self._enter()
        check(fn(self._trk, _p(active), C.byref(rs)), name)
        self._leave()
        return res

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
        lm = TrackLandmarks(n, _p(xyz_world), _p(kp_size), _p(last_disparity), _p(uv_reference), _p(dp_index), None, None, None)
        self._enter()
        check(self._lib.svi_tracker_plan(self._trk, T.ctypes.data_as(_capi.f64p), dp.ctypes.data_as(_capi.f64p) if len(dp) else None, len(dp),
                                         float(motion_scaling), C.byref(lm)), "svi_tracker_plan")
        rec_p, seg_p, total = C.c_void_p(), C.c_void_p(), C.c_int64(0)
        check(self._lib.svi_tracker_records(self._trk, C.byref(rec_p), C.byref(seg_p), C.byref(total)), "svi_tracker_records")
        with torch.cuda.stream(self._ext):
            # the tracker owns the plan; the harness works on copies (they outlive the next frame's plan)
            records = _alias(rec_p.value, (n, TRACK_RECORD_SIZE), torch.uint8, self.device).clone()
            seg = _alias(seg_p.value, (n + 1,), torch.int32, self.device).clone()
        self._leave()
        plan = TrackPlan(records, seg, total.value, kp_size)
        self._frame = plan
        self._frame_keep = (xyz_world, kp_size, last_disparity, uv_reference, dp_index)
        return plan

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

    # ---- the cascades: one call each across the boundary (csrc/track_cascade.hip) -------------------------------------------
    def track_stage1(self, plan, extractor, last_desc_left, last_desc_right, active=None):
        """Stage 1 (:391-486): descriptor at the projected pixel, LEFT first then RIGHT, followed by the stereo search in the
        other image.  Returns a StageResult over all plan.n landmarks; rows that fail keep their last failure code."""
        self._bind(extractor, self._bound[1])
        self._descriptors(last_desc_left, last_desc_right, None)
        return self._run(self._lib.svi_track_stage1, "svi_track_stage1", plan, active)

    def track_stage2(self, plan, detector, extractor, last_desc_left, last_desc_right, active=None):
        """Stage 2 (:489-709): `detector(side, rect)` (rect n x 4 f32: the search rectangle corners) returns the ragged key
        points it found (seg, kp_uv in search-rectangle coordinates); they are shifted by (4s,4s) and described inside the grown
        rectangle (:533-534), matched against the last descriptor (cut-off 50) and verified in the other image."""
        self._bind(extractor, detector)
        self._descriptors(last_desc_left, last_desc_right, None)
        return self._run(self._lib.svi_track_stage2, "svi_track_stage2", plan, active)

    def track_epipolar(self, plan, extractor, last_desc_left, ref_desc_left, active=None, detector=None, last_desc_right=None):
        """trackEpipolar (:794-1315): sampling along the clipped epipolar line (depth 0, then 2 for the landmarks that found
        nothing), _getMatch with the relative (50) and original (100) cut-offs, then _addMeasurementToLandmarkLEFT: the stereo
        search in RIGHT without a descriptor check.  Landmarks whose detection point has not moved have no epipolar line: with
        a detector they are searched by stage 2 (:1026-1290).  Everything else without SVI_TRK_EPI_OK reports SKIPPED."""
        self._bind(extractor, detector)
        self._descriptors(last_desc_left, last_desc_right if last_desc_right is not None else last_desc_left, ref_desc_left)
        return self._run(self._lib.svi_track_epipolar, "svi_track_epipolar", plan, active)

    def track_manual(self, plan, detector, extractor, last_desc_left, last_desc_right, ref_desc_left, active=None):
        """trackManual (:1366-2019): stage 1 -> stage 2 -> epipolar, each only for what the previous one lost.  One StageResult;
        `stage` (int8: 1, 2, 3, 0 = none) tells which stage produced each measurement.  A landmark outside the field of view of
        either camera is not tracked at all (:1415, :2008-2012)."""
        self._bind(extractor, detector)
        self._descriptors(last_desc_left, last_desc_right, ref_desc_left)
        return self._run(self._lib.svi_track_manual, "svi_track_manual", plan, active)

    def pose_stereo_posit(self, plan, detector, extractor, last_desc_left, last_desc_right, solver, T_last, t_imu, T_estimate, active=None):
        """getPoseStereoPosit (:340-760): stage 1 -> stage 2 over the active (bIsOptimal) landmarks and the frame pose from what
        they found, in one call.  -> (StageResult, PositResult); xyz_world of the plan call feeds the solver."""
        self._bind(extractor, detector)
        self._descriptors(last_desc_left, last_desc_right, None)
        if self._frame is None or plan is not self._frame:
            raise ValueError("the cascades run on the most recently planned frame")
        res = StageResult(plan.n, self.device)
        rs = res.c_struct()
        pose = PositResult()
        Tl = np.ascontiguousarray(T_last, np.float64).reshape(12)
        Te = np.ascontiguousarray(T_estimate, np.float64).reshape(12)
        ti = np.ascontiguousarray(t_imu, np.float64).reshape(3)
        if active is not None:
            active = active.to(torch.uint8).contiguous()
        self._enter()
        check(self._lib.svi_track_pose_stereo_posit(self._trk, _p(active), C.byref(solver.params), Tl.ctypes.data_as(_capi.f64p),
                                                    ti.ctypes.data_as(_capi.f64p), Te.ctypes.data_as(_capi.f64p), C.byref(rs), C.byref(pose)),
              "svi_track_pose_stereo_posit")
        self._leave()
        return res, pose

    def add_new_landmarks(self, extractor, uv_left, kp_size, desc_left):
        """addNewLandmarks (:83-193): stereo partner + triangulation of freshly detected key points: getPointTriangulatedInRIGHTFull
        with the search window of CTriangulator::fMinimumSearchRangePixels = 60 (no depth gate, no second descriptor check).
        uv_left n x 2 f32, kp_size n f32, desc_left n x 32 u8 (the detector / extractor output on the LEFT image)."""
        self._bind(extractor, self._bound[1])
        n = uv_left.shape[0]
        res = StageResult(n, self.device)
        if n == 0:
            return res
        uv_left, kp_size, desc_left = uv_left.contiguous(), kp_size.contiguous(), desc_left.contiguous()
        rs = res.c_struct()
        self._enter()
        check(self._lib.svi_track_add_new_landmarks(self._trk, _p(uv_left), _p(kp_size), _p(desc_left), n, C.byref(rs)), "svi_track_add_new_landmarks")
        self._leave()
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
