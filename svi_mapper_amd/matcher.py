"""Host-side mirror of the reference's matcher interface over the C ABI.

Reference shapes mirrored here
  cv::DescriptorMatcher::match(query, train, matches)  as created at src/core/CTriangulator.cpp:12
      -> HammingMatcher.match(query, train) returning DMatch records (queryIdx, trainIdx, imgIdx, distance)
  cv::norm(a, b, cv::NORM_HAMMING)                     src/core/CFundamentalMatcher.cpp:404
      -> HammingMatcher.norm_hamming(a, b)
  CTriangulator::getPointInLEFT                        src/core/CTriangulator.cpp:326-356
      -> Triangulator.get_point_in_left(uvL, uvR)
  CTriangulator::getPointTriangulatedInRIGHT           src/core/CTriangulator.cpp:185-253
      -> Triangulator.get_point_triangulated_in_right(...) on a precomputed descriptor pool
and the batched, device-resident form of SURVEY.md Appendix A (match_dev / match_triangulate_dev).

numpy arrays are treated as host buffers (copied in and out by the library); torch CUDA tensors
are treated as device buffers and only their data_ptr() crosses the boundary.
"""
import collections
import ctypes as C

import numpy as np

from . import _capi
from ._capi import Gate, SviError, check

DMatch = collections.namedtuple("DMatch", "queryIdx trainIdx imgIdx distance")

NO_DIST = 257


class NoMatchFound(Exception):
    """CExceptionNoMatchFound (src/exceptions/CExceptionNoMatchFound.h): control flow of the reference."""


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()  # torch tensor


class HammingMatcher:
    """cv::BFMatcher(cv::NORM_HAMMING) with k = 1, on the MI355X."""

    def __init__(self, device=0, stream=None):
        self._lib = _capi.load_library()
        h = C.c_void_p()
        check(self._lib.svi_matcher_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)),
              "svi_matcher_create")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.svi_matcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return self._lib.svi_matcher_stream(self._h)

    def set_gate_path(self, path):
        """0: gated calls on a small pool bucket it by image row (default); 1: the gate as a per-pair predicate of the scan"""
        check(self._lib.svi_matcher_set_gate_path(self._h, int(path)), "svi_matcher_set_gate_path")

    def shader_clock_mhz(self):
        v = C.c_double(0)
        check(self._lib.svi_debug_shader_clock_mhz(self._h, C.byref(v)), "svi_debug_shader_clock_mhz")
        return v.value

    def synchronize(self):
        check(self._lib.svi_matcher_sync(self._h), "svi_matcher_sync")

    # -- host buffers --------------------------------------------------------------------------
    @staticmethod
    def _desc(a):
        a = np.ascontiguousarray(a, np.uint8)
        if a.ndim == 1:
            a = a.reshape(-1, 32)
        if a.ndim != 2 or a.shape[1] != 32:
            raise ValueError("descriptors must be N x 32 uint8 (BRIEF-256)")
        if a.ctypes.data & 15:  # the ABI wants 16-byte aligned rows
            b = np.empty(a.size + 16, np.uint8)
            off = (-b.ctypes.data) & 15
            b = b[off:off + a.size].reshape(a.shape)
            b[...] = a
            a = b
        return a

    def match_arrays(self, query, train, gate=None, max_dist_exclusive=NO_DIST):
        """(idx int32[nq], dist int32[nq]); idx -1 / dist 257 where the reference would throw."""
        q = self._desc(query)
        t = self._desc(train)
        idx = np.empty(len(q), np.int32)
        dist = np.empty(len(q), np.int32)
        g = None
        keep = []
        if gate is not None:
            quv = np.ascontiguousarray(gate["q_uv"], np.float32).reshape(-1, 2)
            tuv = np.ascontiguousarray(gate["t_uv"], np.float32).reshape(-1, 2)
            umin = np.ascontiguousarray(gate["q_umin"], np.float32).reshape(-1)
            umax = np.ascontiguousarray(gate["q_umax"], np.float32).reshape(-1)
            if len(quv) != len(q) or len(umin) != len(q) or len(umax) != len(q) or len(tuv) != len(t):
                raise ValueError("gate arrays do not match the descriptor counts")
            keep = [quv, tuv, umin, umax]
            g = Gate(_ptr(quv), _ptr(tuv), _ptr(umin), _ptr(umax), float(gate.get("v_tol", 0.0)))
        check(self._lib.svi_match_hamming256(self._h, _ptr(q), len(q), _ptr(t) if len(t) else None, len(t),
                                             C.byref(g) if g is not None else None, int(max_dist_exclusive),
                                             _ptr(idx), _ptr(dist)), "svi_match_hamming256")
        del keep
        return idx, dist

    def match(self, query, train, gate=None):
        """cv::DescriptorMatcher::match: one DMatch per query row that has a candidate (an empty
        pool gives an empty list, which the reference turns into CExceptionNoMatchFound,
        CTriangulator.cpp:95-98)."""
        idx, dist = self.match_arrays(query, train, gate, NO_DIST)
        return [DMatch(i, int(j), 0, float(d)) for i, (j, d) in enumerate(zip(idx, dist)) if j >= 0]

    def norm_hamming(self, a, b):
        a = self._desc(a)
        b = self._desc(b)
        if a.shape != b.shape:
            raise ValueError("shape mismatch")
        d = np.empty(len(a), np.int32)
        check(self._lib.svi_hamming256_pairs(self._h, _ptr(a), _ptr(b), len(a), _ptr(d)), "svi_hamming256_pairs")
        return d

    def triangulate(self, f, cx, cy, duR_flipped, uvL, uvR, min_disparity=0.01):
        uvL = np.ascontiguousarray(uvL, np.float32).reshape(-1, 2)
        uvR = np.ascontiguousarray(uvR, np.float32).reshape(-1, 2)
        xyz = np.zeros((len(uvL), 3))
        ok = np.zeros(len(uvL), np.uint8)
        check(self._lib.svi_triangulate_rectified(self._h, f, cx, cy, duR_flipped, min_disparity, _ptr(uvL), _ptr(uvR),
                                                  len(uvL), _ptr(xyz), _ptr(ok)), "svi_triangulate_rectified")
        return xyz, ok

    # -- device buffers (torch CUDA tensors), asynchronous on self.stream ------------------------
    def match_dev(self, q, t, nq, nt, batch, out_idx, out_dist, gate=None, max_dist_exclusive=NO_DIST):
        g = None
        if gate is not None:
            g = Gate(_ptr(gate["q_uv"]), _ptr(gate["t_uv"]), _ptr(gate["q_umin"]), _ptr(gate["q_umax"]),
                     float(gate.get("v_tol", 0.0)))
        check(self._lib.svi_match_hamming256_dev(self._h, _ptr(q), int(nq), _ptr(t), int(nt), int(batch),
                                                 C.byref(g) if g is not None else None, int(max_dist_exclusive),
                                                 _ptr(out_idx), _ptr(out_dist)), "svi_match_hamming256_dev")

    def match_triangulate_dev(self, q, t, nq, nt, batch, gate, max_dist_exclusive, f, cx, cy, duR_flipped,
                              out_idx, out_dist, out_xyz, out_ok, min_disparity=0.01):
        g = Gate(_ptr(gate["q_uv"]), _ptr(gate["t_uv"]), _ptr(gate["q_umin"]), _ptr(gate["q_umax"]),
                 float(gate.get("v_tol", 0.0)))
        check(self._lib.svi_match_triangulate_dev(self._h, _ptr(q), int(nq), _ptr(t), int(nt), int(batch), C.byref(g),
                                                  int(max_dist_exclusive), f, cx, cy, duR_flipped, min_disparity,
                                                  _ptr(out_idx), _ptr(out_dist), _ptr(out_xyz), _ptr(out_ok)),
              "svi_match_triangulate_dev")

    def match_clouds_dev(self, q, nq, pools, pool_seg, n_clouds, max_pool, out_idx, out_dist, max_dist_exclusive=25):
        """Loop-closure candidates (USING_BF, CTrackerSVI.cpp:1221-1259): query pool vs every past key frame's pool;
        out_idx / out_dist are n_clouds x nq, kept iff MAXIMUM_DISTANCE_HAMMING (25) > distance."""
        check(self._lib.svi_match_clouds_dev(self._h, _ptr(q), int(nq), _ptr(pools), _ptr(pool_seg), int(n_clouds), int(max_pool),
                                             int(max_dist_exclusive), _ptr(out_idx), _ptr(out_dist)), "svi_match_clouds_dev")

    def pairs_dev(self, a, b, n, out):
        check(self._lib.svi_hamming256_pairs_dev(self._h, _ptr(a), _ptr(b), int(n), _ptr(out)),
              "svi_hamming256_pairs_dev")


class Triangulator:
    """CTriangulator (src/core/CTriangulator.{h,cpp}) on precomputed descriptor pools.

    The reference builds the pool by extracting a BRIEF descriptor at every integer pixel of the
    epipolar segment (CTriangulator.cpp:61-83); descriptor extraction is outside this path
    (SURVEY.md §2.1), so the pool descriptors are an input here."""

    min_search_range_px = 60.0   # CTriangulator.h:20
    min_disparity_px = 0.01      # CTriangulator.h:21
    cutoff = 100                 # CTriangulator.cpp:13

    def __init__(self, f, cx, cy, duR, width, matcher=None, device=0):
        self.f, self.cx, self.cy = float(f), float(cx), float(cy)
        self.duR_flipped = -float(duR)
        self.depth_min = self.duR_flipped / float(width)           # CTriangulator.cpp:20
        self.depth_max = self.duR_flipped / self.min_disparity_px  # CTriangulator.cpp:21
        self.matcher = matcher or HammingMatcher(device)

    def get_point_in_left(self, uvL, uvR):
        xyz, ok = self.matcher.triangulate(self.f, self.cx, self.cy, self.duR_flipped, np.asarray(uvL).reshape(1, 2),
                                           np.asarray(uvR).reshape(1, 2), self.min_disparity_px)
        if not ok[0]:
            raise NoMatchFound("<CTriangulator>(getPointInLEFT) zero disparity")
        return xyz[0]

    def get_point_triangulated_in_right(self, pool_desc, u_top_left, v_top_left, keypoint_size, uvL, ref_desc):
        """pool_desc[k] is the descriptor at pixel (u_top_left + 4*size + k, v_top_left + 4*size)
        (CTriangulator.cpp:201-211). Returns (xyz, uvR, matched descriptor)."""
        border = 4.0 * keypoint_size
        if uvL[0] <= u_top_left + border:
            raise NoMatchFound("<CTriangulator>(getPointTriangulatedInRIGHT) insufficient search range")
        pool_desc = np.asarray(pool_desc, np.uint8).reshape(-1, 32)
        if len(pool_desc) == 0:
            raise NoMatchFound("<CTriangulator>(getPointTriangulatedInRIGHT) could not compute descriptors")
        idx, dist = self.matcher.match_arrays(np.asarray(ref_desc, np.uint8).reshape(1, 32), pool_desc, None, self.cutoff)
        if idx[0] < 0:
            raise NoMatchFound("<CTriangulator>(getPointTriangulatedInRIGHT) matching distance")
        uvR = np.array([np.float32(border + idx[0]) + np.float32(u_top_left),
                        np.float32(border) + np.float32(v_top_left)], np.float32)
        return self.get_point_in_left(np.asarray(uvL, np.float32), uvR), uvR, pool_desc[idx[0]]
