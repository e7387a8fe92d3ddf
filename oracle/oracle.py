"""ctypes loader for the CPU oracle (oracle_match.c, oracle_ba.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under svi_mapper_amd/ may import this module.
PARITY UNPINNED (see the C file headers and DESIGN.md "Oracle").

The tracking-schedule functions (track_*) restate src/core/CFundamentalMatcher.cpp one landmark at a time
(oracle_track.c); OracleFundamentalMatcher replays the reference's try/catch cascade around them.

The Python classes deliberately expose the same method names as
svi_mapper_amd.BundleAdjuster / HammingMatcher so the parity tests drive both alike.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f64p = C.POINTER(C.c_double)
_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)


def build(native=False):
    """Compile the oracle with gcc. native=True builds liboracle_native.so with -march=native
    (used by the timed cpu_baseline leg on the box it runs on)."""
    if native:
        subprocess.check_call(["make", "-s", "-C", _HERE, "MARCH=native", "OUT=liboracle_native.so"])
        return os.path.join(_HERE, "liboracle_native.so")
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return os.path.join(_HERE, "liboracle.so")


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def load(path=None):
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    if path is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
    lib = C.CDLL(path)
    lib.orc_ba_create.restype = C.c_void_p
    lib.orc_ba_create.argtypes = [C.c_double] * 5
    lib.orc_ba_destroy.argtypes = [C.c_void_p]
    lib.orc_ba_set_lm.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double]
    lib.orc_ba_set_accel_numeric.argtypes = [C.c_void_p, C.c_int]
    lib.orc_ba_set_imu_offset.argtypes = [C.c_void_p, _f64p]
    lib.orc_ba_aux_jacobians.argtypes = [C.c_void_p, _f64p, _f64p, _f64p, _f64p, _f64p]
    lib.orc_ba_add_pose.argtypes = [C.c_void_p, C.c_int64, _f64p, C.c_int]
    lib.orc_ba_add_landmark.argtypes = [C.c_void_p, C.c_int64, _f64p, C.c_int]
    lib.orc_ba_add_edge_proj.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, _f64p, _f64p, C.c_int]
    lib.orc_ba_add_edge_se3.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _f64p, _f64p, C.c_int]
    lib.orc_ba_add_edge_accel.argtypes = [C.c_void_p, C.c_int64, _f64p, _f64p, _f64p]
    lib.orc_ba_add_edge_lm_lm.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _f64p, _f64p, C.c_int]
    lib.orc_ba_add_keyframe.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _f64p, _f64p, _f64p]
    lib.orc_ba_add_measurements.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _i64p, _f32p, _f32p, _f64p, _i64p]
    lib.orc_ba_initialize.argtypes = [C.c_void_p]
    lib.orc_ba_optimize.argtypes = [C.c_void_p, C.c_int]
    lib.orc_ba_optimize_until.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int,
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.orc_ba_chi2.argtypes = [C.c_void_p, _f64p, _f64p]
    lib.orc_ba_last_plain_chi2.restype = C.c_double
    lib.orc_ba_last_plain_chi2.argtypes = [C.c_void_p]
    lib.orc_ba_lambda.restype = C.c_double
    lib.orc_ba_lambda.argtypes = [C.c_void_p]
    for f in ("num_poses", "num_landmarks", "num_edges", "num_aux", "system_size"):
        getattr(lib, "orc_ba_" + f).restype = C.c_int64
        getattr(lib, "orc_ba_" + f).argtypes = [C.c_void_p]
    for f in ("iterations", "trials"):
        getattr(lib, "orc_ba_" + f).restype = C.c_uint64
        getattr(lib, "orc_ba_" + f).argtypes = [C.c_void_p]
    lib.orc_ba_get_pose.argtypes = [C.c_void_p, C.c_int64, _f64p]
    lib.orc_ba_get_landmark.argtypes = [C.c_void_p, C.c_int64, _f64p]
    lib.orc_ba_get_poses.argtypes = [C.c_void_p, _i64p, _f64p]
    lib.orc_ba_get_landmarks.argtypes = [C.c_void_p, _i64p, _f64p]
    lib.orc_ba_get_edges.argtypes = [C.c_void_p, _i32p, _i64p, _i64p, _f64p, _f64p]
    lib.orc_ba_get_aux.argtypes = [C.c_void_p, _i32p, _i64p, _i64p, _f64p, _f64p]
    lib.orc_ba_edge_jacobians.argtypes = [C.c_void_p, _f64p, _f64p, _f64p]
    lib.orc_ba_dense_system.restype = C.c_int64
    lib.orc_ba_dense_system.argtypes = [C.c_void_p, _f64p, _f64p, C.c_int64, _i32p, _i32p]
    lib.orc_ba_trace.argtypes = [C.c_void_p, _f64p, C.c_int]
    lib.orc_ba_trace_clear.argtypes = [C.c_void_p]
    lib.orc_ba_prune_diverged.restype = C.c_int64
    lib.orc_ba_prune_diverged.argtypes = [C.c_void_p]
    lib.orc_se3_edge.argtypes = [_f64p] * 6
    lib.orc_se3_oplus.argtypes = [_f64p] * 3
    lib.orc_match_hamming256.argtypes = [_u8p, C.c_int, _u8p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p,
                                         C.c_float, C.c_int, _i32p, _i32p]
    lib.orc_hamming256_pairs.argtypes = [_u8p, _u8p, C.c_int, _i32p]
    lib.orc_triangulate_rectified.argtypes = [C.c_double] * 5 + [_f32p, _f32p, C.c_int, _f64p, _u8p]
    vp = C.c_void_p
    lib.orc_track_record_size.restype = C.c_int
    lib.orc_track_plan.argtypes = [vp, _f64p, _f64p, C.c_int, C.c_double, _f64p, _f32p, _f32p, _f64p, _i32p, C.c_int, vp, _i32p]
    lib.orc_track_epipolar_samples.argtypes = [vp, vp, _f32p, _i32p, C.c_int, _i32p, C.c_int, _f32p, _f32p]
    lib.orc_track_stereo_range.argtypes = [C.c_double, C.c_int, _f32p, _f32p, _f32p, _f32p, _u8p, C.c_int, _i32p, _i32p, _f32p]
    lib.orc_track_stereo_candidates.argtypes = [C.c_int, _f32p, C.c_int, _i32p, _f32p]
    lib.orc_match_ragged.argtypes = [_u8p, _u8p, _u8p, C.c_int, _i32p, _u8p, C.c_int, C.c_int, _i32p, _i32p, _i32p]
    lib.orc_track_stereo_verify.argtypes = [vp, _u8p, _u8p, _u8p, _f32p, _f32p, C.c_int, _i32p, _u8p, _f32p, _i32p, _i32p, _i32p,
                                            _f32p, _f64p]
    lib.orc_track_handover.argtypes = [C.c_int, vp, _f32p, _i32p, C.c_int, _i32p, _f32p, _i32p, _f32p, _f32p, _f32p, _u8p]
    lib.orc_brief_integral.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _i32p]
    lib.orc_brief_compute.restype = C.c_int
    lib.orc_brief_compute.argtypes = [_i32p, C.c_int, C.c_int, C.c_void_p, _i32p, _i32p, _f32p, C.c_int, _i32p, _f32p, _u8p]
    lib.orc_landmarks_optimize.argtypes = [vp, _f64p, _f64p, C.c_int, _i32p, _i32p, _f32p, _f32p, _f64p, C.c_int, _f64p, _i32p, _f64p, _i32p]
    lib.orc_stereo_posit.argtypes = [vp, _f64p, _f64p, _f64p, _f64p, _f32p, _f32p, _u8p, C.c_int, vp]
    if path.endswith("liboracle.so"):
        _LIB = lib
    return lib


# ------------------------------------------------------------------------------------------------
# matcher
# ------------------------------------------------------------------------------------------------
def match_hamming256(q, t, gate=None, max_dist_exclusive=257, lib=None):
    """gate = dict(q_uv, t_uv, q_umin, q_umax, v_tol) or None. Returns (idx int32, dist int32)."""
    lib = lib or load()
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.empty(len(q), np.int32)
    dist = np.empty(len(q), np.int32)
    if gate is not None:
        quv = np.ascontiguousarray(gate["q_uv"], np.float32)
        tuv = np.ascontiguousarray(gate["t_uv"], np.float32)
        umin = np.ascontiguousarray(gate["q_umin"], np.float32)
        umax = np.ascontiguousarray(gate["q_umax"], np.float32)
        lib.orc_match_hamming256(_p(q, _u8p), len(q), _p(t, _u8p), len(t), 1, _p(quv, _f32p), _p(tuv, _f32p),
                                 _p(umin, _f32p), _p(umax, _f32p), float(gate.get("v_tol", 0.0)),
                                 int(max_dist_exclusive), _p(idx, _i32p), _p(dist, _i32p))
    else:
        lib.orc_match_hamming256(_p(q, _u8p), len(q), _p(t, _u8p), len(t), 0, None, None, None, None, 0.0,
                                 int(max_dist_exclusive), _p(idx, _i32p), _p(dist, _i32p))
    return idx, dist


def hamming256_pairs(a, b, lib=None):
    lib = lib or load()
    a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32)
    b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
    d = np.empty(len(a), np.int32)
    lib.orc_hamming256_pairs(_p(a, _u8p), _p(b, _u8p), len(a), _p(d, _i32p))
    return d


def triangulate_rectified(f, cx, cy, duR_flipped, uvL, uvR, min_disparity=0.01, lib=None):
    lib = lib or load()
    uvL = np.ascontiguousarray(uvL, np.float32).reshape(-1, 2)
    uvR = np.ascontiguousarray(uvR, np.float32).reshape(-1, 2)
    xyz = np.zeros((len(uvL), 3), np.float64)
    ok = np.zeros(len(uvL), np.uint8)
    lib.orc_triangulate_rectified(f, cx, cy, duR_flipped, min_disparity, _p(uvL, _f32p), _p(uvR, _f32p),
                                  len(uvL), _p(xyz, _f64p), _p(ok, _u8p))
    return xyz, ok


def se3_edge(Xi, Xj, Z, jac=True, lib=None):
    lib = lib or load()
    Xi, Xj, Z = (np.ascontiguousarray(a, np.float64).reshape(12) for a in (Xi, Xj, Z))
    e = np.zeros(6)
    if jac:
        Ji = np.zeros((6, 6))
        Jj = np.zeros((6, 6))
        lib.orc_se3_edge(_p(Xi, _f64p), _p(Xj, _f64p), _p(Z, _f64p), _p(e, _f64p), _p(Ji, _f64p), _p(Jj, _f64p))
        return e, Ji, Jj
    lib.orc_se3_edge(_p(Xi, _f64p), _p(Xj, _f64p), _p(Z, _f64p), _p(e, _f64p), None, None)
    return e


def se3_oplus(T, d, lib=None):
    lib = lib or load()
    T = np.ascontiguousarray(T, np.float64).reshape(12)
    d = np.ascontiguousarray(d, np.float64).reshape(6)
    out = np.zeros(12)
    lib.orc_se3_oplus(_p(T, _f64p), _p(d, _f64p), _p(out, _f64p))
    return out


# ------------------------------------------------------------------------------------------------
# bundle adjustment
# ------------------------------------------------------------------------------------------------
class OracleBA:
    """Same surface as svi_mapper_amd.BundleAdjuster, computed by the CPU restatement."""

    def __init__(self, fx, fy, cx, cy, baseline_m, lib=None, **lm):
        self.lib = lib or load()
        self.h = C.c_void_p(self.lib.orc_ba_create(fx, fy, cx, cy, baseline_m))
        if lm:
            self.lib.orc_ba_set_lm(self.h, lm.get("lm_tau", 1e-5), lm.get("lm_good_step_lower", 1 / 3),
                                   lm.get("lm_good_step_upper", 2 / 3), lm.get("lm_max_trials", 10),
                                   lm.get("cauchy_delta", 1.0))

    def close(self):
        if self.h:
            self.lib.orc_ba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _d(a, n=None):
        a = np.ascontiguousarray(a, np.float64)
        if n is not None:
            a = a.reshape(n)
        return a

    def _chk(self, rc, what):
        if rc != 0:
            raise ValueError("oracle: %s failed (%d)" % (what, rc))

    def set_accel_numeric(self, on):
        self.lib.orc_ba_set_accel_numeric(self.h, int(on))

    def set_imu_offset(self, off):
        """IMU->LEFT offset parameter (Cg2oOptimizer.cpp:213) of the gravity edges that add_keyframe creates"""
        off = self._d(off, 12)
        self.lib.orc_ba_set_imu_offset(self.h, _p(off, _f64p))

    def aux_jacobians(self):
        """pose-only edges in insertion order among their kind: (se3_err n x 6, Ji n x 6 x 6, Jj, acc_err m x 3, acc_J m x 3 x 6)"""
        t, _, _, _, _ = self.get_aux()
        ns, na = int((t == 0).sum()), int((t == 1).sum())
        se, si, sj = np.zeros((ns, 6)), np.zeros((ns, 6, 6)), np.zeros((ns, 6, 6))
        ae, aj = np.zeros((na, 3)), np.zeros((na, 3, 6))
        self.lib.orc_ba_aux_jacobians(self.h, _p(se, _f64p), _p(si, _f64p), _p(sj, _f64p), _p(ae, _f64p), _p(aj, _f64p))
        return se, si, sj, ae, aj

    def add_pose(self, id, T, fixed=False):
        T = self._d(T, 12)
        self._chk(self.lib.orc_ba_add_pose(self.h, int(id), _p(T, _f64p), int(fixed)), "add_pose")

    def add_landmark(self, id, p, fixed=False):
        p = self._d(p, 3)
        self._chk(self.lib.orc_ba_add_landmark(self.h, int(id), _p(p, _f64p), int(fixed)), "add_landmark")

    def add_landmarks(self, ids, p, fixed=False):
        p = self._d(p).reshape(-1, 3)
        for i, pid in enumerate(ids):
            self._chk(self.lib.orc_ba_add_landmark(self.h, int(pid), _p(p[i], _f64p), int(fixed)), "add_landmark")

    def add_edges_bulk(self, type, pose_id, lm_id, z, info_upper, robust):
        z = self._d(z).reshape(-1, 3)
        info = self._d(info_upper).reshape(-1, 6)
        robust = np.broadcast_to(np.asarray(robust, np.int32), (len(z),))
        for i in range(len(z)):
            self._chk(self.lib.orc_ba_add_edge_proj(self.h, int(type[i]), int(pose_id[i]), int(lm_id[i]),
                                                    _p(z[i], _f64p), _p(info[i], _f64p), int(robust[i])), "add_edge")

    def add_edge_se3(self, i, j, Z, info_upper, robust=False):
        Z = self._d(Z, 12)
        info = self._d(info_upper, 21)
        self._chk(self.lib.orc_ba_add_edge_se3(self.h, int(i), int(j), _p(Z, _f64p), _p(info, _f64p), int(robust)),
                  "add_edge_se3")

    def add_edge_accel(self, pose_id, a, off=None, info_upper=(1, 0, 0, 1, 0, 1)):
        a = self._d(a, 3)
        info = self._d(info_upper, 6)
        off = self._d(off, 12) if off is not None else None
        self._chk(self.lib.orc_ba_add_edge_accel(self.h, int(pose_id), _p(a, _f64p), _p(off, _f64p), _p(info, _f64p)),
                  "add_edge_accel")

    def add_edge_lm_lm(self, i, j, z, info_upper, robust=True):
        z = self._d(z, 3)
        info = self._d(info_upper, 6)
        self._chk(self.lib.orc_ba_add_edge_lm_lm(self.h, int(i), int(j), _p(z, _f64p), _p(info, _f64p), int(robust)),
                  "add_edge_lm_lm")

    def add_keyframe(self, id, from_id, T, shift=None, accel=None):
        T = self._d(T, 12)
        shift = self._d(shift, 3) if shift is not None else None
        accel = self._d(accel, 3) if accel is not None else None
        self._chk(self.lib.orc_ba_add_keyframe(self.h, int(id), int(from_id), _p(T, _f64p), _p(shift, _f64p),
                                               _p(accel, _f64p)), "add_keyframe")

    def add_measurements(self, pose_id, lm_id, uv_left, uv_right, xyz_left):
        lm_id = np.ascontiguousarray(lm_id, np.int64)
        uvl = np.ascontiguousarray(uv_left, np.float32).reshape(-1, 2)
        uvr = np.ascontiguousarray(uv_right, np.float32).reshape(-1, 2)
        xyz = self._d(xyz_left).reshape(-1, 3)
        stored = np.zeros(3, np.int64)
        self._chk(self.lib.orc_ba_add_measurements(self.h, int(pose_id), len(lm_id), _p(lm_id, _i64p), _p(uvl, _f32p),
                                                   _p(uvr, _f32p), _p(xyz, _f64p), _p(stored, _i64p)),
                  "add_measurements")
        return stored

    def initialize(self):
        self._chk(self.lib.orc_ba_initialize(self.h), "initialize")

    def optimize(self, iterations):
        r = self.lib.orc_ba_optimize(self.h, int(iterations))
        if r < 0:
            raise ValueError("oracle: optimize before initialize")
        return r

    def optimize_until(self, ratio=0.99, first=1, block=10):
        nom = C.c_uint64(0)
        exe = C.c_uint64(0)
        self._chk(self.lib.orc_ba_optimize_until(self.h, ratio, first, block, C.byref(nom), C.byref(exe)),
                  "optimize_until")
        return nom.value, exe.value

    def chi2(self):
        p = C.c_double(0)
        r = C.c_double(0)
        self.lib.orc_ba_chi2(self.h, C.byref(p), C.byref(r))
        return p.value, r.value

    @property
    def last_plain_chi2(self):
        return self.lib.orc_ba_last_plain_chi2(self.h)

    @property
    def lm_lambda(self):
        return self.lib.orc_ba_lambda(self.h)

    @property
    def num_poses(self):
        return self.lib.orc_ba_num_poses(self.h)

    @property
    def num_landmarks(self):
        return self.lib.orc_ba_num_landmarks(self.h)

    @property
    def num_edges(self):
        return self.lib.orc_ba_num_edges(self.h)

    @property
    def system_size(self):
        return self.lib.orc_ba_system_size(self.h)

    @property
    def iterations(self):
        return self.lib.orc_ba_iterations(self.h)

    @property
    def trials(self):
        return self.lib.orc_ba_trials(self.h)

    def get_pose(self, id):
        T = np.zeros(12)
        self._chk(self.lib.orc_ba_get_pose(self.h, int(id), _p(T, _f64p)), "get_pose")
        return T

    def get_landmark(self, id):
        p = np.zeros(3)
        self._chk(self.lib.orc_ba_get_landmark(self.h, int(id), _p(p, _f64p)), "get_landmark")
        return p

    def get_poses(self):
        """(ids, T[n,12]) sorted by ascending id (the C-ABI's order)."""
        n = self.num_poses
        ids = np.zeros(n, np.int64)
        T = np.zeros((n, 12))
        self.lib.orc_ba_get_poses(self.h, _p(ids, _i64p), _p(T, _f64p))
        o = np.argsort(ids, kind="stable")
        return ids[o], T[o]

    def get_landmarks(self):
        n = self.num_landmarks
        ids = np.zeros(n, np.int64)
        p = np.zeros((n, 3))
        self.lib.orc_ba_get_landmarks(self.h, _p(ids, _i64p), _p(p, _f64p))
        o = np.argsort(ids, kind="stable")
        return ids[o], p[o]

    def get_edges(self):
        n = self.num_edges
        ty = np.zeros(n, np.int32)
        pid = np.zeros(n, np.int64)
        lid = np.zeros(n, np.int64)
        z = np.zeros((n, 3))
        info = np.zeros((n, 6))
        self.lib.orc_ba_get_edges(self.h, _p(ty, _i32p), _p(pid, _i64p), _p(lid, _i64p), _p(z, _f64p), _p(info, _f64p))
        return ty, pid, lid, z, info

    def get_aux(self):
        n = self.lib.orc_ba_num_aux(self.h)
        ty = np.zeros(n, np.int32)
        ia = np.zeros(n, np.int64)
        ib = np.zeros(n, np.int64)
        z = np.zeros((n, 12))
        info = np.zeros((n, 21))
        self.lib.orc_ba_get_aux(self.h, _p(ty, _i32p), _p(ia, _i64p), _p(ib, _i64p), _p(z, _f64p), _p(info, _f64p))
        return ty, ia, ib, z, info

    def edge_jacobians(self):
        n = self.num_edges
        e = np.zeros((n, 3))
        Jp = np.zeros((n, 3, 6))
        Jl = np.zeros((n, 3, 3))
        self.lib.orc_ba_edge_jacobians(self.h, _p(e, _f64p), _p(Jp, _f64p), _p(Jl, _f64p))
        return e, Jp, Jl

    def dense_system(self):
        """H (n,n), b (n), pose_col (by insertion), lm_col (by insertion) at the current estimate."""
        n = self.system_size
        H = np.zeros((n, n))
        b = np.zeros(n)
        pc = np.zeros(self.num_poses, np.int32)
        lc = np.zeros(self.num_landmarks, np.int32)
        r = self.lib.orc_ba_dense_system(self.h, _p(H, _f64p), _p(b, _f64p), n, _p(pc, _i32p), _p(lc, _i32p))
        if r < 0:
            raise ValueError("oracle: dense_system failed")
        return H, b, pc, lc

    def trace(self):
        n = self.lib.orc_ba_trace(self.h, None, 0)
        out = np.zeros(n)
        self.lib.orc_ba_trace(self.h, _p(out, _f64p), n)
        return out.reshape(-1, 5)

    def prune_diverged(self):
        return self.lib.orc_ba_prune_diverged(self.h)


# ------------------------------------------------------------------------------------------------
# temporal tracking schedule (oracle_track.c)
# ------------------------------------------------------------------------------------------------
TRACK_RECORD = np.dtype([("xyz_left", "<f8", 3), ("line", "<f8", 3), ("s3_start", "<f8"), ("uv_left", "<f4", 2), ("uv_right", "<f4", 2),
                         ("search_range", "<f4"), ("s1_roi_left", "<f4", 2), ("s1_roi_right", "<f4", 2), ("s2_left", "<f4", 4),
                         ("s2_right", "<f4", 4), ("s2_ext_left", "<f4", 4), ("s2_ext_right", "<f4", 4), ("s3_count", "<i4"),
                         ("s3_axis", "<i4"), ("status", "<i4")])
(M_OK, M_EMPTY_POOL, M_DISTANCE, M_ORIGINAL, M_RANGE, M_DISPARITY, M_DEPTH, M_OTHER, M_SKIPPED) = range(9)
FOV_LEFT, FOV_RIGHT, EPI_NO_MOTION, EPI_OUT_OF_SIGHT, EPI_BAD_PROJ, EPI_ZERO_LENGTH, EPI_OK = 1, 2, 4, 8, 16, 32, 64


class _Cam(C.Structure):
    _fields_ = [("P_left", C.c_double * 12), ("P_right", C.c_double * 12), ("K_inv", C.c_double * 9), ("width", C.c_double),
                ("height", C.c_double)]


class _StereoParams(C.Structure):
    _fields_ = [("f", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("duR_flipped", C.c_double), ("min_disparity", C.c_double),
                ("depth_min", C.c_double), ("depth_max", C.c_double), ("cutoff_match", C.c_int), ("cutoff_other", C.c_int),
                ("other_inclusive", C.c_int), ("search_in_left", C.c_int)]


def track_camera(P_left, P_right, K_inv, width, height):
    c = _Cam()
    c.P_left[:] = np.asarray(P_left, np.float64).ravel().tolist()
    c.P_right[:] = np.asarray(P_right, np.float64).ravel().tolist()
    c.K_inv[:] = np.asarray(K_inv, np.float64).ravel().tolist()
    c.width, c.height = float(width), float(height)
    return c


def _a(x, dt, shape=None):
    if x is None:
        return None
    x = np.ascontiguousarray(x, dt)
    return x if shape is None else x.reshape(shape)


def track_plan(cam, T_world_to_left, dp_T, motion_scaling, xyz_world, kp_size, last_disparity, uv_reference, dp_index, lib=None):
    lib = lib or load()
    assert lib.orc_track_record_size() == TRACK_RECORD.itemsize
    T = _a(T_world_to_left, np.float64, 12)
    dp = _a(dp_T, np.float64, (-1, 12))
    xyz = _a(xyz_world, np.float64, (-1, 3))
    n = len(xyz)
    kp, dis, uvr, dpi = _a(kp_size, np.float32), _a(last_disparity, np.float32), _a(uv_reference, np.float64, (-1, 2)), _a(dp_index, np.int32)
    rec = np.zeros(n, TRACK_RECORD)
    seg = np.zeros(n + 1, np.int32)
    lib.orc_track_plan(C.byref(cam), _p(T, _f64p), _p(dp, _f64p), len(dp), float(motion_scaling), _p(xyz, _f64p), _p(kp, _f32p), _p(dis, _f32p),
                       _p(uvr, _f64p), _p(dpi, _i32p), n, rec.ctypes.data, _p(seg, _i32p))
    return rec, seg


def track_epipolar_samples(cam, rec, kp_size, seg, depth, sel=None, lib=None):
    lib = lib or load()
    kp = _a(kp_size, np.float32)
    sel = _a(sel, np.int32)
    seg = _a(seg, np.int32)
    n_sel = len(seg) - 1
    out = np.zeros((int(seg[-1]), 2), np.float32)
    roi = np.zeros((n_sel, 4), np.float32)
    lib.orc_track_epipolar_samples(C.byref(cam), rec.ctypes.data, _p(kp, _f32p), _p(sel, _i32p), n_sel, _p(seg, _i32p), int(depth),
                                   _p(out, _f32p), _p(roi, _f32p))
    return out, roi


def track_stereo_range(width, in_left, uv_ref, topleft, kp_size, search_range=None, active=None, lib=None):
    lib = lib or load()
    uv, tl, kp = _a(uv_ref, np.float32, (-1, 2)), _a(topleft, np.float32, (-1, 2)), _a(kp_size, np.float32)
    sr, ac = _a(search_range, np.float32), _a(active, np.uint8)
    n = len(uv)
    seg, st, roi = np.zeros(n + 1, np.int32), np.zeros(n, np.int32), np.zeros((n, 4), np.float32)
    lib.orc_track_stereo_range(float(width), int(in_left), _p(uv, _f32p), _p(tl, _f32p), _p(kp, _f32p), _p(sr, _f32p), _p(ac, _u8p), n,
                               _p(seg, _i32p), _p(st, _i32p), _p(roi, _f32p))
    return seg, st, roi


def track_stereo_candidates(in_left, kp_size, seg, lib=None):
    lib = lib or load()
    kp, seg = _a(kp_size, np.float32), _a(seg, np.int32)
    out = np.zeros((int(seg[-1]), 2), np.float32)
    lib.orc_track_stereo_candidates(int(in_left), _p(kp, _f32p), len(kp), _p(seg, _i32p), _p(out, _f32p))
    return out


def match_ragged(q, original, seg, pool, cutoff, cutoff_original=257, active=None, lib=None):
    lib = lib or load()
    q, orig, pool = _a(q, np.uint8, (-1, 32)), _a(original, np.uint8), _a(pool, np.uint8)
    seg, ac = _a(seg, np.int32), _a(active, np.uint8)
    nq = len(q)
    idx, dist, st = np.zeros(nq, np.int32), np.zeros(nq, np.int32), np.zeros(nq, np.int32)
    lib.orc_match_ragged(_p(q, _u8p), _p(orig, _u8p), _p(ac, _u8p), nq, _p(seg, _i32p), _p(pool, _u8p), int(cutoff), int(cutoff_original),
                         _p(idx, _i32p), _p(dist, _i32p), _p(st, _i32p))
    return idx, dist, st


def stereo_params(f, cx, cy, duR_flipped, min_disparity, depth_min, depth_max, cutoff_match, cutoff_other, other_inclusive, in_left):
    return _StereoParams(f, cx, cy, duR_flipped, min_disparity, depth_min, depth_max, int(cutoff_match), int(cutoff_other),
                         int(other_inclusive), int(in_left))


def track_stereo_verify(prm, ref, last_other, uv_ref, topleft, seg, pool, pool_uv, active=None, lib=None):
    lib = lib or load()
    ref, lo, pool = _a(ref, np.uint8, (-1, 32)), _a(last_other, np.uint8), _a(pool, np.uint8)
    uv, tl, puv = _a(uv_ref, np.float32, (-1, 2)), _a(topleft, np.float32, (-1, 2)), _a(pool_uv, np.float32)
    seg, ac = _a(seg, np.int32), _a(active, np.uint8)
    nq = len(ref)
    idx, dist, st = np.zeros(nq, np.int32), np.zeros(nq, np.int32), np.zeros(nq, np.int32)
    uvo, xyz = np.zeros((nq, 2), np.float32), np.zeros((nq, 3))
    lib.orc_track_stereo_verify(C.byref(prm), _p(ref, _u8p), _p(lo, _u8p), _p(ac, _u8p), _p(uv, _f32p), _p(tl, _f32p), nq, _p(seg, _i32p),
                                _p(pool, _u8p), _p(puv, _f32p), _p(idx, _i32p), _p(dist, _i32p), _p(st, _i32p), _p(uvo, _f32p), _p(xyz, _f64p))
    return idx, dist, st, uvo, xyz


def track_handover(mode, rec, kp_size, sel=None, seg=None, pool_uv=None, idx=None, roi=None, lib=None):
    lib = lib or load()
    kp, sel, seg = _a(kp_size, np.float32), _a(sel, np.int32), _a(seg, np.int32)
    puv, idx, roi = _a(pool_uv, np.float32), _a(idx, np.int32), _a(roi, np.float32)
    n_sel = len(sel) if sel is not None else len(rec)
    uv, tl, ok = np.zeros((n_sel, 2), np.float32), np.zeros((n_sel, 2), np.float32), np.zeros(n_sel, np.uint8)
    lib.orc_track_handover(int(mode), rec.ctypes.data, _p(kp, _f32p), _p(sel, _i32p), n_sel, _p(seg, _i32p), _p(puv, _f32p), _p(idx, _i32p),
                           _p(roi, _f32p), _p(uv, _f32p), _p(tl, _f32p), _p(ok, _u8p))
    return uv, tl, ok


class NoMatch(Exception):
    """CExceptionNoMatchFound / CExceptionNoMatchFoundInternal with the status code the batched path reports."""

    def __init__(self, code):
        super().__init__(code)
        self.code = code


class OracleFundamentalMatcher:
    """The reference's per-landmark cascade, replayed ONE landmark at a time around the oracle_track.c functions:
    nested try / except exactly where CFundamentalMatcher.cpp has them.  extractor(side, roi[4], kp_uv[k,2]) ->
    (kept kp_uv, descriptors) and detector(side, rect[4]) -> kp_uv are host (numpy) callables."""

    def __init__(self, cam, stereo, lib=None):
        """stereo = dict(f, cx, cy, duR_flipped, min_disparity, depth_min, depth_max, width)"""
        self.cam, self.st, self.lib = cam, stereo, lib or load()

    # CTriangulator::getPointTriangulatedInRIGHT / InLEFT + the caller's depth / descriptor checks
    def _stereo(self, extractor, in_left, kp, rng, ref_desc, last_other, uv_ref, topleft, cutoff_other, inclusive):
        seg, st, roi = track_stereo_range(self.st["width"], in_left, uv_ref[None], topleft[None], [kp], [rng] if in_left else None, lib=self.lib)
        if st[0] != M_OK:
            raise NoMatch(int(st[0]))
        cand = track_stereo_candidates(in_left, [kp], seg, lib=self.lib)
        kp_uv, desc = extractor("left" if in_left else "right", roi[0], cand)
        prm = stereo_params(self.st["f"], self.st["cx"], self.st["cy"], self.st["duR_flipped"], self.st["min_disparity"], self.st["depth_min"],
                            self.st["depth_max"], 100, cutoff_other, inclusive, in_left)
        seg1 = np.array([0, len(kp_uv)], np.int32)
        idx, dist, st, uvo, xyz = track_stereo_verify(prm, ref_desc[None], None if last_other is None else last_other[None], uv_ref[None],
                                                      topleft[None], seg1, desc, kp_uv, lib=self.lib)
        if st[0] != M_OK:
            raise NoMatch(int(st[0]))
        return uvo[0], xyz[0], desc[idx[0]]

    def stage1(self, rec, kp_size, extractor, last_l, last_r):
        n = len(rec)
        out = [dict(status=M_SKIPPED) for _ in range(n)]
        for i in range(n):
            r = rec[i]
            if not (r["status"] & FOV_LEFT and r["status"] & FOV_RIGHT):                          # :389
                continue
            one = rec[i:i + 1]
            kp = np.float32(kp_size[i])
            for side in (0, 1):                                                                   # try LEFT, catch -> RIGHT
                try:
                    uv_ref, tl, _ = track_handover(side, one, [kp], lib=self.lib)
                    roi_xy = r["s1_roi_left" if side == 0 else "s1_roi_right"]
                    L = np.float32(8) * kp + np.float32(1)
                    roi = np.array([roi_xy[0], roi_xy[1], L, L], np.float32)
                    kp_uv, desc = extractor("left" if side == 0 else "right", roi, np.array([[np.float32(4) * kp, np.float32(4) * kp]], np.float32))
                    here, there = (last_l[i], last_r[i]) if side == 0 else (last_r[i], last_l[i])
                    idx, dist, st = match_ragged(here[None], None, [0, len(kp_uv)], desc, 25, lib=self.lib)   # 1 == rows && 25 > norm
                    if st[0] != M_OK:
                        raise NoMatch(int(st[0]))
                    uvo, xyz, d_other = self._stereo(extractor, side, kp, r["search_range"], desc[idx[0]], there, uv_ref[0], tl[0], 25, 1)
                    if side == 0:
                        out[i] = dict(status=M_OK, uv_left=r["uv_left"].copy(), uv_right=uvo, xyz=xyz, desc_left=desc[idx[0]], desc_right=d_other)
                    else:
                        out[i] = dict(status=M_OK, uv_left=uvo, uv_right=r["uv_right"].copy(), xyz=xyz, desc_left=d_other, desc_right=desc[idx[0]])
                    break
                except NoMatch as e:
                    out[i] = dict(status=e.code)
        return out

    def stage2(self, rec, kp_size, detector, extractor, last_l, last_r):
        n = len(rec)
        out = [dict(status=M_SKIPPED) for _ in range(n)]
        for i in range(n):
            r = rec[i]
            if not (r["status"] & FOV_LEFT and r["status"] & FOV_RIGHT):
                continue
            one = rec[i:i + 1]
            kp = np.float32(kp_size[i])
            for side in (0, 1):
                name = "left" if side == 0 else "right"
                try:
                    found = detector(name, r["s2_" + name])
                    shifted = (found + np.float32(4) * kp).astype(np.float32)                     # :533
                    c = np.rint(r["s2_ext_" + name]).astype(np.float32)                        # cv::Rect( Point2f, Point2f )
                    kp_uv, desc = extractor(name, np.array([c[0], c[1], c[2] - c[0], c[3] - c[1]], np.float32), shifted)
                    here, there = (last_l[i], last_r[i]) if side == 0 else (last_r[i], last_l[i])
                    seg = np.array([0, len(kp_uv)], np.int32)
                    idx, dist, st = match_ragged(here[None], None, seg, desc, 50, lib=self.lib)   # :540-545
                    if st[0] != M_OK:
                        raise NoMatch(int(st[0]))
                    uv_ref, tl, ok = track_handover(2 + side, one, [kp], None, seg, kp_uv, idx, lib=self.lib)
                    if not ok[0]:
                        raise NoMatch(M_RANGE)                                                    # "out of tracking range"
                    uvo, xyz, d_other = self._stereo(extractor, side, kp, r["search_range"], desc[idx[0]], there, uv_ref[0], tl[0], 50, 0)
                    if side == 0:
                        out[i] = dict(status=M_OK, uv_left=uv_ref[0], uv_right=uvo, xyz=xyz, desc_left=desc[idx[0]], desc_right=d_other)
                    else:
                        out[i] = dict(status=M_OK, uv_left=uvo, uv_right=uv_ref[0], xyz=xyz, desc_left=d_other, desc_right=desc[idx[0]])
                    break
                except NoMatch as e:
                    out[i] = dict(status=e.code)
        return out

    def epipolar(self, rec, kp_size, extractor, last_l, ref_l):
        n = len(rec)
        out = [dict(status=M_SKIPPED) for _ in range(n)]
        for i in range(n):
            r = rec[i]
            if not r["status"] & EPI_OK:
                continue
            one = rec[i:i + 1]
            kp = np.float32(kp_size[i])
            seg = np.array([0, r["s3_count"]], np.int32)
            try:
                depth = 0
                while True:                                                                       # _getMatchSampleRecursiveU/V
                    samples, roi = track_epipolar_samples(self.cam, one, [kp], seg, depth, lib=self.lib)
                    kp_uv, desc = extractor("left", roi[0], samples)
                    seg_e = np.array([0, len(kp_uv)], np.int32)
                    idx, dist, st = match_ragged(last_l[i][None], ref_l[i][None], seg_e, desc, 50, 100, lib=self.lib)   # _getMatch
                    if st[0] == M_OK:
                        break
                    if depth == 2:                                                                # m_uRecursionLimitEpipolarLines
                        raise NoMatch(int(st[0]))
                    depth += 2
                uv_ref, tl, ok = track_handover(4, one, [kp], None, seg_e, kp_uv, idx, roi, lib=self.lib)
                uvo, xyz, d_other = self._stereo(extractor, 0, kp, r["search_range"], desc[idx[0]], None, uv_ref[0], tl[0], -1, 0)
                out[i] = dict(status=M_OK, uv_left=uv_ref[0], uv_right=uvo, xyz=xyz, desc_left=desc[idx[0]], desc_right=d_other)
            except NoMatch as e:
                out[i] = dict(status=e.code)
        return out

    def manual(self, rec, kp_size, detector, extractor, last_l, last_r, ref_l):
        """trackManual (:1366-2019): per landmark stage 1 LEFT/RIGHT, then stage 2 LEFT/RIGHT, then the epipolar search"""
        s1 = self.stage1(rec, kp_size, extractor, last_l, last_r)
        out = []
        for i in range(len(rec)):
            d = dict(s1[i], stage=1 if s1[i]["status"] == M_OK else 0)
            if s1[i]["status"] not in (M_OK, M_SKIPPED):
                one = rec[i:i + 1]
                s2 = self.stage2(one, kp_size[i:i + 1], detector, extractor, last_l[i:i + 1], last_r[i:i + 1])[0]
                d = dict(s2, stage=2 if s2["status"] == M_OK else 0)
                if s2["status"] != M_OK:
                    s3 = self.epipolar(one, kp_size[i:i + 1], extractor, last_l[i:i + 1], ref_l[i:i + 1])[0]
                    if s3["status"] != M_SKIPPED:
                        d = dict(s3, stage=3 if s3["status"] == M_OK else 0)
            out.append(d)
        return out

    def new_landmarks(self, extractor, uv_left, kp_size, desc_left):
        """addNewLandmarks (:109-175): getPointTriangulatedInRIGHTFull for every detected key point"""
        out = []
        prm_depth = dict(self.st)
        for i in range(len(uv_left)):
            kp = np.float32(kp_size[i])
            half = np.float32(4) * kp
            tl = np.array([max(np.float32(0), np.float32(uv_left[i][0]) - np.float32(60.0) - half), np.float32(uv_left[i][1]) - half], np.float32)
            st = self.st
            try:
                self.st = dict(prm_depth, depth_min=-1.0e300, depth_max=1.0e300)
                uvo, xyz, d_other = self._stereo(extractor, 0, kp, np.float32(0), desc_left[i], None, np.asarray(uv_left[i], np.float32), tl, -1, 0)
                out.append(dict(status=M_OK, uv_left=np.asarray(uv_left[i], np.float32), uv_right=uvo, xyz=xyz, desc_left=desc_left[i], desc_right=d_other))
            except NoMatch as e:
                out.append(dict(status=e.code))
            finally:
                self.st = st
        return out


# ------------------------------------------------------------------------------------------------
# CSolverStereoPosit (oracle_posit.c)
# ------------------------------------------------------------------------------------------------
class _PositParams(C.Structure):
    _fields_ = [("P_left", C.c_double * 12), ("P_right", C.c_double * 12), ("min_points", C.c_int), ("min_inliers", C.c_int),
                ("max_iterations", C.c_int), ("max_error_inlier_l2", C.c_double), ("max_error_average_l2", C.c_double),
                ("max_risk", C.c_double), ("convergence_delta", C.c_double), ("min_translation_l2", C.c_double)]


class _PositResult(C.Structure):
    _fields_ = [("T", C.c_double * 12), ("error_average", C.c_double), ("risk", C.c_double), ("status", C.c_int32),
                ("iterations", C.c_int32), ("inliers", C.c_int32), ("n", C.c_int32)]


def posit_params(P_left, P_right, **kw):
    p = _PositParams()
    p.P_left[:] = np.asarray(P_left, np.float64).ravel().tolist()
    p.P_right[:] = np.asarray(P_right, np.float64).ravel().tolist()
    p.min_points, p.min_inliers, p.max_iterations = 25, 15, 1000               # CSolverStereoPosit.h:89-91
    p.max_error_inlier_l2, p.max_error_average_l2, p.max_risk = 10.0, 9.0, 2.0  # :92-94
    p.convergence_delta, p.min_translation_l2 = 1e-5, 0.001                     # :95, :98
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def stereo_posit(prm, T_last, t_imu, T_estimate, xyz_world, uv_left, uv_right, active=None, lib=None):
    lib = lib or load()
    Tl, Te, ti = _a(T_last, np.float64, 12), _a(T_estimate, np.float64, 12), _a(t_imu, np.float64, 3)
    x, ul, ur, ac = _a(xyz_world, np.float64, (-1, 3)), _a(uv_left, np.float32, (-1, 2)), _a(uv_right, np.float32, (-1, 2)), _a(active, np.uint8)
    r = _PositResult()
    lib.orc_stereo_posit(C.byref(prm), _p(Tl, _f64p), _p(ti, _f64p), _p(Te, _f64p), _p(x, _f64p), _p(ul, _f32p), _p(ur, _f32p), _p(ac, _u8p),
                         len(x), C.byref(r))
    return dict(T=np.array(r.T[:]), error_average=r.error_average, risk=r.risk, status=r.status, iterations=r.iterations,
                inliers=r.inliers, n=r.n)


# ------------------------------------------------------------------------------------------------
# CLandmark::optimize (oracle_landmark.c)
# ------------------------------------------------------------------------------------------------
class _LandmarkParams(C.Structure):
    _fields_ = [("min_measurements", C.c_int), ("cap_iterations", C.c_int), ("convergence_delta", C.c_double),
                ("kernel_max_error_l2", C.c_double), ("min_inlier_ratio", C.c_double), ("max_error_average_l2", C.c_double)]


def landmark_params(**kw):
    p = _LandmarkParams(5, 1000, 1e-5, 10.0, 0.5, 9.0)     # CLandmark.h:90-98
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def landmarks_optimize(prm, frame_P_left, frame_P_right, seg, meas_frame, uvl, uvr, xyz, lib=None):
    lib = lib or load()
    PL, PR = _a(frame_P_left, np.float64, (-1, 12)), _a(frame_P_right, np.float64, (-1, 12))
    seg, fr = _a(seg, np.int32), _a(meas_frame, np.int32)
    ul, ur, x = _a(uvl, np.float32, (-1, 2)), _a(uvr, np.float32, (-1, 2)), _a(xyz, np.float64, (-1, 3))
    n = len(x)
    out, st, err, its = np.zeros((n, 3)), np.zeros(n, np.int32), np.zeros(n), np.zeros(n, np.int32)
    lib.orc_landmarks_optimize(C.byref(prm), _p(PL, _f64p), _p(PR, _f64p), len(PL), _p(seg, _i32p), _p(fr, _i32p), _p(ul, _f32p), _p(ur, _f32p),
                               _p(x, _f64p), n, _p(out, _f64p), _p(st, _i32p), _p(err, _f64p), _p(its, _i32p))
    return out, st, err, its


# ------------------------------------------------------------------------------------------------
# BRIEF-256 extraction (oracle_brief.c)
# ------------------------------------------------------------------------------------------------
def brief_integral(image, lib=None):
    lib = lib or load()
    img = np.ascontiguousarray(image, np.uint8)
    h, w = img.shape
    out = np.zeros((h + 1, w + 1), np.int32)
    lib.orc_brief_integral(_p(img, _u8p), w, h, w, _p(out, _i32p))
    return out


def brief_compute(integral, pattern, roi, seg, kp_uv, lib=None):
    """roi n x 4 (x, y, w, h; floats are truncated like cv::Rect does); -> (seg_out, kp_out, desc)"""
    lib = lib or load()
    h, w = integral.shape[0] - 1, integral.shape[1] - 1
    pat = np.ascontiguousarray(pattern, np.int8).reshape(1024)
    roi_i = np.ascontiguousarray(np.trunc(np.asarray(roi, np.float64)).astype(np.int32)).reshape(-1, 4)
    seg, kp = _a(seg, np.int32), _a(kp_uv, np.float32, (-1, 2))
    n = len(roi_i)
    seg_out = np.zeros(n + 1, np.int32)
    kp_out = np.zeros((max(len(kp), 1), 2), np.float32)
    desc = np.zeros((max(len(kp), 1), 32), np.uint8)
    kept = lib.orc_brief_compute(_p(integral, _i32p), w, h, pat.ctypes.data, _p(roi_i, _i32p), _p(seg, _i32p), _p(kp, _f32p), n, _p(seg_out, _i32p),
                                 _p(kp_out, _f32p), _p(desc, _u8p))
    return seg_out, kp_out[:kept], desc[:kept]
