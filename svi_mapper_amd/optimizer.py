"""Host-side mirror of the reference's optimizer interface over the C ABI.

Reference shape mirrored here: Cg2oOptimizer (src/optimization/Cg2oOptimizer.h:63-73, 129-203)
  addLandmarkToGraph(landmark, shift)            -> BundleAdjuster.add_landmark / add_landmarks
  optimize(frame, begin, nClosed, shift)         -> add_keyframe + add_measurements per new keyframe, then
                                                    initialize() + optimize_until()  (= _optimizeUnLimited, :954-980)
  getNumberOfOptimizations / getDurationTotal... -> num_optimizations / duration_total_seconds
and one level down the g2o calls it makes (addVertex / addEdge / initializeOptimization / optimize / chi2).
All arithmetic happens in libsvi_hot.so on the GPU; this file only marshals arrays.
"""
import ctypes as C
import time

import numpy as np

from . import _capi
from ._capi import BaOptions, BaStats, SviError, check, f32p, f64p, i32p, i64p  # noqa: F401


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _d(a, n=None):
    a = np.ascontiguousarray(a, np.float64)
    return a.reshape(n) if n is not None else a


class BundleAdjuster:
    """LM bundle adjustment with the reference's graph rules, on one MI355X (or one shard of a node)."""

    def __init__(self, fx, fy, cx, cy, baseline_m, device=0, stream=None, rank=0, n_ranks=1, profile=False,
                 chol_tile=48, sweep_events=False, **lm):
        self._lib = _capi.load_library()
        o = BaOptions()
        self._lib.svi_ba_options_default(C.byref(o))
        o.fx, o.fy, o.cx, o.cy, o.baseline_m = fx, fy, cx, cy, baseline_m
        o.device, o.rank, o.n_ranks, o.profile, o.chol_tile = device, rank, n_ranks, int(profile), chol_tile
        o.stream = stream
        o.sweep_events = int(sweep_events)
        for k, v in lm.items():
            if not hasattr(o, k):
                raise TypeError("unknown option %r" % k)
            setattr(o, k, v)
        h = C.c_void_p()
        check(self._lib.svi_ba_create(C.byref(o), C.byref(h)), "svi_ba_create")
        self._h = h
        self._hook = None
        self.num_optimizations = 0          # Cg2oOptimizer::getNumberOfOptimizations
        self.duration_total_seconds = 0.0   # Cg2oOptimizer::getDurationTotalSecondsOptimization

    def close(self):
        if getattr(self, "_h", None):
            self._lib.svi_ba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- graph construction ----------------------------------------------------------------------
    def add_pose(self, id, T, fixed=False):
        T = _d(T, 12)
        check(self._lib.svi_ba_add_pose(self._h, int(id), _p(T, f64p), int(fixed)), "svi_ba_add_pose")

    def add_landmark(self, id, p, fixed=False):
        p = _d(p, 3)
        check(self._lib.svi_ba_add_landmark(self._h, int(id), _p(p, f64p), int(fixed)), "svi_ba_add_landmark")

    def add_landmarks(self, ids, p, fixed=False):
        p = _d(p).reshape(-1, 3)
        for i, pid in enumerate(ids):
            check(self._lib.svi_ba_add_landmark(self._h, int(pid), _p(p[i], f64p), int(fixed)), "svi_ba_add_landmark")

    def add_edges_bulk(self, type, pose_id, lm_id, z, info_upper, robust=1):
        ty = np.ascontiguousarray(type, np.int32)
        pid = np.ascontiguousarray(pose_id, np.int64)
        lid = np.ascontiguousarray(lm_id, np.int64)
        z = _d(z).reshape(-1, 3)
        info = _d(info_upper).reshape(-1, 6)
        rb = np.ascontiguousarray(np.broadcast_to(np.asarray(robust, np.int32), (len(ty),)))
        if not (len(pid) == len(lid) == len(z) == len(info) == len(ty)):
            raise ValueError("edge arrays differ in length")
        check(self._lib.svi_ba_add_edges_bulk(self._h, len(ty), _p(ty, i32p), _p(pid, i64p), _p(lid, i64p), _p(z, f64p),
                                              _p(info, f64p), _p(rb, i32p)), "svi_ba_add_edges_bulk")

    def add_edge_se3(self, i, j, Z, info_upper, robust=False):
        Z = _d(Z, 12)
        info = _d(info_upper, 21)
        check(self._lib.svi_ba_add_edge_se3(self._h, int(i), int(j), _p(Z, f64p), _p(info, f64p), int(robust)),
              "svi_ba_add_edge_se3")

    def add_edge_accel(self, pose_id, a, off=None, info_upper=(1, 0, 0, 1, 0, 1)):
        a = _d(a, 3)
        info = _d(info_upper, 6)
        off = _d(off, 12) if off is not None else None
        check(self._lib.svi_ba_add_edge_accel(self._h, int(pose_id), _p(a, f64p), _p(off, f64p), _p(info, f64p)),
              "svi_ba_add_edge_accel")

    def add_edge_lm_lm(self, i, j, z, info_upper, robust=True):
        z = _d(z, 3)
        info = _d(info_upper, 6)
        check(self._lib.svi_ba_add_edge_lm_lm(self._h, int(i), int(j), _p(z, f64p), _p(info, f64p), int(robust)),
              "svi_ba_add_edge_lm_lm")

    def set_imu_offset(self, off):
        """IMU->LEFT offset parameter (Cg2oOptimizer.cpp:213) of the gravity edges that add_keyframe creates"""
        off = _d(off, 12)
        check(self._lib.svi_ba_set_imu_offset(self._h, _p(off, f64p)), "svi_ba_set_imu_offset")

    def add_keyframe(self, id, from_id, T, shift=None, accel=None):
        T = _d(T, 12)
        shift = _d(shift, 3) if shift is not None else None
        accel = _d(accel, 3) if accel is not None else None
        check(self._lib.svi_ba_add_keyframe(self._h, int(id), int(from_id), _p(T, f64p), _p(shift, f64p), _p(accel, f64p)),
              "svi_ba_add_keyframe")

    def add_measurements(self, pose_id, lm_id, uv_left, uv_right, xyz_left):
        lm_id = np.ascontiguousarray(lm_id, np.int64)
        uvl = np.ascontiguousarray(uv_left, np.float32).reshape(-1, 2)
        uvr = np.ascontiguousarray(uv_right, np.float32).reshape(-1, 2)
        xyz = _d(xyz_left).reshape(-1, 3)
        if not (len(uvl) == len(uvr) == len(xyz) == len(lm_id)):
            raise ValueError("measurement arrays differ in length")
        stored = np.zeros(3, np.int64)
        check(self._lib.svi_ba_add_measurements(self._h, int(pose_id), len(lm_id), _p(lm_id, i64p), _p(uvl, f32p),
                                                _p(uvr, f32p), _p(xyz, f64p), _p(stored, i64p)), "svi_ba_add_measurements")
        return stored

    def load_g2o(self, path):
        check(self._lib.svi_ba_load_g2o(self._h, str(path).encode()), "svi_ba_load_g2o")

    def save_g2o(self, path):
        check(self._lib.svi_ba_save_g2o(self._h, str(path).encode()), "svi_ba_save_g2o")

    # -- optimisation ------------------------------------------------------------------------------
    def initialize(self):
        check(self._lib.svi_ba_initialize(self._h), "svi_ba_initialize")

    def optimize(self, iterations):
        done = C.c_int(0)
        check(self._lib.svi_ba_optimize(self._h, int(iterations), C.byref(done)), "svi_ba_optimize")
        return done.value

    def optimize_until(self, ratio=0.99, first=1, block=10):
        """Cg2oOptimizer::_optimizeUnLimited. Returns (nominal, executed) iteration counts."""
        nom = C.c_uint64(0)
        exe = C.c_uint64(0)
        t0 = time.time()
        check(self._lib.svi_ba_optimize_until(self._h, float(ratio), int(first), int(block), C.byref(nom), C.byref(exe)),
              "svi_ba_optimize_until")
        self.duration_total_seconds += time.time() - t0
        self.num_optimizations += 1
        return nom.value, exe.value

    def sync_host(self):
        """Refresh the host copy of the estimates now (the getters do it on demand; collective with several ranks)."""
        check(self._lib.svi_ba_sync_host(self._h), "svi_ba_sync_host")

    def chi2(self):
        p = C.c_double(0)
        r = C.c_double(0)
        check(self._lib.svi_ba_chi2(self._h, C.byref(p), C.byref(r)), "svi_ba_chi2")
        return p.value, r.value

    @property
    def last_plain_chi2(self):
        return self.chi2()[0]

    @property
    def lm_lambda(self):
        v = C.c_double(0)
        check(self._lib.svi_ba_lambda(self._h, C.byref(v)), "svi_ba_lambda")
        return v.value

    def prune_diverged(self):
        n = C.c_int64(0)
        check(self._lib.svi_ba_prune_diverged(self._h, C.byref(n)), "svi_ba_prune_diverged")
        return n.value

    def apply_optimization(self, shift=None):
        """Cg2oOptimizer::_applyOptimizationToLandmarks / ToKeyFrames (Cg2oOptimizer.cpp:1468-1540):
        returns dict(lm_ids, lm_xyz, lm_kept, kf_ids, kf_T, erased); diverged landmarks leave the graph."""
        nl, npz = self.num_landmarks, self.num_poses
        lm_ids, lm_xyz, kept = np.empty(nl, np.int64), np.empty((nl, 3)), np.empty(nl, np.uint8)
        kf_ids, kf_T = np.empty(npz, np.int64), np.empty((npz, 12))
        sh = None if shift is None else np.ascontiguousarray(shift, np.float64).reshape(3)
        erased = C.c_int64(0)
        check(self._lib.svi_ba_apply_optimization(self._h, sh.ctypes.data_as(_capi.f64p) if sh is not None else None,
                                                  lm_ids.ctypes.data_as(_capi.i64p), lm_xyz.ctypes.data_as(_capi.f64p),
                                                  kept.ctypes.data_as(_capi.u8p), kf_ids.ctypes.data_as(_capi.i64p),
                                                  kf_T.ctypes.data_as(_capi.f64p), C.byref(erased)), "svi_ba_apply_optimization")
        return dict(lm_ids=lm_ids, lm_xyz=lm_xyz, lm_kept=kept, kf_ids=kf_ids, kf_T=kf_T, erased=erased.value)

    # -- results -----------------------------------------------------------------------------------
    def _count(self, fn):
        n = C.c_int64(0)
        check(fn(self._h, C.byref(n)), "count")
        return n.value

    @property
    def num_poses(self):
        return self._count(self._lib.svi_ba_num_poses)

    @property
    def num_landmarks(self):
        return self._count(self._lib.svi_ba_num_landmarks)

    @property
    def num_edges(self):
        return self.stats().n_edges_proj

    def get_pose(self, id):
        T = np.zeros(12)
        check(self._lib.svi_ba_get_pose(self._h, int(id), _p(T, f64p)), "svi_ba_get_pose")
        return T

    def get_landmark(self, id):
        p = np.zeros(3)
        check(self._lib.svi_ba_get_landmark(self._h, int(id), _p(p, f64p)), "svi_ba_get_landmark")
        return p

    def get_poses(self):
        n = self.num_poses
        ids = np.zeros(n, np.int64)
        T = np.zeros((n, 12))
        check(self._lib.svi_ba_get_poses(self._h, _p(ids, i64p), _p(T, f64p)), "svi_ba_get_poses")
        return ids, T

    def get_landmarks(self):
        n = self.num_landmarks
        ids = np.zeros(n, np.int64)
        p = np.zeros((n, 3))
        check(self._lib.svi_ba_get_landmarks(self._h, _p(ids, i64p), _p(p, f64p)), "svi_ba_get_landmarks")
        return ids, p

    # -- multi-GPU hook, instrumentation, debug taps -----------------------------------------------
    def set_allreduce(self, fn):
        """fn(dev_ptr:int, count:int, stream:int) -> 0 on success; kept alive by this object."""
        def tramp(_user, buf, count, stream):
            try:
                return int(fn(buf, count, stream) or 0)
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._hook = _capi.ALLREDUCE_FN(tramp)
        check(self._lib.svi_ba_set_allreduce(self._h, self._hook, None), "svi_ba_set_allreduce")

    def set_allreduce_native(self, comm):
        """the library's own RCCL hook (csrc/rccl_hook.cpp): comm = svi_mapper_amd.dist.NativeRccl; no Python on the data path"""
        self._hook = comm           # keeps the communicator alive
        fn = C.cast(self._lib.svi_rccl_allreduce, _capi.ALLREDUCE_FN)
        check(self._lib.svi_ba_set_allreduce(self._h, fn, comm.handle), "svi_ba_set_allreduce")

    def stats(self):
        s = BaStats()
        check(self._lib.svi_ba_get_stats(self._h, C.byref(s)), "svi_ba_get_stats")
        return s

    def phase_times(self):
        ms = np.zeros(len(_capi.SVI_PH_NAMES))
        calls = np.zeros(len(_capi.SVI_PH_NAMES), np.int64)
        check(self._lib.svi_ba_get_phase_times(self._h, _p(ms, f64p), _p(calls, i64p)), "svi_ba_get_phase_times")
        return {n: (float(m), int(c)) for n, m, c in zip(_capi.SVI_PH_NAMES, ms, calls)}

    def reset_phase_times(self):
        check(self._lib.svi_ba_reset_phase_times(self._h), "svi_ba_reset_phase_times")

    def time_sweep(self, reps=50, part=0):
        """mean ms of the Jacobian sweep (HIP events around `reps` back-to-back launches); part = 1 / 2: only the
        landmark-major (K2) / pose-major (K3) kernel"""
        v = C.c_double(0)
        if part:
            check(self._lib.svi_ba_debug_time_sweep_part(self._h, int(reps), int(part), C.byref(v)), "svi_ba_debug_time_sweep_part")
        else:
            check(self._lib.svi_ba_debug_time_sweep(self._h, int(reps), C.byref(v)), "svi_ba_debug_time_sweep")
        return v.value

    def time_sweep_cold(self, reps=20, evict_bytes=640 << 20):
        """mean ms of the sweep with L2 / Infinity Cache flushed before every launch (each sweep between its own events)"""
        v = C.c_double(0)
        check(self._lib.svi_ba_debug_time_sweep_cold(self._h, int(reps), int(evict_bytes), C.byref(v)), "svi_ba_debug_time_sweep_cold")
        return v.value

    def sweep_time(self):
        """(total ms, launches) of the Jacobian sweep inside the LM loop since the last reset (sweep_events / profile)"""
        ms = C.c_double(0)
        n = C.c_int64(0)
        check(self._lib.svi_ba_get_sweep_time(self._h, C.byref(ms), C.byref(n)), "svi_ba_get_sweep_time")
        return ms.value, n.value

    def edge_jacobians(self):
        n = self.stats().n_edges_proj
        e = np.zeros((n, 3))
        Jp = np.zeros((n, 3, 6))
        Jl = np.zeros((n, 3, 3))
        check(self._lib.svi_ba_debug_edge_jacobians(self._h, _p(e, f64p), _p(Jp, f64p), _p(Jl, f64p)),
              "svi_ba_debug_edge_jacobians")
        return e, Jp, Jl

    def aux_jacobians(self):
        """pose-only edges in insertion order among their kind: (se3_err n x 6, Ji n x 6 x 6, Jj, acc_err m x 3, acc_J m x 3 x 6)"""
        st = self.stats()
        ns, na = st.n_edges_se3, st.n_edges_accel
        se, si, sj = np.zeros((ns, 6)), np.zeros((ns, 6, 6)), np.zeros((ns, 6, 6))
        ae, aj = np.zeros((na, 3)), np.zeros((na, 3, 6))
        check(self._lib.svi_ba_debug_aux_jacobians(self._h, _p(se, f64p), _p(si, f64p), _p(sj, f64p), _p(ae, f64p), _p(aj, f64p)),
              "svi_ba_debug_aux_jacobians")
        return se, si, sj, ae, aj

    def reduced_system(self, lam):
        n = 6 * self.stats().n_poses_free
        S = np.zeros((n, n))
        g = np.zeros(n)
        nn = C.c_int64(0)
        check(self._lib.svi_ba_debug_reduced_system(self._h, float(lam), _p(S, f64p), _p(g, f64p), n, C.byref(nn)),
              "svi_ba_debug_reduced_system")
        return S, g
