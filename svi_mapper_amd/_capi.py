"""ctypes binding of libsvi_hot.so (include/svi_hot.h).

This module is deliberately thin: it loads the in-tree shared library and declares argument
types. There is NO fallback: if the library is missing, import fails; if no gfx950 device is
visible, every compute entry point returns SVI_ERR_NO_DEVICE and the wrappers raise SviError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVI_HOT_LIB") or os.path.join(_HERE, "lib", "libsvi_hot.so")   # (the override is for A/B timing of builds)

SVI_OK = 0
SVI_PH_NAMES = ("linearize_lm", "linearize_pose", "pose_edges", "schur", "assemble", "allreduce",
                "cholesky", "backsub_update", "chi2")

f64p = C.POINTER(C.c_double)
f32p = C.POINTER(C.c_float)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p


class SviError(RuntimeError):
    def __init__(self, status, where, detail):
        super().__init__("%s: %s (%d)%s" % (where, _status_string(status), status, (": " + detail) if detail else ""))
        self.status = status


class Gate(C.Structure):
    _fields_ = [("q_uv", vp), ("t_uv", vp), ("q_umin", vp), ("q_umax", vp), ("v_tol", C.c_float)]


class TrackCamera(C.Structure):
    _fields_ = [("P_left", C.c_double * 12), ("P_right", C.c_double * 12), ("K_inv", C.c_double * 9),
                ("width", C.c_double), ("height", C.c_double)]


class TrackLandmarks(C.Structure):
    _fields_ = [("n", C.c_int), ("xyz_world", vp), ("kp_size", vp), ("last_disparity", vp), ("uv_reference", vp), ("dp_index", vp),
                ("last_desc_left", vp), ("last_desc_right", vp), ("ref_desc_left", vp)]


class TrackResult(C.Structure):
    _fields_ = [("status", vp), ("stage", vp), ("uv_left", vp), ("uv_right", vp), ("xyz_left", vp), ("desc_left", vp), ("desc_right", vp)]


EXTRACT_FN = C.CFUNCTYPE(C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int64, vp, vp, vp, C.POINTER(C.c_int64), vp)
DETECT_FN = C.CFUNCTYPE(C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int64, vp, vp, C.POINTER(C.c_int64), vp)


class TrackStereoParams(C.Structure):
    _fields_ = [("f", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("duR_flipped", C.c_double),
                ("min_disparity", C.c_double), ("depth_min", C.c_double), ("depth_max", C.c_double),
                ("cutoff_match", C.c_int), ("cutoff_other", C.c_int), ("other_inclusive", C.c_int),
                ("search_in_left", C.c_int)]


# numpy view of svi_track_record (168 bytes, include/svi_hot.h)
TRACK_RECORD_FIELDS = [("xyz_left", "<f8", 3), ("line", "<f8", 3), ("s3_start", "<f8"), ("uv_left", "<f4", 2),
                       ("uv_right", "<f4", 2), ("search_range", "<f4"), ("s1_roi_left", "<f4", 2),
                       ("s1_roi_right", "<f4", 2), ("s2_left", "<f4", 4), ("s2_right", "<f4", 4),
                       ("s2_ext_left", "<f4", 4), ("s2_ext_right", "<f4", 4), ("s3_count", "<i4"), ("s3_axis", "<i4"),
                       ("status", "<i4")]
TRACK_RECORD_SIZE = 168

# SVI_TRK_* status bits / SVI_TRK_MATCH_* codes
TRK_FOV_LEFT, TRK_FOV_RIGHT, TRK_EPI_NO_MOTION, TRK_EPI_OUT_OF_SIGHT = 1, 2, 4, 8
TRK_EPI_BAD_PROJ, TRK_EPI_ZERO_LENGTH, TRK_EPI_OK = 16, 32, 64
(MATCH_OK, MATCH_EMPTY_POOL, MATCH_DISTANCE, MATCH_ORIGINAL, MATCH_RANGE, MATCH_DISPARITY, MATCH_DEPTH,
 MATCH_OTHER_MISMATCH, MATCH_SKIPPED) = range(9)


class PositParams(C.Structure):
    _fields_ = [("P_left", C.c_double * 12), ("P_right", C.c_double * 12), ("min_points", C.c_int), ("min_inliers", C.c_int),
                ("max_iterations", C.c_int), ("max_error_inlier_l2", C.c_double), ("max_error_average_l2", C.c_double),
                ("max_risk", C.c_double), ("convergence_delta", C.c_double), ("min_translation_l2", C.c_double)]


class PositResult(C.Structure):
    _fields_ = [("T_world_to_left", C.c_double * 12), ("error_average", C.c_double), ("risk", C.c_double), ("status", C.c_int32),
                ("iterations", C.c_int32), ("inliers", C.c_int32), ("n", C.c_int32)]


class LandmarkParams(C.Structure):
    _fields_ = [("min_measurements", C.c_int), ("cap_iterations", C.c_int), ("convergence_delta", C.c_double),
                ("kernel_max_error_l2", C.c_double), ("min_inlier_ratio", C.c_double), ("max_error_average_l2", C.c_double)]


class BaOptions(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("baseline_m", C.c_double), ("cauchy_delta", C.c_double), ("lm_tau", C.c_double),
                ("lm_good_step_lower", C.c_double), ("lm_good_step_upper", C.c_double),
                ("lm_max_trials", C.c_int),
                ("max_depth_xyz_l2", C.c_double), ("max_depth_uvdepth_l2", C.c_double),
                ("max_depth_uvdisp_l2", C.c_double), ("sane_position_l2", C.c_double),
                ("device", C.c_int), ("stream", vp), ("rank", C.c_int), ("n_ranks", C.c_int),
                ("profile", C.c_int), ("chol_tile", C.c_int), ("chol_order", C.c_int), ("sweep_events", C.c_int)]


class BaStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("n_poses", "n_poses_free", "n_landmarks", "n_landmarks_local", "n_edges_proj",
                 "n_edges_proj_local", "n_edges_se3", "n_edges_accel", "n_edges_lmlm", "n_schur_tiles",
                 "n_window_blocks", "chol_n", "chol_tile", "chol_tiles_nnz", "chol_steps", "reduce_doubles")] + \
               [("chol_flops", C.c_double), ("lm_iterations", C.c_uint64), ("lm_trials", C.c_uint64),
                ("chol_failures", C.c_uint64), ("backsolve_timeouts", C.c_uint64)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_size_t, vp)

# name -> (restype, argtypes); every symbol include/svi_hot.h declares
SIGNATURES = {
    "svi_status_string": (C.c_char_p, [C.c_int]),
    "svi_last_error": (C.c_char_p, []),
    "svi_version": (C.c_int, []),
    "svi_device_count": (C.c_int, []),
    "svi_matcher_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
    "svi_matcher_destroy": (C.c_int, [vp]),
    "svi_matcher_sync": (C.c_int, [vp]),
    "svi_matcher_stream": (vp, [vp]),
    "svi_matcher_set_gate_path": (C.c_int, [vp, C.c_int]),
    "svi_debug_shader_clock_mhz": (C.c_int, [vp, f64p]),
    "svi_match_hamming256": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.POINTER(Gate), C.c_int, vp, vp]),
    "svi_match_hamming256_dev": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.POINTER(Gate), C.c_int, vp, vp]),
    "svi_match_clouds_dev": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "svi_hamming256_pairs": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "svi_hamming256_pairs_dev": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "svi_triangulate_rectified": (C.c_int, [vp] + [C.c_double] * 5 + [vp, vp, C.c_int, vp, vp]),
    "svi_triangulate_rectified_dev": (C.c_int, [vp] + [C.c_double] * 5 + [vp, vp, C.c_int, vp, vp]),
    "svi_match_triangulate_dev": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.POINTER(Gate), C.c_int] +
                                  [C.c_double] * 5 + [vp, vp, vp, vp]),
    "svi_track_plan_dev": (C.c_int, [vp, C.POINTER(TrackCamera), f64p, f64p, C.c_int, C.c_double, vp, vp, vp, vp, vp, C.c_int,
                                     vp, vp, i64p]),
    "svi_track_epipolar_samples_dev": (C.c_int, [vp, C.POINTER(TrackCamera), vp, vp, vp, C.c_int, vp, C.c_int, vp, vp]),
    "svi_track_handover_dev": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "svi_track_stereo_range_dev": (C.c_int, [vp, C.c_double, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, i64p]),
    "svi_track_stereo_candidates_dev": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp]),
    "svi_match_ragged_dev": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    "svi_track_stereo_verify_dev": (C.c_int, [vp, C.POINTER(TrackStereoParams), vp, vp, vp, vp, vp, C.c_int, vp, vp, vp,
                                              vp, vp, vp, vp, vp]),
    "svi_brief_create": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "svi_brief_destroy": (C.c_int, [vp]),
    "svi_brief_set_image_dev": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]),
    "svi_brief_integral_dev": (C.c_int, [vp, C.c_int, vp]),
    "svi_brief_compute_dev": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int64, vp, vp, vp, i64p]),
    "svi_posit_params_default": (None, [C.POINTER(PositParams)]),
    "svi_tracker_create": (C.c_int, [vp, C.POINTER(TrackCamera), C.POINTER(vp)]),
    "svi_tracker_destroy": (C.c_int, [vp]),
    "svi_tracker_set_brief": (C.c_int, [vp, vp]),
    "svi_tracker_set_extractor": (C.c_int, [vp, EXTRACT_FN, vp]),
    "svi_tracker_set_detector": (C.c_int, [vp, DETECT_FN, vp, C.c_int64]),
    "svi_tracker_plan": (C.c_int, [vp, f64p, f64p, C.c_int, C.c_double, C.POINTER(TrackLandmarks)]),
    "svi_tracker_set_descriptors": (C.c_int, [vp, vp, vp, vp]),
    "svi_tracker_records": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int64)]),
    "svi_track_stage1": (C.c_int, [vp, vp, C.POINTER(TrackResult)]),
    "svi_track_stage2": (C.c_int, [vp, vp, C.POINTER(TrackResult)]),
    "svi_track_epipolar": (C.c_int, [vp, vp, C.POINTER(TrackResult)]),
    "svi_track_manual": (C.c_int, [vp, vp, C.POINTER(TrackResult)]),
    "svi_track_pose_stereo_posit": (C.c_int, [vp, vp, C.POINTER(PositParams), f64p, f64p, f64p, C.POINTER(TrackResult), C.POINTER(PositResult)]),
    "svi_track_add_new_landmarks": (C.c_int, [vp, vp, vp, vp, C.c_int, C.POINTER(TrackResult)]),
    "svi_stereo_posit_dev": (C.c_int, [vp, C.POINTER(PositParams), f64p, f64p, f64p, vp, vp, vp, vp, C.c_int, C.POINTER(PositResult)]),
    "svi_landmark_params_default": (None, [C.POINTER(LandmarkParams)]),
    "svi_landmarks_optimize_dev": (C.c_int, [vp, C.POINTER(LandmarkParams), vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp]),
    "svi_ba_options_default": (None, [C.POINTER(BaOptions)]),
    "svi_ba_create": (C.c_int, [C.POINTER(BaOptions), C.POINTER(vp)]),
    "svi_ba_destroy": (C.c_int, [vp]),
    "svi_ba_add_pose": (C.c_int, [vp, C.c_int64, f64p, C.c_int]),
    "svi_ba_add_landmark": (C.c_int, [vp, C.c_int64, f64p, C.c_int]),
    "svi_ba_add_edge_xyz": (C.c_int, [vp, C.c_int64, C.c_int64, f64p, f64p, C.c_int]),
    "svi_ba_add_edge_depth": (C.c_int, [vp, C.c_int64, C.c_int64, f64p, f64p, C.c_int]),
    "svi_ba_add_edge_disparity": (C.c_int, [vp, C.c_int64, C.c_int64, f64p, f64p, C.c_int]),
    "svi_ba_add_edges_bulk": (C.c_int, [vp, C.c_int64, i32p, i64p, i64p, f64p, f64p, i32p]),
    "svi_ba_add_edge_se3": (C.c_int, [vp, C.c_int64, C.c_int64, f64p, f64p, C.c_int]),
    "svi_ba_add_edge_accel": (C.c_int, [vp, C.c_int64, f64p, f64p, f64p]),
    "svi_ba_add_edge_lm_lm": (C.c_int, [vp, C.c_int64, C.c_int64, f64p, f64p, C.c_int]),
    "svi_ba_add_keyframe": (C.c_int, [vp, C.c_int64, C.c_int64, f64p, f64p, f64p]),
    "svi_ba_set_imu_offset": (C.c_int, [vp, f64p]),
    "svi_ba_add_measurements": (C.c_int, [vp, C.c_int64, C.c_int64, i64p, f32p, f32p, f64p, i64p]),
    "svi_ba_initialize": (C.c_int, [vp]),
    "svi_ba_optimize": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int)]),
    "svi_ba_optimize_until": (C.c_int, [vp, C.c_double, C.c_int, C.c_int, u64p, u64p]),
    "svi_ba_chi2": (C.c_int, [vp, f64p, f64p]),
    "svi_ba_sync_host": (C.c_int, [vp]),
    "svi_ba_lambda": (C.c_int, [vp, f64p]),
    "svi_ba_get_pose": (C.c_int, [vp, C.c_int64, f64p]),
    "svi_ba_get_landmark": (C.c_int, [vp, C.c_int64, f64p]),
    "svi_ba_num_poses": (C.c_int, [vp, i64p]),
    "svi_ba_num_landmarks": (C.c_int, [vp, i64p]),
    "svi_ba_num_edges": (C.c_int, [vp, i64p]),
    "svi_ba_get_poses": (C.c_int, [vp, i64p, f64p]),
    "svi_ba_get_landmarks": (C.c_int, [vp, i64p, f64p]),
    "svi_ba_prune_diverged": (C.c_int, [vp, i64p]),
    "svi_ba_apply_optimization": (C.c_int, [vp, f64p, i64p, f64p, u8p, i64p, f64p, i64p]),
    "svi_ba_load_g2o": (C.c_int, [vp, C.c_char_p]),
    "svi_ba_save_g2o": (C.c_int, [vp, C.c_char_p]),
    "svi_ba_set_allreduce": (C.c_int, [vp, ALLREDUCE_FN, vp]),
    "svi_debug_set_backsolve_spin_limit": (C.c_int, [C.c_int]),
    "svi_rccl_available": (C.c_int, []),
    "svi_rccl_unique_id": (C.c_int, [vp]),
    "svi_rccl_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "svi_rccl_destroy": (C.c_int, [vp]),
    "svi_rccl_allreduce": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "svi_ba_get_phase_times": (C.c_int, [vp, f64p, i64p]),
    "svi_ba_reset_phase_times": (C.c_int, [vp]),
    "svi_ba_get_stats": (C.c_int, [vp, C.POINTER(BaStats)]),
    "svi_ba_debug_edge_jacobians": (C.c_int, [vp, f64p, f64p, f64p]),
    "svi_ba_debug_aux_jacobians": (C.c_int, [vp, f64p, f64p, f64p, f64p, f64p]),
    "svi_ba_debug_reduced_system": (C.c_int, [vp, C.c_double, f64p, f64p, C.c_int64, i64p]),
    "svi_debug_chol_probe": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, f64p]),
    "svi_ba_debug_time_sweep": (C.c_int, [vp, C.c_int, f64p]),
    "svi_ba_debug_time_sweep_part": (C.c_int, [vp, C.c_int, C.c_int, f64p]),
    "svi_ba_debug_time_sweep_cold": (C.c_int, [vp, C.c_int, C.c_size_t, f64p]),
    "svi_ba_get_sweep_time": (C.c_int, [vp, f64p, i64p]),
}

_lib = None
_hip = None


def _preload_hip_runtime():
    """libsvi_hot.so has no DT_NEEDED on the HIP runtime; bind it to the one this process uses.
    PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64, and two copies in one process cannot
    both drive the GPU, so torch's copy wins whenever torch is importable; otherwise $ROCM_PATH's."""
    global _hip
    if _hip is not None:
        return _hip
    cands = []
    try:
        import torch  # noqa: F401  (also makes sure torch's runtime is the first one loaded)
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:
        pass
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cands += [os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so"]
    err = None
    for c in cands:
        if os.path.sep in c and not os.path.exists(c):
            continue
        try:
            _hip = C.CDLL(c, mode=C.RTLD_GLOBAL)
            return _hip
        except OSError as e:
            err = e
    raise ImportError("svi_mapper_amd: no HIP runtime (libamdhip64) could be loaded: %s" % err)


def load_library(path=None):
    """Load libsvi_hot.so and bind every declared symbol. Raises ImportError if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError("svi_mapper_amd: %s not found - build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` or `make -C svi_mapper_amd/csrc` (hipcc, gfx950). There is no CPU fallback." % p)
    _preload_hip_runtime()
    lib = C.CDLL(p)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:  # header and library disagree: refuse to run on a partial ABI
        raise ImportError("svi_mapper_amd: %s lacks symbols declared in include/svi_hot.h: %s" % (p, ", ".join(missing)))
    if path is None:
        _lib = lib
    return lib


def _status_string(status):
    try:
        return load_library().svi_status_string(int(status)).decode()
    except Exception:
        return "status"


def check(status, where):
    if status != SVI_OK:
        raise SviError(status, where, load_library().svi_last_error().decode())
