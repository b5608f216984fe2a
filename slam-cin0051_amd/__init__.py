"""slam-cin0051_amd -- ctypes binding of libslamx.so, the MI355X-native frontend hot path.

Host-side mirror of the reference's operator interface for this path, in Python for the tests
and bench.py (the C++17 facade with the reference's class names lives in include/slam/):

    FeatureDetector(config).detect / compute / detectAndCompute
        <- slam::FeatureDetector, include/slam/frontend/feature_detector.hpp:47-192
    FeatureMatcher(config).match
        <- slam::FeatureMatcher, include/slam/frontend/feature_matcher.hpp:37-88

Errors follow the reference's exception types: std::invalid_argument -> ValueError,
std::runtime_error -> RuntimeError.  Everything computes on the GPU through the C ABI
(include/slamx.h); if libslamx.so is missing or no HIP device is usable the import / the
Context constructor raise -- there is no CPU fallback.

The directory name contains a hyphen, so load it with importlib (see __graft_entry__.py /
tests/conftest.py) under the module name ``slam_cin0051_amd``.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libslamx.so")

OK, ERR_INVALID_ARGUMENT, ERR_RUNTIME, ERR_HIP, ERR_CAPACITY, ERR_NO_DEVICE = range(6)
STAGES = ("fast", "scan", "expand", "sort", "nms_build", "nms_rounds", "blur", "moments", "trig", "brief", "match", "emit")

KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4")], align=False
)
MATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("distance", "<f4")], align=False)
assert KEYPOINT_DTYPE.itemsize == 20 and MATCH_DTYPE.itemsize == 12


class DetectorConfig(C.Structure):
    _fields_ = [
        ("intensity_threshold", C.c_int32),
        ("contiguous_pixels_threshold", C.c_int32),
        ("non_max_suppression", C.c_int32),
        ("suppression_window_size", C.c_int32),
        ("patch_size", C.c_int32),
        ("num_brief_pairs", C.c_int32),
    ]

    def __repr__(self):
        return "DetectorConfig(%s)" % ", ".join("%s=%d" % (f, getattr(self, f)) for f, _ in self._fields_)


class MatcherConfig(C.Structure):
    _fields_ = [
        ("distance_type", C.c_int32),
        ("filter_matches", C.c_int32),
        ("good_matches_count", C.c_int32),
        ("use_ratio_test", C.c_int32),
        ("ratio_test_threshold", C.c_float),
    ]


class LoopClosureConfig(C.Structure):
    _fields_ = [("min_db_size", C.c_int32), ("min_frames_difference", C.c_int32), ("min_absolute_score", C.c_double),
                ("relative_score_factor", C.c_double), ("min_matches_for_pnp", C.c_int32),
                ("min_inliers_for_pnp", C.c_int32), ("ransac_max_iterations", C.c_int32),
                ("ransac_reprojection_threshold", C.c_double)]


class Camera(C.Structure):
    """slam::Camera (common.hpp:67-190): intrinsics, distortion, image size."""
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("k1", C.c_double),
                ("k2", C.c_double), ("p1", C.c_double), ("p2", C.c_double), ("k3", C.c_double), ("width", C.c_int32),
                ("height", C.c_int32), ("K", C.c_double * 9)]


class PoseConfig(C.Structure):
    """slamx_pose_config (include/slamx.h): the pose stage of a batch."""
    _fields_ = [("K", C.c_double * 9), ("iterations", C.c_int32), ("threshold_px", C.c_double), ("seed", C.c_uint32),
                ("pnp_iterations", C.c_int32), ("pnp_threshold", C.c_double), ("pnp_min_correspondences", C.c_int32),
                ("max_matches", C.c_int32), ("geometry_mode", C.c_int32)]


GEOMETRY_REFERENCE, GEOMETRY_CALIBRATED = 0, 1  # SLAMX_PNP_* (include/slamx.h)
POSE_RESULT_DTYPE = np.dtype([("E", "<f8", (3, 3)), ("R", "<f8", (3, 3)), ("t", "<f8", (3,)), ("pnp_R", "<f8", (3, 3)),
                              ("pnp_t", "<f8", (3,)), ("n_matches", "<i4"), ("n_inliers", "<i4"), ("front", "<i4", (4,)),
                              ("pnp_correspondences", "<i4"), ("pnp_inliers", "<i4"), ("status", "<i4"),
                              ("best_hypothesis", "<i4")], align=True)
POSE_STAGES = ("gather", "hypotheses", "score", "select", "refit", "final", "cheirality", "triangulate", "pnp_gather",
               "pnp_solve", "pnp_score")
POSE_TOO_FEW_MATCHES, POSE_PNP_SKIPPED, POSE_MATCH_OVERFLOW, POSE_NO_TRAIN_FRAME = 1, 2, 4, 8


class BatchShape(C.Structure):
    _fields_ = [
        ("rows", C.c_int32),
        ("cols", C.c_int32),
        ("max_frames", C.c_int32),
        ("max_candidates", C.c_int32),
        ("max_keypoints", C.c_int32),
    ]


# every symbol include/slamx.h declares (tests check that the library exports all of them)
EXPORTS = [
    "slamx_last_error", "slamx_version", "slamx_matcher_mfma_threshold", "slamx_device_count", "slamx_ctx_create", "slamx_ctx_destroy", "slamx_ctx_synchronize",
    "slamx_ctx_stream", "slamx_ctx_device_name", "slamx_ctx_timer_start", "slamx_ctx_timer_stop_ms",
    "slamx_detector_context", "slamx_detector_config_load", "slamx_matcher_config_load", "slamx_loop_closure_config_load", "slamx_detector_create", "slamx_detector_destroy",
    "slamx_detector_pattern", "slamx_detector_descriptor_width", "slamx_detect", "slamx_compute",
    "slamx_detect_and_compute", "slamx_fast_candidates", "slamx_gaussian_blur", "slamx_debug_sort_desc", "slamx_debug_sort_desc_groups",
    "slamx_debug_trig", "slamx_debug_penalty", "slamx_matcher_create", "slamx_matcher_destroy", "slamx_matcher_match",
    "slamx_batch_create", "slamx_batch_destroy", "slamx_batch_frames", "slamx_batch_upload",
    "slamx_host_alloc", "slamx_host_free", "slamx_batch_upload_async", "slamx_batch_upload_async_flags",
    "slamx_batch_detect_and_compute_range", "slamx_batch_match_consecutive_range",
    "slamx_batch_detect_and_compute", "slamx_batch_match_consecutive", "slamx_batch_counts", "slamx_batch_download",
    "slamx_batch_download_matches", "slamx_matcher_filter_matches", "slamx_batch_download_image", "slamx_batch_download_results_async", "slamx_batch_packed_bound", "slamx_batch_download_packed_async",
    "slamx_batch_download_packed_tail", "slamx_batch_set_profiling", "slamx_batch_set_profiling_every", "slamx_batch_stage_ms", "slamx_synth_frame", "slamx_pnp_dlt_batch",
    "slamx_count_inliers_batch", "slamx_pnp_ransac", "slamx_essential_ransac", "slamx_recover_pose", "slamx_triangulate", "slamx_camera_load",
    "slamx_essential_from_samples", "slamx_pnp_from_samples", "slamx_pnp_from_samples_mode", "slamx_pose_create", "slamx_pose_destroy", "slamx_batch_pose",
    "slamx_pose_download", "slamx_pose_download_points", "slamx_pose_download_samples", "slamx_pose_download_correspondences", "slamx_pose_download_hypotheses", "slamx_pose_download_pnp_hypotheses",
    "slamx_pose_set_profiling", "slamx_pose_stage_ms", "slamx_batch_upload_results",
    "slamx_undistort",
    "slamx_pyramid_create", "slamx_pyramid_destroy", "slamx_pyramid_batch", "slamx_pyramid_level_shape",
    "slamx_pyramid_detect_and_compute",
    "slamx_gn_create", "slamx_gn_destroy", "slamx_gn_upload", "slamx_gn_iterate", "slamx_gn_download",
    "slamx_gn_normal_equations", "slamx_gn_set_profiling", "slamx_gn_stage_ms",
    "slamx_mapgather_unique_id", "slamx_mapgather_create", "slamx_mapgather_destroy", "slamx_mapgather_allgather", "slamx_mapgather_reserve", "slamx_mapgather_allgather_fixed",
    "slamx_mapgather_plan", "slamx_mapgather_unpack",
]

_lib = None


def load_library():
    """Load libslamx.so (built by `make -C slam-cin0051_amd` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libslamx.so is not built (%s): run `make -C slam-cin0051_amd` or __graft_entry__.build(). "
            "The slamx hot path has no CPU fallback." % LIB_PATH
        )
    lib = C.CDLL(LIB_PATH)
    lib.slamx_last_error.restype = C.c_char_p
    lib.slamx_version.restype = C.c_char_p
    lib.slamx_ctx_stream.restype = C.c_void_p
    lib.slamx_detector_context.restype = C.c_void_p
    lib.slamx_batch_frames.restype = C.c_void_p
    vp = C.c_void_p
    lib.slamx_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.slamx_ctx_destroy.argtypes = [vp]
    lib.slamx_ctx_synchronize.argtypes = [vp]
    lib.slamx_ctx_stream.argtypes = [vp]
    lib.slamx_detector_context.argtypes = [vp]
    lib.slamx_ctx_device_name.argtypes = [vp, C.c_char_p, C.c_int]
    lib.slamx_ctx_timer_start.argtypes = [vp]
    lib.slamx_ctx_timer_stop_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.slamx_detector_config_load.argtypes = [C.c_char_p, C.POINTER(DetectorConfig)]
    lib.slamx_matcher_config_load.argtypes = [C.c_char_p, C.POINTER(MatcherConfig)]
    lib.slamx_detector_create.argtypes = [vp, C.POINTER(DetectorConfig), C.POINTER(vp)]
    lib.slamx_detector_destroy.argtypes = [vp]
    lib.slamx_detector_pattern.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
    lib.slamx_detector_descriptor_width.argtypes = [vp]
    img = [vp, vp, C.c_int, C.c_int, C.c_ssize_t]
    lib.slamx_detect.argtypes = img + [vp, C.c_int, C.POINTER(C.c_int)]
    lib.slamx_compute.argtypes = img + [vp, C.c_int, vp]
    lib.slamx_detect_and_compute.argtypes = img + [vp, vp, C.c_int, C.POINTER(C.c_int)]
    lib.slamx_fast_candidates.argtypes = img + [vp, C.c_int, C.POINTER(C.c_int)]
    lib.slamx_gaussian_blur.argtypes = img + [vp]
    lib.slamx_debug_sort_desc.argtypes = [vp, vp, C.c_int, vp]
    lib.slamx_debug_sort_desc_groups.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    lib.slamx_debug_trig.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp]
    lib.slamx_debug_penalty.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp]
    lib.slamx_matcher_create.argtypes = [vp, C.POINTER(MatcherConfig), C.POINTER(vp)]
    lib.slamx_matcher_destroy.argtypes = [vp]
    lib.slamx_matcher_match.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, vp,
                                        C.c_int, C.POINTER(C.c_int)]
    lib.slamx_batch_create.argtypes = [vp, C.POINTER(BatchShape), C.POINTER(vp)]
    lib.slamx_batch_destroy.argtypes = [vp]
    lib.slamx_batch_frames.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.slamx_batch_upload.argtypes = [vp, C.c_int, C.c_int, vp, C.c_ssize_t, C.c_ssize_t]
    lib.slamx_batch_upload_async.argtypes = [vp, C.c_int, C.c_int, vp, C.c_ssize_t, C.c_ssize_t]
    lib.slamx_batch_upload_async_flags.argtypes = [vp, C.c_int, C.c_int, vp, C.c_ssize_t, C.c_ssize_t, C.c_int]
    lib.slamx_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    lib.slamx_host_free.argtypes = [vp]
    lib.slamx_host_free.restype = None
    lib.slamx_batch_detect_and_compute.argtypes = [vp, C.c_int]
    lib.slamx_batch_match_consecutive.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    lib.slamx_batch_detect_and_compute_range.argtypes = [vp, C.c_int, C.c_int]
    lib.slamx_batch_match_consecutive_range.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.slamx_batch_counts.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.slamx_batch_download.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int)]
    lib.slamx_batch_download_image.argtypes = [vp, C.c_int, C.c_int, vp, C.c_ssize_t]
    lib.slamx_batch_download_matches.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]
    lib.slamx_batch_download_results_async.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    lib.slamx_mapgather_unique_id.argtypes = [vp]
    lib.slamx_mapgather_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.slamx_mapgather_destroy.argtypes = [vp]
    lib.slamx_mapgather_allgather.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int)]
    lib.slamx_mapgather_reserve.argtypes = [vp, C.c_int]
    lib.slamx_mapgather_allgather_fixed.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int)]
    lib.slamx_mapgather_plan.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.slamx_mapgather_unpack.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int]
    lib.slamx_batch_packed_bound.argtypes = [vp, C.c_int]
    lib.slamx_batch_packed_bound.restype = C.c_size_t
    lib.slamx_batch_download_packed_async.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.slamx_batch_download_packed_tail.argtypes = [vp, vp, C.c_size_t, C.c_size_t]
    lib.slamx_batch_set_profiling.argtypes = [vp, C.c_int]
    lib.slamx_batch_set_profiling_every.argtypes = [vp, C.c_int, C.c_int]
    lib.slamx_batch_stage_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    lib.slamx_pnp_dlt_batch.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    lib.slamx_count_inliers_batch.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_double, vp]
    lib.slamx_pnp_ransac.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, C.c_double, C.c_uint32, vp, vp, C.POINTER(C.c_int32)]
    lib.slamx_essential_ransac.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, C.c_double, C.c_uint32, vp, vp, vp]
    lib.slamx_essential_from_samples.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_double, vp, vp, vp]
    lib.slamx_pnp_from_samples.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_double, vp, vp, C.POINTER(C.c_int32)]
    lib.slamx_pnp_from_samples_mode.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_double, C.c_int, vp, vp,
                                                C.POINTER(C.c_int32), vp, vp]
    lib.slamx_pose_create.argtypes = [vp, C.POINTER(PoseConfig), C.POINTER(vp)]
    lib.slamx_pose_destroy.argtypes = [vp]
    lib.slamx_pose_destroy.restype = None
    lib.slamx_batch_pose.argtypes = [vp, C.c_int, C.c_int]
    lib.slamx_pose_download.argtypes = [vp, C.c_int, vp]
    lib.slamx_pose_download_points.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int)]
    lib.slamx_pose_download_samples.argtypes = [vp, C.c_int, vp, vp]
    lib.slamx_pose_download_correspondences.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int)]
    lib.slamx_pose_download_hypotheses.argtypes = [vp, C.c_int, vp, vp]
    lib.slamx_pose_download_pnp_hypotheses.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.slamx_pose_set_profiling.argtypes = [vp, C.c_int]
    lib.slamx_pose_stage_ms.argtypes = [vp, vp, vp]
    lib.slamx_batch_upload_results.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int]
    lib.slamx_recover_pose.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, vp]
    lib.slamx_triangulate.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, vp]
    lib.slamx_camera_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(Camera)]
    lib.slamx_undistort.argtypes = [vp, C.POINTER(Camera), vp, C.c_int, C.c_int, C.c_ssize_t, vp]
    lib.slamx_pyramid_create.argtypes = [vp, C.POINTER(BatchShape), C.c_int, C.c_double, C.POINTER(vp)]
    lib.slamx_pyramid_destroy.argtypes = [vp]
    lib.slamx_pyramid_destroy.restype = None
    lib.slamx_pyramid_batch.argtypes = [vp, C.c_int]
    lib.slamx_pyramid_batch.restype = vp
    lib.slamx_pyramid_level_shape.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    lib.slamx_pyramid_detect_and_compute.argtypes = [vp, C.c_int]
    lib.slamx_gn_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.slamx_gn_destroy.argtypes = [vp]
    lib.slamx_gn_destroy.restype = None
    lib.slamx_gn_upload.argtypes = [vp, vp, vp, vp, vp]
    lib.slamx_gn_iterate.argtypes = [vp, vp, C.c_double, C.c_int, C.c_int]
    lib.slamx_gn_download.argtypes = [vp, vp, vp, vp, vp]
    lib.slamx_gn_set_profiling.argtypes = [vp, C.c_int]
    lib.slamx_gn_stage_ms.argtypes = [vp, vp, vp]
    lib.slamx_gn_normal_equations.argtypes = [vp, vp, C.c_double, C.c_int, vp, vp, vp, vp]
    lib.slamx_synth_frame.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    _lib = lib
    return lib


class CapacityError(RuntimeError):
    pass


def _check(rc):
    if rc == OK:
        return
    msg = load_library().slamx_last_error().decode("utf-8", "replace")
    if rc == ERR_INVALID_ARGUMENT:
        raise ValueError(msg)  # std::invalid_argument
    if rc == ERR_CAPACITY:
        raise CapacityError(msg)
    raise RuntimeError("slamx error %d: %s" % (rc, msg))  # std::runtime_error


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _as_image(image):
    img = np.asarray(image)
    if img.dtype != np.uint8 or img.ndim != 2:
        raise ValueError("image must be a 2-D uint8 array (EigenGrayMatrix)")
    if img.strides[1] != 1:
        img = np.ascontiguousarray(img)
    return img


def synth_frame(seed, k, rows, cols, n_rect, rect_min=16, rect_max=64):
    """Deterministic synthetic frame k; host-side generator inside libslamx."""
    out = np.empty((rows, cols), np.uint8)
    _check(load_library().slamx_synth_frame(seed, k, rows, cols, n_rect, rect_min, rect_max, _ptr(out)))
    return out


class _PinnedOwner:
    """Keeps a slamx_host_alloc block alive for as long as a numpy view of it exists."""

    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def __del__(self):
        try:
            self.lib.slamx_host_free(self.ptr)
        except Exception:
            pass


def pinned_array(shape, dtype=np.uint8):
    """numpy array in page-locked host memory (slamx_host_alloc): the source of Batch.upload_async."""
    lib = load_library()
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    _check(lib.slamx_host_alloc(n, C.byref(p)))
    owner = _PinnedOwner(lib, p)
    buf = (C.c_uint8 * n).from_address(p.value)
    buf._owner = owner  # the ctypes block (and through numpy's base chain every view) holds the owner
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def load_detector_config(path):
    cfg = DetectorConfig()
    _check(load_library().slamx_detector_config_load(os.fsencode(path), C.byref(cfg)))
    return cfg


def load_camera(path, camera_index=0):
    cam = Camera()
    _check(load_library().slamx_camera_load(os.fsencode(path), camera_index, C.byref(cam)))
    return cam


def load_matcher_config(path):
    cfg = MatcherConfig()
    _check(load_library().slamx_matcher_config_load(os.fsencode(path), C.byref(cfg)))
    return cfg


def load_loop_closure_config(path):
    cfg = LoopClosureConfig()
    _check(load_library().slamx_loop_closure_config_load(os.fsencode(path), C.byref(cfg)))
    return cfg


class _Handle:
    """A C handle and the handles that have to be destroyed BEFORE it (a batch before its detector, a detector
    before its context).  Plain cells with strong links both ways and no finaliser of their own: the wrapper
    objects' __del__ only call close() here.  (WeakSets of the wrapper objects did not do: when the cyclic GC
    collects a parent and a child together -- a failed test's traceback holds both -- it clears the weak
    references first, the parent no longer finds the child, frees its own handle, and the child's destroy call
    then runs on a dangling pointer.)"""

    __slots__ = ("destroy", "value", "parent", "children")

    def __init__(self, destroy, parent=None, value=None):
        self.destroy = destroy  # None: a borrowed handle, only forgotten
        self.value = C.c_void_p(value)
        self.parent = parent
        self.children = []
        if parent is not None:
            parent.children.append(self)

    def close(self):
        if self.value:
            for c in list(self.children):
                c.close()
            if self.destroy is not None:
                self.destroy(self.value)
            self.value = C.c_void_p()
        if self.parent is not None:
            if self in self.parent.children:
                self.parent.children.remove(self)
            self.parent = None


class _Owner:
    """Mixin of the wrapper classes: `h` is the live C handle (NULL once closed, by whoever closed it)."""

    _hd = None

    @property
    def h(self):
        return self._hd.value

    def close(self):
        if self._hd is not None:
            self._hd.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_live_contexts = weakref.WeakSet()


def _close_all_contexts():
    """Handles are released while the HIP runtime is still up: objects that survive until interpreter
    shutdown would otherwise call hipFree after the runtime's own teardown."""
    for ctx in list(_live_contexts):
        try:
            ctx.close()
        except Exception:
            pass


atexit.register(_close_all_contexts)


class Context(_Owner):
    """One per GPU: owns the HIP stream every call is enqueued on."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        self._hd = _Handle(self.lib.slamx_ctx_destroy)  # detectors / matchers hang below it: destroyed before the context
        _check(self.lib.slamx_ctx_create(device, stream, C.byref(self._hd.value)))
        _live_contexts.add(self)

    def synchronize(self):
        _check(self.lib.slamx_ctx_synchronize(self.h))

    @property
    def stream(self):
        return self.lib.slamx_ctx_stream(self.h)

    @property
    def device_name(self):
        buf = C.create_string_buffer(256)
        _check(self.lib.slamx_ctx_device_name(self.h, buf, 256))
        return buf.value.decode()

    def timer_start(self):
        _check(self.lib.slamx_ctx_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float()
        _check(self.lib.slamx_ctx_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    def debug_sort_desc(self, scores, groups=1):
        """The std::sort permutation of `scores` (descending); groups > 1: through the two-launch form of the sort."""
        s = np.ascontiguousarray(scores, dtype=np.uint32)
        perm = np.empty(s.shape[0], np.uint32)
        _check(self.lib.slamx_debug_sort_desc_groups(self.h, _ptr(s), s.shape[0], int(groups), _ptr(perm)))
        return perm

    # ---- small dense pose solves (rows a11-a14) ----
    def pnp_dlt_batch(self, p3d, p2d):
        """LoopClosure::solvePnP for a batch: p3d [H, n, 3], p2d [H, n, 2] -> R [H,3,3], t [H,3], null [H,12]."""
        a = np.ascontiguousarray(p3d, dtype=np.float64)
        b = np.ascontiguousarray(p2d, dtype=np.float64)
        H, n = a.shape[0], a.shape[1]
        R = np.empty((H, 3, 3))
        t = np.empty((H, 3))
        nv = np.empty((H, 12))
        _check(self.lib.slamx_pnp_dlt_batch(self.h, _ptr(a), _ptr(b), n, H, _ptr(R), _ptr(t), _ptr(nv)))
        return R, t, nv

    def count_inliers_batch(self, p3d, p2d, K, R, t, threshold):
        a = np.ascontiguousarray(p3d, dtype=np.float64)
        b = np.ascontiguousarray(p2d, dtype=np.float64)
        Kc = np.ascontiguousarray(K, dtype=np.float64)
        Rc = np.ascontiguousarray(R, dtype=np.float64)
        tc = np.ascontiguousarray(t, dtype=np.float64)
        H = Rc.shape[0]
        counts = np.zeros(H, np.int32)
        _check(self.lib.slamx_count_inliers_batch(self.h, _ptr(a), _ptr(b), a.shape[0], _ptr(Kc), _ptr(Rc), _ptr(tc), H,
                                                  float(threshold), _ptr(counts)))
        return counts

    def pnp_ransac(self, p3d, p2d, K, iterations, threshold, seed):
        a = np.ascontiguousarray(p3d, dtype=np.float64)
        b = np.ascontiguousarray(p2d, dtype=np.float64)
        Kc = np.ascontiguousarray(K, dtype=np.float64)
        R = np.empty((3, 3))
        t = np.empty(3)
        best = C.c_int32()
        _check(self.lib.slamx_pnp_ransac(self.h, _ptr(a), _ptr(b), a.shape[0], _ptr(Kc), iterations, float(threshold),
                                         seed, _ptr(R), _ptr(t), C.byref(best)))
        return R, t, best.value

    def essential_ransac(self, pts1, pts2, K, iterations=512, threshold_px=1.0, seed=0):
        """E (3x3), inlier count, inlier mask from n x 2 float pixel correspondences."""
        a = np.ascontiguousarray(pts1, dtype=np.float32)
        b = np.ascontiguousarray(pts2, dtype=np.float32)
        Kc = np.ascontiguousarray(K, dtype=np.float64)
        E = np.empty((3, 3))
        cnt = C.c_int32()
        mask = np.zeros(a.shape[0], np.uint8)
        _check(self.lib.slamx_essential_ransac(self.h, _ptr(a), _ptr(b), a.shape[0], _ptr(Kc), iterations,
                                               float(threshold_px), seed, _ptr(E), C.byref(cnt), _ptr(mask)))
        return E, cnt.value, mask

    def essential_from_samples(self, pts1, pts2, K, table, threshold_px=1.0):
        """slamx_essential_from_samples: the RANSAC loop on a GIVEN sample table (iterations x 8 indices)."""
        a = np.ascontiguousarray(pts1, dtype=np.float32)
        b = np.ascontiguousarray(pts2, dtype=np.float32)
        Kc = np.ascontiguousarray(K, dtype=np.float64)
        tb = np.ascontiguousarray(table, dtype=np.int32)
        E = np.empty((3, 3))
        cnt = C.c_int32()
        mask = np.zeros(a.shape[0], np.uint8)
        _check(self.lib.slamx_essential_from_samples(self.h, _ptr(a), _ptr(b), a.shape[0], _ptr(Kc), _ptr(tb), tb.shape[0],
                                                     float(threshold_px), _ptr(E), C.byref(cnt), _ptr(mask)))
        return E, cnt.value, mask

    def pnp_from_samples(self, p3d, p2d, K, table, threshold, geometry_mode=None, all_hypotheses=False):
        """-> (R, t, max_inliers); with all_hypotheses also (R_all [H, 3, 3], t_all [H, 3], counts [H]).  geometry_mode None =
        the reference-signature call, GEOMETRY_REFERENCE / GEOMETRY_CALIBRATED = slamx_pnp_from_samples_mode."""
        a = np.ascontiguousarray(p3d, dtype=np.float64)
        b = np.ascontiguousarray(p2d, dtype=np.float64)
        Kc = np.ascontiguousarray(K, dtype=np.float64)
        tb = np.ascontiguousarray(table, dtype=np.int32)
        R = np.empty((3, 3))
        t = np.empty(3)
        best = C.c_int32()
        if geometry_mode is None and not all_hypotheses:
            _check(self.lib.slamx_pnp_from_samples(self.h, _ptr(a), _ptr(b), a.shape[0], _ptr(Kc), _ptr(tb), tb.shape[0],
                                                   float(threshold), _ptr(R), _ptr(t), C.byref(best)))
            return R, t, best.value
        H = tb.shape[0]
        Rt, cnt = np.empty((H, 12)), np.zeros(H, np.int32)
        _check(self.lib.slamx_pnp_from_samples_mode(self.h, _ptr(a), _ptr(b), a.shape[0], _ptr(Kc), _ptr(tb), H, float(threshold),
                                                    int(geometry_mode or 0), _ptr(R), _ptr(t), C.byref(best), _ptr(Rt), _ptr(cnt)))
        if not all_hypotheses:
            return R, t, best.value
        return R, t, best.value, Rt[:, :9].reshape(H, 3, 3).copy(), Rt[:, 9:].copy(), cnt

    def recover_pose(self, E, pts1, pts2, K):
        Ec = np.ascontiguousarray(E, dtype=np.float64)
        Kc = np.ascontiguousarray(K, dtype=np.float64)
        a = np.ascontiguousarray(pts1, dtype=np.float32)
        b = np.ascontiguousarray(pts2, dtype=np.float32)
        R = np.empty((3, 3))
        t = np.empty(3)
        front = np.zeros(4, np.int32)
        _check(self.lib.slamx_recover_pose(self.h, _ptr(Ec), _ptr(a), _ptr(b), a.shape[0], _ptr(Kc), _ptr(R), _ptr(t),
                                           _ptr(front)))
        return R, t, front

    def triangulate(self, P1, P2, pts1, pts2):
        A = np.ascontiguousarray(P1, dtype=np.float64)
        B = np.ascontiguousarray(P2, dtype=np.float64)
        a = np.ascontiguousarray(pts1, dtype=np.float32)
        b = np.ascontiguousarray(pts2, dtype=np.float32)
        n = a.shape[0]
        x4 = np.empty((n, 4), np.float32)
        xyz = np.empty((n, 3))
        _check(self.lib.slamx_triangulate(self.h, _ptr(A), _ptr(B), _ptr(a), _ptr(b), n, _ptr(x4), _ptr(xyz)))
        return x4, xyz

    def undistort(self, cam, image):
        """Camera::undistortImage: uint8 image -> [0,1] doubles, row-major."""
        img = _as_image(image)
        out = np.empty(img.shape, np.float64)
        _check(self.lib.slamx_undistort(self.h, C.byref(cam), _ptr(img), img.shape[0], img.shape[1], img.strides[0],
                                        _ptr(out)))
        return out

    def debug_penalty(self, dist, qx, qy, tx, ty):
        d = np.ascontiguousarray(dist, dtype=np.int32)
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (qx, qy, tx, ty)]
        out = np.empty_like(d)
        _check(self.lib.slamx_debug_penalty(self.h, _ptr(d), *[_ptr(a) for a in arrs], d.shape[0], _ptr(out)))
        return out

    def debug_trig(self, m01, m10):
        a = np.ascontiguousarray(m01, dtype=np.float32)
        b = np.ascontiguousarray(m10, dtype=np.float32)
        deg, c, s = (np.empty_like(a) for _ in range(3))
        _check(self.lib.slamx_debug_trig(self.h, _ptr(a), _ptr(b), a.shape[0], _ptr(deg), _ptr(c), _ptr(s)))
        return deg, c, s


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class FeatureDetector(_Owner):
    """slam::FeatureDetector (feature_detector.hpp:47-192).  config: YAML path or DetectorConfig."""

    def __init__(self, config, ctx=None):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.cfg = config if isinstance(config, DetectorConfig) else load_detector_config(config)
        self._hd = _Handle(self.lib.slamx_detector_destroy, self.ctx._hd)  # batches hang below it
        _check(self.lib.slamx_detector_create(self.ctx.h, C.byref(self.cfg), C.byref(self._hd.value)))
        self.width = self.lib.slamx_detector_descriptor_width(self.h)

    def pattern(self):
        n = C.c_int()
        _check(self.lib.slamx_detector_pattern(self.h, None, 0, C.byref(n)))
        xy = np.empty((n.value, 4), np.int32)
        _check(self.lib.slamx_detector_pattern(self.h, _ptr(xy), n.value, C.byref(n)))
        return xy

    def _sized(self, fn, img, *extra_out):
        cap = max(1024, img.shape[0] * img.shape[1] // 16)
        while True:
            kps = np.empty(cap, KEYPOINT_DTYPE)
            n = C.c_int()
            try:
                rc = fn(kps, cap, n)
                _check(rc)
                return kps, n.value
            except CapacityError:
                if n.value <= cap:
                    raise
                cap = n.value

    def detect(self, image):
        img = _as_image(image)

        def call(kps, cap, n):
            return self.lib.slamx_detect(self.h, _ptr(img), img.shape[0], img.shape[1], img.strides[0], _ptr(kps), cap,
                                         C.byref(n))

        kps, n = self._sized(call, img)
        return kps[:n].copy()

    def compute(self, image, keypoints):
        img = _as_image(image)
        kps = np.ascontiguousarray(keypoints, dtype=KEYPOINT_DTYPE).copy()
        n = kps.shape[0]
        if n == 0:
            return kps, np.zeros((0, 0), np.uint8)  # DescriptorMatrix(0, 0), feature_detector.cpp:22-25
        desc = np.empty((n, self.width), np.uint8)
        _check(self.lib.slamx_compute(self.h, _ptr(img), img.shape[0], img.shape[1], img.strides[0], _ptr(kps), n,
                                      _ptr(desc)))
        return kps, desc

    def detectAndCompute(self, image):
        img = _as_image(image)
        cap = max(1024, img.shape[0] * img.shape[1] // 16)
        while True:
            kps = np.empty(cap, KEYPOINT_DTYPE)
            desc = np.empty((cap, self.width), np.uint8)
            n = C.c_int()
            try:
                _check(self.lib.slamx_detect_and_compute(self.h, _ptr(img), img.shape[0], img.shape[1],
                                                         img.strides[0], _ptr(kps), _ptr(desc), cap, C.byref(n)))
                break
            except CapacityError:
                if n.value <= cap:
                    raise
                cap = n.value
        if n.value == 0:
            return kps[:0].copy(), np.zeros((0, 0), np.uint8)
        return kps[: n.value].copy(), desc[: n.value].copy()

    # intermediate products for the parity tests
    def fast_candidates(self, image):
        img = _as_image(image)

        def call(kps, cap, n):
            return self.lib.slamx_fast_candidates(self.h, _ptr(img), img.shape[0], img.shape[1], img.strides[0],
                                                  _ptr(kps), cap, C.byref(n))

        kps, n = self._sized(call, img)
        return kps[:n].copy()

    def gaussian_blur(self, image):
        img = _as_image(image)
        out = np.empty(img.shape, np.uint8)
        _check(self.lib.slamx_gaussian_blur(self.h, _ptr(img), img.shape[0], img.shape[1], img.strides[0], _ptr(out)))
        return out


class FeatureMatcher(_Owner):
    """slam::FeatureMatcher (feature_matcher.hpp:37-88).  config: YAML path or MatcherConfig."""

    def __init__(self, config, ctx=None):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.cfg = config if isinstance(config, MatcherConfig) else load_matcher_config(config)
        self._hd = _Handle(self.lib.slamx_matcher_destroy, self.ctx._hd)
        _check(self.lib.slamx_matcher_create(self.ctx.h, C.byref(self.cfg), C.byref(self._hd.value)))

    def match(self, descriptors1, descriptors2, keypoints1=None, keypoints2=None):
        d1 = np.ascontiguousarray(descriptors1, dtype=np.uint8)
        d2 = np.ascontiguousarray(descriptors2, dtype=np.uint8)
        n1 = d1.shape[0] if d1.ndim == 2 else 0
        n2 = d2.shape[0] if d2.ndim == 2 else 0
        w1 = d1.shape[1] if d1.ndim == 2 else 0
        w2 = d2.shape[1] if d2.ndim == 2 else 0
        k1 = np.ascontiguousarray(keypoints1, dtype=KEYPOINT_DTYPE) if keypoints1 is not None else None
        k2 = np.ascontiguousarray(keypoints2, dtype=KEYPOINT_DTYPE) if keypoints2 is not None else None
        out = np.empty(max(n1, 1), MATCH_DTYPE)
        n = C.c_int()
        _check(self.lib.slamx_matcher_match(self.h, _ptr(d1), n1, w1, _ptr(d2), n2, w2, _ptr(k1),
                                            0 if k1 is None else k1.shape[0], _ptr(k2),
                                            0 if k2 is None else k2.shape[0], _ptr(out), out.shape[0], C.byref(n)))
        return out[: n.value].copy()


class Batch(_Owner):
    """Device-resident frame batch: frames stay in HBM between the stages."""

    def __init__(self, detector, rows, cols, max_frames, max_candidates=0, max_keypoints=0):
        self.det = detector
        self.lib = detector.lib
        self.shape = BatchShape(rows, cols, max_frames, max_candidates, max_keypoints)
        self._hd = _Handle(self.lib.slamx_batch_destroy, detector._hd)
        _check(self.lib.slamx_batch_create(detector.h, C.byref(self.shape), C.byref(self._hd.value)))
        self.rows, self.cols, self.max_frames = rows, cols, max_frames

    def frames_device(self):
        pitch, fstride = C.c_size_t(), C.c_size_t()
        p = self.lib.slamx_batch_frames(self.h, C.byref(pitch), C.byref(fstride))
        return p, pitch.value, fstride.value

    def upload(self, frames, first=0):
        f = np.ascontiguousarray(frames, dtype=np.uint8)
        if f.ndim == 2:
            f = f[None]
        _check(self.lib.slamx_batch_upload(self.h, first, f.shape[0], _ptr(f), f.strides[1], f.strides[0]))

    def upload_async(self, frames, first=0, slots_idle=False):
        """Queue the copy and return; `frames` must be (a view of) a pinned_array() and stay untouched
        until the context has been synchronised.  slots_idle: SLAMX_UPLOAD_SLOTS_IDLE (nothing queued earlier touches
        the target slots: the copy runs beside that work instead of behind it)."""
        f = frames if frames.ndim == 3 else frames[None]
        if f.dtype != np.uint8 or f.strides[2] != 1:
            raise ValueError("upload_async: uint8 frames with unit pixel stride")
        _check(self.lib.slamx_batch_upload_async_flags(self.h, first, f.shape[0], f.ctypes.data, f.strides[1], f.strides[0],
                                                       1 if slots_idle else 0))

    def detect_and_compute(self, nframes, first=0):
        """Frames [first, first + nframes)."""
        _check(self.lib.slamx_batch_detect_and_compute_range(self.h, first, nframes))

    def match_consecutive(self, matcher, nframes, use_keypoints=True, first_train_frame=-1, first=0):
        """Frames [first, first + nframes), each against its predecessor (frame 0 against first_train_frame, or not at all)."""
        _check(self.lib.slamx_batch_match_consecutive_range(self.h, matcher.h, first, nframes, 1 if use_keypoints else 0,
                                                            first_train_frame))

    def counts(self, nframes):
        a, b, c = (np.zeros(nframes, np.int32) for _ in range(3))
        _check(self.lib.slamx_batch_counts(self.h, nframes, _ptr(a), _ptr(b), _ptr(c)))
        return a, b, c

    def download(self, frame):
        n = C.c_int()
        try:
            _check(self.lib.slamx_batch_download(self.h, frame, None, None, 0, C.byref(n)))
        except CapacityError:
            if n.value == 0:
                raise
        kps = np.empty(n.value, KEYPOINT_DTYPE)
        desc = np.empty((n.value, self.det.width), np.uint8)
        _check(self.lib.slamx_batch_download(self.h, frame, _ptr(kps), _ptr(desc), n.value, C.byref(n)))
        return kps, desc

    def download_image(self, frame, blurred=False):
        out = np.empty((self.rows, self.cols), np.uint8)
        _check(self.lib.slamx_batch_download_image(self.h, frame, 1 if blurred else 0, _ptr(out), out.strides[0]))
        return out

    def download_matches(self, matcher, frame, apply_filter=False):
        n = C.c_int()
        try:
            _check(self.lib.slamx_batch_download_matches(self.h, matcher.h, frame, 1 if apply_filter else 0, None, 0,
                                                         C.byref(n)))
        except CapacityError:
            pass
        out = np.empty(max(n.value, 1), MATCH_DTYPE)
        _check(self.lib.slamx_batch_download_matches(self.h, matcher.h, frame, 1 if apply_filter else 0, _ptr(out),
                                                     out.shape[0], C.byref(n)))
        return out[: n.value].copy()

    def result_buffers(self, nframes, max_keypoints):
        """Page-locked host arrays for download_results_async: (kps, desc, matches, n_keypoints, n_matches)."""
        return (pinned_array((nframes, max_keypoints), KEYPOINT_DTYPE), pinned_array((nframes, max_keypoints, self.det.width)),
                pinned_array((nframes, max_keypoints), MATCH_DTYPE), pinned_array((nframes,), np.int32),
                pinned_array((nframes,), np.int32))

    def download_results_async(self, nframes, bufs):
        """Queue the copy of all results of frames [0, nframes) into result_buffers(); synchronise before reading."""
        kps, desc, matches, nk, nm = bufs
        _check(self.lib.slamx_batch_download_results_async(self.h, nframes, kps.ctypes.data, desc.ctypes.data,
                                                           matches.ctypes.data, nk.ctypes.data, nm.ctypes.data))

    def packed_buffer(self, nframes):
        """Page-locked byte array large enough for the packed results of nframes frames in the worst case."""
        return pinned_array((int(self.lib.slamx_batch_packed_bound(self.h, nframes)),), np.uint8)

    def download_packed_async(self, nframes, buf, nbytes=None):
        """Pack the valid results of frames [0, nframes) on the device and queue the copy of the first nbytes
        (default: all of buf) into the page-locked buf; synchronise, then unpack_results(buf) -- or fetch_packed_tail
        first when header.total_bytes > nbytes."""
        n = buf.nbytes if nbytes is None else min(int(nbytes), buf.nbytes)
        _check(self.lib.slamx_batch_download_packed_async(self.h, nframes, buf.ctypes.data, n))
        return n

    def fetch_packed_tail(self, buf, have):
        _check(self.lib.slamx_batch_download_packed_tail(self.h, buf.ctypes.data, int(have), buf.nbytes))

    def upload_results(self, frame, keypoints, matches):
        """slamx_batch_upload_results: keypoints and matches (queryIdx ascending) of one frame slot from the host."""
        k = np.ascontiguousarray(keypoints, dtype=KEYPOINT_DTYPE)
        m = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
        _check(self.lib.slamx_batch_upload_results(self.h, frame, _ptr(k), k.shape[0], _ptr(m), m.shape[0]))

    def set_profiling(self, n_sets=1, every=1):
        _check(self.lib.slamx_batch_set_profiling_every(self.h, n_sets, every))

    def stage_ms(self, set_index=0):
        ms = (C.c_float * len(STAGES))()
        _check(self.lib.slamx_batch_stage_ms(self.h, set_index, ms))
        return dict(zip(STAGES, [float(x) for x in ms]))


class PoseStage(_Owner):
    """slamx_pose_* (include/slamx.h): essential-matrix RANSAC -> simpleRecoverPose -> triangulatePoints -> PnP RANSAC for
    every frame pair of a batch, device-resident, a fixed number of launches per call."""

    def __init__(self, batch, K, iterations=256, threshold_px=1.0, seed=0, pnp_iterations=0, pnp_threshold=2.0,
                 pnp_min_correspondences=6, max_matches=0, geometry_mode=GEOMETRY_REFERENCE):
        self.batch = batch
        self.lib = batch.lib
        cfg = PoseConfig()
        Kc = np.ascontiguousarray(K, np.float64).reshape(9)
        for i in range(9):
            cfg.K[i] = float(Kc[i])
        cfg.iterations, cfg.threshold_px, cfg.seed = int(iterations), float(threshold_px), int(seed)
        cfg.pnp_iterations, cfg.pnp_threshold = int(pnp_iterations), float(pnp_threshold)
        cfg.pnp_min_correspondences, cfg.max_matches = int(pnp_min_correspondences), int(max_matches)
        cfg.geometry_mode = int(geometry_mode)
        self.cfg = cfg
        self._hd = _Handle(self.lib.slamx_pose_destroy, batch._hd)
        _check(self.lib.slamx_pose_create(batch.h, C.byref(cfg), C.byref(self._hd.value)))

    def run(self, nframes, first_train_frame=-1):
        _check(self.lib.slamx_batch_pose(self.h, nframes, first_train_frame))

    def download(self, nframes):
        out = np.zeros(nframes, POSE_RESULT_DTYPE)
        _check(self.lib.slamx_pose_download(self.h, nframes, _ptr(out)))
        return out

    def points(self, frame):
        """-> (xyz [n, 3], inlier mask [n]) of one frame's pair, parallel to its match list"""
        n = C.c_int()
        try:
            _check(self.lib.slamx_pose_download_points(self.h, frame, None, None, 0, C.byref(n)))
        except CapacityError:
            pass
        xyz = np.empty((n.value, 3))
        mask = np.zeros(n.value, np.uint8)
        _check(self.lib.slamx_pose_download_points(self.h, frame, _ptr(xyz), _ptr(mask), n.value, C.byref(n)))
        return xyz, mask

    def samples(self, frame):
        """-> (idx [iterations, 8], pnp_idx [pnp_iterations, 6]) as the device drew them"""
        idx = np.zeros((self.cfg.iterations, 8), np.int32)
        pidx = np.zeros((max(self.cfg.pnp_iterations, 0), 6), np.int32)
        _check(self.lib.slamx_pose_download_samples(self.h, frame, _ptr(idx), _ptr(pidx) if pidx.size else None))
        return idx, pidx

    def hypotheses(self, frame):
        """-> (E_all [iterations, 3, 3], counts [iterations]) of one frame's pair"""
        E = np.empty((self.cfg.iterations, 3, 3))
        c = np.zeros(self.cfg.iterations, np.int32)
        _check(self.lib.slamx_pose_download_hypotheses(self.h, frame, _ptr(E), _ptr(c)))
        return E, c

    def pnp_hypotheses(self, frame):
        """-> (R [H, 3, 3], t [H, 3], null vector [H, 12], counts [H]) of one frame's PnP hypotheses"""
        H = self.cfg.pnp_iterations
        Rt, nv, c = np.empty((H, 12)), np.empty((H, 12)), np.zeros(H, np.int32)
        _check(self.lib.slamx_pose_download_pnp_hypotheses(self.h, frame, _ptr(Rt), _ptr(nv), _ptr(c)))
        return Rt[:, :9].reshape(H, 3, 3).copy(), Rt[:, 9:].copy(), nv, c

    def correspondences(self, frame):
        n = C.c_int()
        try:
            _check(self.lib.slamx_pose_download_correspondences(self.h, frame, None, None, 0, C.byref(n)))
        except CapacityError:
            pass
        p3, p2 = np.empty((n.value, 3)), np.empty((n.value, 2))
        _check(self.lib.slamx_pose_download_correspondences(self.h, frame, _ptr(p3), _ptr(p2), n.value, C.byref(n)))
        return p3, p2

    def set_profiling(self, every=1):
        _check(self.lib.slamx_pose_set_profiling(self.h, int(every)))

    def stage_ms(self):
        ms = np.zeros(len(POSE_STAGES), np.float32)
        n = np.zeros(len(POSE_STAGES), np.int32)
        _check(self.lib.slamx_pose_stage_ms(self.h, _ptr(ms), _ptr(n)))
        return dict(zip(POSE_STAGES, [float(x) for x in ms])), int(n[0])


PACKED_HEADER_DTYPE = np.dtype([("nframes", "<i4"), ("width", "<i4"), ("total_keypoints", "<i4"), ("total_matches", "<i4"),
                                ("first_off", "<i8"), ("mfirst_off", "<i8"), ("status_off", "<i8"), ("kps_off", "<i8"),
                                ("desc_off", "<i8"), ("matches_off", "<i8"), ("total_bytes", "<i8")])


def packed_header(buf):
    """slamx_packed_header at the start of a packed results buffer (include/slamx.h)."""
    return np.frombuffer(buf, PACKED_HEADER_DTYPE, 1)[0]


def unpack_results(buf):
    """Views into a packed results buffer: (kp_first, match_first, status, kps, desc, matches); frame f owns
    kps[kp_first[f]:kp_first[f + 1]] (desc alike) and matches[match_first[f]:match_first[f + 1]]."""
    h = packed_header(buf)
    if int(h["total_bytes"]) > buf.nbytes:
        raise CapacityError("packed results need %d bytes, the buffer holds %d" % (int(h["total_bytes"]), buf.nbytes))
    F, w, nk, nm = int(h["nframes"]), int(h["width"]), int(h["total_keypoints"]), int(h["total_matches"])
    raw = buf.view(np.uint8).reshape(-1)
    kp_first = np.frombuffer(raw, np.int32, F + 1, int(h["first_off"]))
    m_first = np.frombuffer(raw, np.int32, F + 1, int(h["mfirst_off"]))
    status = np.frombuffer(raw, np.int32, F, int(h["status_off"]))
    kps = np.frombuffer(raw, KEYPOINT_DTYPE, nk, int(h["kps_off"]))
    desc = np.frombuffer(raw, np.uint8, nk * w, int(h["desc_off"])).reshape(nk, w)
    matches = np.frombuffer(raw, MATCH_DTYPE, nm, int(h["matches_off"]))
    return kp_first, m_first, status, kps, desc, matches


class _BorrowedBatch(Batch):
    """A level of a Pyramid: the handle belongs to the pyramid, so close() only forgets it."""

    def __init__(self, detector, handle, rows, cols, max_frames, owner):
        self.det = detector
        self.lib = detector.lib
        self._hd = _Handle(None, owner, handle)  # forgotten (not destroyed) when the pyramid goes
        self.rows, self.cols, self.max_frames = rows, cols, max_frames


class Pyramid(_Owner):
    """slamx_pyramid_* (include/slamx.h): n_levels batches, level l = cvRound(size / scale_factor ** l).
    Frames go into level(0) (upload), merged results come out of it (counts / download /
    match_consecutive / download_matches).  Build-side extension: the reference has no pyramid."""

    def __init__(self, detector, rows, cols, max_frames, n_levels=8, scale_factor=1.2, max_candidates=0, max_keypoints=0):
        self.det = detector
        self.lib = detector.lib
        shape = BatchShape(rows, cols, max_frames, max_candidates, max_keypoints)
        self._hd = _Handle(self.lib.slamx_pyramid_destroy, detector._hd)
        _check(self.lib.slamx_pyramid_create(detector.h, C.byref(shape), n_levels, float(scale_factor), C.byref(self._hd.value)))
        self.n_levels, self.max_frames = n_levels, max_frames
        self._levels = []
        for l in range(n_levels):
            r, c, s = C.c_int(), C.c_int(), C.c_double()
            _check(self.lib.slamx_pyramid_level_shape(self.h, l, C.byref(r), C.byref(c), C.byref(s)))
            self._levels.append(_BorrowedBatch(detector, self.lib.slamx_pyramid_batch(self.h, l), r.value, c.value, max_frames,
                                               self._hd))
        self.shapes = [(b.rows, b.cols) for b in self._levels]

    def level(self, l):
        return self._levels[l]

    def detect_and_compute(self, nframes):
        _check(self.lib.slamx_pyramid_detect_and_compute(self.h, nframes))


def mapgather_plan(counts):
    """slamx_mapgather_plan: (padded per-rank capacity, total) for a list of per-rank point counts."""
    c = np.ascontiguousarray(counts, np.int32)
    cap, total = C.c_int(), C.c_int()
    _check(load_library().slamx_mapgather_plan(_ptr(c), c.shape[0], C.byref(cap), C.byref(total)))
    return cap.value, total.value


def mapgather_unpack(padded, counts, cap_points=None):
    """slamx_mapgather_unpack: [world][padded_cap][3] -> the ranks' points back to back."""
    p = np.ascontiguousarray(padded, np.float64)
    c = np.ascontiguousarray(counts, np.int32)
    total = int(c.sum())
    out = np.empty((total if cap_points is None else cap_points, 3))
    _check(load_library().slamx_mapgather_unpack(_ptr(p), _ptr(c), c.shape[0], p.shape[1], _ptr(out), out.shape[0]))
    return out[:total]


class MapGather(_Owner):
    """slamx_mapgather_* (include/slamx.h): the RCCL all-gather of map points a C++ multi-GPU host uses.  unique_id()
    on rank 0, the 128 bytes handed to every rank, then MapGather(ctx, id, rank, world) on all of them."""

    @staticmethod
    def unique_id():
        buf = np.zeros(128, np.uint8)
        _check(load_library().slamx_mapgather_unique_id(_ptr(buf)))
        return buf

    def __init__(self, ctx, uid, rank, world):
        self.ctx = ctx
        self.lib = ctx.lib
        self.rank, self.world = rank, world
        u = np.ascontiguousarray(uid, np.uint8)
        self._hd = _Handle(self.lib.slamx_mapgather_destroy, ctx._hd)
        _check(self.lib.slamx_mapgather_create(ctx.h, _ptr(u), rank, world, C.byref(self._hd.value)))

    def allgather(self, xyz, cap_points=1 << 20):
        p = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        counts = np.zeros(self.world, np.int32)
        out = np.empty((cap_points, 3))
        total = C.c_int()
        _check(self.lib.slamx_mapgather_allgather(self.h, _ptr(p), p.shape[0], _ptr(counts), _ptr(out), cap_points,
                                                  C.byref(total)))
        return counts, out[: total.value].copy()

    def reserve(self, per_rank_cap):
        _check(self.lib.slamx_mapgather_reserve(self.h, int(per_rank_cap)))

    def allgather_fixed(self, xyz, cap_points=1 << 20):
        """The exchange as ONE collective (after reserve(per_rank_cap) on every rank): count and status travel in the payload."""
        p = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        counts = np.zeros(self.world, np.int32)
        out = np.empty((cap_points, 3))
        total = C.c_int()
        _check(self.lib.slamx_mapgather_allgather_fixed(self.h, _ptr(p), p.shape[0], _ptr(counts), _ptr(out), cap_points,
                                                        C.byref(total)))
        return counts, out[: total.value].copy()


class GaussNewton(_Owner):
    """A batch of sliding windows for slamx_gn_* (include/slamx.h): n_cams poses (world -> camera, R
    row-major + t), up to max_landmarks landmarks and a dense track table per window.  Build-defined
    (BASELINE.json configs[3]); the reference's backend is declaration-only."""

    def __init__(self, n_windows, n_cams, max_landmarks, ctx=None):
        self.ctx = ctx or default_context()
        self.lib = self.ctx.lib
        self.n_windows, self.n_cams, self.max_landmarks = n_windows, n_cams, max_landmarks
        self._hd = _Handle(self.lib.slamx_gn_destroy, self.ctx._hd)  # destroyed before the context (its stream is synchronised on destroy)
        _check(self.lib.slamx_gn_create(self.ctx.h, n_windows, n_cams, max_landmarks, C.byref(self._hd.value)))
        self.iterations = -1

    def upload(self, poses, landmarks, obs, n_landmarks=None):
        W, L, n = self.n_cams, self.max_landmarks, self.n_windows
        poses = np.ascontiguousarray(poses, np.float64).reshape(n, W, 12)
        landmarks = np.ascontiguousarray(landmarks, np.float64).reshape(n, L, 3)
        obs = np.ascontiguousarray(obs, np.float32).reshape(n, L, W, 2)
        nl = None if n_landmarks is None else np.ascontiguousarray(n_landmarks, np.int32).reshape(n)
        _check(self.lib.slamx_gn_upload(self.h, _ptr(poses), _ptr(landmarks), _ptr(obs), _ptr(nl)))
        self.iterations = -1

    def iterate(self, K, lam, n_fixed, iterations):
        k4 = np.ascontiguousarray(K, np.float64).reshape(4)
        _check(self.lib.slamx_gn_iterate(self.h, _ptr(k4), float(lam), n_fixed, iterations))
        self.iterations = iterations

    def download(self):
        """-> poses [n][W][12], landmarks [n][L][3], cost [n][iterations + 1] (or None), status [n]"""
        n = self.n_windows
        poses = np.empty((n, self.n_cams, 12))
        lm = np.empty((n, self.max_landmarks, 3))
        cost = np.empty((n, self.iterations + 1)) if self.iterations >= 0 else None
        status = np.empty(n, np.int32)
        _check(self.lib.slamx_gn_download(self.h, _ptr(poses), _ptr(lm), _ptr(cost), _ptr(status)))
        return poses, lm, cost, status

    def set_profiling(self, on=True, every=1):
        _check(self.lib.slamx_gn_set_profiling(self.h, (max(1, int(every)) if on else 0)))

    def stage_ms(self):
        """-> ({"accumulate": ms, "solve": ms, "backsub": ms}, launches per stage)"""
        ms = np.zeros(3, np.float32)
        n = np.zeros(3, np.int32)
        _check(self.lib.slamx_gn_stage_ms(self.h, _ptr(ms), _ptr(n)))
        names = ("accumulate", "solve", "backsub")
        return dict(zip(names, [float(x) for x in ms])), dict(zip(names, [int(x) for x in n]))

    def normal_equations(self, K, lam, n_fixed):
        """Linearise at the current state without stepping -> S [n][6W][6W], g, delta [n][6W], cost [n]"""
        n, m = self.n_windows, 6 * self.n_cams
        k4 = np.ascontiguousarray(K, np.float64).reshape(4)
        S, g, d, c = np.empty((n, m, m)), np.empty((n, m)), np.empty((n, m)), np.empty(n)
        _check(self.lib.slamx_gn_normal_equations(self.h, _ptr(k4), float(lam), n_fixed, _ptr(S), _ptr(g), _ptr(d), _ptr(c)))
        return S, g, d, c
