"""ctypes binding of the CPU oracle (oracle/slam_oracle.cpp).  TEST INFRASTRUCTURE ONLY: imported
from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_build", "libslam_oracle.so")

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4")])
MATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("distance", "<f4")])


class DetectorParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("intensityThreshold", "contiguousPixelsThreshold", "nonMaxSuppression",
                                       "suppressionWindowSize", "patchSize", "numBRIEFPairs")]


class MatcherParams(C.Structure):
    _fields_ = [("filterMatches", C.c_int), ("goodMatchesCount", C.c_int), ("useRatioTest", C.c_int),
                ("ratioTestThreshold", C.c_float)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.oracle_atan2f.restype = C.c_float
        lib.oracle_atan2f.argtypes = [C.c_float, C.c_float]
        lib.oracle_cosf.restype = C.c_float
        lib.oracle_cosf.argtypes = [C.c_float]
        lib.oracle_sinf.restype = C.c_float
        lib.oracle_sinf.argtypes = [C.c_float]
        lib.oracle_angle_deg.restype = C.c_float
        lib.oracle_angle_deg.argtypes = [C.c_float, C.c_float]
        lib.oracle_angle_cos_sin.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.oracle_const_bits.restype = C.c_uint32
        lib.oracle_trig_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_sort_desc.argtypes = [C.c_void_p, C.c_int, C.c_void_p]

    @staticmethod
    def det_params(cfg):
        """cfg: any object with the six DetectorConfig fields (slamx.DetectorConfig works)."""
        return DetectorParams(cfg.intensity_threshold, cfg.contiguous_pixels_threshold, cfg.non_max_suppression,
                              cfg.suppression_window_size, cfg.patch_size, cfg.num_brief_pairs)

    @staticmethod
    def match_params(cfg):
        return MatcherParams(cfg.filter_matches, cfg.good_matches_count, cfg.use_ratio_test, cfg.ratio_test_threshold)

    def brief_pattern(self, patch, npairs):
        xy = np.zeros((npairs, 4), np.int32)
        n = self.lib.oracle_brief_pattern(patch, npairs, _p(xy), npairs)
        return xy[:n].copy()

    def _kp_call(self, fn, img, P):
        img = np.ascontiguousarray(img, np.uint8)
        cap = max(1024, img.size // 8)
        while True:
            kps = np.empty(cap, KEYPOINT_DTYPE)
            n = fn(_p(img), img.shape[0], img.shape[1], img.strides[0], C.byref(P), _p(kps), cap)
            if n <= cap:
                return kps[:n].copy()
            cap = n

    def fast_candidates(self, img, P):
        return self._kp_call(self.lib.oracle_fast_candidates, img, P)

    def detect(self, img, P):
        return self._kp_call(self.lib.oracle_detect, img, P)

    def gaussian_blur(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.empty(img.shape, np.uint8)
        self.lib.oracle_gaussian_blur(_p(img), img.shape[0], img.shape[1], img.strides[0], _p(out))
        return out

    def compute(self, img, kps, P, want_moments=False):
        img = np.ascontiguousarray(img, np.uint8)
        k = np.ascontiguousarray(kps, KEYPOINT_DTYPE).copy()
        n = k.shape[0]
        width = P.numBRIEFPairs // 8
        desc = np.zeros((n, width), np.uint8)
        mom = np.zeros((n, 2), np.float32)
        self.lib.oracle_compute(_p(img), img.shape[0], img.shape[1], img.strides[0], C.byref(P), _p(k), n, _p(desc),
                                _p(mom))
        if n == 0:
            desc = np.zeros((0, 0), np.uint8)
        return (k, desc, mom) if want_moments else (k, desc)

    def detect_and_compute(self, img, P):
        k = self.detect(img, P)
        return self.compute(img, k, P)

    def _match_call(self, fn, d1, d2, k1, k2, P):
        d1 = np.ascontiguousarray(d1, np.uint8)
        d2 = np.ascontiguousarray(d2, np.uint8)
        k1 = np.ascontiguousarray(k1, KEYPOINT_DTYPE) if k1 is not None else None
        k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE) if k2 is not None else None
        n1, n2 = d1.shape[0], d2.shape[0]
        out = np.empty(max(n1, 1), MATCH_DTYPE)
        n = fn(_p(d1), n1, _p(d2), n2, d1.shape[1] if d1.ndim == 2 else 0, _p(k1), 0 if k1 is None else k1.shape[0],
               _p(k2), 0 if k2 is None else k2.shape[0], C.byref(P), _p(out), out.shape[0])
        if n < 0:
            raise ValueError("Empty descriptors provided.")
        return out[:n].copy()

    def best_matches(self, d1, d2, k1, k2, P):
        return self._match_call(self.lib.oracle_best_matches, d1, d2, k1, k2, P)

    def match(self, d1, d2, k1, k2, P):
        return self._match_call(self.lib.oracle_match, d1, d2, k1, k2, P)

    def trig_chain(self, m01, m10):
        a = np.ascontiguousarray(m01, np.float32)
        b = np.ascontiguousarray(m10, np.float32)
        deg, c, s = (np.empty_like(a) for _ in range(3))
        self.lib.oracle_trig_chain(_p(a), _p(b), a.shape[0], _p(deg), _p(c), _p(s))
        return deg, c, s

    # ---- pose rows a11-a14 (pose_oracle.cpp) ----
    def svd(self, A):
        A = np.ascontiguousarray(A, np.float64)
        m, n = A.shape
        U, S, V = np.empty((m, n)), np.empty(n), np.empty((n, n))
        self.lib.oracle_svd(m, n, _p(A), _p(U), _p(S), _p(V))
        return U, S, V

    def pnp_from_null_vector(self, nullvec):
        """loop_closure.cpp:260-271 from a given DLT null vector (either sign)."""
        p = np.ascontiguousarray(nullvec, np.float64)
        R, t = np.empty((3, 3)), np.empty(3)
        self.lib.oracle_pnp_from_null_vector(_p(p), _p(R), _p(t))
        return R, t

    def solve_pnp(self, p3d, p2d):
        a = np.ascontiguousarray(p3d, np.float64)
        b = np.ascontiguousarray(p2d, np.float64)
        R, t, nv = np.empty((3, 3)), np.empty(3), np.empty(12)
        ok = self.lib.oracle_solve_pnp(a.shape[0], _p(a), _p(b), _p(R), _p(t), _p(nv))
        return ok, R, t, nv

    def pnp_calibrated_from_null_vector(self, nullvec):
        """SLAMX_GEOMETRY_CALIBRATED (build-defined): row-major map, det(R0) > 0, t0 / mean singular value."""
        p = np.ascontiguousarray(nullvec, np.float64)
        R, t = np.empty((3, 3)), np.empty(3)
        self.lib.oracle_pnp_calibrated_from_null_vector(_p(p), _p(R), _p(t))
        return R, t

    def solve_pnp_calibrated(self, p3d, p2d, K):
        a = np.ascontiguousarray(p3d, np.float64)
        b = np.ascontiguousarray(p2d, np.float64)
        R, t, nv = np.empty((3, 3)), np.empty(3), np.empty(12)
        ok = self.lib.oracle_solve_pnp_calibrated(a.shape[0], _p(a), _p(b), _p(np.ascontiguousarray(K, np.float64)), _p(R), _p(t), _p(nv))
        return ok, R, t, nv

    def sampson_count_plain(self, E, pts1, pts2, K, threshold_px):
        """Sampson inlier count of one E in ordinary (unfused) arithmetic: the independent check of the fused chains."""
        a = np.ascontiguousarray(pts1, np.float32)
        b = np.ascontiguousarray(pts2, np.float32)
        self.lib.oracle_sampson_count_plain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double]
        return self.lib.oracle_sampson_count_plain(_p(np.ascontiguousarray(E, np.float64)), _p(a), _p(b), a.shape[0],
                                                   _p(np.ascontiguousarray(K, np.float64)), float(threshold_px))

    def count_inliers(self, p3d, p2d, K, R, t, thr):
        a = np.ascontiguousarray(p3d, np.float64)
        b = np.ascontiguousarray(p2d, np.float64)
        self.lib.oracle_count_inliers.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        return self.lib.oracle_count_inliers(a.shape[0], _p(a), _p(b), _p(np.ascontiguousarray(K, np.float64)),
                                             _p(np.ascontiguousarray(R, np.float64)), _p(np.ascontiguousarray(t, np.float64)), float(thr))

    def essential_ransac(self, pts1, pts2, K, iterations=512, threshold_px=1.0, seed=0):
        a = np.ascontiguousarray(pts1, np.float32)
        b = np.ascontiguousarray(pts2, np.float32)
        n = a.shape[0]
        E = np.empty((3, 3))
        cnt = C.c_int()
        mask = np.zeros(n, np.uint8)
        E_all = np.empty((iterations, 9))
        counts = np.empty(iterations, np.int32)
        self.lib.oracle_essential_ransac.restype = C.c_int
        best = self.lib.oracle_essential_ransac(_p(a), _p(b), n, _p(np.ascontiguousarray(K, np.float64)), iterations,
                                                C.c_double(threshold_px), C.c_uint(seed), _p(E), C.byref(cnt),
                                                _p(mask), _p(E_all), _p(counts))
        return E, cnt.value, mask, best, E_all, counts

    def essential_from_samples(self, pts1, pts2, K, table, threshold_px=1.0):
        """The same loop with the sample table given (iterations x 8 indices)."""
        a = np.ascontiguousarray(pts1, np.float32)
        b = np.ascontiguousarray(pts2, np.float32)
        tb = np.ascontiguousarray(table, np.int32)
        n, iterations = a.shape[0], tb.shape[0]
        E = np.empty((3, 3))
        cnt = C.c_int()
        mask = np.zeros(n, np.uint8)
        E_all = np.empty((iterations, 9))
        counts = np.empty(iterations, np.int32)
        self.lib.oracle_essential_from_samples.restype = C.c_int
        best = self.lib.oracle_essential_from_samples(_p(a), _p(b), n, _p(np.ascontiguousarray(K, np.float64)), _p(tb),
                                                      iterations, C.c_double(threshold_px), _p(E), C.byref(cnt), _p(mask),
                                                      _p(E_all), _p(counts))
        return E, cnt.value, mask, best, E_all, counts

    def recover_pose(self, E, pts1, pts2, K):
        a = np.ascontiguousarray(pts1, np.float32)
        b = np.ascontiguousarray(pts2, np.float32)
        R, t, front = np.empty((3, 3)), np.empty(3), np.zeros(4, np.int32)
        self.lib.oracle_recover_pose(_p(np.ascontiguousarray(E, np.float64)), _p(a), _p(b), a.shape[0],
                                     _p(np.ascontiguousarray(K, np.float64)), _p(R), _p(t), _p(front))
        return R, t, front

    def triangulate(self, P1, P2, pts1, pts2):
        a = np.ascontiguousarray(pts1, np.float32)
        b = np.ascontiguousarray(pts2, np.float32)
        n = a.shape[0]
        x4, xyz = np.empty((n, 4), np.float32), np.empty((n, 3))
        self.lib.oracle_triangulate(_p(np.ascontiguousarray(P1, np.float64)), _p(np.ascontiguousarray(P2, np.float64)),
                                    _p(a), _p(b), n, _p(x4), _p(xyz))
        return x4, xyz

    def undistort(self, img, cam):
        img = np.ascontiguousarray(img, np.uint8)
        out = np.empty(img.shape, np.float64)
        self.lib.oracle_undistort.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_double] * 8 + [C.c_void_p]
        self.lib.oracle_undistort(_p(img), img.shape[0], img.shape[1], img.strides[0], cam.fx, cam.fy, cam.cx, cam.cy,
                                  cam.k1, cam.k2, cam.p1, cam.p2, _p(out))
        return out

    def penalty(self, dist, qx, qy, tx, ty):
        d = np.ascontiguousarray(dist, np.int32)
        arrs = [np.ascontiguousarray(a, np.float32) for a in (qx, qy, tx, ty)]
        out = np.empty_like(d)
        self.lib.oracle_penalty(_p(d), *[_p(a) for a in arrs], d.shape[0], _p(out))
        return out

    def antiqsort(self, n):
        vals = np.empty(n, np.uint32)
        self.lib.oracle_antiqsort(n, _p(vals))
        return vals

    def sort_desc(self, scores):
        s = np.ascontiguousarray(scores, np.uint32)
        perm = np.empty(s.shape[0], np.uint32)
        self.lib.oracle_sort_desc(_p(s), s.shape[0], _p(perm))
        return perm

    # ---- the restated CPU path over a sequence of frames on T threads (oracle/cpu_sequence.cpp)
    def sequence_mt(self, frames, P, M, use_keypoints=True, n_threads=1, reps=1):
        """frames [n][rows][cols] u8; frame k matched against k-1 (frame 0 against the last).  Returns (wall seconds,
        per_frame [n][3] = keypoints, matches, FNV-1a of keypoints + descriptors + matches).  reps: passes of every
        thread over its chunk (n * reps frames processed)."""
        f = np.ascontiguousarray(frames, np.uint8)
        out = np.zeros((f.shape[0], 3), np.uint64)
        fn = self.lib.oracle_sequence_mt
        fn.restype = C.c_double
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        dt = fn(_p(f), f.shape[0], f.shape[1], f.shape[2], C.addressof(P), C.addressof(M), 1 if use_keypoints else 0,
                int(n_threads), int(reps), _p(out))
        return dt, out

    # ---- image pyramid (oracle/pyramid_oracle.cpp; build-defined, no reference counterpart)
    def downscale(self, img, drows, dcols):
        src = np.ascontiguousarray(img, np.uint8)
        dst = np.empty((drows, dcols), np.uint8)
        self.lib.oracle_downscale.restype = None
        self.lib.oracle_downscale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        self.lib.oracle_downscale(_p(src), src.shape[0], src.shape[1], _p(dst), drows, dcols)
        return dst

    def pyramid_sizes(self, rows, cols, n_levels, scale_factor):
        fn = self.lib.oracle_pyramid_level_size
        fn.restype = C.c_int
        fn.argtypes = [C.c_int, C.c_double, C.c_int]
        return [(fn(rows, scale_factor, l), fn(cols, scale_factor, l)) for l in range(n_levels)]

    def pyramid_images(self, img, n_levels, scale_factor):
        sizes = self.pyramid_sizes(img.shape[0], img.shape[1], n_levels, scale_factor)
        out = [np.ascontiguousarray(img, np.uint8)]
        for r, c in sizes[1:]:
            out.append(self.downscale(out[-1], r, c))  # every level from the one above it
        return out

    def pyramid_detect_and_compute(self, img, P, n_levels, scale_factor):
        """The unchanged detector on every level, results concatenated level after level with x, y, size
        multiplied by float32(scale_factor ** level) -- what slamx_pyramid_detect_and_compute returns."""
        kps, descs = [], []
        for l, im in enumerate(self.pyramid_images(img, n_levels, scale_factor)):
            k, d = self.detect_and_compute(im, P)
            k = k.copy()
            sc = np.float32(float(scale_factor) ** l)
            for name in ("x", "y", "size"):
                k[name] = k[name] * sc
            kps.append(k)
            if d.size:
                descs.append(d)
        kp = np.concatenate(kps)
        width = P.numBRIEFPairs // 8
        de = np.concatenate(descs) if descs else np.zeros((0, width), np.uint8)
        return kp, de

    # ---- sliding-window Gauss-Newton (oracle/gn_oracle.cpp; build-defined problem, no reference counterpart)
    def gn_step_dense(self, poses, X, obs, K, lam, n_fixed):
        """One window, full dense normal equations -> dict(delta_c [W,6], delta_x [L,3], cost, S, g, active, rc)"""
        poses = np.ascontiguousarray(poses, np.float64)
        X = np.ascontiguousarray(X, np.float64)
        obs = np.ascontiguousarray(obs, np.float32)
        W, L = poses.shape[0], X.shape[0]
        assert obs.shape == (L, W, 2)
        k4 = np.ascontiguousarray(K, np.float64)
        dc, dx = np.zeros((W, 6)), np.zeros((L, 3))
        S, g = np.zeros((6 * W, 6 * W)), np.zeros(6 * W)
        act = np.zeros(L, np.uint8)
        cost = C.c_double()
        fn = self.lib.oracle_gn_step_dense
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_void_p,
                       C.c_void_p, C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p]
        rc = fn(_p(poses), _p(X), _p(obs), W, L, _p(k4), float(lam), n_fixed, _p(dc), _p(dx), C.byref(cost), _p(S), _p(g), _p(act))
        return dict(delta_c=dc, delta_x=dx, cost=cost.value, S=S, g=g, active=act, rc=rc)

    def gn_apply(self, poses, X, delta_c, delta_x):
        poses = np.array(poses, np.float64, order="C")
        X = np.array(X, np.float64, order="C")
        dc = np.ascontiguousarray(delta_c, np.float64)
        dx = np.ascontiguousarray(delta_x, np.float64)
        self.lib.oracle_gn_apply.restype = None
        self.lib.oracle_gn_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self.lib.oracle_gn_apply(_p(poses), _p(X), poses.shape[0], X.shape[0], _p(dc), _p(dx))
        return poses, X

    def gn_cost(self, poses, X, obs, K, lam=0.0):
        poses = np.ascontiguousarray(poses, np.float64)
        X = np.ascontiguousarray(X, np.float64)
        obs = np.ascontiguousarray(obs, np.float32)
        k4 = np.ascontiguousarray(K, np.float64)
        self.lib.oracle_gn_cost.restype = C.c_double
        self.lib.oracle_gn_cost.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double]
        return self.lib.oracle_gn_cost(_p(poses), _p(X), _p(obs), poses.shape[0], X.shape[0], _p(k4), float(lam))

    def gn_step_schur(self, poses, X, obs, K, lam, n_fixed):
        """One window, eliminated the way a CPU bundle adjuster does; returns (cost before, poses, X)."""
        poses = np.array(poses, np.float64, order="C")
        X = np.array(X, np.float64, order="C")
        obs = np.ascontiguousarray(obs, np.float32)
        k4 = np.ascontiguousarray(K, np.float64)
        fn = self.lib.oracle_gn_step_schur
        fn.restype = C.c_double
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_int]
        c = fn(_p(poses), _p(X), _p(obs), poses.shape[0], X.shape[0], _p(k4), float(lam), n_fixed)
        return c, poses, X


_cached = None


def load():
    global _cached
    if _cached is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _cached = Oracle(C.CDLL(LIB))
    return _cached
