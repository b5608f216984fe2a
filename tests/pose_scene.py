"""Synthetic scenes WITH DEPTH for the pose stage (slamx_batch_pose): a camera moving through a 3-D point cloud,
keypoints = projections (float32) in a per-frame random order, match lists in ascending queryIdx order with wrong
matches mixed in -- the arrays slamx_batch_upload_results writes into a batch's keypoint / match slots.

Used by tests/test_pose_batch_gpu.py, tests/fuzz/fuzz_pose.py and by `bench.py --with-pose` (the c3 workload's own
frames are a planar pattern under a pure image translation: E is degenerate there and nothing has depth).
Host-side numpy only; `slamx` is the product module (for its record dtypes).
"""
import numpy as np

K_KITTI = np.array([[984.2439, 0.0, 690.0], [0.0, 980.8141, 233.1966], [0.0, 0.0, 1.0]])  # test/data/camera.yml K0


def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _frame(slamx, rng, X, K, R, t, noise, prev_slot, keep, outliers):
    """One frame's keypoints and its match list against the frame whose slot table is prev_slot (None: no matches)."""
    n_points = X.shape[0]
    x = (K @ (R @ X.T + t[:, None])).T
    uv = x[:, :2] / x[:, 2:] + rng.normal(0, noise, (n_points, 2))
    order = rng.permutation(n_points)          # keypoint slot i of this frame shows world point order[i]
    slot = np.empty(n_points, np.int64)
    slot[order] = np.arange(n_points)
    kps = np.zeros(n_points, slamx.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = uv[order, 0].astype(np.float32), uv[order, 1].astype(np.float32)
    kps["size"] = 6.0
    matches = np.zeros(0, slamx.MATCH_DTYPE)
    if prev_slot is not None:
        q = np.sort(rng.choice(n_points, int(keep * n_points), replace=False))  # query keypoints that found a match
        tr = prev_slot[order[q]]                                              # the same world point in the train frame
        bad = rng.random(len(q)) < outliers
        tr[bad] = rng.integers(0, n_points, int(bad.sum()))
        matches = np.zeros(len(q), slamx.MATCH_DTYPE)
        matches["queryIdx"], matches["trainIdx"], matches["distance"] = q, tr, 3.0
    return kps, matches, slot


def sequence(slamx, rng, n_frames, n_points, outliers=0.2, noise=0.15, keep=0.85, K=K_KITTI):
    """Keypoints and match lists of a camera moving away through a point cloud (a handful of frames).
    -> per frame (kps, matches); frame 0 has no matches."""
    X = rng.uniform(-4, 4, (n_points, 3)) + np.array([0, 0, 11.0])
    frames = []
    prev_slot = None
    for k in range(n_frames):
        R = rot(*(0.02 * k * rng.uniform(0.5, 1.5, 3)))
        t = np.array([0.25 * k, 0.03 * k, 0.05 * k]) + rng.uniform(-0.02, 0.02, 3)
        kps, matches, prev_slot = _frame(slamx, rng, X, K, R, t, noise, prev_slot, keep, outliers)
        frames.append((kps, matches))
    return frames


def loop_camera(k, n_frames, loops, radius=0.8):
    """World -> camera (R, t) of frame k of a closed trajectory: `loops` turns of a circle in the x-z plane with a
    small vertical figure of eight and a gentle periodic rotation; frame n_frames would be frame 0 again."""
    th = 2.0 * np.pi * loops * k / n_frames
    c = np.array([radius * np.cos(th), 0.35 * radius * np.sin(2 * th), radius * np.sin(th)])
    R = rot(0.03 * np.sin(th), 0.04 * np.cos(th), 0.02 * np.sin(2 * th))
    return R, -R @ c


def loop_sequence(slamx, rng, n_frames, n_points, K, outliers=0.2, noise=0.02, keep=0.85, loops=None):
    """A CLOSED sequence for batches whose frame 0 is matched against the batch's last frame (bench.py's step):
    every frame, frame 0 included, has a match list against its predecessor on the loop, and the baseline between
    neighbours is the same all the way round (about 0.25 world units for 256 frames).  The cloud fills a 1080p view
    at the default K of bench.py (f = 0.9 * 1920).
    -> (frames [(kps, matches)], R [n, 3, 3], t [n, 3]): the generating world -> camera poses."""
    if loops is None:
        loops = max(1, int(round(0.05 * n_frames)))
    X = np.stack([rng.uniform(-3.0, 3.0, n_points), rng.uniform(-1.7, 1.7, n_points), rng.uniform(9.0, 15.0, n_points)], 1)
    poses = [loop_camera(k, n_frames, loops) for k in range(n_frames)]
    made = []
    prev_slot = None
    for k in range(n_frames):
        kps, matches, slot = _frame(slamx, rng, X, K, poses[k][0], poses[k][1], noise, prev_slot, keep, outliers)
        made.append([kps, matches, slot])
        prev_slot = slot
    # close the loop: frame 0's matches against the last frame's slots (same keypoints, new match list)
    kps0, _, slot0 = made[0]
    order0 = np.empty(n_points, np.int64)
    order0[slot0] = np.arange(n_points)
    q = np.sort(rng.choice(n_points, int(keep * n_points), replace=False))
    tr = made[-1][2][order0[q]]
    bad = rng.random(len(q)) < outliers
    tr[bad] = rng.integers(0, n_points, int(bad.sum()))
    m0 = np.zeros(len(q), slamx.MATCH_DTYPE)
    m0["queryIdx"], m0["trainIdx"], m0["distance"] = q, tr, 3.0
    made[0][1] = m0
    return ([(a, b) for a, b, _ in made], np.stack([p[0] for p in poses]), np.stack([p[1] for p in poses]))


def relative_pose(R, t, train, query):
    """(R_rel, t_rel) with x_query = R_rel x_train + t_rel from world -> camera poses."""
    Rr = R[query] @ R[train].T
    return Rr, t[query] - Rr @ t[train]


def rotation_angle_deg(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1.0) / 2.0
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))))


def direction_angle_deg(a, b):
    c = float(np.dot(a, b) / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))))
