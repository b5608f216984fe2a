"""Synthetic sliding windows for the Gauss-Newton tests and bench.py's c4 workload: W cameras on a
gently curving path looking at a cloud of L landmarks, pinhole projections with pixel noise, a
perturbed starting point.  Pure numpy, deterministic per seed."""
import numpy as np


def rodrigues(phi):
    th = np.linalg.norm(phi)
    Kx = np.array([[0, -phi[2], phi[1]], [phi[2], 0, -phi[0]], [-phi[1], phi[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Kx
    return np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx


def make_window(seed, W, L, K=(1700.0, 1700.0, 960.0, 540.0), noise_px=0.5, drop=0.1, pose_sigma=0.01, point_sigma=0.05,
                size=(1920, 1080)):
    """-> dict(poses_gt, X_gt, poses0, X0, obs [L, W, 2] float32 with NaN = not observed, K)"""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K
    poses_gt = np.zeros((W, 12))
    for c in range(W):
        # world -> camera: camera centre moves along x with a slight yaw
        Rc = rodrigues(np.array([0.01 * np.sin(c), 0.02 * c, 0.005 * c]))
        centre = np.array([0.25 * c, 0.02 * np.sin(0.7 * c), 0.05 * c])
        poses_gt[c, :9] = Rc.reshape(9)
        poses_gt[c, 9:] = -Rc @ centre
    X_gt = np.stack([rng.uniform(-4, 6, L), rng.uniform(-2.5, 2.5, L), rng.uniform(4, 12, L)], 1)
    obs = np.full((L, W, 2), np.nan, np.float32)
    for c in range(W):
        Rc = poses_gt[c, :9].reshape(3, 3)
        p = X_gt @ Rc.T + poses_gt[c, 9:]
        u = fx * p[:, 0] / p[:, 2] + cx + rng.normal(0, noise_px, L)
        v = fy * p[:, 1] / p[:, 2] + cy + rng.normal(0, noise_px, L)
        ok = (p[:, 2] > 0.1) & (u >= 0) & (u < size[0]) & (v >= 0) & (v < size[1]) & (rng.random(L) >= drop)
        obs[ok, c, 0] = u[ok]
        obs[ok, c, 1] = v[ok]
    poses0 = poses_gt.copy()
    for c in range(W):
        E = rodrigues(rng.normal(0, pose_sigma, 3))
        poses0[c, :9] = (E @ poses_gt[c, :9].reshape(3, 3)).reshape(9)
        poses0[c, 9:] = E @ poses_gt[c, 9:] + rng.normal(0, pose_sigma, 3)
    X0 = X_gt + rng.normal(0, point_sigma, X_gt.shape)
    return dict(poses_gt=poses_gt, X_gt=X_gt, poses0=poses0, X0=X0, obs=obs, K=np.array(K, np.float64))
