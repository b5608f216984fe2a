"""Per-frame look at the pose stage on bench.py's scene (development aid): errors against the generating motion,
cheirality votes, PnP counts.  python tools/dev/pose_scene_diag.py [frames] [points] [noise_px] [outliers]"""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pose_scene as ps
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
mod = bench.load_product()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
noise = float(sys.argv[3]) if len(sys.argv) > 3 else bench.POSE_SCENE["noise_px"]
outl = float(sys.argv[4]) if len(sys.argv) > 4 else bench.POSE_SCENE["outliers"]
K = np.array([[0.9 * 1920, 0, 960.0], [0, 0.9 * 1920, 540.0], [0, 0, 1.0]])
frames, Rw, tw = ps.loop_sequence(mod, np.random.default_rng(bench.SEED), F, N, K, outliers=outl, noise=noise, loops=max(1, round(0.05 * 256)) if F == 256 else None)
ctx = mod.Context(0)
det = mod.FeatureDetector(mod.DetectorConfig(20, 12, 1, 8, 31, 256), ctx)
b = mod.Batch(det, 64, 64, F, 1024, 16384)
for k, (kp, m) in enumerate(frames):
    b.upload_results(k, kp, m)
pose = mod.PoseStage(b, K, iterations=256, threshold_px=1.0, seed=bench.SEED & 0xFFFFFFFF, pnp_iterations=100, pnp_threshold=2.0,
                     pnp_min_correspondences=20, geometry_mode=mod.GEOMETRY_CALIBRATED)
pose.run(F, F - 1)
res = pose.download(F)
flips = 0
for f in range(F):
    Rr, tr = ps.relative_pose(Rw, tw, (f - 1) % F, f)
    re, de = ps.rotation_angle_deg(res["R"][f], Rr), ps.direction_angle_deg(res["t"][f], tr)
    flips += de > 90
    if f < 24 or de > 5:
        print("f %3d rot %.4f dir %8.4f front %s inl %d/%d pnp %d/%d" % (f, re, de, res["front"][f].tolist(), res["n_inliers"][f], res["n_matches"][f],
                                                                       res["pnp_inliers"][f], res["pnp_correspondences"][f]))
print("flipped translations:", flips, "of", F, " median rot/dir err", np.median([ps.rotation_angle_deg(res["R"][f], ps.relative_pose(Rw, tw, (f - 1) % F, f)[0]) for f in range(F)]),
      np.median([ps.direction_angle_deg(res["t"][f], ps.relative_pose(Rw, tw, (f - 1) % F, f)[1]) for f in range(F)]))
print("pnp share", res["pnp_inliers"].sum() / max(res["pnp_correspondences"].sum(), 1))
