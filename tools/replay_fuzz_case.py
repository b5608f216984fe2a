import importlib, os, sys
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests")); sys.path.insert(0, os.path.join(ROOT,"tools"))
import oracle_binding
import fuzz_detector as fz
slamx = importlib.import_module("slam-cin0051_amd")
oracle = oracle_binding.load()
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(target + 1):
    rows = int(rng.integers(7, 140))
    cols = int(rng.choice([rng.integers(7, 80), rng.integers(200, 300), rng.integers(500, 560)]))
    t = int(rng.choice([rng.integers(0, 8), rng.integers(8, 64), rng.integers(64, 256)]))
    arc = int(rng.integers(0, 17))
    window = int(rng.integers(1, 16))
    img = np.ascontiguousarray(fz.image(rng, rows, cols).astype(np.uint8))
print("case", target, rows, cols, t, arc, window)
cfg = slamx.DetectorConfig(t, arc, 1, window, 31, 256)
det = slamx.FeatureDetector(cfg)
P = oracle.det_params(cfg)
k_ref, d_ref = oracle.detect_and_compute(img, P)
k_got, d_got = det.detectAndCompute(img)
print("keypoints", len(k_ref), len(k_got), "kps equal", np.array_equal(k_got, k_ref))
bad = np.nonzero((d_got != d_ref).any(axis=1))[0]
print("rows differing:", len(bad))
for i in bad[:12]:
    print(i, k_ref[i], "ref", d_ref[i][:8], "got", d_got[i][:8])
xs = np.array([k_ref[i][0] for i in bad]); ys = np.array([k_ref[i][1] for i in bad])
if len(bad):
    print("x range", xs.min(), xs.max(), "y range", ys.min(), ys.max())
np.save("/root/repo/gpurun_out/fuzz_case_img.npy", img)
