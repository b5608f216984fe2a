#!/usr/bin/env python3
"""Replays ONE case of tests/fuzz/fuzz_detector.py (same generator, same seed, the draws of the cases before it consumed on the CPU
only) and says which keypoints' descriptors differ from the oracle's -- how the one-bit difference of seed 404 / case 158 was
pinned to the keypoint at (17, 18) and, with SLAMX_DESCRIBE_TILES=0 / SLAMX_NO_FUSED_DESCRIBE=1, to describe_tile_kernel.
Test infrastructure.  usage: tests/fuzz/replay_fuzz_case.py <seed> <case>"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
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
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "fuzz_case_img.npy"), img)
