import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo/tests")
from conftest import load_product
slamx = load_product()
rows, cols = 1080, 1920
nb = int(sys.argv[1]); B = int(sys.argv[2]); steps = 10
uniq = 32
base = np.stack([slamx.synth_frame(1234, k, rows, cols, 6800, 3, 24) for k in range(uniq)])
pipes = []
for p in range(nb):
    ctx = slamx.Context(0)
    det = slamx.FeatureDetector(slamx.DetectorConfig(20, 12, 1, 12, 31, 256), ctx)
    mat = slamx.FeatureMatcher(slamx.MatcherConfig(0, 0, 20, 1, 0.5), ctx)
    b = slamx.Batch(det, rows, cols, B, max_candidates=32768, max_keypoints=4096)
    for s in range(0, B, uniq):
        b.upload(base[: min(uniq, B - s)], first=s)
    pipes.append((ctx, det, mat, b))
def step():
    for ctx, det, mat, b in pipes:
        b.detect_and_compute(B)
        b.match_consecutive(mat, B, use_keypoints=True, first_train_frame=B - 1)
for _ in range(2): step()
for p in pipes: p[0].synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
for p in pipes: p[0].synchronize()
dt = time.perf_counter() - t0
print("pipes", nb, "B", B, "fps", nb * B * steps / dt, "ms/step", dt / steps * 1e3)
