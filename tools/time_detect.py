"""Stage timings of detect_and_compute on the default bench workload (no result checks: usable with
experimental kernel variants whose outputs are wrong).  usage: time_detect.py [frames] [reps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_product
slamx = load_product()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows, cols = 1080, 1920
d = slamx.FeatureDetector(slamx.DetectorConfig(20, 12, 1, 12, 31, 256))
uniq = 32
base = np.stack([slamx.synth_frame(1234, k, rows, cols, 6800, 3, 24) for k in range(uniq)])
b = slamx.Batch(d, rows, cols, B, max_candidates=32768, max_keypoints=8192)
for s in range(0, B, uniq):
    b.upload(base[: min(uniq, B - s)], first=s)
b.set_profiling(reps)
for _ in range(reps + 2):
    b.detect_and_compute(B)
d.ctx.synchronize()
ms = [b.stage_ms(i) for i in range(reps)]
print(" ".join("%s=%.3f" % (n, sum(m[n] for m in ms) / reps) for n in ms[0]))
