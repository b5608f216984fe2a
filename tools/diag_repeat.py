"""Diagnostic: repeat detect_and_compute on the c2 workload and compare every run with the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_product
import oracle_binding

slamx = load_product()
oracle = oracle_binding.load()
rows, cols, window, n_rect, rmin, rmax = 480, 640, 6, 5000, 3, 16
d = slamx.FeatureDetector(slamx.DetectorConfig(20, 12, 1, window, 31, 256))
Pp = oracle.det_params(d.cfg)
frames = [slamx.synth_frame(0x5EED0051, k, rows, cols, n_rect, rmin, rmax) for k in range(2)]
ref = [oracle.detect_and_compute(f, Pp) for f in frames]
refblur = [oracle.gaussian_blur(f) for f in frames]
b = slamx.Batch(d, rows, cols, 2, max_candidates=131072, max_keypoints=16384)
b.upload(np.stack(frames))
nbad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    b.detect_and_compute(2)
    for k in range(2):
        kg, dg = b.download(k)
        if kg.shape != ref[k][0].shape:
            print("iter", it, "frame", k, "count", kg.shape, ref[k][0].shape); nbad += 1; continue
        bad = np.nonzero([x.tobytes() != y.tobytes() for x, y in zip(kg, ref[k][0])])[0]
        badd = np.nonzero((dg != ref[k][1]).any(axis=1))[0]
        if len(bad) or len(badd):
            nbad += 1
            print("iter", it, "frame", k, "kp bad", bad[:5], [ (kg[i], ref[k][0][i]) for i in bad[:3]], "desc bad", badd[:5])
    for k in range(2):
        up = b.download_image(k)
        if not np.array_equal(up, frames[k]):
            yy, xx = np.nonzero(up != frames[k]); nbad += 1
            print("iter", it, "UPLOADED frame", k, "differs at", list(zip(yy[:8], xx[:8])), len(yy))
        bb = b.download_image(k, blurred=True)
        if not np.array_equal(bb, refblur[k]):
            yy, xx = np.nonzero(bb != refblur[k]); nbad += 1
            print("iter", it, "BATCH blur frame", k, "differs at", list(zip(yy[:8], xx[:8])), len(yy))
    for k in range(2):
        g = d.gaussian_blur(frames[k])
        if not np.array_equal(g, refblur[k]):
            yy, xx = np.nonzero(g != refblur[k])
            nbad += 1
            print("iter", it, "blur frame", k, "differs at", list(zip(yy[:5], xx[:5])))
kp0 = ref[0][0]
k1, d1 = d.compute(frames[0], kp0.copy())
bad = np.nonzero([x.tobytes() != y.tobytes() for x, y in zip(k1, kp0)])[0]
print("single-frame compute on oracle keypoints: bad", bad[:5], [(k1[i], kp0[i]) for i in bad[:3]])
k2, d2 = d.detectAndCompute(frames[0])
bad = np.nonzero([x.tobytes() != y.tobytes() for x, y in zip(k2, kp0)])[0]
print("single-frame detectAndCompute: bad", bad[:5])
print("mismatching runs:", nbad)
