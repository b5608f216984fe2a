"""Latency of ONE detect call on a dense 1080p frame (c3's frames: ~80k candidates, more than LDS holds) -- development aid."""
import importlib.util, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
mod = bench.load_product()
ctx = mod.Context(0)
det = mod.FeatureDetector(mod.DetectorConfig(20, 12, 1, 8, 31, 256), ctx)
for name, nrect, rmin, rmax in (("c3 frame", 32000, 3, 16), ("1080p_2k frame", 6800, 3, 24)):
    img = mod.synth_frame(bench.SEED, 0, 1080, 1920, nrect, rmin, rmax)
    k = det.detect(img)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); k = det.detect(img); ts.append((time.perf_counter() - t0) * 1e6)
    print("%s: %d keypoints, detect median %.0f us (min %.0f)" % (name, len(k), sorted(ts)[len(ts) // 2], min(ts)))
