#!/usr/bin/env python3
"""How long the HOST spends inside each call of the end-to-end leg (is any of the "async" calls synchronous?).
    python tools/e2e_host_times.py [workload] [lanes]"""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    n_lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    wl = bench.WORKLOADS[name]
    B = wl["batch"]
    mod = bench.load_product()
    frames = np.stack([mod.synth_frame(bench.SEED, k, wl["rows"], wl["cols"], wl["n_rect"], wl["rmin"], wl["rmax"]) for k in range(B)])
    pin = mod.pinned_array(frames.shape)
    pin[...] = frames
    lanes = []
    for _ in range(n_lanes):
        c = mod.Context(0)
        d = mod.FeatureDetector(mod.DetectorConfig(20, 12, 1, wl["window"], 31, 256), c)
        m = mod.FeatureMatcher(mod.MatcherConfig(0, 0, 20, 1, 0.5), c)
        lanes.append((c, mod.Batch(d, wl["rows"], wl["cols"], B, wl["max_cand"], wl["max_kp"]), m))
    for c, b, m in lanes:  # warm up
        b.upload_async(pin)
        b.detect_and_compute(B)
        b.match_consecutive(m, B, use_keypoints=True, first_train_frame=B - 1)
        c.synchronize()
    T = {"upload": [], "detect": [], "match": [], "sync": []}
    t_all = time.perf_counter()
    for i in range(8):
        c, b, m = lanes[i % n_lanes]
        t0 = time.perf_counter()
        if i >= n_lanes:
            c.synchronize()
        t1 = time.perf_counter()
        b.upload_async(pin)
        t2 = time.perf_counter()
        b.detect_and_compute(B)
        t3 = time.perf_counter()
        b.match_consecutive(m, B, use_keypoints=True, first_train_frame=B - 1)
        t4 = time.perf_counter()
        T["sync"].append(t1 - t0); T["upload"].append(t2 - t1); T["detect"].append(t3 - t2); T["match"].append(t4 - t3)
    for c, _, _ in lanes:
        c.synchronize()
    print("total %.2f ms for 8 batches" % ((time.perf_counter() - t_all) * 1e3))
    for k, v in T.items():
        print("%-8s host ms per call: %s" % (k, " ".join("%.2f" % (x * 1e3) for x in v)))


if __name__ == "__main__":
    main()
