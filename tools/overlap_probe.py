#!/usr/bin/env python3
"""How much of the matcher's time hides beside the next batch's detector: two contexts (two HIP streams) on one card, batch A
runs detect+describe while batch B runs the match of its (already described) frames.
    python tools/overlap_probe.py [workload] [reps]"""
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
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    wl = bench.WORKLOADS[name]
    B = wl["batch"]
    mod = bench.load_product()
    frames = np.stack([mod.synth_frame(bench.SEED, k, wl["rows"], wl["cols"], wl["n_rect"], wl["rmin"], wl["rmax"]) for k in range(B)])
    lanes = []
    for _ in range(2):
        c = mod.Context(0)
        d = mod.FeatureDetector(mod.DetectorConfig(20, 12, 1, wl["window"], 31, 256), c)
        m = mod.FeatureMatcher(mod.MatcherConfig(0, 0, 20, 1, 0.5), c)
        b = mod.Batch(d, wl["rows"], wl["cols"], B, wl["max_cand"], wl["max_kp"])
        b.upload(frames)
        b.detect_and_compute(B)
        b.match_consecutive(m, B, use_keypoints=True, first_train_frame=B - 1)
        c.synchronize()
        lanes.append((c, b, m))

    def detect(i):
        lanes[i][1].detect_and_compute(B)

    def match(i):
        lanes[i][1].match_consecutive(lanes[i][2], B, use_keypoints=True, first_train_frame=B - 1)

    def timed(fn):
        fn()
        for lane in lanes:
            lane[0].synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        for lane in lanes:
            lane[0].synchronize()
        return (time.perf_counter() - t0) * 1e3 / reps

    t_d = timed(lambda: detect(0))
    t_m = timed(lambda: match(1))
    t_dm = timed(lambda: (detect(0), match(1)))
    t_serial = timed(lambda: (detect(0), match(0)))
    t_dd = timed(lambda: (detect(0), detect(1)))
    print("%s: detect %.2f ms, match %.2f ms, one stream detect+match %.2f ms, detect || match on two streams %.2f ms, "
          "detect || detect %.2f ms (2 x %.2f)" % (name, t_d, t_m, t_serial, t_dm, t_dd, t_d), flush=True)


if __name__ == "__main__":
    main()
