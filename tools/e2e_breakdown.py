#!/usr/bin/env python3
"""Where the end-to-end leg of bench.py (pcie_inclusive) loses link rate: the same rotating lanes with the step cut short.
    python tools/e2e_breakdown.py [workload] [lanes] [steps]"""
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
    name = sys.argv[1] if len(sys.argv) > 1 else "1080p_2k"
    n_lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
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
    outs = [lane[1].packed_buffer(B) for lane in lanes] if os.environ.get("E2E_NO_OUTS") is None else [np.zeros(64, np.uint8)]
    modes = ("upload", "upload+detect", "upload+detect+match", "upload+detect+match+download", "detect+match (frames resident)")
    if os.environ.get("E2E_NO_OUTS") is not None:
        modes = ("upload+detect+match", "upload", "upload+detect+match")
    for what in modes:
        def step(i, guess):
            c, b, m = lanes[i % n_lanes]
            if what.startswith("upload"):
                b.upload_async(pin)
            if "detect" in what:
                b.detect_and_compute(B)
            if "match" in what:
                b.match_consecutive(m, B, use_keypoints=True, first_train_frame=B - 1)
            if "download" in what:
                b.download_packed_async(B, outs[i % n_lanes], guess)
        for i in range(n_lanes):
            step(i, outs[0].nbytes)
        for lane in lanes:
            lane[0].synchronize()
        guess = outs[0].nbytes
        if "download" in what:
            guess = int(int(mod.packed_header(outs[0])["total_bytes"]) * 1.1) + 4096
        t0 = time.perf_counter()
        for i in range(steps):
            if i >= n_lanes:
                lanes[i % n_lanes][0].synchronize()
            step(i, guess)
        for lane in lanes:
            lane[0].synchronize()
        dt = time.perf_counter() - t0
        print("%-36s %2d lanes: %7.0f frames/s  %6.2f ms per batch of %d  %5.1f GB/s of frames" %
              (what, n_lanes, steps * B / dt, dt * 1e3 / steps, B, steps * frames.nbytes / 1e9 / dt), flush=True)


if __name__ == "__main__":
    main()
