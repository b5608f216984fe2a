#!/usr/bin/env python3
"""Times the match stage alone on a bench workload's frames (detect + describe once, then N match calls).
    python tools/time_match.py <workload> [batch] [reps] [n_rect]      (n_rect: another corner density for the same frame size)
The matcher's environment switch applies: SLAMX_MATCH_MFMA_MIN_KEYPOINTS (0: the matrix-core scan for every
pair, 100000: the popcount scan).  A library built with FLAGS_match=-DSLAMX_MATCH_STATS also reports the slow-path volume,
one built with FLAGS_match=-DSLAMX_MATCH_CLOCKS where a wave spends its time."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def main():
    name = sys.argv[1]
    wl = bench.WORKLOADS[name]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else wl["batch"]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    if len(sys.argv) > 4:
        wl = dict(wl, n_rect=int(sys.argv[4]))
    mod = bench.load_product()
    ctx = mod.Context(0)
    det = mod.FeatureDetector(mod.DetectorConfig(20, 12, 1, wl["window"], 31, 256), ctx)
    m = mod.FeatureMatcher(mod.MatcherConfig(0, 0, 20, 1, 0.5), ctx)
    frames = np.stack([mod.synth_frame(bench.SEED, k, wl["rows"], wl["cols"], wl["n_rect"], wl["rmin"], wl["rmax"]) for k in range(B)])
    if "levels" in wl:  # the pyramid workload: merged keypoints of all levels in level 0's batch
        pyr = mod.Pyramid(det, wl["rows"], wl["cols"], B, wl["levels"], wl["scale"], wl["max_cand"], wl["max_kp"])
        batch = pyr.level(0)
        batch.upload(frames)
        pyr.detect_and_compute(B)
    else:
        batch = mod.Batch(det, wl["rows"], wl["cols"], B, wl["max_cand"], wl["max_kp"])
        batch.upload(frames)
        batch.detect_and_compute(B)
    for _ in range(2):
        batch.match_consecutive(m, B, use_keypoints=True, first_train_frame=B - 1)
    ctx.synchronize()
    ts = []
    for _ in range(reps):
        ctx.timer_start()
        batch.match_consecutive(m, B, use_keypoints=True, first_train_frame=B - 1)
        ts.append(ctx.timer_stop_ms())
    nc, nk, nm = batch.counts(B)
    print("%s B=%d keypoints/frame %.0f matches/frame %.0f: match stage %.3f ms (min %.3f) [MFMA_MIN=%s]" % (
        name, B, nk.mean(), nm.mean(), float(np.median(ts)), min(ts), os.environ.get("SLAMX_MATCH_MFMA_MIN_KEYPOINTS")))
    lib = mod.load_library()
    if hasattr(lib, "slamx_debug_match_stats"):  # development build (-DSLAMX_MATCH_STATS)
        import ctypes

        out = (ctypes.c_ulonglong * 8)()
        lib.slamx_debug_match_stats(out)
        batch.match_consecutive(m, B, use_keypoints=True, first_train_frame=B - 1)
        lib.slamx_debug_match_stats(out)
        w = max(out[0], 1)
        print("spatial scan: %d waves, per query: %.1f tile entries / 4, %.1f pairs queued, %.1f reached the keys; %.1f stages per wave"
              % (out[0], out[1] / w / 64 * 4, out[2] / w / 64, out[3] / w / 64, out[4] / w))
        print("per wave: %.1f tile entries, %.1f queue rounds, %.1f stage-kind flips, %.1f settle calls" % (
            out[1] / w, out[5] / w, out[6] / w, out[7] / w))
    if hasattr(lib, "slamx_debug_match_clocks"):  # development build (-DSLAMX_MATCH_CLOCKS)
        import ctypes

        out = (ctypes.c_ulonglong * 8)()
        lib.slamx_debug_match_clocks(out)
        batch.match_consecutive(m, B, use_keypoints=True, first_train_frame=B - 1)
        lib.slamx_debug_match_clocks(out)
        w = max(out[0], 1)
        us = [out[i] / w / 100.0 for i in range(8)]  # s_memtime: 100 MHz
        print("spatial scan, per wave (%d waves): setup %.1f us, stage order %.1f, startup tiles %.1f (slow path %.1f), steady scan %.1f "
              "(slow path %.1f), drain %.1f" % (out[0], us[1], us[2], us[3], us[7], us[4], us[5], us[6]))
    print("checksum", int(nm.sum()), int(batch.download_matches(m, B - 1)["trainIdx"].astype(np.int64).sum()))


if __name__ == "__main__":
    main()
