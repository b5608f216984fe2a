#!/usr/bin/env python3
"""Differential fuzz of the image pyramid (slamx_pyramid_*: integer bilinear downscale, the unchanged detector on every
level, results merged into level 0's arrays) against the CPU restatement: random frame shapes, level counts (1..8), scale
factors (1.05..2.5), thresholds, windows and image statistics, two or three frames per batch.  Test infrastructure.
usage: tests/fuzz/fuzz_pyramid.py [cases] [seed]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
import oracle_binding
from fuzz_detector import image


def main(cases=None, seed=None):
    """cases / seed: from the command line when None (tests/test_fuzz_smoke_gpu.py passes fixed ones)."""
    if cases is None:
        cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    if seed is None:
        seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    slamx = importlib.import_module("slam-cin0051_amd")
    oracle = oracle_binding.load()
    rng = np.random.default_rng(seed)
    done = 0
    for case in range(cases):
        rows = int(rng.integers(24, 200))
        cols = int(rng.choice([rng.integers(24, 120), rng.integers(200, 330), rng.integers(480, 700)]))
        levels = int(rng.integers(1, 9))
        scale = float(rng.choice([1.2, rng.uniform(1.05, 1.5), rng.uniform(1.5, 2.5)]))
        if min(rows, cols) / scale ** (levels - 1) < 9:  # the library refuses levels smaller than 8 pixels
            levels = max(1, int(np.log(min(rows, cols) / 9.0) / np.log(scale)) + 1)
        t = int(rng.choice([rng.integers(2, 12), rng.integers(12, 64)]))
        window = int(rng.integers(2, 13))
        arc = int(rng.integers(5, 13))
        n = int(rng.integers(2, 4))
        frames = np.stack([np.ascontiguousarray(image(rng, rows, cols).astype(np.uint8)) for _ in range(n)])
        cfg = slamx.DetectorConfig(t, arc, 1, window, 31, 256)
        det = slamx.FeatureDetector(cfg)
        P = oracle.det_params(cfg)
        refs = [oracle.pyramid_detect_and_compute(frames[k], P, levels, scale) for k in range(n)]
        if max(len(r[0]) for r in refs) > 20000:
            det.close() if hasattr(det, "close") else None
            continue
        tag = "case %d: %dx%d, %d levels, scale %.3f, t=%d arc=%d window=%d" % (case, rows, cols, levels, scale, t, arc, window)
        pyr = slamx.Pyramid(det, rows, cols, n, levels, scale, max_candidates=rows * cols, max_keypoints=max(1024, rows * cols))  # (an upper level may hold more keypoints per pixel than level 0:
        # its capacity is the linear share of this one; a tight figure is refused loudly, which is not what is fuzzed here)
        b = pyr.level(0)
        b.upload(frames)
        pyr.detect_and_compute(n)
        for k in range(n):
            k_got, d_got = b.download(k)
            k_ref, d_ref = refs[k]
            assert len(k_got) == len(k_ref) and k_got.tobytes() == k_ref.tobytes(), "keypoints differ, %s, frame %d (%d vs %d)" % (tag, k, len(k_got), len(k_ref))
            assert len(k_ref) == 0 or np.array_equal(d_got, d_ref), "descriptors differ, %s, frame %d" % (tag, k)
        done += 1
        del pyr
        det.close() if hasattr(det, "close") else None
    print("fuzz_pyramid: %d cases identical (seed %d)" % (done, seed))


if __name__ == "__main__":
    main()
