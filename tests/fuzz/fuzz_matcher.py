#!/usr/bin/env python3
"""Differential fuzz of the matcher against the CPU restatement (oracle/): random set sizes, descriptor widths and
live-bit prefixes, clusters of duplicates and near-duplicates (ties in best and second), integer / fractional /
far-apart keypoints (jump penalty), every filter / ratio configuration, both all-pairs scans (xor + popcount and
the int8 matrix cores).  Test infrastructure (uses oracle/ as the checker).
usage: tests/fuzz/fuzz_matcher.py [cases] [seed]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding


def descriptors(rng, n, width, bits, pool):
    d = np.zeros((n, width), np.uint8)
    k = min(width, 8)
    raw = rng.integers(0, 256, (n, k), dtype=np.uint8)
    if bits < 8 * k:
        mask = np.frombuffer(np.uint64((1 << bits) - 1).tobytes(), np.uint8)[:k]
        raw &= mask
    d[:, :k] = raw
    if width > 8 and bits >= 64:  # full-width descriptors: random tails as well
        d[:, 8:] = rng.integers(0, 256, (n, width - 8), dtype=np.uint8)
    if pool is not None and len(pool):  # copies and one-bit neighbours of descriptors of the other set
        m = rng.integers(0, n // 2 + 2)
        idx = rng.integers(0, n, m)
        d[idx] = pool[rng.integers(0, len(pool), m)]
        flip = idx[: m // 2]
        if bits >= 1 and len(flip):
            d[flip, 0] ^= (np.uint8(1) << rng.integers(0, min(bits, 8), len(flip)).astype(np.uint8))
    return d


def keypoints(rng, slamx, n, kind):
    k = np.zeros(n, slamx.KEYPOINT_DTYPE)
    if kind == 0:  # integer coordinates in a 1080p frame
        k["x"], k["y"] = rng.integers(0, 1920, n), rng.integers(0, 1080, n)
    elif kind == 1:  # fractional, some far outside
        k["x"], k["y"] = rng.random(n) * 4000 - 1000, rng.random(n) * 3000 - 1000
    elif kind == 2:  # a tight cluster: nobody is farther than the jump radius
        k["x"], k["y"] = 900 + rng.random(n) * 300, 500 + rng.random(n) * 300
    else:  # clumps on a 4k canvas with a few NaN coordinates (no penalty in the reference: NaN > 500 is false)
        c = rng.integers(0, 6, n)
        k["x"], k["y"] = (c % 3) * 1500 + rng.random(n) * 200, (c // 3) * 1500 + rng.random(n) * 200
        bad = rng.integers(0, n, max(1, n // 50))
        k["x"][bad] = np.nan
    return k


def main(cases=None, seed=None):
    """cases / seed: from the command line when None (tests/test_fuzz_smoke_gpu.py passes fixed ones)."""
    if cases is None:
        cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    if seed is None:
        seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    slamx = importlib.import_module("slam-cin0051_amd")
    oracle = oracle_binding.load()
    rng = np.random.default_rng(seed)
    for case in range(cases):
        sizes = lambda: [rng.integers(1, 40), rng.integers(40, 700), rng.integers(700, 3000), rng.integers(700, 3000),
                         rng.integers(3000, 7000)]  # (the last class: the matrix-core scan by the default rule)
        n1 = int(rng.choice(sizes()))
        n2 = int(rng.choice(sizes()))
        width = int(rng.choice([1, 3, 4, 8, 12, 16, 20, 32, 33, 64, 96, 128]))
        bits = int(rng.choice([rng.integers(1, 16), 31, 32, 33, 46, 62, 63, 64]))
        bits = min(bits, 8 * min(width, 8))
        d1 = descriptors(rng, n1, width, bits, None)
        d2 = descriptors(rng, n2, width, bits, d1)
        kind = int(rng.integers(0, 5))  # 4 = no keypoints
        k1 = keypoints(rng, slamx, n1, kind) if kind < 4 else None
        k2 = keypoints(rng, slamx, n2, kind) if kind < 4 else None
        filt, ratio_on = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        good = int(rng.choice([1, 3, 20, 5000]))
        ratio = float(rng.choice([0.0, 0.5, 0.7, 0.9, 1.0]))
        if rng.integers(0, 2):
            os.environ["SLAMX_MATCH_MFMA_MIN_KEYPOINTS"] = "0"  # the matrix-core scan for every size it supports
        else:
            os.environ.pop("SLAMX_MATCH_MFMA_MIN_KEYPOINTS", None)
        m = slamx.FeatureMatcher(slamx.MatcherConfig(0, filt, good, ratio_on, ratio))
        got = m.match(d1, d2, k1, k2)
        ref = oracle.match(d1, d2, k1, k2, oracle.match_params(m.cfg))
        tag = "case %d: %d x %d, width %d, %d live bits, keypoints %d, filter %d/%d, ratio %d/%.1f, mfma override %s" % (
            case, n1, n2, width, bits, kind, filt, good, ratio_on, ratio, os.environ.get("SLAMX_MATCH_MFMA_MIN_KEYPOINTS"))
        assert len(got) == len(ref) and np.array_equal(got.view(np.uint8), ref.view(np.uint8)), "matches differ, " + tag
    os.environ.pop("SLAMX_MATCH_MFMA_MIN_KEYPOINTS", None)
    print("fuzz_matcher: %d cases identical (seed %d)" % (cases, seed))


if __name__ == "__main__":
    main()
