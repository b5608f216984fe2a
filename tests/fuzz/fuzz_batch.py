#!/usr/bin/env python3
"""Differential fuzz of the BATCH path -- what bench.py's headline measures -- against the CPU restatement: 2..9 random
frames of one random shape (noise, blocks, gradients, rectangles; some frames flat, some repeated) through
Batch.detect_and_compute and Batch.match_consecutive with random matcher configurations, with and without keypoints,
first_train_frame = none / the last frame / a middle frame, the matrix-core scan forced or left to the size rule, then the
packed download; every frame's keypoints, descriptors, raw and filtered matches against the oracle's.  Test infrastructure.
usage: tests/fuzz/fuzz_batch.py [cases] [seed]        FUZZ_ONLY=<case> replays one case"""
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
    only = int(os.environ["FUZZ_ONLY"]) if os.environ.get("FUZZ_ONLY") else None
    done = 0
    for case in range(cases) if only is None else [only]:
        rng = np.random.default_rng([seed, case])
        rows = int(rng.integers(24, 160))
        cols = int(rng.choice([rng.integers(24, 120), rng.integers(200, 330), rng.integers(480, 640)]))
        n = int(rng.integers(2, 10))
        t = int(rng.choice([rng.integers(3, 12), rng.integers(12, 50)]))
        window = int(rng.integers(2, 13))
        arc = int(rng.integers(5, 13))
        base = [np.ascontiguousarray(image(rng, rows, cols).astype(np.uint8)) for _ in range(int(rng.integers(1, 4)))]
        frames = []
        for k in range(n):
            pick = int(rng.integers(0, 6))
            if pick == 0:
                frames.append(np.full((rows, cols), int(rng.integers(0, 256)), np.uint8))  # no corner at all
            elif pick == 1 and frames:
                frames.append(frames[-1].copy())                                            # the same frame again: distance 0 everywhere
            else:
                f = base[int(rng.integers(0, len(base)))].astype(np.int32)
                f = np.roll(f, (int(rng.integers(-3, 4)), int(rng.integers(-5, 6))), (0, 1)) + rng.integers(-2, 3, (rows, cols))
                frames.append(np.clip(f, 0, 255).astype(np.uint8))
        frames = np.stack(frames)
        patch, pairs = 31, 256
        if rng.integers(0, 4) == 0:  # other descriptor widths through the batch's matcher (1 ... 128 bytes), other patches
            patch = int(rng.choice([5, 9, 15, 21, 31, 33, 63]))
            pairs = int(rng.choice([8, 24, 64, 128, 256, 512, 1024]))
        cfg = slamx.DetectorConfig(t, arc, 1, window, patch, pairs)
        det = slamx.FeatureDetector(cfg)
        P = oracle.det_params(cfg)
        refs = [oracle.detect_and_compute(frames[k], P) for k in range(n)]
        if max(len(r[0]) for r in refs) > 6000:
            continue
        filt, ratio_on = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        good = int(rng.choice([1, 5, 20, 5000]))
        ratio = float(rng.choice([0.5, 0.7, 0.9, 1.0]))
        use_kp = bool(rng.integers(0, 4) != 0)
        ftf = int(rng.choice([-1, n - 1, rng.integers(0, n)]))
        if rng.integers(0, 2):
            os.environ["SLAMX_MATCH_MFMA_MIN_KEYPOINTS"] = "0"
        else:
            os.environ.pop("SLAMX_MATCH_MFMA_MIN_KEYPOINTS", None)
        m = slamx.FeatureMatcher(slamx.MatcherConfig(0, filt, good, ratio_on, ratio))
        mp = oracle.match_params(m.cfg)
        tag = "case %d: %d frames of %dx%d, t=%d arc=%d window=%d patch=%d pairs=%d, matcher filter %d/%d ratio %d/%.1f, keypoints %d, first_train_frame %d, mfma %s" % (
            case, n, rows, cols, t, arc, window, patch, pairs, filt, good, ratio_on, ratio, use_kp, ftf, os.environ.get("SLAMX_MATCH_MFMA_MIN_KEYPOINTS"))
        b = slamx.Batch(det, rows, cols, n, max_candidates=rows * cols, max_keypoints=max(64, max(len(r[0]) for r in refs) + 8))
        b.upload(frames)
        for rnd in range(2):  # the second run: the host knows the batch's density (hint), the matcher's scratch exists
            b.detect_and_compute(n)
            b.match_consecutive(m, n, use_keypoints=use_kp, first_train_frame=ftf)
            nc, nk, nm = b.counts(n)
            for k in range(n):
                k_ref, d_ref = refs[k]
                k_got, d_got = b.download(k)
                assert len(k_got) == len(k_ref) and k_got.tobytes() == k_ref.tobytes(), "keypoints differ, frame %d run %d, %s" % (k, rnd, tag)
                assert len(k_ref) == 0 or np.array_equal(d_got, d_ref), "descriptors differ, frame %d run %d, %s" % (k, rnd, tag)
                tr = k - 1 if k > 0 else ftf
                got_raw = b.download_matches(m, k, apply_filter=False)
                got_flt = b.download_matches(m, k, apply_filter=True)
                if tr < 0 or len(k_ref) == 0 or len(refs[tr][0]) == 0:
                    assert len(got_raw) == 0 and len(got_flt) == 0, "matches without a train set, frame %d, %s" % (k, tag)
                    continue
                kq, kt = (k_ref, refs[tr][0]) if use_kp else (None, None)
                raw_p = oracle.match_params(slamx.MatcherConfig(0, 0, good, ratio_on, ratio))
                ref_raw = oracle.match(d_ref, refs[tr][1], kq, kt, raw_p)
                ref_flt = oracle.match(d_ref, refs[tr][1], kq, kt, mp)
                assert len(got_flt) == len(ref_flt) and got_flt.tobytes() == ref_flt.tobytes(), "filtered matches differ, frame %d run %d, %s" % (k, rnd, tag)
                assert nm[k] == len(got_raw), "match count, frame %d, %s" % (k, tag)
                assert got_raw.tobytes() == ref_raw.tobytes(), "raw matches differ, frame %d run %d, %s" % (k, rnd, tag)
        done += 1
        del b
    os.environ.pop("SLAMX_MATCH_MFMA_MIN_KEYPOINTS", None)
    print("fuzz_batch: %d cases identical (seed %d)" % (done, seed))


if __name__ == "__main__":
    main()
