#!/usr/bin/env python3
"""Generates tests/golden/oracle_golden.npz: outputs of the CPU oracle (oracle/slam_oracle.cpp) on the
reference's own input fixtures (gray PNGs + shipped YAML, copied under tests/golden/ref_data).

The reference cannot be built in this image and holds no golden vectors of its own (SURVEY.md
8c), so these vectors pin the ORACLE (libstdc++ GCC 11.4 / glibc 2.35 behaviour inherited through
the same std:: / libm calls the reference makes); they guard against toolchain drift between this
container and the GPU box and against accidental edits of the restatement.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding  # noqa: E402
from conftest import load_gray  # noqa: E402

# all 14 mode-L PNGs of the reference's test data (SURVEY.md 8d "parity inputs")
IMAGES = ["images/%010d.png" % k for k in range(10)] + ["images_test_loop/%d.png" % k for k in range(4)]
# frame pairs the matcher vectors are taken on: consecutive frames, as test_feature_matcher.cpp:17-21 pairs them
PAIRS = [("images/%010d.png" % k, "images/%010d.png" % (k + 1), "kitti%d%d" % (k, k + 1)) for k in range(9)] + \
        [("images_test_loop/%d.png" % k, "images_test_loop/%d.png" % (k + 1), "loop%d%d" % (k, k + 1)) for k in range(3)]


def build():
    o = oracle_binding.load()
    P = oracle_binding.DetectorParams(20, 12, 1, 12, 31, 256)       # test/data/feature_detector.yml
    M = oracle_binding.MatcherParams(1, 20, 1, 0.5)                  # test/data/feature_matcher.yml
    out = {"pattern": o.brief_pattern(31, 256)}
    feats = {}
    for name in IMAGES:
        img = load_gray(os.path.join(HERE, "ref_data", name))
        key = name.replace("/", "_").replace(".png", "")
        cand = o.fast_candidates(img, P)
        kps = o.detect(img, P)
        kps2, desc, mom = o.compute(img, kps, P, want_moments=True)
        out[key + "_cand_xys"] = np.stack([cand["x"], cand["y"], cand["response"]], 1).astype(np.uint16)
        out[key + "_kps"] = kps2.view(np.uint8).reshape(len(kps2), 20)
        out[key + "_desc"] = desc
        out[key + "_moments"] = mom.astype(np.int32)
        out[key + "_blur_sum"] = np.array([int(o.gaussian_blur(img).astype(np.uint64).sum())], np.uint64)
        feats[name] = (kps2, desc)
    for a, b, tag in PAIRS:
        (k1, d1), (k2, d2) = feats[a], feats[b]
        out[tag + "_match_kp"] = o.match(d1, d2, k1, k2, M).view(np.uint8).reshape(-1, 12)       # test_feature_matcher.cpp:62
        out[tag + "_match_nokp"] = o.match(d1, d2, None, None, M).view(np.uint8).reshape(-1, 12)  # test_pose_estimator.cpp:138
        M0 = oracle_binding.MatcherParams(0, 20, 1, 0.5)
        out[tag + "_best_kp"] = o.match(d1, d2, k1, k2, M0).view(np.uint8).reshape(-1, 12)
    # ---- build-side extensions (no reference counterpart): the oracle's own outputs, to catch drift / accidental edits
    img = load_gray(os.path.join(HERE, "ref_data", "images_test_loop/0.png"))
    levels = o.pyramid_images(img, 5, 1.2)
    out["pyramid_loop0_level_shapes"] = np.array([l.shape for l in levels], np.int32)
    out["pyramid_loop0_level_sums"] = np.array([int(l.astype(np.uint64).sum()) for l in levels], np.uint64)
    out["pyramid_loop0_level4"] = levels[4]
    pk, pd = o.pyramid_detect_and_compute(img, P, 5, 1.2)
    out["pyramid_loop0_kps"] = pk.view(np.uint8).reshape(len(pk), 20)
    out["pyramid_loop0_desc"] = pd
    return out


def build_gn():
    """Floating-point vectors of the Gauss-Newton oracle (one window, 4 cameras, 30 landmarks): kept apart from the
    exact vectors because they are compared with a tolerance (1e-12 relative)."""
    import gn_scene

    o = oracle_binding.load()
    w = gn_scene.make_window(20261004, 4, 30)
    r = o.gn_step_dense(w["poses0"], w["X0"], w["obs"], w["K"], 1e-3, 2)
    assert r["rc"] == 0
    return {"gn_S": r["S"], "gn_g": r["g"], "gn_delta_c": r["delta_c"], "gn_delta_x": r["delta_x"],
            "gn_cost": np.array([r["cost"]]), "gn_active": r["active"], "gn_obs": w["obs"], "gn_poses0": w["poses0"],
            "gn_X0": w["X0"]}


if __name__ == "__main__":
    data = build()
    path = os.path.join(HERE, "oracle_golden.npz")
    np.savez_compressed(path, **data)
    np.savez_compressed(os.path.join(HERE, "gn_golden.npz"), **build_gn())
    print("wrote", path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in data.items() if "kps" in k})
