"""bench.py on the GPU box: the N > 1 code path of main() rehearsed on ONE card (two ranks share device 0, gloo
for the collectives -- RCCL refuses two ranks on one device, and 8-GPU runs are the driver's), plus the shape of
the default JSON line."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(argv, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=timeout)
    lines = [json.loads(l) for l in p.stdout.decode().splitlines() if l.startswith("{")]
    return p.returncode, lines, p.stderr.decode()


def test_two_ranks_on_one_card_run_the_real_step():
    """Both ranks create their own context on device 0 next to torch, run the real detect / describe / match step on
    their own stream (seed + rank), triangulate their newest pair and gather the map points: everything of the
    multi-GPU path except RCCL itself."""
    rc, lines, err = _bench(["--gpus", "2", "--backend", "gloo", "--share-device", "--workload", "c2", "--batch", "64",
                             "--steps", "3", "--warmup", "1"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["frames_per_step_per_gpu"] == 64
    assert d["value"] == pytest.approx(2 * 64 * 3 / (d["ms_per_step"] * 3e-3), rel=1e-6)  # whole-job aggregate
    assert d["config"]["keypoints_per_frame"] > 500 and d["map_points_gathered"] > 0
    assert "cpu_baseline" not in d  # rank 0 at N == 1 only


def test_single_rank_line_has_the_contract_keys():
    rc, lines, err = _bench(["--workload", "c2", "--batch", "64", "--steps", "4", "--warmup", "1", "--no-extras",
                             "--no-cpu-baseline", "--no-pcie"])
    assert rc == 0, err[-2000:]
    d = lines[-1]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"]
    assert d["n_gpus"] == 1 and d["dtype"] == "u8" and d["vs_baseline"] is None


def test_pose_stage_inside_the_timed_step():
    """--with-pose: BASELINE.json configs[2]'s second half (essential-matrix RANSAC, simpleRecoverPose, triangulatePoints,
    PnP RANSAC for every frame pair) runs inside the step and is reported per kernel, with its own ceilings."""
    rc, lines, err = _bench(["--workload", "c2", "--batch", "64", "--steps", "4", "--warmup", "1", "--no-extras",
                             "--no-cpu-baseline", "--no-pcie", "--with-pose", "--pose-iterations", "64", "--pnp-iterations", "32"])
    assert rc == 0, err[-2000:]
    d = lines[-1]
    p = d["pose"]
    assert p["hypotheses_per_pair"] == 64 and p["pnp_hypotheses_per_frame"] == 32 and p["pairs_with_a_pose"] == 64
    assert p["matches_per_pair"] > 300 and p["essential_inliers_per_pair"] > 0.5 * p["matches_per_pair"]
    assert 0 < p["ms_per_step"] < d["ms_per_step"] and set(p["stage_ms_per_step"]) >= {"hypotheses", "score", "refit", "cheirality", "pnp_solve"}
    for r in p["rooflines"].values():
        assert 0 < r["frac"] <= 1.0
    for r in d["kernel_rooflines"].values():
        assert 0 < r["frac"] <= 1.0


def test_c3_pose_leg_runs_on_a_scene_with_depth():
    """the pose stage of `--workload c3 --with-pose` (the driver line's other_workloads.c3_pose) is timed
    on keypoint / match arrays of a scene WITH DEPTH at c3's size (10,000 keypoints per frame, a fifth of the matches
    wrong) -- not on the c3 frames' own match lists, where E is degenerate and every PnP hypothesis fails: every pair has
    a pose, the essential step finds the correct matches, the (calibrated) PnP step more than half of its correspondences,
    the recovered motion is the generating one; and the reference-form PnP hypotheses, measured beside it, score nothing
    (the reference's behaviour on real geometry).  The same arrays are compared with the oracle pair by pair in
    tests/test_pose_batch_gpu.py::test_bench_pose_leg_against_the_oracle."""
    rc, lines, err = _bench(["--workload", "c3", "--batch", "16", "--steps", "2", "--warmup", "1", "--no-extras",
                             "--no-cpu-baseline", "--no-pcie", "--with-pose"])
    assert rc == 0, err[-2000:]
    d = lines[-1]
    p = d["pose"]
    assert "scene with depth" in d["config"]["pose_input"]
    assert p["pairs_with_a_pose"] == 16 and p["all_pairs_have_a_pose"] and p["pnp_frames_skipped"] == 0
    assert 8000 < p["matches_per_pair"] < 9000
    assert 0.80 * p["matches_per_pair"] < p["essential_inliers_per_pair"] < 0.90 * p["matches_per_pair"]
    assert p["geometry_mode"] == "calibrated"
    assert p["pnp_inliers_per_frame"] > 0.5 * p["pnp_correspondences_per_frame"] > 2000
    m = p["recovered_motion"]
    # (what a 1 px Sampson threshold leaves of 0.02 px keypoints: the handful of wrong matches within the threshold of the
    # epipolar lines weigh as much in the refit as all correct ones together; and none of the translations is mirrored)
    assert m["essential"]["pairs"] == 16 and m["essential"]["rotation_error_deg_max"] < 1.0
    assert m["essential"]["translation_direction_error_deg_median"] < 3.0 and m["essential"]["translation_direction_error_deg_max"] < 45.0
    assert m["pnp"]["frames"] >= 14 and m["pnp"]["rotation_error_deg_median"] < 0.5 and m["pnp"]["translation_relative_error_median"] < 0.1
    o = p["other_geometry_mode"]
    assert o["geometry_mode"] == "reference" and o["pnp_inlier_share_of_correspondences"] < 0.01
    assert p["stage_ms_per_step"]["pnp_score"] > 0 and p["stage_ms_per_step"]["score"] > 0
    assert d["roofline_over_1"] == {}
