"""The C++17 facade (include/slam/...: the reference's class names over the C ABI) exercised by the
counterparts of the reference's own test executables (tools/test_feature_*.cpp), on the same
images and YAML files; results are dumped and compared bit for bit with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_gray

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "tools", "bin")


def _run(args, env=None):
    p = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                       env=dict(os.environ, **env) if env else None)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode()


@pytest.mark.parametrize("env", [None, {"SLAMX_NO_FUSED_DESCRIBE": "1"}, {"SLAMX_DESCRIBE_TILES": "2"}, {"SLAMX_DESCRIBE_TILES": "0"}],
                         ids=["fused-describe", "three-kernel-describe", "region-tile-describe", "per-keypoint-tiles-only"])
def test_feature_detector_program(slamx, oracle, ref_data, tmp_path, env):
    """Every describe path (the fused orientation + BRIEF kernel with per-keypoint tiles, the shared region tiles that
    dense frames take -- forced here for this sparse frame --, and the moments / trig / brief kernels they replace for the
    standard pattern) gives the oracle's keypoints and descriptors."""
    img_path = os.path.join(ref_data, "images", "0000000000.png")
    out = str(tmp_path / "det.bin")
    text = _run([os.path.join(BIN, "test_feature_detector"), os.path.join(ref_data, "feature_detector.yml"), img_path, out], env)
    assert "Detected" in text and "DetectAndCompute" in text
    img = load_gray(img_path)
    P = oracle.det_params(slamx.load_detector_config(os.path.join(ref_data, "feature_detector.yml")))
    k_ref, d_ref = oracle.detect_and_compute(img, P)
    raw = open(out, "rb").read()
    off = 0
    for _ in range(2):  # detect+compute, then detectAndCompute
        (n,) = struct.unpack_from("<i", raw, off)
        off += 4
        kps = np.frombuffer(raw, slamx.KEYPOINT_DTYPE, n, off)
        off += 20 * n
        assert kps.tobytes() == k_ref.tobytes()
    for _ in range(2):
        r, c = struct.unpack_from("<ii", raw, off)
        off += 8
        d = np.frombuffer(raw, np.uint8, r * c, off).reshape(r, c)
        off += r * c
        assert np.array_equal(d, d_ref)
    assert off == len(raw)


def test_feature_matcher_program(slamx, oracle, ref_data, tmp_path):
    a = os.path.join(ref_data, "images", "0000000000.png")
    b = os.path.join(ref_data, "images", "0000000001.png")
    out = str(tmp_path / "match.bin")
    text = _run([os.path.join(BIN, "test_feature_matcher"), os.path.join(ref_data, "feature_detector.yml"),
                 os.path.join(ref_data, "feature_matcher.yml"), a, b, out])
    assert "Found" in text
    P = oracle.det_params(slamx.load_detector_config(os.path.join(ref_data, "feature_detector.yml")))
    M = oracle.match_params(slamx.load_matcher_config(os.path.join(ref_data, "feature_matcher.yml")))
    k1, d1 = oracle.detect_and_compute(load_gray(a), P)
    k2, d2 = oracle.detect_and_compute(load_gray(b), P)
    raw = open(out, "rb").read()
    (ex,) = struct.unpack_from("<i", raw, 0)
    assert ex == 3, "std::invalid_argument / std::runtime_error not raised as the reference does"
    off = 4
    for ref in (oracle.match(d1, d2, k1, k2, M), oracle.match(d1, d2, None, None, M)):
        (n,) = struct.unpack_from("<i", raw, off)
        off += 4
        m = np.frombuffer(raw, slamx.MATCH_DTYPE, n, off)
        off += 12 * n
        assert m.tobytes() == ref.tobytes()


def test_cli_runs_the_pipeline(ref_data):
    text = _run([os.path.join(BIN, "cli"), "-c", ref_data, "-v", os.path.join(ref_data, "images_test_loop")])
    lines = [l for l in text.splitlines() if l.startswith("frame ")]
    assert len(lines) == 4 and "59 keypoints" in lines[0]
    assert _run([os.path.join(BIN, "cli"), "-h"]).startswith("cli -c")


def test_cli_pyramid_option(ref_data, oracle, slamx):
    """-p 4: the facade's detectAndComputePyramid; the keypoint count of the first frame is the oracle's."""
    text = _run([os.path.join(BIN, "cli"), "-c", ref_data, "-v", os.path.join(ref_data, "images_test_loop"), "-p", "4"])
    lines = [l for l in text.splitlines() if l.startswith("frame ")]
    img = load_gray(os.path.join(ref_data, "images_test_loop", "0.png"))
    P = oracle.det_params(slamx.load_detector_config(os.path.join(ref_data, "feature_detector.yml")))
    k_ref, _ = oracle.pyramid_detect_and_compute(img, P, 4, 1.2)
    assert len(lines) == 4 and ("%d keypoints" % len(k_ref)) in lines[0] and len(k_ref) > 59


def test_cli_estimates_the_pose_between_kitti_frames(ref_data):
    """camera.yml present -> PoseEstimator::estimate + triangulatePoints on the matches (the two KITTI
    frames are 0.1 s apart each on a forward-driving car: small rotation, translation mostly along z)."""
    text = _run([os.path.join(BIN, "cli"), "-c", ref_data, "-v", os.path.join(ref_data, "images")])
    pose = [l for l in text.splitlines() if l.strip().startswith("pose:")]
    assert len(pose) == 9, text  # ten frames, nine consecutive pairs
    for line in pose:
        t = [float(x) for x in line.split("(")[1].split(")")[0].split()]
        trace = float(line.split("trace(R) =")[1].split(",")[0])
        assert trace > 2.99 and abs(t[2]) > 0.9, line


def test_preprocessing_program(slamx, oracle, ref_data, tmp_path):
    """Counterpart of the reference's test_preprocessing: Camera::undistortImage on one image, then
    slam::Preprocessor over a frame directory with its timestamps.txt (preprocessor.cpp:24-141)."""
    import calendar
    import shutil
    import time

    frames = tmp_path / "frames"
    frames.mkdir()
    names = ["0000000000.png", "0000000001.png"]
    for n in names:
        shutil.copy(os.path.join(ref_data, "images", n), frames / n)
    stamps = open(os.path.join(ref_data, "timestamps.txt")).read().splitlines()[:2]
    (frames / "timestamps.txt").write_text("\n".join(stamps) + "\n")
    prefix = str(tmp_path / "pre")
    cam_yml = os.path.join(ref_data, "camera.yml")
    text = _run([os.path.join(BIN, "test_preprocessing"), cam_yml, os.path.join(ref_data, "images", names[0]), str(frames), "2",
                 prefix])
    cam = slamx.load_camera(cam_yml)
    lines = [l for l in text.splitlines() if l.startswith("frame ")]
    assert len(lines) == 2
    for k, n in enumerate(names):
        img = load_gray(os.path.join(ref_data, "images", n))
        ref = oracle.undistort(img, cam)
        if k == 0:
            got = np.fromfile(prefix + "_undistorted.f64", np.float64).reshape(img.shape)
            assert np.array_equal(got, ref)
        fields = lines[k].split()
        assert int(fields[2]) == img.shape[0] and int(fields[4]) == img.shape[1]
        assert abs(float(fields[8]) - ref.sum()) < 1e-6 * ref.sum()
        # "2011-09-26 13:02:25.454043904": local-time mktime + nanoseconds, truncated to milliseconds
        main, nano = stamps[k].split(".")
        want_ms = int(time.mktime(time.strptime(main, "%Y-%m-%d %H:%M:%S"))) * 1000 + int(nano) // 1000000
        assert int(fields[6]) == want_ms
    # a third yield is past the end of the stream: the program reports it and fails, as the reference's does
    p = subprocess.run([os.path.join(BIN, "test_preprocessing"), cam_yml, os.path.join(ref_data, "images", names[0]),
                        str(frames), "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"No frames yielded" in p.stderr
    # frame count / timestamp count mismatch is an error (preprocessor.cpp:79-81)
    (frames / "timestamps.txt").write_text(stamps[0] + "\n")
    p = subprocess.run([os.path.join(BIN, "test_preprocessing"), cam_yml, os.path.join(ref_data, "images", names[0]),
                        str(frames), "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"Number of timestamps does not match" in p.stderr


@pytest.mark.parametrize("colour", ["mono", "420jpeg", "444"])
def test_preprocessing_program_on_an_uncompressed_video(slamx, oracle, ref_data, tmp_path, colour):
    """The video branch of slam::Preprocessor (preprocessor.cpp:84-93, :124-131): cv::VideoCapture in the reference; this
    build reads the one container that needs no codec, uncompressed YUV4MPEG2 -- frame k = the k-th luma plane, found by
    seeking as the reference seeks, its position k / frame rate in whole milliseconds -- and refuses every other file with the
    reference's message for a capture that cannot be opened."""
    names = ["0000000000.png", "0000000001.png", "0000000002.png"]
    imgs = [load_gray(os.path.join(ref_data, "images", n)) for n in names]
    rows, cols = imgs[0].shape
    video = tmp_path / "sequence.y4m"
    with open(video, "wb") as f:
        f.write(("YUV4MPEG2 W%d H%d F30000:1001 Ip A1:1 C%s XYSCSS=%s\n" % (cols, rows, colour, colour.upper())).encode())
        for k, img in enumerate(imgs):
            f.write(b"FRAME\n")
            f.write(img.tobytes())
            if colour.startswith("420"):
                f.write(bytes([100 + k]) * (2 * ((rows + 1) // 2) * ((cols + 1) // 2)))
            elif colour == "444":
                f.write(bytes([100 + k]) * (2 * rows * cols))
    cam_yml = os.path.join(ref_data, "camera.yml")
    cam = slamx.load_camera(cam_yml)
    text = _run([os.path.join(BIN, "test_preprocessing"), cam_yml, os.path.join(ref_data, "images", names[0]), str(video), "3"])
    lines = [l for l in text.splitlines() if l.startswith("frame ")]
    assert len(lines) == 3
    for k, img in enumerate(imgs):
        ref = oracle.undistort(img, cam)
        fields = lines[k].split()
        assert int(fields[2]) == rows and int(fields[4]) == cols
        assert abs(float(fields[8]) - ref.sum()) < 1e-9 * ref.sum()
        assert int(fields[6]) == int(1000.0 * k * 1001 / 30000)
    # a fourth yield is past the end; a file that is no YUV4MPEG2 stream cannot be opened, as in the reference without a codec
    p = subprocess.run([os.path.join(BIN, "test_preprocessing"), cam_yml, os.path.join(ref_data, "images", names[0]), str(video), "4"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"No frames yielded" in p.stderr
    p = subprocess.run([os.path.join(BIN, "test_preprocessing"), cam_yml, os.path.join(ref_data, "images", names[0]),
                        os.path.join(ref_data, "images", names[1]), "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"Could not open video file" in p.stderr
    if colour == "mono":
        cut = tmp_path / "cut.y4m"
        cut.write_bytes(open(video, "rb").read()[:-1000])  # the last frame cut short: two whole frames remain
        text = _run([os.path.join(BIN, "test_preprocessing"), cam_yml, os.path.join(ref_data, "images", names[0]), str(cut), "2"])
        assert len([l for l in text.splitlines() if l.startswith("frame ")]) == 2


@pytest.mark.parametrize("min_matches", [20, 4])
def test_loop_closure_program(slamx, ref_data, tmp_path, min_matches):
    """slam::LoopClosure (facade) on the reference's loop-test images, replayed step by step through
    the Python binding: same candidate scores, same verdict, same transform."""
    cfg = open(os.path.join(ref_data, "loop_closure.yml")).read().replace("MinMatchesForPnP: 20", "MinMatchesForPnP: %d" % min_matches)
    cfg = cfg.replace("MinInliersForPnP: 5", "MinInliersForPnP: 4")
    yml = tmp_path / "loop_closure.yml"
    yml.write_text(cfg)
    det_yml = os.path.join(ref_data, "feature_detector.yml")
    mat_yml = os.path.join(ref_data, "feature_matcher.yml")
    cam_yml = os.path.join(ref_data, "camera.yml")
    images = os.path.join(ref_data, "images_test_loop")
    seed = 7
    text = _run([os.path.join(BIN, "test_loop_closure"), str(yml), det_yml, mat_yml, cam_yml, images, "4", str(seed)])
    # replay
    P = slamx.load_loop_closure_config(str(yml))
    det = slamx.FeatureDetector(det_yml)
    mat = slamx.FeatureMatcher(mat_yml)
    cam = slamx.load_camera(cam_yml)
    K = np.array(cam.K).reshape(3, 3)
    frames = [det.detectAndCompute(load_gray(os.path.join(images, "%d.png" % i))) for i in range(4)]
    for i, (k, _) in enumerate(frames):
        assert "keyframe %d: %d keypoints" % (i, len(k)) in text
    kq, dq = frames[3]
    best, max_score, second = -1, 0.0, 0.0
    for i in range(4):
        if abs(3 - i) < P.min_frames_difference:
            continue
        score = len(mat.match(dq, frames[i][1])) / len(dq)
        if score > max_score:
            second, max_score, best = max_score, score, i
        elif score > second:
            second = score
    want = None
    if best != -1 and max_score >= P.min_absolute_score and max_score >= P.relative_score_factor * second:
        kt, dt = frames[best]
        m = mat.match(dq, dt, kq, kt)
        if len(m) >= P.min_matches_for_pnp:
            p2 = np.stack([kq["x"][m["queryIdx"]], kq["y"][m["queryIdx"]]], 1).astype(np.float64)
            p3 = np.stack([kt["x"][m["trainIdx"]], kt["y"][m["trainIdx"]], np.ones(len(m))], 1).astype(np.float64)
            R, t, inl = det.ctx.pnp_ransac(p3, p2, K, P.ransac_max_iterations, P.ransac_reprojection_threshold, seed)
            if inl >= P.min_inliers_for_pnp:
                want = (best, R, t)
    if want is None:
        assert "no loop" in text
    else:
        assert "loop with keyframe %d" % want[0] in text
        rows = [list(map(float, l.split()[1:])) for l in text.splitlines() if l.startswith("T ")]
        T = np.array(rows)
        assert np.abs(T[:3, :3] - want[1]).max() < 1e-9 and np.abs(T[:3, 3] - want[2]).max() < 1e-9
        assert np.array_equal(T[3], [0, 0, 0, 1])


def test_loop_closure_config_checks(slamx, ref_data, tmp_path):
    good = open(os.path.join(ref_data, "loop_closure.yml")).read()
    for old, new, msg in [("MinDbSize: 2", "MinDbSize: -1", "'MinDbSize' must be a non-negative integer."),
                          ("MinFramesDifference: 2", "MinFramesDifference: 0", "'MinFramesDifference' must be a positive integer."),
                          ("MinMatchesForPnP: 20", "MinMatchesForPnP: 3", "'MinMatchesForPnP' must be greater than 3 for PnP."),
                          ("MinInliersForPnP: 5", "MinInliersForPnP: 30", "'MinInliersForPnP' cannot be greater than 'MinMatchesForPnP'."),
                          ("RansacMaxIterations: 100", "RansacMaxIterations: 0", "'RansacMaxIterations' must be a positive integer."),
                          ("RansacReprojectionThreshold: 2.0", "RansacReprojectionThreshold: 0.0",
                           "'RansacReprojectionThreshold' must be a positive value.")]:
        assert old in good
        p = tmp_path / "bad.yml"
        p.write_text(good.replace(old, new))
        with pytest.raises(RuntimeError, match=msg.replace("(", ".").replace(")", ".")):
            slamx.load_loop_closure_config(str(p))
    ok = slamx.load_loop_closure_config(os.path.join(ref_data, "loop_closure.yml"))
    assert (ok.min_db_size, ok.min_frames_difference, ok.min_matches_for_pnp, ok.min_inliers_for_pnp, ok.ransac_max_iterations) == (2, 2, 20, 5, 100)
    assert ok.min_absolute_score == 0.005 and ok.relative_score_factor == 1.5 and ok.ransac_reprojection_threshold == 2.0


def test_backend_program():
    """slam::Map / slam::Backend (declared, never implemented, by the reference): Backend::run() optimises the newest
    window of keyframes on its thread with the GPU Gauss-Newton and writes poses and points back into the map."""
    text = _run([os.path.join(BIN, "test_backend")])
    assert text.startswith("host checks: ok\ncost ")
    assert "failing pass: threw 1, failed() 1, map untouched 1" in text


def test_reference_typed_call_sequences(ref_data):
    """tools/reference_call_sequence.cpp: the reference's own call sequences (test_pose_estimator.cpp:113-177,
    test_loop_closure.cpp:46-79) written with cv::Mat / cv::KeyPoint / cv::DMatch / Eigen::Vector3d against the facade's
    reference-typed overloads; every result is compared bit for bit with the plain-typed overload of the same call."""
    text = _run([os.path.join(BIN, "reference_call_sequence"), ref_data])
    assert "cv-typed == plain-typed: yes" in text and "Eigen-typed == plain-typed: yes" in text
    assert "in front" in text


def test_cli_g1_gathers_its_map_points_over_rccl(ref_data):
    """tools/cli -g 1: the multi-GPU driver (one host thread + context per GPU, frame chunks with a one-frame halo)
    with a world of one: slamx_mapgather_unique_id / _create / _allgather run on the real librccl (ncclCommInitRank,
    two ncclAllGather), everything but a second rank -- which needs a second GPU (the driver's 8-GPU node)."""
    text = _run([os.path.join(BIN, "cli"), "-c", ref_data, "-v", os.path.join(ref_data, "images"), "-g", "1"])
    plain = _run([os.path.join(BIN, "cli"), "-c", ref_data, "-v", os.path.join(ref_data, "images")])
    frames = lambda t: [l for l in t.splitlines() if l.startswith("frame ") or l.strip().startswith("pose:")]
    assert frames(text) == frames(plain)  # same results with and without the gather
    line = [l for l in text.splitlines() if "map points gathered over RCCL" in l]
    assert len(line) == 1 and "1 GPU, map points gathered over RCCL: 180 (per GPU: 180)" in line[0]


def test_cli_refuses_more_gpus_than_there_are(ref_data):
    p = subprocess.run([os.path.join(BIN, "cli"), "-c", ref_data, "-v", os.path.join(ref_data, "images"), "-g", "9"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0 and b"GPU(s) visible" in p.stderr


def test_mapgather_world_of_one(slamx, gpu_ctx):
    """slamx_mapgather_* through the C ABI on one rank: counts and payload come back unchanged, twice (the buffers are
    reused and grow), an empty contribution, and a too small output is reported after the collectives."""
    g = slamx.MapGather(gpu_ctx, slamx.MapGather.unique_id(), 0, 1)
    pts = np.arange(21.0).reshape(7, 3)
    counts, out = g.allgather(pts)
    assert counts.tolist() == [7] and np.array_equal(out, pts)
    big = np.random.default_rng(0).normal(size=(5000, 3))
    counts, out = g.allgather(big)
    assert counts.tolist() == [5000] and np.array_equal(out, big)
    counts, out = g.allgather(np.zeros((0, 3)))
    assert counts.tolist() == [0] and out.shape == (0, 3)
    with pytest.raises(slamx.CapacityError):
        g.allgather(big, cap_points=100)
    # the one-collective form: count and status in the header row of a fixed-capacity block
    with pytest.raises(ValueError):
        g.allgather_fixed(pts)  # no capacity agreed yet
    g.reserve(6000)
    for p in (pts, big, np.zeros((0, 3)), pts[:1]):
        counts, out = g.allgather_fixed(p)
        assert counts.tolist() == [len(p)] and np.array_equal(out, p)
    with pytest.raises(slamx.CapacityError):
        g.allgather_fixed(np.zeros((6001, 3)))  # more than the reserved capacity: refused AFTER the collective ran
    counts, out = g.allgather_fixed(big)  # ... and the communicator is still usable
    assert counts.tolist() == [5000] and np.array_equal(out, big)
    with pytest.raises(slamx.CapacityError):
        g.allgather_fixed(big, cap_points=100)
    g.close()


def test_facade_latency_program(ref_data):
    """tools/facade_latency.cpp (what bench.py reports as `facade_latency`): medians of the reference-signature calls."""
    import json

    imgs = [os.path.join(ref_data, "images", "000000000%d.png" % i) for i in (0, 1)]
    text = _run([os.path.join(BIN, "facade_latency"), os.path.join(ref_data, "feature_detector.yml"),
                 os.path.join(ref_data, "feature_matcher.yml")] + imgs + ["5"])
    d = json.loads(text.strip().splitlines()[-1])
    assert d["rows"] == 512 and d["cols"] == 1392 and d["keypoints"][0] > 1000 and d["matches"] == 20
    for k in ("detect", "compute", "detectAndCompute", "match", "match_without_keypoints"):
        assert 0 < d[k]["min_us"] <= d[k]["median_us"] <= d[k]["max_us"] < 1e6


@pytest.mark.parametrize("folder,repeat", [("images", 1), ("images_test_loop", 3)])
def test_frame_batch_program(ref_data, folder, repeat):
    """slam::FrameBatch (include/slam/frontend/frame_batch.hpp, an extension) hands back, frame by frame, exactly what
    FeatureDetector::detectAndCompute and FeatureMatcher::match return for the same frames (tools/test_frame_batch.cpp
    compares keypoints, descriptors and matches byte for byte; a difference makes it exit with 1)."""
    imgs = sorted(os.path.join(ref_data, folder, n) for n in os.listdir(os.path.join(ref_data, folder)) if n.endswith(".png"))
    text = _run([os.path.join(BIN, "test_frame_batch"), os.path.join(ref_data, "feature_detector.yml"),
                 os.path.join(ref_data, "feature_matcher.yml"), str(repeat)] + imgs)
    assert "batch == frame by frame: yes" in text, text
    assert ("%d frames of" % (len(imgs) * repeat)) in text
    # the pose stage of the same batch through FrameBatch::estimatePoses (camera.yml sits beside the detector's YAML): every
    # pair that has a pose is consistent, and on the KITTI frames (a car driving forward) every pair has one along z
    line = [l for l in text.splitlines() if l.startswith("pose stage")]
    assert len(line) == 1, text
    nums = [int(x) for x in __import__("re").findall(r"(\d+) with a pose, (\d+) consistent .*?, (\d+) with", line[0])[0]]
    assert nums[0] == nums[1] and nums[0] > 0
    if folder == "images":
        assert nums[0] == len(imgs) - 1 and nums[2] == nums[0], line[0]


@pytest.mark.parametrize("batch", [3, 4, 16])
def test_cli_batch_mode_prints_the_frame_by_frame_lines(ref_data, batch):
    """cli -b N: the sequence through slam::FrameBatch N frames at a time (the last frame of a batch recomputed as the
    first of the next) -- keypoint counts, match counts, poses and map-point counts line for line those of the run that
    calls detectAndCompute / match once per frame."""
    args = [os.path.join(BIN, "cli"), "-c", ref_data, "-v", os.path.join(ref_data, "images")]
    want = _run(args).splitlines()
    got = _run(args + ["-b", str(batch)]).splitlines()
    assert len(want) == 10 + 9 + 1 and want[:-1] == got[:-1], "\n".join(got)  # (the last line carries the run time)
    assert want[-1].split(" in ")[0] == got[-1].split(" in ")[0]
