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


def _run(args):
    p = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode()


def test_feature_detector_program(slamx, oracle, ref_data, tmp_path):
    img_path = os.path.join(ref_data, "images", "0000000000.png")
    out = str(tmp_path / "det.bin")
    text = _run([os.path.join(BIN, "test_feature_detector"), os.path.join(ref_data, "feature_detector.yml"), img_path, out])
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


def test_cli_estimates_the_pose_between_kitti_frames(ref_data):
    """camera.yml present -> PoseEstimator::estimate + triangulatePoints on the matches (the two KITTI
    frames are 0.1 s apart on a forward-driving car: small rotation, translation mostly along z)."""
    text = _run([os.path.join(BIN, "cli"), "-c", ref_data, "-v", os.path.join(ref_data, "images")])
    pose = [l for l in text.splitlines() if l.strip().startswith("pose:")]
    assert len(pose) == 1, text
    t = [float(x) for x in pose[0].split("(")[1].split(")")[0].split()]
    trace = float(pose[0].split("trace(R) =")[1].split(",")[0])
    assert trace > 2.99 and abs(t[2]) > 0.9


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
