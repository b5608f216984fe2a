"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Everything here is integer / byte / index work or float work that the reference defines
operation by operation, so every comparison is exact equality (np.array_equal on the raw
records), never a tolerance.
"""
import os

import numpy as np
import pytest

from conftest import load_gray

pytestmark = pytest.mark.gpu

# all 14 mode-L PNGs of the reference's test data: ten 1392x512 KITTI frames (0000000000.png is the
# reference's test_feature_detector input), four 640x480 loop frames
REF_IMAGES = ["images/%010d.png" % k for k in range(10)] + ["images_test_loop/%d.png" % k for k in range(4)]


def cfg_of(slamx, t=20, arc=12, nms=1, window=12, patch=31, pairs=256):
    return slamx.DetectorConfig(t, arc, nms, window, patch, pairs)


@pytest.fixture(scope="module")
def det(slamx, ref_data):
    return slamx.FeatureDetector(os.path.join(ref_data, "feature_detector.yml"))


@pytest.fixture(scope="module")
def P(slamx, oracle, det):
    return oracle.det_params(det.cfg)


def synth(slamx, k=0, rows=480, cols=640, n_rect=300, seed=0x5EED0051, rmin=16, rmax=64):
    return slamx.synth_frame(seed, k, rows, cols, n_rect, rmin, rmax)


def assert_records_equal(a, b, what):
    assert a.shape == b.shape, "%s: count %s vs %s" % (what, a.shape, b.shape)
    if not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
        bad = np.nonzero([x.tobytes() != y.tobytes() for x, y in zip(a, b)])[0]
        raise AssertionError("%s: %d records differ, first at %d: %s vs %s" % (what, len(bad), bad[0], a[bad[0]], b[bad[0]]))


# ---------------------------------------------------------------- a1
def test_pattern_matches_oracle(det, oracle):
    pat = det.pattern()
    ref = oracle.brief_pattern(31, 256)
    assert pat.shape == (46, 4)
    assert np.array_equal(pat, ref)


# ---------------------------------------------------------------- a2 + score
@pytest.mark.parametrize("name", REF_IMAGES)
def test_fast_candidates_reference_images(det, oracle, P, ref_data, name):
    img = load_gray(os.path.join(ref_data, name))
    assert_records_equal(det.fast_candidates(img), oracle.fast_candidates(img, P), "candidates " + name)


@pytest.mark.parametrize("shape", [(7, 7), (6, 40), (40, 6), (8, 70), (33, 129), (100, 64), (65, 191), (480, 640)])
@pytest.mark.parametrize("arc,t", [(12, 20), (9, 10), (16, 5), (0, 30), (1, 40)])
def test_fast_candidates_shapes_and_thresholds(slamx, oracle, shape, arc, t):
    rows, cols = shape
    rng = np.random.default_rng(rows * 1000 + cols + arc)
    base = synth(slamx, 0, max(rows, 8), max(cols, 8), 40, rmin=3, rmax=12)[:rows, :cols]
    img = np.ascontiguousarray(np.where(rng.random((rows, cols)) < 0.05, rng.integers(0, 256, (rows, cols)), base).astype(np.uint8))
    d = slamx.FeatureDetector(cfg_of(slamx, t=t, arc=arc))
    Pp = oracle.det_params(d.cfg)
    assert_records_equal(d.fast_candidates(img), oracle.fast_candidates(img, Pp), "candidates %s arc=%d t=%d" % (shape, arc, t))


@pytest.mark.parametrize("arc,t", [(16, 8), (12, 30), (14, 20)])
def test_fast_candidates_dense_noise(slamx, oracle, arc, t):
    """Uniform noise: 55-75 % of the pixels survive the cardinal pretest, far more than the per-tile
    survivor list holds, so the tiles take the register fallback of fast_detect_kernel."""
    rng = np.random.default_rng(arc * 100 + t)
    img = rng.integers(0, 256, (200, 700), dtype=np.uint8)
    d = slamx.FeatureDetector(cfg_of(slamx, t=t, arc=arc))
    Pp = oracle.det_params(d.cfg)
    got, ref = d.fast_candidates(img), oracle.fast_candidates(img, Pp)
    assert len(ref) > 2000
    assert_records_equal(got, ref, "dense noise arc=%d t=%d" % (arc, t))


@pytest.mark.parametrize("t,arc", [(0, 9), (1, 2), (2, 3), (3, 5), (4, 7), (5, 8), (7, 10), (62, 11), (63, 13), (64, 15),
                                   (127, 12), (250, 4), (255, 6), (21, 12), (22, 12), (23, 12)])
def test_fast_quick_reject_quantisation(slamx, oracle, t, arc):
    """The quick reject of fast_detect_kernel compares pixels reduced to 6 bits against (t + 1) >> 2 and must
    never drop a pixel the reference's pretest lets through: thresholds at and around the multiples of four, at
    both ends of the range, centre / circle differences of exactly t - 1, t, t + 1, t + 2 at every alignment
    modulo 4, saturated pixels, and arc lengths that take every path of the log-step run test."""
    rng = np.random.default_rng(1000 * t + arc)
    rows, cols = 96, 520  # three tile columns, the last one partial
    base = rng.integers(0, 256, (rows, cols))
    # blocks of constant level with single pixels offset by t-1 .. t+2 in both directions
    level = rng.integers(0, 256, (rows // 8 + 1, cols // 8 + 1)).repeat(8, 0).repeat(8, 1)[:rows, :cols]
    delta = rng.choice([-(t + 2), -(t + 1), -t, -(t - 1), 0, 0, 0, t - 1, t, t + 1, t + 2], (rows, cols))
    flat = np.clip(level + np.where(rng.random((rows, cols)) < 0.4, delta, 0), 0, 255)
    img = np.where(rng.random((rows, cols)) < 0.15, base, flat)
    img[:, :40] = np.where(rng.random((rows, 40)) < 0.5, 255, 0)  # saturated ends
    img = np.ascontiguousarray(img.astype(np.uint8))
    d = slamx.FeatureDetector(cfg_of(slamx, t=t, arc=arc))
    Pp = oracle.det_params(d.cfg)
    got, ref = d.fast_candidates(img), oracle.fast_candidates(img, Pp)
    assert_records_equal(got, ref, "quick reject t=%d arc=%d" % (t, arc))


@pytest.mark.parametrize("shape", [(64, 96), (200, 700)])
def test_more_than_a_quarter_of_the_pixels_are_corners(slamx, oracle, shape):
    """ContiguousPixelsThreshold 0 (accepted by the reference's validator) on noise: every pixel that
    passes the cardinal pretest is a corner, well over the first sizing of the implicit one-frame
    batch (a quarter of the pixels).  The host-buffer API then resizes for "every pixel" and runs
    the frame again instead of failing -- the reference has no such limit."""
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    d = slamx.FeatureDetector(cfg_of(slamx, t=8, arc=0, window=4))
    Pp = oracle.det_params(d.cfg)
    ref = oracle.fast_candidates(img, Pp)
    assert len(ref) > shape[0] * shape[1] // 4
    assert_records_equal(d.fast_candidates(img), ref, "candidates beyond the first sizing")
    if shape[0] * shape[1] < 10000:  # the oracle's greedy loop is O(M^2)
        assert_records_equal(d.detect(img), oracle.detect(img, Pp), "detect beyond the first sizing")
        k_ref, d_ref = oracle.detect_and_compute(img, Pp)
        k_got, d_got = slamx.FeatureDetector(d.cfg).detectAndCompute(img)  # a fresh detector: first sizing again
        assert_records_equal(k_got, k_ref, "detectAndCompute beyond the first sizing")
        assert np.array_equal(d_got, d_ref)


def test_fast_candidates_strided_input(det, oracle, P, slamx):
    big = synth(slamx, 3, 200, 300, 120, rmin=8, rmax=30)
    view = big[10:150, 17:260]  # non-contiguous rows
    assert_records_equal(det.fast_candidates(view), oracle.fast_candidates(np.ascontiguousarray(view), P), "strided")


# ---------------------------------------------------------------- std::sort emulation
def _check_sort(gpu_ctx, oracle, scores):
    """The permutation of the one-launch sort (a workgroup per frame) AND of the two-launch form that launches of few
    frames take (round 5: the top levels in sort_split_kernel, the pieces on 2 / 16 workgroups of sort_pieces_kernel)."""
    scores = np.asarray(scores, np.uint32)
    ref = oracle.sort_desc(scores)
    for groups in (1, 2, 16):
        got = gpu_ctx.debug_sort_desc(scores, groups)
        assert np.array_equal(got, ref), "introsort permutation differs (n=%d, %d group(s)), first at %d" % (
            len(scores), groups, int(np.nonzero(got != ref)[0][0]))


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 18, 31, 32, 33, 63, 64, 65, 66, 100, 127, 128, 129, 130, 200, 255, 256,
                               257, 500, 1000, 1023, 1024, 1025, 4097, 11329, 16384, 16385, 19112, 19116, 19120, 24576, 24577,
                               28671, 28672, 28673, 40000, 57345, 70000])
def test_sort_random_with_ties(gpu_ctx, oracle, n):
    """(sizes around 19114 and 28672: where the partition lists move from LDS to global memory, and where an array stops
    fitting LDS in one piece -- order.hip, SORT_LDS_KEYS_MAX)"""
    rng = np.random.default_rng(n)
    for spread in (3, 40, 1600, 4081):
        _check_sort(gpu_ctx, oracle, rng.integers(0, spread, n))


@pytest.mark.parametrize("n", [17, 64, 65, 129, 1000, 5000, 19500, 28672, 30000])
def test_sort_structured(gpu_ctx, oracle, n):
    i = np.arange(n)
    cases = [
        np.zeros(n), np.full(n, 4080), i % 4096, (n - i) % 4096, (i // 7) % 4096, (i % 2) * 100, (i % 3 == 0) * 7,
        np.minimum(i, n - 1 - i) % 4096, np.maximum(i, n - 1 - i) % 4096, (i * 2654435761 >> 7) % 11,
    ]
    for c in cases:
        _check_sort(gpu_ctx, oracle, c)


def test_sort_depth_limit_heapsort(gpu_ctx, oracle):
    """McIlroy's adversary drives libstdc++'s introsort into its heapsort fallback
    (std::__partial_sort at depth limit 0); the device replays the same heap code."""
    for n in (100, 1000, 4096):
        vals = oracle.antiqsort(n)  # a permutation of 0..n-1 that kills ascending std::sort
        assert sorted(vals.tolist()) == list(range(n))
        _check_sort(gpu_ctx, oracle, (n - 1) - vals)  # same comparison sequence under 'greater'
        _check_sort(gpu_ctx, oracle, vals)
        _check_sort(gpu_ctx, oracle, ((n - 1) - vals) // 3)  # with ties


def test_sort_depth_limit_in_the_global_levels(gpu_ctx, oracle):
    """The same adversary on an array that does not fit LDS (order.hip: 28672 keys): every partition peels off a few
    keys, so the depth limit runs out while the segment is still in global memory and one lane heap-sorts ~30k keys
    there; the sorted segment goes straight to the output, the peeled-off leaves arrive as pieces without levels."""
    for n in (30000, 70000):
        vals = np.minimum(oracle.antiqsort(n), 4095)  # scores have 12 bits; the first ~30 pivots are the smallest values,
        _check_sort(gpu_ctx, oracle, 4095 - vals)      # everything above them compares the same clamped or not
        _check_sort(gpu_ctx, oracle, vals)


def test_sort_kitti_scores(gpu_ctx, oracle, det, P, ref_data):
    img = load_gray(os.path.join(ref_data, "images/0000000000.png"))
    cand = oracle.fast_candidates(img, P)
    _check_sort(gpu_ctx, oracle, cand["response"].astype(np.uint32))


# ---------------------------------------------------------------- a3/a7 detect
@pytest.mark.parametrize("name", REF_IMAGES)
def test_detect_reference_images(det, oracle, P, ref_data, name):
    img = load_gray(os.path.join(ref_data, name))
    assert_records_equal(det.detect(img), oracle.detect(img, P), "detect " + name)


@pytest.mark.parametrize("window", [1, 2, 6, 8, 12, 25, 70])
def test_detect_windows_synthetic(slamx, oracle, window):
    d = slamx.FeatureDetector(cfg_of(slamx, window=window))
    Pp = oracle.det_params(d.cfg)
    for k, (rows, cols, nr) in enumerate([(480, 640, 3000), (240, 333, 900), (128, 1000, 1500)]):
        img = synth(slamx, k, rows, cols, nr, rmin=3, rmax=20)
        assert_records_equal(d.detect(img), oracle.detect(img, Pp), "detect window=%d %dx%d" % (window, rows, cols))


@pytest.mark.parametrize("window", [2, 3, 5, 12, 24])
@pytest.mark.parametrize("arc,t", [(12, 30), (9, 25)])
def test_detect_dense_candidates(slamx, oracle, window, arc, t):
    """Uniform noise: thousands of candidates, crowded everywhere and heavily tied -- the filter waves of
    the greedy NMS run on a grid that changes under them, and early batches fill on every chunk."""
    rng = np.random.default_rng(window * 1000 + arc * 10 + t)
    img = rng.integers(0, 256, (150, 420), dtype=np.uint8)
    d = slamx.FeatureDetector(cfg_of(slamx, t=t, arc=arc, window=window))
    Pp = oracle.det_params(d.cfg)
    assert len(oracle.fast_candidates(img, Pp)) > 1500
    assert_records_equal(d.detect(img), oracle.detect(img, Pp), "dense detect window=%d arc=%d t=%d" % (window, arc, t))


def test_detect_filter_wave_count_follows_the_frame(slamx, oracle):
    """Batches with a candidate capacity of >= 65536 run the greedy NMS with 1 + 7 waves; a frame with >= 32768
    candidates uses all seven filter waves, the others three (nms.hip, NMS_DENSE_CANDIDATES).  One batch with
    frames on both sides of that count, heavily tied noise; the same frames through a small-capacity batch (the
    1 + 3 instantiation) where they fit."""
    rng = np.random.default_rng(20251)
    rows, cols = 420, 640
    noise = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    d = slamx.FeatureDetector(cfg_of(slamx, t=30, arc=9, window=5))
    Pp = oracle.det_params(d.cfg)
    frames = []
    for keep_rows in (rows, 300, 150, 330):  # the rest of the frame is flat: no corners there
        f = np.full((rows, cols), 128, np.uint8)
        f[:keep_rows] = noise[:keep_rows]
        frames.append(f)
    counts = [len(oracle.fast_candidates(f, Pp)) for f in frames]
    b = slamx.Batch(d, rows, cols, len(frames), max_candidates=131072, max_keypoints=16384)
    b.upload(np.stack(frames))
    b.detect_and_compute(len(frames))
    nc, nk, _ = b.counts(len(frames))
    assert nc.tolist() == counts and max(counts) >= 32768 > min(counts)
    for k, f in enumerate(frames):
        want_k, want_d = oracle.detect_and_compute(f, Pp)
        got_k, got_d = b.download(k)
        assert_records_equal(got_k, want_k, "frame %d (%d candidates)" % (k, counts[k]))
        assert np.array_equal(got_d, want_d)
    small = [k for k, c in enumerate(counts) if c <= 30000]
    if small:
        b2 = slamx.Batch(d, rows, cols, len(small), max_candidates=30000, max_keypoints=16384)
        b2.upload(np.stack([frames[k] for k in small]))
        b2.detect_and_compute(len(small))
        for j, k in enumerate(small):
            assert_records_equal(b2.download(j)[0], b.download(k)[0], "frame %d through the small instantiation" % k)


def test_detect_without_nms(slamx, oracle):
    d = slamx.FeatureDetector(cfg_of(slamx, nms=0))
    img = synth(slamx, 1, 300, 400, 1500, rmin=3, rmax=20)
    assert_records_equal(d.detect(img), oracle.detect(img, oracle.det_params(d.cfg)), "detect nms=0")


def test_detect_no_corners(det):
    img = np.full((64, 96), 128, np.uint8)
    assert det.detect(img).shape[0] == 0
    kps, desc = det.detectAndCompute(img)
    assert kps.shape[0] == 0 and desc.shape == (0, 0)  # DescriptorMatrix(0, 0)


# ---------------------------------------------------------------- a4 blur
@pytest.mark.parametrize("name", REF_IMAGES[:1] + REF_IMAGES[10:11])
def test_blur_reference_images(det, oracle, ref_data, name):
    img = load_gray(os.path.join(ref_data, name))
    assert np.array_equal(det.gaussian_blur(img), oracle.gaussian_blur(img))


@pytest.mark.parametrize("shape", [(5, 5), (4, 9), (9, 4), (6, 64), (17, 65), (70, 130), (480, 640)])
def test_blur_shapes(det, oracle, shape):
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(det.gaussian_blur(img), oracle.gaussian_blur(img))


def test_blur_half_integer_candidates(det, oracle):
    """Extreme / two-level images: sums land close to .5 boundaries far more often."""
    rng = np.random.default_rng(5)
    for levels in ((0, 255), (0, 1), (127, 128), (254, 255)):
        img = rng.choice(np.array(levels, np.uint8), size=(256, 512))
        assert np.array_equal(det.gaussian_blur(img), oracle.gaussian_blur(img))


# ---------------------------------------------------------------- libm restatement
def test_trig_integer_moment_grid(gpu_ctx, oracle):
    rng = np.random.default_rng(11)
    n = 400000
    m01 = rng.integers(-1200000, 1200001, n).astype(np.float32)
    m10 = rng.integers(-1200000, 1200001, n).astype(np.float32)
    # axis-aligned and tiny cases (hard part 2: 0 / +-90 / 180 degrees)
    special = np.array([(0, 1), (1, 0), (-1, 0), (0, -1), (1, 1), (0, 0), (-1, -1), (1, -1), (-1, 1), (5, 0), (0, -7),
                        (1, 1000000), (1000000, 1), (-1, 1000000), (1000000, -1), (3, -4), (-0.0, 1), (0, -0.0)], np.float32)
    m01[: len(special)] = special[:, 0]
    m10[: len(special)] = special[:, 1]
    small = rng.integers(-40, 41, (20000, 2)).astype(np.float32)
    m01[1000:21000] = small[:, 0]
    m10[1000:21000] = small[:, 1]
    got = gpu_ctx.debug_trig(m01, m10)
    ref = oracle.trig_chain(m01, m10)
    for g, r, name in zip(got, ref, ("angle", "cos", "sin")):
        bad = np.nonzero(g.view(np.uint32) != r.view(np.uint32))[0]
        assert bad.size == 0, "%s differs from glibc at %d inputs, e.g. m01=%r m10=%r: %r vs %r" % (
            name, bad.size, m01[bad[0]], m10[bad[0]], g[bad[0]], r[bad[0]])


def test_trig_known_answers(gpu_ctx):
    """SURVEY.md 8c known answers for the (m01, m10) -> deg -> cos, sin chain."""
    m01 = np.array([0, 1, -1, 0, 1, 0], np.float32)
    m10 = np.array([1, 0, 0, -1, 1, 0], np.float32)
    deg, c, s = gpu_ctx.debug_trig(m01, m10)
    assert deg.tolist() == [0.0, 90.0, -90.0, 180.0, 45.0, 0.0]
    assert c.view(np.uint32).tolist() == [0x3F800000, 0xB33BBD2E, 0xB33BBD2E, 0xBF800000, 0x3F3504F3, 0x3F800000]
    assert s.view(np.uint32).tolist()[:5] == [0x00000000, 0x3F800000, 0xBF800000, 0xB3BBBD2E, 0x3F3504F3]


# ---------------------------------------------------------------- a5/a6/a7 compute
@pytest.mark.parametrize("name", REF_IMAGES)
def test_compute_reference_images(det, oracle, P, ref_data, name):
    img = load_gray(os.path.join(ref_data, name))
    kps = oracle.detect(img, P)
    k_ref, d_ref = oracle.compute(img, kps, P)
    k_got, d_got = det.compute(img, kps)
    assert_records_equal(k_got, k_ref, "angles " + name)
    assert np.array_equal(d_got, d_ref), "descriptors " + name


def test_compute_border_keypoints(det, oracle, P, slamx):
    """Keypoints 0..25 px from every border: patch-leaves-image zero rows and the
    skipped-pair bit shifting of feature_detector.cpp:268-280."""
    img = synth(slamx, 2, 120, 160, 300, rmin=3, rmax=16)
    pts = [(x, y) for y in list(range(0, 26)) + list(range(94, 120)) for x in (0, 3, 14, 15, 16, 20, 80, 139, 143, 144, 145, 159)]
    pts += [(x, y) for x in list(range(0, 26)) + list(range(134, 160)) for y in (0, 15, 16, 21, 60, 98, 103, 104, 119)]
    kps = np.zeros(len(pts), slamx.KEYPOINT_DTYPE)
    kps["x"] = [p[0] for p in pts]
    kps["y"] = [p[1] for p in pts]
    kps["size"] = 6.0
    k_ref, d_ref = oracle.compute(img, kps, P)
    k_got, d_got = det.compute(img, kps)
    assert_records_equal(k_got, k_ref, "border angles")
    assert np.array_equal(d_got, d_ref)
    assert (d_ref.sum(axis=1) == 0).any() and (d_ref.sum(axis=1) != 0).any()


def test_compute_fractional_coordinates(det, oracle, P, slamx):
    img = synth(slamx, 4, 200, 200, 600, rmin=3, rmax=16)
    rng = np.random.default_rng(3)
    kps = np.zeros(300, slamx.KEYPOINT_DTYPE)
    kps["x"] = rng.uniform(0, 200, 300).astype(np.float32)
    kps["y"] = rng.uniform(0, 200, 300).astype(np.float32)
    kps["angle"] = 123.0  # overwritten by compute
    k_ref, d_ref = oracle.compute(img, kps, P)
    k_got, d_got = det.compute(img, kps)
    assert_records_equal(k_got, k_ref, "fractional angles")
    assert np.array_equal(d_got, d_ref)


@pytest.mark.parametrize("patch,pairs", [(31, 256), (7, 8), (101, 256), (31, 1024)])
def test_compute_unrepresentable_coordinates(slamx, oracle, patch, pairs):
    """Caller-supplied keypoints whose coordinates an int cannot hold (undefined in the reference's
    static_cast<int>): they must behave as "outside the image" -- angle 0, zero descriptor -- and
    never reach a tile load; the ordinary keypoints between them are unaffected."""
    d = slamx.FeatureDetector(cfg_of(slamx, patch=patch, pairs=pairs, window=6))
    Pp = oracle.det_params(d.cfg)
    img = synth(slamx, 6, 240, 320, 1000, rmin=3, rmax=16)
    good = oracle.detect(img, Pp)[:200]
    wild = [3.0e9, -3.0e9, 2147483520.0, 2147483648.0, -2147483648.0, 4294967296.0, 1.0e30, -1.0e30, np.inf, -np.inf,
            np.nan, 2147483648.0 - 15 * 128, 65536.0, 1048577.0, -1.0, -16.0]
    kps = np.zeros(len(good) + 3 * len(wild), slamx.KEYPOINT_DTYPE)
    kps["size"] = 6.0
    kps["angle"] = 77.0
    is_wild = np.zeros(len(kps), bool)
    assert len(good) >= 150
    for j, w in enumerate(wild):  # wild x, wild y, both -- spread between the ordinary keypoints
        for k, (x, y) in enumerate(((w, 100.0), (100.0, w), (w, w))):
            i = 4 * (3 * j + k) + 1
            kps[i]["x"], kps[i]["y"] = x, y
            is_wild[i] = True
    kps[~is_wild] = good[: (~is_wild).sum()]
    k_got, d_got = d.compute(img, kps)
    assert (k_got["angle"][is_wild] == 0.0).all()
    assert not d_got[is_wild].any()
    k_ref, d_ref = oracle.compute(img, kps[~is_wild], Pp)
    assert_records_equal(k_got[~is_wild], k_ref, "ordinary keypoints between wild ones")
    assert np.array_equal(d_got[~is_wild], d_ref)


@pytest.mark.parametrize("patch,pairs", [(31, 256), (15, 64), (7, 8), (51, 512), (31, 1024), (101, 256)])
def test_compute_other_patterns(slamx, oracle, patch, pairs):
    d = slamx.FeatureDetector(cfg_of(slamx, patch=patch, pairs=pairs, window=6))
    Pp = oracle.det_params(d.cfg)
    assert np.array_equal(d.pattern(), oracle.brief_pattern(patch, pairs))
    img = synth(slamx, 5, 240, 320, 1000, rmin=3, rmax=16)
    kps = oracle.detect(img, Pp)
    k_ref, d_ref = oracle.compute(img, kps, Pp)
    k_got, d_got = d.compute(img, kps)
    assert_records_equal(k_got, k_ref, "angles patch=%d" % patch)
    assert np.array_equal(d_got, d_ref)


@pytest.mark.parametrize("name", REF_IMAGES[:1] + REF_IMAGES[10:11])
def test_detect_and_compute(det, oracle, P, ref_data, name):
    img = load_gray(os.path.join(ref_data, name))
    k_ref, d_ref = oracle.detect_and_compute(img, P)
    k_got, d_got = det.detectAndCompute(img)
    assert_records_equal(k_got, k_ref, "detectAndCompute kps")
    assert np.array_equal(d_got, d_ref)


# ---------------------------------------------------------------- a8-a10 match
@pytest.fixture(scope="module")
def kitti_pair(det, oracle, P, ref_data):
    a = load_gray(os.path.join(ref_data, "images/0000000000.png"))
    b = load_gray(os.path.join(ref_data, "images/0000000001.png"))
    return oracle.detect_and_compute(a, P), oracle.detect_and_compute(b, P)


def mcfg(slamx, filt=1, good=20, ratio_on=1, ratio=0.5):
    return slamx.MatcherConfig(0, filt, good, ratio_on, ratio)


@pytest.mark.parametrize("with_kp", [True, False])
@pytest.mark.parametrize("filt,good,ratio_on,ratio", [(1, 20, 1, 0.5), (0, 1, 1, 0.5), (0, 1, 0, 0.0), (1, 5000, 1, 0.9),
                                                       (1, 3, 0, 1.0), (0, 1, 1, 1.0), (0, 1, 1, 0.0)])
def test_match_kitti_pair(slamx, oracle, kitti_pair, with_kp, filt, good, ratio_on, ratio):
    (k1, d1), (k2, d2) = kitti_pair
    m = slamx.FeatureMatcher(mcfg(slamx, filt, good, ratio_on, ratio))
    mp = oracle.match_params(m.cfg)
    got = m.match(d1, d2, k1 if with_kp else None, k2 if with_kp else None)
    ref = oracle.match(d1, d2, k1 if with_kp else None, k2 if with_kp else None, mp)
    assert_records_equal(got, ref, "matches")


def test_match_shipped_config(slamx, oracle, kitti_pair, ref_data):
    """test_feature_matcher.cpp: the shipped YAML, keypoints passed."""
    (k1, d1), (k2, d2) = kitti_pair
    m = slamx.FeatureMatcher(os.path.join(ref_data, "feature_matcher.yml"))
    assert (m.cfg.filter_matches, m.cfg.good_matches_count, m.cfg.use_ratio_test) == (1, 20, 1)
    got = m.match(d1, d2, k1, k2)
    assert_records_equal(got, oracle.match(d1, d2, k1, k2, oracle.match_params(m.cfg)), "shipped matcher")
    assert len(got) <= 20


# widths 20, 28, 36..60 bytes have 5, 7, 9..15 words: the kernel is instantiated for 6, 8 and 16
# and must read the words past the descriptor's own as zero, not from the next descriptor
@pytest.mark.parametrize("width", [1, 3, 4, 8, 12, 16, 20, 24, 28, 32, 33, 36, 44, 48, 60, 64])
@pytest.mark.parametrize("n1,n2", [(1, 1), (1, 300), (300, 1), (257, 255), (700, 513)])
def test_match_random_descriptors(slamx, oracle, width, n1, n2):
    rng = np.random.default_rng(width * 100000 + n1 * 1000 + n2)
    d1 = rng.integers(0, 256, (n1, width), dtype=np.uint8)
    d2 = rng.integers(0, 256, (n2, width), dtype=np.uint8)
    d2[rng.integers(0, n2, n2 // 3 + 1)] = d1[rng.integers(0, n1, n2 // 3 + 1)]  # exact duplicates -> ties
    k1 = np.zeros(n1, slamx.KEYPOINT_DTYPE)
    k2 = np.zeros(n2, slamx.KEYPOINT_DTYPE)
    for k in (k1, k2):
        k["x"] = rng.integers(0, 1920, len(k))
        k["y"] = rng.integers(0, 1080, len(k))
    for kp in (True, False):
        for ratio_on, ratio in ((1, 0.8), (0, 0.5)):
            m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, ratio_on, ratio))
            a1, a2 = (k1, k2) if kp else (None, None)
            assert_records_equal(m.match(d1, d2, a1, a2), oracle.match(d1, d2, a1, a2, oracle.match_params(m.cfg)),
                                 "random match w=%d kp=%s" % (width, kp))


@pytest.fixture
def mfma_scan_for_all_sizes(monkeypatch):
    """The matrix-core scan is only taken for train sets of >= 1024 keypoints (>= 768 when they spread over more than
    1000 px; smaller ones are faster on the popcount scan); the override sends every size through it."""
    monkeypatch.setenv("SLAMX_MATCH_MFMA_MIN_KEYPOINTS", "0")


@pytest.mark.parametrize("bits", [1, 7, 31, 32, 33, 46, 61, 62, 63, 64])
@pytest.mark.parametrize("n1,n2", [(1, 1), (5, 1), (1, 700), (15, 17), (16, 16), (250, 270), (257, 255), (700, 513), (2100, 2300)])
def test_match_live_bit_prefix(slamx, oracle, bits, n1, n2, mfma_scan_for_all_sizes):
    """Descriptors whose bits >= `bits` are zero (what the detector produces: 46 live bits): up to 62 live bits
    the all-pairs distances come from the int8 matrix cores (match_mfma_kernel), above that from xor + popcount.
    Few distinct values on purpose -- ties in best AND second, duplicates, all-zero rows -- with and without the
    jump penalty, with and without the ratio test."""
    rng = np.random.default_rng(bits * 7919 + n1 * 31 + n2)
    width = 32

    def make(n):
        d = np.zeros((n, width), np.uint8)
        raw = rng.integers(0, 256, (n, 8), dtype=np.uint8)
        if bits < 64:
            mask = np.frombuffer(np.uint64((1 << bits) - 1).tobytes(), np.uint8)
            raw &= mask
        d[:, :8] = raw
        return d

    d1, d2 = make(n1), make(n2)
    d2[rng.integers(0, n2, n2 // 3 + 1)] = d1[rng.integers(0, n1, n2 // 3 + 1)]  # exact duplicates -> ties
    d2[rng.integers(0, n2, n2 // 10 + 1)] = 0
    if bits >= 7:  # near-duplicates: many equal small distances
        k = rng.integers(0, n2, n2 // 4 + 1)
        d2[k] = d1[rng.integers(0, n1, len(k))]
        d2[k, 0] ^= np.uint8(1) << rng.integers(0, min(bits, 8), len(k)).astype(np.uint8)
    k1 = np.zeros(n1, slamx.KEYPOINT_DTYPE)
    k2 = np.zeros(n2, slamx.KEYPOINT_DTYPE)
    for k in (k1, k2):
        k["x"] = rng.integers(0, 1920, len(k))
        k["y"] = rng.integers(0, 1080, len(k))
    for kp in (True, False):
        for ratio_on, ratio in ((1, 0.8), (0, 0.5)):
            m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, ratio_on, ratio))
            a1, a2 = (k1, k2) if kp else (None, None)
            assert_records_equal(m.match(d1, d2, a1, a2), oracle.match(d1, d2, a1, a2, oracle.match_params(m.cfg)),
                                 "live bits %d kp=%s ratio=%s" % (bits, kp, ratio_on))


@pytest.mark.parametrize("width,bits", [(4, 20), (4, 32), (8, 50), (8, 62), (16, 46), (64, 46)])
def test_match_matrix_core_scan_descriptor_widths(slamx, oracle, width, bits, mfma_scan_for_all_sizes):
    """One-word and two-word descriptors and long ones whose tail is zero through the matrix-core scan."""
    rng = np.random.default_rng(width * 100 + bits)
    n1, n2 = 300, 1000
    mask = np.frombuffer(np.uint64((1 << bits) - 1).tobytes(), np.uint8)[: min(width, 8)]

    def make(n):
        d = np.zeros((n, width), np.uint8)
        d[:, : min(width, 8)] = rng.integers(0, 256, (n, min(width, 8)), dtype=np.uint8) & mask
        return d

    d1, d2 = make(n1), make(n2)
    d2[rng.integers(0, n2, 300)] = d1[rng.integers(0, n1, 300)]
    k1 = np.zeros(n1, slamx.KEYPOINT_DTYPE)
    k2 = np.zeros(n2, slamx.KEYPOINT_DTYPE)
    for k in (k1, k2):
        k["x"] = rng.integers(0, 1920, len(k))
        k["y"] = rng.integers(0, 1080, len(k))
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 1, 0.9))
    for a1, a2 in ((k1, k2), (None, None)):
        assert_records_equal(m.match(d1, d2, a1, a2), oracle.match(d1, d2, a1, a2, oracle.match_params(m.cfg)),
                             "width %d, %d live bits" % (width, bits))


def test_match_large_train_set_takes_the_matrix_core_scan(slamx, oracle):
    """7000 train descriptors with 46 live bits: above the built-in threshold (1024), no override."""
    rng = np.random.default_rng(77)
    n1, n2 = 900, 7000
    d1 = np.zeros((n1, 32), np.uint8)
    d2 = np.zeros((n2, 32), np.uint8)
    mask = np.frombuffer(np.uint64((1 << 46) - 1).tobytes(), np.uint8)
    d1[:, :8] = rng.integers(0, 256, (n1, 8), dtype=np.uint8) & mask
    d2[:, :8] = rng.integers(0, 256, (n2, 8), dtype=np.uint8) & mask
    d2[rng.integers(0, n2, 3000)] = d1[rng.integers(0, n1, 3000)]
    k1 = np.zeros(n1, slamx.KEYPOINT_DTYPE)
    k2 = np.zeros(n2, slamx.KEYPOINT_DTYPE)
    for k in (k1, k2):
        k["x"] = rng.integers(0, 1920, len(k))
        k["y"] = rng.integers(0, 1080, len(k))
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 1, 0.8))
    assert_records_equal(m.match(d1, d2, k1, k2), oracle.match(d1, d2, k1, k2, oracle.match_params(m.cfg)), "large train set")


@pytest.mark.parametrize("n1,n2", [(300, 65535), (300, 65536), (300, 65537), (200, 90000), (66000, 40)])
def test_match_train_sets_around_the_16_bit_index_limit(slamx, oracle, n1, n2):
    """The matrix-core scan keeps train indices in 16 bits of its pair entries: sets of up to 65536 descriptors take
    it, larger ones the popcount scan; the best match of many queries sits at the very end of the train set."""
    rng = np.random.default_rng(n1 * 7 + n2)
    mask = np.frombuffer(np.uint64((1 << 46) - 1).tobytes(), np.uint8)
    d1 = np.zeros((n1, 32), np.uint8)
    d2 = np.zeros((n2, 32), np.uint8)
    d1[:, :8] = rng.integers(0, 256, (n1, 8), dtype=np.uint8) & mask
    d2[:, :8] = rng.integers(0, 256, (n2, 8), dtype=np.uint8) & mask
    tail = min(n1, n2) // 2
    d2[n2 - tail:] = d1[:tail]                      # exact matches in the last positions
    d2[rng.integers(0, n2, n2 // 50 + 1)] = d1[rng.integers(0, n1, n2 // 50 + 1)]
    k1 = np.zeros(n1, slamx.KEYPOINT_DTYPE)
    k2 = np.zeros(n2, slamx.KEYPOINT_DTYPE)
    for k in (k1, k2):
        k["x"] = rng.integers(0, 1920, len(k))
        k["y"] = rng.integers(0, 1080, len(k))
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 1, 0.9))
    for a1, a2 in ((k1, k2), (None, None)):
        assert_records_equal(m.match(d1, d2, a1, a2), oracle.match(d1, d2, a1, a2, oracle.match_params(m.cfg)),
                             "%d x %d descriptors" % (n1, n2))


def test_batch_matching_through_the_matrix_core_scan(slamx, oracle, mfma_scan_for_all_sizes):
    """A device-resident batch (46 live bits by construction) with the override: frames matched to their
    predecessors, ragged keypoint counts, train sets of several LDS stages."""
    d = slamx.FeatureDetector(cfg_of(slamx, window=6))
    Pp = oracle.det_params(d.cfg)
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 1, 0.7))
    mp = oracle.match_params(m.cfg)
    frames = [synth(slamx, k, 360, 500, 300 + 400 * k, rmin=3, rmax=20) for k in range(4)]
    b = slamx.Batch(d, 360, 500, 4, 32768, 8192)
    b.upload(np.stack(frames))
    b.detect_and_compute(4)
    b.match_consecutive(m, 4, use_keypoints=True, first_train_frame=3)
    ref = [oracle.detect_and_compute(f, Pp) for f in frames]
    for k in range(4):
        t = (k - 1) % 4
        want = oracle.match(ref[k][1], ref[t][1], ref[k][0], ref[t][0], mp)
        assert_records_equal(b.download_matches(m, k), want, "frame %d vs %d" % (k, t))
    b.match_consecutive(m, 4, use_keypoints=False, first_train_frame=-1)
    for k in range(1, 4):
        want = oracle.match(ref[k][1], ref[k - 1][1], None, None, mp)
        assert_records_equal(b.download_matches(m, k), want, "no keypoints, frame %d" % k)


def test_batch_with_both_scans_in_one_launch(slamx, oracle, monkeypatch):
    """Frames of very different keypoint counts in one batch, threshold between them: some frame pairs are taken
    by match_kernel, the others by match_spatial_kernel -- chosen on the device by the size of the pair's train set."""
    monkeypatch.setenv("SLAMX_MATCH_MFMA_MIN_KEYPOINTS", "600")
    d = slamx.FeatureDetector(cfg_of(slamx, window=6))
    Pp = oracle.det_params(d.cfg)
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 1, 0.7))
    mp = oracle.match_params(m.cfg)
    frames = [synth(slamx, k, 360, 500, n, rmin=3, rmax=20) for k, n in enumerate((80, 1500, 120, 1700, 60))]
    b = slamx.Batch(d, 360, 500, 5, 32768, 8192)
    b.upload(np.stack(frames))
    b.detect_and_compute(5)
    b.match_consecutive(m, 5, use_keypoints=True, first_train_frame=4)
    ref = [oracle.detect_and_compute(f, Pp) for f in frames]
    sizes = [len(r[0]) for r in ref]
    assert min(sizes) < 600 < max(sizes)
    for k in range(5):
        t = (k - 1) % 5
        want = oracle.match(ref[k][1], ref[t][1], ref[k][0], ref[t][0], mp)
        assert_records_equal(b.download_matches(m, k), want, "frame %d (%d kp) vs %d (%d kp)" % (k, sizes[k], t, sizes[t]))


def test_batch_scan_choice_by_extent(slamx, oracle):
    """Default thresholds: train sets of 1024 keypoints or more take the matrix-core scan, those of 768..1023 only when
    their keypoints spread over more than 1000 px (decided per frame on the device).  Wide and narrow frames on both
    sides of those counts alternate in one batch, so both scans run in one launch and every pair sees the other kind
    as its partner; all results are the oracle's."""
    d = slamx.FeatureDetector(cfg_of(slamx, window=6))
    Pp = oracle.det_params(d.cfg)
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 1, 0.7))
    mp = oracle.match_params(m.cfg)
    rows, cols = 400, 1500
    frames = []
    for k, (n, narrow) in enumerate(((2200, False), (5200, True), (5500, False), (300, False), (5400, True), (7000, True))):
        f = synth(slamx, k, rows, cols, n, rmin=3, rmax=20)
        if narrow:
            f[:, 600:] = 128  # corners only in the left 600 columns: diagonal of the keypoint box < 1000 px
        frames.append(f)
    F = len(frames)
    b = slamx.Batch(d, rows, cols, F, 65536, 8192)
    b.upload(np.stack(frames))
    b.detect_and_compute(F)
    b.match_consecutive(m, F, use_keypoints=True, first_train_frame=F - 1)
    ref = [oracle.detect_and_compute(f, Pp) for f in frames]
    sizes = [len(r[0]) for r in ref]
    assert all(768 <= sizes[k] < 1024 for k in (0, 1, 4)) and sizes[3] < 768 and sizes[2] >= 1024 and sizes[5] >= 1024, sizes
    assert ref[1][0]["x"].max() < 640 and ref[0][0]["x"].max() > 1400
    for k in range(F):
        t = (k - 1) % F
        want = oracle.match(ref[k][1], ref[t][1], ref[k][0], ref[t][0], mp)
        assert_records_equal(b.download_matches(m, k), want, "frame %d (%d kp) vs %d (%d kp)" % (k, sizes[k], t, sizes[t]))


@pytest.mark.parametrize("n1,n2,wide", [(3300, 3500, True), (1500, 3100, False), (2500, 900, True), (900, 2500, True),
                                         (1200, 1000, False), (1000, 1030, False)])
def test_match_default_scan_rule_through_the_host_api(slamx, oracle, n1, n2, wide):
    """slamx_matcher_match with the DEFAULT thresholds (no environment override): train sets of >= 1024 descriptors, and
    wide ones of >= 768, go through the matrix-core scan in spatial order, the others through xor + popcount -- decided
    on the device.  Descriptors with the detector's 46 live bits, duplicates and near-duplicates, keypoints on a 1080p
    canvas (wide) or in a 600 x 500 corner of it; every result is the oracle's."""
    rng = np.random.default_rng(n1 * 7 + n2)

    def make(n):
        d = np.zeros((n, 32), np.uint8)
        raw = rng.integers(0, 256, (n, 8), dtype=np.uint8)
        raw &= np.frombuffer(np.uint64((1 << 46) - 1).tobytes(), np.uint8)
        d[:, :8] = raw
        return d

    d1, d2 = make(n1), make(n2)
    k = rng.integers(0, n2, n2 // 2)
    d2[k] = d1[rng.integers(0, n1, len(k))]
    d2[k[: len(k) // 2], 1] ^= np.uint8(1) << rng.integers(0, 8, len(k) // 2).astype(np.uint8)
    k1 = np.zeros(n1, slamx.KEYPOINT_DTYPE)
    k2 = np.zeros(n2, slamx.KEYPOINT_DTYPE)
    for kp in (k1, k2):
        kp["x"] = rng.integers(0, 1920 if wide else 600, len(kp))
        kp["y"] = rng.integers(0, 1080 if wide else 500, len(kp))
    for cfg in ((0, 1, 1, 0.8), (1, 20, 1, 0.5), (0, 1, 0, 0.0)):
        m = slamx.FeatureMatcher(mcfg(slamx, *cfg))
        mp = oracle.match_params(m.cfg)
        assert_records_equal(m.match(d1, d2, k1, k2), oracle.match(d1, d2, k1, k2, mp), "with keypoints %s" % (cfg,))
        assert_records_equal(m.match(d1, d2), oracle.match(d1, d2, None, None, mp), "without keypoints %s" % (cfg,))


def test_match_spatial_scan_with_wild_coordinates(slamx, oracle, mfma_scan_for_all_sizes):
    """The spatial scan orders by keypoint position and classes stages by their bounding boxes; the RESULT must not
    depend on either: NaN coordinates (the reference's `distance > 500` is false for NaN: no penalty), coordinates of
    +-3e9, negative ones, everything at one point, and a clean set for comparison.  (Infinite coordinates are left
    out: the reference then converts an infinite float to int, which is undefined behaviour -- INT_MIN on x86.)"""
    rng = np.random.default_rng(77)
    n1, n2 = 700, 900
    d1 = np.zeros((n1, 32), np.uint8)
    d2 = np.zeros((n2, 32), np.uint8)
    d1[:, :5] = rng.integers(0, 256, (n1, 5), dtype=np.uint8)
    d2[:, :5] = rng.integers(0, 256, (n2, 5), dtype=np.uint8)
    d2[rng.integers(0, n2, 300)] = d1[rng.integers(0, n1, 300)]
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 1, 0.9))
    mp = oracle.match_params(m.cfg)

    def kps(n, kind):
        k = np.zeros(n, slamx.KEYPOINT_DTYPE)
        k["x"] = rng.uniform(0, 1920, n).astype(np.float32)
        k["y"] = rng.uniform(0, 1080, n).astype(np.float32)
        if kind == "nan":
            k["x"][rng.integers(0, n, n // 7)] = np.nan
            k["y"][rng.integers(0, n, n // 9)] = np.nan
        elif kind == "huge":
            k["x"][rng.integers(0, n, n // 5)] = 3.0e9
            k["y"][rng.integers(0, n, n // 5)] = -3.0e9
            k["x"][rng.integers(0, n, n // 5)] *= -1.0
        elif kind == "point":
            k["x"] = 7.0
            k["y"] = 7.0
        return k

    with np.errstate(invalid="ignore", over="ignore"):
        for kq, kt in (("nan", "clean"), ("clean", "nan"), ("nan", "nan"), ("huge", "clean"), ("clean", "huge"), ("point", "clean"),
                       ("point", "point"), ("clean", "clean")):
            k1, k2 = kps(n1, kq), kps(n2, kt)
            assert_records_equal(m.match(d1, d2, k1, k2), oracle.match(d1, d2, k1, k2, mp), "query %s, train %s" % (kq, kt))


def test_match_penalty_fractional_keypoints(slamx, oracle):
    rng = np.random.default_rng(9)
    d1 = rng.integers(0, 256, (400, 32), dtype=np.uint8)
    d2 = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    k1 = np.zeros(400, slamx.KEYPOINT_DTYPE)
    k2 = np.zeros(500, slamx.KEYPOINT_DTYPE)
    for k in (k1, k2):
        k["x"] = rng.uniform(0, 4000, len(k)).astype(np.float32)
        k["y"] = rng.uniform(0, 3000, len(k)).astype(np.float32)
    k2["x"][:50] = k1["x"][:50] + 300.0  # exactly 500 px away (3-4-5): boundary of '>'
    k2["y"][:50] = k1["y"][:50] + 400.0
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 0, 0.5))
    assert_records_equal(m.match(d1, d2, k1, k2), oracle.match(d1, d2, k1, k2, oracle.match_params(m.cfg)), "penalty")


def test_penalty_arithmetic_exact(gpu_ctx, oracle):
    """sqrtf, divide, multiply and truncation of feature_matcher.cpp:161-170 on 2M random tuples
    (correctly rounded sqrt/divide are required: a 1-ulp sqrt flips truncations)."""
    rng = np.random.default_rng(21)
    n = 2_000_000
    dist = rng.integers(0, 257, n).astype(np.int32)
    qx, tx = (rng.uniform(0, 4096, n).astype(np.float32) for _ in range(2))
    qy, ty = (rng.uniform(0, 2200, n).astype(np.float32) for _ in range(2))
    # integer-pixel coordinates and exact-boundary cases (distance exactly 500)
    qx[: n // 2] = np.round(qx[: n // 2])
    qy[: n // 2] = np.round(qy[: n // 2])
    tx[: n // 2] = np.round(tx[: n // 2])
    ty[: n // 2] = np.round(ty[: n // 2])
    tx[:1000] = qx[:1000] + 300.0
    ty[:1000] = qy[:1000] + 400.0
    got = gpu_ctx.debug_penalty(dist, qx, qy, tx, ty)
    ref = oracle.penalty(dist, qx, qy, tx, ty)
    bad = np.nonzero(got != ref)[0]
    assert bad.size == 0, "penalty differs at %d tuples, e.g. %r" % (bad.size, (dist[bad[0]], qx[bad[0]], qy[bad[0]], tx[bad[0]], ty[bad[0]], got[bad[0]], ref[bad[0]]))
    assert (ref > dist).mean() > 0.3  # the penalty branch is well exercised


def test_match_errors(slamx):
    m = slamx.FeatureMatcher(mcfg(slamx))
    d = np.zeros((4, 32), np.uint8)
    with pytest.raises(ValueError):  # std::invalid_argument("Empty descriptors provided.")
        m.match(np.zeros((0, 32), np.uint8), d)
    with pytest.raises(ValueError):
        m.match(d, np.zeros((0, 32), np.uint8))
    with pytest.raises(RuntimeError):  # "Descriptor dimensions must match."
        m.match(d, np.zeros((4, 16), np.uint8))
    l2 = slamx.FeatureMatcher(slamx.MatcherConfig(1, 0, 1, 0, 0.5))
    with pytest.raises(RuntimeError):  # L2 is refused by match()
        l2.match(d, d)


# ---------------------------------------------------------------- device-resident batches
def test_block_ids_spill_into_grid_z():
    """Kernels that take (frame, block) off a 3-D grid put the blocks of a frame in grid.y; beyond 65535 blocks
    per frame (frames of half a gigapixel) the rest spills into grid.z.  SLAMX_GRID_Y_LIMIT lowers that limit to 5
    in a child process, so that ordinary frames (21 FAST tiles, dozens of describe / downscale blocks) take the
    spill path through FAST, expand, describe and the pyramid's downscale -- bit-exact against the oracle."""
    import subprocess
    import sys

    env = dict(os.environ, SLAMX_GRID_Y_LIMIT="5")
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(here, "test_parity_gpu.py"),
                        os.path.join(here, "test_pyramid_gpu.py"), "-k",
                        "batch_equals_single_frame or pipeline_odd_shapes or levels_match_the_cpu_downscale or pyramid_batch_with_matching or "
                        "fast_candidates_shapes_and_thresholds"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and " passed" in out, out[-2000:]


def test_nms_state_route_still_gives_the_oracle_s_keypoints():
    """The greedy NMS builds its kept-candidate bitmap after the loop in the grid's own LDS words; only candidate sets too
    large for grid + table go through the `state` array in global memory (until round 5 every frame of 65536+ candidates
    did).  In a child process SLAMX_NMS_STATE_ROUTE=1 sends every frame that way: the detector tests, dense and tied
    frames and the batch path must still give the oracle's keypoints in the oracle's order."""
    import subprocess
    import sys

    env = dict(os.environ, SLAMX_NMS_STATE_ROUTE="1")
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(here, "test_parity_gpu.py"), "-k",
                        "detect_windows_synthetic or detect_dense_candidates or filter_wave_count or batch_equals_single_frame or "
                        "frontend_against_committed_golden_vectors"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0 and " passed" in out, out[-2000:]


def test_region_tile_describe_on_every_shape():
    """describe_tile_kernel (orientation + BRIEF from shared 240 x 64-pixel region tiles) is what dense frames take; in a
    child process SLAMX_DESCRIBE_TILES=2 sends EVERY frame through it, so that the odd image shapes (narrower than a
    region, shorter than a band, keypoints within 21 pixels of every border), the reference's frames, the batch path
    and the pyramid levels exercise its region binning, halo loads and border masks -- bit-exact against the oracle."""
    import subprocess
    import sys

    env = dict(os.environ, SLAMX_DESCRIBE_TILES="2")
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(here, "test_parity_gpu.py"),
                        os.path.join(here, "test_pyramid_gpu.py"), "-k",
                        "batch_equals_single_frame or pipeline_odd_shapes or test_detect_and_compute or pyramid_batch_with_matching or dense_small_frames or "
                        "frontend_against_committed_golden_vectors"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0 and " passed" in out, out[-2000:]


@pytest.mark.parametrize("rows,cols,window,t", [(37, 553, 5, 15), (40, 300, 2, 10), (64, 240, 3, 8), (100, 100, 2, 12), (22, 700, 2, 5)])
def test_dense_small_frames_through_the_region_tiles(slamx, oracle, rows, cols, window, t):
    """Dense frames take describe_tile_kernel by themselves (>= 20 keypoints per 240 x 64-pixel region).  Small noisy
    frames put keypoints within 21 pixels of EVERY border and of the top-left corner, where the tile's 16-byte pieces
    straddle x = 0: in image row 0 such a piece starts at a negative buffer offset and used to be zero-filled as a whole,
    columns 0..7 of row 0 included (tests/fuzz/fuzz_detector.py seed 404 case 158: one descriptor bit of the keypoint at
    (17, 18) on a 37 x 553 frame).  Keypoints, angles and descriptors must be the oracle's."""
    rng = np.random.default_rng(rows * 1000 + cols)
    d = slamx.FeatureDetector(cfg_of(slamx, t=t, arc=5, window=window))
    Pp = oracle.det_params(d.cfg)
    for k in range(4):
        if k % 2 == 0:
            img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        else:  # blocks with small noise: corners at block boundaries, including the first rows and columns
            b = int(rng.integers(3, 9))
            lvl = rng.integers(0, 256, (rows // b + 1, cols // b + 1)).repeat(b, 0).repeat(b, 1)[:rows, :cols]
            img = np.clip(lvl + rng.integers(-6, 7, (rows, cols)), 0, 255).astype(np.uint8)
        img = np.ascontiguousarray(img)
        k_ref, d_ref = oracle.detect_and_compute(img, Pp)
        k_got, d_got = d.detectAndCompute(img)
        assert len(k_ref) >= 20 * ((cols + 239) // 240) * ((rows + 63) // 64), "the frame must be dense enough for the region tiles"
        assert_records_equal(k_got, k_ref, "%dx%d frame %d keypoints" % (rows, cols, k))
        assert np.array_equal(d_got, d_ref), "%dx%d frame %d: %d descriptor rows differ" % (
            rows, cols, k, int((d_got != d_ref).any(axis=1).sum()))


@pytest.mark.parametrize("patch,pairs", [(5, 24), (9, 8), (15, 64), (21, 512), (25, 1024), (33, 256), (63, 128)])
def test_other_patch_sizes_and_descriptor_widths_in_a_batch(slamx, oracle, patch, pairs):
    """Every describe path on keypoints 3 pixels from the border: PatchSize 5 ... 63 and NumBRIEFPairs 8 ... 1024 select
    the fused kernels (radius <= 15, <= 64 surviving pairs, width a multiple of 8), the three-kernel path (other widths
    or more pairs) or the generic kernel (radius > 15).  moments_kernel used to read its fixed 31-row tile through plain
    pointers: with a small patch the tile of an "inside" keypoint starts above the frame -- before the batch's first
    frame that is memory the batch does not own (tests/fuzz/fuzz_detector.py seed 81 case 52 faulted there)."""
    rng = np.random.default_rng(patch * 10000 + pairs)
    rows, cols = 72, 525
    d = slamx.FeatureDetector(cfg_of(slamx, t=30, arc=9, window=4, patch=patch, pairs=pairs))
    Pp = oracle.det_params(d.cfg)
    noise = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    flat = np.full((rows, cols), 90, np.uint8)
    few = flat.copy()
    few[20:31, 100:113] = 200
    frames = [noise, flat, few, noise]
    refs = [oracle.detect_and_compute(f, Pp) for f in frames]
    assert len(refs[0][0]) > 50 and len(refs[1][0]) == 0
    b = slamx.Batch(d, rows, cols, 4, max_candidates=rows * cols, max_keypoints=len(refs[0][0]) + 8)
    b.upload(np.stack(frames))
    b.detect_and_compute(4)
    for f in range(4):
        k_b, d_b = b.download(f)
        assert_records_equal(k_b, refs[f][0], "patch %d pairs %d frame %d" % (patch, pairs, f))
        assert len(refs[f][0]) == 0 or np.array_equal(d_b, refs[f][1])
    k1, d1 = d.detectAndCompute(noise)
    assert_records_equal(k1, refs[0][0], "single frame")
    assert np.array_equal(d1, refs[0][1])


def test_batch_equals_single_frame(slamx, oracle):
    d = slamx.FeatureDetector(cfg_of(slamx, window=6))
    Pp = oracle.det_params(d.cfg)
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 1, 0.7))
    mp = oracle.match_params(m.cfg)
    n = 5
    frames = np.stack([synth(slamx, k, 240, 320, 1200, rmin=3, rmax=16) for k in range(n)])
    b = slamx.Batch(d, 240, 320, n, max_candidates=20000, max_keypoints=8000)
    b.upload(frames)
    b.detect_and_compute(n)
    b.match_consecutive(m, n, use_keypoints=True)
    ncand, nkp, nm = b.counts(n)
    prev = None
    for k in range(n):
        k_ref, d_ref = oracle.detect_and_compute(frames[k], Pp)
        k_got, d_got = b.download(k)
        assert nkp[k] == len(k_ref) and ncand[k] == len(oracle.fast_candidates(frames[k], Pp))
        assert_records_equal(k_got, k_ref, "batch frame %d" % k)
        assert np.array_equal(d_got, d_ref)
        if prev is not None:
            ref_m = oracle.match(d_ref, prev[1], k_ref, prev[0], mp)
            assert_records_equal(b.download_matches(m, k), ref_m, "batch matches %d" % k)
            assert nm[k] == len(ref_m)
        prev = (k_ref, d_ref)
    assert nm[0] == 0


@pytest.mark.parametrize("window", [3, 12])
def test_batch_of_identical_frames_is_timing_independent(slamx, oracle, window):
    """The greedy NMS has waves that filter candidates against a grid another wave is writing, the sort has a
    work queue: the result must not depend on how the waves interleave.  256 copies of two crowded frames run
    side by side (different CUs, different neighbours, different timing), three times over; every copy must
    give the oracle's keypoints, descriptors and matches."""
    d = slamx.FeatureDetector(cfg_of(slamx, t=25, arc=9, window=window))
    Pp = oracle.det_params(d.cfg)
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 1, 0.7))
    rng = np.random.default_rng(window)
    noise = rng.integers(0, 256, (150, 420), dtype=np.uint8)
    rects = synth(slamx, 7, 150, 420, 900, rmin=3, rmax=14)
    n = 256
    frames = np.stack([noise if k % 2 == 0 else rects for k in range(n)])
    refs = [oracle.detect_and_compute(frames[k], Pp) for k in range(2)]
    ref_m = [oracle.match(refs[1][1], refs[0][1], refs[1][0], refs[0][0], oracle.match_params(m.cfg)),
             oracle.match(refs[0][1], refs[1][1], refs[0][0], refs[1][0], oracle.match_params(m.cfg))]
    b = slamx.Batch(d, 150, 420, n, max_candidates=16384, max_keypoints=8192)
    b.upload(frames)
    for rep in range(3):
        b.detect_and_compute(n)
        b.match_consecutive(m, n, use_keypoints=True)
        for k in range(n):
            kps, desc = b.download(k)
            rk, rd = refs[k % 2]
            assert_records_equal(kps, rk, "rep %d frame %d" % (rep, k))
            assert np.array_equal(desc, rd), "descriptors rep %d frame %d" % (rep, k)
            if k > 0:
                # frame k (odd: rects vs noise; even: noise vs rects) against frame k - 1
                assert_records_equal(b.download_matches(m, k), ref_m[0] if k % 2 == 1 else ref_m[1],
                                     "matches rep %d frame %d" % (rep, k))


def test_batch_upload_host_buffer_reusable_on_return(slamx):
    """slamx.h: the host buffer may be reused as soon as slamx_batch_upload returns."""
    d = slamx.FeatureDetector(cfg_of(slamx))
    frames = np.stack([synth(slamx, k) for k in range(2)])
    b = slamx.Batch(d, 480, 640, 2)
    for _ in range(10):
        tmp = frames.copy()
        b.upload(tmp)
        tmp[:] = 0
        for k in range(2):
            assert np.array_equal(b.download_image(k), frames[k])


def test_batch_upload_async_from_pinned_memory(slamx, oracle):
    """Streaming ingest: the copy is queued, not waited for; pageable memory is refused."""
    det = slamx.FeatureDetector(cfg_of(slamx))
    Pp = oracle.det_params(det.cfg)
    frames = np.stack([synth(slamx, k, 120, 200, 300, rmin=3, rmax=16) for k in range(3)])
    pin = slamx.pinned_array(frames.shape)
    pin[...] = frames
    batch = slamx.Batch(det, 120, 200, 3, 8192, 4096)
    batch.upload_async(pin)
    batch.detect_and_compute(3)
    for k in range(3):
        kps, desc = batch.download(k)
        rk, rd = oracle.detect_and_compute(frames[k], Pp)
        assert_records_equal(kps, rk, "async upload frame %d" % k)
        assert np.array_equal(desc, rd)
    # rows with padding (a cropped view of a wider pinned image): the strided copy path
    wide = slamx.pinned_array((3, 120, 300))
    wide[:, :, :200] = frames
    batch.upload_async(wide[:, :, :200])
    batch.detect_and_compute(3)
    kps, _ = batch.download(1)
    assert_records_equal(kps, oracle.detect_and_compute(frames[1], Pp)[0], "async strided upload")
    with pytest.raises(ValueError):
        batch.upload_async(frames)  # pageable


def test_batch_download_results_async(slamx, oracle):
    """The streaming download: all results of a batch into page-locked memory with five queued copies."""
    d = slamx.FeatureDetector(cfg_of(slamx, window=8))
    Pp = oracle.det_params(d.cfg)
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 1, 0.8))
    frames = [synth(slamx, k, 200, 300, 250, rmin=4, rmax=24) for k in range(3)]
    b = slamx.Batch(d, 200, 300, 3, 8192, 1024)
    b.upload(np.stack(frames))
    b.detect_and_compute(3)
    b.match_consecutive(m, 3, use_keypoints=True, first_train_frame=2)
    bufs = b.result_buffers(3, 1024)
    b.download_results_async(3, bufs)
    d.ctx.synchronize()
    kps, desc, matches, nk, nm = bufs
    ref = [oracle.detect_and_compute(f, Pp) for f in frames]
    for k in range(3):
        assert nk[k] == len(ref[k][0])
        assert_records_equal(kps[k, : nk[k]], ref[k][0], "keypoints of frame %d" % k)
        assert np.array_equal(desc[k, : nk[k]], ref[k][1])
        t = (k - 1) % 3
        want = oracle.match(ref[k][1], ref[t][1], ref[k][0], ref[t][0], oracle.match_params(m.cfg))
        assert_records_equal(matches[k, : nm[k]], want, "matches of frame %d" % k)
    with pytest.raises(ValueError):  # pageable memory is refused, as for the uploads
        b.download_results_async(3, (np.empty((3, 1024), slamx.KEYPOINT_DTYPE),) + bufs[1:])


@pytest.mark.parametrize("mode", ["", "copy"])
def test_batch_download_packed_async(slamx, oracle, monkeypatch, mode):
    """The packed streaming download: a device-side pack step gathers the valid keypoints / descriptors / matches of
    all frames into one buffer (prefix of the counts + copy kernel) and brings it to page-locked memory -- the copy
    kernel's own stores over PCIe (default) or one queued copy (SLAMX_PACKED_DOWNLOAD=copy); frames with different
    counts, an empty frame in the middle, the tail call when the host buffer was cut short, and a batch no matcher
    has run on."""
    if mode:
        monkeypatch.setenv("SLAMX_PACKED_DOWNLOAD", mode)
    else:
        monkeypatch.delenv("SLAMX_PACKED_DOWNLOAD", raising=False)
    d = slamx.FeatureDetector(cfg_of(slamx, window=8))
    Pp = oracle.det_params(d.cfg)
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 1, 0.8))
    frames = [synth(slamx, k, 200, 300, n, rmin=4, rmax=24) for k, n in enumerate((250, 120, 0, 400, 60))]
    frames[2][:] = 128  # no corners at all
    F = len(frames)
    b = slamx.Batch(d, 200, 300, F, 8192, 1024)
    b.upload(np.stack(frames))
    b.detect_and_compute(F)
    buf = b.packed_buffer(F)
    b.download_packed_async(F, buf)  # before any matcher has run: no matches section
    d.ctx.synchronize()
    kp_first, m_first, status, kps, desc, matches = slamx.unpack_results(buf)
    ref = [oracle.detect_and_compute(f, Pp) for f in frames]
    assert np.diff(kp_first).tolist() == [len(r[0]) for r in ref] and not m_first.any() and len(matches) == 0
    b.match_consecutive(m, F, use_keypoints=True, first_train_frame=F - 1)
    buf[:] = 0xEE
    b.download_packed_async(F, buf)
    d.ctx.synchronize()
    kp_first, m_first, status, kps, desc, matches = slamx.unpack_results(buf)
    h = slamx.packed_header(buf)
    assert h["nframes"] == F and h["width"] == 32 and not status.any()
    assert h["total_bytes"] < 0.35 * buf.nbytes  # far below the capacity-padded worst case
    assert kp_first[0] == 0 and kp_first[2] == kp_first[3]  # the empty frame owns an empty range
    mp = oracle.match_params(m.cfg)
    for k in range(F):
        sl = slice(kp_first[k], kp_first[k + 1])
        assert_records_equal(kps[sl], ref[k][0], "keypoints of frame %d" % k)
        assert np.array_equal(desc[sl], ref[k][1]) if len(ref[k][0]) else desc[sl].shape[0] == 0
        t = (k - 1) % F
        got = matches[m_first[k]:m_first[k + 1]]
        if len(ref[k][0]) and len(ref[t][0]):
            assert_records_equal(got, oracle.match(ref[k][1], ref[t][1], ref[k][0], ref[t][0], mp), "matches of frame %d" % k)
        else:
            assert len(got) == 0
    # a first copy that was cut short (a streaming caller guessing too low), completed by the tail call
    total = int(h["total_bytes"])
    short = b.packed_buffer(F)
    short[:] = 0
    have = b.download_packed_async(F, short, total // 3)
    d.ctx.synchronize()
    assert slamx.packed_header(short)["total_bytes"] == total and have < total
    assert not short[have:].any()  # nothing beyond what the caller allowed
    b.fetch_packed_tail(short, have)
    assert np.array_equal(short[:total], buf[:total])
    with pytest.raises(ValueError):  # pageable memory is refused
        b.download_packed_async(F, np.empty(buf.nbytes, np.uint8))


def test_batch_capacity_overflow_is_loud(slamx):
    d = slamx.FeatureDetector(cfg_of(slamx, window=6))
    frame = synth(slamx, 0, 240, 320, 1200, rmin=3, rmax=16)
    b = slamx.Batch(d, 240, 320, 1, max_candidates=64, max_keypoints=64)
    b.upload(frame)
    b.detect_and_compute(1)
    with pytest.raises(slamx.CapacityError):
        b.counts(1)


# ---------------------------------------------------------------- a15 Camera::undistortImage
def test_camera_yaml_and_undistort(slamx, gpu_ctx, oracle, ref_data):
    """test_preprocessing.cpp:15-29: load camera.yml, undistort the first KITTI frame; the
    u8 -> [0,1] -> u8 round trip is exact."""
    cam = slamx.load_camera(os.path.join(ref_data, "camera.yml"))
    assert (cam.width, cam.height) == (1392, 512)
    assert (cam.fx, cam.fy, cam.cx, cam.cy) == (984.2439, 980.8141, 690.0, 233.1966)
    assert (cam.k1, cam.k2, cam.p1, cam.p2, cam.k3) == (-0.3728755, 0.2037299, 0.002219027, 0.001383707, -0.07233722)
    img = load_gray(os.path.join(ref_data, "images/0000000000.png"))
    got = gpu_ctx.undistort(cam, img)
    ref = oracle.undistort(img, cam)
    assert np.array_equal(got, ref)
    back = (got * 255.0).astype(np.uint8)  # test_preprocessing.cpp:28-29
    assert set(np.unique(back)) <= set(np.unique(img)) | {0}
    with pytest.raises(RuntimeError, match="does not match camera image size"):
        gpu_ctx.undistort(cam, img[:100])


@pytest.mark.parametrize("shape", [(37, 53), (64, 64), (45, 700), (101, 257), (250, 1000), (333, 249), (500, 47)])
@pytest.mark.parametrize("window", [3, 12])
def test_pipeline_odd_shapes(slamx, oracle, shape, window):
    """detectAndCompute + match on image sizes that are not multiples of any tile (ragged last tiles,
    keypoints within 21 pixels of every border, images narrower than one tile)."""
    rows, cols = shape
    d = slamx.FeatureDetector(cfg_of(slamx, window=window))
    Pp = oracle.det_params(d.cfg)
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 20, 1, 0.7))
    mp = oracle.match_params(m.cfg)
    imgs = [np.ascontiguousarray(synth(slamx, k, 512, 1024, 2500, rmin=3, rmax=14)[:rows, :cols]) for k in range(2)]
    got = [d.detectAndCompute(i) for i in imgs]
    ref = [oracle.detect_and_compute(i, Pp) for i in imgs]
    for k in range(2):
        assert_records_equal(got[k][0], ref[k][0], "%s window %d frame %d keypoints" % (shape, window, k))
        assert np.array_equal(got[k][1], ref[k][1])
    if len(ref[0][0]) and len(ref[1][0]):
        want = oracle.match(ref[1][1], ref[0][1], ref[1][0], ref[0][0], mp)
        assert_records_equal(m.match(got[1][1], got[0][1], got[1][0], got[0][0]), want, "%s matches" % (shape,))


# ---------------------------------------------------------------- BASELINE.json full-size workloads
@pytest.mark.parametrize("name,rows,cols,window,n_rect,rmin,rmax", [
    ("1080p_2k", 1080, 1920, 12, 6800, 3, 24),   # bench.py default workload (the metric's config)
    ("c2", 480, 640, 6, 5000, 3, 16),            # BASELINE.json configs[1]
    ("c3", 1080, 1920, 8, 32000, 3, 16),         # BASELINE.json configs[2]: ~80k candidates, ~10k keypoints
    ("uhd", 2160, 3840, 10, 40000, 3, 20),       # beyond the bench sizes: 3840x2160, ~100k candidates, ~14k keypoints
])
def test_full_size_workloads_bit_exact(slamx, oracle, name, rows, cols, window, n_rect, rmin, rmax):
    """The bench workloads themselves, two consecutive frames each: candidates, keypoint order,
    angles, descriptors and matches against the oracle (which needs a few seconds at c3)."""
    d = slamx.FeatureDetector(cfg_of(slamx, window=window))
    Pp = oracle.det_params(d.cfg)
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 20, 1, 0.5))
    mp = oracle.match_params(m.cfg)
    frames = [synth(slamx, k, rows, cols, n_rect, rmin=rmin, rmax=rmax) for k in range(2)]
    b = slamx.Batch(d, rows, cols, 2, max_candidates=131072, max_keypoints=16384)
    b.upload(np.stack(frames))
    b.detect_and_compute(2)
    b.match_consecutive(m, 2, use_keypoints=True, first_train_frame=1)
    ref = [oracle.detect_and_compute(f, Pp) for f in frames]
    for k in range(2):
        assert np.array_equal(b.download_image(k), frames[k]), "%s frame %d: upload" % (name, k)
        assert np.array_equal(b.download_image(k, blurred=True), oracle.gaussian_blur(frames[k])), "%s frame %d: blur" % (name, k)
        k_got, d_got = b.download(k)
        assert_records_equal(k_got, ref[k][0], "%s frame %d keypoints" % (name, k))
        assert np.array_equal(d_got, ref[k][1])
    for q, t_ in ((1, 0), (0, 1)):
        want = oracle.match(ref[q][1], ref[t_][1], ref[q][0], ref[t_][0], mp)
        assert_records_equal(b.download_matches(m, q), want, "%s matches %d->%d" % (name, q, t_))
    assert len(ref[0][0]) > 1500


def test_c1_job_bit_exact(slamx, oracle, ref_data):
    """BASELINE.json configs[0] -- the reference's own CPU-runnable case -- as bench.py --workload c1 runs it: the whole
    job of 50 synthetic 640x480 frames (~500 keypoints each) with the SHIPPED detector YAML (window 12), every frame
    detected, described and matched to its predecessor (frame 0 to frame 49).  Every frame's keypoints, descriptors
    and matches are hashed and compared with the CPU sequence runner (oracle/cpu_sequence.cpp)."""
    import importlib.util
    from conftest import ROOT

    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    wl = bench.WORKLOADS["c1"]
    B = wl["batch"]
    assert (B, wl["rows"], wl["cols"], wl["window"]) == (50, 480, 640, 12)
    d = slamx.FeatureDetector(os.path.join(ref_data, "feature_detector.yml"))
    assert d.cfg.suppression_window_size == wl["window"]
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 20, 1, 0.5))
    frames = np.stack([slamx.synth_frame(bench.SEED, k, wl["rows"], wl["cols"], wl["n_rect"], wl["rmin"], wl["rmax"]) for k in range(B)])
    b = slamx.Batch(d, wl["rows"], wl["cols"], B, wl["max_cand"], wl["max_kp"])
    b.upload(frames)
    b.detect_and_compute(B)
    b.match_consecutive(m, B, use_keypoints=True, first_train_frame=B - 1)
    _, want = oracle.sequence_mt(frames, oracle.det_params(d.cfg), oracle.match_params(m.cfg), True, 8)
    ncand, nkp, nmatch = b.counts(B)
    assert nkp.tolist() == want[:, 0].astype(int).tolist()
    assert nmatch.tolist() == want[:, 1].astype(int).tolist()
    assert 400 < nkp.mean() < 600
    for k in range(B):
        kps, desc = b.download(k)
        mt = b.download_matches(m, k)
        h = 0xCBF29CE484222325
        for byte in kps.tobytes() + desc.tobytes() + mt.tobytes():
            h = ((h ^ byte) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        assert h == int(want[k, 2]), "frame %d" % k


# ---------------------------------------------------------------- HIP path against the COMMITTED vectors (no oracle in the loop)
def _golden():
    from conftest import ROOT

    return np.load(os.path.join(ROOT, "tests", "golden", "oracle_golden.npz"))


@pytest.mark.parametrize("name", REF_IMAGES)
def test_frontend_against_committed_golden_vectors(slamx, det, ref_data, name):
    """tests/golden/oracle_golden.npz (generator: tests/golden/make_golden.py, run in the build container) holds
    candidates, keypoints (order, response, angle bits), descriptors and a blur checksum of every reference PNG.
    The HIP path is compared with those bytes directly: on the GPU box this does not depend on which oracle .so
    travelled there."""
    img = load_gray(os.path.join(ref_data, name))
    key = name.replace("/", "_").replace(".png", "")
    with _golden() as g:
        cand = det.fast_candidates(img)
        got = np.stack([cand["x"], cand["y"], cand["response"]], 1).astype(np.uint16)
        assert np.array_equal(got, g[key + "_cand_xys"]), "candidates " + name
        kps, desc = det.detectAndCompute(img)
        assert np.array_equal(kps.view(np.uint8).reshape(len(kps), 20), g[key + "_kps"]), "keypoints " + name
        assert np.array_equal(desc, g[key + "_desc"]), "descriptors " + name
        assert int(det.gaussian_blur(img).astype(np.uint64).sum()) == int(g[key + "_blur_sum"][0]), "blur " + name


def test_matcher_against_committed_golden_vectors(slamx, det, ref_data):
    """Matches of every consecutive reference frame pair with the shipped matcher YAML (filter on: top-20 through the
    host's std::partial_sort), with and without keypoints, plus the unfiltered best matches -- against the committed
    bytes."""
    import importlib.util
    from conftest import ROOT

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    m_ship = slamx.FeatureMatcher(os.path.join(ref_data, "feature_matcher.yml"))
    m_all = slamx.FeatureMatcher(slamx.MatcherConfig(0, 0, 20, 1, 0.5))
    feats = {}
    with _golden() as g:
        for a, b, tag in mg.PAIRS:
            for n in (a, b):
                if n not in feats:
                    feats[n] = det.detectAndCompute(load_gray(os.path.join(ref_data, n)))
            (k1, d1), (k2, d2) = feats[a], feats[b]
            assert m_ship.match(d1, d2, k1, k2).tobytes() == g[tag + "_match_kp"].tobytes(), tag + " with keypoints"
            assert m_ship.match(d1, d2).tobytes() == g[tag + "_match_nokp"].tobytes(), tag + " without keypoints"
            assert m_all.match(d1, d2, k1, k2).tobytes() == g[tag + "_best_kp"].tobytes(), tag + " unfiltered"


@pytest.mark.parametrize("shape,dense", [((200, 300), False), ((480, 640), True)])
def test_batch_in_chunks_equals_whole_batch(slamx, shape, dense):
    """slamx_batch_detect_and_compute_range / _match_consecutive_range / _upload_async_flags(SLOTS_IDLE) (round 5: what
    slam::FrameBatch::processSequence pipelines a sequence with): a batch filled and processed chunk by chunk -- uneven
    chunks, a one-frame chunk, an empty frame at a chunk's start -- must leave exactly the bytes the whole-batch calls
    leave: every count, keypoint, descriptor and match, through the packed download.  Sparse frames (popcount matcher,
    per-keypoint describe tiles) and dense ones (matrix-core matcher in spatial order, region tiles)."""
    rows, cols = shape
    d = slamx.FeatureDetector(cfg_of(slamx, window=6 if dense else 8))
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 1, 0.8))
    counts = (5000, 4000, 0, 4500, 5200, 3000, 4800, 5100, 100, 4900, 5000) if dense else (250, 120, 0, 400, 60, 300, 10, 350, 200, 90, 310)
    frames = np.stack([synth(slamx, k, rows, cols, n, rmin=3 if dense else 4, rmax=16 if dense else 24) for k, n in enumerate(counts)])
    frames[2][:] = 77
    F = len(frames)
    pin = slamx.pinned_array(frames.shape)
    pin[...] = frames

    def run(chunks):
        b = slamx.Batch(d, rows, cols, F, 32768, 4096)
        for first, n in chunks:
            b.upload_async(pin[first:first + n], first, slots_idle=len(chunks) > 1)
            b.detect_and_compute(n, first)
            b.match_consecutive(m, n, use_keypoints=True, first_train_frame=-1, first=first)
        buf = b.packed_buffer(F)
        b.download_packed_async(F, buf)
        d.ctx.synchronize()
        total = int(slamx.packed_header(buf)["total_bytes"])
        nc, nk, nm = b.counts(F)
        return bytes(buf[:total]), nc.tolist(), nk.tolist(), nm.tolist()

    whole = run([(0, F)])
    assert max(whole[2]) > (1024 if dense else 50) and whole[3][0] == 0 and sum(whole[3]) > 0
    for chunks in ([(0, 3), (3, 1), (4, 5), (9, 2)], [(0, 1), (1, 1), (2, 8), (10, 1)], [(0, 6), (6, 5)]):
        assert run(chunks) == whole, chunks
