"""GPU parity of the image pyramid (pyramid.hip, slamx_pyramid_*) against oracle/pyramid_oracle.cpp +
the per-level CPU detector.  Integer / byte work: every comparison is exact.  Build-side extension
(the reference has no pyramid): level 0 is the reference's detector, levels >= 1 are unpinned."""
import os

import numpy as np
import pytest

from conftest import load_gray
from test_parity_gpu import assert_records_equal, cfg_of, mcfg, synth

pytestmark = pytest.mark.gpu


def run_pyramid(slamx, det, frames, n_levels, scale, max_cand=0, max_kp=0):
    frames = np.stack(frames)
    n, rows, cols = frames.shape
    pyr = slamx.Pyramid(det, rows, cols, n, n_levels, scale, max_cand, max_kp)
    pyr.level(0).upload(frames)
    pyr.detect_and_compute(n)
    return pyr


@pytest.mark.parametrize("shape,n_levels,scale", [((480, 640), 8, 1.2), ((1080, 1920), 8, 1.2), ((133, 211), 5, 1.2),
                                                   ((240, 320), 4, 2.0), ((97, 64), 3, 1.5), ((512, 1392), 8, 1.2)])
def test_levels_match_the_cpu_downscale(slamx, oracle, shape, n_levels, scale):
    det = slamx.FeatureDetector(cfg_of(slamx, window=6))
    img = synth(slamx, 1, shape[0], shape[1], max(40, shape[0] * shape[1] // 800), rmin=3, rmax=20)
    pyr = run_pyramid(slamx, det, [img, img[::-1].copy()], n_levels, scale)
    ref = oracle.pyramid_images(img, n_levels, scale)
    assert pyr.shapes == [r.shape for r in ref] == oracle.pyramid_sizes(shape[0], shape[1], n_levels, scale)
    for l in range(n_levels):
        got = pyr.level(l).download_image(0)
        assert np.array_equal(got, ref[l]), "level %d differs at %d pixels" % (l, int((got != ref[l]).sum()))
    flipped = oracle.pyramid_images(img[::-1].copy(), n_levels, scale)
    assert np.array_equal(pyr.level(n_levels - 1).download_image(1), flipped[-1])


@pytest.mark.parametrize("name", ["images/0000000000.png", "images_test_loop/0.png"])
def test_pyramid_features_reference_images(slamx, oracle, ref_data, name):
    det = slamx.FeatureDetector(os.path.join(ref_data, "feature_detector.yml"))
    P = oracle.det_params(det.cfg)
    img = load_gray(os.path.join(ref_data, name))
    pyr = run_pyramid(slamx, det, [img], 8, 1.2)
    k_got, d_got = pyr.level(0).download(0)
    k_ref, d_ref = oracle.pyramid_detect_and_compute(img, P, 8, 1.2)
    assert len(k_ref) > len(oracle.detect(img, P))  # the lower levels add keypoints
    assert_records_equal(k_got, k_ref, "pyramid keypoints " + name)
    assert np.array_equal(d_got, d_ref)
    sizes = np.unique(k_got["size"])
    assert len(sizes) > 3 and sizes[0] == 6.0  # Keypoint::size carries the level: 6 * 1.2 ** l


@pytest.mark.parametrize("window,n_levels,scale", [(6, 8, 1.2), (12, 4, 1.5), (8, 1, 1.2)])
def test_pyramid_batch_with_matching(slamx, oracle, window, n_levels, scale):
    """A batch of frames through the pyramid, every frame matched to its predecessor on the merged sets."""
    det = slamx.FeatureDetector(cfg_of(slamx, window=window))
    P = oracle.det_params(det.cfg)
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 1, 1, 0.8))
    mp = oracle.match_params(m.cfg)
    frames = [synth(slamx, k, 300, 420, 700, rmin=3, rmax=24) for k in range(4)]
    pyr = run_pyramid(slamx, det, frames, n_levels, scale, max_cand=32768)
    b = pyr.level(0)
    b.match_consecutive(m, len(frames), use_keypoints=True, first_train_frame=-1)
    ref = [oracle.pyramid_detect_and_compute(f, P, n_levels, scale) for f in frames]
    ncand, nkp, nmatch = b.counts(len(frames))
    for k, (kr, dr) in enumerate(ref):
        kg, dg = b.download(k)
        assert_records_equal(kg, kr, "frame %d" % k)
        assert np.array_equal(dg, dr)
        assert nkp[k] == len(kr)
        if k > 0:
            want = oracle.match(dr, ref[k - 1][1], kr, ref[k - 1][0], mp)
            assert_records_equal(b.download_matches(m, k), want, "matches of frame %d" % k)
    if n_levels == 1:  # a one-level pyramid is the plain batch
        plain = slamx.Batch(det, 300, 420, 4)
        plain.upload(np.stack(frames))
        plain.detect_and_compute(4)
        for k in range(4):
            assert_records_equal(plain.download(k)[0], b.download(k)[0], "one level = plain batch")
    # a second pass over the same frames restarts from level 0 (counts are not accumulated)
    pyr.detect_and_compute(len(frames))
    assert np.array_equal(b.counts(len(frames))[1], nkp)


@pytest.mark.parametrize("pairs", [8, 24, 64])
def test_pyramid_narrow_descriptors(slamx, oracle, pairs):
    """Descriptor widths that are not a multiple of 4 bytes (1, 3 bytes) take the byte-wise merge kernel."""
    det = slamx.FeatureDetector(cfg_of(slamx, window=8, patch=15, pairs=pairs))
    P = oracle.det_params(det.cfg)
    img = synth(slamx, 2, 240, 360, 500, rmin=3, rmax=20)
    pyr = run_pyramid(slamx, det, [img], 4, 1.3, max_cand=16384)
    k_got, d_got = pyr.level(0).download(0)
    k_ref, d_ref = oracle.pyramid_detect_and_compute(img, P, 4, 1.3)
    assert d_ref.shape[1] == pairs // 8
    assert_records_equal(k_got, k_ref, "keypoints, %d pairs" % pairs)
    assert np.array_equal(d_got, d_ref)


def test_pyramid_merged_capacity_overflow_is_loud(slamx):
    det = slamx.FeatureDetector(cfg_of(slamx, window=6))
    img = synth(slamx, 0, 300, 420, 700, rmin=3, rmax=24)
    pyr = run_pyramid(slamx, det, [img], 6, 1.2, max_cand=8192, max_kp=0)
    n = pyr.level(0).counts(1)[1][0]
    own = slamx.FeatureDetector(det.cfg).detect(img).shape[0]
    assert n > own
    tight = run_pyramid(slamx, det, [img], 6, 1.2, max_cand=8192, max_kp=own + 10)  # level 0 fits, the merge does not
    with pytest.raises(slamx.CapacityError):
        tight.level(0).counts(1)


def test_pyramid_argument_errors(slamx):
    det = slamx.FeatureDetector(cfg_of(slamx))
    with pytest.raises(ValueError):
        slamx.Pyramid(det, 100, 100, 1, 0, 1.2)
    with pytest.raises(ValueError):
        slamx.Pyramid(det, 100, 100, 1, 4, 1.0)
    with pytest.raises(ValueError):
        slamx.Pyramid(det, 20, 20, 1, 8, 1.5)  # the last levels would be smaller than 8 pixels


def test_full_size_pyramid_workload_bit_exact(slamx, oracle):
    """bench.py's c3p workload itself (1080p, 8 levels, shipped YAML, ~12.5k keypoints per frame over all
    levels), two consecutive frames: merged keypoints, descriptors and the matches of both directions (1.6e8
    pairs each, through the matrix-core scan) against the oracle."""
    det = slamx.FeatureDetector(cfg_of(slamx, window=12))
    P = oracle.det_params(det.cfg)
    m = slamx.FeatureMatcher(mcfg(slamx, 0, 20, 1, 0.5))
    mp = oracle.match_params(m.cfg)
    frames = [synth(slamx, k, 1080, 1920, 6800, rmin=3, rmax=24) for k in range(2)]
    pyr = run_pyramid(slamx, det, frames, 8, 1.2, max_cand=32768, max_kp=16384)
    b = pyr.level(0)
    b.match_consecutive(m, 2, use_keypoints=True, first_train_frame=1)
    ref = [oracle.pyramid_detect_and_compute(f, P, 8, 1.2) for f in frames]
    for k in range(2):
        k_got, d_got = b.download(k)
        assert_records_equal(k_got, ref[k][0], "c3p frame %d keypoints" % k)
        assert np.array_equal(d_got, ref[k][1])
    assert len(ref[0][0]) > 11000
    for q, t_ in ((1, 0), (0, 1)):
        want = oracle.match(ref[q][1], ref[t_][1], ref[q][0], ref[t_][0], mp)
        assert_records_equal(b.download_matches(m, q), want, "c3p matches %d->%d" % (q, t_))


def test_pyramid_against_committed_golden_vectors(slamx, ref_data):
    """tests/golden/oracle_golden.npz holds the 5-level pyramid of images_test_loop/0.png: level shapes, pixel sums,
    the smallest level and the merged keypoints / descriptors -- compared without the oracle in the loop."""
    from conftest import ROOT

    det = slamx.FeatureDetector(os.path.join(ref_data, "feature_detector.yml"))
    img = load_gray(os.path.join(ref_data, "images_test_loop/0.png"))
    pyr = run_pyramid(slamx, det, [img], 5, 1.2)
    with np.load(os.path.join(ROOT, "tests", "golden", "oracle_golden.npz")) as g:
        assert [list(s) for s in pyr.shapes] == g["pyramid_loop0_level_shapes"].tolist()
        sums = [int(pyr.level(l).download_image(0).astype(np.uint64).sum()) for l in range(5)]
        assert sums == g["pyramid_loop0_level_sums"].tolist()
        assert np.array_equal(pyr.level(4).download_image(0), g["pyramid_loop0_level4"])
        k, d = pyr.level(0).download(0)
        assert np.array_equal(k.view(np.uint8).reshape(len(k), 20), g["pyramid_loop0_kps"])
        assert np.array_equal(d, g["pyramid_loop0_desc"])
