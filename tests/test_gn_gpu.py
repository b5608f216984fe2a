"""GPU tests of the sliding-window Gauss-Newton step (gn.hip, slamx_gn_*) against the independent
dense CPU statement in oracle/gn_oracle.cpp.  Floating point (f64 on the matrix cores): north_star
asks for pose / landmark values within 1e-5 relative; the tolerances below are much tighter and are
written next to every comparison.  The problem is build-defined (no reference counterpart)."""
import numpy as np
import pytest

import gn_scene

pytestmark = pytest.mark.gpu

S_RTOL = 1e-9      # entries of the reduced system, relative to its largest entry
STEP_ATOL = 1e-8   # camera step / stepped poses (unit-scale quantities)
X_ATOL = 1e-7      # landmark coordinates (metres, values up to ~12)
COST_RTOL = 1e-9


def oracle_run(oracle, w, lam, n_fixed, iters):
    poses, X = w["poses0"].copy(), w["X0"].copy()
    costs = []
    for _ in range(iters):
        r = oracle.gn_step_dense(poses, X, w["obs"], w["K"], lam, n_fixed)
        assert r["rc"] == 0
        costs.append(r["cost"])
        poses, X = oracle.gn_apply(poses, X, r["delta_c"], r["delta_x"])
    costs.append(oracle.gn_cost(poses, X, w["obs"], w["K"], lam))
    return poses, X, np.array(costs)


def upload(slamx, windows, L=None):
    W = windows[0]["poses0"].shape[0]
    L = L or max(w["X0"].shape[0] for w in windows)
    n = len(windows)
    poses = np.stack([w["poses0"] for w in windows])
    X = np.zeros((n, L, 3))
    obs = np.full((n, L, W, 2), np.nan, np.float32)
    nl = np.zeros(n, np.int32)
    for i, w in enumerate(windows):
        k = w["X0"].shape[0]
        X[i, :k] = w["X0"]
        obs[i, :k] = w["obs"]
        nl[i] = k
    gn = slamx.GaussNewton(n, W, L)
    gn.upload(poses, X, obs, nl)
    return gn


@pytest.mark.parametrize("W", [1, 2, 3, 5, 8, 9, 10])
@pytest.mark.parametrize("L", [1, 7, 8, 9, 64, 301])
def test_reduced_system_matches_the_dense_oracle(slamx, oracle, W, L):
    lam, n_fixed = 1e-3, min(2, W)
    w = gn_scene.make_window(100 * W + L, W, L)
    gn = upload(slamx, [w])
    S, g, d, c = gn.normal_equations(w["K"], lam, n_fixed)
    r = oracle.gn_step_dense(w["poses0"], w["X0"], w["obs"], w["K"], lam, n_fixed)
    assert r["rc"] == 0
    scale = max(np.abs(r["S"]).max(), 1.0)
    np.testing.assert_allclose(S[0], r["S"], rtol=0, atol=S_RTOL * scale)
    np.testing.assert_allclose(g[0], r["g"], rtol=0, atol=S_RTOL * max(np.abs(r["g"]).max(), 1.0))
    assert c[0] == pytest.approx(r["cost"], rel=COST_RTOL, abs=1e-12)
    # a handful of landmarks under up to 10 cameras is barely constrained beyond the damping: the solve then
    # amplifies the 1e-9 agreement of (S, g) -- still an order below north_star's 1e-5
    np.testing.assert_allclose(d[0], r["delta_c"].reshape(-1), rtol=0, atol=STEP_ATOL if L >= 64 else 1e-6)
    np.testing.assert_array_equal(S[0], S[0].T)  # assembled from mirrored tiles: exactly symmetric


@pytest.mark.parametrize("W,L,lam,n_fixed", [(10, 200, 0.0, 2), (10, 200, 1e-2, 1), (6, 150, 1e-4, 2), (3, 90, 0.0, 2), (9, 77, 1.0, 0)])
def test_iterations_follow_the_oracle(slamx, oracle, W, L, lam, n_fixed):
    w = gn_scene.make_window(7 * W + L, W, L)
    iters = 5
    gn = upload(slamx, [w])
    gn.iterate(w["K"], lam, n_fixed, iters)
    poses, X, cost, status = gn.download()
    p_ref, X_ref, c_ref = oracle_run(oracle, w, lam, n_fixed, iters)
    assert status[0] == 0
    np.testing.assert_allclose(cost[0], c_ref, rtol=1e-7, atol=1e-9)  # costs drop by orders of magnitude: relative
    np.testing.assert_allclose(poses[0], p_ref, rtol=0, atol=STEP_ATOL)
    np.testing.assert_allclose(X[0], X_ref, rtol=0, atol=X_ATOL)
    if W >= 6:  # (an undamped step of one free camera over 90 noisy landmarks may overshoot: the oracle does too)
        assert cost[0, -1] < cost[0, 0]
    if n_fixed > 0:
        np.testing.assert_array_equal(poses[0, :n_fixed], w["poses0"][:n_fixed])  # held cameras: bit-identical


def test_batch_of_ragged_windows(slamx, oracle):
    """Windows of different landmark counts in one batch (0, 1, a partial chunk, several slices), each equal to
    the same window solved alone by the oracle; the capacity beyond n_landmarks holds garbage on purpose."""
    W, lam, n_fixed, iters = 7, 1e-3, 1, 3
    counts = [0, 1, 5, 8, 13, 64, 257, 1000]
    windows = [gn_scene.make_window(1000 + k, W, max(k, 1)) for k in counts]
    for w, k in zip(windows, counts):
        if k == 0:
            w["X0"], w["obs"], w["X_gt"] = w["X0"][:0], w["obs"][:0], w["X_gt"][:0]
    gn = upload(slamx, windows, L=1024)
    gn.iterate(windows[0]["K"], lam, n_fixed, iters)
    poses, X, cost, status = gn.download()
    for i, (w, k) in enumerate(zip(windows, counts)):
        p_ref, X_ref, c_ref = oracle_run(oracle, w, lam, n_fixed, iters)
        # (a few landmarks under 7 cameras are barely constrained: the solve amplifies rounding, see above)
        np.testing.assert_allclose(poses[i], p_ref, rtol=0, atol=STEP_ATOL if k >= 64 else 1e-6,
                                   err_msg="window with %d landmarks" % k)
        np.testing.assert_allclose(X[i, :k], X_ref, rtol=0, atol=X_ATOL if k >= 64 else 1e-5)
        np.testing.assert_allclose(cost[i], c_ref, rtol=1e-7 if k >= 64 else 1e-5, atol=1e-9)
        assert not X[i, k:].any()  # landmark slots beyond n_landmarks are never touched
    assert not status.any()


def test_landmarks_that_take_no_part(slamx, oracle):
    """Never seen, seen once (lambda = 0), behind a camera: excluded exactly as the oracle excludes them."""
    w = gn_scene.make_window(42, 5, 120, drop=0.3)
    w["obs"][3] = np.nan
    w["obs"][5, 1:] = np.nan
    w["X0"][7] = [0.0, 0.0, -5.0]   # behind every camera
    w["obs"][9, 2] = [np.inf, 3.0]  # non-finite pixel: that observation is dropped
    gn = upload(slamx, [w])
    S, g, d, c = gn.normal_equations(w["K"], 0.0, 2)
    r = oracle.gn_step_dense(w["poses0"], w["X0"], w["obs"], w["K"], 0.0, 2)
    assert r["rc"] == 0 and r["active"][3] == 0 and r["active"][5] == 0 and r["active"][7] == 0 and 20 < r["active"].sum() < 118
    np.testing.assert_allclose(S[0], r["S"], rtol=0, atol=S_RTOL * np.abs(r["S"]).max())
    np.testing.assert_allclose(d[0], r["delta_c"].reshape(-1), rtol=0, atol=STEP_ATOL)
    assert c[0] == pytest.approx(r["cost"], rel=COST_RTOL)
    gn.iterate(w["K"], 0.0, 2, 1)
    poses, X, cost, status = gn.download()
    p_ref, X_ref = oracle.gn_apply(w["poses0"], w["X0"], r["delta_c"], r["delta_x"])
    np.testing.assert_allclose(X[0], X_ref, rtol=0, atol=X_ATOL)
    idle = r["active"] == 0
    np.testing.assert_array_equal(X[0][idle], w["X0"][idle])


def test_singular_window_is_skipped_loudly(slamx):
    """No damping, no held camera: the gauge is free, the reduced system is not positive definite.  The step is
    skipped (poses and landmarks untouched), the status bit is set, the other window of the batch is unaffected."""
    a = gn_scene.make_window(1, 4, 0 + 50)
    b = gn_scene.make_window(2, 4, 50)
    b["obs"][:] = np.nan  # nothing observed at all: S = 0
    gn = upload(slamx, [a, b])
    gn.iterate(a["K"], 1e-3, 1, 2)
    _, _, _, status = gn.download()
    assert not status.any()
    gn = upload(slamx, [a, b])
    gn.iterate(a["K"], 0.0, 0, 2)
    poses, X, cost, status = gn.download()
    assert status[1] == 1
    np.testing.assert_array_equal(poses[1], b["poses0"])
    np.testing.assert_array_equal(X[1], b["X0"])
    assert cost[1].tolist() == [0.0, 0.0, 0.0]


def test_skipped_step_moves_no_landmark(slamx):
    """A window WITH observations whose factorisation fails: camera 3 sees nothing and there is no damping, so its
    6 x 6 diagonal block is exactly zero and the Cholesky stops there.  slamx.h says a skipped step leaves the
    window alone -- the landmarks too, although every one of them is active (seen by cameras 0..2) and the
    back-substitution with delta_c = 0 would move it by A_l^-1 b_l."""
    a = gn_scene.make_window(7, 4, 60)
    a["obs"][:, 3, :] = np.nan
    good = gn_scene.make_window(8, 4, 60)
    gn = upload(slamx, [a, good])
    gn.iterate(a["K"], 0.0, 2, 2)  # two held cameras fix the gauge incl. scale: the other window is regular
    poses, X, cost, status = gn.download()
    assert status.tolist() == [1, 0]
    np.testing.assert_array_equal(poses[0], a["poses0"])
    np.testing.assert_array_equal(X[0], a["X0"])
    assert cost[0, 0] > 0 and cost[0, 0] == cost[0, 1] == cost[0, 2]  # same state, same cost, every iteration
    assert np.abs(X[1] - good["X0"]).max() > 0 and cost[1, -1] < cost[1, 0]  # the other window stepped


def test_runs_are_bit_identical(slamx):
    """Per-wave tiles are summed in slice order, no atomics: two runs of a many-slice batch agree bit for bit."""
    windows = [gn_scene.make_window(50 + k, 10, 700) for k in range(6)]
    out = []
    for _ in range(2):
        gn = upload(slamx, windows)
        gn.iterate(windows[0]["K"], 1e-3, 1, 3)
        out.append(gn.download())
    for x, y in zip(out[0], out[1]):
        np.testing.assert_array_equal(x, y)


def test_noisy_windows_recover_the_trajectory(slamx):
    """End to end on simulated data: 0.5 px noise, 10 % dropped observations, start 1 cm / 0.6 degrees off."""
    windows = [gn_scene.make_window(300 + k, 10, 1500) for k in range(4)]
    for w in windows:
        w["poses0"][:2] = w["poses_gt"][:2]
    gn = upload(slamx, windows)
    gn.iterate(windows[0]["K"], 1e-6, 2, 6)
    poses, X, cost, status = gn.download()
    assert not status.any()
    for i, w in enumerate(windows):
        assert cost[i, -1] < 0.02 * cost[i, 0]
        n_obs = np.isfinite(w["obs"][:, :, 0]).sum()
        assert cost[i, -1] / (2 * n_obs) < 0.5 ** 2 * 1.2  # mean squared residual ~ the noise variance
        err0 = np.abs(w["poses0"] - w["poses_gt"]).max()
        err = np.abs(poses[i] - w["poses_gt"]).max()
        assert err < 0.1 * err0 and err < 5e-3  # what 0.5 px of noise over ~1000 landmarks leaves


def test_argument_errors(slamx):
    with pytest.raises(ValueError):
        slamx.GaussNewton(1, 11, 10)  # more than SLAMX_GN_MAX_CAMS
    with pytest.raises(ValueError):
        slamx.GaussNewton(0, 4, 10)
    w = gn_scene.make_window(1, 4, 10)
    gn = upload(slamx, [w])
    with pytest.raises(ValueError):
        gn.iterate(w["K"], -1.0, 1, 1)
    with pytest.raises(ValueError):
        gn.iterate(w["K"], 0.0, 5, 1)
    with pytest.raises(ValueError):
        gn.iterate(w["K"], 0.0, 1, 65)
    with pytest.raises(ValueError):
        gn.upload(w["poses0"][None], np.zeros((1, 10, 3)), np.zeros((1, 10, 4, 2), np.float32), np.array([11], np.int32))


def test_full_size_windows_of_the_c4_workload(slamx, oracle):
    """bench.py's c4 windows themselves (10 poses x 2048 landmarks, ~13.4k observations): one step against the CPU
    statement with per-landmark elimination (the dense one would need a 6204 x 6204 Cholesky per window; the two
    CPU formulations are compared with each other at small sizes in tests/test_gn_cpu.py)."""
    W, L, lam, n_fixed = 10, 2048, 1e-4, 2
    windows = [gn_scene.make_window(0x5EED0051 * 1000 + k, W, L) for k in range(3)]
    for w in windows:
        w["poses0"][:n_fixed] = w["poses_gt"][:n_fixed]
    gn = upload(slamx, windows)
    gn.iterate(windows[0]["K"], lam, n_fixed, 1)
    poses, X, cost, status = gn.download()
    assert not status.any()
    for i, w in enumerate(windows):
        c0, p_ref, X_ref = oracle.gn_step_schur(w["poses0"], w["X0"], w["obs"], w["K"], lam, n_fixed)
        assert cost[i, 0] == pytest.approx(c0, rel=COST_RTOL)
        np.testing.assert_allclose(poses[i], p_ref, rtol=0, atol=STEP_ATOL)
        np.testing.assert_allclose(X[i], X_ref, rtol=0, atol=X_ATOL)
        assert cost[i, 1] == pytest.approx(oracle.gn_cost(p_ref, X_ref, w["obs"], w["K"], lam), rel=1e-7)
        assert cost[i, 1] < 0.05 * cost[i, 0]


def test_against_committed_golden_vectors(slamx):
    """tests/golden/gn_golden.npz: inputs and the oracle's reduced system / step of one window, committed -- the
    GPU against the fixture without the oracle in the loop (1e-9 relative on S and g, 1e-8 on the step)."""
    import os

    from conftest import ROOT

    with np.load(os.path.join(ROOT, "tests", "golden", "gn_golden.npz")) as g:
        gn = slamx.GaussNewton(1, 4, 30)
        gn.upload(g["gn_poses0"][None], g["gn_X0"][None], g["gn_obs"][None])
        S, gv, d, c = gn.normal_equations(np.array([1700.0, 1700.0, 960.0, 540.0]), 1e-3, 2)
        np.testing.assert_allclose(S[0], g["gn_S"], rtol=0, atol=S_RTOL * np.abs(g["gn_S"]).max())
        np.testing.assert_allclose(gv[0], g["gn_g"], rtol=0, atol=S_RTOL * np.abs(g["gn_g"]).max())
        np.testing.assert_allclose(d[0], g["gn_delta_c"].reshape(-1), rtol=0, atol=1e-6)
        assert c[0] == pytest.approx(float(g["gn_cost"][0]), rel=COST_RTOL)
