"""CPU checks of the Gauss-Newton oracle (oracle/gn_oracle.cpp).  The problem is build-defined
(BASELINE.json configs[3]; the reference has no counterpart), so the oracle is pinned by
mathematics instead of reference vectors: two independent formulations must agree, the analytic
step must be a descent direction that finite differences confirm, and noise-free windows must
converge to the generating poses."""
import numpy as np
import pytest

import gn_scene


def run_dense(oracle, w, lam, n_fixed, iters):
    poses, X = w["poses0"].copy(), w["X0"].copy()
    costs = []
    for _ in range(iters):
        r = oracle.gn_step_dense(poses, X, w["obs"], w["K"], lam, n_fixed)
        assert r["rc"] == 0
        costs.append(r["cost"])
        poses, X = oracle.gn_apply(poses, X, r["delta_c"], r["delta_x"])
    costs.append(oracle.gn_cost(poses, X, w["obs"], w["K"], lam))
    return poses, X, costs


@pytest.mark.parametrize("W,L", [(3, 30), (5, 40), (10, 60)])
def test_dense_and_eliminated_formulations_agree(oracle, W, L):
    # two held cameras fix the seven gauge freedoms (scale included), so the system is well conditioned
    # and the two elimination orders agree to rounding
    w = gn_scene.make_window(11 + W, W, L)
    r = oracle.gn_step_dense(w["poses0"], w["X0"], w["obs"], w["K"], 1e-3, 2)
    assert r["rc"] == 0
    pd, Xd = oracle.gn_apply(w["poses0"], w["X0"], r["delta_c"], r["delta_x"])
    c, ps, Xs = oracle.gn_step_schur(w["poses0"], w["X0"], w["obs"], w["K"], 1e-3, 2)
    assert c == pytest.approx(r["cost"], rel=1e-12)
    np.testing.assert_allclose(ps, pd, rtol=0, atol=1e-7)  # north_star asks for 1e-5 relative
    np.testing.assert_allclose(Xs, Xd, rtol=0, atol=1e-6)
    # the reduced system derived from the dense H solves to the same camera step
    d = np.linalg.solve(r["S"], r["g"])
    np.testing.assert_allclose(d, r["delta_c"].reshape(-1), rtol=0, atol=1e-7 * max(1.0, np.abs(d).max()))


def test_step_agrees_with_finite_differences(oracle):
    """b = -J^T r: for lambda -> 0 and a tiny trust region, cost(x + t delta) - cost(x) = -2 t delta^T b + O(t^2)
    with delta^T b = delta^T H delta > 0; checked with a central difference along the step."""
    w = gn_scene.make_window(3, 4, 25, drop=0.0)
    r = oracle.gn_step_dense(w["poses0"], w["X0"], w["obs"], w["K"], 0.0, 2)
    assert r["rc"] == 0

    def cost_at(t):
        p, X = oracle.gn_apply(w["poses0"], w["X0"], t * r["delta_c"], t * r["delta_x"])
        return oracle.gn_cost(p, X, w["obs"], w["K"])

    t = 1e-4
    slope = (cost_at(t) - cost_at(-t)) / (2 * t)
    # along the Gauss-Newton step: d cost / dt at 0 = 2 r^T J delta = -2 delta^T H delta, and the quadratic model
    # predicts cost(1) ~ cost(0) - delta^T H delta  =>  slope ~ -2 (cost(0) - model minimum)
    assert slope < 0
    predicted_drop = -slope / 2
    actual_drop = r["cost"] - cost_at(1.0)
    assert actual_drop == pytest.approx(predicted_drop, rel=0.05)


@pytest.mark.parametrize("W", [3, 10])
def test_noise_free_window_converges_to_the_generating_poses(oracle, W):
    w = gn_scene.make_window(5, W, 80, noise_px=0.0, drop=0.05)
    w["poses0"][:2] = w["poses_gt"][:2]  # the two held cameras carry the gauge: they must start at the truth
    poses, X, costs = run_dense(oracle, w, 0.0, 2, 6)
    # observations are float32 pixels: the residual floor is ~1e-4 px per observation
    assert costs[-1] < 1e-5 * costs[0] and costs[-1] < 1e-4
    np.testing.assert_allclose(poses, w["poses_gt"], atol=2e-5)
    part = oracle.gn_step_dense(poses, X, w["obs"], w["K"], 0.0, 2)["active"].astype(bool)
    assert part.sum() >= 0.7 * len(part)
    np.testing.assert_allclose(X[part], w["X_gt"][part], atol=2e-4)
    np.testing.assert_array_equal(X[~part], w["X0"][~part])  # landmarks that take no part are left alone


def test_unobserved_and_single_view_landmarks_take_no_part(oracle):
    w = gn_scene.make_window(9, 4, 20, drop=0.0)
    w["obs"][3] = np.nan          # never seen
    w["obs"][5, 1:] = np.nan      # one view: the 3x3 block has rank 2
    r = oracle.gn_step_dense(w["poses0"], w["X0"], w["obs"], w["K"], 0.0, 2)
    assert r["rc"] == 0
    views = np.isfinite(w["obs"][:, :, 0]).sum(axis=1)
    assert views[3] == 0 and views[5] == 1 and (views >= 2).sum() >= 8
    np.testing.assert_array_equal(r["active"], (views >= 2).astype(np.uint8))
    assert not r["delta_x"][3].any() and not r["delta_x"][5].any()
    # with damping both blocks are positive definite: the single view then moves, the unseen one has a zero gradient
    r = oracle.gn_step_dense(w["poses0"], w["X0"], w["obs"], w["K"], 1e-2, 1)
    assert r["active"].all() and not r["delta_x"][3].any() and r["delta_x"][5].any()
    # what takes no part is invisible: removing those landmarks altogether gives the same camera step
    r0 = oracle.gn_step_dense(w["poses0"], w["X0"], w["obs"], w["K"], 0.0, 2)
    keep = views >= 2
    r1 = oracle.gn_step_dense(w["poses0"], w["X0"][keep], w["obs"][keep], w["K"], 0.0, 2)
    np.testing.assert_allclose(r1["delta_c"], r0["delta_c"], rtol=0, atol=1e-12)
    assert r1["cost"] == pytest.approx(r0["cost"], rel=1e-14)
