// gn_oracle.cpp -- CPU checker for the sliding-window Gauss-Newton step (BASELINE.json configs[3]).
//
// TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
//
// PARITY UNPINNED, and more than that: NO REFERENCE COUNTERPART.  The reference has no
// Gauss-Newton, no Jacobians and no bundle adjustment (include/slam/backend/backend.hpp:10-18 and
// map.hpp:9-21 are declaration-only stubs).  BASELINE.json's north_star and
// configs[3] ask for "sliding-window Gauss-Newton J^T J (6-DoF x N landmarks) on MFMA", so the
// build defines the problem and this file states the same mathematics a
// second, independent way: the FULL normal equations of the window, dense, with NO Schur
// complement -- H = J^T J + lambda I over all 6W + 3L unknowns, solved by one dense Cholesky -- so
// that an error in the GPU's block elimination cannot hide behind an identical restatement.
// The reduced camera system (S, g) the GPU forms on the matrix cores is derived here from that
// dense H by explicit per-landmark elimination, for a direct comparison of the MFMA output.
//
// Model (the usual reprojection error of a calibrated pinhole camera):
//   p = R_c X_l + t_c,   r = (fx p.x / p.z + cx - u,  fy p.y / p.z + cy - v)
//   pose perturbation (rho, phi) acting on the left:  p' = exp(phi^) p + rho
//     => dp/drho = I, dp/dphi = -[p]x ;  dp/dX = R_c
//   an observation is used iff (u, v) are finite and p.z > 1e-9
//   H delta = b,  H = J^T J + lambda I,  b = -J^T r
//   gauge: the first n_fixed cameras are held (their rows / columns become the identity, b = 0)
//   a landmark whose 3x3 block H_ll + lambda I fails the pivot test below (never seen, or seen from
//   one place only, with lambda = 0) takes no part: its observations are left out of H, b and the cost
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Obs {
    bool valid;
    double r[2];
    double Jc[2][6];
    double Jl[2][3];
};

Obs linearize(const double* pose, const double* X, const float* uv, const double* K) {
    Obs o{};
    const double* R = pose;
    const double* t = pose + 9;
    const double u = static_cast<double>(uv[0]), v = static_cast<double>(uv[1]);
    double p[3];
    for (int i = 0; i < 3; ++i) {
        p[i] = R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2] + t[i];
    }
    o.valid = std::isfinite(u) && std::isfinite(v) && p[2] > 1e-9;
    if (!o.valid) {
        return o;
    }
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const double iz = 1.0 / p[2];
    o.r[0] = fx * p[0] * iz + cx - u;
    o.r[1] = fy * p[1] * iz + cy - v;
    const double Jp[2][3] = {{fx * iz, 0.0, -fx * p[0] * iz * iz}, {0.0, fy * iz, -fy * p[1] * iz * iz}};
    for (int a = 0; a < 2; ++a) {
        const double* j = Jp[a];
        o.Jc[a][0] = j[0];
        o.Jc[a][1] = j[1];
        o.Jc[a][2] = j[2];
        // -(j [p]x)
        o.Jc[a][3] = j[2] * p[1] - j[1] * p[2];
        o.Jc[a][4] = j[0] * p[2] - j[2] * p[0];
        o.Jc[a][5] = j[1] * p[0] - j[0] * p[1];
        for (int k = 0; k < 3; ++k) {
            o.Jl[a][k] = j[0] * R[k] + j[1] * R[3 + k] + j[2] * R[6 + k];
        }
    }
    return o;
}

// pivot test shared with the GPU: the landmark is optimised iff the Cholesky
// pivots of A = H_ll + lambda I satisfy p1 > 0, p2 > 1e-9 A11, p3 > 1e-9 A22
bool landmark_active(const double A[3][3]) {
    const double p1 = A[0][0];
    if (!(p1 > 0.0)) {
        return false;
    }
    const double l10 = A[1][0] / std::sqrt(p1);
    const double l20 = A[2][0] / std::sqrt(p1);
    const double p2 = A[1][1] - l10 * l10;
    if (!(p2 > 1e-9 * A[1][1])) {
        return false;
    }
    const double l21 = (A[2][1] - l20 * l10) / std::sqrt(p2);
    const double p3 = A[2][2] - l20 * l20 - l21 * l21;
    return p3 > 1e-9 * A[2][2];
}

bool cholesky_solve(std::vector<double>& A, std::vector<double>& b, int n) {  // A = L L^T in place (lower), b <- solution
    for (int j = 0; j < n; ++j) {
        double d = A[static_cast<size_t>(j) * n + j];
        for (int k = 0; k < j; ++k) {
            d -= A[static_cast<size_t>(j) * n + k] * A[static_cast<size_t>(j) * n + k];
        }
        if (!(d > 0.0)) {
            return false;
        }
        d = std::sqrt(d);
        A[static_cast<size_t>(j) * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[static_cast<size_t>(i) * n + j];
            for (int k = 0; k < j; ++k) {
                s -= A[static_cast<size_t>(i) * n + k] * A[static_cast<size_t>(j) * n + k];
            }
            A[static_cast<size_t>(i) * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[static_cast<size_t>(i)];
        for (int k = 0; k < i; ++k) {
            s -= A[static_cast<size_t>(i) * n + k] * b[static_cast<size_t>(k)];
        }
        b[static_cast<size_t>(i)] = s / A[static_cast<size_t>(i) * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[static_cast<size_t>(i)];
        for (int k = i + 1; k < n; ++k) {
            s -= A[static_cast<size_t>(k) * n + i] * b[static_cast<size_t>(k)];
        }
        b[static_cast<size_t>(i)] = s / A[static_cast<size_t>(i) * n + i];
    }
    return true;
}

void rodrigues(const double* phi, double* E /*3x3 row-major*/) {
    const double th2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    const double th = std::sqrt(th2);
    double a, b;  // exp(phi^) = I + a [phi]x + b [phi]x^2
    if (th < 1e-8) {
        a = 1.0 - th2 / 6.0;
        b = 0.5 - th2 / 24.0;
    } else {
        a = std::sin(th) / th;
        b = (1.0 - std::cos(th)) / th2;
    }
    const double x = phi[0], y = phi[1], z = phi[2];
    E[0] = 1.0 - b * (y * y + z * z);
    E[1] = -a * z + b * x * y;
    E[2] = a * y + b * x * z;
    E[3] = a * z + b * x * y;
    E[4] = 1.0 - b * (x * x + z * z);
    E[5] = -a * x + b * y * z;
    E[6] = -a * y + b * x * z;
    E[7] = a * x + b * y * z;
    E[8] = 1.0 - b * (x * x + y * y);
}

}  // namespace

extern "C" {

// One damped Gauss-Newton step of ONE window, full dense normal equations.
//   poses [W][12] (R row-major, t), X [L][3], obs [L][W][2] float (NaN = not observed), K = fx fy cx cy
// Outputs (any may be null): delta_c [W][6] (rho, phi), delta_x [L][3], cost (sum r^2 at the
// linearisation point), S [6W][6W] and g [6W] (reduced camera system after eliminating the
// landmarks, gauge applied), active [L] (1 = landmark optimised).  Returns 0, or 1 when the dense
// Cholesky fails.
int oracle_gn_step_dense(const double* poses, const double* X, const float* obs, int W, int L, const double* K,
                         double lambda, int n_fixed, double* delta_c, double* delta_x, double* cost, double* S, double* g,
                         uint8_t* active) {
    const int nc = 6 * W;
    const int n = nc + 3 * L;
    std::vector<double> H(static_cast<size_t>(n) * n, 0.0), b(static_cast<size_t>(n), 0.0);
    double c = 0.0;
    std::vector<uint8_t> act(static_cast<size_t>(L), 0);
    for (int l = 0; l < L; ++l) {
        // pass 1: does the landmark take part?
        double A[3][3] = {{lambda, 0, 0}, {0, lambda, 0}, {0, 0, lambda}};
        for (int w = 0; w < W; ++w) {
            const Obs o = linearize(poses + 12 * w, X + 3 * l, obs + (static_cast<size_t>(l) * W + w) * 2, K);
            if (!o.valid) {
                continue;
            }
            for (int a = 0; a < 2; ++a) {
                for (int i = 0; i < 3; ++i) {
                    for (int j = 0; j < 3; ++j) {
                        A[i][j] += o.Jl[a][i] * o.Jl[a][j];
                    }
                }
            }
        }
        act[static_cast<size_t>(l)] = landmark_active(A) ? 1 : 0;
        if (active != nullptr) {
            active[l] = act[static_cast<size_t>(l)];
        }
        if (!act[static_cast<size_t>(l)]) {
            continue;
        }
        // pass 2: its observations enter the normal equations
        for (int w = 0; w < W; ++w) {
            const Obs o = linearize(poses + 12 * w, X + 3 * l, obs + (static_cast<size_t>(l) * W + w) * 2, K);
            if (!o.valid) {
                continue;
            }
            int idx[9];
            double J[2][9];
            for (int k = 0; k < 6; ++k) {
                idx[k] = 6 * w + k;
                J[0][k] = o.Jc[0][k];
                J[1][k] = o.Jc[1][k];
            }
            for (int k = 0; k < 3; ++k) {
                idx[6 + k] = nc + 3 * l + k;
                J[0][6 + k] = o.Jl[0][k];
                J[1][6 + k] = o.Jl[1][k];
            }
            for (int a = 0; a < 2; ++a) {
                c += o.r[a] * o.r[a];
                for (int i = 0; i < 9; ++i) {
                    b[static_cast<size_t>(idx[i])] -= J[a][i] * o.r[a];
                    for (int j = 0; j < 9; ++j) {
                        H[static_cast<size_t>(idx[i]) * n + idx[j]] += J[a][i] * J[a][j];
                    }
                }
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        H[static_cast<size_t>(i) * n + i] += lambda;
    }
    if (cost != nullptr) {
        *cost = c;
    }
    // held unknowns: fixed cameras and landmarks that take no part
    std::vector<uint8_t> held(static_cast<size_t>(n), 0);
    for (int w = 0; w < W && w < n_fixed; ++w) {
        for (int k = 0; k < 6; ++k) {
            held[static_cast<size_t>(6 * w + k)] = 1;
        }
    }
    for (int l = 0; l < L; ++l) {
        if (!act[static_cast<size_t>(l)]) {
            for (int k = 0; k < 3; ++k) {
                held[static_cast<size_t>(nc + 3 * l + k)] = 1;
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        if (held[static_cast<size_t>(i)]) {
            for (int j = 0; j < n; ++j) {
                H[static_cast<size_t>(i) * n + j] = 0.0;
                H[static_cast<size_t>(j) * n + i] = 0.0;
            }
            H[static_cast<size_t>(i) * n + i] = 1.0;
            b[static_cast<size_t>(i)] = 0.0;
        }
    }
    if (S != nullptr || g != nullptr) {
        // S = H_cc - sum_l H_cl H_ll^-1 H_lc ; g = b_c - sum_l H_cl H_ll^-1 b_l   (H_ll block diagonal)
        std::vector<double> Sm(static_cast<size_t>(nc) * nc), gv(static_cast<size_t>(nc));
        for (int i = 0; i < nc; ++i) {
            gv[static_cast<size_t>(i)] = b[static_cast<size_t>(i)];
            for (int j = 0; j < nc; ++j) {
                Sm[static_cast<size_t>(i) * nc + j] = H[static_cast<size_t>(i) * n + j];
            }
        }
        for (int l = 0; l < L; ++l) {
            const int o = nc + 3 * l;
            double A[3][3], inv[3][3];
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) {
                    A[i][j] = H[static_cast<size_t>(o + i) * n + o + j];
                }
            }
            const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                               A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
            inv[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det;
            inv[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
            inv[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
            inv[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det;
            inv[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det;
            inv[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
            inv[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det;
            inv[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
            inv[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
            for (int i = 0; i < nc; ++i) {
                double ei[3];  // row i of H_cl H_ll^-1
                for (int k = 0; k < 3; ++k) {
                    ei[k] = 0.0;
                    for (int m = 0; m < 3; ++m) {
                        ei[k] += H[static_cast<size_t>(i) * n + o + m] * inv[m][k];
                    }
                }
                for (int k = 0; k < 3; ++k) {
                    gv[static_cast<size_t>(i)] -= ei[k] * b[static_cast<size_t>(o + k)];
                }
                for (int j = 0; j < nc; ++j) {
                    double s = 0.0;
                    for (int k = 0; k < 3; ++k) {
                        s += ei[k] * H[static_cast<size_t>(o + k) * n + j];
                    }
                    Sm[static_cast<size_t>(i) * nc + j] -= s;
                }
            }
        }
        if (S != nullptr) {
            std::memcpy(S, Sm.data(), Sm.size() * sizeof(double));
        }
        if (g != nullptr) {
            std::memcpy(g, gv.data(), gv.size() * sizeof(double));
        }
    }
    if (!cholesky_solve(H, b, n)) {
        return 1;
    }
    if (delta_c != nullptr) {
        std::memcpy(delta_c, b.data(), sizeof(double) * nc);
    }
    if (delta_x != nullptr) {
        std::memcpy(delta_x, b.data() + nc, sizeof(double) * 3 * L);
    }
    return 0;
}

// Apply a step: R <- exp(phi^) R, t <- exp(phi^) t + rho ; X <- X + delta_x.
void oracle_gn_apply(double* poses, double* X, int W, int L, const double* delta_c, const double* delta_x) {
    for (int w = 0; w < W; ++w) {
        double E[9], Rn[9], tn[3];
        double* R = poses + 12 * w;
        double* t = R + 9;
        rodrigues(delta_c + 6 * w + 3, E);
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                Rn[3 * i + j] = E[3 * i] * R[j] + E[3 * i + 1] * R[3 + j] + E[3 * i + 2] * R[6 + j];
            }
            tn[i] = E[3 * i] * t[0] + E[3 * i + 1] * t[1] + E[3 * i + 2] * t[2] + delta_c[6 * w + i];
        }
        std::memcpy(R, Rn, sizeof(Rn));
        std::memcpy(t, tn, sizeof(tn));
    }
    for (int i = 0; i < 3 * L; ++i) {
        X[i] += delta_x[i];
    }
}

// Sum of squared residuals over the landmarks that take part (pivot test with the given lambda).
double oracle_gn_cost(const double* poses, const double* X, const float* obs, int W, int L, const double* K, double lambda) {
    double c = 0.0;
    for (int l = 0; l < L; ++l) {
        double A[3][3] = {{lambda, 0, 0}, {0, lambda, 0}, {0, 0, lambda}};
        double cl = 0.0;
        for (int w = 0; w < W; ++w) {
            const Obs o = linearize(poses + 12 * w, X + 3 * l, obs + (static_cast<size_t>(l) * W + w) * 2, K);
            if (o.valid) {
                cl += o.r[0] * o.r[0] + o.r[1] * o.r[1];
                for (int a = 0; a < 2; ++a) {
                    for (int i = 0; i < 3; ++i) {
                        for (int j = 0; j < 3; ++j) {
                            A[i][j] += o.Jl[a][i] * o.Jl[a][j];
                        }
                    }
                }
            }
        }
        if (landmark_active(A)) {
            c += cl;
        }
    }
    return c;
}

// The same step organised the way a CPU bundle adjuster does it (per-landmark elimination into
// the reduced camera system, 6W x 6W Cholesky, back-substitution): O(L W^2) instead of
// O((6W + 3L)^3).  Used as the CPU baseline of bench.py's c4 workload and cross-checked against
// oracle_gn_step_dense in tests/test_cpu.py.  Updates poses and X in place; returns the cost at
// the linearisation point (negative when the reduced system is not positive definite).
double oracle_gn_step_schur(double* poses, double* X, const float* obs, int W, int L, const double* K, double lambda,
                            int n_fixed) {
    const int nc = 6 * W;
    std::vector<double> S(static_cast<size_t>(nc) * nc, 0.0), g(static_cast<size_t>(nc), 0.0);
    std::vector<Obs> O(static_cast<size_t>(W));
    struct Lm {
        bool active;
        double A[3][3];
        double bl[3];
    };
    std::vector<Lm> lm(static_cast<size_t>(L));
    double c = 0.0;
    auto inv3 = [](const double A[3][3], double inv[3][3]) {
        const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                           A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
        inv[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det;
        inv[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
        inv[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
        inv[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det;
        inv[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det;
        inv[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
        inv[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det;
        inv[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
        inv[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
    };
    for (int l = 0; l < L; ++l) {
        Lm& m = lm[static_cast<size_t>(l)];
        std::memset(&m, 0, sizeof(m));
        for (int w = 0; w < W; ++w) {
            Obs& o = O[static_cast<size_t>(w)];
            o = linearize(poses + 12 * w, X + 3 * l, obs + (static_cast<size_t>(l) * W + w) * 2, K);
            if (!o.valid) {
                continue;
            }
            for (int a = 0; a < 2; ++a) {
                for (int i = 0; i < 3; ++i) {
                    m.bl[i] -= o.Jl[a][i] * o.r[a];
                    for (int j = 0; j < 3; ++j) {
                        m.A[i][j] += o.Jl[a][i] * o.Jl[a][j];
                    }
                }
            }
        }
        for (int i = 0; i < 3; ++i) {
            m.A[i][i] += lambda;
        }
        m.active = landmark_active(m.A);
        if (!m.active) {
            continue;  // takes no part: nothing of it enters S, g or the cost
        }
        for (int w = 0; w < W; ++w) {
            const Obs& o = O[static_cast<size_t>(w)];
            if (!o.valid) {
                continue;
            }
            for (int a = 0; a < 2; ++a) {
                c += o.r[a] * o.r[a];
                for (int i = 0; i < 6; ++i) {
                    g[static_cast<size_t>(6 * w + i)] -= o.Jc[a][i] * o.r[a];
                    for (int j = 0; j < 6; ++j) {
                        S[static_cast<size_t>(6 * w + i) * nc + 6 * w + j] += o.Jc[a][i] * o.Jc[a][j];
                    }
                }
            }
        }
        double inv[3][3];
        inv3(m.A, inv);
        // E_w = Jc^T Jl (6x3);  S -= E_w inv E_v^T ; g -= E_w inv bl
        std::vector<double> E(static_cast<size_t>(W) * 18, 0.0), EI(static_cast<size_t>(W) * 18, 0.0);
        for (int w = 0; w < W; ++w) {
            const Obs& o = O[static_cast<size_t>(w)];
            if (!o.valid) {
                continue;
            }
            for (int i = 0; i < 6; ++i) {
                for (int k = 0; k < 3; ++k) {
                    E[static_cast<size_t>(w) * 18 + 3 * i + k] = o.Jc[0][i] * o.Jl[0][k] + o.Jc[1][i] * o.Jl[1][k];
                }
            }
            for (int i = 0; i < 6; ++i) {
                for (int k = 0; k < 3; ++k) {
                    double s = 0.0;
                    for (int q = 0; q < 3; ++q) {
                        s += E[static_cast<size_t>(w) * 18 + 3 * i + q] * inv[q][k];
                    }
                    EI[static_cast<size_t>(w) * 18 + 3 * i + k] = s;
                }
            }
        }
        for (int w = 0; w < W; ++w) {
            if (!O[static_cast<size_t>(w)].valid) {
                continue;
            }
            for (int i = 0; i < 6; ++i) {
                const double* ei = &EI[static_cast<size_t>(w) * 18 + 3 * i];
                g[static_cast<size_t>(6 * w + i)] -= ei[0] * m.bl[0] + ei[1] * m.bl[1] + ei[2] * m.bl[2];
                for (int v = 0; v < W; ++v) {
                    if (!O[static_cast<size_t>(v)].valid) {
                        continue;
                    }
                    for (int j = 0; j < 6; ++j) {
                        const double* ej = &E[static_cast<size_t>(v) * 18 + 3 * j];
                        S[static_cast<size_t>(6 * w + i) * nc + 6 * v + j] -= ei[0] * ej[0] + ei[1] * ej[1] + ei[2] * ej[2];
                    }
                }
            }
        }
    }
    for (int i = 0; i < nc; ++i) {
        S[static_cast<size_t>(i) * nc + i] += lambda;
    }
    for (int i = 0; i < nc && i < 6 * n_fixed; ++i) {
        for (int j = 0; j < nc; ++j) {
            S[static_cast<size_t>(i) * nc + j] = 0.0;
            S[static_cast<size_t>(j) * nc + i] = 0.0;
        }
        S[static_cast<size_t>(i) * nc + i] = 1.0;
        g[static_cast<size_t>(i)] = 0.0;
    }
    if (!cholesky_solve(S, g, nc)) {
        return -1.0;
    }
    // back-substitution, against the OLD poses
    std::vector<double> dx(static_cast<size_t>(L) * 3, 0.0);
    for (int l = 0; l < L; ++l) {
        const Lm& m = lm[static_cast<size_t>(l)];
        if (!m.active) {
            continue;
        }
        double rhs[3] = {m.bl[0], m.bl[1], m.bl[2]};
        for (int w = 0; w < W; ++w) {
            const Obs o = linearize(poses + 12 * w, X + 3 * l, obs + (static_cast<size_t>(l) * W + w) * 2, K);
            if (!o.valid) {
                continue;
            }
            for (int k = 0; k < 3; ++k) {
                for (int i = 0; i < 6; ++i) {
                    rhs[k] -= (o.Jc[0][i] * o.Jl[0][k] + o.Jc[1][i] * o.Jl[1][k]) * g[static_cast<size_t>(6 * w + i)];
                }
            }
        }
        double inv[3][3];
        inv3(m.A, inv);
        for (int k = 0; k < 3; ++k) {
            dx[static_cast<size_t>(3 * l + k)] = inv[k][0] * rhs[0] + inv[k][1] * rhs[1] + inv[k][2] * rhs[2];
        }
    }
    oracle_gn_apply(poses, X, W, L, g.data(), dx.data());
    return c;
}

}  // extern "C"
