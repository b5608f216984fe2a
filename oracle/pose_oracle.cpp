// pose_oracle.cpp -- TEST INFRASTRUCTURE ONLY (part of the oracle, see slam_oracle.cpp).
//
// CPU restatement of the small dense pose solves, SURVEY.md 8a rows a11-a14:
//   a11  LoopClosure::solvePnP                 src/backend/loop_closure.cpp:238-274
//   a12  RANSAC inlier scoring                 src/backend/loop_closure.cpp:201-215
//   a13  decomposeEssential / triangulateSimple / simpleRecoverPose
//                                              src/frontend/simple_pose_recover.cpp:6-98
//   a14  slam::triangulate + triangulatePoints include/slam/common/common.hpp:201-221,
//                                              src/frontend/pose_estimator.cpp:69-104
//
// PARITY UNPINNED: the reference delegates the arithmetic of these rows to third-party SVDs
// (Eigen 3.4.0 JacobiSVD at loop_closure.cpp:257,265; OpenCV 4.12 cv::SVD at
// simple_pose_recover.cpp:7,29 and common.hpp:215), neither present in this image, and no
// reference test pins a value (test_pose_estimator.cpp:168 only checks R^T R = I to 1e-6).  The
// SVD here is a one-sided Jacobi (Hestenes) in double with singular values sorted descending, as
// both libraries order them; singular vectors are unique only up to a common sign per pair, so
// consumers compare sign-invariant quantities (see tests/test_pose_gpu.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

namespace {

// A (m x n, row-major, m >= n) = U diag(S) V^T.  U: m x n, V: n x n, S descending.
void svd_jacobi(int m, int n, const double* A, double* U, double* S, double* V) {
    std::vector<double> W(A, A + static_cast<size_t>(m) * n);
    std::vector<double> Vt(static_cast<size_t>(n) * n, 0.0);  // V stored row-major
    for (int i = 0; i < n; ++i) {
        Vt[static_cast<size_t>(i) * n + i] = 1.0;
    }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int k = 0; k < m; ++k) {
                    const double a = W[static_cast<size_t>(k) * n + p];
                    const double b = W[static_cast<size_t>(k) * n + q];
                    alpha += a * a;
                    beta += b * b;
                    gamma += a * b;
                }
                if (gamma == 0.0) {
                    continue;
                }
                off = std::max(off, std::fabs(gamma) / std::sqrt(std::max(alpha * beta, 1e-300)));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t);
                const double s = c * t;
                for (int k = 0; k < m; ++k) {
                    const double a = W[static_cast<size_t>(k) * n + p];
                    const double b = W[static_cast<size_t>(k) * n + q];
                    W[static_cast<size_t>(k) * n + p] = c * a - s * b;
                    W[static_cast<size_t>(k) * n + q] = s * a + c * b;
                }
                for (int k = 0; k < n; ++k) {
                    const double a = Vt[static_cast<size_t>(k) * n + p];
                    const double b = Vt[static_cast<size_t>(k) * n + q];
                    Vt[static_cast<size_t>(k) * n + p] = c * a - s * b;
                    Vt[static_cast<size_t>(k) * n + q] = s * a + c * b;
                }
            }
        }
        if (off < 1e-15) {
            break;
        }
    }
    std::vector<double> sv(static_cast<size_t>(n));
    std::vector<int> order(static_cast<size_t>(n));
    for (int j = 0; j < n; ++j) {
        double nn = 0.0;
        for (int k = 0; k < m; ++k) {
            nn += W[static_cast<size_t>(k) * n + j] * W[static_cast<size_t>(k) * n + j];
        }
        sv[static_cast<size_t>(j)] = std::sqrt(nn);
        order[static_cast<size_t>(j)] = j;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sv[static_cast<size_t>(a)] > sv[static_cast<size_t>(b)]; });
    for (int jj = 0; jj < n; ++jj) {
        const int j = order[static_cast<size_t>(jj)];
        S[jj] = sv[static_cast<size_t>(j)];
        for (int k = 0; k < m; ++k) {
            U[static_cast<size_t>(k) * n + jj] = S[jj] > 0.0 ? W[static_cast<size_t>(k) * n + j] / S[jj] : 0.0;
        }
        for (int k = 0; k < n; ++k) {
            V[static_cast<size_t>(k) * n + jj] = Vt[static_cast<size_t>(k) * n + j];
        }
    }
}

double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
void mul3(const double* A, const double* B, double* C) {  // C = A B
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) {
                s += A[i * 3 + k] * B[k * 3 + j];
            }
            C[i * 3 + j] = s;
        }
    }
}
// full 3x3 SVD with an orthonormal U even for rank-deficient input (U completed by cross products)
void svd3(const double* M, double* U, double* S, double* V) {
    svd_jacobi(3, 3, M, U, S, V);
    auto col = [&](double* Q, int j, double* out) {
        out[0] = Q[j];
        out[1] = Q[3 + j];
        out[2] = Q[6 + j];
    };
    if (S[2] <= 1e-14 * std::max(S[0], 1e-300)) {
        double u0[3], u1[3];
        col(U, 0, u0);
        col(U, 1, u1);
        const double u2[3] = {u0[1] * u1[2] - u0[2] * u1[1], u0[2] * u1[0] - u0[0] * u1[2], u0[0] * u1[1] - u0[1] * u1[0]};
        U[2] = u2[0];
        U[5] = u2[1];
        U[8] = u2[2];
    }
}

// simple_pose_recover.cpp:21-32 triangulateSimple; X = vt.row(3) / vt(3,3)
void triangulate_simple(const double* P1, const double* P2, float x1, float y1, float x2, float y2, double* X) {
    double A[16];
    for (int j = 0; j < 4; ++j) {
        A[0 * 4 + j] = static_cast<double>(x1) * P1[8 + j] - P1[0 + j];
        A[1 * 4 + j] = static_cast<double>(y1) * P1[8 + j] - P1[4 + j];
        A[2 * 4 + j] = static_cast<double>(x2) * P2[8 + j] - P2[0 + j];
        A[3 * 4 + j] = static_cast<double>(y2) * P2[8 + j] - P2[4 + j];
    }
    double U[16], S[4], V[16];
    svd_jacobi(4, 4, A, U, S, V);
    for (int k = 0; k < 4; ++k) {
        X[k] = V[k * 4 + 3];
    }
}

}  // namespace

extern "C" {

int oracle_pose_rows_available(void) { return 1; }

// generic SVD access for the tests (row-major)
void oracle_svd(int m, int n, const double* A, double* U, double* S, double* V) { svd_jacobi(m, n, A, U, S, V); }

// Second half of solvePnP, loop_closure.cpp:260-271, from a GIVEN null vector p (12 doubles): the column-major re-map,
// R = U diag(1, 1, det(U V^T)) V^T of R0's SVD, t = t0 / ||R0||_F.  The reference's result is not covariant with the
// sign of p (R of -p is neither R nor -R), and Eigen's JacobiSVD leaves that sign unpinned, so the
// tests run this half on whichever sign the device's null vector has and compare R, t for EVERY hypothesis.
int oracle_pnp_from_null_vector(const double* p, double* R, double* t) {
    // :260  P = Map<Matrix<double,3,4>>(p.data()) -- COLUMN-major map of the row-major DLT vector
    double P[12];  // row-major 3x4
    for (int c = 0; c < 4; ++c) {
        for (int r = 0; r < 3; ++r) {
            P[r * 4 + c] = p[c * 3 + r];
        }
    }
    double R0[9], t0[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            R0[r * 3 + c] = P[r * 4 + c];
        }
        t0[r] = P[r * 4 + 3];
    }
    double Ur[9], Sr[3], Vr[9];
    svd3(R0, Ur, Sr, Vr);
    double Vt[9], UVt[9];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            Vt[i * 3 + j] = Vr[j * 3 + i];
        }
    }
    mul3(Ur, Vt, UVt);
    const double det = det3(UVt);
    double D[9] = {1, 0, 0, 0, 1, 0, 0, 0, det};
    double UD[9];
    mul3(Ur, D, UD);
    mul3(UD, Vt, R);
    double fro = 0.0;
    for (double x : R0) {
        fro += x * x;
    }
    fro = std::sqrt(fro);  // R.norm() == Frobenius norm
    for (int k = 0; k < 3; ++k) {
        t[k] = t0[k] / fro;
    }
    return 1;
}

// a11 solvePnP, loop_closure.cpp:238-274.  p3d: n x 3, p2d: n x 2.  Returns 0 if n < 6.
// nullvec (optional, 12 doubles) receives the DLT null vector p before the column-major re-map.
int oracle_solve_pnp(int n, const double* p3d, const double* p2d, double* R, double* t, double* nullvec) {
    if (n < 6) {
        return 0;
    }
    std::vector<double> A(static_cast<size_t>(2 * n) * 12, 0.0);
    for (int i = 0; i < n; ++i) {
        const double X = p3d[3 * i], Y = p3d[3 * i + 1], Z = p3d[3 * i + 2];
        const double u = p2d[2 * i], v = p2d[2 * i + 1];
        const double r0[12] = {X, Y, Z, 1, 0, 0, 0, 0, -u * X, -u * Y, -u * Z, -u};
        const double r1[12] = {0, 0, 0, 0, X, Y, Z, 1, -v * X, -v * Y, -v * Z, -v};
        std::memcpy(&A[static_cast<size_t>(2 * i) * 12], r0, sizeof(r0));
        std::memcpy(&A[static_cast<size_t>(2 * i + 1) * 12], r1, sizeof(r1));
    }
    std::vector<double> U(static_cast<size_t>(2 * n) * 12), S(12), V(144);
    svd_jacobi(2 * n, 12, A.data(), U.data(), S.data(), V.data());
    double p[12];
    for (int k = 0; k < 12; ++k) {
        p[k] = V[static_cast<size_t>(k) * 12 + 11];  // svd.matrixV().col(11)
    }
    if (nullvec != nullptr) {
        std::memcpy(nullvec, p, sizeof(p));
    }
    return oracle_pnp_from_null_vector(p, R, t);
}

// SLAMX_GEOMETRY_CALIBRATED (include/slamx.h) -- BUILD-DEFINED, no reference counterpart, PARITY UNPINNED: what solvePnP
// becomes when its three defects are removed (hard part 6): the DLT on K^-1-normalised image points, the
// ROW-major 3x4 map of the DLT vector, the null vector's sign chosen so that det(R0) > 0, the same orthonormalisation,
// t = t0 / mean singular value of R0.  From a GIVEN null vector (the sign handling makes the result covariant, but the
// tests feed the device's vector anyway so that only this half is compared here).
int oracle_pnp_calibrated_from_null_vector(const double* p, double* R, double* t) {
    double R0[9], t0[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            R0[r * 3 + c] = p[r * 4 + c];
        }
        t0[r] = p[r * 4 + 3];
    }
    if (det3(R0) < 0.0) {
        for (double& x : R0) {
            x = -x;
        }
        for (double& x : t0) {
            x = -x;
        }
    }
    double Ur[9], Sr[3], Vr[9];
    svd3(R0, Ur, Sr, Vr);
    double Vt[9], UVt[9];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            Vt[i * 3 + j] = Vr[j * 3 + i];
        }
    }
    mul3(Ur, Vt, UVt);
    const double det = det3(UVt);
    double D[9] = {1, 0, 0, 0, 1, 0, 0, 0, det};
    double UD[9];
    mul3(Ur, D, UD);
    mul3(UD, Vt, R);
    const double scale = (Sr[0] + Sr[1] + Sr[2]) / 3.0;
    for (int k = 0; k < 3; ++k) {
        t[k] = t0[k] / scale;
    }
    return 1;
}

// The whole calibrated solve on n >= 6 points: p2d in pixels, K upper triangular with K[8] = 1.
int oracle_solve_pnp_calibrated(int n, const double* p3d, const double* p2d, const double* K, double* R, double* t,
                                double* nullvec) {
    if (n < 6) {
        return 0;
    }
    std::vector<double> A(static_cast<size_t>(2 * n) * 12, 0.0);
    for (int i = 0; i < n; ++i) {
        const double X = p3d[3 * i], Y = p3d[3 * i + 1], Z = p3d[3 * i + 2];
        const double v = (p2d[2 * i + 1] - K[5]) / K[4];
        const double u = (p2d[2 * i] - K[2] - K[1] * v) / K[0];
        const double r0[12] = {X, Y, Z, 1, 0, 0, 0, 0, -u * X, -u * Y, -u * Z, -u};
        const double r1[12] = {0, 0, 0, 0, X, Y, Z, 1, -v * X, -v * Y, -v * Z, -v};
        std::memcpy(&A[static_cast<size_t>(2 * i) * 12], r0, sizeof(r0));
        std::memcpy(&A[static_cast<size_t>(2 * i + 1) * 12], r1, sizeof(r1));
    }
    std::vector<double> U(static_cast<size_t>(2 * n) * 12), S(12), V(144);
    svd_jacobi(2 * n, 12, A.data(), U.data(), S.data(), V.data());
    double p[12];
    for (int k = 0; k < 12; ++k) {
        p[k] = V[static_cast<size_t>(k) * 12 + 11];
    }
    if (nullvec != nullptr) {
        std::memcpy(nullvec, p, sizeof(p));
    }
    return oracle_pnp_calibrated_from_null_vector(p, R, t);
}

// a12 inlier count of one (R,t), loop_closure.cpp:201-215
int oracle_count_inliers(int n, const double* p3d, const double* p2d, const double* K, const double* R, const double* t,
                         double threshold) {
    int inliers = 0;
    for (int j = 0; j < n; ++j) {
        double tp[3];
        for (int r = 0; r < 3; ++r) {
            tp[r] = R[r * 3] * p3d[3 * j] + R[r * 3 + 1] * p3d[3 * j + 1] + R[r * 3 + 2] * p3d[3 * j + 2] + t[r];
        }
        if (tp[2] <= 0) {
            continue;
        }
        const double q[3] = {tp[0] / tp[2], tp[1] / tp[2], tp[2] / tp[2]};
        const double px = K[0] * q[0] + K[1] * q[1] + K[2] * q[2];
        const double py = K[3] * q[0] + K[4] * q[1] + K[5] * q[2];
        const double ex = p2d[2 * j] - px;
        const double ey = p2d[2 * j + 1] - py;
        if (std::sqrt(ex * ex + ey * ey) < threshold) {
            inliers++;
        }
    }
    return inliers;
}

// a13 decomposeEssential, simple_pose_recover.cpp:6-18
void oracle_decompose_essential(const double* E, double* R1, double* R2, double* t) {
    double U[9], S[3], V[9];
    svd3(E, U, S, V);
    const double W[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    const double Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    double Vt[9], UW[9];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            Vt[i * 3 + j] = V[j * 3 + i];
        }
    }
    mul3(U, W, UW);
    mul3(UW, Vt, R1);
    mul3(U, Wt, UW);
    mul3(UW, Vt, R2);
    t[0] = U[2];
    t[1] = U[5];
    t[2] = U[8];
    if (det3(R1) < 0) {
        for (int k = 0; k < 9; ++k) {
            R1[k] = -R1[k];
        }
    }
    if (det3(R2) < 0) {
        for (int k = 0; k < 9; ++k) {
            R2[k] = -R2[k];
        }
    }
}

// a13 simpleRecoverPose, simple_pose_recover.cpp:35-98.  pts: n x 2 float (normalised points, as
// estimate() passes them, pose_estimator.cpp:59-66).  front (optional, 4 ints) gets the counts.
void oracle_recover_pose(const double* E, const float* pts1, const float* pts2, int n, const double* K, double* R,
                         double* t, int* front_out) {
    double R1[9], R2[9], t_[3];
    oracle_decompose_essential(E, R1, R2, t_);
    const double* Rs[4] = {R1, R2, R1, R2};
    const double sg[4] = {1, 1, -1, -1};
    double KP0[12];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 4; ++c) {
            KP0[r * 4 + c] = c < 3 ? K[r * 3 + c] : 0.0;  // K * [I|0]
        }
    }
    int best = 0, maxFront = -1;
    for (int i = 0; i < 4; ++i) {
        double P[12], KP[12];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) {
                P[r * 4 + c] = Rs[i][r * 3 + c];
            }
            P[r * 4 + 3] = sg[i] * t_[r];
        }
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 4; ++c) {
                KP[r * 4 + c] = K[r * 3] * P[c] + K[r * 3 + 1] * P[4 + c] + K[r * 3 + 2] * P[8 + c];
            }
        }
        int front = 0;
        for (int j = 0; j < n; ++j) {
            double X[4];
            triangulate_simple(KP0, KP, pts1[2 * j], pts1[2 * j + 1], pts2[2 * j], pts2[2 * j + 1], X);
            // "X / X.at<double>(3,0)" is a cv::MatExpr scale by the reciprocal (third-party; only the
            // signs of z1, z2 are consumed)
            const double inv = 1.0 / X[3];
            for (int k = 0; k < 4; ++k) {
                X[k] = X[k] * inv;
            }
            const double z1 = X[2];
            const double z2 = KP[8] * X[0] + KP[9] * X[1] + KP[10] * X[2] + KP[11] * X[3];
            if (z1 > 0 && z2 > 0) {
                front++;
            }
        }
        if (front_out != nullptr) {
            front_out[i] = front;
        }
        if (front > maxFront) {
            maxFront = front;
            best = i;
        }
    }
    std::memcpy(R, Rs[best], sizeof(double) * 9);
    for (int k = 0; k < 3; ++k) {
        t[k] = sg[best] * t_[k];
    }
}

// a14 slam::triangulate (common.hpp:201-221) + the conversion loop of triangulatePoints
// (pose_estimator.cpp:94-101).  P1, P2: 3x4 row-major; pts: n x 2 float (pixel coordinates).
// x4 (optional): n x 4 float homogeneous (the CV_32F matrix, column i = point i, BEFORE the /= w);
// xyz: n x 3 double.
void oracle_triangulate(const double* P1, const double* P2, const float* pts1, const float* pts2, int n, float* x4,
                        double* xyz) {
    for (int i = 0; i < n; ++i) {
        double X[4];
        triangulate_simple(P1, P2, pts1[2 * i], pts1[2 * i + 1], pts2[2 * i], pts2[2 * i + 1], X);
        float xf[4];
        for (int k = 0; k < 4; ++k) {
            xf[k] = static_cast<float>(X[k]);  // copyTo a CV_32F column
            if (x4 != nullptr) {
                x4[4 * i + k] = xf[k];
            }
        }
        // "x /= x.at<float>(3,0)" on a CV_32F Mat is convertTo(x, type, 1.0 / w): every element is
        // multiplied in float by float(1.0 / double(w)) (OpenCV mat.inl.hpp; third-party, unpinned)
        const float alpha = static_cast<float>(1.0 / static_cast<double>(xf[3]));
        for (int k = 0; k < 3; ++k) {
            xyz[3 * i + k] = static_cast<double>(xf[k] * alpha);
        }
    }
}

// f3: the essential-matrix RANSAC of this build (cv::findEssentialMat, pose_estimator.cpp:37-47, is
// third-party: PARITY UNPINNED BY THE REFERENCE).  CPU restatement of the algorithm the HIP path
// runs -- same sampling, same conditioning, null vector by the one-sided Jacobi SVD, same Sampson
// test -- used to check the GPU implementation, not to pin it against OpenCV.
// pts: n x 2 float pixel coordinates.  E_all (optional): iterations x 9, counts_all (optional).
// The loop with the sample table GIVEN (idx: iterations x 8 indices): what oracle_essential_ransac runs after drawing
// its table, and what the batched GPU stage -- which draws its tables on the device -- is compared with.
int oracle_essential_from_samples(const float* pts1, const float* pts2, int n, const double* K, const int* table, int iterations,
                                  double threshold_px, double* E, int* n_inliers, unsigned char* mask, double* E_all,
                                  int* counts_all) {
    if (n < 8 || iterations <= 0) {
        return -1;
    }
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    std::vector<double> x1(static_cast<size_t>(n) * 2), x2(static_cast<size_t>(n) * 2);
    for (int i = 0; i < n; ++i) {
        x1[2 * i] = (static_cast<double>(pts1[2 * i]) - cx) / fx;
        x1[2 * i + 1] = (static_cast<double>(pts1[2 * i + 1]) - cy) / fy;
        x2[2 * i] = (static_cast<double>(pts2[2 * i]) - cx) / fx;
        x2[2 * i + 1] = (static_cast<double>(pts2[2 * i + 1]) - cy) / fy;
    }
    const double thr = threshold_px / (0.5 * (fx + fy));
    const double thr2 = thr * thr;
    int best = -1, best_count = -1;
    std::vector<double> Ebest(9, 0.0);
    double last_cost = 0.0;  // MSAC cost of the model `count` saw last: sum of min(d^2, thr^2), d = Sampson distance
    auto count = [&](const double* e, unsigned char* flags) {
        int c = 0;
        last_cost = 0.0;
        for (int i = 0; i < n; ++i) {
            const double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
            // chains of fused multiply-adds, innermost term first: the estimator is the build's own, and this is how it
            // defines the Sampson test (18 FP64 operations on the GPU instead of 33)
            const double ex0 = std::fma(e[0], u1, std::fma(e[1], v1, e[2]));
            const double ex1 = std::fma(e[3], u1, std::fma(e[4], v1, e[5]));
            const double ex2 = std::fma(e[6], u1, std::fma(e[7], v1, e[8]));
            const double et0 = std::fma(e[0], u2, std::fma(e[3], v2, e[6]));
            const double et1 = std::fma(e[1], u2, std::fma(e[4], v2, e[7]));
            const double r = std::fma(u2, ex0, std::fma(v2, ex1, ex2));
            const double den = std::fma(ex0, ex0, std::fma(ex1, ex1, std::fma(et0, et0, et1 * et1)));
            const bool in = den > 0.0 && r * r < thr2 * den;
            c += in ? 1 : 0;
            last_cost += in ? (r * r) / den : thr2;
            if (flags != nullptr) {
                flags[i] = in ? 1 : 0;
            }
        }
        return c;
    };
    for (int it = 0; it < iterations; ++it) {
        const int* idx = table + static_cast<size_t>(it) * 8;
        double c1x = 0, c1y = 0, c2x = 0, c2y = 0;
        for (int i = 0; i < 8; ++i) {
            c1x += x1[2 * idx[i]];
            c1y += x1[2 * idx[i] + 1];
            c2x += x2[2 * idx[i]];
            c2y += x2[2 * idx[i] + 1];
        }
        c1x /= 8.0;
        c1y /= 8.0;
        c2x /= 8.0;
        c2y /= 8.0;
        double d1 = 0, d2 = 0;
        for (int i = 0; i < 8; ++i) {
            const double ux = x1[2 * idx[i]] - c1x, uy = x1[2 * idx[i] + 1] - c1y;
            const double vx = x2[2 * idx[i]] - c2x, vy = x2[2 * idx[i] + 1] - c2y;
            d1 += std::sqrt(ux * ux + uy * uy);
            d2 += std::sqrt(vx * vx + vy * vy);
        }
        const double s1 = d1 > 0.0 ? std::sqrt(2.0) * 8.0 / d1 : 1.0;
        const double s2 = d2 > 0.0 ? std::sqrt(2.0) * 8.0 / d2 : 1.0;
        double A[9 * 9] = {0};  // 8 constraint rows + a zero row: the one-sided Jacobi wants m >= n
        for (int r = 0; r < 8; ++r) {
            const double p1[3] = {(x1[2 * idx[r]] - c1x) * s1, (x1[2 * idx[r] + 1] - c1y) * s1, 1.0};
            const double p2[3] = {(x2[2 * idx[r]] - c2x) * s2, (x2[2 * idx[r] + 1] - c2y) * s2, 1.0};
            for (int c = 0; c < 9; ++c) {
                A[r * 9 + c] = p2[c / 3] * p1[c % 3];
            }
        }
        double U[81], S[9], V[81];
        svd_jacobi(9, 9, A, U, S, V);
        double En[9];
        for (int k = 0; k < 9; ++k) {
            En[k] = V[k * 9 + 8];
        }
        const double T1[9] = {s1, 0, -s1 * c1x, 0, s1, -s1 * c1y, 0, 0, 1};
        const double T2t[9] = {s2, 0, 0, 0, s2, 0, -s2 * c2x, -s2 * c2y, 1};
        double tmp[9], E0[9], U3[9], S3[3], V3[9], Vt3[9], UD[9], Ef[9];
        mul3(En, T1, tmp);
        mul3(T2t, tmp, E0);
        svd3(E0, U3, S3, V3);
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                Vt3[i * 3 + j] = V3[j * 3 + i];
            }
        }
        const double D[9] = {1, 0, 0, 0, 1, 0, 0, 0, 0};
        mul3(U3, D, UD);
        mul3(UD, Vt3, Ef);
        const int c = count(Ef, nullptr);
        if (E_all != nullptr) {
            std::copy(Ef, Ef + 9, E_all + static_cast<size_t>(it) * 9);
        }
        if (counts_all != nullptr) {
            counts_all[it] = c;
        }
        if (c > best_count) {
            best_count = c;
            best = it;
            std::copy(Ef, Ef + 9, Ebest.begin());
        }
    }
    // refit on the inliers of the winner (normal equations of the conditioned 8-point system), kept iff its MSAC cost
    // is not higher than the winner's (round 5; until then: iff it had at least as many inliers -- a count contest that an
    // accurate refit loses by the handful of wrong matches lying near the winner's epipolar lines)
    std::vector<unsigned char> flags(static_cast<size_t>(n)), flags2(static_cast<size_t>(n));
    count(Ebest.data(), flags.data());
    double best_cost = last_cost;
    constexpr int kRefitRounds = 1;  // POSE_REFIT_ROUNDS (slam-cin0051_amd/csrc/slamx_internal.h)
    for (int round = 0; round < kRefitRounds; ++round) {
    double Efit[9];
    std::copy(Ebest.begin(), Ebest.end(), Efit);
    if (best_count >= 8) {
        double c1x = 0, c1y = 0, c2x = 0, c2y = 0, cnt = 0;
        for (int i = 0; i < n; ++i) {
            if (flags[i]) {
                c1x += x1[2 * i];
                c1y += x1[2 * i + 1];
                c2x += x2[2 * i];
                c2y += x2[2 * i + 1];
                cnt += 1.0;
            }
        }
        c1x /= cnt;
        c1y /= cnt;
        c2x /= cnt;
        c2y /= cnt;
        double d1 = 0, d2 = 0;
        for (int i = 0; i < n; ++i) {
            if (flags[i]) {
                const double ux = x1[2 * i] - c1x, uy = x1[2 * i + 1] - c1y;
                const double vx = x2[2 * i] - c2x, vy = x2[2 * i + 1] - c2y;
                d1 += std::sqrt(ux * ux + uy * uy);
                d2 += std::sqrt(vx * vx + vy * vy);
            }
        }
        const double s1 = d1 > 0.0 ? std::sqrt(2.0) * cnt / d1 : 1.0;
        const double s2 = d2 > 0.0 ? std::sqrt(2.0) * cnt / d2 : 1.0;
        double G[81] = {0};
        for (int i = 0; i < n; ++i) {
            if (!flags[i]) {
                continue;
            }
            const double p1[3] = {(x1[2 * i] - c1x) * s1, (x1[2 * i + 1] - c1y) * s1, 1.0};
            const double p2[3] = {(x2[2 * i] - c2x) * s2, (x2[2 * i + 1] - c2y) * s2, 1.0};
            double a[9];
            for (int c = 0; c < 9; ++c) {
                a[c] = p2[c / 3] * p1[c % 3];
            }
            for (int r = 0; r < 9; ++r) {
                for (int c = 0; c < 9; ++c) {
                    G[r * 9 + c] += a[r] * a[c];
                }
            }
        }
        double U[81], S[9], V[81];
        svd_jacobi(9, 9, G, U, S, V);
        double En[9];
        for (int k = 0; k < 9; ++k) {
            En[k] = V[k * 9 + 8];
        }
        const double T1[9] = {s1, 0, -s1 * c1x, 0, s1, -s1 * c1y, 0, 0, 1};
        const double T2t[9] = {s2, 0, 0, 0, s2, 0, -s2 * c2x, -s2 * c2y, 1};
        double tmp[9], E0[9], U3[9], S3[3], V3[9], Vt3[9], UD[9];
        mul3(En, T1, tmp);
        mul3(T2t, tmp, E0);
        svd3(E0, U3, S3, V3);
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                Vt3[i * 3 + j] = V3[j * 3 + i];
            }
        }
        const double D[9] = {1, 0, 0, 0, 1, 0, 0, 0, 0};
        mul3(U3, D, UD);
        mul3(UD, Vt3, Efit);
    }
    const int fit_count = count(Efit, flags2.data());
    if (last_cost <= best_cost) {  // the refit becomes the current model: the next round refits on ITS inliers
        std::copy(Efit, Efit + 9, Ebest.begin());
        flags = flags2;
        best_count = fit_count;
        best_cost = last_cost;
    }
    }  // rounds
    std::copy(Ebest.begin(), Ebest.end(), E);
    if (n_inliers != nullptr) {
        *n_inliers = best_count;
    }
    if (mask != nullptr) {
        std::copy(flags.begin(), flags.end(), mask);
    }
    return best;
}

// Sampson inlier count of ONE essential matrix in ordinary arithmetic (every product and sum rounded on its own, the
// textbook expression) -- deliberately NOT the fused chains the GPU kernel and oracle_essential_from_samples share, so
// that an algebra error common to both would show.  Differs from the fused count only for correspondences
// within rounding of the threshold.
int oracle_sampson_count_plain(const double* E, const float* pts1, const float* pts2, int n, const double* K, double threshold_px) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double thr = threshold_px / (0.5 * (fx + fy));
    int c = 0;
    for (int i = 0; i < n; ++i) {
        const double a[3] = {(static_cast<double>(pts1[2 * i]) - cx) / fx, (static_cast<double>(pts1[2 * i + 1]) - cy) / fy, 1.0};
        const double b[3] = {(static_cast<double>(pts2[2 * i]) - cx) / fx, (static_cast<double>(pts2[2 * i + 1]) - cy) / fy, 1.0};
        double Ea[3], Etb[3];
        for (int r = 0; r < 3; ++r) {
            Ea[r] = E[r * 3] * a[0] + E[r * 3 + 1] * a[1] + E[r * 3 + 2] * a[2];
            Etb[r] = E[r] * b[0] + E[3 + r] * b[1] + E[6 + r] * b[2];
        }
        const double r = b[0] * Ea[0] + b[1] * Ea[1] + b[2] * Ea[2];
        const double den = Ea[0] * Ea[0] + Ea[1] * Ea[1] + Etb[0] * Etb[0] + Etb[1] * Etb[1];
        if (den > 0.0 && (r * r) / den < thr * thr) {
            c++;
        }
    }
    return c;
}

int oracle_essential_ransac(const float* pts1, const float* pts2, int n, const double* K, int iterations,
                            double threshold_px, unsigned seed, double* E, int* n_inliers, unsigned char* mask,
                            double* E_all, int* counts_all) {
    if (n < 8 || iterations <= 0) {
        return -1;
    }
    // the sampling rule of the reference's own RANSAC (loop_closure.cpp:177-193) with an explicit seed
    std::mt19937 rng(seed);
    std::uniform_int_distribution<int> dist(0, n - 1);
    std::vector<int> table(static_cast<size_t>(iterations) * 8);
    for (int it = 0; it < iterations; ++it) {
        int* idx = &table[static_cast<size_t>(it) * 8];
        int have = 0;
        while (have < 8) {
            const int c = dist(rng);
            if (std::find(idx, idx + have, c) == idx + have) {
                idx[have++] = c;
            }
        }
    }
    return oracle_essential_from_samples(pts1, pts2, n, K, table.data(), iterations, threshold_px, E, n_inliers, mask, E_all,
                                         counts_all);
}

}  // extern "C"
