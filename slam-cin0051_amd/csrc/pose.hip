// pose.hip -- the small dense pose solves, batched on the GPU.
//
// Reference (paths relative to the reference root):
//   a11 LoopClosure::solvePnP            src/backend/loop_closure.cpp:238-274  (DLT, JacobiSVD)
//   a12 RANSAC inlier scoring            src/backend/loop_closure.cpp:201-215
//   a13 decomposeEssential / triangulateSimple / simpleRecoverPose
//                                        src/frontend/simple_pose_recover.cpp:6-98   (cv::SVD)
//   a14 slam::triangulate                include/slam/common/common.hpp:201-221      (cv::SVD)
//
// PARITY UNPINNED by the reference (third-party SVDs, no value assertions): results are compared
// with the oracle's double-precision Jacobi SVD at 1e-5 relative on sign-invariant quantities.
//
// MI355X mapping.  The DLT system A (2n x 12) is the one dense contraction of the path, and it
// runs on the FP64 matrix cores (v_mfma_f64_16x16x4_f64) -- but NOT as a one-shot eigen-solve of
// A^T A, which squares the condition number (pixel-scale rows give cond(A) ~ 1e5, so the null
// vector of A^T A is only good to ~1e-4).  Instead each wave runs a PRECONDITIONED Jacobi SVD:
//     repeat 3x:  G = A^T A          (MFMA, 16x16 tile, K = 2n in steps of 4)
//                 G = V_p^T D V_p    (cyclic Jacobi on the 12x12 Gram matrix, lanes = rows, LDS)
//                 A <- A V_p ; V <- V V_p        (MFMA)
// After the first pass the columns of A are nearly orthogonal, so the next Gram matrix is nearly
// diagonal and its small entries are computed with small RELATIVE error: the iteration converges
// to the singular vectors with SVD-level accuracy while every O(n * 12^2) contraction stays on MFMA.
// The 3x3 and 4x4 problems (orthonormalisation, essential-matrix decomposition, triangulation) are
// one-sided Jacobi SVDs in registers, one per lane.
#include <algorithm>

#include "slamx_internal.h"

#pragma clang fp contract(off)

namespace slamx {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

// 1 / x, 1 / sqrt(x) and sqrt(x) from the hardware estimates plus Newton steps (a few ulps): the IEEE-exact division and
// square-root sequences cost ~25-30 instructions each, and a Jacobi rotation has seven of them.  Used where a result is
// an iterate of a self-correcting iteration (rotation angles), never where the reference's rounding is observable.
__device__ __forceinline__ double rcp_fast(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}
__device__ __forceinline__ double rsq_fast(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}

// ---------------- per-lane one-sided Jacobi (Hestenes) iteration on a tiny M x N matrix ----------------
// Rotates the columns of W (and of Vt, which starts as the identity) until every pair is orthogonal to 1e-15 relative:
// W = U diag(S) Vt^T.  Same cyclic order as the CPU checker's svd_jacobi; a pair that is already orthogonal to that level
// is left alone (the checker rotates it by ~1e-16), and the sweep that finds nothing to rotate ends the iteration.
template <int M, int N>
__device__ __forceinline__ void jacobi_columns(double* W, double* Vt) {
#pragma unroll
    for (int i = 0; i < N * N; ++i) {
        Vt[i] = (i / N == i % N) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < N - 1; ++p) {
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
                for (int k = 0; k < M; ++k) {
                    const double a = W[k * N + p];
                    const double b = W[k * N + q];
                    alpha += a * a;
                    beta += b * b;
                    gamma += a * b;
                }
                if (gamma * gamma > 1e-30 * (alpha * beta)) {
                    rotated = true;
                    const double zeta = (beta - alpha) * rcp_fast(2.0 * gamma);
                    const double az = fabs(zeta);
                    const double q1 = 1.0 + az * az;
                    // |zeta| + sqrt(1 + zeta^2); beyond 1e150 the square overflows and the sum is 2 |zeta| to 1e-300
                    const double den = az > 1e150 ? 2.0 * az : az + q1 * rsq_fast(q1);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) * rcp_fast(den);
                    const double c = rsq_fast(1.0 + t * t);
                    const double s = c * t;
#pragma unroll
                    for (int k = 0; k < M; ++k) {
                        const double a = W[k * N + p];
                        const double b = W[k * N + q];
                        W[k * N + p] = c * a - s * b;
                        W[k * N + q] = s * a + c * b;
                    }
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        const double a = Vt[k * N + p];
                        const double b = Vt[k * N + q];
                        Vt[k * N + p] = c * a - s * b;
                        Vt[k * N + q] = s * a + c * b;
                    }
                }
            }
        }
        if (!rotated) {
            break;
        }
    }
}

// W = U diag(S) V^T, S descending (ties keep column order), U columns of zero singular values are left zero.
template <int M, int N>
__device__ void svd_small(const double* A, double* U, double* S, double* V) {
    double W[M * N];
    double Vt[N * N];
#pragma unroll
    for (int i = 0; i < M * N; ++i) {
        W[i] = A[i];
    }
    jacobi_columns<M, N>(W, Vt);
    double sv[N];
    int order[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double nn = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) {
            nn += W[k * N + j] * W[k * N + j];
        }
        sv[j] = sqrt(nn);
        order[j] = j;
    }
    // stable insertion sort, descending
#pragma unroll
    for (int i = 1; i < N; ++i) {
#pragma unroll
        for (int j = i; j > 0; --j) {
            if (sv[order[j]] > sv[order[j - 1]]) {
                const int tmp = order[j];
                order[j] = order[j - 1];
                order[j - 1] = tmp;
            }
        }
    }
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
        const int j = order[jj];
        S[jj] = sv[j];
#pragma unroll
        for (int k = 0; k < M; ++k) {
            U[k * N + jj] = sv[j] > 0.0 ? W[k * N + j] / sv[j] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            V[k * N + jj] = Vt[k * N + j];
        }
    }
}

// Only the right singular vector of the SMALLEST singular value (the last column of V above): no square roots, no U.
// Among equal smallest values the stable descending sort puts the highest column last.
template <int M, int N>
__device__ void smallest_right_singular_vector(const double* A, double* x) {
    double W[M * N];
    double Vt[N * N];
#pragma unroll
    for (int i = 0; i < M * N; ++i) {
        W[i] = A[i];
    }
    jacobi_columns<M, N>(W, Vt);
    double nn[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) {
            v += W[k * N + j] * W[k * N + j];
        }
        nn[j] = v;
    }
    double best = nn[0];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        x[k] = Vt[k * N];
    }
#pragma unroll
    for (int j = 1; j < N; ++j) {
        if (nn[j] <= best) {
            best = nn[j];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                x[k] = Vt[k * N + j];
            }
        }
    }
}

__device__ __forceinline__ double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
__device__ __forceinline__ void mul3(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s += A[i * 3 + k] * B[k * 3 + j];
            }
            C[i * 3 + j] = s;
        }
    }
}
// 3x3 SVD with U completed to an orthonormal basis when the matrix has rank 2 (essential matrices)
__device__ void svd3(const double* M, double* U, double* S, double* V) {
    svd_small<3, 3>(M, U, S, V);
    if (S[2] <= 1e-14 * fmax(S[0], 1e-300)) {
        const double u0[3] = {U[0], U[3], U[6]};
        const double u1[3] = {U[1], U[4], U[7]};
        U[2] = u0[1] * u1[2] - u0[2] * u1[1];
        U[5] = u0[2] * u1[0] - u0[0] * u1[2];
        U[8] = u0[0] * u1[1] - u0[1] * u1[0];
    }
}

// triangulateSimple / slam::triangulate: null vector of the 4x4 DLT (vt.row(3))
__device__ void triangulate_dlt(const double* P1, const double* P2, float x1, float y1, float x2, float y2, double* X) {
    double A[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        A[0 * 4 + j] = static_cast<double>(x1) * P1[8 + j] - P1[0 + j];
        A[1 * 4 + j] = static_cast<double>(y1) * P1[8 + j] - P1[4 + j];
        A[2 * 4 + j] = static_cast<double>(x2) * P2[8 + j] - P2[0 + j];
        A[3 * 4 + j] = static_cast<double>(y2) * P2[8 + j] - P2[4 + j];
    }
    smallest_right_singular_vector<4, 4>(A, X);
}

// ---- the cheirality vote's two signs without an SVD, where that is safe ----
// simpleRecoverPose consumes only sign(z1) and sign(z2) of the triangulated point, X = smallest right singular vector of
// the 4 x 4 DLT matrix A.  Screen: the smallest eigenpair of G = A^T A from its characteristic polynomial (Newton from 0
// converges monotonically to the smallest root of a polynomial whose roots are all real and non-negative) and the adjugate
// of G - lambda I -- ~650 instructions instead of the ~4000 of the Jacobi iteration -- and accept its signs only when a
// residual bound certifies the direction of X to 5e-5 and X2, X3 and the second depth are farther than 1e-3 |X| from zero;
// every other correspondence goes through the Jacobi iteration.  Returns 1 / 0 for "in front of both" / not, -1 for
// "not decided".
__device__ __forceinline__ int cheirality_screen(const double* A, const double* row2 /* third row of K [R|t] */) {
    double g[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = i; j < 4; ++j) {
            const double v = __builtin_fma(A[i], A[j], __builtin_fma(A[4 + i], A[4 + j], __builtin_fma(A[8 + i], A[8 + j], A[12 + i] * A[12 + j])));
            g[i][j] = v;
            g[j][i] = v;
        }
    }
    // 2 x 2 minors of rows (0, 1) and (2, 3); det, and with the diagonal shifted the adjugate
    auto minors = [&](double d0, double d1, double d2, double d3, double* sm, double* tm) {
        sm[0] = d0 * d1 - g[0][1] * g[1][0];
        sm[1] = d0 * g[1][2] - g[0][2] * g[1][0];
        sm[2] = d0 * g[1][3] - g[0][3] * g[1][0];
        sm[3] = g[0][1] * g[1][2] - g[0][2] * d1;
        sm[4] = g[0][1] * g[1][3] - g[0][3] * d1;
        sm[5] = g[0][2] * g[1][3] - g[0][3] * g[1][2];
        tm[5] = d2 * d3 - g[2][3] * g[3][2];
        tm[4] = g[2][1] * d3 - g[2][3] * g[3][1];
        tm[3] = g[2][1] * g[3][2] - d2 * g[3][1];
        tm[2] = g[2][0] * d3 - g[2][3] * g[3][0];
        tm[1] = g[2][0] * g[3][2] - d2 * g[3][0];
        tm[0] = g[2][0] * g[3][1] - g[2][1] * g[3][0];
    };
    double sm[6], tm[6];
    minors(g[0][0], g[1][1], g[2][2], g[3][3], sm, tm);
    const double c0 = sm[0] * tm[5] - sm[1] * tm[4] + sm[2] * tm[3] + sm[3] * tm[2] - sm[4] * tm[1] + sm[5] * tm[0];
    const double c3 = g[0][0] + g[1][1] + g[2][2] + g[3][3];
    const double c2 = sm[0] + (g[0][0] * g[2][2] - g[0][2] * g[0][2]) + (g[0][0] * g[3][3] - g[0][3] * g[0][3]) +
                      (g[1][1] * g[2][2] - g[1][2] * g[1][2]) + (g[1][1] * g[3][3] - g[1][3] * g[1][3]) + tm[5];
    // principal 3 x 3 minors (without index k) = the diagonal of the adjugate
    const double m3 = g[2][2] * sm[0] - g[1][2] * sm[1] + g[0][2] * sm[3];   // rows / cols 0 1 2
    const double m2 = g[3][3] * sm[0] - g[1][3] * sm[2] + g[0][3] * sm[4];   // 0 1 3
    const double m1 = g[0][0] * tm[5] - g[0][2] * tm[2] + g[0][3] * tm[1];   // 0 2 3
    const double m0 = g[1][1] * tm[5] - g[1][2] * tm[4] + g[1][3] * tm[3];   // 1 2 3
    const double c1 = m0 + m1 + m2 + m3;
    if (!(c1 > 0.0) || !(c3 > 0.0) || !(c0 == c0)) {
        return -1;
    }
    // smallest root of l^4 - c3 l^3 + c2 l^2 - c1 l + c0
    double lam = 0.0;
    bool conv = false;
    const double eps = 1e-15 * c3;
    for (int it = 0; it < 48; ++it) {
        const double pv = (((lam - c3) * lam + c2) * lam - c1) * lam + c0;
        const double dv = ((4.0 * lam - 3.0 * c3) * lam + 2.0 * c2) * lam - c1;
        if (!(dv < 0.0)) {
            break;  // past the first turning point: not the regime this screen is for
        }
        const double step = pv * rcp_fast(dv);
        lam -= step;
        if (fabs(step) <= eps) {
            conv = true;
            break;
        }
    }
    if (!conv || !(lam >= -eps)) {
        return -1;
    }
    minors(g[0][0] - lam, g[1][1] - lam, g[2][2] - lam, g[3][3] - lam, sm, tm);
    const double d0 = g[0][0] - lam, d1 = g[1][1] - lam, d2 = g[2][2] - lam, d3 = g[3][3] - lam;
    // diagonal of adj(G - lambda I): the column of the largest one is the best-conditioned null vector
    const double a00 = d1 * tm[5] - g[1][2] * tm[4] + g[1][3] * tm[3];
    const double a11 = d0 * tm[5] - g[0][2] * tm[2] + g[0][3] * tm[1];
    const double a22 = d3 * sm[0] - g[1][3] * sm[2] + g[0][3] * sm[4];
    const double a33 = d2 * sm[0] - g[1][2] * sm[1] + g[0][2] * sm[3];
    const double amax = fmax(fmax(fabs(a00), fabs(a11)), fmax(fabs(a22), fabs(a33)));
    if (!(amax > 1e-6 * c1)) {
        return -1;  // the two smallest eigenvalues (nearly) coincide: no well-defined vector from the adjugate
    }
    double X[4];
    if (fabs(a00) == amax) {  // column 0 of the adjugate (cofactors of row 0, symmetric)
        X[0] = a00;
        X[1] = -g[1][0] * tm[5] + g[1][2] * tm[2] - g[1][3] * tm[1];
        X[2] = g[1][0] * tm[4] - d1 * tm[2] + g[1][3] * tm[0];
        X[3] = -g[1][0] * tm[3] + d1 * tm[1] - g[1][2] * tm[0];
    } else if (fabs(a11) == amax) {
        X[0] = -g[0][1] * tm[5] + g[0][2] * tm[4] - g[0][3] * tm[3];
        X[1] = a11;
        X[2] = -d0 * tm[4] + g[0][1] * tm[2] - g[0][3] * tm[0];
        X[3] = d0 * tm[3] - g[0][1] * tm[1] + g[0][2] * tm[0];
    } else if (fabs(a22) == amax) {
        X[0] = g[3][1] * sm[5] - g[3][2] * sm[4] + d3 * sm[3];
        X[1] = -g[3][0] * sm[5] + g[3][2] * sm[2] - d3 * sm[1];
        X[2] = a22;
        X[3] = -g[3][0] * sm[3] + g[3][1] * sm[1] - g[3][2] * sm[0];
    } else {
        X[0] = -g[2][1] * sm[5] + d2 * sm[4] - g[2][3] * sm[3];
        X[1] = g[2][0] * sm[5] - d2 * sm[2] + g[2][3] * sm[1];
        X[2] = -g[2][0] * sm[4] + g[2][1] * sm[2] - g[2][3] * sm[0];
        X[3] = a33;
    }
    const double nx = fabs(X[0]) + fabs(X[1]) + fabs(X[2]) + fabs(X[3]);
    // A-posteriori bound on the direction error of X: |G X - lambda X| / gap, with gap >= amax / c2 (amax <= l1 l2 (l3 - l4),
    // c2 >= l1 l2).  It covers the rounding of the coefficients (c0 = det G loses digits when l4 << l1) as well as an
    // unconverged root; beyond 5e-5 the correspondence goes to the Jacobi iteration.
    double resid = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        resid += fabs(g[k][0] * X[0] + g[k][1] * X[1] + g[k][2] * X[2] + g[k][3] * X[3] - lam * X[k]);
    }
    if (!(resid * c2 < 2.5e-5 * amax * nx)) {
        return -1;
    }
    const double z2n = row2[0] * X[0] + row2[1] * X[1] + row2[2] * X[2] + row2[3] * X[3];
    const double rn = fabs(row2[0]) + fabs(row2[1]) + fabs(row2[2]) + fabs(row2[3]);
    const double tol = 1e-3 * nx;
    if (!(fabs(X[2]) > tol) || !(fabs(X[3]) > tol) || !(fabs(z2n) > tol * rn)) {
        return -1;
    }
    const bool front1 = (X[2] > 0.0) == (X[3] > 0.0);
    const bool front2 = (z2n > 0.0) == (X[3] > 0.0);
    return front1 && front2 ? 1 : 0;
}

// ---------------- a11: DLT PnP, one wave per hypothesis ----------------
constexpr int PNP_MAX_ROWS = 64;  // 2n <= 64
constexpr int LD = 16;            // all LDS matrices are padded to 16 columns

// LDS work area of one wave's preconditioned Jacobi SVD: A holds up to ROWS rows
template <int ROWS>
struct JacobiWork {
    double A[ROWS * LD];
    double G[LD * LD];
    double Vp[LD * LD];
    double Vt[LD * LD];  // accumulated V
    double T[(ROWS > LD ? ROWS : LD) * LD];
};
typedef JacobiWork<PNP_MAX_ROWS> PnpWork;  // solvePnP with up to 32 points (slamx_pnp_dlt_batch)

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// D(16x16) = X^T X over `rows` rows of X (rows x 16): FP64 MFMA, K in steps of 4
__device__ void mfma_gram(const double* X, int rows, double* D, int lane) {
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    const int col = lane & 15;
    const int kk = lane >> 4;
    for (int k0 = 0; k0 < rows; k0 += 4) {
        const int r = k0 + kk;
        const double a = r < rows ? X[r * LD + col] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        D[(kk + 4 * r) * LD + col] = acc[r];  // C/D: col = lane & 15, row = (lane >> 4) + 4 * reg
    }
}

// D(rows x 16) = X(rows x 16) * Y(16 x 16): FP64 MFMA, one 16-row tile at a time
__device__ void mfma_mul(const double* X, int rows, const double* Y, double* D, int lane) {
    const int col = lane & 15;
    const int kk = lane >> 4;
    for (int t0 = 0; t0 < rows; t0 += 16) {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4) {
            const int xr = t0 + col;  // A operand: X[i = lane & 15][k = lane >> 4]
            const double a = xr < rows ? X[xr * LD + k0 + kk] : 0.0;
            const double b = Y[(k0 + kk) * LD + col];  // B operand: Y[k = lane >> 4][j = lane & 15]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dr = t0 + kk + 4 * r;
            if (dr < rows) {
                D[dr * LD + col] = acc[r];
            }
        }
    }
}

// cyclic Jacobi on the symmetric n x n matrix G (LDS), accumulating the rotations in Vp (= I on entry)
__device__ void jacobi_sym(double* G, double* Vp, int n, int lane) {
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                const double app = G[p * LD + p];
                const double aqq = G[q * LD + q];
                const double apq = G[p * LD + q];
                // (angles from hardware reciprocals + Newton steps: the iteration corrects itself, see rcp_fast)
                const double pq2 = fabs(app * aqq);
                if (apq == 0.0 || apq * apq <= 1e-34 * pq2) {
                    continue;  // wave-uniform
                }
                off = fmax(off, fabs(apq) * rsq_fast(fmax(pq2, 1e-300)));
                const double zeta = (aqq - app) * rcp_fast(2.0 * apq);
                const double az = fabs(zeta);
                const double q1 = 1.0 + az * az;
                const double den = az > 1e150 ? 2.0 * az : az + q1 * rsq_fast(q1);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) * rcp_fast(den);
                const double c = rsq_fast(1.0 + t * t);
                const double s = c * t;
                double gkp = 0.0, gkq = 0.0, vkp = 0.0, vkq = 0.0;
                if (lane < n) {
                    gkp = G[lane * LD + p];
                    gkq = G[lane * LD + q];
                }
                if (lane < LD) {
                    vkp = Vp[lane * LD + p];
                    vkq = Vp[lane * LD + q];
                }
                wave_sync();
                if (lane < n && lane != p && lane != q) {
                    const double np_ = c * gkp - s * gkq;
                    const double nq_ = s * gkp + c * gkq;
                    G[lane * LD + p] = np_;
                    G[p * LD + lane] = np_;
                    G[lane * LD + q] = nq_;
                    G[q * LD + lane] = nq_;
                }
                if (lane == 0) {
                    G[p * LD + p] = app - t * apq;
                    G[q * LD + q] = aqq + t * apq;
                    G[p * LD + q] = 0.0;
                    G[q * LD + p] = 0.0;
                }
                if (lane < LD) {
                    Vp[lane * LD + p] = c * vkp - s * vkq;
                    Vp[lane * LD + q] = s * vkp + c * vkq;
                }
                wave_sync();
            }
        }
        if (off < 1e-15) {
            break;
        }
    }
}

// Right singular vector of the smallest singular value of w.A (rows x ncols, ncols <= 12, zero padded
// to 16 columns): three passes of {G = A^T A on MFMA; cyclic Jacobi on G; A <- A Vp, Vt <- Vt Vp}.
// Returns the column of w.Vt that holds it (wave-uniform).
template <typename Work>
__device__ int wave_null_column(Work& w, int rows, int ncols, int lane) {
    for (int e = lane; e < LD * LD; e += 64) {
        w.Vt[e] = (e / LD == e % LD) ? 1.0 : 0.0;
    }
    wave_sync();
    for (int pass = 0; pass < 3; ++pass) {
        mfma_gram(w.A, rows, w.G, lane);
        for (int e = lane; e < LD * LD; e += 64) {
            w.Vp[e] = (e / LD == e % LD) ? 1.0 : 0.0;
        }
        wave_sync();
        jacobi_sym(w.G, w.Vp, ncols, lane);
        mfma_mul(w.A, rows, w.Vp, w.T, lane);
        wave_sync();
        for (int e = lane; e < rows * LD; e += 64) {
            w.A[e] = w.T[e];
        }
        mfma_mul(w.Vt, LD, w.Vp, w.T, lane);
        wave_sync();
        for (int e = lane; e < LD * LD; e += 64) {
            w.Vt[e] = w.T[e];
        }
        wave_sync();
    }
    // smallest singular value: column norms of the final A (= diag of its Gram matrix)
    mfma_gram(w.A, rows, w.G, lane);
    wave_sync();
    int jmin = 0;
    double smin = w.G[0];
    for (int j = 1; j < ncols; ++j) {
        const double dj = w.G[j * LD + j];
        if (dj < smin) {
            smin = dj;
            jmin = j;
        }
    }
    return jmin;
}

// One hypothesis, one wave: build A (loop_closure.cpp:249-255) from n_points correspondences, zero padded to 16 columns;
// pt3(i) / pt2(i) return pointers to the i-th 3-D point and pixel.  Lane 0 writes R (9), t (3) and optionally the DLT
// null vector (12).
template <typename Work, typename P3, typename P2>
__device__ void wave_solve_pnp(Work& w, int n_points, P3 pt3, P2 pt2, int lane, double* R_out, double* t_out, double* null_out) {
    const int rows = 2 * n_points;
    for (int e = lane; e < rows * LD; e += 64) {
        const int r = e / LD;
        const int c = e - r * LD;
        const int i = r >> 1;
        const double* P3p = pt3(i);
        const double* P2p = pt2(i);
        const double X = P3p[0], Y = P3p[1], Z = P3p[2];
        const double uv = (r & 1) ? P2p[1] : P2p[0];
        double val = 0.0;
        const int cc = (r & 1) ? c - 4 : c;  // second row of the pair shifts the [X Y Z 1] block by 4
        if (cc >= 0 && cc < 4) {
            val = cc == 0 ? X : cc == 1 ? Y : cc == 2 ? Z : 1.0;
        } else if (c >= 8 && c < 12) {
            const double base = c == 8 ? X : c == 9 ? Y : c == 10 ? Z : 1.0;
            val = -uv * base;
        }
        w.A[e] = val;
    }
    const int jmin = wave_null_column(w, rows, 12, lane);
    if (lane == 0) {
        double p[12];
        for (int k = 0; k < 12; ++k) {
            p[k] = w.Vt[k * LD + jmin];
            if (null_out != nullptr) {
                null_out[k] = p[k];
            }
        }
        // :260 column-major 3x4 map of the row-major DLT vector (reference behaviour, kept)
        double R0[9], t0[3];
        for (int c = 0; c < 3; ++c) {
            for (int r = 0; r < 3; ++r) {
                R0[r * 3 + c] = p[c * 3 + r];
            }
        }
        for (int r = 0; r < 3; ++r) {
            t0[r] = p[9 + r];
        }
        double Ur[9], Sr[3], Vr[9], Vrt[9], UVt[9], UD[9], Rm[9];
        svd3(R0, Ur, Sr, Vr);
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                Vrt[i * 3 + j] = Vr[j * 3 + i];
            }
        }
        mul3(Ur, Vrt, UVt);
        const double det = det3(UVt);
        const double D[9] = {1, 0, 0, 0, 1, 0, 0, 0, det};
        mul3(Ur, D, UD);
        mul3(UD, Vrt, Rm);
        double fro = 0.0;
        for (int k = 0; k < 9; ++k) {
            fro += R0[k] * R0[k];
        }
        fro = sqrt(fro);
        for (int k = 0; k < 9; ++k) {
            R_out[k] = Rm[k];
        }
        for (int k = 0; k < 3; ++k) {
            t_out[k] = t0[k] / fro;
        }
    }
    wave_sync();
}

__global__ __launch_bounds__(256) void pnp_dlt_kernel(const double* __restrict__ p3d, const double* __restrict__ p2d,
                                                      int n_points, int n_hyp, double* __restrict__ R_out,
                                                      double* __restrict__ t_out, double* __restrict__ null_out) {
    extern __shared__ double pose_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    PnpWork& w = *reinterpret_cast<PnpWork*>(pose_lds + static_cast<size_t>(wave) * (sizeof(PnpWork) / sizeof(double)));
    for (int h = blockIdx.x * 4 + wave; h < n_hyp; h += gridDim.x * 4) {  // wave-uniform
        const double* b3 = p3d + static_cast<size_t>(h) * n_points * 3;
        const double* b2 = p2d + static_cast<size_t>(h) * n_points * 2;
        wave_solve_pnp(
            w, n_points, [&](int i) { return b3 + 3 * i; }, [&](int i) { return b2 + 2 * i; }, lane,
            R_out + static_cast<size_t>(h) * 9, t_out + static_cast<size_t>(h) * 3,
            null_out != nullptr ? null_out + static_cast<size_t>(h) * 12 : nullptr);
    }
}

// ---------------- a12: inlier counts, one workgroup per hypothesis ----------------
__global__ __launch_bounds__(256) void inlier_kernel(const double* __restrict__ p3d, const double* __restrict__ p2d, int n,
                                                     const double* __restrict__ K, const double* __restrict__ R,
                                                     const double* __restrict__ t, double threshold,
                                                     int* __restrict__ counts) {
    __shared__ int total;
    const int h = blockIdx.x;
    if (threadIdx.x == 0) {
        total = 0;
    }
    __syncthreads();
    const double* Rh = R + static_cast<size_t>(h) * 9;
    const double* th = t + static_cast<size_t>(h) * 3;
    int mine = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        double tp[3];
        for (int r = 0; r < 3; ++r) {
            tp[r] = Rh[r * 3] * p3d[3 * j] + Rh[r * 3 + 1] * p3d[3 * j + 1] + Rh[r * 3 + 2] * p3d[3 * j + 2] + th[r];
        }
        if (tp[2] <= 0) {
            continue;
        }
        const double q0 = tp[0] / tp[2], q1 = tp[1] / tp[2], q2 = tp[2] / tp[2];
        const double px = K[0] * q0 + K[1] * q1 + K[2] * q2;
        const double py = K[3] * q0 + K[4] * q1 + K[5] * q2;
        const double ex = p2d[2 * j] - px;
        const double ey = p2d[2 * j + 1] - py;
        if (sqrt(ex * ex + ey * ey) < threshold) {
            mine++;
        }
    }
    if (mine != 0) {
        atomicAdd(&total, mine);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        counts[h] = total;
    }
}

// ---------------- a13: decomposeEssential + cheirality vote ----------------
// poses: 4 x {R (9), t (3)}; KP: 4 x 12 (K [R|t]); KP0: 12
__global__ void decompose_kernel(const double* __restrict__ E, const double* __restrict__ K, double* __restrict__ poses,
                                 double* __restrict__ KP, double* __restrict__ KP0) {
    if (threadIdx.x != 0 || blockIdx.x != 0) {
        return;
    }
    double U[9], S[3], V[9], Vt[9], UW[9], R1[9], R2[9];
    svd3(E, U, S, V);
    const double W[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    const double Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            Vt[i * 3 + j] = V[j * 3 + i];
        }
    }
    mul3(U, W, UW);
    mul3(UW, Vt, R1);
    mul3(U, Wt, UW);
    mul3(UW, Vt, R2);
    const double t_[3] = {U[2], U[5], U[8]};
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
    for (int i = 0; i < 4; ++i) {
        const double* Rs = (i & 1) ? R2 : R1;
        const double sg = i < 2 ? 1.0 : -1.0;
        double P[12];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) {
                P[r * 4 + c] = Rs[r * 3 + c];
                poses[i * 12 + r * 3 + c] = Rs[r * 3 + c];
            }
            P[r * 4 + 3] = sg * t_[r];
            poses[i * 12 + 9 + r] = sg * t_[r];
        }
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 4; ++c) {
                KP[i * 12 + r * 4 + c] = K[r * 3] * P[c] + K[r * 3 + 1] * P[4 + c] + K[r * 3 + 2] * P[8 + c];
            }
        }
    }
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 4; ++c) {
            KP0[r * 4 + c] = c < 3 ? K[r * 3 + c] : 0.0;
        }
    }
}

// one lane per (hypothesis, point): simple_pose_recover.cpp:69-82
__global__ __launch_bounds__(256) void cheirality_kernel(const double* __restrict__ KP, const double* __restrict__ KP0,
                                                         const float* __restrict__ pts1, const float* __restrict__ pts2,
                                                         int n, int* __restrict__ front) {
    const int i = blockIdx.y;
    int mine = 0;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        double X[4];
        triangulate_dlt(KP0, KP + i * 12, pts1[2 * j], pts1[2 * j + 1], pts2[2 * j], pts2[2 * j + 1], X);
        const double inv = 1.0 / X[3];
        for (int k = 0; k < 4; ++k) {
            X[k] = X[k] * inv;
        }
        const double z1 = X[2];
        const double* kp = KP + i * 12;
        const double z2 = kp[8] * X[0] + kp[9] * X[1] + kp[10] * X[2] + kp[11] * X[3];
        if (z1 > 0 && z2 > 0) {
            mine++;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        mine += __shfl_xor(mine, d);
    }
    if ((threadIdx.x & 63) == 0 && mine != 0) {
        atomicAdd(&front[i], mine);
    }
}

// ---------------- a14: triangulate ----------------
__global__ __launch_bounds__(256) void triangulate_kernel(const double* __restrict__ P1, const double* __restrict__ P2,
                                                          const float* __restrict__ pts1, const float* __restrict__ pts2,
                                                          int n, float* __restrict__ x4, double* __restrict__ xyz) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double X[4];
        triangulate_dlt(P1, P2, pts1[2 * i], pts1[2 * i + 1], pts2[2 * i], pts2[2 * i + 1], X);
        float xf[4];
        for (int k = 0; k < 4; ++k) {
            xf[k] = static_cast<float>(X[k]);
            if (x4 != nullptr) {
                x4[4 * i + k] = xf[k];
            }
        }
        // cv "x /= w" on CV_32F == multiply by float(1.0 / double(w))
        const float alpha = static_cast<float>(1.0 / static_cast<double>(xf[3]));
        for (int k = 0; k < 3; ++k) {
            xyz[3 * i + k] = static_cast<double>(__fmul_rn(xf[k], alpha));
        }
    }
}

// =====================================================================================================================
// The pose stage of a batch: every kernel below runs over ALL frame pairs of the batch in one launch (blockIdx.y or
// blockIdx.x = frame slot) and reads its sizes from device memory, so the host enqueues a fixed sequence of launches
// with no allocation, copy or synchronisation in between.  The one-pair entry points (slamx_essential_from_samples,
// slamx_pnp_from_samples) run the SAME kernels on a workspace of one slot.
//
// Parallelisation.  A RANSAC hypothesis is a tiny dense problem; 256 pairs x 256 hypotheses are 65k of them.  The
// 8-point solve maps ONE HYPOTHESIS TO ONE LANE, everything in registers: the null vector of the 8 x 9 constraint matrix
// comes from 8 Householder reflections (A Q = [L 0], z = Q e_9: exact null vector of whatever rank A has, backward
// stable, ~1.5k flops, no iteration) -- a wave per hypothesis (the first version: Gram matrix on MFMA + cyclic Jacobi
// in LDS) spends 64 lanes on the scalar rotation arithmetic of one problem, ~250k wave instructions per hypothesis.  The
// scoring kernels hold a correspondence per lane and stream the hypotheses through LDS.  What IS a dense contraction
// stays on the FP64 matrix cores: the 9 x 9 normal matrix of the refit over all inliers of a pair (essential_refit_
// kernel) and the DLT systems of solvePnP (pnp_hyp_kernel, A^T A of loop_closure.cpp:249-258).
// =====================================================================================================================

// Sampson test of one correspondence under one essential matrix (normalised units), as cv-style RANSACs score it:
// r^2 / (|E x1|_xy^2 + |E^T x2|_xy^2) < thr2.  Same operations, same order as the CPU checker of the tests.
// A pointer into the CONSTANT address space: a load through it with a wave-uniform address is a scalar load (s_load),
// and its result a scalar operand.  Only for arrays no lane writes while the kernel runs (the hypotheses of the scoring
// kernels, written by the kernel before).
typedef const double __attribute__((address_space(4))) * uniform_doubles;
__device__ __forceinline__ uniform_doubles as_uniform(const double* p) { return (uniform_doubles)p; }

template <typename Ptr>
__device__ __forceinline__ bool sampson_inlier(Ptr e, double u1, double v1, double u2, double v2, double thr2) {
    // Every product-sum is a chain of fused multiply-adds, innermost term first (the estimator is this build's own, so
    // its rounding is this build's to define: the CPU checker evaluates the same chain with std::fma) -- 18 FP64
    // instructions per test instead of 33.
    const double ex0 = __builtin_fma(e[0], u1, __builtin_fma(e[1], v1, e[2]));
    const double ex1 = __builtin_fma(e[3], u1, __builtin_fma(e[4], v1, e[5]));
    const double ex2 = __builtin_fma(e[6], u1, __builtin_fma(e[7], v1, e[8]));
    const double et0 = __builtin_fma(e[0], u2, __builtin_fma(e[3], v2, e[6]));
    const double et1 = __builtin_fma(e[1], u2, __builtin_fma(e[4], v2, e[7]));
    const double r = __builtin_fma(u2, ex0, __builtin_fma(v2, ex1, ex2));
    const double den = __builtin_fma(ex0, ex0, __builtin_fma(ex1, ex1, __builtin_fma(et0, et0, et1 * et1)));
    return (den > 0.0) & (r * r < thr2 * den);  // (no short circuit: straight-line code)
}

// The same test, and the correspondence's term of the MSAC cost of the model: its squared Sampson distance when it is an
// inlier, the squared threshold otherwise (pose_final_kernel compares the refit with the winning sample on that cost).
template <typename Ptr>
__device__ __forceinline__ bool sampson_inlier_cost(Ptr e, double u1, double v1, double u2, double v2, double thr2, double* cost) {
    const double ex0 = __builtin_fma(e[0], u1, __builtin_fma(e[1], v1, e[2]));
    const double ex1 = __builtin_fma(e[3], u1, __builtin_fma(e[4], v1, e[5]));
    const double ex2 = __builtin_fma(e[6], u1, __builtin_fma(e[7], v1, e[8]));
    const double et0 = __builtin_fma(e[0], u2, __builtin_fma(e[3], v2, e[6]));
    const double et1 = __builtin_fma(e[1], u2, __builtin_fma(e[4], v2, e[7]));
    const double r = __builtin_fma(u2, ex0, __builtin_fma(v2, ex1, ex2));
    const double den = __builtin_fma(ex0, ex0, __builtin_fma(ex1, ex1, __builtin_fma(et0, et0, et1 * et1)));
    const bool in = (den > 0.0) & (r * r < thr2 * den);
    *cost = in ? (r * r) / den : thr2;
    return in;
}

// En (conditioned null vector, row-major 3 x 3) -> essential matrix: undo the conditioning (E = T2^T En T1,
// T = [s 0 -s cx; 0 s -s cy; 0 0 1]) and project onto singular values (1, 1, 0).
__device__ void essential_from_null_vector(const double* En, double s1, double c1x, double c1y, double s2, double c2x,
                                           double c2y, double* Ef) {
    const double T1[9] = {s1, 0, -s1 * c1x, 0, s1, -s1 * c1y, 0, 0, 1};
    const double T2t[9] = {s2, 0, 0, 0, s2, 0, -s2 * c2x, -s2 * c2y, 1};
    double tmp[9], E0[9], U[9], S[3], V[9], Vt3[9], UD[9];
    mul3(En, T1, tmp);
    mul3(T2t, tmp, E0);
    svd3(E0, U, S, V);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            Vt3[i * 3 + j] = V[j * 3 + i];
        }
    }
    const double D[9] = {1, 0, 0, 0, 1, 0, 0, 0, 0};
    mul3(U, D, UD);
    mul3(UD, Vt3, Ef);
}

// ---- correspondences of every pair, from where the keypoints and matches are ----
// pair p = QUERY frame p against its train frame (p - 1, or first_train_frame for p = 0).  "Frame 1" of
// PoseEstimator::estimate / triangulatePoints is the TRAIN frame (the older one), "frame 2" the query frame.
__global__ __launch_bounds__(256) void pose_gather_kernel(PoseSource src, PoseWork w, PoseParams prm, int first_train_frame) {
    const int p = blockIdx.y;
    const int tf = p > 0 ? p - 1 : first_train_frame;
    int n = tf >= 0 ? src.n_matches[p] : 0;
    int status = tf >= 0 ? 0 : SLAMX_POSE_NO_TRAIN_FRAME;
    if (n > prm.cap) {
        n = prm.cap;
        status |= SLAMX_POSE_MATCH_OVERFLOW;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        w.n[p] = n;
        slamx_pose_result r{};  // E = 0; R = I, t = 0 until a pose is found (the reference leaves R, t untouched)
        r.R[0] = r.R[4] = r.R[8] = 1.0;
        r.pnp_R[0] = r.pnp_R[4] = r.pnp_R[8] = 1.0;
        r.n_matches = n;
        r.best_hypothesis = -1;
        r.status = status | (n < 8 && tf >= 0 ? SLAMX_POSE_TOO_FEW_MATCHES : 0);
        w.res[p] = r;
    }
    const slamx_match* mm = src.matches + static_cast<size_t>(p) * src.maxK;
    const slamx_keypoint* kq = src.kps + static_cast<size_t>(p) * src.maxK;
    const slamx_keypoint* kt = src.kps + static_cast<size_t>(tf < 0 ? 0 : tf) * src.maxK;
    const size_t base = static_cast<size_t>(p) * prm.cap;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const slamx_match m = mm[i];
        const float2 a = make_float2(kt[m.trainIdx].x, kt[m.trainIdx].y);
        const float2 b = make_float2(kq[m.queryIdx].x, kq[m.queryIdx].y);
        w.p1[base + i] = a;
        w.p2[base + i] = b;
        // pose_estimator.cpp:60-65: (pt.x - cx) / fx in double
        w.x1[base + i] = make_double2((static_cast<double>(a.x) - prm.cx) / prm.fx, (static_cast<double>(a.y) - prm.cy) / prm.fy);
        w.x2[base + i] = make_double2((static_cast<double>(b.x) - prm.cx) / prm.fx, (static_cast<double>(b.y) - prm.cy) / prm.fy);
    }
}

// one pair whose pixel points (p1, p2) and count (n[0]) came from the host
__global__ __launch_bounds__(256) void pose_load_points_kernel(PoseWork w, PoseParams prm) {
    const int n = w.n[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float2 a = w.p1[i], b = w.p2[i];
        w.x1[i] = make_double2((static_cast<double>(a.x) - prm.cx) / prm.fx, (static_cast<double>(a.y) - prm.cy) / prm.fy);
        w.x2[i] = make_double2((static_cast<double>(b.x) - prm.cx) / prm.fx, (static_cast<double>(b.y) - prm.cy) / prm.fy);
    }
}

// ---- f3: essential matrix from 8 correspondences, ONE LANE per hypothesis ----
// The reference calls cv::findEssentialMat(points1, points2, K, cv::RANSAC) (pose_estimator.cpp:37-47, third-party).
// This build's own estimator: normalised 8-point algorithm on K^-1-normalised points -- Hartley conditioning of the 8
// samples, null vector of the 8 x 9 constraint matrix, projection onto the essential manifold by a 3 x 3 SVD --,
// hypotheses scored by the Sampson distance.
__global__ __launch_bounds__(64) void essential_hyp_kernel(PoseWork w, PoseParams prm) {
    const int p = blockIdx.y;
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = w.n[p];
    if (h >= prm.iterations || n < 8) {
        return;
    }
    const size_t hh = static_cast<size_t>(p) * prm.iterations + h;
    int idx[8];
    if (prm.device_sampling != 0) {
        pose_draw_sample(prm.seed, 0, p, h, n, 8, idx);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            w.idx[hh * 8 + i] = idx[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = w.idx[hh * 8 + i];
            idx[i] = c < 0 ? 0 : (c >= n ? n - 1 : c);  // a caller's table: keep the loads inside the pair
        }
    }
    const double2* X1 = w.x1 + static_cast<size_t>(p) * prm.cap;
    const double2* X2 = w.x2 + static_cast<size_t>(p) * prm.cap;
    double a[16], b[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double2 u = X1[idx[i]], v = X2[idx[i]];
        a[2 * i] = u.x;
        a[2 * i + 1] = u.y;
        b[2 * i] = v.x;
        b[2 * i + 1] = v.y;
    }
    // Hartley conditioning: centroid to the origin, mean distance sqrt(2)
    double c1x = 0, c1y = 0, c2x = 0, c2y = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c1x += a[2 * i];
        c1y += a[2 * i + 1];
        c2x += b[2 * i];
        c2y += b[2 * i + 1];
    }
    c1x /= 8.0;
    c1y /= 8.0;
    c2x /= 8.0;
    c2y /= 8.0;
    double d1 = 0, d2 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double ux = a[2 * i] - c1x, uy = a[2 * i + 1] - c1y;
        const double vx = b[2 * i] - c2x, vy = b[2 * i + 1] - c2y;
        d1 += sqrt(ux * ux + uy * uy);
        d2 += sqrt(vx * vx + vy * vy);
    }
    const double s1 = d1 > 0.0 ? sqrt(2.0) * 8.0 / d1 : 1.0;
    const double s2 = d2 > 0.0 ? sqrt(2.0) * 8.0 / d2 : 1.0;
    // constraint rows x2^T E x1 = 0, E row-major: [u2 u1, u2 v1, u2, v2 u1, v2 v1, v2, u1, v1, 1]
    double A[8][9];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const double u1 = (a[2 * r] - c1x) * s1, v1 = (a[2 * r + 1] - c1y) * s1;
        const double u2 = (b[2 * r] - c2x) * s2, v2 = (b[2 * r + 1] - c2y) * s2;
        A[r][0] = u2 * u1;
        A[r][1] = u2 * v1;
        A[r][2] = u2;
        A[r][3] = v2 * u1;
        A[r][4] = v2 * v1;
        A[r][5] = v2;
        A[r][6] = u1;
        A[r][7] = v1;
        A[r][8] = 1.0;
    }
    // Null vector by Householder reflections from the right: A H_0 ... H_7 = [L 0], so z = H_0 ... H_7 e_8 satisfies
    // A z = 0 whatever the rank of A.  H_k = I - beta_k v_k v_k^T acts on columns k..8; v_k replaces row k's tail.
    double beta[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double s = 0.0;
#pragma unroll
        for (int c = k; c < 9; ++c) {
            s += A[k][c] * A[k][c];
        }
        const double nrm = sqrt(s);
        const double alpha = A[k][k] >= 0.0 ? -nrm : nrm;
        A[k][k] -= alpha;  // v = x - alpha e_1 (no cancellation: alpha has the opposite sign of x_0)
        double vtv = 0.0;
#pragma unroll
        for (int c = k; c < 9; ++c) {
            vtv += A[k][c] * A[k][c];
        }
        const double bk = vtv > 0.0 ? 2.0 / vtv : 0.0;
        beta[k] = bk;
#pragma unroll
        for (int j = k + 1; j < 8; ++j) {
            double d = 0.0;
#pragma unroll
            for (int c = k; c < 9; ++c) {
                d += A[j][c] * A[k][c];
            }
            d *= bk;
#pragma unroll
            for (int c = k; c < 9; ++c) {
                A[j][c] -= d * A[k][c];
            }
        }
    }
    double z[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1.0};
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        double d = 0.0;
#pragma unroll
        for (int c = k; c < 9; ++c) {
            d += z[c] * A[k][c];
        }
        d *= beta[k];
#pragma unroll
        for (int c = k; c < 9; ++c) {
            z[c] -= d * A[k][c];
        }
    }
    double Ef[9];
    essential_from_null_vector(z, s1, c1x, c1y, s2, c2x, c2y, Ef);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        w.Eh[hh * 9 + k] = Ef[k];
    }
    w.counts[hh] = 0;  // sampson_score_kernel adds into it
}

// ---- every hypothesis against every correspondence of its pair ----
// One wave per workgroup; a lane holds SCORE_PPT correspondences in registers.  A hypothesis is wave-uniform: its nine
// doubles arrive through the scalar cache (s_load) and enter the FP64 instructions as scalar operands -- no LDS staging,
// no barriers.  counts[p][h] += number of correspondences of this wave that pass (integer sums: order-free); lane l keeps
// the wave's count for hypothesis h0 + l and one vector atomic per 64 hypotheses writes them out.
constexpr int SCORE_PPT = 4;
__global__ __launch_bounds__(64) void sampson_score_kernel(PoseWork w, PoseParams prm) {
    const int p = blockIdx.y;
    const int n = w.n[p];
    const int first = blockIdx.x * (64 * SCORE_PPT);
    if (n < 8 || first >= n) {
        return;
    }
    const int lane = threadIdx.x;
    const double2* X1 = w.x1 + static_cast<size_t>(p) * prm.cap;
    const double2* X2 = w.x2 + static_cast<size_t>(p) * prm.cap;
    double u1[SCORE_PPT], v1[SCORE_PPT], u2[SCORE_PPT], v2[SCORE_PPT];
    bool live[SCORE_PPT];
#pragma unroll
    for (int q = 0; q < SCORE_PPT; ++q) {
        const int i = first + q * 64 + lane;
        live[q] = i < n;
        const double2 a = live[q] ? X1[i] : make_double2(0.0, 0.0);
        const double2 b = live[q] ? X2[i] : make_double2(0.0, 0.0);
        u1[q] = a.x;
        v1[q] = a.y;
        u2[q] = b.x;
        v2[q] = b.y;
    }
    const double* __restrict__ Eh = w.Eh + static_cast<size_t>(p) * prm.iterations * 9;
    int* counts = w.counts + static_cast<size_t>(p) * prm.iterations;
    const double thr2 = prm.thr2;
    // Two register sets take turns: while one hypothesis is evaluated the next one's nine doubles are on their way
    // through the scalar cache (a scalar load takes ~200 cycles to return; a single set would need 18 s_mov per step)
    const int H = prm.iterations;
    auto fetch = [&](int h, double* e9) {
        const uniform_doubles e = as_uniform(Eh + static_cast<size_t>(min(h, H - 1)) * 9);  // wave-uniform address
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            e9[j] = e[j];
        }
    };
    auto count = [&](const double* e9) {
        int c = 0;
#pragma unroll
        for (int q = 0; q < SCORE_PPT; ++q) {
            const bool in = live[q] & sampson_inlier(e9, u1[q], v1[q], u2[q], v2[q], thr2);
            c += __builtin_popcountll(__ballot(in));
        }
        return c;
    };
    double ea[9], eb[9];
    fetch(0, ea);
    for (int h0 = 0; h0 < H; h0 += 64) {
        const int nh = min(64, H - h0);
        int mine = 0;
        for (int k = 0; k < nh; k += 2) {
            fetch(h0 + k + 1, eb);
            const int ca = count(ea);
            mine += lane == k ? ca : 0;
            fetch(h0 + k + 2, ea);
            const int cb = count(eb);
            mine += ((lane == k + 1) & (k + 1 < nh)) ? cb : 0;
        }
        if (lane < nh && mine != 0) {
            atomicAdd(&counts[h0 + lane], mine);
        }
    }
}

// ---- first maximum, its matrix and its inlier flags; one workgroup per pair ----
__global__ __launch_bounds__(256) void pose_select_kernel(PoseWork w, PoseParams prm) {
    __shared__ int best_c[256], best_i[256];
    __shared__ double eb[9];
    const int p = blockIdx.x;
    const int n = w.n[p];
    if (n < 8) {
        return;  // uniform
    }
    const int* counts = w.counts + static_cast<size_t>(p) * prm.iterations;
    int bc = -1, bi = 0x7fffffff;
    for (int h = threadIdx.x; h < prm.iterations; h += 256) {  // ascending: strict '>' keeps the first maximum
        const int c = counts[h];
        if (c > bc) {
            bc = c;
            bi = h;
        }
    }
    best_c[threadIdx.x] = bc;
    best_i[threadIdx.x] = bi;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (static_cast<int>(threadIdx.x) < d) {
            const int oc = best_c[threadIdx.x + d], oi = best_i[threadIdx.x + d];
            if (oc > best_c[threadIdx.x] || (oc == best_c[threadIdx.x] && oi < best_i[threadIdx.x])) {
                best_c[threadIdx.x] = oc;
                best_i[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    const int best = best_i[0];
    if (threadIdx.x < 9) {
        const double v = w.Eh[(static_cast<size_t>(p) * prm.iterations + best) * 9 + threadIdx.x];
        eb[threadIdx.x] = v;
        w.Ebest[static_cast<size_t>(p) * 9 + threadIdx.x] = v;
        w.Efit[static_cast<size_t>(p) * 9 + threadIdx.x] = v;  // the refit replaces it when it has 8 inliers to work with
    }
    if (threadIdx.x == 0) {
        w.sel[p * 4 + 0] = best;
        w.sel[p * 4 + 1] = best_c[0];
    }
    __syncthreads();
    const double2* X1 = w.x1 + static_cast<size_t>(p) * prm.cap;
    const double2* X2 = w.x2 + static_cast<size_t>(p) * prm.cap;
    uint8_t* fa = w.flags_a + static_cast<size_t>(p) * prm.cap;
    double e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        e[k] = eb[k];
    }
    for (int i = threadIdx.x; i < n; i += 256) {
        const double2 a = X1[i], b = X2[i];
        fa[i] = sampson_inlier(e, a.x, a.y, b.x, b.y, prm.thr2) ? 1 : 0;
    }
}

// ---- least-squares refit of the winning hypothesis on its inliers; one wave per pair ----
// The same conditioned 8-point equations over all inliers: the 9 x 9 normal matrix is accumulated on the FP64 matrix
// cores in chunks of 64 rows (A^T A: the dense contraction of this step), smallest eigenvector by cyclic Jacobi,
// projection onto the essential manifold.
// deterministic sum of a double over the workgroup's 256 threads: xor butterfly inside each wave (every lane ends with the
// same value), the four wave sums added in wave order by everyone
__device__ __forceinline__ double block_sum4(double v, double* wave_part /* [4] in LDS */, int wave) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        v += __shfl_xor(v, d);
    }
    __syncthreads();  // wave_part may still be read from the previous call
    if ((threadIdx.x & 63) == 0) {
        wave_part[wave] = v;
    }
    __syncthreads();
    return ((wave_part[0] + wave_part[1]) + wave_part[2]) + wave_part[3];
}

__global__ __launch_bounds__(256) void essential_refit_kernel(PoseWork w, PoseParams prm) {
    extern __shared__ double pose_lds[];  // [4][64 x 16] chunk rows of the four waves, then [4][16 x 16] their Gram tiles
    __shared__ double wave_part[4];
    __shared__ double G[LD * LD], Vp[LD * LD];
    const int p = blockIdx.x;
    const int n = w.n[p];
    if (n < 8) {
        return;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* x1 = reinterpret_cast<const double*>(w.x1 + static_cast<size_t>(p) * prm.cap);
    const double* x2 = reinterpret_cast<const double*>(w.x2 + static_cast<size_t>(p) * prm.cap);
    const uint8_t* flags = w.flags_a + static_cast<size_t>(p) * prm.cap;
    double* E_out = w.Efit + static_cast<size_t>(p) * 9;
    // conditioning over the inliers: centroids, then mean distances
    double acc[5] = {0, 0, 0, 0, 0};  // sum u1, v1, u2, v2, count
    for (int i = tid; i < n; i += 256) {
        if (flags[i] != 0) {
            acc[0] += x1[2 * i];
            acc[1] += x1[2 * i + 1];
            acc[2] += x2[2 * i];
            acc[3] += x2[2 * i + 1];
            acc[4] += 1.0;
        }
    }
    double tot[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        tot[k] = block_sum4(acc[k], wave_part, wave);
    }
    const double cnt = tot[4];
    if (cnt < 8.0) {
        return;  // uniform; Efit keeps the minimal-sample model
    }
    const double c1x = tot[0] / cnt, c1y = tot[1] / cnt, c2x = tot[2] / cnt, c2y = tot[3] / cnt;
    double dd[2] = {0, 0};
    for (int i = tid; i < n; i += 256) {
        if (flags[i] != 0) {
            const double ux = x1[2 * i] - c1x, uy = x1[2 * i + 1] - c1y;
            const double vx = x2[2 * i] - c2x, vy = x2[2 * i + 1] - c2y;
            dd[0] += sqrt(ux * ux + uy * uy);
            dd[1] += sqrt(vx * vx + vy * vy);
        }
    }
    const double d1 = block_sum4(dd[0], wave_part, wave);
    const double d2 = block_sum4(dd[1], wave_part, wave);
    const double s1 = d1 > 0.0 ? sqrt(2.0) * cnt / d1 : 1.0;
    const double s2 = d2 > 0.0 ? sqrt(2.0) * cnt / d2 : 1.0;
    // normal matrix: wave k takes the chunks k, k + 4, ... of 64 correspondences (outliers contribute zero rows), lane =
    // row, one accumulator tile per wave over all its chunks; the four tiles are added in wave order afterwards
    double* rows = pose_lds + static_cast<size_t>(wave) * 64 * LD;
    d4 gacc = {0.0, 0.0, 0.0, 0.0};
    const int col = lane & 15;
    const int kk = lane >> 4;
    for (int base = wave * 64; base < n; base += 256) {
        const int i = base + lane;
        const bool in = i < n && flags[i] != 0;
        const double p1[3] = {in ? (x1[2 * i] - c1x) * s1 : 0.0, in ? (x1[2 * i + 1] - c1y) * s1 : 0.0, in ? 1.0 : 0.0};
        const double p2[3] = {in ? (x2[2 * i] - c2x) * s2 : 0.0, in ? (x2[2 * i + 1] - c2y) * s2 : 0.0, in ? 1.0 : 0.0};
        wave_sync();  // the previous chunk's rows have been read
#pragma unroll
        for (int c = 0; c < LD; ++c) {
            rows[lane * LD + c] = c < 9 ? p2[c / 3] * p1[c % 3] : 0.0;
        }
        wave_sync();
#pragma unroll
        for (int k0 = 0; k0 < 64; k0 += 4) {
            const double a = rows[(k0 + kk) * LD + col];
            gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, gacc, 0, 0, 0);
        }
    }
    double* tiles = pose_lds + 4 * 64 * LD;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        tiles[wave * LD * LD + (kk + 4 * r) * LD + col] = gacc[r];  // C/D: col = lane & 15, row = (lane >> 4) + 4 * reg
    }
    __syncthreads();
    if (wave != 0) {
        return;
    }
    for (int e = lane; e < LD * LD; e += 64) {
        G[e] = ((tiles[e] + tiles[LD * LD + e]) + tiles[2 * LD * LD + e]) + tiles[3 * LD * LD + e];
        Vp[e] = (e / LD == e % LD) ? 1.0 : 0.0;
    }
    wave_sync();
    jacobi_sym(G, Vp, 9, lane);
    wave_sync();
    if (lane == 0) {
        int jmin = 0;
        for (int j = 1; j < 9; ++j) {
            if (G[j * LD + j] < G[jmin * LD + jmin]) {
                jmin = j;
            }
        }
        double En[9], Ef[9];
        for (int k = 0; k < 9; ++k) {
            En[k] = Vp[k * LD + jmin];
        }
        essential_from_null_vector(En, s1, c1x, c1y, s2, c2x, c2y, Ef);
        for (int k = 0; k < 9; ++k) {
            E_out[k] = Ef[k];
        }
    }
}

// decomposeEssential (simple_pose_recover.cpp:6-18) and the four projection matrices K [R|t] of :61-65
__device__ void decompose_essential(const double* E, const double* K, double* poses, double* KP) {
    double U[9], S[3], V[9], Vt[9], UW[9], R1[9], R2[9];
    svd3(E, U, S, V);
    const double W[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    const double Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            Vt[i * 3 + j] = V[j * 3 + i];
        }
    }
    mul3(U, W, UW);
    mul3(UW, Vt, R1);
    mul3(U, Wt, UW);
    mul3(UW, Vt, R2);
    const double t_[3] = {U[2], U[5], U[8]};
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
    for (int i = 0; i < 4; ++i) {
        const double* Rs = (i & 1) ? R2 : R1;
        const double sg = i < 2 ? 1.0 : -1.0;
        double P[12];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) {
                P[r * 4 + c] = Rs[r * 3 + c];
                poses[i * 12 + r * 3 + c] = Rs[r * 3 + c];
            }
            P[r * 4 + 3] = sg * t_[r];
            poses[i * 12 + 9 + r] = sg * t_[r];
        }
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 4; ++c) {
                KP[i * 12 + r * 4 + c] = K[r * 3] * P[c] + K[r * 3 + 1] * P[4 + c] + K[r * 3 + 2] * P[8 + c];
            }
        }
    }
}

// ---- is the refit the better model?  Then the result's matrix, flags and the four candidate poses; one workgroup per
//      pair.  "Better" is the MSAC cost (sum over the correspondences of min(d^2, thr^2), d = Sampson distance): until round
//      5 the refit was kept only if it had at least as many INLIERS as the winning sample -- but the winner of a count
//      contest among near-equal models is the one that also collects a few wrong matches lying near its epipolar lines,
//      so an accurate refit lost to it by a handful of counts and the stage returned a minimal-sample model (rotation
//      error 0.04 deg, translation direction 1.6 deg on bench.py's scene; with the cost rule 0.001 / 0.02 deg).
//      Sums in a fixed order (thread-strided partial sums, shuffles, four waves in wave order): no floating-point atomics.
__global__ __launch_bounds__(256) void pose_final_kernel(PoseWork w, PoseParams prm) {
    __shared__ int total;
    __shared__ double wave_part[4];
    const int p = blockIdx.x;
    const int n = w.n[p];
    if (n < 8) {
        return;
    }
    if (threadIdx.x == 0) {
        total = 0;
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double2* X1 = w.x1 + static_cast<size_t>(p) * prm.cap;
    const double2* X2 = w.x2 + static_cast<size_t>(p) * prm.cap;
    const uint8_t* fa = w.flags_a + static_cast<size_t>(p) * prm.cap;
    uint8_t* fb = w.flags_b + static_cast<size_t>(p) * prm.cap;
    double e[9], eb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        e[k] = w.Efit[static_cast<size_t>(p) * 9 + k];
        eb[k] = w.Ebest[static_cast<size_t>(p) * 9 + k];
    }
    int mine = 0;
    double cost_fit = 0.0, cost_best = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double2 a = X1[i], b = X2[i];
        double cf, cb;
        const bool in = sampson_inlier_cost(e, a.x, a.y, b.x, b.y, prm.thr2, &cf);
        (void)sampson_inlier_cost(eb, a.x, a.y, b.x, b.y, prm.thr2, &cb);
        fb[i] = in ? 1 : 0;
        mine += in ? 1 : 0;
        cost_fit += cf;
        cost_best += cb;
    }
    if (mine != 0) {
        atomicAdd(&total, mine);
    }
    const double sum_fit = block_sum4(cost_fit, wave_part, wave);
    const double sum_best = block_sum4(cost_best, wave_part, wave);
    __syncthreads();
    const bool use_fit = sum_fit <= sum_best;  // the refit is kept iff its MSAC cost is not higher
    // whichever model is kept becomes the pair's current one -- matrix, flags and count -- so that the next (refit, final)
    // round (POSE_REFIT_ROUNDS of them) refits on ITS inliers: a local-optimisation loop, each round either improves the
    // cost or leaves everything as it was
    for (int i = threadIdx.x; i < n; i += 256) {
        if (use_fit) {
            const_cast<uint8_t*>(fa)[i] = fb[i];
        } else {
            fb[i] = fa[i];
        }
    }
    if (threadIdx.x == 0) {
        const int count_best = w.sel[p * 4 + 1];
        w.sel[p * 4 + 2] = total;
        w.sel[p * 4 + 3] = use_fit ? 1 : 0;
        slamx_pose_result& r = w.res[p];
        double E[9];
        for (int k = 0; k < 9; ++k) {
            E[k] = use_fit ? e[k] : eb[k];
            r.E[k] = E[k];
        }
        if (use_fit) {
            for (int k = 0; k < 9; ++k) {
                w.Ebest[static_cast<size_t>(p) * 9 + k] = E[k];
            }
            w.sel[p * 4 + 1] = total;
        }
        r.n_inliers = use_fit ? total : count_best;
        r.best_hypothesis = w.sel[p * 4 + 0];
        // SLAMX_GEOMETRY_CALIBRATED: the vote's projection matrices are [R|t] themselves (K = identity), see slamx.h
        const double Kid[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        decompose_essential(E, prm.geometry_mode == SLAMX_GEOMETRY_CALIBRATED ? Kid : prm.K, w.poses4 + static_cast<size_t>(p) * 48,
                            w.KP + static_cast<size_t>(p) * 48);
        for (int i = 0; i < 4; ++i) {
            w.front[p * 4 + i] = 0;
        }
    }
}

// ---- simpleRecoverPose's vote (simple_pose_recover.cpp:67-82): one lane per (candidate, correspondence) ----
// The reference triangulates the NORMALISED points (float, pose_estimator.cpp:59-66) against K [I|0] and K [R|t].
__global__ __launch_bounds__(256) void cheirality_pairs_kernel(PoseWork w, PoseParams prm) {
    const int p = blockIdx.z;
    const int i = blockIdx.y;
    const int n = w.n[p];
    if (n < 8 || static_cast<int>(blockIdx.x * blockDim.x) >= n) {
        return;
    }
    double KP0[12];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            KP0[r * 4 + c] = c < 3 ? (prm.geometry_mode == SLAMX_GEOMETRY_CALIBRATED ? (r == c ? 1.0 : 0.0) : prm.K[r * 3 + c]) : 0.0;
        }
    }
    double kp[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        kp[k] = w.KP[static_cast<size_t>(p) * 48 + i * 12 + k];
    }
    const double2* X1 = w.x1 + static_cast<size_t>(p) * prm.cap;
    const double2* X2 = w.x2 + static_cast<size_t>(p) * prm.cap;
    int mine = 0;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const double2 a = X1[j], b = X2[j];
        const float fx1 = static_cast<float>(a.x), fy1 = static_cast<float>(a.y), fx2 = static_cast<float>(b.x),
                    fy2 = static_cast<float>(b.y);
        double A[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            A[0 * 4 + c] = static_cast<double>(fx1) * KP0[8 + c] - KP0[0 + c];
            A[1 * 4 + c] = static_cast<double>(fy1) * KP0[8 + c] - KP0[4 + c];
            A[2 * 4 + c] = static_cast<double>(fx2) * kp[8 + c] - kp[0 + c];
            A[3 * 4 + c] = static_cast<double>(fy2) * kp[8 + c] - kp[4 + c];
        }
        int verdict = cheirality_screen(A, kp + 8);
        if (verdict < 0) {  // (a few correspondences in a thousand: the reference's own sequence)
            double X[4];
            smallest_right_singular_vector<4, 4>(A, X);
            const double inv = 1.0 / X[3];
            for (int k = 0; k < 4; ++k) {
                X[k] = X[k] * inv;
            }
            const double z1 = X[2];
            const double z2 = kp[8] * X[0] + kp[9] * X[1] + kp[10] * X[2] + kp[11] * X[3];
            verdict = (z1 > 0 && z2 > 0) ? 1 : 0;
        }
        mine += verdict;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        mine += __shfl_xor(mine, d);
    }
    if ((threadIdx.x & 63) == 0 && mine != 0) {
        atomicAdd(&w.front[p * 4 + i], mine);
    }
}

// ---- the pose the vote selects (first maximum from maxFront = -1) and triangulatePoints with it ----
// pose_estimator.cpp:69-104: K [I|0] and K [R|t] (sums in k order from 0.0), PIXEL points, float32 homogeneous store.
__global__ __launch_bounds__(256) void triangulate_pairs_kernel(PoseWork w, PoseParams prm) {
    const int p = blockIdx.y;
    const int n = w.n[p];
    if (n < 8) {
        return;
    }
    int best = 0, maxFront = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = w.front[p * 4 + i];
        if (f > maxFront) {
            maxFront = f;
            best = i;
        }
    }
    const double* pose = w.poses4 + static_cast<size_t>(p) * 48 + best * 12;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        slamx_pose_result& r = w.res[p];
        for (int k = 0; k < 9; ++k) {
            r.R[k] = pose[k];
        }
        for (int k = 0; k < 3; ++k) {
            r.t[k] = pose[9 + k];
        }
        for (int i = 0; i < 4; ++i) {
            r.front[i] = w.front[p * 4 + i];
        }
    }
    if (static_cast<int>(blockIdx.x * blockDim.x) >= n) {
        return;
    }
    double T1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double T2[12];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            T2[r * 4 + c] = pose[r * 3 + c];
        }
        T2[r * 4 + 3] = pose[9 + r];
    }
    double P1[12], P2[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s1 += prm.K[i * 3 + k] * T1[k * 4 + j];
                s2 += prm.K[i * 3 + k] * T2[k * 4 + j];
            }
            P1[i * 4 + j] = s1;
            P2[i * 4 + j] = s2;
        }
    }
    const float2* A = w.p1 + static_cast<size_t>(p) * prm.cap;
    const float2* B = w.p2 + static_cast<size_t>(p) * prm.cap;
    double* xyz = w.xyz + static_cast<size_t>(p) * prm.cap * 3;
    float* x4 = w.x4 != nullptr ? w.x4 + static_cast<size_t>(p) * prm.cap * 4 : nullptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float2 a = A[i], b = B[i];
        double X[4];
        triangulate_dlt(P1, P2, a.x, a.y, b.x, b.y, X);
        float xf[4];
        for (int k = 0; k < 4; ++k) {
            xf[k] = static_cast<float>(X[k]);
            if (x4 != nullptr) {
                x4[4 * i + k] = xf[k];
            }
        }
        // cv "x /= w" on CV_32F == multiply by float(1.0 / double(w))
        const float alpha = static_cast<float>(1.0 / static_cast<double>(xf[3]));
        for (int k = 0; k < 3; ++k) {
            xyz[3 * static_cast<size_t>(i) + k] = static_cast<double>(__fmul_rn(xf[k], alpha));
        }
    }
}

// ---- a12 for the batch: the 3D-2D correspondences of LoopClosure::verifyGeometricConsistency (loop_closure.cpp:166-171)
// points2d = the QUERY frame's keypoint of every match, points3d = the map point of the match's TRAIN keypoint.  The
// train frame's map points are those its own pair triangulated, one per match of the train frame, i.e. per QUERY
// keypoint of that pair; match lists are in ascending queryIdx order, so the map point of keypoint k is found by a
// binary search.  Matches whose train keypoint has no map point drop out; the order of the others is kept.
__global__ __launch_bounds__(1024) void pnp_gather_kernel(PoseSource src, PoseWork w, PoseParams prm, int first_train_frame) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int p = blockIdx.x;
    const int tf = p > 0 ? p - 1 : first_train_frame;
    const int n = w.n[p];
    // map points of the train frame (none when its pair had fewer than 8 matches).  A train slot outside this call's
    // frames (frame 0 against first_train_frame >= nframes) was not triangulated by this call: whatever an earlier call
    // left there belongs to other match lists, so frame 0 then has no PnP step (SLAMX_POSE_PNP_SKIPPED)
    const int nt = tf >= 0 && tf < static_cast<int>(gridDim.x) ? w.n[tf] : 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int carry = 0;
    if (tf >= 0 && nt >= 8) {
        const slamx_match* mq = src.matches + static_cast<size_t>(p) * src.maxK;
        const slamx_match* mt = src.matches + static_cast<size_t>(tf) * src.maxK;
        const slamx_keypoint* kq = src.kps + static_cast<size_t>(p) * src.maxK;
        const double* xyz = w.xyz + static_cast<size_t>(tf) * prm.cap * 3;
        double* c3 = w.c3 + static_cast<size_t>(p) * prm.cap * 3;
        double* c2 = w.c2 + static_cast<size_t>(p) * prm.cap * 2;
        for (int base = 0; base < n; base += 1024) {
            const int j = base + static_cast<int>(threadIdx.x);
            int found = -1;
            slamx_match m{};
            if (j < n) {
                m = mq[j];
                int lo = 0, hi = nt - 1;
                while (lo <= hi) {
                    const int mid = (lo + hi) >> 1;
                    const int q = mt[mid].queryIdx;
                    if (q == m.trainIdx) {
                        found = mid;
                        break;
                    }
                    if (q < m.trainIdx) {
                        lo = mid + 1;
                    } else {
                        hi = mid - 1;
                    }
                }
            }
            const unsigned long long bal = __ballot(found >= 0);
            const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) {
                wave_tot[wv] = __builtin_popcountll(bal);
            }
            __syncthreads();
            int woff = 0, tot = 0;
            for (int k = 0; k < 16; ++k) {
                const int c = wave_tot[k];
                woff += k < wv ? c : 0;
                tot += c;
            }
            if (found >= 0) {
                const size_t pos = static_cast<size_t>(carry + woff + before);
                c3[pos * 3 + 0] = xyz[static_cast<size_t>(found) * 3 + 0];
                c3[pos * 3 + 1] = xyz[static_cast<size_t>(found) * 3 + 1];
                c3[pos * 3 + 2] = xyz[static_cast<size_t>(found) * 3 + 2];
                c2[pos * 2 + 0] = static_cast<double>(kq[m.queryIdx].x);
                c2[pos * 2 + 1] = static_cast<double>(kq[m.queryIdx].y);
            }
            carry += tot;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        w.m[p] = carry;
        w.res[p].pnp_correspondences = carry;
        const int need = prm.pnp_min > 6 ? prm.pnp_min : 6;
        if (carry < need) {
            w.res[p].status |= SLAMX_POSE_PNP_SKIPPED;
        }
        carry_s = carry;
    }
    (void)carry_s;
}

// ---- a11 for the batch: solvePnP (loop_closure.cpp:238-274) on a 6-point sample, one LANE per hypothesis ----
// The 12 x 12 DLT matrix of a lane lives in LDS (element e of lane l at [e * 64 + l]: conflict-free), 72 KB per wave.
// p = right singular vector of its smallest singular value: Householder QR (A = Q R; the right singular vectors of R are
// A's), then inverse iteration with R^T R -- two triangular solves per step, no squaring of the condition number (the
// pixel-scale rows give cond(A) ~ 1e5, which is why a one-shot eigen-solve of A^T A would not do) -- until the direction
// moves by less than 1e-14 or 40 steps (a sample whose two smallest singular values coincide has no defined answer).
// A wave per hypothesis (the first version: Gram matrix on MFMA + cyclic Jacobi in LDS, kept for slamx_pnp_dlt_batch)
// took 4.9 ms for 256 x 100 hypotheses; this takes two orders of magnitude less.
constexpr int DLT_N = 12;
constexpr size_t PNP_HYP_LDS = static_cast<size_t>(DLT_N) * DLT_N * 64 * sizeof(double);
__global__ __launch_bounds__(64) void pnp_hyp_kernel(PoseWork w, PoseParams prm) {
    extern __shared__ double pose_lds[];
    const int p = blockIdx.y;
    const int lane = threadIdx.x;
    const int m = w.m[p];
    const int need = prm.pnp_min > 6 ? prm.pnp_min : 6;
    if (m < need) {
        return;  // uniform
    }
    const int h_raw = blockIdx.x * 64 + lane;
    const bool active = h_raw < prm.pnp_iterations;
    const int h = active ? h_raw : prm.pnp_iterations - 1;  // spare lanes redo the last hypothesis and write nothing
    const size_t hh = static_cast<size_t>(p) * prm.pnp_iterations + h;
    int idx[6];
    if (prm.device_sampling != 0) {
        pose_draw_sample(prm.seed, 1, p, h, m, 6, idx);
        if (active) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                w.pidx[hh * 6 + k] = idx[k];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int c = w.pidx[hh * 6 + k];
            idx[k] = c < 0 ? 0 : (c >= m ? m - 1 : c);
        }
    }
    const double* c3 = w.c3 + static_cast<size_t>(p) * prm.cap * 3;
    const double* c2 = w.c2 + static_cast<size_t>(p) * prm.cap * 2;
    double* A = pose_lds + lane;  // A[r][c] at A[(r * 12 + c) * 64]
#define AL(r, c) A[((r) * DLT_N + (c)) * 64]
    // rows of loop_closure.cpp:249-255
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double X = c3[3 * static_cast<size_t>(idx[i])], Y = c3[3 * static_cast<size_t>(idx[i]) + 1],
                     Z = c3[3 * static_cast<size_t>(idx[i]) + 2];
        double u = c2[2 * static_cast<size_t>(idx[i])], v = c2[2 * static_cast<size_t>(idx[i]) + 1];
        if (prm.geometry_mode == SLAMX_GEOMETRY_CALIBRATED) {  // (uniform) K^-1 of an upper-triangular K with K[8] = 1
            v = (v - prm.K[5]) / prm.K[4];
            u = (u - prm.K[2] - prm.K[1] * v) / prm.K[0];
        }
        const double r0[12] = {X, Y, Z, 1.0, 0.0, 0.0, 0.0, 0.0, -u * X, -u * Y, -u * Z, -u};
        const double r1[12] = {0.0, 0.0, 0.0, 0.0, X, Y, Z, 1.0, -v * X, -v * Y, -v * Z, -v};
#pragma unroll
        for (int c = 0; c < DLT_N; ++c) {
            AL(2 * i, c) = r0[c];
            AL(2 * i + 1, c) = r1[c];
        }
    }
    // Householder QR, column by column; R ends up in the upper triangle
    double dmax = 0.0;
#pragma unroll
    for (int k = 0; k < DLT_N; ++k) {
        double v[DLT_N];
        double s = 0.0;
#pragma unroll
        for (int r = k; r < DLT_N; ++r) {
            v[r] = AL(r, k);
            s += v[r] * v[r];
        }
        if (k == DLT_N - 1) {
            dmax = fmax(dmax, fabs(v[k]));
            break;
        }
        const double nrm = sqrt(s);
        const double alpha = v[k] >= 0.0 ? -nrm : nrm;
        v[k] -= alpha;
        double vtv = 0.0;
#pragma unroll
        for (int r = k; r < DLT_N; ++r) {
            vtv += v[r] * v[r];
        }
        const double bk = vtv > 0.0 ? 2.0 / vtv : 0.0;
#pragma unroll
        for (int j = k + 1; j < DLT_N; ++j) {
            double col[DLT_N];
            double d = 0.0;
#pragma unroll
            for (int r = k; r < DLT_N; ++r) {
                col[r] = AL(r, j);
                d += v[r] * col[r];
            }
            d *= bk;
#pragma unroll
            for (int r = k; r < DLT_N; ++r) {
                AL(r, j) = col[r] - d * v[r];
            }
        }
        AL(k, k) = alpha;
        dmax = fmax(dmax, nrm);
    }
    // reciprocal diagonal; a vanishing pivot (exact data: A is singular) is replaced by a tiny one of the same sign
    double rd[DLT_N];
    const double tiny = dmax * 1e-18 + 1e-300;
#pragma unroll
    for (int k = 0; k < DLT_N; ++k) {
        const double d = AL(k, k);
        rd[k] = 1.0 / (fabs(d) < tiny ? (d < 0.0 ? -tiny : tiny) : d);
    }
    double x[DLT_N];
#pragma unroll
    for (int k = 0; k < DLT_N; ++k) {
        x[k] = 0.28867513459481287;  // 1 / sqrt(12)
    }
    bool done = false;
    for (int it = 0; it < 40; ++it) {
        if (__ballot(!done) == 0ull) {
            break;  // every lane of the wave has converged
        }
        double y[DLT_N], z[DLT_N];
        // R^T y = x (forward)
#pragma unroll
        for (int i = 0; i < DLT_N; ++i) {
            double acc = x[i];
#pragma unroll
            for (int j = 0; j < i; ++j) {
                acc -= AL(j, i) * y[j];
            }
            y[i] = acc * rd[i];
        }
        // R z = y (backward)
#pragma unroll
        for (int i = DLT_N - 1; i >= 0; --i) {
            double acc = y[i];
#pragma unroll
            for (int j = i + 1; j < DLT_N; ++j) {
                acc -= AL(i, j) * z[j];
            }
            z[i] = acc * rd[i];
        }
        double nn = 0.0, dot = 0.0;
#pragma unroll
        for (int k = 0; k < DLT_N; ++k) {
            nn += z[k] * z[k];
            dot += z[k] * x[k];
        }
        const double scale = (dot < 0.0 ? -1.0 : 1.0) * rsq_fast(nn);  // keep the direction's sign from step to step
        double moved = 0.0;
#pragma unroll
        for (int k = 0; k < DLT_N; ++k) {
            const double nx = z[k] * scale;
            moved = fmax(moved, fabs(nx - x[k]));
            x[k] = done ? x[k] : nx;
        }
        done = done || !(moved > 1e-14) || !(nn > 0.0) || !(nn < 1e300);
    }
#undef AL
    if (!active) {
        return;
    }
    // :260 column-major 3x4 map of the row-major DLT vector (reference behaviour, kept), :262-271 orthonormalisation
    double R0[9], t0[3];
    const bool calibrated = prm.geometry_mode == SLAMX_GEOMETRY_CALIBRATED;  // uniform
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            R0[r * 3 + c] = calibrated ? x[r * 4 + c] : x[c * 3 + r];
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        t0[r] = calibrated ? x[r * 4 + 3] : x[9 + r];
    }
    if (calibrated && det3(R0) < 0.0) {  // the null vector's sign: the one whose left 3 x 3 block is orientation-preserving
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            R0[k] = -R0[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            t0[k] = -t0[k];
        }
    }
    double Ur[9], Sr[3], Vr[9], Vrt[9], UVt[9], UD[9], Rm[9];
    svd3(R0, Ur, Sr, Vr);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            Vrt[i * 3 + j] = Vr[j * 3 + i];
        }
    }
    mul3(Ur, Vrt, UVt);
    const double det = det3(UVt);
    const double D[9] = {1, 0, 0, 0, 1, 0, 0, 0, det};
    mul3(Ur, D, UD);
    mul3(UD, Vrt, Rm);
    double fro = 0.0;
    for (int k = 0; k < 9; ++k) {
        fro += R0[k] * R0[k];
    }
    fro = sqrt(fro);
    if (calibrated) {
        fro = (Sr[0] + Sr[1] + Sr[2]) / 3.0;  // the scale of a scaled rotation (||.||_F is sqrt(3) times that)
    }
    double* out = w.pRt + hh * 12;
    for (int k = 0; k < 9; ++k) {
        out[k] = Rm[k];
    }
    for (int k = 0; k < 3; ++k) {
        out[9 + k] = t0[k] / fro;
    }
    if (w.pnull != nullptr) {
        for (int k = 0; k < 12; ++k) {
            w.pnull[hh * 12 + k] = x[k];
        }
    }
    w.pcounts[hh] = 0;
}

// reprojection test of loop_closure.cpp:201-215 for one (R, t): Rt = R row-major (9) then t (3)
template <typename Ptr>
__device__ __forceinline__ bool pnp_inlier(Ptr Rt, const double* K, double X, double Y, double Z, double u, double v,
                                           double threshold) {
    double tp[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        tp[r] = Rt[r * 3] * X + Rt[r * 3 + 1] * Y + Rt[r * 3 + 2] * Z + Rt[9 + r];
    }
    if (tp[2] <= 0) {
        return false;
    }
    const double q0 = tp[0] / tp[2], q1 = tp[1] / tp[2], q2 = tp[2] / tp[2];
    const double px = K[0] * q0 + K[1] * q1 + K[2] * q2;
    const double py = K[3] * q0 + K[4] * q1 + K[5] * q2;
    const double ex = u - px;
    const double ey = v - py;
    return sqrt(ex * ex + ey * ey) < threshold;
}

// The same decision from reciprocals instead of three divisions and a square root: compare the squared error computed
// with a few-ulp reciprocal against the squared threshold; only when the two are closer than 1e-9 relative (or anything is
// not finite) does the reference's own sequence decide.  The verdict is ALWAYS pnp_inlier's.
template <typename Ptr>
__device__ __forceinline__ bool pnp_inlier_quick(Ptr Rt, const double* K, double X, double Y, double Z, double u, double v,
                                                 double threshold, double thr2) {
    double tp[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        tp[r] = Rt[r * 3] * X + Rt[r * 3 + 1] * Y + Rt[r * 3 + 2] * Z + Rt[9 + r];
    }
    if (!(tp[2] > 0)) {
        return tp[2] <= 0 ? false : pnp_inlier(Rt, K, X, Y, Z, u, v, threshold);  // (NaN depth: the exact sequence)
    }
    const double inv = rcp_fast(tp[2]);
    const double q0 = tp[0] * inv, q1 = tp[1] * inv;
    const double ex = u - (K[0] * q0 + K[1] * q1 + K[2]);
    const double ey = v - (K[3] * q0 + K[4] * q1 + K[5]);
    const double e2 = ex * ex + ey * ey;
    const double gap = fabs(e2 - thr2);
    if (gap > 1e-9 * (e2 + thr2)) {
        return e2 < thr2;
    }
    return pnp_inlier(Rt, K, X, Y, Z, u, v, threshold);
}

// every PnP hypothesis against every correspondence of its frame (the structure of sampson_score_kernel: one wave per
// workgroup, the hypothesis' twelve doubles through the scalar cache)
__global__ __launch_bounds__(64) void pnp_score_kernel(PoseWork w, PoseParams prm) {
    const int p = blockIdx.y;
    const int m = w.m[p];
    const int need = prm.pnp_min > 6 ? prm.pnp_min : 6;
    const int first = blockIdx.x * (64 * SCORE_PPT);
    if (m < need || first >= m) {
        return;
    }
    const int lane = threadIdx.x;
    const double* c3 = w.c3 + static_cast<size_t>(p) * prm.cap * 3;
    const double* c2 = w.c2 + static_cast<size_t>(p) * prm.cap * 2;
    double X[SCORE_PPT], Y[SCORE_PPT], Z[SCORE_PPT], u[SCORE_PPT], v[SCORE_PPT];
    bool live[SCORE_PPT];
#pragma unroll
    for (int q = 0; q < SCORE_PPT; ++q) {
        const int i = first + q * 64 + lane;
        live[q] = i < m;
        const size_t ii = static_cast<size_t>(live[q] ? i : 0);
        X[q] = c3[3 * ii];
        Y[q] = c3[3 * ii + 1];
        Z[q] = c3[3 * ii + 2];
        u[q] = c2[2 * ii];
        v[q] = c2[2 * ii + 1];
    }
    const double* __restrict__ src = w.pRt + static_cast<size_t>(p) * prm.pnp_iterations * 12;
    int* counts = w.pcounts + static_cast<size_t>(p) * prm.pnp_iterations;
    const double thr = prm.pnp_threshold, thr2 = prm.pnp_threshold * prm.pnp_threshold;
    double nxt[12];
    {
        const uniform_doubles e = as_uniform(src);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            nxt[j] = e[j];
        }
    }
    const int H = prm.pnp_iterations;
    for (int h0 = 0; h0 < H; h0 += 64) {
        const int nh = min(64, H - h0);
        int mine = 0;
        for (int k = 0; k < nh; ++k) {
            double cur[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                cur[j] = nxt[j];
            }
            const uniform_doubles e = as_uniform(src + static_cast<size_t>(min(h0 + k + 1, H - 1)) * 12);  // wave-uniform
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                nxt[j] = e[j];
            }
            int c = 0;
#pragma unroll
            for (int q = 0; q < SCORE_PPT; ++q) {
                const bool in = live[q] & pnp_inlier_quick(cur, prm.K, X[q], Y[q], Z[q], u[q], v[q], thr, thr2);
                c += __builtin_popcountll(__ballot(in));
            }
            mine += lane == k ? c : 0;
        }
        if (lane < nh && mine != 0) {
            atomicAdd(&counts[h0 + lane], mine);
        }
    }
}

// loop_closure.cpp:217-221: strict '>' from maxInliers = 0, identity / zero as the initial best; one wave per frame
__global__ __launch_bounds__(64) void pnp_select_kernel(PoseWork w, PoseParams prm) {
    const int p = blockIdx.x;
    const int m = w.m[p];
    const int need = prm.pnp_min > 6 ? prm.pnp_min : 6;
    if (m < need) {
        return;
    }
    const int lane = threadIdx.x;
    const int* counts = w.pcounts + static_cast<size_t>(p) * prm.pnp_iterations;
    int bc = 0, bi = 0x7fffffff;
    for (int h = lane; h < prm.pnp_iterations; h += 64) {
        const int c = counts[h];
        if (c > bc) {
            bc = c;
            bi = h;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const int oc = __shfl_xor(bc, d), oi = __shfl_xor(bi, d);
        if (oc > bc || (oc == bc && oi < bi)) {
            bc = oc;
            bi = oi;
        }
    }
    if (lane == 0) {
        slamx_pose_result& r = w.res[p];
        r.pnp_inliers = bc;
        if (bc > 0) {
            const double* Rt = w.pRt + (static_cast<size_t>(p) * prm.pnp_iterations + bi) * 12;
            for (int k = 0; k < 9; ++k) {
                r.pnp_R[k] = Rt[k];
            }
            for (int k = 0; k < 3; ++k) {
                r.pnp_t[k] = Rt[9 + k];
            }
        }
    }
}

}  // namespace

void launch_pnp_dlt(const double* p3d, const double* p2d, int n_points, int n_hyp, double* R, double* t, double* nullvec,
                    hipStream_t stream) {
    const size_t lds = 4 * sizeof(PnpWork);
    static std::atomic<unsigned long long> lds_done{0};
    allow_dynamic_lds(reinterpret_cast<const void*>(pnp_dlt_kernel), lds, lds_done);
    int blocks = (n_hyp + 3) / 4;
    if (blocks > 2048) {
        blocks = 2048;
    }
    hipLaunchKernelGGL(pnp_dlt_kernel, dim3(blocks), dim3(256), lds, stream, p3d, p2d, n_points, n_hyp, R, t, nullvec);
}

void launch_inliers(const double* p3d, const double* p2d, int n, const double* K, const double* R, const double* t,
                    double threshold, int n_hyp, int* counts, hipStream_t stream) {
    hipLaunchKernelGGL(inlier_kernel, dim3(n_hyp), dim3(256), 0, stream, p3d, p2d, n, K, R, t, threshold, counts);
}

void launch_recover_pose(const double* E, const double* K, const float* pts1, const float* pts2, int n, double* poses,
                         double* KP, double* KP0, int* front, hipStream_t stream) {
    hipLaunchKernelGGL(decompose_kernel, dim3(1), dim3(64), 0, stream, E, K, poses, KP, KP0);
    int bx = (n + 255) / 256;
    if (bx > 1024) {
        bx = 1024;
    }
    hipLaunchKernelGGL(cheirality_kernel, dim3(bx < 1 ? 1 : bx, 4), dim3(256), 0, stream, KP, KP0, pts1, pts2, n, front);
}

void launch_triangulate(const double* P1, const double* P2, const float* pts1, const float* pts2, int n, float* x4,
                        double* xyz, hipStream_t stream) {
    int bx = (n + 255) / 256;
    if (bx > 2048) {
        bx = 2048;
    }
    hipLaunchKernelGGL(triangulate_kernel, dim3(bx < 1 ? 1 : bx), dim3(256), 0, stream, P1, P2, pts1, pts2, n, x4, xyz);
}

namespace {
int pose_blocks(int cap, int per_block) {
    const int b = (cap + per_block - 1) / per_block;
    return b < 1 ? 1 : b;
}
template <typename Kernel>
void allow_lds(Kernel k, size_t bytes, std::atomic<unsigned long long>& done) {
    allow_dynamic_lds(reinterpret_cast<const void*>(k), bytes, done);
}
}  // namespace

void launch_pose_gather(const PoseSource& src, const PoseWork& w, const PoseParams& prm, int nframes, int first_train_frame,
                        hipStream_t stream) {
    hipLaunchKernelGGL(pose_gather_kernel, dim3(std::min(pose_blocks(prm.cap, 256), 64), nframes), dim3(256), 0, stream, src, w,
                       prm, first_train_frame);
}

void launch_pose_load_points(const PoseWork& w, const PoseParams& prm, hipStream_t stream) {
    hipLaunchKernelGGL(pose_load_points_kernel, dim3(std::min(pose_blocks(prm.cap, 256), 256)), dim3(256), 0, stream, w, prm);
}

void launch_essential_hypotheses(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream) {
    hipLaunchKernelGGL(essential_hyp_kernel, dim3((prm.iterations + 63) / 64, npairs), dim3(64), 0, stream, w, prm);
}

void launch_sampson_score(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream) {
    hipLaunchKernelGGL(sampson_score_kernel, dim3(pose_blocks(prm.cap, 64 * SCORE_PPT), npairs), dim3(64), 0, stream, w, prm);
}

void launch_pose_select(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream) {
    hipLaunchKernelGGL(pose_select_kernel, dim3(npairs), dim3(256), 0, stream, w, prm);
}

void launch_essential_refit_pairs(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream) {
    const size_t lds = (4 * 64 * LD + 4 * LD * LD) * sizeof(double);  // chunk rows + Gram tiles of the four waves
    static std::atomic<unsigned long long> lds_done{0};
    allow_lds(essential_refit_kernel, lds, lds_done);
    hipLaunchKernelGGL(essential_refit_kernel, dim3(npairs), dim3(256), lds, stream, w, prm);
}

void launch_pose_final(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream) {
    hipLaunchKernelGGL(pose_final_kernel, dim3(npairs), dim3(256), 0, stream, w, prm);
}

void launch_cheirality_pairs(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream) {
    hipLaunchKernelGGL(cheirality_pairs_kernel, dim3(pose_blocks(prm.cap, 256), 4, npairs), dim3(256), 0, stream, w, prm);
}

void launch_triangulate_pairs(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream) {
    hipLaunchKernelGGL(triangulate_pairs_kernel, dim3(pose_blocks(prm.cap, 256), npairs), dim3(256), 0, stream, w, prm);
}

void launch_pnp_gather(const PoseSource& src, const PoseWork& w, const PoseParams& prm, int nframes, int first_train_frame,
                       hipStream_t stream) {
    hipLaunchKernelGGL(pnp_gather_kernel, dim3(nframes), dim3(1024), 0, stream, src, w, prm, first_train_frame);
}

void launch_pnp_hypotheses(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream) {
    static std::atomic<unsigned long long> lds_done{0};
    allow_lds(pnp_hyp_kernel, PNP_HYP_LDS, lds_done);
    hipLaunchKernelGGL(pnp_hyp_kernel, dim3((prm.pnp_iterations + 63) / 64, npairs), dim3(64), PNP_HYP_LDS, stream, w, prm);
}

void launch_pnp_score(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream) {
    hipLaunchKernelGGL(pnp_score_kernel, dim3(pose_blocks(prm.cap, 64 * SCORE_PPT), npairs), dim3(64), 0, stream, w, prm);
}

void launch_pnp_select(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream) {
    hipLaunchKernelGGL(pnp_select_kernel, dim3(npairs), dim3(64), 0, stream, w, prm);
}

}  // namespace slamx
