// gn.hip -- sliding-window Gauss-Newton: J^T J / J^T r accumulation and the landmark elimination
// on the FP64 matrix cores, reduced camera system solve, back-substitution.
//
// NO REFERENCE COUNTERPART: the reference's backend is declaration-only
// (include/slam/backend/backend.hpp:10-18, map.hpp:9-21).  BASELINE.json's
// north_star / configs[3] name "sliding-window Gauss-Newton J^T J (6-DoF x N landmarks) on MFMA",
// so the problem is defined by this build and checked against an independent
// dense CPU statement of the same mathematics (oracle/gn_oracle.cpp).  PARITY UNPINNED.
//
// Problem: a window of W <= 10 camera poses (world -> camera, R row-major + t) and L landmarks, a
// dense track table obs[l][c] = (u, v) (NaN = not observed), one calibrated pinhole camera.
//   p = R_c X_l + t_c,  r = (fx p.x / p.z + cx - u, fy p.y / p.z + cy - v),  used iff p.z > 1e-9
//   left perturbation (rho, phi): dp/drho = I, dp/dphi = -[p]x, dp/dX = R_c  =>  J_c (2x6), J_l (2x3)
//   H delta = b with H = J^T J + lambda I, b = -J^T r; the first n_fixed cameras are held; a landmark
//   whose 3x3 block fails the pivot test (never seen, or one view with lambda = 0) takes no part.
//
// MI355X mapping.  Per landmark the 3x3 block A_l = sum J_l^T J_l + lambda I is factored
// (A_l = L L^T) and the camera-landmark blocks E_cl = J_c^T J_l are scaled to F_cl = E_cl L^-T,
// y_l = L^-1 b_l, so that the whole elimination is ONE symmetric rank-k update
//     S = H_cc + lambda I - F F^T,   g = b_c - F y,     F: 6W x 3L
// -- the dense contraction of the window (K = 3L, 64 x 64 output with y as row 60) -- and the camera
// blocks H_cc = sum J_c^T J_c, b_c = -sum J_c^T r and the cost sum r^2 are Gram matrices of the
// stacked [J_c | r] rows.  Both run on v_mfma_f64_16x16x4_f64:
//   * a wave takes 4 landmarks at a time, lane = (landmark, camera): it linearises its observation,
//     reduces the landmark block over the 16 lanes of the landmark with DPP, factors it, and writes
//     its F rows and [J_c | r] rows into two small LDS panels (7.5 KB + 5 KB per wave, so that LDS
//     leaves room for the two waves per SIMD the registers allow: one wave's VALU phase then runs
//     under the other's MFMA phase);
//   * 10 lower-triangle tiles of F F^T (3 k-steps each) and 5 camera-pair Gram tiles (2 k-steps)
//     are accumulated in 15 x 4 f64 accumulator registers per lane over ALL chunks of the wave;
//   * the per-wave tiles are written once, summed in slice order by gn_solve_kernel (no atomics:
//     results are run-to-run identical), which assembles S, applies the gauge, factors the
//     <= 60 x 60 system in LDS and writes the camera step and the new poses;
//   * gn_backsub_kernel re-linearises (same device function, same values) and updates the landmarks.
#include "slamx_internal.h"

// Nothing in this file is pinned to a reference that computes without FMA (the Makefile's -ffp-contract=off is
// for the parity-pinned front end): contraction is allowed here, results are compared with the CPU statement
// at 1e-9 relative.
#pragma clang fp contract(fast)

namespace slamx {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int GN_CHUNK = 4;                 // landmarks per wave step (x 16 camera slots = 64 lanes)
constexpr int GN_K = GN_CHUNK * 3;          // panel columns (k) per step
constexpr int GN_Y_ROW = 60;                // row of the F panel that carries y_l
// doubles between consecutive k of the F panel: 80 = 160 dwords = 32 banks mod 64, so the four
// k-groups of an operand read (16 lanes x 8 bytes each) cover all 64 banks exactly twice
constexpr int GN_PSTRIDE = 80;
constexpr int GN_SYRK_TILES = 10;           // lower triangle of the 4 x 4 tiles of F F^T
constexpr int GN_GRAM_TILES = 5;            // camera pairs (2 x (6 Jacobian columns + r + pad))
constexpr int GN_TILES = GN_SYRK_TILES + GN_GRAM_TILES;
constexpr int GN_TILE_ELEMS = 256;

__device__ __forceinline__ void gn_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a landmark (one DPP row); every lane ends with the same bits: each step
// adds two values that both partners hold, and IEEE addition commutes
__device__ __forceinline__ double sum16(double v) {
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);  // row_half_mirror: lane i <-> 7 - i of each group of 8
    v += dpp_f64<0x140>(v);  // row_mirror: lane i <-> 15 - i of the row
    return v;
}

// 1 / x and 1 / sqrt(x) from the hardware estimates plus two Newton steps (full f64 accuracy to a few ulps);
// the IEEE-exact division / square root sequences cost ~25 instructions each, and the kernel has 4 per lane and chunk.
__device__ __forceinline__ double gn_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}
__device__ __forceinline__ double gn_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}

struct GnObs {
    bool valid;
    double r[2];
    double Jc[2][6];
    double Jl[2][3];
};

// pose: R row-major (9) + t (3); K = fx fy cx cy.  Same expressions as oracle/gn_oracle.cpp.
__device__ __forceinline__ GnObs gn_linearize(const double* pose, double X0, double X1, double X2, float u_f, float v_f,
                                              double fx, double fy, double cx, double cy) {
    GnObs o;
    const double u = static_cast<double>(u_f), v = static_cast<double>(v_f);
    double p[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        p[i] = pose[3 * i] * X0 + pose[3 * i + 1] * X1 + pose[3 * i + 2] * X2 + pose[9 + i];
    }
    // finite(u), finite(v): x - x == 0 fails for NaN and +-inf
    o.valid = (u - u == 0.0) && (v - v == 0.0) && p[2] > 1e-9;
    const double iz = gn_rcp(o.valid ? p[2] : 1.0);
    o.r[0] = fx * p[0] * iz + cx - u;
    o.r[1] = fy * p[1] * iz + cy - v;
    const double Jp[2][3] = {{fx * iz, 0.0, -fx * p[0] * iz * iz}, {0.0, fy * iz, -fy * p[1] * iz * iz}};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const double j0 = Jp[a][0], j1 = Jp[a][1], j2 = Jp[a][2];
        o.Jc[a][0] = j0;
        o.Jc[a][1] = j1;
        o.Jc[a][2] = j2;
        o.Jc[a][3] = j2 * p[1] - j1 * p[2];
        o.Jc[a][4] = j0 * p[2] - j2 * p[0];
        o.Jc[a][5] = j1 * p[0] - j0 * p[1];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o.Jl[a][k] = j0 * pose[k] + j1 * pose[3 + k] + j2 * pose[6 + k];
        }
    }
    if (!o.valid) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            o.r[a] = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                o.Jc[a][k] = 0.0;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o.Jl[a][k] = 0.0;
            }
        }
    }
    return o;
}

// Cholesky factor of the landmark block A = sum J_l^T J_l + lambda I (a = A00 A10 A11 A20 A21 A22) with the
// pivot test of DESIGN.md 4.10 (the same as oracle/gn_oracle.cpp:landmark_active): the landmark takes
// part iff p1 > 0, p2 > 1e-9 A11, p3 > 1e-9 A22.  lo = 1/l00, l10, 1/l11, l20, l21, 1/l22: the diagonal is
// kept as reciprocals because every F row and y are forward substitutions with it (an f64 division is
// ~25 instructions; 39 of them per lane and chunk were the bulk of the kernel's VALU work).
__device__ __forceinline__ bool gn_factor3(const double* a, double* lo) {
    const double p1 = a[0];
    bool ok = p1 > 0.0;
    const double i00 = gn_rsqrt(ok ? p1 : 1.0);
    const double l10 = a[1] * i00;
    const double l20 = a[3] * i00;
    const double p2 = a[2] - l10 * l10;
    ok = ok && p2 > 1e-9 * a[2];
    const double i11 = gn_rsqrt(ok ? p2 : 1.0);
    const double l21 = (a[4] - l20 * l10) * i11;
    const double p3 = a[5] - l20 * l20 - l21 * l21;
    ok = ok && p3 > 1e-9 * a[5];
    const double i22 = gn_rsqrt(ok ? p3 : 1.0);
    // a landmark that takes no part gets zero reciprocals: every forward substitution with them yields exact
    // zeros (finite operands), so no select is needed on the 21 values derived from them
    lo[0] = ok ? i00 : 0.0;
    lo[1] = l10;
    lo[2] = ok ? i11 : 0.0;
    lo[3] = l20;
    lo[4] = l21;
    lo[5] = ok ? i22 : 0.0;
    return ok;
}

struct GnView {
    const double* poses;      // [n_windows][W][12]
    double* landmarks;        // [n_windows][maxL][3]
    const float2* obs;        // [n_windows][maxL][W]
    const int* n_landmarks;   // [n_windows]
    double* partial;          // [n_windows][nslices][GN_TILES][256]
    const double* delta;      // [n_windows][64]   camera step (back-substitution)
    int W, maxL, nslices;
    double fx, fy, cx, cy, lambda;
};

// lane = (landmark ll = lane >> 4 of the chunk, camera q = lane & 15)
// The poses of the window sit in LDS (16 slots, absent cameras = camera 0: never used, their observations
// are NaN): 24 doubles per lane would push the accumulate kernel over 256 registers.
struct GnLane {
    int ll, q;
    const double* pose;
};
constexpr int GN_POSE_SLOTS = 16;

__device__ __forceinline__ void gn_load_poses(const GnView& g, int w, int lane, double* pose_lds, GnLane* L) {
    L->ll = lane >> 4;
    L->q = lane & 15;
    for (int e = lane; e < GN_POSE_SLOTS * 12; e += 64) {
        const int c = e / 12;
        pose_lds[e] = g.poses[(static_cast<size_t>(w) * g.W + (c < g.W ? c : 0)) * 12 + (e - 12 * c)];
    }
    L->pose = pose_lds + L->q * 12;
}

// ---------------------------------------------------------------- accumulate
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void gn_accumulate_kernel(GnView g) {
    __shared__ double panelF[GN_K * GN_PSTRIDE];             // [k][row], rows 0..59 cameras, 60 = y
    __shared__ double panelG[GN_GRAM_TILES * GN_CHUNK * 2 * 16];  // [camera pair][k = 2 ll + a][16 columns]
    __shared__ double pose_lds[GN_POSE_SLOTS * 12];
    const int w = static_cast<int>(blockIdx.x) / g.nslices;
    const int slice = static_cast<int>(blockIdx.x) - w * g.nslices;
    const int lane = threadIdx.x;
    const int nL = g.n_landmarks[w];
    const int nchunks = (nL + GN_CHUNK - 1) / GN_CHUNK;
    GnLane L;
    gn_load_poses(g, w, lane, pose_lds, &L);
    for (int e = lane; e < GN_K * GN_PSTRIDE; e += 64) {
        panelF[e] = 0.0;  // rows of absent cameras and rows 61..63 are never written: they stay zero
    }
    constexpr int GTILE = GN_CHUNK * 2 * 16;  // doubles per camera-pair panel
    for (int e = lane; e < GN_GRAM_TILES * GTILE; e += 64) {
        panelG[e] = 0.0;
    }
    gn_wave_sync();
    d4 acc[GN_TILES];
#pragma unroll
    for (int t = 0; t < GN_TILES; ++t) {
        acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    }
    const int col = lane & 15;
    const int kk = lane >> 4;
    const int c = L.q;  // this lane's camera
    for (int chunk = slice; chunk < nchunks; chunk += g.nslices) {
        const int l = chunk * GN_CHUNK + L.ll;
        const bool lvalid = l < nL;
        const size_t lbase = static_cast<size_t>(w) * g.maxL + (lvalid ? l : 0);
        const double X0 = g.landmarks[lbase * 3 + 0], X1 = g.landmarks[lbase * 3 + 1], X2 = g.landmarks[lbase * 3 + 2];
        const float nanf_ = __builtin_nanf("");
        const float2 uv = (lvalid && c < g.W) ? g.obs[lbase * g.W + c] : make_float2(nanf_, nanf_);
        const GnObs o = gn_linearize(L.pose, X0, X1, X2, uv.x, uv.y, g.fx, g.fy, g.cx, g.cy);
        double a[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const double* j = o.Jl[r];
            a[0] += j[0] * j[0];
            a[1] += j[1] * j[0];
            a[2] += j[1] * j[1];
            a[3] += j[2] * j[0];
            a[4] += j[2] * j[1];
            a[5] += j[2] * j[2];
            bl[0] -= j[0] * o.r[r];
            bl[1] -= j[1] * o.r[r];
            bl[2] -= j[2] * o.r[r];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            a[k] = sum16(a[k]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bl[k] = sum16(bl[k]);
        }
        a[0] += g.lambda;
        a[2] += g.lambda;
        a[5] += g.lambda;
        double lo[6];
        const bool active = gn_factor3(a, lo);
        // y = L^-1 b_l
        const double y0 = bl[0] * lo[0];
        const double y1 = (bl[1] - lo[1] * y0) * lo[2];
        const double y2 = (bl[2] - lo[3] * y0 - lo[4] * y1) * lo[5];
        const double part = active ? 1.0 : 0.0;
        double* pf = panelF + (L.ll * 3) * GN_PSTRIDE;
        if (c == 0) {
            pf[GN_Y_ROW] = y0;
            pf[GN_PSTRIDE + GN_Y_ROW] = y1;
            pf[2 * GN_PSTRIDE + GN_Y_ROW] = y2;
        }
        if (c < g.W) {
            double* gp = panelG + (c >> 1) * GTILE + (L.ll * 2) * 16 + (c & 1) * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                // F rows 6c + i: solve L f = e for the row e of E = J_c^T J_l
                const double e0 = o.Jc[0][i] * o.Jl[0][0] + o.Jc[1][i] * o.Jl[1][0];
                const double e1 = o.Jc[0][i] * o.Jl[0][1] + o.Jc[1][i] * o.Jl[1][1];
                const double e2 = o.Jc[0][i] * o.Jl[0][2] + o.Jc[1][i] * o.Jl[1][2];
                const double f0 = e0 * lo[0];
                const double f1 = (e1 - lo[1] * f0) * lo[2];
                const double f2 = (e2 - lo[3] * f0 - lo[4] * f1) * lo[5];
                pf[6 * c + i] = f0;
                pf[GN_PSTRIDE + 6 * c + i] = f1;
                pf[2 * GN_PSTRIDE + 6 * c + i] = f2;
                // a landmark that takes no part leaves its observations out of H_cc, b_c and the cost too
                gp[i] = part * o.Jc[0][i];
                gp[16 + i] = part * o.Jc[1][i];
            }
            gp[6] = part * o.r[0];
            gp[16 + 6] = part * o.r[1];
        }
        gn_wave_sync();
        // F F^T: operand element (row 16 t + col, k = 4 ks + kk) serves as A of tile row t and B of tile column t
#pragma unroll
        for (int ks = 0; ks < GN_K / 4; ++ks) {
            const double* src = panelF + (ks * 4 + kk) * GN_PSTRIDE + col;
            const double f0 = src[0], f1 = src[16], f2 = src[32], f3 = src[48];
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, f0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f0, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, f1, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(f2, f0, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(f2, f1, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(f2, f2, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(f3, f0, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f64_16x16x4f64(f3, f1, acc[7], 0, 0, 0);
            acc[8] = __builtin_amdgcn_mfma_f64_16x16x4f64(f3, f2, acc[8], 0, 0, 0);
            acc[9] = __builtin_amdgcn_mfma_f64_16x16x4f64(f3, f3, acc[9], 0, 0, 0);
        }
        // [J_c | r]^T [J_c | r] per camera pair: k = the 8 residual rows of the chunk
#pragma unroll
        for (int tp = 0; tp < GN_GRAM_TILES; ++tp) {
#pragma unroll
            for (int ks = 0; ks < GN_CHUNK * 2 / 4; ++ks) {
                const double v = panelG[tp * GTILE + (ks * 4 + kk) * 16 + col];
                acc[GN_SYRK_TILES + tp] = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc[GN_SYRK_TILES + tp], 0, 0, 0);
            }
        }
        gn_wave_sync();
    }
    double* out = g.partial + (static_cast<size_t>(w) * g.nslices + slice) * GN_TILES * GN_TILE_ELEMS;
#pragma unroll
    for (int t = 0; t < GN_TILES; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            out[t * GN_TILE_ELEMS + r * 64 + lane] = acc[t][r];  // element (row = kk + 4 r, col) of the tile
        }
    }
}

// ---------------------------------------------------------------- solve
struct GnSolveView {
    const double* partial;
    const double* poses;   // current
    double* poses_next;    // written: exp(delta) applied (or a copy when the step is skipped)
    double* delta;         // [n_windows][64]
    double* cost;          // [n_windows]  cost at the linearisation point
    int* status;           // [n_windows]  bit 0 set when the reduced system was not positive definite
    double* S_out;         // optional [n_windows][6W][6W]
    double* g_out;         // optional [n_windows][6W]
    int W, nslices, n_fixed, cost_only;
    double lambda;
};

__device__ __forceinline__ void gn_rodrigues(const double* phi, double* E) {
    const double x = phi[0], y = phi[1], z = phi[2];
    const double th2 = x * x + y * y + z * z;
    const double th = sqrt(th2);
    double a, b;
    if (th < 1e-8) {
        a = 1.0 - th2 / 6.0;
        b = 0.5 - th2 / 24.0;
    } else {
        a = sin(th) / th;
        b = (1.0 - cos(th)) / th2;
    }
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

__global__ __launch_bounds__(256) void gn_solve_kernel(GnSolveView s) {
    constexpr int LDS = 65;
    __shared__ double FF[64 * LDS];   // F F^T, later S and its Cholesky factor
    __shared__ double GM[GN_GRAM_TILES * 256];
    __shared__ double gv[64];
    __shared__ int bad;
    const int w = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = 6 * s.W;
    // sum the per-wave tiles in slice order
    const double* src = s.partial + static_cast<size_t>(w) * s.nslices * GN_TILES * GN_TILE_ELEMS;
    const int r4 = tid >> 6, ln = tid & 63;
    const int row = (ln >> 4) + 4 * r4, colx = ln & 15;
    for (int t = 0; t < GN_TILES; ++t) {
        double v = 0.0;
        for (int k = 0; k < s.nslices; ++k) {
            v += src[(static_cast<size_t>(k) * GN_TILES + t) * GN_TILE_ELEMS + tid];
        }
        if (t < GN_SYRK_TILES) {
            // t -> (ti, tj), ti >= tj: 0:(0,0) 1:(1,0) 2:(1,1) 3:(2,0) 4:(2,1) 5:(2,2) 6:(3,0) 7:(3,1) 8:(3,2) 9:(3,3)
            const int ti = t < 1 ? 0 : t < 3 ? 1 : t < 6 ? 2 : 3;
            const int tj = t - ti * (ti + 1) / 2;
            FF[(16 * ti + row) * LDS + 16 * tj + colx] = v;
            if (ti != tj) {
                FF[(16 * tj + colx) * LDS + 16 * ti + row] = v;
            }
        } else {
            GM[(t - GN_SYRK_TILES) * 256 + row * 16 + colx] = v;
        }
    }
    if (tid == 0) {
        bad = 0;
    }
    __syncthreads();
    if (tid == 0) {
        double c = 0.0;
        for (int cam = 0; cam < s.W; ++cam) {
            const int o = (cam & 1) * 8 + 6;
            c += GM[(cam >> 1) * 256 + o * 16 + o];
        }
        s.cost[w] = c;
    }
    // g = b_c - F y = -(J_c^T r) - (F F^T)[row 60]
    double gi = 0.0;
    if (tid < n) {
        const int cam = tid / 6, i = tid - 6 * cam;
        const int o = (cam & 1) * 8;
        gi = -GM[(cam >> 1) * 256 + (o + i) * 16 + o + 6] - FF[GN_Y_ROW * LDS + tid];
        if (cam < s.n_fixed) {
            gi = 0.0;
        }
    }
    __syncthreads();  // row 60 of FF read before S overwrites nothing of it (S lives in rows < 60), kept for clarity
    // S = H_cc + lambda I - F F^T, gauge applied
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        const int ci = i / 6, cj = j / 6;
        double v = -FF[i * LDS + j];
        if (ci == cj) {
            const int o = (ci & 1) * 8;
            v += GM[(ci >> 1) * 256 + (o + i - 6 * ci) * 16 + o + j - 6 * cj];
            if (i == j) {
                v += s.lambda;
            }
        }
        if (ci < s.n_fixed || cj < s.n_fixed) {
            v = i == j ? 1.0 : 0.0;
        }
        FF[i * LDS + j] = v;
    }
    if (tid < n) {
        gv[tid] = gi;
    }
    __syncthreads();
    if (s.S_out != nullptr) {
        for (int e = tid; e < n * n; e += 256) {
            s.S_out[static_cast<size_t>(w) * n * n + e] = FF[(e / n) * LDS + e % n];
        }
    }
    if (s.g_out != nullptr && tid < n) {
        s.g_out[static_cast<size_t>(w) * n + tid] = gv[tid];
    }
    if (s.cost_only != 0) {
        return;
    }
    // right-looking Cholesky of the n x n system in LDS (lower triangle), then the two substitutions
    for (int j = 0; j < n; ++j) {
        __syncthreads();
        const double d = FF[j * LDS + j];
        if (!(d > 0.0)) {
            if (tid == 0) {
                bad = 1;
            }
            break;  // uniform: every thread reads the same d
        }
        const double sd = sqrt(d);
        __syncthreads();
        if (tid == 0) {
            FF[j * LDS + j] = sd;
        }
        for (int i = j + 1 + tid; i < n; i += 256) {
            FF[i * LDS + j] /= sd;
        }
        __syncthreads();
        const int m = n - j - 1;
        for (int e = tid; e < m * m; e += 256) {
            const int i = j + 1 + e / m, k = j + 1 + e % m;
            if (k <= i) {
                FF[i * LDS + k] -= FF[i * LDS + j] * FF[k * LDS + j];
            }
        }
    }
    __syncthreads();
    const bool failed = bad != 0;
    if (!failed) {
        for (int j = 0; j < n; ++j) {  // L z = g
            __syncthreads();
            if (tid == 0) {
                gv[j] /= FF[j * LDS + j];
            }
            __syncthreads();
            if (tid > j && tid < n) {
                gv[tid] -= FF[tid * LDS + j] * gv[j];
            }
        }
        for (int j = n - 1; j >= 0; --j) {  // L^T delta = z
            __syncthreads();
            if (tid == 0) {
                gv[j] /= FF[j * LDS + j];
            }
            __syncthreads();
            if (tid < j) {
                gv[tid] -= FF[j * LDS + tid] * gv[j];
            }
        }
    }
    __syncthreads();
    if (tid < 64) {
        // slot 63 (n <= 60) carries "this step was skipped" to gn_backsub_kernel: with delta_c = 0 the
        // back-substitution would still move every landmark by A_l^-1 b_l
        s.delta[static_cast<size_t>(w) * 64 + tid] = tid == 63 ? (failed ? 1.0 : 0.0) : ((tid < n && !failed) ? gv[tid] : 0.0);
    }
    if (tid == 0 && failed) {
        atomicOr(&s.status[w], 1);
    }
    // new poses: R <- exp(phi^) R, t <- exp(phi^) t + rho
    if (tid < s.W) {
        const double* P = s.poses + (static_cast<size_t>(w) * s.W + tid) * 12;
        double* Q = s.poses_next + (static_cast<size_t>(w) * s.W + tid) * 12;
        double d6[6];
        for (int k = 0; k < 6; ++k) {
            d6[k] = failed ? 0.0 : gv[6 * tid + k];
        }
        double E[9];
        gn_rodrigues(d6 + 3, E);
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                Q[3 * i + j] = E[3 * i] * P[j] + E[3 * i + 1] * P[3 + j] + E[3 * i + 2] * P[6 + j];
            }
            Q[9 + i] = E[3 * i] * P[9] + E[3 * i + 1] * P[10] + E[3 * i + 2] * P[11] + d6[i];
        }
    }
}

// ---------------------------------------------------------------- back-substitution
// delta_X_l = A_l^-1 (b_l - sum_c E_cl^T delta_c), against the poses of the linearisation point
__global__ __launch_bounds__(64) void gn_backsub_kernel(GnView g) {
    const int w = static_cast<int>(blockIdx.x) / g.nslices;
    const int slice = static_cast<int>(blockIdx.x) - w * g.nslices;
    const int lane = threadIdx.x;
    if (g.delta[static_cast<size_t>(w) * 64 + 63] != 0.0) {
        return;  // gn_solve_kernel skipped this window's step (status bit 0): poses AND landmarks stay
    }
    const int nL = g.n_landmarks[w];
    const int nchunks = (nL + GN_CHUNK - 1) / GN_CHUNK;
    __shared__ double pose_lds[GN_POSE_SLOTS * 12];
    GnLane L;
    gn_load_poses(g, w, lane, pose_lds, &L);
    gn_wave_sync();
    const int c = L.q;
    double dc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        dc[k] = c < g.W ? g.delta[static_cast<size_t>(w) * 64 + 6 * c + k] : 0.0;
    }
    for (int chunk = slice; chunk < nchunks; chunk += g.nslices) {
        const int l = chunk * GN_CHUNK + L.ll;
        const bool lvalid = l < nL;
        const size_t lbase = static_cast<size_t>(w) * g.maxL + (lvalid ? l : 0);
        const double X0 = g.landmarks[lbase * 3 + 0], X1 = g.landmarks[lbase * 3 + 1], X2 = g.landmarks[lbase * 3 + 2];
        double a[6] = {0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0}, ed[3] = {0, 0, 0};
        const float nanf_ = __builtin_nanf("");
        const float2 uv = (lvalid && c < g.W) ? g.obs[lbase * g.W + c] : make_float2(nanf_, nanf_);
        const GnObs o = gn_linearize(L.pose, X0, X1, X2, uv.x, uv.y, g.fx, g.fy, g.cx, g.cy);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const double* j = o.Jl[r];
            a[0] += j[0] * j[0];
            a[1] += j[1] * j[0];
            a[2] += j[1] * j[1];
            a[3] += j[2] * j[0];
            a[4] += j[2] * j[1];
            a[5] += j[2] * j[2];
            bl[0] -= j[0] * o.r[r];
            bl[1] -= j[1] * o.r[r];
            bl[2] -= j[2] * o.r[r];
            // (E^T delta_c) = J_l^T (J_c delta_c)
            double jd = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                jd += o.Jc[r][k] * dc[k];
            }
            ed[0] += j[0] * jd;
            ed[1] += j[1] * jd;
            ed[2] += j[2] * jd;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            a[k] = sum16(a[k]);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bl[k] = sum16(bl[k]);
            ed[k] = sum16(ed[k]);
        }
        a[0] += g.lambda;
        a[2] += g.lambda;
        a[5] += g.lambda;
        double lo[6];
        const bool active = gn_factor3(a, lo);
        const double r0 = bl[0] - ed[0], r1 = bl[1] - ed[1], r2 = bl[2] - ed[2];
        const double z0 = r0 * lo[0];
        const double z1 = (r1 - lo[1] * z0) * lo[2];
        const double z2 = (r2 - lo[3] * z0 - lo[4] * z1) * lo[5];
        const double d2 = z2 * lo[5];
        const double d1 = (z1 - lo[4] * d2) * lo[2];
        const double d0 = (z0 - lo[1] * d1 - lo[3] * d2) * lo[0];
        if (c == 0 && lvalid && active) {
            g.landmarks[lbase * 3 + 0] = X0 + d0;
            g.landmarks[lbase * 3 + 1] = X1 + d1;
            g.landmarks[lbase * 3 + 2] = X2 + d2;
        }
    }
}

}  // namespace

void launch_gn_accumulate(const GnLaunch& a, hipStream_t stream) {
    GnView g{};
    g.poses = a.poses;
    g.landmarks = a.landmarks;
    g.obs = reinterpret_cast<const float2*>(a.obs);
    g.n_landmarks = a.n_landmarks;
    g.partial = a.partial;
    g.delta = a.delta;
    g.W = a.n_cams;
    g.maxL = a.max_landmarks;
    g.nslices = a.nslices;
    g.fx = a.K[0];
    g.fy = a.K[1];
    g.cx = a.K[2];
    g.cy = a.K[3];
    g.lambda = a.lambda;
    hipLaunchKernelGGL(gn_accumulate_kernel, dim3(static_cast<unsigned>(a.n_windows) * a.nslices), dim3(64), 0, stream, g);
}

void launch_gn_solve(const GnLaunch& a, double* poses_next, int cost_only, double* S_out, double* g_out, hipStream_t stream) {
    GnSolveView s{};
    s.partial = a.partial;
    s.poses = a.poses;
    s.poses_next = poses_next;
    s.delta = a.delta;
    s.cost = a.cost;
    s.status = a.status;
    s.S_out = S_out;
    s.g_out = g_out;
    s.W = a.n_cams;
    s.nslices = a.nslices;
    s.n_fixed = a.n_fixed;
    s.cost_only = cost_only;
    s.lambda = a.lambda;
    hipLaunchKernelGGL(gn_solve_kernel, dim3(static_cast<unsigned>(a.n_windows)), dim3(256), 0, stream, s);
}

void launch_gn_backsub(const GnLaunch& a, hipStream_t stream) {
    GnView g{};
    g.poses = a.poses;
    g.landmarks = a.landmarks;
    g.obs = reinterpret_cast<const float2*>(a.obs);
    g.n_landmarks = a.n_landmarks;
    g.partial = a.partial;
    g.delta = a.delta;
    g.W = a.n_cams;
    g.maxL = a.max_landmarks;
    g.nslices = a.nslices;
    g.fx = a.K[0];
    g.fy = a.K[1];
    g.cx = a.K[2];
    g.cy = a.K[3];
    g.lambda = a.lambda;
    hipLaunchKernelGGL(gn_backsub_kernel, dim3(static_cast<unsigned>(a.n_windows) * a.nslices), dim3(64), 0, stream, g);
}

}  // namespace slamx
