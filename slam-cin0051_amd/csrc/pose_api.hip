// pose_api.hip -- host side of the pose entry points of the C ABI (include/slamx.h): the small dense solves of rows
// a11-a14 with host pointers (slamx_pnp_dlt_batch, slamx_count_inliers_batch, slamx_recover_pose, slamx_triangulate),
// the two RANSAC loops for one frame pair (slamx_essential_ransac / _from_samples, slamx_pnp_ransac / _from_samples) and
// the pose stage of a device-resident batch (slamx_pose_*, slamx_batch_pose).
//
// Host logic only: argument checks, the reference's std::mt19937 sampling for the seeded one-pair calls
// (loop_closure.cpp:177-193), device memory and kernel sequencing.  No call here allocates device memory in its steady
// state: the one-pair calls carve a grow-only arena that belongs to the context (re-allocated only when a call needs
// more than any call before it), the batch stage owns a workspace sized at slamx_pose_create.
#include <algorithm>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "api_common.h"

using namespace slamx;

struct slamx_pose {
    slamx_ctx* ctx = nullptr;
    slamx_batch* batch = nullptr;  // nullptr: the one-pair workspace of a context
    slamx_pose_config cfg{};
    PoseParams prm{};
    PoseWork w{};
    int slots = 0;                // frame slots (pairs) the arrays hold
    std::vector<void*> blocks;    // every device allocation of the workspace
    // profiling: event pairs around every launch
    int prof_every = 0;
    long long calls = 0;
    std::vector<hipEvent_t> ev;   // [set][SLAMX_POSE_STAGE_COUNT + 1]
    int sets_used = 0;
    static constexpr int MAX_SETS = 64;
};

namespace {

template <typename T>
int pose_alloc(slamx_pose* p, T** out, size_t count) {
    SLAMX_TRY(dev_alloc(out, count));
    p->blocks.push_back(*out);
    return SLAMX_OK;
}

void pose_free(slamx_pose* p) {
    if (p == nullptr) {
        return;
    }
    for (void* b : p->blocks) {
        (void)hipFree(b);
    }
    for (hipEvent_t e : p->ev) {
        if (e != nullptr) {
            (void)hipEventDestroy(e);
        }
    }
    delete p;
}

// SLAMX_GEOMETRY_CALIBRATED normalises image points with the closed-form inverse of an upper-triangular K
int check_geometry_mode(int mode, const double* K) {
    if (mode != SLAMX_GEOMETRY_REFERENCE && mode != SLAMX_GEOMETRY_CALIBRATED) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "geometry_mode: SLAMX_GEOMETRY_REFERENCE or SLAMX_GEOMETRY_CALIBRATED");
    }
    if (mode == SLAMX_GEOMETRY_CALIBRATED &&
        !(K[0] > 0.0 && K[4] > 0.0 && K[3] == 0.0 && K[6] == 0.0 && K[7] == 0.0 && K[8] == 1.0)) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "SLAMX_GEOMETRY_CALIBRATED needs an upper-triangular K with K[8] = 1 and positive focal lengths");
    }
    return SLAMX_OK;
}

int fill_params(const slamx_pose_config& c, int cap, PoseParams* prm) {
    if (c.iterations < 1 || c.iterations > 65536 || !(c.threshold_px > 0.0) || c.pnp_iterations < 0 || c.pnp_iterations > 65536 ||
        (c.pnp_iterations > 0 && !(c.pnp_threshold > 0.0))) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pose config: 1 <= iterations <= 65536, thresholds > 0, 0 <= pnp_iterations <= 65536");
    }
    if (!(c.K[0] > 0.0) || !(c.K[4] > 0.0)) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pose config: focal lengths must be positive");
    }
    std::memcpy(prm->K, c.K, sizeof(prm->K));
    prm->fx = c.K[0];
    prm->fy = c.K[4];
    prm->cx = c.K[2];
    prm->cy = c.K[5];
    // cv::findEssentialMat scales its pixel threshold by the focal length; same convention
    const double thr = c.threshold_px / (0.5 * (prm->fx + prm->fy));
    prm->thr2 = thr * thr;
    prm->pnp_threshold = c.pnp_threshold;
    prm->iterations = c.iterations;
    prm->pnp_iterations = c.pnp_iterations;
    prm->pnp_min = c.pnp_min_correspondences;
    prm->seed = c.seed;
    prm->cap = cap;
    prm->device_sampling = 1;
    SLAMX_TRY(check_geometry_mode(c.geometry_mode, c.K));
    prm->geometry_mode = c.geometry_mode;
    return SLAMX_OK;
}

// All arrays of a workspace of `slots` frame slots, `cap` correspondences each.
int pose_workspace(slamx_pose* p, int slots, int cap, bool with_x4) {
    const size_t P = static_cast<size_t>(slots), C = static_cast<size_t>(cap);
    const size_t H = static_cast<size_t>(p->prm.iterations), Hp = static_cast<size_t>(std::max(p->prm.pnp_iterations, 1));
    PoseWork& w = p->w;
    SLAMX_TRY(pose_alloc(p, &w.n, P));
    SLAMX_TRY(pose_alloc(p, &w.x1, P * C));
    SLAMX_TRY(pose_alloc(p, &w.x2, P * C));
    SLAMX_TRY(pose_alloc(p, &w.p1, P * C));
    SLAMX_TRY(pose_alloc(p, &w.p2, P * C));
    SLAMX_TRY(pose_alloc(p, &w.idx, P * H * 8));
    SLAMX_TRY(pose_alloc(p, &w.Eh, P * H * 9));
    SLAMX_TRY(pose_alloc(p, &w.counts, P * H));
    SLAMX_TRY(pose_alloc(p, &w.flags_a, P * C));
    SLAMX_TRY(pose_alloc(p, &w.flags_b, P * C));
    SLAMX_TRY(pose_alloc(p, &w.Ebest, P * 9));
    SLAMX_TRY(pose_alloc(p, &w.Efit, P * 9));
    SLAMX_TRY(pose_alloc(p, &w.sel, P * 4));
    SLAMX_TRY(pose_alloc(p, &w.poses4, P * 48));
    SLAMX_TRY(pose_alloc(p, &w.KP, P * 48));
    SLAMX_TRY(pose_alloc(p, &w.front, P * 4));
    SLAMX_TRY(pose_alloc(p, &w.res, P));
    SLAMX_TRY(pose_alloc(p, &w.xyz, P * C * 3));
    w.x4 = nullptr;
    if (with_x4) {
        SLAMX_TRY(pose_alloc(p, &w.x4, P * C * 4));
    }
    SLAMX_TRY(pose_alloc(p, &w.m, P));
    SLAMX_TRY(pose_alloc(p, &w.c3, P * C * 3));
    SLAMX_TRY(pose_alloc(p, &w.c2, P * C * 2));
    SLAMX_TRY(pose_alloc(p, &w.pidx, P * Hp * 6));
    SLAMX_TRY(pose_alloc(p, &w.pRt, P * Hp * 12));
    SLAMX_TRY(pose_alloc(p, &w.pnull, P * Hp * 12));
    SLAMX_TRY(pose_alloc(p, &w.pcounts, P * Hp));
    p->slots = slots;
    hipStream_t s = p->ctx->stream;
    HIP_TRY(hipMemsetAsync(w.n, 0, P * sizeof(int), s));
    HIP_TRY(hipMemsetAsync(w.m, 0, P * sizeof(int), s));
    HIP_TRY(hipMemsetAsync(w.res, 0, P * sizeof(slamx_pose_result), s));
    HIP_TRY(hipMemsetAsync(w.sel, 0, P * 4 * sizeof(int), s));
    return SLAMX_OK;
}

// The grow-only scratch arena of a context: `bytes` of device memory valid until the next arena_get on the context.
int arena_get(slamx_ctx* ctx, size_t bytes, uint8_t** out) {
    if (bytes > ctx->arena_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));  // (rare: only when a call needs more than any before it)
        if (ctx->arena != nullptr) {
            (void)hipFree(ctx->arena);
            ctx->arena = nullptr;
            ctx->arena_bytes = 0;
        }
        const size_t want = std::max<size_t>(bytes + bytes / 2, 1 << 20);
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->arena), want));
        ctx->arena_bytes = want;
    }
    *out = ctx->arena;
    return SLAMX_OK;
}
struct Carver {  // 256-byte aligned pieces of an arena
    uint8_t* base = nullptr;
    size_t at = 0;
    template <typename T>
    T* take(size_t count) {
        T* p = reinterpret_cast<T*>(base + at);
        at += (count * sizeof(T) + 255) / 256 * 256;
        return p;
    }
};
size_t aligned(size_t bytes) { return (bytes + 255) / 256 * 256; }

// The one-pair workspace of a context, for at least `n` correspondences, `iterations` essential and `pnp_iterations`
// PnP hypotheses; rebuilt (rarely) when a call asks for more than it holds.
int single_workspace(slamx_ctx* ctx, int n, int iterations, int pnp_iterations, slamx_pose** out) {
    slamx_pose* sp = ctx->single_pose;
    if (sp == nullptr || sp->prm.cap < n || sp->prm.iterations < iterations || sp->prm.pnp_iterations < pnp_iterations) {
        const int cap = std::max(n + n / 2, sp != nullptr ? sp->prm.cap : 1024);
        const int H = std::max(iterations, sp != nullptr ? sp->prm.iterations : 512);
        const int Hp = std::max(pnp_iterations, sp != nullptr ? sp->prm.pnp_iterations : 128);
        if (sp != nullptr) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            pose_free(sp);
            ctx->single_pose = nullptr;
        }
        std::unique_ptr<slamx_pose, void (*)(slamx_pose*)> np(new slamx_pose(), pose_free);
        np->ctx = ctx;
        np->prm.cap = cap;
        np->prm.iterations = H;
        np->prm.pnp_iterations = Hp;
        SLAMX_TRY(pose_workspace(np.get(), 1, cap, false));
        ctx->single_pose = np.release();
        sp = ctx->single_pose;
    }
    *out = sp;
    return SLAMX_OK;
}

void rec(slamx_pose* p, int set, int idx) {
    if (set >= 0) {
        (void)hipEventRecord(p->ev[static_cast<size_t>(set) * (SLAMX_POSE_STAGE_COUNT + 1) + idx], p->ctx->stream);
    }
}

}  // namespace

namespace slamx {
void pose_release_context(slamx_ctx* ctx) {
    if (ctx->single_pose != nullptr) {
        pose_free(ctx->single_pose);
        ctx->single_pose = nullptr;
    }
    if (ctx->arena != nullptr) {
        (void)hipFree(ctx->arena);
        ctx->arena = nullptr;
        ctx->arena_bytes = 0;
    }
}
}  // namespace slamx

extern "C" {

// ------------------------------------------------------------------ a11 / a12: solvePnP, inlier counts
int slamx_pnp_dlt_batch(slamx_ctx* ctx, const double* p3d, const double* p2d, int n_points, int n_hyp, double* R,
                        double* t, double* nullvec) try {
    if (n_points < 6) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "solvePnP needs at least 6 points");  // returns false in the reference
    }
    if (n_points > 32 || n_hyp < 0 || p3d == nullptr || p2d == nullptr || R == nullptr || t == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pnp arguments invalid (6 <= n_points <= 32)");
    }
    if (n_hyp == 0) {
        return SLAMX_OK;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t H = static_cast<size_t>(n_hyp);
    Carver c;
    SLAMX_TRY(arena_get(ctx, aligned(H * n_points * 24) + aligned(H * n_points * 16) + aligned(H * 72) + aligned(H * 24) + aligned(H * 96),
                        &c.base));
    double* d3 = c.take<double>(H * n_points * 3);
    double* d2 = c.take<double>(H * n_points * 2);
    double* dR = c.take<double>(H * 9);
    double* dt = c.take<double>(H * 3);
    double* dn = c.take<double>(H * 12);
    HIP_TRY(hipMemcpyAsync(d3, p3d, H * n_points * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d2, p2d, H * n_points * 2 * sizeof(double), hipMemcpyHostToDevice, s));
    launch_pnp_dlt(d3, d2, n_points, n_hyp, dR, dt, dn, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(R, dR, H * 9 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(t, dt, H * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (nullvec != nullptr) {
        HIP_TRY(hipMemcpyAsync(nullvec, dn, H * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_count_inliers_batch(slamx_ctx* ctx, const double* p3d, const double* p2d, int n, const double* K,
                              const double* R, const double* t, int n_hyp, double threshold, int32_t* counts) try {
    if (n < 0 || n_hyp < 0 || K == nullptr || counts == nullptr || (n > 0 && (p3d == nullptr || p2d == nullptr)) ||
        (n_hyp > 0 && (R == nullptr || t == nullptr)) || !(threshold >= 0.0)) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "inlier arguments invalid");
    }
    if (n_hyp == 0) {
        return SLAMX_OK;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t H = static_cast<size_t>(n_hyp), N = static_cast<size_t>(n);
    Carver c;
    SLAMX_TRY(arena_get(ctx, aligned(N * 24) + aligned(N * 16) + aligned(72) + aligned(H * 72) + aligned(H * 24) + aligned(H * 4), &c.base));
    double* d3 = c.take<double>(N * 3);
    double* d2 = c.take<double>(N * 2);
    double* dK = c.take<double>(9);
    double* dR = c.take<double>(H * 9);
    double* dt = c.take<double>(H * 3);
    int* dc = c.take<int>(H);
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(d3, p3d, N * 3 * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d2, p2d, N * 2 * sizeof(double), hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipMemcpyAsync(dK, K, 9 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dR, R, H * 9 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dt, t, H * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    launch_inliers(d3, d2, n, dK, dR, dt, threshold, n_hyp, dc, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(counts, dc, H * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SLAMX_OK;
} SLAMX_CATCH

// ------------------------------------------------------------------ the RANSAC of verifyGeometricConsistency, one frame
int slamx_pnp_from_samples(slamx_ctx* ctx, const double* p3d, const double* p2d, int n, const double* K, const int32_t* idx,
                           int iterations, double threshold, double* R, double* t, int32_t* max_inliers) {
    return slamx_pnp_from_samples_mode(ctx, p3d, p2d, n, K, idx, iterations, threshold, SLAMX_GEOMETRY_REFERENCE, R, t, max_inliers,
                                       nullptr, nullptr);
}

int slamx_pnp_from_samples_mode(slamx_ctx* ctx, const double* p3d, const double* p2d, int n, const double* K,
                                const int32_t* idx, int iterations, double threshold, int geometry_mode, double* R, double* t,
                                int32_t* max_inliers, double* Rt_all, int32_t* counts) try {
    if (n < 6 || iterations <= 0 || iterations > 65536 || p3d == nullptr || p2d == nullptr || K == nullptr || R == nullptr ||
        t == nullptr || idx == nullptr || !(threshold > 0.0)) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "ransac arguments invalid (needs >= 6 correspondences)");
    }
    SLAMX_TRY(check_geometry_mode(geometry_mode, K));
    for (size_t i = 0; i < static_cast<size_t>(iterations) * 6; ++i) {
        if (idx[i] < 0 || idx[i] >= n) {
            return fail(SLAMX_ERR_INVALID_ARGUMENT, "pnp sample table: index out of range");
        }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    slamx_pose* sp = nullptr;
    SLAMX_TRY(single_workspace(ctx, n, 1, iterations, &sp));
    PoseParams prm = sp->prm;  // capacities of the workspace ...
    std::memcpy(prm.K, K, sizeof(prm.K));
    prm.pnp_iterations = iterations;  // ... the sizes of this call
    prm.pnp_threshold = threshold;
    prm.pnp_min = 6;
    prm.device_sampling = 0;
    prm.geometry_mode = geometry_mode;
    const PoseWork& w = sp->w;
    slamx_pose_result init{};
    init.pnp_R[0] = init.pnp_R[4] = init.pnp_R[8] = 1.0;  // :173-175 identity / zero until a hypothesis scores
    HIP_TRY(hipMemcpyAsync(w.res, &init, sizeof(init), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(w.m, &n, sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(w.c3, p3d, static_cast<size_t>(n) * 24, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(w.c2, p2d, static_cast<size_t>(n) * 16, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(w.pidx, idx, static_cast<size_t>(iterations) * 6 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    launch_pnp_hypotheses(w, prm, 1, s);
    launch_pnp_score(w, prm, 1, s);
    launch_pnp_select(w, prm, 1, s);
    HIP_TRY(hipGetLastError());
    slamx_pose_result r{};
    HIP_TRY(hipMemcpyAsync(&r, w.res, sizeof(r), hipMemcpyDeviceToHost, s));
    if (Rt_all != nullptr) {
        HIP_TRY(hipMemcpyAsync(Rt_all, w.pRt, static_cast<size_t>(iterations) * 12 * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    if (counts != nullptr) {
        HIP_TRY(hipMemcpyAsync(counts, w.pcounts, static_cast<size_t>(iterations) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(R, r.pnp_R, 72);
    std::memcpy(t, r.pnp_t, 24);
    if (max_inliers != nullptr) {
        *max_inliers = r.pnp_inliers;
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_pnp_ransac(slamx_ctx* ctx, const double* p3d, const double* p2d, int n, const double* K, int iterations,
                     double threshold, uint32_t seed, double* R, double* t, int32_t* max_inliers) try {
    if (n < 6 || iterations <= 0 || iterations > 65536) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "ransac arguments invalid (needs >= 6 correspondences)");
    }
    // loop_closure.cpp:177-193: the same generator, distribution and distinct-index rule
    std::mt19937 rng(seed);
    std::uniform_int_distribution<int> dist(0, n - 1);
    constexpr int sampleSize = 6;
    std::vector<int32_t> table(static_cast<size_t>(iterations) * sampleSize);
    for (int i = 0; i < iterations; ++i) {
        std::vector<int> sampleIndices;
        while (sampleIndices.size() < sampleSize) {
            const int idx = dist(rng);
            if (std::find(sampleIndices.begin(), sampleIndices.end(), idx) == sampleIndices.end()) {
                table[static_cast<size_t>(i) * sampleSize + sampleIndices.size()] = idx;
                sampleIndices.push_back(idx);
            }
        }
    }
    return slamx_pnp_from_samples(ctx, p3d, p2d, n, K, table.data(), iterations, threshold, R, t, max_inliers);
} SLAMX_CATCH

// ------------------------------------------------------------------ f3: essential matrix by RANSAC, one frame pair
int slamx_essential_from_samples(slamx_ctx* ctx, const float* pts1, const float* pts2, int n, const double* K,
                                 const int32_t* idx, int iterations, double threshold_px, double* E, int32_t* n_inliers,
                                 uint8_t* inlier_mask) try {
    if (iterations <= 0 || iterations > 65536 || pts1 == nullptr || pts2 == nullptr || K == nullptr || E == nullptr ||
        idx == nullptr || threshold_px <= 0.0) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "essential ransac arguments invalid");
    }
    if (n < 8) {
        // PoseEstimator::estimate refuses fewer than 8 matches (pose_estimator.cpp:21-25)
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "Cannot estimate pose, not enough matches. Required at least 8.");
    }
    for (size_t i = 0; i < static_cast<size_t>(iterations) * 8; ++i) {
        if (idx[i] < 0 || idx[i] >= n) {
            return fail(SLAMX_ERR_INVALID_ARGUMENT, "essential sample table: index out of range");
        }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    slamx_pose* sp = nullptr;
    SLAMX_TRY(single_workspace(ctx, n, iterations, 0, &sp));
    slamx_pose_config cfg{};
    std::memcpy(cfg.K, K, sizeof(cfg.K));
    cfg.iterations = iterations;
    cfg.threshold_px = threshold_px;
    PoseParams prm{};
    SLAMX_TRY(fill_params(cfg, sp->prm.cap, &prm));
    prm.device_sampling = 0;
    const PoseWork& w = sp->w;
    slamx_pose_result init{};
    HIP_TRY(hipMemcpyAsync(w.res, &init, sizeof(init), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(w.n, &n, sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(w.p1, pts1, static_cast<size_t>(n) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(w.p2, pts2, static_cast<size_t>(n) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(w.idx, idx, static_cast<size_t>(iterations) * 8 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    launch_pose_load_points(w, prm, s);
    launch_essential_hypotheses(w, prm, 1, s);
    launch_sampson_score(w, prm, 1, s);
    launch_pose_select(w, prm, 1, s);
    for (int round = 0; round < POSE_REFIT_ROUNDS; ++round) {
        launch_essential_refit_pairs(w, prm, 1, s);
        launch_pose_final(w, prm, 1, s);
    }
    HIP_TRY(hipGetLastError());
    slamx_pose_result r{};
    HIP_TRY(hipMemcpyAsync(&r, w.res, sizeof(r), hipMemcpyDeviceToHost, s));
    if (inlier_mask != nullptr) {
        HIP_TRY(hipMemcpyAsync(inlier_mask, w.flags_b, static_cast<size_t>(n), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(E, r.E, 72);
    if (n_inliers != nullptr) {
        *n_inliers = r.n_inliers;
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_essential_ransac(slamx_ctx* ctx, const float* pts1, const float* pts2, int n, const double* K, int iterations,
                           double threshold_px, uint32_t seed, double* E, int32_t* n_inliers, uint8_t* inlier_mask) try {
    if (iterations <= 0 || iterations > 65536) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "essential ransac arguments invalid");
    }
    if (n < 8) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "Cannot estimate pose, not enough matches. Required at least 8.");
    }
    // the sampling rule of the reference's own RANSAC (loop_closure.cpp:177-193): mt19937,
    // uniform_int_distribution, indices redrawn until distinct -- here with an explicit seed
    std::mt19937 rng(seed);
    std::uniform_int_distribution<int> dist(0, n - 1);
    constexpr int sampleSize = 8;
    std::vector<int32_t> table(static_cast<size_t>(iterations) * sampleSize);
    for (int i = 0; i < iterations; ++i) {
        int32_t* row = &table[static_cast<size_t>(i) * sampleSize];
        int have = 0;
        while (have < sampleSize) {
            const int c = dist(rng);
            if (std::find(row, row + have, c) == row + have) {
                row[have++] = c;
            }
        }
    }
    return slamx_essential_from_samples(ctx, pts1, pts2, n, K, table.data(), iterations, threshold_px, E, n_inliers, inlier_mask);
} SLAMX_CATCH

// ------------------------------------------------------------------ a13 / a14 with host pointers
int slamx_recover_pose(slamx_ctx* ctx, const double* E, const float* pts1, const float* pts2, int n, const double* K,
                       double* R, double* t, int32_t* front) try {
    if (n < 0 || E == nullptr || K == nullptr || R == nullptr || t == nullptr || (n > 0 && (pts1 == nullptr || pts2 == nullptr))) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "recover pose arguments invalid");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t N = static_cast<size_t>(n);
    Carver c;
    SLAMX_TRY(arena_get(ctx, 2 * aligned(72) + 2 * aligned(N * 8) + 2 * aligned(48 * 8) + aligned(96) + aligned(16), &c.base));
    double* dE = c.take<double>(9);
    double* dK = c.take<double>(9);
    float* d1 = c.take<float>(N * 2);
    float* d2 = c.take<float>(N * 2);
    double* dposes = c.take<double>(48);
    double* dKP = c.take<double>(48);
    double* dKP0 = c.take<double>(12);
    int* dfront = c.take<int>(4);
    HIP_TRY(hipMemcpyAsync(dE, E, 72, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dK, K, 72, hipMemcpyHostToDevice, s));
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(d1, pts1, N * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d2, pts2, N * 8, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipMemsetAsync(dfront, 0, 4 * sizeof(int), s));
    launch_recover_pose(dE, dK, d1, d2, n, dposes, dKP, dKP0, dfront, s);
    HIP_TRY(hipGetLastError());
    double poses[48];
    int fr[4];
    HIP_TRY(hipMemcpyAsync(poses, dposes, sizeof(poses), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(fr, dfront, sizeof(fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    int best = 0, maxFront = -1;  // simple_pose_recover.cpp:67-82: first maximum wins
    for (int i = 0; i < 4; ++i) {
        if (fr[i] > maxFront) {
            maxFront = fr[i];
            best = i;
        }
    }
    std::memcpy(R, poses + best * 12, 72);
    std::memcpy(t, poses + best * 12 + 9, 24);
    if (front != nullptr) {
        std::memcpy(front, fr, sizeof(fr));
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_triangulate(slamx_ctx* ctx, const double* P1, const double* P2, const float* pts1, const float* pts2, int n,
                      float* x4, double* xyz) try {
    if (n < 0 || P1 == nullptr || P2 == nullptr || xyz == nullptr || (n > 0 && (pts1 == nullptr || pts2 == nullptr))) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "triangulate arguments invalid");
    }
    if (n == 0) {
        return SLAMX_OK;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const size_t N = static_cast<size_t>(n);
    Carver c;
    SLAMX_TRY(arena_get(ctx, 2 * aligned(96) + 2 * aligned(N * 8) + aligned(N * 16) + aligned(N * 24), &c.base));
    double* dP1 = c.take<double>(12);
    double* dP2 = c.take<double>(12);
    float* d1 = c.take<float>(N * 2);
    float* d2 = c.take<float>(N * 2);
    float* dx4 = c.take<float>(N * 4);
    double* dxyz = c.take<double>(N * 3);
    HIP_TRY(hipMemcpyAsync(dP1, P1, 96, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dP2, P2, 96, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d1, pts1, N * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d2, pts2, N * 8, hipMemcpyHostToDevice, s));
    launch_triangulate(dP1, dP2, d1, d2, n, dx4, dxyz, s);
    HIP_TRY(hipGetLastError());
    if (x4 != nullptr) {
        HIP_TRY(hipMemcpyAsync(x4, dx4, N * 16, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipMemcpyAsync(xyz, dxyz, N * 24, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SLAMX_OK;
} SLAMX_CATCH

// ------------------------------------------------------------------ the pose stage of a batch
int slamx_pose_create(slamx_batch* b, const slamx_pose_config* cfg, slamx_pose** out) try {
    if (b == nullptr || cfg == nullptr || out == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "null argument");
    }
    *out = nullptr;
    if (cfg->max_matches < 0) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pose config: max_matches must not be negative");
    }
    const int cap = cfg->max_matches > 0 ? std::min(cfg->max_matches, b->v.maxK) : b->v.maxK;
    HIP_TRY(hipSetDevice(b->det->ctx->device));
    std::unique_ptr<slamx_pose, void (*)(slamx_pose*)> p(new slamx_pose(), pose_free);
    p->ctx = b->det->ctx;
    p->batch = b;
    p->cfg = *cfg;
    SLAMX_TRY(fill_params(*cfg, cap, &p->prm));
    SLAMX_TRY(pose_workspace(p.get(), b->shape.max_frames, cap, false));
    *out = p.release();
    return SLAMX_OK;
} SLAMX_CATCH

void slamx_pose_destroy(slamx_pose* p) {
    if (p == nullptr) {
        return;
    }
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    pose_free(p);
}

int slamx_batch_pose(slamx_pose* p, int nframes, int first_train_frame) try {
    if (p == nullptr || p->batch == nullptr || nframes < 0 || nframes > p->slots || first_train_frame >= p->slots) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "batch_pose: nframes / first_train_frame out of range");
    }
    slamx_batch* b = p->batch;
    if (b->matches == nullptr) {
        return fail(SLAMX_ERR_RUNTIME, "batch_pose: the batch holds no matches (slamx_batch_match_consecutive first)");
    }
    if (nframes == 0) {
        return SLAMX_OK;
    }
    HIP_TRY(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    PoseSource src{};
    src.kps = b->v.kps;
    src.matches = b->matches;
    src.n_matches = b->n_matches;
    src.maxK = b->v.maxK;
    const PoseWork& w = p->w;
    const PoseParams& prm = p->prm;
    int set = -1;
    if (p->prof_every > 0) {
        if (p->calls % p->prof_every == 0 && p->sets_used < slamx_pose::MAX_SETS) {
            set = p->sets_used++;
        }
        p->calls++;
    }
    rec(p, set, 0);
    launch_pose_gather(src, w, prm, nframes, first_train_frame, s);
    rec(p, set, 1);
    launch_essential_hypotheses(w, prm, nframes, s);
    rec(p, set, 2);
    launch_sampson_score(w, prm, nframes, s);
    rec(p, set, 3);
    launch_pose_select(w, prm, nframes, s);
    rec(p, set, 4);
    // POSE_REFIT_ROUNDS x (refit on the current model's inliers, keep it iff its MSAC cost is not higher); the "refit"
    // bracket holds every launch of the loop but the last pose_final_kernel
    for (int round = 0; round < POSE_REFIT_ROUNDS; ++round) {
        launch_essential_refit_pairs(w, prm, nframes, s);
        if (round + 1 < POSE_REFIT_ROUNDS) {
            launch_pose_final(w, prm, nframes, s);
        }
    }
    rec(p, set, 5);
    launch_pose_final(w, prm, nframes, s);
    rec(p, set, 6);
    launch_cheirality_pairs(w, prm, nframes, s);
    rec(p, set, 7);
    launch_triangulate_pairs(w, prm, nframes, s);
    rec(p, set, 8);
    if (prm.pnp_iterations > 0) {
        launch_pnp_gather(src, w, prm, nframes, first_train_frame, s);
        rec(p, set, 9);
        launch_pnp_hypotheses(w, prm, nframes, s);
        rec(p, set, 10);
        launch_pnp_score(w, prm, nframes, s);
        launch_pnp_select(w, prm, nframes, s);
        rec(p, set, 11);
    } else {
        rec(p, set, 9);
        rec(p, set, 10);
        rec(p, set, 11);
    }
    HIP_TRY(hipGetLastError());
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_pose_download(slamx_pose* p, int nframes, slamx_pose_result* out) try {
    if (p == nullptr || out == nullptr || nframes < 0 || nframes > p->slots) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pose_download: arguments invalid");
    }
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipMemcpyAsync(out, p->w.res, sizeof(slamx_pose_result) * static_cast<size_t>(nframes), hipMemcpyDeviceToHost,
                           p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_pose_download_points(slamx_pose* p, int frame, double* xyz, uint8_t* inlier_mask, int cap, int* n) try {
    if (p == nullptr || frame < 0 || frame >= p->slots || n == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pose_download_points: arguments invalid");
    }
    HIP_TRY(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    slamx_pose_result r{};
    HIP_TRY(hipMemcpyAsync(&r, p->w.res + frame, sizeof(r), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const int have = (r.status & (SLAMX_POSE_TOO_FEW_MATCHES | SLAMX_POSE_NO_TRAIN_FRAME)) != 0 ? 0 : r.n_matches;
    *n = have;
    if (have > cap) {
        return fail(SLAMX_ERR_CAPACITY, "pose_download_points: " + std::to_string(have) + " points, capacity " + std::to_string(cap));
    }
    const size_t base = static_cast<size_t>(frame) * p->prm.cap;
    if (have > 0 && xyz != nullptr) {
        HIP_TRY(hipMemcpyAsync(xyz, p->w.xyz + base * 3, static_cast<size_t>(have) * 24, hipMemcpyDeviceToHost, s));
    }
    if (have > 0 && inlier_mask != nullptr) {
        HIP_TRY(hipMemcpyAsync(inlier_mask, p->w.flags_b + base, static_cast<size_t>(have), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_pose_download_samples(slamx_pose* p, int frame, int32_t* idx, int32_t* pnp_idx) try {
    if (p == nullptr || frame < 0 || frame >= p->slots) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pose_download_samples: arguments invalid");
    }
    HIP_TRY(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    if (idx != nullptr) {
        const size_t cnt = static_cast<size_t>(p->prm.iterations) * 8;
        HIP_TRY(hipMemcpyAsync(idx, p->w.idx + static_cast<size_t>(frame) * cnt, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    if (pnp_idx != nullptr && p->prm.pnp_iterations > 0) {
        const size_t cnt = static_cast<size_t>(p->prm.pnp_iterations) * 6;
        HIP_TRY(hipMemcpyAsync(pnp_idx, p->w.pidx + static_cast<size_t>(frame) * cnt, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_pose_download_hypotheses(slamx_pose* p, int frame, double* E_all, int32_t* counts) try {
    if (p == nullptr || frame < 0 || frame >= p->slots) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pose_download_hypotheses: arguments invalid");
    }
    HIP_TRY(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    const size_t H = static_cast<size_t>(p->prm.iterations);
    if (E_all != nullptr) {
        HIP_TRY(hipMemcpyAsync(E_all, p->w.Eh + static_cast<size_t>(frame) * H * 9, H * 72, hipMemcpyDeviceToHost, s));
    }
    if (counts != nullptr) {
        HIP_TRY(hipMemcpyAsync(counts, p->w.counts + static_cast<size_t>(frame) * H, H * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_pose_download_pnp_hypotheses(slamx_pose* p, int frame, double* Rt_all, double* null_all, int32_t* counts) try {
    if (p == nullptr || frame < 0 || frame >= p->slots || p->prm.pnp_iterations <= 0) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pose_download_pnp_hypotheses: arguments invalid (or no PnP step configured)");
    }
    HIP_TRY(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    const size_t H = static_cast<size_t>(p->prm.pnp_iterations);
    if (Rt_all != nullptr) {
        HIP_TRY(hipMemcpyAsync(Rt_all, p->w.pRt + static_cast<size_t>(frame) * H * 12, H * 96, hipMemcpyDeviceToHost, s));
    }
    if (null_all != nullptr) {
        HIP_TRY(hipMemcpyAsync(null_all, p->w.pnull + static_cast<size_t>(frame) * H * 12, H * 96, hipMemcpyDeviceToHost, s));
    }
    if (counts != nullptr) {
        HIP_TRY(hipMemcpyAsync(counts, p->w.pcounts + static_cast<size_t>(frame) * H, H * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_pose_download_correspondences(slamx_pose* p, int frame, double* p3d, double* p2d, int cap, int* n) try {
    if (p == nullptr || frame < 0 || frame >= p->slots || n == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pose_download_correspondences: arguments invalid");
    }
    HIP_TRY(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    int m = 0;
    HIP_TRY(hipMemcpyAsync(&m, p->w.m + frame, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *n = m;
    if (m > cap) {
        return fail(SLAMX_ERR_CAPACITY, "pose_download_correspondences: " + std::to_string(m) + " rows, capacity " + std::to_string(cap));
    }
    const size_t base = static_cast<size_t>(frame) * p->prm.cap;
    if (m > 0 && p3d != nullptr) {
        HIP_TRY(hipMemcpyAsync(p3d, p->w.c3 + base * 3, static_cast<size_t>(m) * 24, hipMemcpyDeviceToHost, s));
    }
    if (m > 0 && p2d != nullptr) {
        HIP_TRY(hipMemcpyAsync(p2d, p->w.c2 + base * 2, static_cast<size_t>(m) * 16, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_pose_set_profiling(slamx_pose* p, int on) try {
    if (p == nullptr || on < 0) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pose_set_profiling: arguments invalid");
    }
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    p->prof_every = on;
    p->calls = 0;
    p->sets_used = 0;
    const size_t want = on > 0 ? static_cast<size_t>(slamx_pose::MAX_SETS) * (SLAMX_POSE_STAGE_COUNT + 1) : 0;
    while (p->ev.size() < want) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        p->ev.push_back(e);
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_pose_stage_ms(slamx_pose* p, float* avg_ms, int32_t* launches) try {
    if (p == nullptr || avg_ms == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pose_stage_ms: arguments invalid");
    }
    HIP_TRY(hipSetDevice(p->ctx->device));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    for (int k = 0; k < SLAMX_POSE_STAGE_COUNT; ++k) {
        double sum = 0.0;
        for (int set = 0; set < p->sets_used; ++set) {
            float ms = 0.0f;
            const size_t o = static_cast<size_t>(set) * (SLAMX_POSE_STAGE_COUNT + 1) + k;
            HIP_TRY(hipEventElapsedTime(&ms, p->ev[o], p->ev[o + 1]));
            sum += ms;
        }
        avg_ms[k] = p->sets_used > 0 ? static_cast<float>(sum / p->sets_used) : 0.0f;
        if (launches != nullptr) {
            launches[k] = p->sets_used;
        }
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_batch_upload_results(slamx_batch* b, int frame, const slamx_keypoint* kps, int n_kp, const slamx_match* matches,
                               int n_matches) try {
    if (b == nullptr || frame < 0 || frame >= b->shape.max_frames || n_kp < 0 || n_kp > b->v.maxK || n_matches < 0 ||
        n_matches > b->v.maxK || (n_kp > 0 && kps == nullptr) || (n_matches > 0 && matches == nullptr)) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "upload_results: frame / counts out of range");
    }
    for (int i = 0; i < n_matches; ++i) {
        if (matches[i].queryIdx < 0 || matches[i].queryIdx >= n_kp || matches[i].trainIdx < 0 || matches[i].trainIdx >= b->v.maxK ||
            (i > 0 && matches[i].queryIdx <= matches[i - 1].queryIdx)) {
            return fail(SLAMX_ERR_INVALID_ARGUMENT, "upload_results: match indices out of range or queryIdx not ascending");
        }
    }
    HIP_TRY(hipSetDevice(b->det->ctx->device));
    hipStream_t s = b->det->ctx->stream;
    SLAMX_TRY(batch_match_arrays(b, s));
    const size_t base = static_cast<size_t>(frame) * b->v.maxK;
    if (n_kp > 0) {
        HIP_TRY(hipMemcpyAsync(b->v.kps + base, kps, static_cast<size_t>(n_kp) * sizeof(slamx_keypoint), hipMemcpyHostToDevice, s));
    }
    if (n_matches > 0) {
        HIP_TRY(hipMemcpyAsync(b->matches + base, matches, static_cast<size_t>(n_matches) * sizeof(slamx_match), hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipMemcpyAsync(b->v.n_kp + frame, &n_kp, sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->n_matches + frame, &n_matches, sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // pageable host memory: consumed when the call returns
    return SLAMX_OK;
} SLAMX_CATCH

}  // extern "C"
