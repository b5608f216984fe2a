// api_common.h -- shared by the host-side translation units of the C ABI (slamx_api.hip, pose_api.hip): the
// thread-local error slot, the status macros, device allocation helpers and the handle structs.
#pragma once

#include <exception>
#include <new>
#include <string>
#include <vector>

#include "slamx_internal.h"

struct slamx_ctx;
namespace slamx {
std::string& last_error_slot();  // thread-local, defined in slamx_api.hip (what slamx_last_error() returns)
void pose_release_context(slamx_ctx* ctx);  // pose_api.hip: frees what the pose entry points keep on a context
}

namespace {

inline int fail(int code, const std::string& msg) {
    slamx::last_error_slot() = msg;
    return code;
}

// No C++ exception crosses the C ABI: every int-returning entry point is a function-try-block that ends in SLAMX_CATCH
// (host allocations -- std::vector, std::string, new -- are the only sources; "out of memory" fits the small-string
// buffer, so reporting it allocates nothing).
inline int caught_exception() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        try {
            slamx::last_error_slot().assign("out of memory");
        } catch (...) {
        }
    } catch (const std::exception& e) {
        try {
            slamx::last_error_slot().assign(e.what());
        } catch (...) {
        }
    } catch (...) {
        try {
            slamx::last_error_slot().assign("unknown exception");
        } catch (...) {
        }
    }
    return SLAMX_ERR_RUNTIME;
}
#define SLAMX_CATCH                  \
    catch (...) {                    \
        return caught_exception();   \
    }

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            return fail(SLAMX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
        }                                                                                                  \
    } while (0)

#define SLAMX_TRY(expr)       \
    do {                      \
        int rc_ = (expr);     \
        if (rc_ != SLAMX_OK) { \
            return rc_;       \
        }                     \
    } while (0)

template <typename T>
int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) {
        count = 1;
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    return SLAMX_OK;
}
template <typename T>
void dev_free(T*& p) {
    if (p != nullptr) {
        (void)hipFree(p);
        p = nullptr;
    }
}

// Scoped device buffer (one-off calls outside the hot path: parity helpers, undistort).
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p != nullptr) {
            (void)hipFree(p);
        }
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes > 0 ? bytes : 1); }
    template <typename T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

}  // namespace

struct slamx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    std::string name;
    int n_cus = 0;
    // second queue of the context: the blur of a batch beside its sort + NMS (blur_beside_sort), created on first use
    hipStream_t side = nullptr;
    hipEvent_t e_fork = nullptr, e_join = nullptr;
    // third queue: the uploads of slamx_batch_upload_async (a queue that only ever carries copies), created on first use
    hipStream_t copy = nullptr;
    hipEvent_t e_copy_before = nullptr, e_copy_after = nullptr;
    // grow-only device scratch of the host-pointer pose entry points (pose_api.hip): carved per call, re-allocated only
    // when a call needs more than any call before it
    uint8_t* arena = nullptr;
    size_t arena_bytes = 0;
    struct slamx_pose* single_pose = nullptr;  // one-pair workspace behind slamx_essential_ransac / slamx_pnp_ransac
};

struct slamx_detector {
    slamx_ctx* ctx = nullptr;
    slamx_detector_config cfg{};
    std::vector<int32_t> pattern;  // x1,y1,x2,y2 per surviving pair
    int8_t* pattern_dev = nullptr;
    uint2* moment_weights_dev = nullptr;  // moment_weight_table() for PatchSize / 2 <= 15
    double* weights_dev = nullptr;
    slamx_batch* single = nullptr;  // lazily created 1-frame batch behind the host-buffer API
    // Page-locked staging of that batch (hipHostMalloc, grown on demand): [packed results | image rows | keypoints in].
    // A host-buffer call copies its image (and keypoints) here, queues ONE asynchronous upload per array and the
    // kernels behind it, lets the pack kernel write status, count, keypoints and descriptors back into the block and
    // synchronises once (until round 4: a synchronous pageable upload and four synchronous copies back).
    uint8_t* stage = nullptr;
    size_t stage_bytes = 0;
};

struct slamx_matcher {
    slamx_ctx* ctx = nullptr;
    slamx_matcher_config cfg{};
    // scratch behind slamx_matcher_match: a 2-"frame" layout, frame 0 = train, frame 1 = query
    int cap = 0, width = 0;
    uint8_t* desc = nullptr;
    slamx_keypoint* kps = nullptr;
    int* n_kp = nullptr;
    slamx_match* matches = nullptr;
    int* n_matches = nullptr;
    int2* best = nullptr;
    int* second = nullptr;
    uint8_t* xdesc = nullptr;  // [2][cap][64]
    // desc | kps | n_kp above are carved out of ONE device block and matches | n_matches out of another, mirrored by ONE
    // page-locked host block: a call is one copy in, the kernels, one copy out and one synchronisation (until round 4:
    // five copies from pageable memory, three synchronisations -- 176 us per call on 1.1k keypoints, mostly those)
    uint8_t* in_blob = nullptr;
    uint8_t* out_blob = nullptr;
    uint8_t* stage = nullptr;  // hipHostMalloc: [in_blob image][out tail image]
    size_t in_bytes = 0, out_tail_bytes = 0;
};

struct slamx_batch {
    slamx_detector* det = nullptr;
    slamx_batch_shape shape{};
    slamx::BatchView v{};
    uint8_t* frames = nullptr;
    // matcher outputs (allocated on first use)
    slamx_match* matches = nullptr;
    int* n_matches = nullptr;
    int2* best = nullptr;
    int* second = nullptr;
    uint8_t* xdesc = nullptr;  // [F][maxK][64] expanded descriptors of the matrix-core matcher
    uint8_t* pack = nullptr;   // packed results of slamx_batch_download_packed_async (worst-case size, on first use)
    size_t pack_bytes = 0;
    // profiling: n_sets event sets of NEV events; event i of a set brackets the stages
    // 0|FAST|1|SCAN|2|EXPAND|3|SORT|4|NMS_BUILD|5|NMS_ROUNDS|6|BLUR|7|MOMENTS|8|TRIG|9|BRIEF|10|MATCH|11|EMIT|12
    static constexpr int NEV = SLAMX_STAGE_COUNT + 1;
    int n_sets = 0;
    std::vector<hipEvent_t> ev;
    std::vector<char> ev_valid;
    long long calls = 0;  // detect_and_compute calls since profiling was switched on
    int cur_set = 0;
    int every = 1;        // only every `every`-th call records events (an event between two kernels keeps
                          // the second from ramping up while the first drains: ~9 % on the bench step)
    bool active = false;  // the current call records
    int last_frames = 0;
    int last_status = 0;  // per-frame status bits seen by the most recent slamx_batch_download
    // What the host last SAW of the batch's keypoint counts (slamx_batch_counts): -1 nothing yet, 0 no frame reached the
    // density from which the describe stage uses shared region tiles, 1 some frame did.  A speed hint only: with 0 the
    // two launches of that path are skipped (25-30 us per 512-frame step of sparse frames) and the per-keypoint kernel
    // takes every frame, whatever its density.
    int dense_hint = -1;
};

namespace {
// The matcher outputs of a batch, on first use: all four arrays or none (a failed allocation leaves the batch as it was).
inline int batch_match_arrays(slamx_batch* b, hipStream_t stream) {
    if (b->matches != nullptr) {
        return SLAMX_OK;
    }
    const size_t F = static_cast<size_t>(b->shape.max_frames), K = static_cast<size_t>(b->v.maxK);
    slamx_match* matches = nullptr;
    int* n_matches = nullptr;
    int2* best = nullptr;
    int* second = nullptr;
    int rc = dev_alloc(&matches, F * K);
    rc = rc != SLAMX_OK ? rc : dev_alloc(&n_matches, F);
    rc = rc != SLAMX_OK ? rc : dev_alloc(&best, F * slamx::MATCH_MAX_SPLIT * K);
    rc = rc != SLAMX_OK ? rc : dev_alloc(&second, F * slamx::MATCH_MAX_SPLIT * K);
    if (rc == SLAMX_OK && hipMemsetAsync(n_matches, 0, F * sizeof(int), stream) != hipSuccess) {
        rc = fail(SLAMX_ERR_HIP, "hipMemsetAsync(n_matches) failed");
    }
    if (rc != SLAMX_OK) {
        dev_free(matches);
        dev_free(n_matches);
        dev_free(best);
        dev_free(second);
        return rc;
    }
    b->matches = matches;
    b->n_matches = n_matches;
    b->best = best;
    b->second = second;
    return SLAMX_OK;
}
}  // namespace


