// slamx_api.hip -- host side of the C ABI declared in include/slamx.h.
//
// Host logic only: configuration, the BRIEF pattern and blur weights (constructor-time work the
// reference also does on the CPU, with the same <random> / std::exp calls), device memory and
// kernel sequencing, and the order-defining std::partial_sort / std::sort of
// filterAndSortMatches (feature_matcher.cpp:191-204, kept on the host with the SAME libstdc++
// calls: SURVEY.md section 7, hard part 4).  Every pixel / descriptor / pair operation of the
// hot path runs in the HIP kernels; there is no CPU implementation to fall back to.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "api_common.h"
#include <dlfcn.h>
#include "yaml_lite.hpp"

using namespace slamx;

namespace slamx {
std::string& last_error_slot() {
    thread_local std::string g_last_error;
    return g_last_error;
}
}  // namespace slamx

namespace {

// generateBRIEFPattern, src/frontend/feature_detector.cpp:286-313 -- constructor-time, host.
std::vector<int32_t> make_pattern(int patch_size, int num_pairs) {
    std::vector<int32_t> out;
    const float scale = static_cast<float>(patch_size) / 2.0F;
    std::default_random_engine generator;
    std::normal_distribution<float> distribution(0.0F, 1.0F);
    for (int i = 0; i < num_pairs; ++i) {
        const float x1 = distribution(generator) * scale;
        const float y1 = distribution(generator) * scale;
        const float x2 = distribution(generator) * scale;
        const float y2 = distribution(generator) * scale;
        if (std::abs(x1) < scale && std::abs(y1) < scale && std::abs(x2) < scale && std::abs(y2) < scale) {
            out.push_back(static_cast<int>(x1));
            out.push_back(static_cast<int>(y1));
            out.push_back(static_cast<int>(x2));
            out.push_back(static_cast<int>(y2));
        }
    }
    return out;
}

// the kernel of gaussianBlur(image, 5, 1.0), feature_detector.cpp:321-335
void make_blur_weights(double* w25) {
    const int half = 2;
    const double sigma = 1.0;
    double sum = 0.0;
    for (int i = -half; i <= half; ++i) {
        for (int j = -half; j <= half; ++j) {
            const double value = std::exp(-((i * i) + (j * j)) / (2 * sigma * sigma));
            w25[(i + half) * 5 + (j + half)] = value;
            sum += value;
        }
    }
    for (int k = 0; k < 25; ++k) {
        w25[k] /= sum;
    }
}

int validate_detector_config(const slamx_detector_config& c) {
    // messages follow include/slam/frontend/feature_detector.hpp:60-93
    if (c.intensity_threshold < 0 || c.intensity_threshold > 255) {
        return fail(SLAMX_ERR_RUNTIME, "Intensity threshold must be in the range [0, 255].");
    }
    if (c.contiguous_pixels_threshold < 0 || c.contiguous_pixels_threshold > 16) {
        return fail(SLAMX_ERR_RUNTIME, "Contiguous pixels threshold must be in the range [0, 16].");
    }
    if (c.non_max_suppression != 0 && c.non_max_suppression != 1) {
        return fail(SLAMX_ERR_RUNTIME, "Non-max suppression must be either 0 (false) or 1 (true).");
    }
    if (c.suppression_window_size <= 0) {
        return fail(SLAMX_ERR_RUNTIME, "Suppression window size must be a positive integer.");
    }
    if (c.patch_size <= 0 || c.patch_size % 2 == 0) {
        return fail(SLAMX_ERR_RUNTIME, "Patch size must be a positive odd integer.");
    }
    if (c.num_brief_pairs <= 0 || c.num_brief_pairs % 8 != 0) {
        return fail(SLAMX_ERR_RUNTIME, "Number of BRIEF pairs must be a positive multiple of 8.");
    }
    // limits of this implementation (not of the reference)
    if (c.num_brief_pairs > MAX_PATTERN) {
        return fail(SLAMX_ERR_RUNTIME, "NumBRIEFPairs above 1024 is not supported by the MI355X build.");
    }
    if (c.patch_size > 255) {
        return fail(SLAMX_ERR_RUNTIME, "PatchSize above 255 is not supported by the MI355X build.");
    }
    return SLAMX_OK;
}

DetectorParams params_of(const slamx_detector* d) {
    DetectorParams p;
    p.threshold = d->cfg.intensity_threshold;
    p.arc = d->cfg.contiguous_pixels_threshold;
    p.nms = d->cfg.non_max_suppression;
    p.window = d->cfg.suppression_window_size;
    p.patch = d->cfg.patch_size;
    p.npairs = d->cfg.num_brief_pairs;
    p.pattern_len = static_cast<int>(d->pattern.size() / 4);
    p.moment_weights = d->moment_weights_dev;
    return p;
}

int status_to_error(int st, int frame) {
    if (st & STATUS_CAND_OVERFLOW) {
        return fail(SLAMX_ERR_CAPACITY, "frame " + std::to_string(frame) + ": FAST candidates exceed max_candidates");
    }
    if (st & STATUS_KP_OVERFLOW) {
        return fail(SLAMX_ERR_CAPACITY, "frame " + std::to_string(frame) + ": keypoints exceed max_keypoints");
    }
    if (st & STATUS_SEG_OVERFLOW) {
        return fail(SLAMX_ERR_CAPACITY, "frame " + std::to_string(frame) + ": sort segment list overflow");
    }
    return SLAMX_OK;
}

void record(slamx_batch* b, int idx) {
    if (b->n_sets > 0 && b->active) {
        const size_t k = static_cast<size_t>(b->cur_set) * slamx_batch::NEV + idx;
        (void)hipEventRecord(b->ev[k], b->det->ctx->stream);
        b->ev_valid[k] = 1;
    }
}
hipEvent_t event_slot(slamx_batch* b, int idx) {  // mark as recorded by a launcher
    if (b->n_sets <= 0 || !b->active) {
        return nullptr;
    }
    const size_t k = static_cast<size_t>(b->cur_set) * slamx_batch::NEV + idx;
    b->ev_valid[k] = 1;
    return b->ev[k];
}
bool recorded(const slamx_batch* b, int idx) {
    return b->n_sets > 0 && b->active && b->ev_valid[static_cast<size_t>(b->cur_set) * slamx_batch::NEV + idx] != 0;
}
void begin_set(slamx_batch* b) {
    if (b->n_sets > 0) {
        b->active = b->calls % b->every == 0;
        if (b->active) {
            b->cur_set = static_cast<int>((b->calls / b->every) % b->n_sets);
            for (int i = 0; i < slamx_batch::NEV; ++i) {
                b->ev_valid[static_cast<size_t>(b->cur_set) * slamx_batch::NEV + i] = 0;
            }
        }
        b->calls++;
    }
}

// detect part: FAST bitmap -> prefix -> candidates -> exact sort -> greedy NMS
void enqueue_detect(slamx_batch* b, int nframes) {
    hipStream_t s = b->det->ctx->stream;
    const DetectorParams p = params_of(b->det);
    record(b, 0);
    launch_fast_detect(b->v, p, nframes, s);
    record(b, 1);
    launch_scan(b->v, nframes, s);
    record(b, 2);
    launch_expand(b->v, p, nframes, s);
    record(b, 3);
    if (p.nms != 0) {
        launch_sort(b->v, nframes, s, !nms_uses_greedy_path(b->v, p));
        record(b, 4);
        launch_nms(b->v, p, nframes, s, event_slot(b, 5));
        record(b, 6);
    } else {
        launch_passthrough_keypoints(b->v, nframes, s);
        record(b, 4);
        record(b, 5);
        record(b, 6);
    }
}

// FAST bitmap -> prefix -> candidates, without events (the overlapped form below is not taken on profiled calls)
void enqueue_detect_until_sort(slamx_batch* b, int nframes) {
    hipStream_t s = b->det->ctx->stream;
    const DetectorParams p = params_of(b->det);
    launch_fast_detect(b->v, p, nframes, s);
    launch_scan(b->v, nframes, s);
    launch_expand(b->v, p, nframes, s);
}

// Whether this call queues the blur beside sort + NMS (see slamx_batch_detect_and_compute): NMS on, the sort a single
// round of workgroups, a batch large enough to be worth two more events, and no per-stage events on this call.
// SLAMX_BLUR_BESIDE_SORT=0 / 1 switches it off / on for every batch of at least two frames (experiments, tests).
bool blur_beside_sort(slamx_batch* b, int nframes) {
    static const int forced = std::getenv("SLAMX_BLUR_BESIDE_SORT") ? std::atoi(std::getenv("SLAMX_BLUR_BESIDE_SORT")) : -1;
    slamx_ctx* c = b->det->ctx;
    if (forced == 0 || b->det->cfg.non_max_suppression == 0 || (b->n_sets > 0 && b->active) || nframes < 2) {
        return false;
    }
    if (forced != 1 && (nframes < 16 || nframes > c->n_cus)) {
        return false;
    }
    if (c->side == nullptr) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
            hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, lo) != hipSuccess) {
            c->side = nullptr;
        }
    }
    if (c->side != nullptr && c->e_fork == nullptr && hipEventCreateWithFlags(&c->e_fork, hipEventDisableTiming) != hipSuccess) {
        c->e_fork = nullptr;
    }
    if (c->side != nullptr && c->e_join == nullptr && hipEventCreateWithFlags(&c->e_join, hipEventDisableTiming) != hipSuccess) {
        c->e_join = nullptr;
    }
    if (c->side == nullptr || c->e_fork == nullptr || c->e_join == nullptr) {
        (void)hipGetLastError();
        return false;  // the serial order is always available; what was created is released with the context
    }
    return true;
}

// compute part: blur -> orientation + BRIEF
void enqueue_compute(slamx_batch* b, int nframes, bool after_nms) {
    hipStream_t s = b->det->ctx->stream;
    const DetectorParams p = params_of(b->det);
    if (!recorded(b, 6)) {
        record(b, 6);
    }
    launch_blur(b->v, b->det->weights_dev, nframes, s);
    record(b, 7);
    hipEvent_t e_mom = event_slot(b, 8);
    hipEvent_t e_trig = event_slot(b, 9);
    launch_describe(b->v, p, b->det->pattern_dev, nframes, after_nms && p.nms != 0, s, e_mom, e_trig, b->dense_hint != 0);
    record(b, 10);
}

// Candidate capacity of the implicit one-frame batch behind the host-buffer API, which has no
// capacity knob.  First sizing: a quarter of the pixels (natural images give 1-2 %); `every_pixel`
// is the second sizing, taken when a frame overflowed the first: every pixel a candidate, up to
// the hard limit of 2^20 (the sort key holds a 20-bit raster index, SCORE_SHIFT).
long long single_capacity(int rows, int cols, bool every_pixel) {
    const long long px = static_cast<long long>(rows) * cols;
    long long cap = every_pixel ? px : px / 4;
    if (cap < 1024) {
        cap = 1024;
    }
    if (cap > MAX_CANDIDATES_HARD) {
        cap = MAX_CANDIDATES_HARD;
    }
    return cap;
}

int ensure_single(slamx_detector* det, int rows, int cols, int min_keypoints, bool every_pixel = false) {
    slamx_batch* b = det->single;
    const long long cap = single_capacity(rows, cols, every_pixel);
    if (b != nullptr && b->shape.rows == rows && b->shape.cols == cols && b->v.maxK >= min_keypoints &&
        b->v.maxM >= cap) {
        return SLAMX_OK;
    }
    if (b != nullptr) {
        slamx_batch_destroy(b);
        det->single = nullptr;
    }
    slamx_batch_shape sh{};
    sh.rows = rows;
    sh.cols = cols;
    sh.max_frames = 1;
    sh.max_candidates = static_cast<int>(cap);
    sh.max_keypoints = std::max(static_cast<int>(cap), min_keypoints);
    return slamx_batch_create(det, &sh, &det->single);
}

// true when the one-frame batch overflowed its candidate capacity and a larger one can be had:
// the caller re-runs the frame once with every_pixel sizing (the reference has no such limit)
bool single_can_grow(const slamx_detector* det, int rc, int rows, int cols) {
    const slamx_batch* b = det->single;
    return rc == SLAMX_ERR_CAPACITY && b != nullptr && (b->last_status & STATUS_CAND_OVERFLOW) != 0 &&
           b->v.maxM < single_capacity(rows, cols, true);
}

// The staging block of the one-frame batch (see slamx_detector::stage): out = the packed results (worst case of the
// batch's capacity), img = rows x cols bytes, kps_in = the keypoints slamx_compute is given.
struct SingleStage {
    uint8_t* out = nullptr;
    size_t out_bytes = 0;
    uint8_t* img = nullptr;
    slamx_keypoint* kps_in = nullptr;
    int* n_in = nullptr;
};
int single_stage(slamx_detector* det, SingleStage* st) {
    const slamx_batch* b = det->single;
    const auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t out_bytes = up(packed_bound_bytes(1, b->v.maxK, b->v.width, true));
    const size_t img_bytes = up(static_cast<size_t>(b->v.rows) * b->v.cols);
    const size_t kps_bytes = up(static_cast<size_t>(b->v.maxK) * sizeof(slamx_keypoint));
    const size_t want = out_bytes + img_bytes + kps_bytes + 256;
    if (det->stage_bytes < want) {
        HIP_TRY(hipStreamSynchronize(det->ctx->stream));
        if (det->stage != nullptr) {
            (void)hipHostFree(det->stage);
            det->stage = nullptr;
            det->stage_bytes = 0;
        }
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&det->stage), want, hipHostMallocDefault));
        det->stage_bytes = want;
    }
    st->out = det->stage;
    st->out_bytes = out_bytes;
    st->img = det->stage + out_bytes;
    st->kps_in = reinterpret_cast<slamx_keypoint*>(det->stage + out_bytes + img_bytes);
    st->n_in = reinterpret_cast<int*>(det->stage + out_bytes + img_bytes + kps_bytes);
    return SLAMX_OK;
}

// image -> staging block -> frame 0 of the one-frame batch (asynchronous: the block is page-locked and ours)
int single_upload(slamx_detector* det, const SingleStage& st, const uint8_t* img, ptrdiff_t stride) {
    slamx_batch* b = det->single;
    const size_t cols = static_cast<size_t>(b->v.cols);
    if (static_cast<size_t>(stride) == cols) {
        std::memcpy(st.img, img, cols * b->v.rows);
    } else {
        for (int r = 0; r < b->v.rows; ++r) {
            std::memcpy(st.img + r * cols, img + static_cast<ptrdiff_t>(r) * stride, cols);
        }
    }
    HIP_TRY(hipMemcpy2DAsync(b->frames, static_cast<size_t>(b->v.pitch), st.img, cols, cols, static_cast<size_t>(b->v.rows),
                             hipMemcpyHostToDevice, det->ctx->stream));
    return SLAMX_OK;
}

// status, count, keypoints and descriptors of frame 0: packed into the staging block by the pack kernel, one
// synchronisation, then plain copies into the caller's buffers.  Same results and errors as slamx_batch_download.
int single_download(slamx_detector* det, const SingleStage& st, slamx_keypoint* kps, uint8_t* desc, int cap, int* n) {
    slamx_batch* b = det->single;
    hipStream_t s = det->ctx->stream;
    void* mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, st.out, 0) != hipSuccess || mapped == nullptr) {
        (void)hipGetLastError();
        return slamx_batch_download(b, 0, kps, desc, cap, n);  // (page-locked memory the device cannot address: the copies)
    }
    if (b->pack == nullptr) {
        const size_t pcap = packed_bound_bytes(b->shape.max_frames, b->v.maxK, b->v.width, true);
        SLAMX_TRY(dev_alloc(&b->pack, pcap));
        b->pack_bytes = pcap;
    }
    launch_pack_results(b->v, b->matches, b->n_matches, 1, b->pack, s, static_cast<uint8_t*>(mapped), st.out_bytes);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    slamx_packed_header h;
    std::memcpy(&h, st.out, sizeof(h));
    int status = 0;
    std::memcpy(&status, st.out + h.status_off, sizeof(int));
    b->last_status = status;
    SLAMX_TRY(status_to_error(status, 0));
    const int cnt = h.total_keypoints;
    if (n != nullptr) {
        *n = cnt;
    }
    if (cnt > cap) {
        return fail(SLAMX_ERR_CAPACITY, "keypoint buffer too small");
    }
    if (kps != nullptr && cnt > 0) {
        std::memcpy(kps, st.out + h.kps_off, sizeof(slamx_keypoint) * static_cast<size_t>(cnt));
    }
    if (desc != nullptr && cnt > 0) {
        std::memcpy(desc, st.out + h.desc_off, static_cast<size_t>(cnt) * b->v.width);
    }
    return SLAMX_OK;
}

int check_image(const uint8_t* img, int rows, int cols, ptrdiff_t stride) {
    if (img == nullptr || rows <= 0 || cols <= 0 || stride < cols) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "image pointer/shape/stride invalid");
    }
    if (rows > 65535 || cols > 65535) {
        return fail(SLAMX_ERR_RUNTIME, "images larger than 65535 in a dimension are not supported");
    }
    return SLAMX_OK;
}

// filterAndSortMatches, src/frontend/feature_matcher.cpp:191-204 -- same libstdc++ calls.
void filter_and_sort(std::vector<slamx_match>& m, int good_matches_count) {
    const auto pred = [](const slamx_match& a, const slamx_match& b) { return a.distance < b.distance; };
    if (m.size() > static_cast<size_t>(good_matches_count)) {
        std::partial_sort(m.begin(), m.begin() + good_matches_count, m.end(), pred);
        m.erase(m.begin() + good_matches_count, m.end());
    } else {
        std::sort(m.begin(), m.end(), pred);
    }
}

// Train-set split of the matcher: enough workgroups for ~8 resident waves per SIMD (256 CUs x 4
// SIMDs) even when only ~half of the maxK query slots of a frame are used; 512 queries per
// 4-wave workgroup.
int match_split_for(int npairs, int maxK) {
    const long long waves = static_cast<long long>(npairs > 0 ? npairs : 1) * ((maxK + 511) / 512) * 4;
    long long s = (8192 + waves - 1) / waves;
    if (s < 1) {
        s = 1;
    }
    if (s > MATCH_MAX_SPLIT) {
        s = MATCH_MAX_SPLIT;
    }
    return static_cast<int>(s);
}

// Train-set size from which the matrix-core scan is taken; SLAMX_MATCH_MFMA_MIN_KEYPOINTS overrides the
// built-in threshold (the parity tests run small sets through both scans that way).
int match_mfma_min_keypoints() {
    const char* e = std::getenv("SLAMX_MATCH_MFMA_MIN_KEYPOINTS");
    if (e != nullptr && *e != '\0') {
        return std::atoi(e);
    }
    return MATCH_MFMA_MIN_KEYPOINTS;
}
// ... and the lower size for frames whose keypoints are spread wide (never above the general threshold)
int match_mfma_min_keypoints_wide() { return std::min(MATCH_MFMA_MIN_KEYPOINTS_WIDE, match_mfma_min_keypoints()); }

uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace

extern "C" {

const char* slamx_last_error(void) { return slamx::last_error_slot().c_str(); }
const char* slamx_version(void) { return "slamx 0.2 (gfx950)"; }
int slamx_matcher_mfma_threshold(void) { return match_mfma_min_keypoints(); }

int slamx_device_count(void) try {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
} SLAMX_CATCH

int slamx_ctx_create(int device_ordinal, void* hip_stream, slamx_ctx** out) try {
    if (out == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "out is null");
    }
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        return fail(SLAMX_ERR_NO_DEVICE, "no HIP device available: the slamx hot path has no CPU fallback");
    }
    if (device_ordinal < 0 || device_ordinal >= count) {
        return fail(SLAMX_ERR_NO_DEVICE, "device ordinal out of range");
    }
    HIP_TRY(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
    auto* c = new slamx_ctx();
    c->device = device_ordinal;
    c->name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    c->n_cus = prop.multiProcessorCount;
    if (hip_stream != nullptr) {
        c->stream = static_cast<hipStream_t>(hip_stream);
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return fail(SLAMX_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
        }
        c->owns_stream = true;
    }
    hipError_t ee = hipEventCreate(&c->t0);
    if (ee == hipSuccess) {
        ee = hipEventCreate(&c->t1);
    }
    if (ee != hipSuccess) {
        slamx_ctx_destroy(c);
        return fail(SLAMX_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(ee));
    }
    *out = c;
    return SLAMX_OK;
} SLAMX_CATCH

void slamx_ctx_destroy(slamx_ctx* ctx) {
    if (ctx == nullptr) {
        return;
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream != nullptr) {
        (void)hipStreamSynchronize(ctx->stream);
    }
    slamx::pose_release_context(ctx);  // the one-pair pose workspace and the scratch arena (pose_api.hip)
    if (ctx->t0 != nullptr) {
        (void)hipEventDestroy(ctx->t0);
    }
    if (ctx->t1 != nullptr) {
        (void)hipEventDestroy(ctx->t1);
    }
    if (ctx->side != nullptr) {
        (void)hipStreamSynchronize(ctx->side);
        (void)hipStreamDestroy(ctx->side);
    }
    if (ctx->copy != nullptr) {
        (void)hipStreamSynchronize(ctx->copy);
        (void)hipStreamDestroy(ctx->copy);
    }
    if (ctx->e_copy_before != nullptr) {
        (void)hipEventDestroy(ctx->e_copy_before);
    }
    if (ctx->e_copy_after != nullptr) {
        (void)hipEventDestroy(ctx->e_copy_after);
    }
    if (ctx->e_fork != nullptr) {
        (void)hipEventDestroy(ctx->e_fork);
    }
    if (ctx->e_join != nullptr) {
        (void)hipEventDestroy(ctx->e_join);
    }
    if (ctx->owns_stream && ctx->stream != nullptr) {
        (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

int slamx_ctx_synchronize(slamx_ctx* ctx) try {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SLAMX_OK;
} SLAMX_CATCH
void* slamx_ctx_stream(slamx_ctx* ctx) { return ctx->stream; }
int slamx_ctx_device_name(slamx_ctx* ctx, char* buf, int cap) try {
    if (buf == nullptr || cap <= 0) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "buffer");
    }
    std::snprintf(buf, static_cast<size_t>(cap), "%s", ctx->name.c_str());
    return SLAMX_OK;
} SLAMX_CATCH
int slamx_ctx_timer_start(slamx_ctx* ctx) try {
    HIP_TRY(hipEventRecord(ctx->t0, ctx->stream));
    return SLAMX_OK;
} SLAMX_CATCH
int slamx_ctx_timer_stop_ms(slamx_ctx* ctx, float* ms) try {
    HIP_TRY(hipEventRecord(ctx->t1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->t1));
    HIP_TRY(hipEventElapsedTime(ms, ctx->t0, ctx->t1));
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_detector_config_load(const char* yaml_path, slamx_detector_config* out) try {
    YamlFile y;
    if (yaml_path == nullptr || !y.open(yaml_path)) {
        return fail(SLAMX_ERR_RUNTIME,
                    std::string("Could not open feature detector file: ") + (yaml_path ? yaml_path : "(null)"));
    }
    out->intensity_threshold = y.get_int("IntensityThreshold");
    out->contiguous_pixels_threshold = y.get_int("ContiguousPixelsThreshold");
    out->non_max_suppression = y.get_int("NonMaxSuppression");
    out->suppression_window_size = y.get_int("SuppressionWindowSize");
    out->patch_size = y.get_int("PatchSize");
    out->num_brief_pairs = y.get_int("NumBRIEFPairs");
    return validate_detector_config(*out);
} SLAMX_CATCH

int slamx_matcher_config_load(const char* yaml_path, slamx_matcher_config* out) try {
    YamlFile y;
    if (yaml_path == nullptr || !y.open(yaml_path)) {
        return fail(SLAMX_ERR_RUNTIME,
                    std::string("Could not open feature matcher config file: ") + (yaml_path ? yaml_path : "(null)"));
    }
    const std::string dt = y.get_string("DistanceType");
    if (dt == "HAMMING") {
        out->distance_type = 0;
    } else if (dt == "L2") {
        out->distance_type = 1;
    } else {
        return fail(SLAMX_ERR_RUNTIME, "Invalid distance type. Must be 'HAMMING' or 'L2'.");
    }
    out->filter_matches = y.get_int("FilterMatches");
    out->good_matches_count = y.get_int("GoodMatchesCount");
    out->use_ratio_test = y.get_int("UseRatioTest");
    out->ratio_test_threshold = static_cast<float>(y.get_double("RatioTestThreshold"));
    return SLAMX_OK;  // range checks happen in slamx_matcher_create, as in the constructor
} SLAMX_CATCH

int slamx_loop_closure_config_load(const char* yaml_path, slamx_loop_closure_config* out) try {
    YamlFile y;
    if (yaml_path == nullptr || out == nullptr || !y.open(yaml_path)) {
        return fail(SLAMX_ERR_RUNTIME,
                    std::string("Could not open Loop Closure config file: ") + (yaml_path ? yaml_path : "(null)"));
    }
    out->min_db_size = y.get_int("MinDbSize");
    if (out->min_db_size < 0) {
        return fail(SLAMX_ERR_RUNTIME, "'MinDbSize' must be a non-negative integer.");
    }
    out->min_frames_difference = y.get_int("MinFramesDifference");
    if (out->min_frames_difference <= 0) {
        return fail(SLAMX_ERR_RUNTIME, "'MinFramesDifference' must be a positive integer.");
    }
    out->min_absolute_score = y.get_double("MinAbsoluteScore");
    if (out->min_absolute_score < 0.0) {
        return fail(SLAMX_ERR_RUNTIME, "'MinAbsoluteScore' must be non-negative.");
    }
    out->relative_score_factor = y.get_double("RelativeScoreFactor");
    if (out->relative_score_factor < 0.0) {
        return fail(SLAMX_ERR_RUNTIME, "'RelativeScoreFactor' must be non-negative.");
    }
    out->min_matches_for_pnp = y.get_int("MinMatchesForPnP");
    if (out->min_matches_for_pnp <= 3) {
        return fail(SLAMX_ERR_RUNTIME, "'MinMatchesForPnP' must be greater than 3 for PnP.");
    }
    out->min_inliers_for_pnp = y.get_int("MinInliersForPnP");
    if (out->min_inliers_for_pnp <= 3) {
        return fail(SLAMX_ERR_RUNTIME, "'MinInliersForPnP' must be greater than 3 for PnP.");
    }
    if (out->min_inliers_for_pnp > out->min_matches_for_pnp) {
        return fail(SLAMX_ERR_RUNTIME, "'MinInliersForPnP' cannot be greater than 'MinMatchesForPnP'.");
    }
    out->ransac_max_iterations = y.get_int("RansacMaxIterations");
    if (out->ransac_max_iterations <= 0) {
        return fail(SLAMX_ERR_RUNTIME, "'RansacMaxIterations' must be a positive integer.");
    }
    out->ransac_reprojection_threshold = y.get_double("RansacReprojectionThreshold");
    if (out->ransac_reprojection_threshold <= 0.0) {
        return fail(SLAMX_ERR_RUNTIME, "'RansacReprojectionThreshold' must be a positive value.");
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_detector_create(slamx_ctx* ctx, const slamx_detector_config* cfg, slamx_detector** out) try {
    if (ctx == nullptr || cfg == nullptr || out == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "null argument");
    }
    *out = nullptr;
    SLAMX_TRY(validate_detector_config(*cfg));
    HIP_TRY(hipSetDevice(ctx->device));
    auto* d = new slamx_detector();
    d->ctx = ctx;
    d->cfg = *cfg;
    d->pattern = make_pattern(cfg->patch_size, cfg->num_brief_pairs);
    std::vector<int8_t> p8(d->pattern.size() + 4, 0);
    for (size_t i = 0; i < d->pattern.size(); ++i) {
        p8[i] = static_cast<int8_t>(d->pattern[i]);  // |coord| < patch/2 <= 127
    }
    double w25[25];
    make_blur_weights(w25);
    int rc = dev_alloc(&d->pattern_dev, p8.size());
    if (rc == SLAMX_OK) {
        rc = dev_alloc(&d->weights_dev, 25);
    }
    if (rc != SLAMX_OK) {
        slamx_detector_destroy(d);
        return rc;
    }
    hipError_t e = hipMemcpy(d->pattern_dev, p8.data(), p8.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        e = hipMemcpy(d->weights_dev, w25, sizeof(w25), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && cfg->patch_size / 2 <= 15) {
        std::vector<uint32_t> mw(2 * MOMENT_TABLE_ENTRIES);
        moment_weight_table(cfg->patch_size / 2, mw.data());
        rc = dev_alloc(&d->moment_weights_dev, MOMENT_TABLE_ENTRIES);
        if (rc != SLAMX_OK) {
            slamx_detector_destroy(d);
            return rc;
        }
        e = hipMemcpy(d->moment_weights_dev, mw.data(), mw.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        slamx_detector_destroy(d);  // the host object and its device buffers must not outlive a failed create
        return fail(SLAMX_ERR_HIP, std::string("detector constants upload: ") + hipGetErrorString(e));
    }
    *out = d;
    return SLAMX_OK;
} SLAMX_CATCH

void slamx_detector_destroy(slamx_detector* det) {
    if (det == nullptr) {
        return;
    }
    if (det->single != nullptr) {
        slamx_batch_destroy(det->single);
    }
    dev_free(det->pattern_dev);
    dev_free(det->moment_weights_dev);
    dev_free(det->weights_dev);
    if (det->stage != nullptr) {
        (void)hipHostFree(det->stage);
    }
    delete det;
}

int slamx_detector_pattern(const slamx_detector* det, int32_t* xy4, int cap_pairs, int* n_pairs) try {
    const int n = static_cast<int>(det->pattern.size() / 4);
    if (n_pairs != nullptr) {
        *n_pairs = n;
    }
    if (xy4 != nullptr) {
        const int m = std::min(n, cap_pairs);
        std::memcpy(xy4, det->pattern.data(), static_cast<size_t>(m) * 4 * sizeof(int32_t));
    }
    return n <= cap_pairs || xy4 == nullptr ? SLAMX_OK : fail(SLAMX_ERR_CAPACITY, "pattern buffer too small");
} SLAMX_CATCH

int slamx_detector_descriptor_width(const slamx_detector* det) { return det->cfg.num_brief_pairs / 8; }
slamx_ctx* slamx_detector_context(const slamx_detector* det) { return det != nullptr ? det->ctx : nullptr; }

// ------------------------------------------------------------------ batches
int slamx_batch_create(slamx_detector* det, const slamx_batch_shape* shape, slamx_batch** out) try {
    if (det == nullptr || shape == nullptr || out == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "null argument");
    }
    *out = nullptr;
    if (shape->rows <= 0 || shape->cols <= 0 || shape->max_frames <= 0 || shape->rows > 65535 ||
        shape->cols > 65535) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "batch shape invalid");
    }
    HIP_TRY(hipSetDevice(det->ctx->device));
    auto* b = new slamx_batch();
    b->det = det;
    b->shape = *shape;
    const long long px = static_cast<long long>(shape->rows) * shape->cols;
    long long maxM = shape->max_candidates > 0 ? shape->max_candidates : std::max(1024LL, px / 16);
    maxM = std::min<long long>(maxM, MAX_CANDIDATES_HARD);
    long long maxK = shape->max_keypoints > 0 ? shape->max_keypoints : maxM;
    b->shape.max_candidates = static_cast<int>(maxM);
    b->shape.max_keypoints = static_cast<int>(maxK);
    BatchView& v = b->v;
    const size_t F = static_cast<size_t>(shape->max_frames);
    v.rows = shape->rows;
    v.cols = shape->cols;
    v.pitch = (shape->cols + 63) / 64 * 64;
    v.frame_stride = static_cast<size_t>(v.pitch) * v.rows;
    v.nw = (shape->cols + 63) / 64;
    v.maxM = static_cast<int>(maxM);
    v.maxK = static_cast<int>(maxK);
    v.maxSeg = static_cast<int>(maxM / 16 + 16);
    v.width = det->cfg.num_brief_pairs / 8;
    const size_t nwords = static_cast<size_t>(v.rows) * v.nw;
    int rc = SLAMX_OK;
    uint8_t* blurred = nullptr;
    auto A = [&](auto** p, size_t n) {
        if (rc == SLAMX_OK) {
            rc = dev_alloc(p, n);
        }
    };
    A(&b->frames, F * v.frame_stride + 64);
    A(&blurred, F * v.frame_stride + 64);
    A(&v.masks, F * nwords);
    A(&v.wordoff, F * nwords);
    A(&v.wordcnt, F * nwords);
    v.fast_tiles = ((v.cols + FAST_TILE_W - 1) / FAST_TILE_W) * ((v.rows + FAST_TILE_H - 1) / FAST_TILE_H);
    A(&v.tile_count, F * v.fast_tiles);
    A(&v.tile_list, F * v.fast_tiles * CORNER_LIST_CAP);
    A(&v.n_cand, F);
    A(&v.cand_xy, F * v.maxM);
    A(&v.keys, F * v.maxM);
    A(&v.sorted, F * v.maxM);
    A(&v.rank_of, F * v.maxM);
    A(&v.state, F * v.maxM);
    v.segstart_stride = ((static_cast<size_t>(v.maxM) + 32) / 32 + 1) * 4;
    A(&v.segstart, F * v.segstart_stride);
    A(&v.seglist, F * 3 * v.maxSeg);
    A(&v.kps, F * v.maxK);
    A(&v.n_kp, F);
    A(&v.desc, F * v.maxK * v.width);
    A(&v.visit, F * v.maxK);
    v.nbands = (v.rows + 63) / 64;
    {
        const size_t regions = static_cast<size_t>((v.cols + 239) / 240) * v.nbands;
        if (regions <= 4096) {  // (describe.hip: RT_MAX_REGIONS; larger frames keep the per-keypoint tiles)
            A(&v.region_first, F * (regions + 1));
            A(&v.region_list, F * v.maxK);
        }
    }
    A(&v.moments, F * v.maxK);
    A(&v.cos_sin, F * v.maxK);
    A(&v.status, F);
    A(&v.nbr_cnt, F * v.maxM);
    A(&v.nbr, F * v.maxM * NBR_CAP);
    v.frames = b->frames;
    v.blurred = blurred;
    if (rc != SLAMX_OK) {
        slamx_batch_destroy(b);
        return rc;
    }
    hipError_t e = hipMemsetAsync(v.n_kp, 0, F * sizeof(int), det->ctx->stream);
    if (e == hipSuccess) {
        e = hipMemsetAsync(v.n_cand, 0, F * sizeof(int), det->ctx->stream);
    }
    if (e == hipSuccess) {
        e = hipMemsetAsync(v.status, 0, F * sizeof(int), det->ctx->stream);
    }
    if (e == hipSuccess) {
        // pitch padding is read (never used) by the tile loaders: keep it defined
        e = hipMemsetAsync(b->frames, 0, F * v.frame_stride + 64, det->ctx->stream);
    }
    if (e != hipSuccess) {
        slamx_batch_destroy(b);
        return fail(SLAMX_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
    }
    *out = b;
    return SLAMX_OK;
} SLAMX_CATCH

void slamx_batch_destroy(slamx_batch* b) {
    if (b == nullptr) {
        return;
    }
    if (b->det != nullptr) {
        (void)hipSetDevice(b->det->ctx->device);
        (void)hipStreamSynchronize(b->det->ctx->stream);
    }
    BatchView& v = b->v;
    dev_free(b->frames);
    dev_free(v.blurred);
    dev_free(v.masks);
    dev_free(v.wordoff);
    dev_free(v.wordcnt);
    dev_free(v.tile_count);
    dev_free(v.tile_list);
    dev_free(v.n_cand);
    dev_free(v.cand_xy);
    dev_free(v.keys);
    dev_free(v.sorted);
    dev_free(v.rank_of);
    dev_free(v.state);
    dev_free(v.segstart);
    dev_free(v.seglist);
    dev_free(v.kps);
    dev_free(v.n_kp);
    dev_free(v.desc);
    dev_free(v.visit);
    dev_free(v.region_first);
    dev_free(v.region_list);
    dev_free(v.moments);
    dev_free(v.cos_sin);
    dev_free(v.status);
    dev_free(b->pack);
    dev_free(v.nbr_cnt);
    dev_free(v.nbr);
    dev_free(b->matches);
    dev_free(b->n_matches);
    dev_free(b->best);
    dev_free(b->second);
    dev_free(b->xdesc);
    for (auto& e : b->ev) {
        if (e != nullptr) {
            (void)hipEventDestroy(e);
        }
    }
    delete b;
}

uint8_t* slamx_batch_frames(slamx_batch* b, size_t* pitch, size_t* frame_stride) {
    if (pitch != nullptr) {
        *pitch = static_cast<size_t>(b->v.pitch);
    }
    if (frame_stride != nullptr) {
        *frame_stride = b->v.frame_stride;
    }
    return b->frames;
}

int slamx_batch_upload(slamx_batch* b, int first, int n, const uint8_t* host, ptrdiff_t stride,
                       ptrdiff_t frame_stride) try {
    if (first < 0 || n < 0 || first + n > b->shape.max_frames || host == nullptr || stride < b->v.cols) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "upload range/stride invalid");
    }
    for (int f = 0; f < n; ++f) {
        HIP_TRY(hipMemcpy2DAsync(b->frames + static_cast<size_t>(first + f) * b->v.frame_stride,
                                 static_cast<size_t>(b->v.pitch), host + static_cast<ptrdiff_t>(f) * frame_stride,
                                 static_cast<size_t>(stride), static_cast<size_t>(b->v.cols),
                                 static_cast<size_t>(b->v.rows), hipMemcpyHostToDevice, b->det->ctx->stream));
    }
    // pageable host memory: the runtime is still reading `host` when the async call returns
    // (measured: a buffer overwritten right after the call corrupts 99 of 100 uploads)
    HIP_TRY(hipStreamSynchronize(b->det->ctx->stream));
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_host_alloc(size_t bytes, void** out) try {
    if (out == nullptr || bytes == 0) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "host_alloc: out is null or size is 0");
    }
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return SLAMX_OK;
} SLAMX_CATCH

void slamx_host_free(void* p) {
    if (p != nullptr) {
        (void)hipHostFree(p);
    }
}

int slamx_batch_upload_async(slamx_batch* b, int first, int n, const uint8_t* host, ptrdiff_t stride,
                             ptrdiff_t frame_stride) {
    return slamx_batch_upload_async_flags(b, first, n, host, stride, frame_stride, 0);
}

int slamx_batch_upload_async_flags(slamx_batch* b, int first, int n, const uint8_t* host, ptrdiff_t stride,
                                   ptrdiff_t frame_stride, int flags) try {
    if (first < 0 || n < 0 || first + n > b->shape.max_frames || host == nullptr || stride < b->v.cols) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "upload range/stride invalid");
    }
    if (n == 0) {
        return SLAMX_OK;
    }
    HIP_TRY(hipSetDevice(b->det->ctx->device));
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, host) != hipSuccess || attr.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "upload_async needs page-locked host memory (slamx_host_alloc)");
    }
    // The copy goes on a stream of its own (SLAMX_UPLOAD_STREAM=0: on the compute stream): ordered behind the work already
    // queued on the context (the frames it overwrites may still be read) and in front of whatever is queued next.  A queue
    // that only ever carries copies keeps them on the DMA engines beside other contexts' kernels; on the compute stream the
    // same copy was seen to run serially with those kernels once the stream's history had been copies only
    // (tools/e2e_breakdown.py: 10.5 against 15.0 ms per 256-frame c3 batch for the same calls, depending on what ran before).
    slamx_ctx* c = b->det->ctx;
    static const bool own_stream = std::getenv("SLAMX_UPLOAD_STREAM") == nullptr || std::atoi(std::getenv("SLAMX_UPLOAD_STREAM")) != 0;
    hipStream_t s = c->stream;
    if (own_stream) {
        if (c->copy == nullptr) {
            HIP_TRY(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&c->e_copy_before, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->e_copy_after, hipEventDisableTiming));
        }
        if ((flags & SLAMX_UPLOAD_SLOTS_IDLE) == 0) {
            HIP_TRY(hipEventRecord(c->e_copy_before, c->stream));
            HIP_TRY(hipStreamWaitEvent(c->copy, c->e_copy_before, 0));
        }  // (else: nothing queued earlier touches these slots, the caller says: the copy may run beside that work)
        s = c->copy;
    }
    uint8_t* dst = b->frames + static_cast<size_t>(first) * b->v.frame_stride;
    const bool dense = stride == b->v.pitch && frame_stride == static_cast<ptrdiff_t>(b->v.frame_stride);
    if (dense) {  // rows and frames back to back on both sides: one copy
        HIP_TRY(hipMemcpyAsync(dst, host, static_cast<size_t>(n) * b->v.frame_stride, hipMemcpyHostToDevice, s));
    } else if (frame_stride == stride * static_cast<ptrdiff_t>(b->v.rows) && b->v.frame_stride == static_cast<size_t>(b->v.pitch) * b->v.rows) {
        // the host's frames follow each other row after row (a frame width that is no multiple of 64: 1392-pixel KITTI
        // rows against a pitch of 1408): both sides are ONE array of n * rows rows, so one 2-D copy moves them all (round 5:
        // until then a 2-D copy per frame -- 250 of them for a KITTI sequence, 20 us each)
        HIP_TRY(hipMemcpy2DAsync(dst, static_cast<size_t>(b->v.pitch), host, static_cast<size_t>(stride), static_cast<size_t>(b->v.cols),
                                 static_cast<size_t>(b->v.rows) * static_cast<size_t>(n), hipMemcpyHostToDevice, s));
    } else {
        for (int f = 0; f < n; ++f) {
            HIP_TRY(hipMemcpy2DAsync(dst + static_cast<size_t>(f) * b->v.frame_stride, static_cast<size_t>(b->v.pitch),
                                     host + static_cast<ptrdiff_t>(f) * frame_stride, static_cast<size_t>(stride),
                                     static_cast<size_t>(b->v.cols), static_cast<size_t>(b->v.rows), hipMemcpyHostToDevice, s));
        }
    }
    if (own_stream) {
        HIP_TRY(hipEventRecord(c->e_copy_after, c->copy));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->e_copy_after, 0));
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_batch_detect_and_compute(slamx_batch* b, int nframes) try {
    if (nframes < 0 || nframes > b->shape.max_frames) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "nframes out of range");
    }
    if (nframes == 0) {
        return SLAMX_OK;
    }
    HIP_TRY(hipSetDevice(b->det->ctx->device));
    begin_set(b);
    if (blur_beside_sort(b, nframes)) {
        // The blur needs the frames only, so it is queued on the context's second stream right before the sort and
        // joined before the describe kernel.  Taken when the sort is ONE round of workgroups (a frame per CU, each
        // for the whole launch: 110 KB of LDS and 64 registers leave room for two blur workgroups beside it):
        // 7.87 -> 7.52 ms per 256 frames of 80k candidates.  With more frames than CUs the blur's workgroups take
        // the LDS the next round of sort workgroups waits for and nothing is gained.
        slamx_ctx* c = b->det->ctx;
        const DetectorParams p = params_of(b->det);
        enqueue_detect_until_sort(b, nframes);
        (void)hipEventRecord(c->e_fork, c->stream);
        (void)hipStreamWaitEvent(c->side, c->e_fork, 0);
        launch_blur(b->v, b->det->weights_dev, nframes, c->side);
        (void)hipEventRecord(c->e_join, c->side);
        launch_sort(b->v, nframes, c->stream, !nms_uses_greedy_path(b->v, p));
        launch_nms(b->v, p, nframes, c->stream, nullptr);
        (void)hipStreamWaitEvent(c->stream, c->e_join, 0);
        launch_describe(b->v, p, b->det->pattern_dev, nframes, true, c->stream, nullptr, nullptr, b->dense_hint != 0);
    } else {
        enqueue_detect(b, nframes);
        enqueue_compute(b, nframes, true);
    }
    b->last_frames = nframes;
    HIP_TRY(hipGetLastError());
    return SLAMX_OK;
} SLAMX_CATCH

namespace {
// The view of frames [first, ...) of a batch: every per-frame array moved on by `first` frames (strides as
// slamx_batch_create lays them out), so that a kernel launched for `count` frames works on slots first .. first + count - 1.
BatchView shifted_view(const BatchView& v, int first) {
    BatchView s = v;
    const size_t f = static_cast<size_t>(first);
    const size_t nwords = static_cast<size_t>(v.rows) * v.nw;
    auto move = [&](auto*& p, size_t per_frame) {
        if (p != nullptr) {
            p += f * per_frame;
        }
    };
    move(s.frames, v.frame_stride);
    move(s.blurred, v.frame_stride);
    move(s.masks, nwords);
    move(s.wordoff, nwords);
    move(s.wordcnt, nwords);
    move(s.tile_count, static_cast<size_t>(v.fast_tiles));
    move(s.tile_list, static_cast<size_t>(v.fast_tiles) * CORNER_LIST_CAP);
    move(s.n_cand, 1);
    move(s.cand_xy, static_cast<size_t>(v.maxM));
    move(s.keys, static_cast<size_t>(v.maxM));
    move(s.sorted, static_cast<size_t>(v.maxM));
    move(s.rank_of, static_cast<size_t>(v.maxM));
    move(s.state, static_cast<size_t>(v.maxM));
    move(s.nbr_cnt, static_cast<size_t>(v.maxM));
    move(s.nbr, static_cast<size_t>(v.maxM) * NBR_CAP);
    move(s.segstart, v.segstart_stride);
    move(s.seglist, static_cast<size_t>(3) * v.maxSeg);
    move(s.kps, static_cast<size_t>(v.maxK));
    move(s.n_kp, 1);
    move(s.desc, static_cast<size_t>(v.maxK) * v.width);
    move(s.visit, static_cast<size_t>(v.maxK));
    move(s.region_first, static_cast<size_t>((v.cols + 239) / 240) * v.nbands + 1);
    move(s.region_list, static_cast<size_t>(v.maxK));
    move(s.moments, static_cast<size_t>(v.maxK));
    move(s.cos_sin, static_cast<size_t>(v.maxK));
    move(s.status, 1);
    return s;
}
}  // namespace

int slamx_batch_detect_and_compute_range(slamx_batch* b, int first, int count) try {
    if (b == nullptr || first < 0 || count < 0 || first + count > b->shape.max_frames) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "detect_and_compute_range: first / count out of range");
    }
    if (count == 0) {
        return SLAMX_OK;
    }
    if (first == 0) {
        return slamx_batch_detect_and_compute(b, count);
    }
    // the same launches on the view of the slots from `first` on (kernel arguments are taken at launch)
    const BatchView whole = b->v;
    const int frames_before = b->last_frames;
    b->v = shifted_view(whole, first);
    const int rc = slamx_batch_detect_and_compute(b, count);
    b->v = whole;
    b->last_frames = std::max(frames_before, first + count);
    return rc;
} SLAMX_CATCH

int slamx_batch_match_consecutive(slamx_batch* b, slamx_matcher* m, int nframes, int use_keypoints,
                                  int first_train_frame) {
    return slamx_batch_match_consecutive_range(b, m, 0, nframes, use_keypoints, first_train_frame);
}

int slamx_batch_match_consecutive_range(slamx_batch* b, slamx_matcher* m, int first, int nframes, int use_keypoints,
                                        int first_train_frame) try {
    if (b == nullptr || m == nullptr || first < 0 || nframes < 0 || first + nframes > b->shape.max_frames ||
        first_train_frame >= b->shape.max_frames) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "nframes / first_train_frame out of range");
    }
    if (m->cfg.distance_type != 0) {
        return fail(SLAMX_ERR_RUNTIME, "L2 distance requires float descriptors. Use the float overload.");
    }
    HIP_TRY(hipSetDevice(b->det->ctx->device));
    const size_t F = static_cast<size_t>(b->shape.max_frames);
    SLAMX_TRY(batch_match_arrays(b, b->det->ctx->stream));
    const int pattern_len = static_cast<int>(b->det->pattern.size() / 4);
    // Two all-pairs scans with identical results (match.hip): int8 matrix cores in spatial order, and xor + popcount.
    // Which is faster depends on the size of the train set and on how far its keypoints spread (slamx_internal.h:
    // MATCH_MFMA_MIN_KEYPOINTS); both are only known on the device, so when the batch's capacity allows BOTH kernels
    // are launched and every frame pair is taken by one of them (MatchView::mfma_flag, set by scan_choice_kernel).
    const bool use_mfma = pattern_len <= 62 && b->v.width % 4 == 0 && b->v.maxK >= match_mfma_min_keypoints_wide();
    if (use_mfma && b->xdesc == nullptr) {
        SLAMX_TRY(dev_alloc(&b->xdesc, match_scratch_bytes(F, b->v.maxK)));
    }
    MatchView mv{};
    mv.desc = b->v.desc;
    mv.kps = use_keypoints != 0 ? b->v.kps : nullptr;
    mv.n_kp = b->v.n_kp;
    mv.matches = b->matches;
    mv.n_matches = b->n_matches;
    mv.best = b->best;
    mv.second = b->second;
    mv.maxK = b->v.maxK;
    mv.width = b->v.width;
    // descriptors of a batch come from describe_kernel: bits >= pattern_len are never set
    mv.live_words = std::max(1, (pattern_len + 31) / 32);
    mv.live_bits = std::max(1, pattern_len);
    match_scratch_carve(use_mfma ? b->xdesc : nullptr, F, b->v.maxK, &mv);
    mv.mfma_min_train = match_mfma_min_keypoints();
    mv.mfma_min_wide = match_mfma_min_keypoints_wide();
    // (the matrix-core path takes 64 queries per workgroup, the popcount path 512)
    mv.nsplit = match_split_for(nframes, b->v.maxK);  // one split for both scans: they share the per-chunk result slots
    mv.use_ratio = m->cfg.use_ratio_test;
    mv.ratio = m->cfg.ratio_test_threshold;
    if (!recorded(b, 10)) {
        record(b, 10);
    }
    hipEvent_t mid = event_slot(b, 11);
    if (first > 0) {
        // frames first .. first + nframes - 1, each against its predecessor (frame first - 1 was described by an earlier call)
        launch_match_consecutive(mv, first, nframes, -1, b->det->ctx->stream, mid);
    } else if (first_train_frame >= 0) {
        launch_match_consecutive(mv, 0, nframes, first_train_frame, b->det->ctx->stream, mid);
    } else {
        launch_match_consecutive(mv, 1, nframes - 1, -1, b->det->ctx->stream, mid);
        HIP_TRY(hipMemsetAsync(b->n_matches, 0, sizeof(int), b->det->ctx->stream));
    }
    record(b, 12);
    HIP_TRY(hipGetLastError());
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_batch_counts(slamx_batch* b, int nframes, int32_t* n_candidates, int32_t* n_keypoints, int32_t* n_matches) try {
    if (nframes < 0 || nframes > b->shape.max_frames) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "nframes out of range");
    }
    hipStream_t s = b->det->ctx->stream;
    std::vector<int> status(static_cast<size_t>(nframes));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(status.data(), b->v.status, sizeof(int) * nframes, hipMemcpyDeviceToHost));
    if (n_candidates != nullptr) {
        HIP_TRY(hipMemcpy(n_candidates, b->v.n_cand, sizeof(int) * nframes, hipMemcpyDeviceToHost));
    }
    if (n_keypoints != nullptr) {
        HIP_TRY(hipMemcpy(n_keypoints, b->v.n_kp, sizeof(int) * nframes, hipMemcpyDeviceToHost));
        const int tile_min = describe_tile_min_keypoints(b->v);
        int dense = 0;
        for (int f = 0; f < nframes; ++f) {
            dense |= n_keypoints[f] >= tile_min ? 1 : 0;
        }
        b->dense_hint = dense;
    }
    if (n_matches != nullptr) {
        if (b->n_matches != nullptr) {
            HIP_TRY(hipMemcpy(n_matches, b->n_matches, sizeof(int) * nframes, hipMemcpyDeviceToHost));
        } else {
            std::memset(n_matches, 0, sizeof(int) * nframes);
        }
    }
    for (int f = 0; f < nframes; ++f) {
        SLAMX_TRY(status_to_error(status[static_cast<size_t>(f)], f));
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_batch_download(slamx_batch* b, int frame, slamx_keypoint* kps, uint8_t* desc, int cap, int* n) try {
    if (frame < 0 || frame >= b->shape.max_frames) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "frame out of range");
    }
    HIP_TRY(hipStreamSynchronize(b->det->ctx->stream));
    int st = 0;
    int cnt = 0;
    HIP_TRY(hipMemcpy(&st, b->v.status + frame, sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&cnt, b->v.n_kp + frame, sizeof(int), hipMemcpyDeviceToHost));
    b->last_status = st;
    SLAMX_TRY(status_to_error(st, frame));
    if (n != nullptr) {
        *n = cnt;
    }
    if (cnt > cap) {
        return fail(SLAMX_ERR_CAPACITY, "keypoint buffer too small");
    }
    if (kps != nullptr && cnt > 0) {
        HIP_TRY(hipMemcpy(kps, b->v.kps + static_cast<size_t>(frame) * b->v.maxK, sizeof(slamx_keypoint) * cnt,
                          hipMemcpyDeviceToHost));
    }
    if (desc != nullptr && cnt > 0) {
        HIP_TRY(hipMemcpy(desc, b->v.desc + static_cast<size_t>(frame) * b->v.maxK * b->v.width,
                          static_cast<size_t>(cnt) * b->v.width, hipMemcpyDeviceToHost));
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_batch_download_results_async(slamx_batch* b, int nframes, slamx_keypoint* kps, uint8_t* desc, slamx_match* matches,
                                       int32_t* n_keypoints, int32_t* n_matches) try {
    if (b == nullptr || nframes < 0 || nframes > b->shape.max_frames) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "nframes out of range");
    }
    if (nframes == 0) {
        return SLAMX_OK;
    }
    HIP_TRY(hipSetDevice(b->det->ctx->device));
    hipStream_t s = b->det->ctx->stream;
    const size_t n = static_cast<size_t>(nframes);
    const void* dsts[5] = {kps, desc, matches, n_keypoints, n_matches};
    for (const void* p : dsts) {
        if (p == nullptr) {
            continue;
        }
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, p) != hipSuccess || attr.type != hipMemoryTypeHost) {
            (void)hipGetLastError();
            return fail(SLAMX_ERR_INVALID_ARGUMENT, "download_results_async needs page-locked host memory (slamx_host_alloc)");
        }
    }
    if ((matches != nullptr || n_matches != nullptr) && b->matches == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "no matches computed on this batch");
    }
    // whole capacity-padded arrays, one copy each: the counts say how much of every frame's slice is valid
    if (kps != nullptr) {
        HIP_TRY(hipMemcpyAsync(kps, b->v.kps, n * b->v.maxK * sizeof(slamx_keypoint), hipMemcpyDeviceToHost, s));
    }
    if (desc != nullptr) {
        HIP_TRY(hipMemcpyAsync(desc, b->v.desc, n * b->v.maxK * b->v.width, hipMemcpyDeviceToHost, s));
    }
    if (matches != nullptr) {
        HIP_TRY(hipMemcpyAsync(matches, b->matches, n * b->v.maxK * sizeof(slamx_match), hipMemcpyDeviceToHost, s));
    }
    if (n_keypoints != nullptr) {
        HIP_TRY(hipMemcpyAsync(n_keypoints, b->v.n_kp, n * sizeof(int), hipMemcpyDeviceToHost, s));
    }
    if (n_matches != nullptr) {
        HIP_TRY(hipMemcpyAsync(n_matches, b->n_matches, n * sizeof(int), hipMemcpyDeviceToHost, s));
    }
    return SLAMX_OK;
} SLAMX_CATCH

size_t slamx_batch_packed_bound(slamx_batch* b, int nframes) {
    if (b == nullptr || nframes < 0 || nframes > b->shape.max_frames) {
        return 0;
    }
    return packed_bound_bytes(nframes, b->v.maxK, b->v.width, true);
}

int slamx_batch_download_packed_async(slamx_batch* b, int nframes, void* host, size_t bytes) try {
    if (b == nullptr || nframes <= 0 || nframes > b->shape.max_frames || host == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "download_packed_async: nframes out of range or no host buffer");
    }
    HIP_TRY(hipSetDevice(b->det->ctx->device));
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, host) != hipSuccess || attr.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "download_packed_async needs page-locked host memory (slamx_host_alloc)");
    }
    const size_t bound = packed_bound_bytes(nframes, b->v.maxK, b->v.width, true);
    if (bytes < sizeof(slamx_packed_header)) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "download_packed_async: the host buffer cannot even hold the header");
    }
    if (b->pack == nullptr) {
        const size_t cap = packed_bound_bytes(b->shape.max_frames, b->v.maxK, b->v.width, true);
        SLAMX_TRY(dev_alloc(&b->pack, cap));
        b->pack_bytes = cap;
    }
    hipStream_t s = b->det->ctx->stream;
    // Page-locked memory the device can address (slamx_host_alloc's always is): the pack kernel stores the results
    // there itself, beside its HBM copy -- no copy-engine transfer, which takes its turn with the uploads of the
    // batches behind.  SLAMX_PACKED_DOWNLOAD=copy keeps the pack + hipMemcpyAsync form.
    const char* mode = std::getenv("SLAMX_PACKED_DOWNLOAD");
    void* mapped = nullptr;
    if ((mode == nullptr || std::strcmp(mode, "copy") != 0) &&
        (hipHostGetDevicePointer(&mapped, host, 0) != hipSuccess || mapped == nullptr)) {
        (void)hipGetLastError();
        mapped = nullptr;
    }
    if (mapped != nullptr) {
        launch_pack_results(b->v, b->matches, b->n_matches, nframes, b->pack, s, static_cast<uint8_t*>(mapped),
                            std::min(bytes, bound));
        HIP_TRY(hipGetLastError());
        return SLAMX_OK;
    }
    launch_pack_results(b->v, b->matches, b->n_matches, nframes, b->pack, s);
    HIP_TRY(hipGetLastError());
    // (on the compute stream: a queue of its own for this copy, as the uploads have, was measured and is far worse --
    // 21.6 against 12.3 ms per 256-frame c3 batch end to end)
    HIP_TRY(hipMemcpyAsync(host, b->pack, std::min(bytes, bound), hipMemcpyDeviceToHost, s));
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_batch_download_packed_tail(slamx_batch* b, void* host, size_t from, size_t host_capacity) try {
    if (b == nullptr || host == nullptr || b->pack == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "download_packed_tail: nothing has been packed on this batch");
    }
    HIP_TRY(hipSetDevice(b->det->ctx->device));
    HIP_TRY(hipStreamSynchronize(b->det->ctx->stream));
    slamx_packed_header h{};
    HIP_TRY(hipMemcpy(&h, b->pack, sizeof(h), hipMemcpyDeviceToHost));
    const size_t total = static_cast<size_t>(h.total_bytes);
    if (total > b->pack_bytes || total > host_capacity) {
        return fail(SLAMX_ERR_CAPACITY, "download_packed_tail: the host buffer is smaller than header.total_bytes");
    }
    from &= ~static_cast<size_t>(15);  // (the kernel-written form leaves a cut dword out: start at the section grain)
    if (from < total) {
        HIP_TRY(hipMemcpy(static_cast<uint8_t*>(host) + from, b->pack + from, total - from, hipMemcpyDeviceToHost));
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_batch_download_image(slamx_batch* b, int frame, int blurred, uint8_t* out, ptrdiff_t stride) try {
    if (frame < 0 || frame >= b->shape.max_frames || out == nullptr || stride < b->v.cols) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "frame out of range or output buffer invalid");
    }
    HIP_TRY(hipStreamSynchronize(b->det->ctx->stream));
    const uint8_t* src = (blurred != 0 ? b->v.blurred : b->v.frames) + static_cast<size_t>(frame) * b->v.frame_stride;
    HIP_TRY(hipMemcpy2D(out, static_cast<size_t>(stride), src, static_cast<size_t>(b->v.pitch),
                        static_cast<size_t>(b->v.cols), static_cast<size_t>(b->v.rows), hipMemcpyDeviceToHost));
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_batch_download_matches(slamx_batch* b, slamx_matcher* m, int frame, int apply_filter, slamx_match* out,
                                 int cap, int* n) try {
    if (frame < 0 || frame >= b->shape.max_frames || b->matches == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "frame out of range or no matches computed");
    }
    HIP_TRY(hipStreamSynchronize(b->det->ctx->stream));
    int cnt = 0;
    HIP_TRY(hipMemcpy(&cnt, b->n_matches + frame, sizeof(int), hipMemcpyDeviceToHost));
    std::vector<slamx_match> v(static_cast<size_t>(cnt));
    if (cnt > 0) {
        HIP_TRY(hipMemcpy(v.data(), b->matches + static_cast<size_t>(frame) * b->v.maxK, sizeof(slamx_match) * cnt,
                          hipMemcpyDeviceToHost));
    }
    if (apply_filter != 0 && m != nullptr && m->cfg.filter_matches != 0) {
        filter_and_sort(v, m->cfg.good_matches_count);
    }
    if (n != nullptr) {
        *n = static_cast<int>(v.size());
    }
    if (static_cast<int>(v.size()) > cap) {
        return fail(SLAMX_ERR_CAPACITY, "match buffer too small");
    }
    if (out != nullptr && !v.empty()) {
        std::memcpy(out, v.data(), sizeof(slamx_match) * v.size());
    }
    return SLAMX_OK;
} SLAMX_CATCH

// filterAndSortMatches (feature_matcher.cpp:191-204) on a caller's raw match list, in place, on the host: the step
// slamx_batch_download_matches applies after its copy, for callers that already hold the raw lists (the packed download)
int slamx_matcher_filter_matches(const slamx_matcher* m, slamx_match* matches, int n, int* n_out) try {
    if (m == nullptr || n < 0 || (n > 0 && matches == nullptr) || n_out == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "filter_matches: arguments invalid");
    }
    *n_out = n;
    if (m->cfg.filter_matches != 0 && n > 0) {
        std::vector<slamx_match> v(matches, matches + n);
        filter_and_sort(v, m->cfg.good_matches_count);
        std::memcpy(matches, v.data(), sizeof(slamx_match) * v.size());
        *n_out = static_cast<int>(v.size());
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_batch_set_profiling(slamx_batch* b, int n_sets) try {
    return slamx_batch_set_profiling_every(b, n_sets, 1);
} SLAMX_CATCH

int slamx_batch_set_profiling_every(slamx_batch* b, int n_sets, int every) try {
    if (every < 1) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "profiling stride must be at least 1");
    }
    if (n_sets < 0 || n_sets > 4096) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "n_sets out of range");
    }
    HIP_TRY(hipSetDevice(b->det->ctx->device));
    HIP_TRY(hipStreamSynchronize(b->det->ctx->stream));
    for (auto& e : b->ev) {
        if (e != nullptr) {
            (void)hipEventDestroy(e);
        }
    }
    b->ev.assign(static_cast<size_t>(n_sets) * slamx_batch::NEV, nullptr);
    b->ev_valid.assign(static_cast<size_t>(n_sets) * slamx_batch::NEV, 0);
    b->n_sets = n_sets;
    b->calls = 0;
    b->cur_set = 0;
    b->every = every;
    b->active = false;
    for (auto& e : b->ev) {
        HIP_TRY(hipEventCreate(&e));
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_batch_stage_ms(slamx_batch* b, int set_index, float* ms) try {
    if (set_index < 0 || set_index >= b->n_sets) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "profiling set out of range");
    }
    HIP_TRY(hipStreamSynchronize(b->det->ctx->stream));
    const size_t base = static_cast<size_t>(set_index) * slamx_batch::NEV;
    for (int s = 0; s < SLAMX_STAGE_COUNT; ++s) {
        ms[s] = 0.0f;
        if (b->ev_valid[base + s] && b->ev_valid[base + s + 1]) {
            float t = 0.0f;
            if (hipEventElapsedTime(&t, b->ev[base + s], b->ev[base + s + 1]) == hipSuccess) {
                ms[s] = t;
            }
        }
    }
    return SLAMX_OK;
} SLAMX_CATCH

// ------------------------------------------------------------------ single-frame host API
int slamx_detect(slamx_detector* det, const uint8_t* img, int rows, int cols, ptrdiff_t stride, slamx_keypoint* kps,
                 int cap, int* n) try {
    SLAMX_TRY(check_image(img, rows, cols, stride));
    for (int attempt = 0;; ++attempt) {
        SLAMX_TRY(ensure_single(det, rows, cols, 0, attempt > 0));
        slamx_batch* b = det->single;
        HIP_TRY(hipSetDevice(det->ctx->device));
        SingleStage st;
        SLAMX_TRY(single_stage(det, &st));
        SLAMX_TRY(single_upload(det, st, img, stride));
        HIP_TRY(hipMemsetAsync(b->v.status, 0, sizeof(int), det->ctx->stream));
        begin_set(b);
        enqueue_detect(b, 1);
        HIP_TRY(hipGetLastError());
        const int rc = single_download(det, st, kps, nullptr, cap, n);
        if (attempt == 0 && single_can_grow(det, rc, rows, cols)) {
            continue;
        }
        return rc;
    }
} SLAMX_CATCH

int slamx_compute(slamx_detector* det, const uint8_t* img, int rows, int cols, ptrdiff_t stride, slamx_keypoint* kps,
                  int n, uint8_t* desc) try {
    if (n < 0) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "negative keypoint count");
    }
    if (n == 0) {
        return SLAMX_OK;  // compute() on an empty vector yields a 0x0 matrix (feature_detector.cpp:22-25)
    }
    SLAMX_TRY(check_image(img, rows, cols, stride));
    if (kps == nullptr || desc == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "null keypoints/descriptors");
    }
    SLAMX_TRY(ensure_single(det, rows, cols, n));
    slamx_batch* b = det->single;
    hipStream_t s = det->ctx->stream;
    HIP_TRY(hipSetDevice(det->ctx->device));
    SingleStage st;
    SLAMX_TRY(single_stage(det, &st));
    SLAMX_TRY(single_upload(det, st, img, stride));
    std::memcpy(st.kps_in, kps, sizeof(slamx_keypoint) * static_cast<size_t>(n));
    *st.n_in = n;
    HIP_TRY(hipMemcpyAsync(b->v.kps, st.kps_in, sizeof(slamx_keypoint) * static_cast<size_t>(n), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->v.n_kp, st.n_in, sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(b->v.status, 0, sizeof(int), s));
    begin_set(b);
    enqueue_compute(b, 1, false);
    HIP_TRY(hipGetLastError());
    int cnt = 0;
    return single_download(det, st, kps, desc, n, &cnt);
} SLAMX_CATCH

int slamx_detect_and_compute(slamx_detector* det, const uint8_t* img, int rows, int cols, ptrdiff_t stride,
                             slamx_keypoint* kps, uint8_t* desc, int cap, int* n) try {
    SLAMX_TRY(check_image(img, rows, cols, stride));
    for (int attempt = 0;; ++attempt) {
        SLAMX_TRY(ensure_single(det, rows, cols, 0, attempt > 0));
        slamx_batch* b = det->single;
        HIP_TRY(hipSetDevice(det->ctx->device));
        SingleStage st;
        SLAMX_TRY(single_stage(det, &st));
        SLAMX_TRY(single_upload(det, st, img, stride));
        HIP_TRY(hipMemsetAsync(b->v.status, 0, sizeof(int), det->ctx->stream));
        SLAMX_TRY(slamx_batch_detect_and_compute(b, 1));
        const int rc = single_download(det, st, kps, desc, cap, n);
        if (attempt == 0 && single_can_grow(det, rc, rows, cols)) {
            continue;
        }
        return rc;
    }
} SLAMX_CATCH

int slamx_fast_candidates(slamx_detector* det, const uint8_t* img, int rows, int cols, ptrdiff_t stride,
                          slamx_keypoint* kps, int cap, int* n) try {
    SLAMX_TRY(check_image(img, rows, cols, stride));
    hipStream_t s = det->ctx->stream;
    const DetectorParams p = params_of(det);
    slamx_batch* b = nullptr;
    int st = 0, m = 0;
    for (int attempt = 0;; ++attempt) {
        SLAMX_TRY(ensure_single(det, rows, cols, 0, attempt > 0));
        b = det->single;
        SLAMX_TRY(slamx_batch_upload(b, 0, 1, img, stride, 0));
        HIP_TRY(hipMemsetAsync(b->v.status, 0, sizeof(int), s));
        launch_fast_detect(b->v, p, 1, s);
        launch_scan(b->v, 1, s);
        launch_expand(b->v, p, 1, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(&st, b->v.status, sizeof(int), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&m, b->v.n_cand, sizeof(int), hipMemcpyDeviceToHost));
        b->last_status = st;
        if (attempt == 0 && single_can_grow(det, status_to_error(st, 0), rows, cols)) {
            continue;
        }
        break;
    }
    SLAMX_TRY(status_to_error(st, 0));
    if (n != nullptr) {
        *n = m;
    }
    if (m > cap) {
        return fail(SLAMX_ERR_CAPACITY, "candidate buffer too small");
    }
    std::vector<uint32_t> xy(static_cast<size_t>(m)), keys(static_cast<size_t>(m));
    if (m > 0) {
        HIP_TRY(hipMemcpy(xy.data(), b->v.cand_xy, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(keys.data(), b->v.keys, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < m; ++i) {
        kps[i].x = static_cast<float>(xy[static_cast<size_t>(i)] & 0xffffu);
        kps[i].y = static_cast<float>(xy[static_cast<size_t>(i)] >> 16);
        kps[i].size = 6.0f;
        kps[i].angle = 0.0f;
        kps[i].response = static_cast<float>(keys[static_cast<size_t>(i)] >> SCORE_SHIFT);
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_gaussian_blur(slamx_detector* det, const uint8_t* img, int rows, int cols, ptrdiff_t stride, uint8_t* out) try {
    SLAMX_TRY(check_image(img, rows, cols, stride));
    SLAMX_TRY(ensure_single(det, rows, cols, 0));
    slamx_batch* b = det->single;
    hipStream_t s = det->ctx->stream;
    SLAMX_TRY(slamx_batch_upload(b, 0, 1, img, stride, 0));
    launch_blur(b->v, det->weights_dev, 1, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy2DAsync(out, static_cast<size_t>(cols), b->v.blurred, static_cast<size_t>(b->v.pitch),
                             static_cast<size_t>(cols), static_cast<size_t>(rows), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_debug_sort_desc(slamx_ctx* ctx, const uint32_t* scores, int n, uint32_t* perm_out) {
    return slamx_debug_sort_desc_groups(ctx, scores, n, 1, perm_out);
}

int slamx_debug_sort_desc_groups(slamx_ctx* ctx, const uint32_t* scores, int n, int groups, uint32_t* perm_out) try {
    if (n < 0 || n > MAX_CANDIDATES_HARD || groups < 1 || groups > 64) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "n / groups out of range");
    }
    if (n == 0) {
        return SLAMX_OK;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<uint32_t> keys(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
        if (scores[i] > 4095u) {
            return fail(SLAMX_ERR_INVALID_ARGUMENT, "score above 4095");
        }
        keys[static_cast<size_t>(i)] = (scores[i] << SCORE_SHIFT) | static_cast<uint32_t>(i);
    }
    uint32_t *dk = nullptr, *ds = nullptr;
    uint8_t* dseg = nullptr;
    int2* dlist = nullptr;
    int* dstatus = nullptr;
    int* dn = nullptr;
    uint32_t* drank = nullptr;
    const int maxSeg = n / 16 + 16;
    int rc = dev_alloc(&dk, n);
    if (rc == SLAMX_OK && groups > 1) rc = dev_alloc(&dn, 1);
    if (rc == SLAMX_OK && groups > 1) rc = dev_alloc(&drank, n);
    if (rc == SLAMX_OK) rc = dev_alloc(&ds, n);
    if (rc == SLAMX_OK) rc = dev_alloc(&dseg, (static_cast<size_t>(n) + 32) / 32 * 4 + 8);
    if (rc == SLAMX_OK) rc = dev_alloc(&dlist, static_cast<size_t>(3) * maxSeg);
    if (rc == SLAMX_OK) rc = dev_alloc(&dstatus, 1);
    int st = 0;
    hipError_t e = hipSuccess;
    if (rc == SLAMX_OK) {
        e = hipMemcpyAsync(dk, keys.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && groups > 1) {
            e = hipMemcpyAsync(dn, &n, sizeof(int), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(dstatus, 0, sizeof(int), ctx->stream);
        }
        if (e == hipSuccess) {
            if (groups > 1) {
                launch_debug_sort_split(dk, ds, dseg, dlist, maxSeg, n, dstatus, dn, drank, groups, ctx->stream);
            } else {
                launch_debug_sort(dk, ds, dseg, dlist, maxSeg, n, dstatus, ctx->stream);
            }
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(keys.data(), ds, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&st, dstatus, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    dev_free(dk);
    dev_free(ds);
    dev_free(dseg);
    dev_free(dlist);
    dev_free(dstatus);
    dev_free(dn);
    dev_free(drank);
    if (rc != SLAMX_OK) {
        return rc;
    }
    if (e != hipSuccess) {
        return fail(SLAMX_ERR_HIP, std::string("debug sort: ") + hipGetErrorString(e));
    }
    if (st != 0) {
        return fail(SLAMX_ERR_CAPACITY, "sort segment list overflow");
    }
    for (int i = 0; i < n; ++i) {
        perm_out[i] = keys[static_cast<size_t>(i)] & INDEX_MASK;
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_debug_trig(slamx_ctx* ctx, const float* m01, const float* m10, int n, float* angle_deg, float* cos_out,
                     float* sin_out) try {
    if (n <= 0) {
        return SLAMX_OK;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    float* d[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = SLAMX_OK;
    for (auto& p : d) {
        if (rc == SLAMX_OK) rc = dev_alloc(&p, n);
    }
    hipError_t e = hipSuccess;
    if (rc == SLAMX_OK) {
        const size_t bytes = sizeof(float) * static_cast<size_t>(n);
        e = hipMemcpyAsync(d[0], m01, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d[1], m10, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            launch_debug_trig(d[0], d[1], n, d[2], d[3], d[4], ctx->stream);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(angle_deg, d[2], bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(cos_out, d[3], bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sin_out, d[4], bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    for (auto& p : d) {
        dev_free(p);
    }
    if (rc != SLAMX_OK) {
        return rc;
    }
    if (e != hipSuccess) {
        return fail(SLAMX_ERR_HIP, std::string("debug trig: ") + hipGetErrorString(e));
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_debug_penalty(slamx_ctx* ctx, const int32_t* dist, const float* qx, const float* qy, const float* tx,
                        const float* ty, int n, int32_t* out) try {
    if (n <= 0) {
        return SLAMX_OK;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    void* d[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const void* src[5] = {dist, qx, qy, tx, ty};
    const size_t bytes = 4 * static_cast<size_t>(n);
    hipError_t e = hipSuccess;
    for (auto& p : d) {
        if (e == hipSuccess) e = hipMalloc(&p, bytes);
    }
    for (int k = 0; k < 5 && e == hipSuccess; ++k) {
        e = hipMemcpyAsync(d[k], src[k], bytes, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) {
        launch_debug_penalty(static_cast<int*>(d[0]), static_cast<float*>(d[1]), static_cast<float*>(d[2]),
                             static_cast<float*>(d[3]), static_cast<float*>(d[4]), n, static_cast<int*>(d[5]), ctx->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d[5], bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    for (auto& p : d) {
        if (p != nullptr) (void)hipFree(p);
    }
    if (e != hipSuccess) {
        return fail(SLAMX_ERR_HIP, std::string("debug penalty: ") + hipGetErrorString(e));
    }
    return SLAMX_OK;
} SLAMX_CATCH

// ------------------------------------------------------------------ matcher
int slamx_matcher_create(slamx_ctx* ctx, const slamx_matcher_config* cfg, slamx_matcher** out) try {
    if (ctx == nullptr || cfg == nullptr || out == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "null argument");
    }
    *out = nullptr;
    // messages follow src/frontend/feature_matcher.cpp:27-57
    if (cfg->distance_type != 0 && cfg->distance_type != 1) {
        return fail(SLAMX_ERR_RUNTIME, "Invalid distance type. Must be 'HAMMING' or 'L2'.");
    }
    if (cfg->filter_matches != 0 && cfg->filter_matches != 1) {
        return fail(SLAMX_ERR_RUNTIME, "FilterMatches must be either 0 (false) or 1 (true).");
    }
    if (cfg->filter_matches != 0 && cfg->good_matches_count <= 0) {
        return fail(SLAMX_ERR_RUNTIME, "GoodMatchesCount must be positive when filtering is enabled.");
    }
    if (cfg->use_ratio_test != 0 && cfg->use_ratio_test != 1) {
        return fail(SLAMX_ERR_RUNTIME, "UseRatioTest must be either 0 (false) or 1 (true).");
    }
    if (cfg->ratio_test_threshold < 0.0F || cfg->ratio_test_threshold > 1.0F) {
        return fail(SLAMX_ERR_RUNTIME, "RatioTestThreshold must be in the range [0, 1].");
    }
    auto* m = new slamx_matcher();
    m->ctx = ctx;
    m->cfg = *cfg;
    *out = m;
    return SLAMX_OK;
} SLAMX_CATCH

void slamx_matcher_destroy(slamx_matcher* m) {
    if (m == nullptr) {
        return;
    }
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    dev_free(m->in_blob);   // desc, kps, n_kp
    dev_free(m->out_blob);  // matches, n_matches
    dev_free(m->best);
    dev_free(m->second);
    dev_free(m->xdesc);
    if (m->stage != nullptr) {
        (void)hipHostFree(m->stage);
    }
    delete m;
}

int slamx_matcher_match(slamx_matcher* m, const uint8_t* d1, int n1, int width1, const uint8_t* d2, int n2, int width2,
                const slamx_keypoint* kp1, int nkp1, const slamx_keypoint* kp2, int nkp2, slamx_match* out, int cap,
                int* n) try {
    if (n != nullptr) {
        *n = 0;
    }
    // validateInputs, src/frontend/feature_matcher.cpp:97-111
    if (n1 <= 0 || n2 <= 0) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "Empty descriptors provided.");
    }
    if (m->cfg.distance_type != 0) {
        return fail(SLAMX_ERR_RUNTIME, "DescriptorMatrix (uint8_t) requires HAMMING distance.");
    }
    if (width1 != width2) {
        return fail(SLAMX_ERR_RUNTIME, "Descriptor dimensions must match.");
    }
    if (width1 <= 0 || d1 == nullptr || d2 == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "descriptor pointer/width invalid");
    }
    const bool use_kp = nkp1 > 0 && nkp2 > 0;  // :150
    if (use_kp && (nkp1 < n1 || nkp2 < n2 || kp1 == nullptr || kp2 == nullptr)) {
        // the reference indexes keypoints1[i] / keypoints2[j] unchecked (UB when shorter)
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "keypoint vectors shorter than the descriptor sets");
    }
    HIP_TRY(hipSetDevice(m->ctx->device));
    hipStream_t s = m->ctx->stream;
    const int need = std::max(n1, n2);
    if (need > m->cap || width1 != m->width) {
        HIP_TRY(hipStreamSynchronize(s));
        dev_free(m->in_blob);
        dev_free(m->out_blob);
        dev_free(m->best);
        dev_free(m->second);
        dev_free(m->xdesc);
        if (m->stage != nullptr) {
            (void)hipHostFree(m->stage);
            m->stage = nullptr;
        }
        m->cap = 0;
        const size_t c = static_cast<size_t>(need);
        // in:  desc [2][c][width] (padded to 16 bytes) | kps [2][c] | n_kp [2]        out: matches [2][c] | n_matches [2]
        const size_t desc_bytes = (2 * c * width1 + 64 + 15) / 16 * 16;
        m->in_bytes = desc_bytes + 2 * c * sizeof(slamx_keypoint) + 16;
        m->out_tail_bytes = c * sizeof(slamx_match) + 16;  // the query slot's matches and both counts: what a call brings back
        SLAMX_TRY(dev_alloc(&m->in_blob, m->in_bytes));
        SLAMX_TRY(dev_alloc(&m->out_blob, 2 * c * sizeof(slamx_match) + 16));
        m->desc = m->in_blob;
        m->kps = reinterpret_cast<slamx_keypoint*>(m->in_blob + desc_bytes);
        m->n_kp = reinterpret_cast<int*>(m->in_blob + desc_bytes + 2 * c * sizeof(slamx_keypoint));
        m->matches = reinterpret_cast<slamx_match*>(m->out_blob);
        m->n_matches = reinterpret_cast<int*>(m->out_blob + 2 * c * sizeof(slamx_match));
        SLAMX_TRY(dev_alloc(&m->best, 2 * MATCH_MAX_SPLIT * c));
        SLAMX_TRY(dev_alloc(&m->second, 2 * MATCH_MAX_SPLIT * c));
        SLAMX_TRY(dev_alloc(&m->xdesc, match_scratch_bytes(2, need)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->stage), m->in_bytes + m->out_tail_bytes, hipHostMallocDefault));
        std::memset(m->stage, 0, m->in_bytes + m->out_tail_bytes);
        m->cap = need;
        m->width = width1;
    }
    const size_t c = static_cast<size_t>(m->cap);
    {
        // the inputs into the page-locked image of the device block (frame 0 = train, frame 1 = query), then ONE copy
        const size_t desc_bytes = (2 * c * width1 + 64 + 15) / 16 * 16;
        uint8_t* st = m->stage;
        std::memcpy(st, d2, static_cast<size_t>(n2) * width1);
        std::memcpy(st + c * width1, d1, static_cast<size_t>(n1) * width1);
        if (use_kp) {
            std::memcpy(st + desc_bytes, kp2, sizeof(slamx_keypoint) * n2);
            std::memcpy(st + desc_bytes + c * sizeof(slamx_keypoint), kp1, sizeof(slamx_keypoint) * n1);
        }
        const int counts[2] = {n2, n1};
        std::memcpy(st + desc_bytes + 2 * c * sizeof(slamx_keypoint), counts, sizeof(counts));
        HIP_TRY(hipMemcpyAsync(m->in_blob, st, m->in_bytes, hipMemcpyHostToDevice, s));
    }
    MatchView mv{};
    mv.desc = m->desc;
    mv.kps = use_kp ? m->kps : nullptr;
    mv.n_kp = m->n_kp;
    mv.matches = m->matches;
    mv.n_matches = m->n_matches;
    mv.best = m->best;
    mv.second = m->second;
    mv.maxK = m->cap;
    mv.width = width1;
    mv.nsplit = match_split_for(1, m->cap);
    mv.live_words = 0;
    mv.live_bits = 0;
    mv.xdesc = nullptr;
    if (width1 % 4 == 0) {
        // highest 32-bit word that is non-zero in any descriptor of either set
        int live = 1;
        for (int s2 = 0; s2 < 2; ++s2) {
            const uint8_t* d = s2 == 0 ? d1 : d2;
            const int cnt = s2 == 0 ? n1 : n2;
            const int words = width1 / 4;
            for (int i = 0; i < cnt; ++i) {
                const uint32_t* row = reinterpret_cast<const uint32_t*>(d + static_cast<size_t>(i) * width1);
                for (int w = words - 1; w >= live; --w) {
                    uint32_t val;
                    std::memcpy(&val, row + w, 4);
                    if (val != 0) {
                        live = w + 1;
                        break;
                    }
                }
            }
        }
        mv.live_words = live;
        if (live <= 2) {
            // at most 64 bits in use: the exact position of the highest one decides whether the
            // matrix-core path (<= 62 live bits) applies
            uint32_t top = 0;
            for (int s2 = 0; s2 < 2; ++s2) {
                const uint8_t* d = s2 == 0 ? d1 : d2;
                const int cnt = s2 == 0 ? n1 : n2;
                for (int i = 0; i < cnt; ++i) {
                    uint32_t val;
                    std::memcpy(&val, d + static_cast<size_t>(i) * width1 + 4 * (live - 1), 4);
                    top |= val;
                }
            }
            int hb = 0;
            while (top != 0) {
                ++hb;
                top >>= 1;
            }
            mv.live_bits = std::max(1, 32 * (live - 1) + hb);
            if (n2 >= match_mfma_min_keypoints_wide()) {  // see slamx_batch_match_consecutive: the device decides
                match_scratch_carve(m->xdesc, 2, m->cap, &mv);
                mv.mfma_min_train = match_mfma_min_keypoints();
                mv.mfma_min_wide = match_mfma_min_keypoints_wide();
                mv.nsplit = match_split_for(1, m->cap * 2);
            }
        }
    }
    mv.use_ratio = m->cfg.use_ratio_test;
    mv.ratio = m->cfg.ratio_test_threshold;
    launch_match_consecutive(mv, 1, 1, -1, s, nullptr);
    HIP_TRY(hipGetLastError());
    // the query slot's matches (at most n1) and the counts behind them: ONE copy into the page-locked block, one wait
    uint8_t* back = m->stage + m->in_bytes;
    HIP_TRY(hipMemcpyAsync(back, m->matches + c, m->out_tail_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    int cnt = 0;
    std::memcpy(&cnt, back + c * sizeof(slamx_match) + sizeof(int), sizeof(int));  // n_matches[1]
    cnt = std::min(std::max(cnt, 0), n1);
    std::vector<slamx_match> v(static_cast<size_t>(cnt));
    if (cnt > 0) {
        std::memcpy(v.data(), back, sizeof(slamx_match) * cnt);
    }
    if (m->cfg.filter_matches != 0) {
        filter_and_sort(v, m->cfg.good_matches_count);
    }
    if (n != nullptr) {
        *n = static_cast<int>(v.size());
    }
    if (static_cast<int>(v.size()) > cap) {
        return fail(SLAMX_ERR_CAPACITY, "match buffer too small");
    }
    if (out != nullptr && !v.empty()) {
        std::memcpy(out, v.data(), sizeof(slamx_match) * v.size());
    }
    return SLAMX_OK;
} SLAMX_CATCH

// (the pose solves -- slamx_pnp_* / slamx_essential_ransac / slamx_recover_pose / slamx_triangulate / slamx_pose_* --
// live in pose_api.hip)
// ------------------------------------------------------------------ image pyramid (pyramid.hip)
}  // extern "C"

struct slamx_pyramid {
    slamx_detector* det = nullptr;
    std::vector<slamx_batch*> levels;
    std::vector<double> scale;         // scale_factor^l
    std::vector<uint32_t*> xtab, ytab;  // per level >= 1: source index | weight << 16 per destination column / row
};

namespace {

// centre-aligned source position of destination index d: ((2d + 1) Ws - Wd) / (2 Wd), as index + 11-bit weight
std::vector<uint32_t> scale_table(int dst, int src) {
    std::vector<uint32_t> t(static_cast<size_t>(dst));
    for (int d = 0; d < dst; ++d) {
        const long long num = (2LL * d + 1) * src - dst;
        const long long den = 2LL * dst;
        long long i0 = 0, f = 0;
        if (num > 0) {
            i0 = num / den;
            f = ((num % den) * 2048) / den;
        }
        if (i0 >= src - 1) {
            i0 = src - 1;
            f = 0;
        }
        t[static_cast<size_t>(d)] = static_cast<uint32_t>(i0) | (static_cast<uint32_t>(f) << 16);
    }
    return t;
}

}  // namespace

extern "C" {

int slamx_pyramid_create(slamx_detector* det, const slamx_batch_shape* shape, int n_levels, double scale_factor,
                         slamx_pyramid** out) try {
    if (det == nullptr || shape == nullptr || out == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "null argument");
    }
    *out = nullptr;
    if (n_levels < 1 || n_levels > SLAMX_PYRAMID_MAX_LEVELS || !(scale_factor > 1.0) || !(scale_factor <= 4.0)) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pyramid: 1 <= n_levels <= 16 and 1 < scale_factor <= 4");
    }
    auto* p = new slamx_pyramid();
    p->det = det;
    const double px0 = static_cast<double>(shape->rows) * shape->cols;
    for (int l = 0; l < n_levels; ++l) {
        const double sc = std::pow(scale_factor, l);
        slamx_batch_shape sh = *shape;
        sh.rows = static_cast<int>(std::lrint(shape->rows / sc));
        sh.cols = static_cast<int>(std::lrint(shape->cols / sc));
        if (sh.rows < 8 || sh.cols < 8) {
            slamx_pyramid_destroy(p);
            return fail(SLAMX_ERR_INVALID_ARGUMENT, "pyramid: level " + std::to_string(l) + " is smaller than 8 pixels");
        }
        if (l > 0) {
            // Downscaling moves structures closer together and averages the noise away: raw corners per
            // PIXEL go up level by level, their number per level hardly falls (measured on the bench
            // frames: 14k, 20k, 23k, 24k, 24k, 24k, 22k, 19k).  The candidate capacity is therefore the
            // caller's on every level (bounded by the level's pixels); kept keypoints are bounded by the
            // suppression radius, i.e. by the area: they get the linear scale, with a floor.
            const double r = std::sqrt((static_cast<double>(sh.rows) * sh.cols) / px0);
            if (shape->max_candidates > 0) {
                sh.max_candidates = static_cast<int>(std::min<long long>(shape->max_candidates, static_cast<long long>(sh.rows) * sh.cols));
                sh.max_candidates = std::max(sh.max_candidates, 1024);
            }
            if (shape->max_keypoints > 0) {
                sh.max_keypoints = std::max(256, static_cast<int>(shape->max_keypoints * r) + 64);
            }
        }
        slamx_batch* b = nullptr;
        const int rc = slamx_batch_create(det, &sh, &b);
        if (rc != SLAMX_OK) {
            slamx_pyramid_destroy(p);
            return rc;
        }
        p->levels.push_back(b);
        p->scale.push_back(sc);
        p->xtab.push_back(nullptr);
        p->ytab.push_back(nullptr);
        if (l > 0) {
            const slamx_batch* prev = p->levels[static_cast<size_t>(l) - 1];
            std::vector<uint32_t> xt = scale_table(sh.cols, prev->shape.cols);
            xt.resize((xt.size() + 3) / 4 * 4 + 4, xt.back());  // downscale_kernel reads four entries at a time
            const std::vector<uint32_t> yt = scale_table(sh.rows, prev->shape.rows);
            int rc2 = dev_alloc(&p->xtab[static_cast<size_t>(l)], xt.size());
            if (rc2 == SLAMX_OK) {
                rc2 = dev_alloc(&p->ytab[static_cast<size_t>(l)], yt.size());
            }
            hipError_t e = hipSuccess;
            if (rc2 == SLAMX_OK) {
                e = hipMemcpy(p->xtab[static_cast<size_t>(l)], xt.data(), xt.size() * 4, hipMemcpyHostToDevice);
                if (e == hipSuccess) {
                    e = hipMemcpy(p->ytab[static_cast<size_t>(l)], yt.data(), yt.size() * 4, hipMemcpyHostToDevice);
                }
            }
            if (rc2 != SLAMX_OK || e != hipSuccess) {
                slamx_pyramid_destroy(p);
                return rc2 != SLAMX_OK ? rc2 : fail(SLAMX_ERR_HIP, std::string("pyramid tables: ") + hipGetErrorString(e));
            }
        }
    }
    *out = p;
    return SLAMX_OK;
} SLAMX_CATCH

void slamx_pyramid_destroy(slamx_pyramid* p) {
    if (p == nullptr) {
        return;
    }
    for (slamx_batch* b : p->levels) {
        slamx_batch_destroy(b);
    }
    for (auto& t : p->xtab) {
        dev_free(t);
    }
    for (auto& t : p->ytab) {
        dev_free(t);
    }
    delete p;
}

slamx_batch* slamx_pyramid_batch(slamx_pyramid* p, int level) {
    if (p == nullptr || level < 0 || level >= static_cast<int>(p->levels.size())) {
        return nullptr;
    }
    return p->levels[static_cast<size_t>(level)];
}

int slamx_pyramid_level_shape(const slamx_pyramid* p, int level, int* rows, int* cols, double* scale) try {
    if (p == nullptr || level < 0 || level >= static_cast<int>(p->levels.size())) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pyramid: level out of range");
    }
    const slamx_batch* b = p->levels[static_cast<size_t>(level)];
    if (rows != nullptr) {
        *rows = b->shape.rows;
    }
    if (cols != nullptr) {
        *cols = b->shape.cols;
    }
    if (scale != nullptr) {
        *scale = p->scale[static_cast<size_t>(level)];
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_pyramid_detect_and_compute(slamx_pyramid* p, int nframes) try {
    if (p == nullptr || nframes < 0 || nframes > p->levels[0]->shape.max_frames) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "pyramid: nframes out of range");
    }
    if (nframes == 0) {
        return SLAMX_OK;
    }
    HIP_TRY(hipSetDevice(p->det->ctx->device));
    hipStream_t s = p->det->ctx->stream;
    const size_t nl = p->levels.size();
    for (size_t l = 1; l < nl; ++l) {
        launch_downscale(p->levels[l - 1]->v, p->levels[l]->v, p->xtab[l], p->ytab[l], nframes, s);
    }
    for (size_t l = 0; l < nl; ++l) {
        SLAMX_TRY(slamx_batch_detect_and_compute(p->levels[l], nframes));
    }
    for (size_t l = 1; l < nl; ++l) {
        launch_merge_level(p->levels[l]->v, p->levels[0]->v, static_cast<float>(p->scale[l]), nframes, s);
    }
    if (nl > 1) {
        // Level 0's stage events: the boundary between BRIEF and MATCH moves behind the other levels and the merges, so
        // that a match call that follows is timed as the matcher alone (until round 3 its bracket started at the end of
        // level 0's describe kernel and held levels 1.. as well: the "11.4 ms matcher" of the pyramid workload).  The
        // BRIEF stage of a pyramid's level 0 therefore reads: its BRIEF kernel + all of levels 1.. + the merges.
        record(p->levels[0], 10);
    }
    HIP_TRY(hipGetLastError());
    return SLAMX_OK;
} SLAMX_CATCH

// ------------------------------------------------------------------ sliding-window Gauss-Newton (gn.hip)
}  // extern "C"

struct slamx_gn {
    slamx_ctx* ctx = nullptr;
    int n_windows = 0, n_cams = 0, max_landmarks = 0, nslices = 1;
    double* poses[2] = {nullptr, nullptr};  // [cur] = linearisation point, the other receives the step
    int cur = 0;
    double* landmarks = nullptr;
    float* obs = nullptr;
    int* n_landmarks = nullptr;
    double* partial = nullptr;
    double* delta = nullptr;
    double* cost = nullptr;  // [SLAMX_GN_MAX_ITERATIONS + 1][n_windows]
    int* status = nullptr;
    int last_iterations = -1;
    // profiling: event pairs around every kernel of the iterations since it was switched on
    // (stage 0 accumulate, 1 solve, 2 back-substitution), on the context stream
    bool profiling = false;
    int prof_every = 1;       // only every prof_every-th iteration records (an event between two kernels keeps the
    long long prof_iter = 0;  // second from starting while the first drains)
    bool prof_now = false;
    std::vector<hipEvent_t> ev[3];
    size_t ev_used[3] = {0, 0, 0};
};

namespace {

// record the next unused event of a stage (grown on demand); pairs are (2k, 2k + 1)
void gn_mark(slamx_gn* g, int stage) {
    if (!g->profiling || !g->prof_now) {
        return;
    }
    if (g->ev_used[stage] == g->ev[stage].size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) {
            return;
        }
        g->ev[stage].push_back(e);
    }
    (void)hipEventRecord(g->ev[stage][g->ev_used[stage]++], g->ctx->stream);
}

GnLaunch gn_launch_of(const slamx_gn* g, const double* K, double lambda, int n_fixed, int cost_slot) {
    GnLaunch a{};
    a.poses = g->poses[g->cur];
    a.landmarks = g->landmarks;
    a.obs = g->obs;
    a.n_landmarks = g->n_landmarks;
    a.partial = g->partial;
    a.delta = g->delta;
    a.cost = g->cost + static_cast<size_t>(cost_slot) * g->n_windows;
    a.status = g->status;
    a.n_windows = g->n_windows;
    a.n_cams = g->n_cams;
    a.max_landmarks = g->max_landmarks;
    a.nslices = g->nslices;
    a.n_fixed = n_fixed;
    for (int k = 0; k < 4; ++k) {
        a.K[k] = K[k];
    }
    a.lambda = lambda;
    return a;
}

int gn_check_step_args(const slamx_gn* g, const double* K, double lambda, int n_fixed) {
    if (g == nullptr || K == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "gn: null argument");
    }
    if (!(lambda >= 0.0) || n_fixed < 0 || n_fixed > g->n_cams) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "gn: lambda must be >= 0 and 0 <= n_fixed <= n_cams");
    }
    if (!(K[0] > 0.0) || !(K[1] > 0.0)) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "gn: focal lengths must be positive");
    }
    return SLAMX_OK;
}

}  // namespace

extern "C" {

int slamx_gn_create(slamx_ctx* ctx, int n_windows, int n_cams, int max_landmarks, slamx_gn** out) try {
    if (ctx == nullptr || out == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "gn: null argument");
    }
    *out = nullptr;
    if (n_windows <= 0 || n_cams < 1 || n_cams > SLAMX_GN_MAX_CAMS || max_landmarks <= 0) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "gn: need n_windows > 0, 1 <= n_cams <= 10, max_landmarks > 0");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    auto* g = new slamx_gn();
    g->ctx = ctx;
    g->n_windows = n_windows;
    g->n_cams = n_cams;
    g->max_landmarks = max_landmarks;
    // waves per window: all of them resident at once (1024 SIMDs x 2 waves of 250 registers) or a whole
    // number of such rounds; never more than there are chunks
    const int chunks = (max_landmarks + 3) / 4;
    int slices = 4096 / n_windows;
    slices = std::max(1, std::min(std::min(slices, 32), chunks));
    g->nslices = slices;
    const size_t nw = static_cast<size_t>(n_windows);
    int rc = dev_alloc(&g->poses[0], nw * n_cams * 12);
    if (rc == SLAMX_OK) rc = dev_alloc(&g->poses[1], nw * n_cams * 12);
    if (rc == SLAMX_OK) rc = dev_alloc(&g->landmarks, nw * max_landmarks * 3);
    if (rc == SLAMX_OK) rc = dev_alloc(&g->obs, nw * max_landmarks * n_cams * 2);
    if (rc == SLAMX_OK) rc = dev_alloc(&g->n_landmarks, nw);
    if (rc == SLAMX_OK) rc = dev_alloc(&g->partial, nw * slices * 15 * 256);
    if (rc == SLAMX_OK) rc = dev_alloc(&g->delta, nw * 64);
    if (rc == SLAMX_OK) rc = dev_alloc(&g->cost, nw * (SLAMX_GN_MAX_ITERATIONS + 1));
    if (rc == SLAMX_OK) rc = dev_alloc(&g->status, nw);
    if (rc != SLAMX_OK) {
        slamx_gn_destroy(g);
        return rc;
    }
    *out = g;
    return SLAMX_OK;
} SLAMX_CATCH

void slamx_gn_destroy(slamx_gn* g) {
    if (g == nullptr) {
        return;
    }
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    dev_free(g->poses[0]);
    dev_free(g->poses[1]);
    dev_free(g->landmarks);
    dev_free(g->obs);
    dev_free(g->n_landmarks);
    dev_free(g->partial);
    dev_free(g->delta);
    dev_free(g->cost);
    dev_free(g->status);
    for (auto& v : g->ev) {
        for (hipEvent_t e : v) {
            (void)hipEventDestroy(e);
        }
    }
    delete g;
}

int slamx_gn_set_profiling(slamx_gn* g, int on) try {
    if (g == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "gn: null argument");
    }
    HIP_TRY(hipStreamSynchronize(g->ctx->stream));
    g->profiling = on != 0;
    g->prof_every = on > 1 ? on : 1;  // on = n > 1: every n-th iteration records
    g->prof_iter = 0;
    for (auto& u : g->ev_used) {
        u = 0;
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_gn_stage_ms(slamx_gn* g, float* avg_ms, int32_t* launches) try {
    if (g == nullptr || avg_ms == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "gn: null argument");
    }
    HIP_TRY(hipStreamSynchronize(g->ctx->stream));
    for (int st = 0; st < 3; ++st) {
        double sum = 0.0;
        int n = 0;
        for (size_t k = 0; k + 1 < g->ev_used[st]; k += 2) {
            float t = 0.0f;
            if (hipEventElapsedTime(&t, g->ev[st][k], g->ev[st][k + 1]) == hipSuccess) {
                sum += t;
                ++n;
            }
        }
        avg_ms[st] = n > 0 ? static_cast<float>(sum / n) : 0.0f;
        if (launches != nullptr) {
            launches[st] = n;
        }
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_gn_upload(slamx_gn* g, const double* poses, const double* landmarks, const float* obs,
                    const int32_t* n_landmarks) try {
    if (g == nullptr || poses == nullptr || landmarks == nullptr || obs == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "gn: null argument");
    }
    const size_t nw = static_cast<size_t>(g->n_windows);
    std::vector<int> nl(nw, g->max_landmarks);
    if (n_landmarks != nullptr) {
        for (size_t w = 0; w < nw; ++w) {
            if (n_landmarks[w] < 0 || n_landmarks[w] > g->max_landmarks) {
                return fail(SLAMX_ERR_INVALID_ARGUMENT, "gn: n_landmarks out of range");
            }
            nl[w] = n_landmarks[w];
        }
    }
    HIP_TRY(hipSetDevice(g->ctx->device));
    hipStream_t s = g->ctx->stream;
    g->cur = 0;
    HIP_TRY(hipMemcpyAsync(g->poses[0], poses, nw * g->n_cams * 12 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(g->landmarks, landmarks, nw * g->max_landmarks * 3 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(g->obs, obs, nw * g->max_landmarks * g->n_cams * 2 * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(g->n_landmarks, nl.data(), nw * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(g->status, 0, nw * sizeof(int), s));
    HIP_TRY(hipStreamSynchronize(s));  // pageable sources: the caller may reuse them on return
    g->last_iterations = -1;
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_gn_iterate(slamx_gn* g, const double* K, double lambda, int n_fixed, int iterations) try {
    SLAMX_TRY(gn_check_step_args(g, K, lambda, n_fixed));
    if (iterations < 0 || iterations > SLAMX_GN_MAX_ITERATIONS) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "gn: iterations must be in [0, 64]");
    }
    HIP_TRY(hipSetDevice(g->ctx->device));
    hipStream_t s = g->ctx->stream;
    for (int it = 0; it <= iterations; ++it) {
        const GnLaunch a = gn_launch_of(g, K, lambda, n_fixed, it);
        g->prof_now = g->profiling && (g->prof_iter++ % g->prof_every) == 0;
        gn_mark(g, 0);
        launch_gn_accumulate(a, s);
        gn_mark(g, 0);
        const bool last = it == iterations;  // the final pass only evaluates the cost
        gn_mark(g, 1);
        launch_gn_solve(a, g->poses[g->cur ^ 1], last ? 1 : 0, nullptr, nullptr, s);
        gn_mark(g, 1);
        if (!last) {
            gn_mark(g, 2);
            launch_gn_backsub(a, s);
            gn_mark(g, 2);
            g->cur ^= 1;
        }
    }
    HIP_TRY(hipGetLastError());
    g->last_iterations = iterations;
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_gn_download(slamx_gn* g, double* poses, double* landmarks, double* cost, int32_t* status) try {
    if (g == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "gn: null argument");
    }
    HIP_TRY(hipSetDevice(g->ctx->device));
    hipStream_t s = g->ctx->stream;
    const size_t nw = static_cast<size_t>(g->n_windows);
    HIP_TRY(hipStreamSynchronize(s));
    if (poses != nullptr) {
        HIP_TRY(hipMemcpy(poses, g->poses[g->cur], nw * g->n_cams * 12 * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (landmarks != nullptr) {
        HIP_TRY(hipMemcpy(landmarks, g->landmarks, nw * g->max_landmarks * 3 * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (cost != nullptr) {
        if (g->last_iterations < 0) {
            return fail(SLAMX_ERR_INVALID_ARGUMENT, "gn: no slamx_gn_iterate call to report costs of");
        }
        // device layout [iteration][window] -> caller layout [window][iteration]
        const int ni = g->last_iterations + 1;
        std::vector<double> tmp(nw * ni);
        HIP_TRY(hipMemcpy(tmp.data(), g->cost, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t w = 0; w < nw; ++w) {
            for (int i = 0; i < ni; ++i) {
                cost[w * ni + i] = tmp[static_cast<size_t>(i) * nw + w];
            }
        }
    }
    if (status != nullptr) {
        HIP_TRY(hipMemcpy(status, g->status, nw * sizeof(int), hipMemcpyDeviceToHost));
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_gn_normal_equations(slamx_gn* g, const double* K, double lambda, int n_fixed, double* S, double* gvec,
                              double* delta, double* cost) try {
    SLAMX_TRY(gn_check_step_args(g, K, lambda, n_fixed));
    HIP_TRY(hipSetDevice(g->ctx->device));
    hipStream_t s = g->ctx->stream;
    const size_t nw = static_cast<size_t>(g->n_windows);
    const size_t n = static_cast<size_t>(6) * g->n_cams;
    DevBuf dS, dg;
    HIP_TRY(dS.alloc(nw * n * n * sizeof(double)));
    HIP_TRY(dg.alloc(nw * n * sizeof(double)));
    // slot SLAMX_GN_MAX_ITERATIONS of the cost array is scratch here; the stepped poses go to the spare buffer
    const GnLaunch a = gn_launch_of(g, K, lambda, n_fixed, SLAMX_GN_MAX_ITERATIONS);
    launch_gn_accumulate(a, s);
    launch_gn_solve(a, g->poses[g->cur ^ 1], 0, dS.as<double>(), dg.as<double>(), s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    if (S != nullptr) {
        HIP_TRY(hipMemcpy(S, dS.p, nw * n * n * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (gvec != nullptr) {
        HIP_TRY(hipMemcpy(gvec, dg.p, nw * n * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (delta != nullptr) {
        std::vector<double> d64(nw * 64);
        HIP_TRY(hipMemcpy(d64.data(), g->delta, d64.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t w = 0; w < nw; ++w) {
            std::memcpy(delta + w * n, d64.data() + w * 64, n * sizeof(double));
        }
    }
    if (cost != nullptr) {
        HIP_TRY(hipMemcpy(cost, a.cost, nw * sizeof(double), hipMemcpyDeviceToHost));
    }
    return SLAMX_OK;
} SLAMX_CATCH

// ------------------------------------------------------------------ camera
int slamx_camera_load(const char* yaml_path, int camera_index, slamx_camera* out) try {
    YamlFile y;
    if (yaml_path == nullptr || !y.open(yaml_path)) {
        return fail(SLAMX_ERR_RUNTIME, std::string("Could not open calibration file: ") + (yaml_path ? yaml_path : "(null)"));
    }
    const std::string kKey = "K" + std::to_string(camera_index);
    const std::string dKey = "D" + std::to_string(camera_index);
    const YamlNode* k = y.node(kKey);
    const YamlNode* d = y.node(dKey);
    if (k == nullptr || d == nullptr || k->kind != YamlNode::MATRIX || d->kind != YamlNode::MATRIX ||
        k->mat.data.size() < 9) {
        return fail(SLAMX_ERR_RUNTIME, "Could not find keys " + kKey + " or " + dKey + " in file.");
    }
    std::memset(out, 0, sizeof(*out));
    for (int i = 0; i < 9; ++i) {
        out->K[i] = k->mat.data[static_cast<size_t>(i)];
    }
    out->fx = out->K[0];
    out->fy = out->K[4];
    out->cx = out->K[2];
    out->cy = out->K[5];
    const auto& D = d->mat.data;
    out->k1 = D.size() > 0 ? D[0] : 0.0;
    out->k2 = D.size() > 1 ? D[1] : 0.0;
    out->p1 = D.size() > 2 ? D[2] : 0.0;
    out->p2 = D.size() > 3 ? D[3] : 0.0;
    out->k3 = D.size() > 4 ? D[4] : 0.0;
    const YamlNode* sz = y.node("ImageSize");
    if (sz != nullptr && sz->kind == YamlNode::SEQ && sz->seq.size() >= 2) {
        out->width = static_cast<int32_t>(sz->seq[0]);
        out->height = static_cast<int32_t>(sz->seq[1]);
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_undistort(slamx_ctx* ctx, const slamx_camera* cam, const uint8_t* img, int rows, int cols, ptrdiff_t stride,
                    double* out) try {
    if (cam == nullptr || out == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "null argument");
    }
    if (img == nullptr || rows <= 0 || cols <= 0) {
        return fail(SLAMX_ERR_RUNTIME, "Input image is empty.");  // common.hpp:130-132
    }
    if (rows != cam->height || cols != cam->width) {
        return fail(SLAMX_ERR_RUNTIME, "Input image size does not match camera image size.");  // :133-135
    }
    if (stride < cols) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "stride smaller than the row length");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DevBuf dimg, dout;
    const size_t pitch = (static_cast<size_t>(cols) + 63) / 64 * 64;
    HIP_TRY(dimg.alloc(pitch * rows));
    HIP_TRY(dout.alloc(static_cast<size_t>(rows) * cols * sizeof(double)));
    HIP_TRY(hipMemcpy2DAsync(dimg.p, pitch, img, static_cast<size_t>(stride), static_cast<size_t>(cols),
                             static_cast<size_t>(rows), hipMemcpyHostToDevice, s));
    launch_undistort(dimg.as<uint8_t>(), rows, cols, static_cast<int>(pitch), *cam, dout.as<double>(), s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, dout.p, static_cast<size_t>(rows) * cols * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SLAMX_OK;
} SLAMX_CATCH

// ------------------------------------------------------------------ synthetic frames (host)
int slamx_synth_frame(uint64_t seed, int k, int rows, int cols, int n_rect, int rect_min, int rect_max, uint8_t* out) try {
    if (rows <= 0 || cols <= 0 || out == nullptr || rect_min <= 0 || rect_max < rect_min || k < 0 || n_rect < 0) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "synthetic frame arguments invalid");
    }
    const int CW = cols + 64;
    const int CH = rows + 32;
    std::vector<uint8_t> canvas(static_cast<size_t>(CW) * CH, 96);
    uint64_t s = seed;
    for (int r = 0; r < n_rect; ++r) {
        const int w = rect_min + static_cast<int>(splitmix64(s) % static_cast<uint64_t>(rect_max - rect_min + 1));
        const int h = rect_min + static_cast<int>(splitmix64(s) % static_cast<uint64_t>(rect_max - rect_min + 1));
        const int x0 = static_cast<int>(splitmix64(s) % static_cast<uint64_t>(CW));
        const int y0 = static_cast<int>(splitmix64(s) % static_cast<uint64_t>(CH));
        const int g = static_cast<int>(splitmix64(s) % 194ull);
        const uint8_t level = static_cast<uint8_t>(g < 65 ? 16 + g : 112 + (g - 65));  // {16..80} U {112..240}
        const int x1 = std::min(CW, x0 + w);
        const int y1 = std::min(CH, y0 + h);
        for (int y = y0; y < y1; ++y) {
            std::memset(&canvas[static_cast<size_t>(y) * CW + x0], level, static_cast<size_t>(x1 - x0));
        }
    }
    const int ox = (2 * k) % 64;
    const int oy = k % 32;
    uint64_t ns = seed ^ (0x9E3779B97F4A7C15ull * static_cast<uint64_t>(k + 1));
    for (int y = 0; y < rows; ++y) {
        for (int x = 0; x < cols; ++x) {
            const int base = canvas[static_cast<size_t>(y + oy) * CW + (x + ox)];
            const int noise = static_cast<int>(splitmix64(ns) % 9ull) - 4;
            int val = base + noise;
            val = val < 0 ? 0 : (val > 255 ? 255 : val);
            out[static_cast<size_t>(y) * cols + x] = static_cast<uint8_t>(val);
        }
    }
    return SLAMX_OK;
} SLAMX_CATCH


// ---------------------------------------------------------------- map-point all-gather over RCCL
int slamx_mapgather_plan(const int32_t* counts, int world, int* padded_cap, int* total) try {
    if (counts == nullptr || world < 1) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "mapgather_plan: no counts");
    }
    long long sum = 0;
    int cap = 1;  // an empty rank still sends one (zero) point: RCCL needs a non-empty, equal-sized send buffer
    for (int r = 0; r < world; ++r) {
        if (counts[r] < 0) {
            return fail(SLAMX_ERR_INVALID_ARGUMENT, "mapgather_plan: negative count");
        }
        sum += counts[r];
        cap = std::max(cap, static_cast<int>(counts[r]));
    }
    if (sum > 0x7fffffffLL) {
        return fail(SLAMX_ERR_CAPACITY, "mapgather_plan: more than 2^31 points");
    }
    if (padded_cap != nullptr) {
        *padded_cap = cap;
    }
    if (total != nullptr) {
        *total = static_cast<int>(sum);
    }
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_mapgather_unpack(const double* padded, const int32_t* counts, int world, int padded_cap, double* out,
                           int cap_points) try {
    int cap = 0, total = 0;
    SLAMX_TRY(slamx_mapgather_plan(counts, world, &cap, &total));
    if (padded == nullptr || padded_cap < cap) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "mapgather_unpack: padded payload smaller than the largest count");
    }
    if (total > cap_points || (out == nullptr && total > 0)) {
        return fail(SLAMX_ERR_CAPACITY, "mapgather_unpack: output holds fewer points than were gathered");
    }
    size_t at = 0;
    for (int r = 0; r < world; ++r) {
        const size_t n = static_cast<size_t>(counts[r]) * 3;
        if (n > 0) {
            std::memcpy(out + at, padded + static_cast<size_t>(r) * padded_cap * 3, n * sizeof(double));
        }
        at += n;
    }
    return SLAMX_OK;
} SLAMX_CATCH

}  // extern "C" (C++ helpers of the gather follow)

namespace {
// librccl, bound on first use.  Signatures from /opt/rocm/include/rccl/rccl.h (ncclResult_t and the enums are ints;
// ncclUniqueId is 128 opaque bytes passed BY VALUE; ncclInt32 = 2, ncclFloat64 = 8).
struct RcclId {
    char internal[SLAMX_MAPGATHER_ID_BYTES];
};
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        // SLAMX_RCCL_NO_FALLBACK=1: only $SLAMX_RCCL_LIBRARY is tried (the missing-library path has a test)
        const char* nf = std::getenv("SLAMX_RCCL_NO_FALLBACK");
        const bool fallback = nf == nullptr || *nf == '\0' || *nf == '0';
        const char* names[] = {std::getenv("SLAMX_RCCL_LIBRARY"), fallback ? "librccl.so.1" : nullptr,
                               fallback ? "librccl.so" : nullptr, fallback ? "/opt/rocm/lib/librccl.so.1" : nullptr};
        std::string why = "?";
        for (const char* n : names) {
            if (n == nullptr || *n == '\0') {
                continue;
            }
            if ((a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) {
                break;
            }
            const char* e = dlerror();  // ONE call: dlerror() clears the state it returns
            if (e != nullptr) {
                why = e;
            }
        }
        if (a.lib == nullptr) {
            a.error = "librccl not found (" + why + "); set SLAMX_RCCL_LIBRARY";
            return a;
        }
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.lib, "ncclAllGather"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
        if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString) {
            a.error = "librccl lacks an ncclGetUniqueId / CommInitRank / CommDestroy / AllGather / GetErrorString symbol";
        }
        return a;
    }();
    return api;
}
#define RCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        const int r_ = (expr);                                                                          \
        if (r_ != 0) {                                                                                  \
            return fail(SLAMX_ERR_RUNTIME, std::string(#expr) + ": " + rccl().GetErrorString(r_));      \
        }                                                                                               \
    } while (0)
}  // namespace

struct slamx_mapgather {
    slamx_ctx* ctx = nullptr;
    void* comm = nullptr;
    int rank = 0, world = 1;
    int32_t* d_counts = nullptr;  // [3 world + 3]: layout in slamx_mapgather_create
    double* d_send = nullptr;     // [cap][3]
    double* d_recv = nullptr;     // [world][cap][3]
    int cap = 0;
    std::vector<double> h_recv;
    std::vector<int32_t> h_counts;
    // the one-collective form (slamx_mapgather_reserve / _allgather_fixed): blocks of fixed_rows x 3 doubles, row 0 = header
    size_t fixed_rows = 0;
    double* d_fsend = nullptr;  // [fixed_rows][3]
    double* d_frecv = nullptr;  // [world][fixed_rows][3]
    double* h_fsend = nullptr;  // page-locked: this rank's block, then the gathered blocks
};

extern "C" {

int slamx_mapgather_unique_id(uint8_t id[SLAMX_MAPGATHER_ID_BYTES]) try {
    if (id == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "mapgather_unique_id: null");
    }
    if (!rccl().error.empty()) {
        return fail(SLAMX_ERR_RUNTIME, rccl().error);
    }
    RcclId u{};
    RCCL_TRY(rccl().GetUniqueId(&u));
    std::memcpy(id, u.internal, SLAMX_MAPGATHER_ID_BYTES);
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_mapgather_create(slamx_ctx* ctx, const uint8_t id[SLAMX_MAPGATHER_ID_BYTES], int rank, int world,
                           slamx_mapgather** out) try {
    if (ctx == nullptr || id == nullptr || out == nullptr || world < 1 || rank < 0 || rank >= world) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "mapgather_create: bad context / id / rank / world");
    }
    if (!rccl().error.empty()) {
        return fail(SLAMX_ERR_RUNTIME, rccl().error);
    }
    HIP_TRY(hipSetDevice(ctx->device));
    auto g = std::make_unique<slamx_mapgather>();
    g->ctx = ctx;
    g->rank = rank;
    g->world = world;
    RcclId u{};
    std::memcpy(u.internal, id, SLAMX_MAPGATHER_ID_BYTES);
    // Everything that can fail locally comes BEFORE the rendezvous: a rank that returned early from here would leave
    // the others waiting in ncclCommInitRank for ever.
    // d_counts: [0, 2 world) the gathered {count, status} pairs, [2 world, 2 world + 2) this rank's own pair,
    // [2 world + 2, 3 world + 2) the gathered "buffers ready" flags, [3 world + 2] this rank's flag
    const int rc_alloc = dev_alloc(&g->d_counts, 3 * static_cast<size_t>(world) + 3);
    if (rc_alloc != SLAMX_OK) {
        return rc_alloc;
    }
    g->h_counts.assign(3 * static_cast<size_t>(world) + 3, 0);
    const int r_init = rccl().CommInitRank(&g->comm, world, u, rank);  // blocks until all `world` ranks have called it
    if (r_init != 0) {
        dev_free(g->d_counts);
        return fail(SLAMX_ERR_RUNTIME, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r_init));
    }
    *out = g.release();
    return SLAMX_OK;
} SLAMX_CATCH

void slamx_mapgather_destroy(slamx_mapgather* g) {
    if (g == nullptr) {
        return;
    }
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    if (g->comm != nullptr) {
        (void)rccl().CommDestroy(g->comm);
    }
    dev_free(g->d_counts);
    dev_free(g->d_send);
    dev_free(g->d_recv);
    dev_free(g->d_fsend);
    dev_free(g->d_frecv);
    if (g->h_fsend != nullptr) {
        (void)hipHostFree(g->h_fsend);
    }
    delete g;
}

int slamx_mapgather_allgather(slamx_mapgather* g, const double* xyz, int n_points, int32_t* counts, double* out,
                              int cap_points, int* total) try {
    if (g == nullptr || n_points < 0 || (xyz == nullptr && n_points > 0) || counts == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "mapgather_allgather: bad arguments");
    }
    // No rank may leave between the collectives: the others would wait in the next one for ever.  A LOCAL failure (a HIP
    // call, an allocation) is therefore remembered, the rank goes on through every collective -- the first carries its
    // status beside its count, the second tells whether every rank has its buffers, and only when all have does the
    // payload travel -- and the error is reported at the end, on every rank.  (A failing ncclAllGather itself means the
    // communicator is gone: nothing this rank does can release the others then.)
    int local_rc = SLAMX_OK;
    std::string local_msg;
    auto soft = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && local_rc == SLAMX_OK) {
            local_rc = SLAMX_ERR_HIP;
            local_msg = std::string(what) + ": " + hipGetErrorString(e);
        }
    };
    soft(hipSetDevice(g->ctx->device), "hipSetDevice");
    hipStream_t s = g->ctx->stream;
    constexpr int kInt32 = 2, kFloat64 = 8;  // ncclInt32, ncclFloat64
    const size_t W = static_cast<size_t>(g->world);
    int32_t* d_pairs = g->d_counts;
    int32_t* d_mine = g->d_counts + 2 * W;
    int32_t* d_ready = g->d_counts + 2 * W + 2;
    int32_t* d_myready = g->d_counts + 3 * W + 2;
    // 1. the counts, each with the rank's status so far
    const int32_t mine[2] = {n_points, local_rc};
    soft(hipMemcpyAsync(d_mine, mine, sizeof(mine), hipMemcpyHostToDevice, s), "hipMemcpyAsync(count)");
    RCCL_TRY(rccl().AllGather(d_mine, d_pairs, 2, kInt32, g->comm, s));
    soft(hipMemcpyAsync(g->h_counts.data(), d_pairs, sizeof(int32_t) * 2 * W, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(counts)");
    soft(hipStreamSynchronize(s), "hipStreamSynchronize");
    bool peers_ok = true;
    for (size_t r = 0; r < W; ++r) {
        counts[r] = local_rc == SLAMX_OK ? g->h_counts[2 * r] : 0;
        peers_ok = peers_ok && g->h_counts[2 * r + 1] == 0;
    }
    int cap = 1, tot = 0;
    if (local_rc == SLAMX_OK) {
        const int rc_plan = slamx_mapgather_plan(counts, g->world, &cap, &tot);
        if (rc_plan != SLAMX_OK) {
            local_rc = rc_plan;
            local_msg = slamx_last_error();
        }
    }
    if (total != nullptr) {
        *total = tot;
    }
    // 2. buffers for the payload, padded to the largest count (every rank computes the same `cap` from the same counts)
    if (local_rc == SLAMX_OK && cap > g->cap) {
        const int grown = std::max(cap, 2 * g->cap);
        dev_free(g->d_send);
        dev_free(g->d_recv);
        g->cap = 0;
        int rc_a = dev_alloc(&g->d_send, static_cast<size_t>(grown) * 3);
        if (rc_a == SLAMX_OK) {
            rc_a = dev_alloc(&g->d_recv, W * grown * 3);
        }
        if (rc_a == SLAMX_OK) {
            g->cap = grown;
        } else {
            local_rc = rc_a;
            local_msg = slamx_last_error();
            dev_free(g->d_send);
            dev_free(g->d_recv);
        }
    }
    if (local_rc == SLAMX_OK) {
        soft(hipMemsetAsync(g->d_send, 0, static_cast<size_t>(cap) * 3 * sizeof(double), s), "hipMemsetAsync(send)");
        if (n_points > 0) {
            soft(hipMemcpyAsync(g->d_send, xyz, static_cast<size_t>(n_points) * 3 * sizeof(double), hipMemcpyHostToDevice, s),
                 "hipMemcpyAsync(points)");
        }
    }
    const int32_t ready = local_rc == SLAMX_OK && peers_ok ? 1 : 0;
    soft(hipMemcpyAsync(d_myready, &ready, sizeof(ready), hipMemcpyHostToDevice, s), "hipMemcpyAsync(ready)");
    RCCL_TRY(rccl().AllGather(d_myready, d_ready, 1, kInt32, g->comm, s));
    soft(hipMemcpyAsync(g->h_counts.data() + 2 * W + 2, d_ready, sizeof(int32_t) * W, hipMemcpyDeviceToHost, s),
         "hipMemcpyAsync(ready flags)");
    soft(hipStreamSynchronize(s), "hipStreamSynchronize");
    int first_not_ready = -1;
    for (size_t r = 0; r < W && first_not_ready < 0; ++r) {
        if (g->h_counts[2 * W + 2 + r] == 0) {
            first_not_ready = static_cast<int>(r);
        }
    }
    if (local_rc != SLAMX_OK) {
        return fail(local_rc, "mapgather_allgather: " + local_msg + " (the collectives were entered; no payload travelled)");
    }
    if (first_not_ready >= 0) {
        // every rank sees the same flags and skips the payload collective together
        return fail(SLAMX_ERR_RUNTIME, "mapgather_allgather: rank " + std::to_string(first_not_ready) +
                                           " reported a local failure; no payload travelled");
    }
    // 3. the payload
    RCCL_TRY(rccl().AllGather(g->d_send, g->d_recv, static_cast<size_t>(cap) * 3, kFloat64, g->comm, s));
    g->h_recv.resize(W * cap * 3);
    HIP_TRY(hipMemcpyAsync(g->h_recv.data(), g->d_recv, g->h_recv.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // 4. unpack (reports a too small `out` only now: no rank may leave before the collectives)
    return slamx_mapgather_unpack(g->h_recv.data(), counts, g->world, cap, out, cap_points);
} SLAMX_CATCH

// ---- the same exchange as ONE collective (round 5): every rank sends a fixed-capacity block whose first
// row carries {count, status}; nothing is read back between collectives because there is only one.
int slamx_mapgather_reserve(slamx_mapgather* g, int per_rank_cap) try {
    if (g == nullptr || per_rank_cap < 0) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "mapgather_reserve: bad arguments");
    }
    HIP_TRY(hipSetDevice(g->ctx->device));
    const size_t rows = static_cast<size_t>(per_rank_cap) + 1;  // + the header row
    if (g->fixed_rows >= rows) {
        return SLAMX_OK;
    }
    HIP_TRY(hipStreamSynchronize(g->ctx->stream));
    dev_free(g->d_fsend);
    dev_free(g->d_frecv);
    if (g->h_fsend != nullptr) {
        (void)hipHostFree(g->h_fsend);
        g->h_fsend = nullptr;
    }
    g->fixed_rows = 0;
    SLAMX_TRY(dev_alloc(&g->d_fsend, rows * 3));
    SLAMX_TRY(dev_alloc(&g->d_frecv, static_cast<size_t>(g->world) * rows * 3));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&g->h_fsend), (static_cast<size_t>(g->world) + 1) * rows * 3 * sizeof(double), hipHostMallocDefault));
    g->fixed_rows = rows;
    return SLAMX_OK;
} SLAMX_CATCH

int slamx_mapgather_allgather_fixed(slamx_mapgather* g, const double* xyz, int n_points, int32_t* counts, double* out,
                                    int cap_points, int* total) try {
    if (g == nullptr || n_points < 0 || (xyz == nullptr && n_points > 0) || counts == nullptr) {
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "mapgather_allgather_fixed: bad arguments");
    }
    if (g->fixed_rows == 0) {
        // (the same on every rank of a correct program: nobody enters the collective)
        return fail(SLAMX_ERR_INVALID_ARGUMENT, "mapgather_allgather_fixed: call slamx_mapgather_reserve with the agreed capacity first");
    }
    // A local problem travels in the header row; the rank still enters the collective (with no points), and every rank
    // reports the failure afterwards -- nobody is left waiting.
    int local_rc = SLAMX_OK;
    std::string local_msg;
    const size_t rows = g->fixed_rows, W = static_cast<size_t>(g->world);
    if (static_cast<size_t>(n_points) + 1 > rows) {
        local_rc = SLAMX_ERR_CAPACITY;
        local_msg = "more points (" + std::to_string(n_points) + ") than the reserved capacity per rank (" + std::to_string(rows - 1) + ")";
        n_points = 0;
    }
    if (hipSetDevice(g->ctx->device) != hipSuccess && local_rc == SLAMX_OK) {
        local_rc = SLAMX_ERR_HIP;
        local_msg = "hipSetDevice";
    }
    hipStream_t s = g->ctx->stream;
    double* mine = g->h_fsend;  // page-locked: [rows][3], then the gathered [world][rows][3]
    mine[0] = static_cast<double>(n_points);
    mine[1] = static_cast<double>(local_rc);
    mine[2] = 0.0;
    if (n_points > 0) {
        std::memcpy(mine + 3, xyz, static_cast<size_t>(n_points) * 3 * sizeof(double));
    }
    // only the rows in use cross to the device; the rows behind them keep whatever an earlier call left (never read)
    const size_t used = (static_cast<size_t>(n_points) + 1) * 3;
    const bool up_ok = hipMemcpyAsync(g->d_fsend, mine, used * sizeof(double), hipMemcpyHostToDevice, s) == hipSuccess;
    constexpr int kFloat64 = 8;  // ncclFloat64
    RCCL_TRY(rccl().AllGather(g->d_fsend, g->d_frecv, rows * 3, kFloat64, g->comm, s));
    double* all = g->h_fsend + rows * 3;
    HIP_TRY(hipMemcpyAsync(all, g->d_frecv, W * rows * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));  // the one synchronisation of the exchange
    int tot = 0, first_bad = -1;
    for (size_t r = 0; r < W; ++r) {
        const double* blk = all + r * rows * 3;
        const int c = static_cast<int>(blk[0]);
        if (blk[1] != 0.0 && first_bad < 0) {
            first_bad = static_cast<int>(r);
        }
        counts[r] = c < 0 || static_cast<size_t>(c) + 1 > rows ? 0 : c;
        tot += counts[r];
    }
    if (total != nullptr) {
        *total = tot;
    }
    if (local_rc != SLAMX_OK) {
        return fail(local_rc, "mapgather_allgather_fixed: " + local_msg + " (the collective was entered with no points)");
    }
    if (!up_ok) {
        return fail(SLAMX_ERR_HIP, "mapgather_allgather_fixed: the upload of this rank's block failed");
    }
    if (first_bad >= 0) {
        return fail(SLAMX_ERR_RUNTIME, "mapgather_allgather_fixed: rank " + std::to_string(first_bad) + " reported a local failure");
    }
    if (tot > cap_points) {
        return fail(SLAMX_ERR_CAPACITY, "mapgather_allgather_fixed: " + std::to_string(tot) + " points gathered, room for " + std::to_string(cap_points));
    }
    if (out != nullptr) {
        size_t at = 0;
        for (size_t r = 0; r < W; ++r) {
            std::memcpy(out + at * 3, all + r * rows * 3 + 3, static_cast<size_t>(counts[r]) * 3 * sizeof(double));
            at += static_cast<size_t>(counts[r]);
        }
    }
    return SLAMX_OK;
} SLAMX_CATCH

}  // extern "C"
