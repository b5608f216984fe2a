/* slamx.h -- C ABI of the MI355X-native frontend hot path of daviyan5/SLAM-CIN0051.
 *
 * This is the drop-in boundary.  The reference has no FFI layer:
 * its seam is the public C++ class API of the headers under include/slam/frontend.  Each
 * entry point below names the reference interface it replaces (file:line,
 * relative to the reference root); include/slam/... in this repo is the C++17
 * facade with the reference's own class names built on top of these calls, and
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - every function returns an int status (SLAMX_OK == 0); the facade maps
 *     SLAMX_ERR_INVALID_ARGUMENT -> std::invalid_argument and every other
 *     failure -> std::runtime_error, matching the reference's exception types
 *     (feature_detector.hpp:55-93, feature_matcher.cpp:97-111);
 *   - slamx_last_error() returns a thread-local message for the last failure;
 *   - all compute runs on the GPU.  There is no CPU fallback: with no usable HIP
 *     device slamx_ctx_create fails with SLAMX_ERR_NO_DEVICE;
 *   - handles are not thread-safe; distinct contexts are independent.
 */
#ifndef SLAMX_H
#define SLAMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAMX_OK 0
#define SLAMX_ERR_INVALID_ARGUMENT 1 /* std::invalid_argument in the reference  */
#define SLAMX_ERR_RUNTIME 2          /* std::runtime_error in the reference      */
#define SLAMX_ERR_HIP 3              /* a HIP runtime call failed                */
#define SLAMX_ERR_CAPACITY 4         /* an output or workspace capacity exceeded */
#define SLAMX_ERR_NO_DEVICE 5        /* no usable gfx950 device                  */

/* slam::Keypoint, include/slam/frontend/feature_detector.hpp:28-38 (5 floats, 20 B). */
typedef struct slamx_keypoint {
    float x, y, size, angle, response;
} slamx_keypoint;

/* slam::Match, include/slam/frontend/feature_matcher.hpp:18-25 (12 B). */
typedef struct slamx_match {
    int32_t queryIdx, trainIdx;
    float distance;
} slamx_match;

/* The six YAML keys slam::FeatureDetector reads, feature_detector.hpp:60-93. */
typedef struct slamx_detector_config {
    int32_t intensity_threshold;         /* IntensityThreshold        [0,255]       */
    int32_t contiguous_pixels_threshold; /* ContiguousPixelsThreshold [0,16]        */
    int32_t non_max_suppression;         /* NonMaxSuppression         0|1           */
    int32_t suppression_window_size;     /* SuppressionWindowSize     > 0           */
    int32_t patch_size;                  /* PatchSize                 odd, > 0      */
    int32_t num_brief_pairs;             /* NumBRIEFPairs             multiple of 8 */
} slamx_detector_config;

/* The five YAML keys slam::FeatureMatcher reads, feature_matcher.cpp:25-57. */
typedef struct slamx_matcher_config {
    int32_t distance_type; /* 0 = HAMMING, 1 = L2 (L2 is accepted at construction and
                              refused by match(), as in feature_matcher.cpp:83-87)    */
    int32_t filter_matches;
    int32_t good_matches_count;
    int32_t use_ratio_test;
    float ratio_test_threshold;
} slamx_matcher_config;

/* The eight YAML keys slam::LoopClosure reads, src/backend/loop_closure.cpp:30-80. */
typedef struct slamx_loop_closure_config {
    int32_t min_db_size;
    int32_t min_frames_difference;
    double min_absolute_score;
    double relative_score_factor;
    int32_t min_matches_for_pnp;
    int32_t min_inliers_for_pnp;
    int32_t ransac_max_iterations;
    double ransac_reprojection_threshold;
} slamx_loop_closure_config;

/* Capacities of a device-resident frame batch (all per frame except max_frames). */
typedef struct slamx_batch_shape {
    int32_t rows, cols;
    int32_t max_frames;
    int32_t max_candidates; /* raw FAST corners per frame (0 = rows*cols/16)        */
    int32_t max_keypoints;  /* keypoints kept per frame   (0 = max_candidates)      */
} slamx_batch_shape;

typedef struct slamx_ctx slamx_ctx;
typedef struct slamx_detector slamx_detector;
typedef struct slamx_matcher slamx_matcher;
typedef struct slamx_batch slamx_batch;

const char* slamx_last_error(void);
const char* slamx_version(void);
/* Train-set size from which the matcher computes the all-pairs Hamming distances of a frame pair on the
 * int8 matrix cores instead of with xor + popcount (descriptors with <= 62 live bits only; results are
 * identical either way; in a batch the choice is made per frame pair on the device, a batch whose keypoint
 * capacity is below the value never takes the matrix-core scan).  Built-in 1024, overridden by the
 * environment variable SLAMX_MATCH_MFMA_MIN_KEYPOINTS. */
int slamx_matcher_mfma_threshold(void);

/* ---- context: one per GPU; owns (or borrows) the HIP stream all work is enqueued on. ---- */
/* hip_stream == NULL: the context creates its own non-blocking stream. */
/* Number of usable HIP devices (0 when there is none or the runtime cannot be reached). */
int slamx_device_count(void);
int slamx_ctx_create(int device_ordinal, void* hip_stream, slamx_ctx** out);
void slamx_ctx_destroy(slamx_ctx* ctx);
int slamx_ctx_synchronize(slamx_ctx* ctx);
void* slamx_ctx_stream(slamx_ctx* ctx);
int slamx_ctx_device_name(slamx_ctx* ctx, char* buf, int cap);
/* hipEvent pair on the context's stream (bench.py times kernels on THIS stream). */
int slamx_ctx_timer_start(slamx_ctx* ctx);
int slamx_ctx_timer_stop_ms(slamx_ctx* ctx, float* ms); /* synchronizes */

/* ---- configuration files (the OpenCV FileStorage YAML-1.0 subset the reference uses) ---- */
/* replaces the cv::FileStorage reads of feature_detector.hpp:54-94 */
int slamx_detector_config_load(const char* yaml_path, slamx_detector_config* out);
/* replaces feature_matcher.cpp:18-59 */
int slamx_matcher_config_load(const char* yaml_path, slamx_matcher_config* out);
/* LoopClosure::loadParameters (loop_closure.cpp:27-95): same keys, same range checks, same messages. */
int slamx_loop_closure_config_load(const char* yaml_path, slamx_loop_closure_config* out);

/* ---- slam::FeatureDetector ---- */
/* FeatureDetector::FeatureDetector, feature_detector.hpp:53-107 (validation + pattern). */
int slamx_detector_create(slamx_ctx* ctx, const slamx_detector_config* cfg, slamx_detector** out);
void slamx_detector_destroy(slamx_detector* det);
/* generateBRIEFPattern, feature_detector.cpp:286-313: xy4 receives {x1,y1,x2,y2} per pair. */
int slamx_detector_pattern(const slamx_detector* det, int32_t* xy4, int cap_pairs, int* n_pairs);
int slamx_detector_descriptor_width(const slamx_detector* det); /* NumBRIEFPairs / 8 */
/* The context a detector was created on (what its batches run on and what slamx_ctx_synchronize must be given). */
slamx_ctx* slamx_detector_context(const slamx_detector* det);

/* FeatureDetector::detect, feature_detector.cpp:8-18.  img: row-major uint8, host memory.
 * Writes up to cap keypoints; *n is the true count (SLAMX_ERR_CAPACITY if *n > cap). */
int slamx_detect(slamx_detector* det, const uint8_t* img, int rows, int cols, ptrdiff_t stride,
                 slamx_keypoint* kps, int cap, int* n);
/* FeatureDetector::compute, feature_detector.cpp:20-47.  kps[i].angle is overwritten;
 * desc is n x descriptor_width, row-major, host memory.  n == 0 is a no-op. */
int slamx_compute(slamx_detector* det, const uint8_t* img, int rows, int cols, ptrdiff_t stride,
                  slamx_keypoint* kps, int n, uint8_t* desc);
/* FeatureDetector::detectAndCompute, feature_detector.cpp:49-54. */
int slamx_detect_and_compute(slamx_detector* det, const uint8_t* img, int rows, int cols, ptrdiff_t stride,
                             slamx_keypoint* kps, uint8_t* desc, int cap, int* n);

/* Intermediate products, exposed for the parity tests (same kernels as above):
 * raster-ordered FAST candidates with their scores (detectFASTKeypoints + computeFASTScore,
 * feature_detector.cpp:56-68, 190-203) and the blurred image (gaussianBlur, :315-364). */
int slamx_fast_candidates(slamx_detector* det, const uint8_t* img, int rows, int cols, ptrdiff_t stride,
                          slamx_keypoint* kps, int cap, int* n);
int slamx_gaussian_blur(slamx_detector* det, const uint8_t* img, int rows, int cols, ptrdiff_t stride,
                        uint8_t* out /* rows*cols dense */);
/* The std::sort call of applyNonMaxSuppression (feature_detector.cpp:160-161) on its own:
 * sorts n scores (each an integer 0..4095) descending exactly as libstdc++'s introsort
 * permutes them; perm_out[r] = input index that lands at sorted position r. */
int slamx_debug_sort_desc(slamx_ctx* ctx, const uint32_t* scores, int n, uint32_t* perm_out);
/* ... through the two-launch form of the sort that launches of few frames take (groups > 1: sort_split_kernel, then
 * `groups` workgroups of sort_pieces_kernel; groups == 1: the call above).  Same permutation. */
int slamx_debug_sort_desc_groups(slamx_ctx* ctx, const uint32_t* scores, int n, int groups, uint32_t* perm_out);
/* The glibc libm calls of computeOrientation / computeBRIEFDescriptor
 * (feature_detector.cpp:229, 247-249) as evaluated on the device. */
int slamx_debug_trig(slamx_ctx* ctx, const float* m01, const float* m10, int n, float* angle_deg, float* cos_out,
                     float* sin_out);

/* The jump-radius penalty of findBestMatchesHamming (feature_matcher.cpp:161-170) for n
 * (hamming distance, query xy, train xy) tuples: out[i] = the penalised integer distance. */
int slamx_debug_penalty(slamx_ctx* ctx, const int32_t* dist, const float* qx, const float* qy, const float* tx,
                        const float* ty, int n, int32_t* out);

/* ---- slam::FeatureMatcher ---- */
/* FeatureMatcher::FeatureMatcher, feature_matcher.cpp:18-69. */
int slamx_matcher_create(slamx_ctx* ctx, const slamx_matcher_config* cfg, slamx_matcher** out);
void slamx_matcher_destroy(slamx_matcher* m);
/* FeatureMatcher::match, feature_matcher.cpp:71-95.  d1: n1 x width1, d2: n2 x width2, host
 * memory.  kp1/kp2 may be NULL with nkp == 0 (the "{}" defaults): the jump-radius penalty is
 * applied only when both are non-empty (:150).  Errors: n1 == 0 or n2 == 0 ->
 * SLAMX_ERR_INVALID_ARGUMENT; width mismatch or L2 -> SLAMX_ERR_RUNTIME. */
int slamx_matcher_match(slamx_matcher* m, const uint8_t* d1, int n1, int width1, const uint8_t* d2, int n2, int width2,
                const slamx_keypoint* kp1, int nkp1, const slamx_keypoint* kp2, int nkp2, slamx_match* out, int cap,
                int* n);

/* ---- device-resident batches: frames stay in HBM, nothing returns to the host until asked ---- */
int slamx_batch_create(slamx_detector* det, const slamx_batch_shape* shape, slamx_batch** out);
void slamx_batch_destroy(slamx_batch* b);
/* Device pointer of frame 0; frame f row r starts at base + f*frame_stride + r*pitch. */
uint8_t* slamx_batch_frames(slamx_batch* b, size_t* pitch, size_t* frame_stride);
/* Host -> HBM copy of n frames (row-major, given stride) into slots first..first+n-1, ordered after
 * the work already queued on the batch; the host buffer may be reused as soon as the call returns. */
int slamx_batch_upload(slamx_batch* b, int first, int n, const uint8_t* host, ptrdiff_t stride,
                       ptrdiff_t frame_stride);
/* Streaming ingest (the reference hands every frame over as a host cv::Mat, preprocessor.cpp:24-82):
 * page-locked host memory, and the same copy as slamx_batch_upload that does NOT wait -- it is queued on
 * the context stream, ordered before the compute calls that follow, and reads `host` at PCIe speed while
 * other contexts compute.  `host` must come from slamx_host_alloc and stay untouched until
 * slamx_ctx_synchronize (or any download) on this context has returned; anything else is refused
 * with SLAMX_ERR_INVALID_ARGUMENT, because the runtime stages pageable memory lazily (see above). */
int slamx_host_alloc(size_t bytes, void** out);
void slamx_host_free(void* p);
int slamx_batch_upload_async(slamx_batch* b, int first, int n, const uint8_t* host, ptrdiff_t stride,
                             ptrdiff_t frame_stride);
/* The same with flags.  SLAMX_UPLOAD_SLOTS_IDLE: the caller guarantees that no work queued earlier on the batch's context
 * reads or writes the slots [first, first + n) -- the copy is then NOT ordered behind that work and runs beside it on the
 * context's copy queue (a caller that fills a batch chunk by chunk while earlier chunks are being processed:
 * slam::FrameBatch::processSequence).  Work queued AFTER the call still waits for the copy. */
#define SLAMX_UPLOAD_SLOTS_IDLE 1
int slamx_batch_upload_async_flags(slamx_batch* b, int first, int n, const uint8_t* host, ptrdiff_t stride,
                                   ptrdiff_t frame_stride, int flags);
/* detectAndCompute on frames 0..nframes-1 (async on the context stream). */
int slamx_batch_detect_and_compute(slamx_batch* b, int nframes);
/* match(desc[k], desc[k-1], kps[k], kps[k-1]) for k = 1..nframes-1; frame 0 is matched against
 * frame first_train_frame of the same batch (the slot a streaming caller keeps the previous
 * batch's last frame in), or not at all when first_train_frame < 0.  use_keypoints selects the
 * jump-radius penalty.  Results stay on the device in query order, before
 * filterAndSortMatches (async). */
int slamx_batch_match_consecutive(slamx_batch* b, slamx_matcher* m, int nframes, int use_keypoints,
                                  int first_train_frame);
/* The two calls above for the slots [first, first + count) only (frames handed over chunk by chunk): detect + describe
 * of those frames; then each of them matched against its predecessor -- frame `first` against frame first - 1, which an
 * earlier call described, or, when first == 0, against first_train_frame as above.  Results are exactly those of the
 * whole-batch calls (a frame's results do not depend on its neighbours in the launch). */
int slamx_batch_detect_and_compute_range(slamx_batch* b, int first, int count);
int slamx_batch_match_consecutive_range(slamx_batch* b, slamx_matcher* m, int first, int count, int use_keypoints,
                                        int first_train_frame);
/* Synchronize and copy out.  A frame whose candidate/keypoint count exceeded the shape's
 * capacity makes these return SLAMX_ERR_CAPACITY. */
int slamx_batch_counts(slamx_batch* b, int nframes, int32_t* n_candidates, int32_t* n_keypoints, int32_t* n_matches);
int slamx_batch_download(slamx_batch* b, int frame, slamx_keypoint* kps, uint8_t* desc, int cap, int* n);
/* Matches of frame k against frame k-1, with the matcher's filter step applied on request. */
/* The frame as it sits in HBM (blurred == 0) or its 5x5-blurred copy made by the last
 * detect_and_compute (blurred != 0): what gaussianBlur (feature_detector.cpp:315-364) returned. */
int slamx_batch_download_image(slamx_batch* b, int frame, int blurred, uint8_t* out, ptrdiff_t stride);
int slamx_batch_download_matches(slamx_batch* b, slamx_matcher* m, int frame, int apply_filter, slamx_match* out,
                                 int cap, int* n);
/* The matcher's filterAndSortMatches step (feature_matcher.cpp:191-204: partial_sort + truncate to GoodMatchesCount, or
 * sort, when FilterMatches is set; nothing otherwise) on a raw match list the caller already holds on the host (a frame's
 * range of the packed download): in place, *n_out = the number of matches left.  Host only: no device work. */
int slamx_matcher_filter_matches(const slamx_matcher* m, slamx_match* matches, int n, int* n_out);
/* The streaming form of the downloads: queues the copy of the results of frames [0, nframes) into
 * PAGE-LOCKED host memory (slamx_host_alloc) on the context stream and returns; any pointer may be
 * NULL.  Layout = the device's capacity-padded arrays: kps [nframes][max_keypoints],
 * desc [nframes][max_keypoints][width], matches [nframes][max_keypoints]; n_keypoints / n_matches
 * [nframes] say how much of each frame's slice is valid.  Synchronise the context before reading. */
int slamx_batch_download_results_async(slamx_batch* b, int nframes, slamx_keypoint* kps, uint8_t* desc,
                                       slamx_match* matches, int32_t* n_keypoints, int32_t* n_matches);

/* The PACKED form of the streaming download.  A device-side pack step (a prefix over the per-frame counts, then one
 * copy kernel) gathers the VALID keypoints, descriptors and matches of frames [0, nframes) -- what the reference
 * returns per frame in its std::vector<Keypoint> / DescriptorMatrix / std::vector<Match> (feature_detector.cpp:49-54,
 * feature_matcher.cpp:71-95) -- into ONE contiguous device buffer, and the first `bytes` of it are copied into
 * PAGE-LOCKED `host` on the context stream (about 116 KB per 1080p frame of 2k keypoints instead of the 262 KB of
 * the capacity-padded arrays).  Buffer layout, every section 16-byte aligned, offsets in bytes from its start:
 *     slamx_packed_header
 *     int32 kp_first[nframes + 1]      exclusive prefix of the keypoint counts: frame f owns [kp_first[f], kp_first[f+1])
 *     int32 match_first[nframes + 1]   the same for the matches (all zero when no matcher has run on the batch)
 *     int32 status[nframes]            per-frame overflow bits (0 = complete)
 *     slamx_keypoint kps[total_keypoints]; uint8 desc[total_keypoints][width]; slamx_match matches[total_matches]
 * `host` is page-locked memory (slamx_host_alloc) of at least `bytes` bytes; at most `bytes` of it are written.  By
 * default the pack kernel stores the results there itself, so only header.total_bytes cross PCIe even when `bytes`
 * is the worst case (slamx_batch_packed_bound()); with SLAMX_PACKED_DOWNLOAD=copy the first `bytes` of the packed
 * buffer are copied, whatever they hold, and a streaming caller passes a little more than the previous batch needed.
 * After synchronising the caller reads header.total_bytes: if that exceeds `bytes`,
 * slamx_batch_download_packed_tail fetches the rest of the same packed buffer (synchronous), valid until the next pack
 * of this batch. */
typedef struct slamx_packed_header {
    int32_t nframes, width, total_keypoints, total_matches;
    int64_t first_off, mfirst_off, status_off, kps_off, desc_off, matches_off, total_bytes;
} slamx_packed_header;
size_t slamx_batch_packed_bound(slamx_batch* b, int nframes);
int slamx_batch_download_packed_async(slamx_batch* b, int nframes, void* host, size_t bytes);
int slamx_batch_download_packed_tail(slamx_batch* b, void* host, size_t from, size_t host_capacity);

/* Per-stage device time from hipEvents recorded on the context stream between the kernels (no
 * synchronization is added).  slamx_batch_set_profiling(b, n_sets) keeps n_sets event sets
 * (0 = off); the i-th detect_and_compute call after it records into set i % n_sets, and the
 * match call that follows uses the same set.  slamx_batch_stage_ms synchronizes and reads one
 * set.  Every stage except NMS_ROUNDS is exactly one kernel launch. */
#define SLAMX_STAGE_FAST 0       /* fast_detect_kernel                         */
#define SLAMX_STAGE_SCAN 1       /* scan_kernel                                */
#define SLAMX_STAGE_EXPAND 2     /* expand_kernel                              */
#define SLAMX_STAGE_SORT 3       /* sort_kernel                                */
#define SLAMX_STAGE_NMS_BUILD 4  /* nms_build_kernel                           */
#define SLAMX_STAGE_NMS_ROUNDS 5 /* nms_round_kernel x2 + nms_finish_kernel    */
#define SLAMX_STAGE_BLUR 6       /* blur_kernel                                */
#define SLAMX_STAGE_MOMENTS 7    /* moments_kernel (or the fused describe_kernel for large patches) */
#define SLAMX_STAGE_TRIG 8       /* trig_kernel                                */
#define SLAMX_STAGE_BRIEF 9      /* brief_kernel                               */
#define SLAMX_STAGE_MATCH 10     /* match_kernel                               */
#define SLAMX_STAGE_EMIT 11      /* emit_kernel                                */
#define SLAMX_STAGE_COUNT 12
int slamx_batch_set_profiling(slamx_batch* b, int n_sets);
/* Same, but only every `every`-th detect_and_compute call (and the match that follows it) records:
 * an event between two kernels keeps the second from starting while the first drains, which costs
 * ~9 % of a 13-kernel step; call k * every lands in set k % n_sets. */
int slamx_batch_set_profiling_every(slamx_batch* b, int n_sets, int every);
int slamx_batch_stage_ms(slamx_batch* b, int set_index, float* ms /* SLAMX_STAGE_COUNT entries */);

/* ---- small dense pose solves (src/backend/loop_closure.cpp, src/frontend/simple_pose_recover.cpp) ----
 * Host pointers, double precision, row-major 3x3 / 3x4 matrices.  The reference delegates these
 * to Eigen::JacobiSVD / cv::SVD (third-party, sign conventions unpinned): results agree with a
 * double-precision SVD to ~1e-12; singular-vector signs are this implementation's own. */
/* LoopClosure::solvePnP (loop_closure.cpp:238-274) for n_hyp independent point sets of n_points
 * (>= 6, <= 32) each: p3d [n_hyp][n_points][3], p2d [n_hyp][n_points][2] -> R [n_hyp][9],
 * t [n_hyp][3]; nullvec (optional) [n_hyp][12] = the DLT null vector.  The 2n x 12 system is
 * contracted on the FP64 matrix cores. */
int slamx_pnp_dlt_batch(slamx_ctx* ctx, const double* p3d, const double* p2d, int n_points, int n_hyp, double* R,
                        double* t, double* nullvec);
/* Inlier count of n_hyp poses over n correspondences (loop_closure.cpp:201-215). */
int slamx_count_inliers_batch(slamx_ctx* ctx, const double* p3d, const double* p2d, int n, const double* K,
                              const double* R, const double* t, int n_hyp, double threshold, int32_t* counts);
/* The RANSAC loop of verifyGeometricConsistency (loop_closure.cpp:173-222) with an explicit seed
 * (the reference seeds std::mt19937 from std::random_device): sampling with the same
 * std::mt19937 / std::uniform_int_distribution calls on the host, all solves and counts in two
 * GPU launches, first maximum wins.  max_inliers == 0 leaves R = I, t = 0 as the reference does. */
int slamx_pnp_ransac(slamx_ctx* ctx, const double* p3d, const double* p2d, int n, const double* K, int iterations,
                     double threshold, uint32_t seed, double* R, double* t, int32_t* max_inliers);
/* Essential matrix by RANSAC, the step PoseEstimator::estimate delegates to cv::findEssentialMat
 * (src/frontend/pose_estimator.cpp:37-47; third-party in the reference, so this is the build's own
 * estimator: normalised 8-point hypotheses, Sampson-distance scoring, all hypotheses solved and scored
 * in two launches).  pts1/pts2: n x 2 float PIXEL coordinates of the matched keypoints, K 3x3
 * row-major; threshold_px is divided by the mean focal length as OpenCV does; sampling as in the
 * reference's own RANSAC (std::mt19937 + uniform_int_distribution, distinct indices) from `seed`.
 * Out: E 3x3 row-major with singular values (1,1,0), the inlier count of the winning hypothesis (first
 * maximum) and optionally one flag per correspondence.  n < 8 -> SLAMX_ERR_INVALID_ARGUMENT
 * (pose_estimator.cpp:21-25).  Feed E to slamx_recover_pose with the normalised points. */
int slamx_essential_ransac(slamx_ctx* ctx, const float* pts1, const float* pts2, int n, const double* K, int iterations,
                           double threshold_px, uint32_t seed, double* E, int32_t* n_inliers, uint8_t* inlier_mask);

/* simpleRecoverPose (simple_pose_recover.cpp:35-98): E 3x3, pts1/pts2 n x 2 float (normalised
 * points as PoseEstimator::estimate passes them), K 3x3 -> R 3x3, t 3; front (optional) gets the
 * four cheirality counts. */
int slamx_recover_pose(slamx_ctx* ctx, const double* E, const float* pts1, const float* pts2, int n, const double* K,
                       double* R, double* t, int32_t* front);
/* slam::triangulate + the conversion loop of PoseEstimator::triangulatePoints (common.hpp:201-221,
 * pose_estimator.cpp:94-101): P1, P2 3x4, pts n x 2 float pixel coordinates -> x4 (optional)
 * n x 4 float homogeneous, xyz n x 3 double. */
int slamx_triangulate(slamx_ctx* ctx, const double* P1, const double* P2, const float* pts1, const float* pts2, int n,
                      float* x4, double* xyz);

/* The same two RANSAC loops with the sample table given by the caller instead of drawn from a seed: idx holds
 * `iterations` rows of 8 (essential) / 6 (PnP) indices into the correspondences.  slamx_essential_ransac and
 * slamx_pnp_ransac are exactly these calls on the table their std::mt19937 produces; the batched stage below draws its
 * tables on the device (slamx_pose_download_samples) and must give the same result as these calls fed with them. */
int slamx_essential_from_samples(slamx_ctx* ctx, const float* pts1, const float* pts2, int n, const double* K,
                                 const int32_t* idx, int iterations, double threshold_px, double* E, int32_t* n_inliers,
                                 uint8_t* inlier_mask);
int slamx_pnp_from_samples(slamx_ctx* ctx, const double* p3d, const double* p2d, int n, const double* K, const int32_t* idx,
                           int iterations, double threshold, double* R, double* t, int32_t* max_inliers);
/* ... with the hypothesis form named (geometry_mode: SLAMX_GEOMETRY_REFERENCE = the call above,
 * SLAMX_GEOMETRY_CALIBRATED: see slamx_pose_config below); Rt_all (optional) [iterations][12] and counts (optional) [iterations] receive every
 * hypothesis' pose (R row-major, then t) and inlier count. */
int slamx_pnp_from_samples_mode(slamx_ctx* ctx, const double* p3d, const double* p2d, int n, const double* K,
                                const int32_t* idx, int iterations, double threshold, int geometry_mode, double* R, double* t,
                                int32_t* max_inliers, double* Rt_all, int32_t* counts);

/* ---- the pose stage of a device-resident batch (BASELINE.json configs[2]: "+ 2-view essential-matrix / PnP pose") ----
 * For every frame f of the batch whose matches against its predecessor are on the device (slamx_batch_match_consecutive:
 * query = f, train = f - 1, or first_train_frame for f = 0), in a fixed number of launches over ALL frames and with no
 * host round trip, allocation or synchronisation:
 *   1. PoseEstimator::estimate (src/frontend/pose_estimator.cpp:18-67) with (keypoints1, keypoints2) = (train frame,
 *      query frame): the correspondences are gathered from the keypoint and match arrays where they are, normalised by
 *      K^-1; the essential matrix comes from this build's own RANSAC (cv::findEssentialMat is third-party: normalised
 *      8-point hypotheses, Sampson scoring, first maximum, refit on the inliers -- the algorithm of slamx_essential_ransac,
 *      same kernels); then simpleRecoverPose (src/frontend/simple_pose_recover.cpp:35-98) on all correspondences.
 *      Fewer than 8 matches: no pose (pose_estimator.cpp:21-25), SLAMX_POSE_TOO_FEW_MATCHES, R = I, t = 0.
 *   2. PoseEstimator::triangulatePoints (pose_estimator.cpp:69-104) with K [I|0], K [R|t]: one map point per match.
 *   3. if pnp_iterations > 0: the RANSAC of LoopClosure::verifyGeometricConsistency (src/backend/loop_closure.cpp:
 *      173-222) for frame f against the map points of its train frame (those triangulated in step 2 of the pair whose
 *      QUERY frame the train frame is): 3D point of match.trainIdx, 2D keypoint of match.queryIdx, for every match whose
 *      train keypoint has a map point; 6-point DLT solvePnP hypotheses (loop_closure.cpp:238-274), reprojection inlier
 *      count, first maximum from maxInliers = 0 (identity / zero when nothing scores).  Fewer than
 *      max(6, pnp_min_correspondences) correspondences: SLAMX_POSE_PNP_SKIPPED -- also for frame 0 when its train
 *      slot first_train_frame is not one of this call's frames [0, nframes): that slot's map points are not this call's.
 * Samples are drawn on the device by a counter-based generator (splitmix64 of (seed, frame slot, hypothesis, draw),
 * indices redrawn until distinct as loop_closure.cpp:185-193 does); slamx_pose_download_samples returns the tables.
 * The match lists are taken as slamx_batch_match_consecutive left them (query order, before filterAndSortMatches). */
typedef struct slamx_pose_config {
    double K[9];               /* intrinsics, row-major                                                     */
    int32_t iterations;        /* essential-matrix hypotheses per frame pair, 1..65536                       */
    double threshold_px;       /* Sampson threshold in pixels (divided by the mean focal length)             */
    uint32_t seed;
    int32_t pnp_iterations;    /* RansacMaxIterations (loop_closure.yml); 0 = no PnP step                    */
    double pnp_threshold;      /* RansacReprojectionThreshold                                                */
    int32_t pnp_min_correspondences; /* MinMatchesForPnP; values below 6 mean 6                              */
    int32_t max_matches;       /* capacity per frame pair; 0 = the batch's max_keypoints                     */
    int32_t geometry_mode;     /* SLAMX_GEOMETRY_REFERENCE (0, default) or SLAMX_GEOMETRY_CALIBRATED         */
} slamx_pose_config;
/* Two steps of the reference's pose code mix units, so that on a scene with real depth and a real K they return
 * something else than the motion (hard part 6).  geometry_mode says which form the stage runs:
 * SLAMX_GEOMETRY_REFERENCE -- both steps exactly as the reference writes them:
 *   - simpleRecoverPose triangulates the K^-1-NORMALISED points it is handed (pose_estimator.cpp:53-66) against
 *     K [I|0] and K [R|t] (simple_pose_recover.cpp:61-82): the cheirality vote then gives all correspondences to one
 *     candidate whatever the geometry, R comes out right and the SIGN of t is arbitrary;
 *   - LoopClosure::solvePnP (loop_closure.cpp:238-274): the DLT on PIXEL coordinates, the COLUMN-major 3x4 map of the
 *     row-major DLT vector (:260), t = t0 / ||R0||_F (:271: sqrt(3) times the scale of a scaled rotation) -- no
 *     hypothesis of this form reprojects a real scene within the threshold.
 *   This mode reproduces the reference, arbitrary sign and zero inliers included; the parity tests use it.
 * SLAMX_GEOMETRY_CALIBRATED (build-defined, NO reference counterpart, parity unpinned) -- the unit-consistent forms:
 *   - the cheirality vote triangulates the normalised points against [I|0] and [R|t] (= simpleRecoverPose called with
 *     K = identity; slamx_recover_pose with an identity K is the one-pair form);
 *   - the PnP DLT runs on K^-1-normalised image points (K upper triangular with K[8] = 1), the ROW-major 3x4 map, the sign
 *     of the null vector chosen so that det(R0) > 0, the same orthonormalisation R = U diag(1, 1, det(U V^T)) V^T,
 *     t = t0 / mean singular value of R0.  The inlier test (loop_closure.cpp:201-215) is the reference's in both modes.
 * Everything else of the stage (gather, essential-matrix RANSAC, decomposeEssential, triangulatePoints) is common. */
#define SLAMX_GEOMETRY_REFERENCE 0
#define SLAMX_GEOMETRY_CALIBRATED 1
#define SLAMX_POSE_TOO_FEW_MATCHES 1 /* fewer than 8 matches: no essential matrix, no map points             */
#define SLAMX_POSE_PNP_SKIPPED 2     /* fewer correspondences than the PnP step needs                        */
#define SLAMX_POSE_MATCH_OVERFLOW 4  /* more matches than max_matches: the first max_matches were used        */
#define SLAMX_POSE_NO_TRAIN_FRAME 8  /* frame 0 with first_train_frame < 0: nothing to estimate               */
typedef struct slamx_pose_result {
    double E[9], R[9], t[3];   /* essential matrix (singular values 1,1,0), pose of the query frame's camera  */
    double pnp_R[9], pnp_t[3];
    int32_t n_matches, n_inliers, front[4], pnp_correspondences, pnp_inliers, status, best_hypothesis;
} slamx_pose_result;
typedef struct slamx_pose slamx_pose;
/* The stage's workspace is sized for the batch's max_frames x max_matches at creation; destroy it BEFORE its batch. */
int slamx_pose_create(slamx_batch* b, const slamx_pose_config* cfg, slamx_pose** out);
void slamx_pose_destroy(slamx_pose* p);
/* Enqueue the stage for frames [0, nframes) on the context stream (async; no allocation, no synchronisation). */
int slamx_batch_pose(slamx_pose* p, int nframes, int first_train_frame);
/* Synchronise and copy out: one result per frame. */
int slamx_pose_download(slamx_pose* p, int nframes, slamx_pose_result* out);
/* Map points of one frame's pair, parallel to its match list (n = n_matches): xyz n x 3 doubles, inlier_mask (optional)
 * n bytes = the essential matrix's inlier flags. */
int slamx_pose_download_points(slamx_pose* p, int frame, double* xyz, uint8_t* inlier_mask, int cap, int* n);
/* The sample tables the device drew for one frame: idx [iterations][8], pnp_idx [pnp_iterations][6] (either may be NULL). */
int slamx_pose_download_samples(slamx_pose* p, int frame, int32_t* idx, int32_t* pnp_idx);
/* Intermediate products of step 1 for the parity tests: the essential matrix of every hypothesis of one frame
 * (E_all [iterations][9]) and its Sampson inlier count (counts [iterations]); either may be NULL. */
int slamx_pose_download_hypotheses(slamx_pose* p, int frame, double* E_all, int32_t* counts);
/* ... and of step 3: pose (Rt_all [pnp_iterations][12]: R row-major, then t), DLT null vector before the column-major
 * re-map (null_all [pnp_iterations][12]) and inlier count of every PnP hypothesis of one frame; any may be NULL. */
int slamx_pose_download_pnp_hypotheses(slamx_pose* p, int frame, double* Rt_all, double* null_all, int32_t* counts);
/* The 3D-2D correspondences step 3 formed for one frame (for tests): p3d n x 3, p2d n x 2 doubles. */
int slamx_pose_download_correspondences(slamx_pose* p, int frame, double* p3d, double* p2d, int cap, int* n);
/* Per-kernel device time of the stage (hipEvents on the context stream around every launch of the calls since
 * profiling was switched on; on = n > 1 records every n-th call only). */
#define SLAMX_POSE_STAGE_GATHER 0      /* pose_gather_kernel                                                 */
#define SLAMX_POSE_STAGE_HYPOTHESES 1  /* essential_hyp_kernel (sampling + 8-point solve, lane per hypothesis) */
#define SLAMX_POSE_STAGE_SCORE 2       /* sampson_score_kernel: every hypothesis x every correspondence       */
#define SLAMX_POSE_STAGE_SELECT 3      /* pose_select_kernel (first maximum, inlier flags of the winner)      */
#define SLAMX_POSE_STAGE_REFIT 4       /* essential_refit_kernel (normal matrix of the inliers on FP64 MFMA)  */
#define SLAMX_POSE_STAGE_FINAL 5       /* pose_final_kernel (refit accepted or not, decomposeEssential)       */
#define SLAMX_POSE_STAGE_CHEIRALITY 6  /* cheirality_kernel (4 candidates x every correspondence)             */
#define SLAMX_POSE_STAGE_TRIANGULATE 7 /* triangulate_pairs_kernel                                            */
#define SLAMX_POSE_STAGE_PNP_GATHER 8  /* pnp_gather_kernel                                                   */
#define SLAMX_POSE_STAGE_PNP_SOLVE 9   /* pnp_hyp_kernel (sampling + 12 x 12 DLT, lane per hypothesis)        */
#define SLAMX_POSE_STAGE_PNP_SCORE 10  /* pnp_score_kernel + pnp_select_kernel                                */
#define SLAMX_POSE_STAGE_COUNT 11
int slamx_pose_set_profiling(slamx_pose* p, int on);
int slamx_pose_stage_ms(slamx_pose* p, float* avg_ms /* SLAMX_POSE_STAGE_COUNT */, int32_t* launches /* same, optional */);
/* Keypoints and matches of one frame slot written from the host (a caller that has them from elsewhere -- the keyframe
 * database of LoopClosure::addKeyframe, loop_closure.cpp:96-109 -- and the parity tests of the pose stage): kps n_kp
 * keypoints, matches n_matches rows (queryIdx ascending, as the matcher leaves them) against the slot's train frame.
 * Descriptors of the slot are left as they are. */
int slamx_batch_upload_results(slamx_batch* b, int frame, const slamx_keypoint* kps, int n_kp, const slamx_match* matches,
                               int n_matches);

/* ---- image pyramid (BASELINE.json north_star / configs[2]; NO reference counterpart) ----
 * The reference detects on the full-resolution image only (feature_detector.cpp:8-18, Keypoint::size
 * is the constant 6.0, feature_detector.hpp:18,35).  A pyramid is a stack of batches: level 0 has the
 * caller's shape, level l is cvRound(size / scale_factor^l) (sizes computed in double as ORB-style
 * pyramids do) and is an integer bilinear downscale of level l-1 (pyramid.hip).
 * slamx_pyramid_detect_and_compute builds the levels from the frames in level 0, runs the unchanged
 * detector on every level and appends the results of levels >= 1 to level 0's arrays, level after
 * level: x, y and size multiplied by float(scale_factor^l), angle / response / descriptor as they
 * are.  Frames go in and merged results come out through the level-0 batch (slamx_pyramid_batch(p, 0)
 * with slamx_batch_upload / _match_consecutive / _counts / _download); shape.max_keypoints is the
 * merged capacity; shape.max_candidates applies to every level (downscaled levels hold about as many raw
 * corners as level 0). */
typedef struct slamx_pyramid slamx_pyramid;
#define SLAMX_PYRAMID_MAX_LEVELS 16
int slamx_pyramid_create(slamx_detector* det, const slamx_batch_shape* shape, int n_levels, double scale_factor,
                         slamx_pyramid** out);
void slamx_pyramid_destroy(slamx_pyramid* p);
slamx_batch* slamx_pyramid_batch(slamx_pyramid* p, int level);
int slamx_pyramid_level_shape(const slamx_pyramid* p, int level, int* rows, int* cols, double* scale);
int slamx_pyramid_detect_and_compute(slamx_pyramid* p, int nframes);

/* ---- sliding-window Gauss-Newton (BASELINE.json configs[3]; NO reference counterpart) ----
 * The reference's backend is declaration-only (include/slam/backend/backend.hpp:10-18,
 * include/slam/backend/map.hpp:9-21), so there is no interface to mirror: this is the entry point a
 * Backend::run / Map implementation would call.  A batch of independent windows, each with n_cams
 * (<= 10) poses (world -> camera, 12 doubles: R row-major, t), up to max_landmarks landmarks and a
 * dense track table obs[landmark][cam] = (u, v) float pixel coordinates, NaN = not observed.
 * One iteration = linearise the reprojection error, accumulate J^T J / J^T r and eliminate the
 * landmarks on the FP64 matrix cores (gn.hip), solve the reduced camera system (6 n_cams unknowns,
 * the first n_fixed cameras held, lambda added to every diagonal), back-substitute the landmarks,
 * apply the step (R <- exp(phi^) R, t <- exp(phi^) t + rho, X <- X + dX).  Device-resident between
 * calls; results are run-to-run identical (no atomics). */
typedef struct slamx_gn slamx_gn;
#define SLAMX_GN_MAX_CAMS 10
#define SLAMX_GN_MAX_ITERATIONS 64
int slamx_gn_create(slamx_ctx* ctx, int n_windows, int n_cams, int max_landmarks, slamx_gn** out);
void slamx_gn_destroy(slamx_gn* gn);
/* poses [n_windows][n_cams][12], landmarks [n_windows][max_landmarks][3],
 * obs [n_windows][max_landmarks][n_cams][2], n_landmarks [n_windows] or NULL (= max_landmarks). */
int slamx_gn_upload(slamx_gn* gn, const double* poses, const double* landmarks, const float* obs,
                    const int32_t* n_landmarks);
/* Enqueue `iterations` (0..64) steps plus a final cost evaluation; K = fx fy cx cy. */
int slamx_gn_iterate(slamx_gn* gn, const double* K, double lambda, int n_fixed, int iterations);
/* Any pointer may be NULL.  cost [n_windows][iterations + 1] of the last slamx_gn_iterate call
 * (sum of squared residuals before every step and after the last); status [n_windows]: bit 0 =
 * some step was skipped because the reduced system was not positive definite. */
int slamx_gn_download(slamx_gn* gn, double* poses, double* landmarks, double* cost, int32_t* status);
/* Per-kernel device time of the iterations enqueued since profiling was switched on: hipEvent pairs on the
 * context stream around every launch.  avg_ms[3] = accumulate, solve, back-substitution; launches[3]
 * (optional) = how many launches each average covers.  on = 0: off; 1: every iteration records; n > 1: every
 * n-th iteration (an event between two kernels keeps the second from starting while the first drains). */
int slamx_gn_set_profiling(slamx_gn* gn, int on);
int slamx_gn_stage_ms(slamx_gn* gn, float* avg_ms, int32_t* launches);
/* Linearise at the current state WITHOUT stepping: S [n_windows][6 n_cams][6 n_cams], g and delta
 * [n_windows][6 n_cams] (S delta = g), cost [n_windows].  For parity tests and callers that run
 * their own step control. */
int slamx_gn_normal_equations(slamx_gn* gn, const double* K, double lambda, int n_fixed, double* S, double* g,
                              double* delta, double* cost);

/* ---- slam::Camera (include/slam/common/common.hpp:67-190) ---- */
typedef struct slamx_camera {
    double fx, fy, cx, cy;     /* K(0,0), K(1,1), K(0,2), K(1,2)                           */
    double k1, k2, p1, p2, k3; /* D(0..4); k3 is loaded and unused, as in the reference     */
    int32_t width, height;     /* ImageSize                                                */
    double K[9];               /* the full 3x3 intrinsic matrix, row-major                 */
} slamx_camera;
/* Camera::Camera(configPath, cameraIndex), common.hpp:76-119: keys K<i>, D<i>, ImageSize. */
int slamx_camera_load(const char* yaml_path, int camera_index, slamx_camera* out);
/* Camera::undistortImage, common.hpp:127-173: nearest-neighbour inverse mapping of the
 * [0,1]-scaled image.  img: rows x cols uint8 (must equal ImageSize), out: rows x cols double,
 * ROW-major (the reference returns a column-major MatrixXd; same values). */
int slamx_undistort(slamx_ctx* ctx, const slamx_camera* cam, const uint8_t* img, int rows, int cols, ptrdiff_t stride,
                    double* out);

/* ---- the one exchange of the multi-GPU path: all-gather of map points over RCCL ----
 * Frames shard across GPUs with no collective on the data path; at the end of a job (or per keyframe batch) every
 * rank contributes the map points it triangulated (PoseEstimator::triangulatePoints, pose_estimator.cpp:69-104:
 * n x 3 doubles, n differs per rank) and receives everyone's.  The reference has no multi-GPU code; this is the
 * C-ABI a C++17 host uses (one host thread or process + one context per GPU; tools/cli -g N); bench.py's Python
 * ranks do the same exchange through torch.distributed.  librccl is loaded on first use (dlopen: $SLAMX_RCCL_LIBRARY,
 * then librccl.so.1 / librccl.so), so the library has no link-time dependency on it.
 *   rank 0:   slamx_mapgather_unique_id(id); hand the 128 bytes to the other ranks (shared memory, a file, a socket)
 *   all:      slamx_mapgather_create(ctx, id, rank, world, &g)       -- ncclCommInitRank: every rank must call it
 *             slamx_mapgather_allgather(g, xyz, n, counts, out, cap, &total)
 * allgather = ncclAllGather of the counts, then ncclAllGather of the payload padded to the largest count (xGMI
 * messages of KB to MB: latency-bound), on the context's stream; returns when `out` holds the points of rank 0,
 * rank 1, ... back to back (total x 3 doubles) and counts[r] says how many are rank r's.  total > cap_points ->
 * SLAMX_ERR_CAPACITY after the collectives have run (no rank is left waiting), *total set. */
#define SLAMX_MAPGATHER_ID_BYTES 128
typedef struct slamx_mapgather slamx_mapgather;
int slamx_mapgather_unique_id(uint8_t id[SLAMX_MAPGATHER_ID_BYTES]);
int slamx_mapgather_create(slamx_ctx* ctx, const uint8_t id[SLAMX_MAPGATHER_ID_BYTES], int rank, int world,
                           slamx_mapgather** out);
void slamx_mapgather_destroy(slamx_mapgather* g);
int slamx_mapgather_allgather(slamx_mapgather* g, const double* xyz, int n_points, int32_t* counts /* [world] */,
                              double* out, int cap_points, int* total);
/* The same exchange as ONE collective, for callers whose ranks agree on a capacity beforehand (a batch's max_keypoints:
 * a frame pair has at most that many matches, hence map points): every rank sends a block of per_rank_cap + 1 rows of three
 * doubles whose first row carries {count, status}; one ncclAllGather, one copy of the gathered blocks to the host, one
 * synchronisation -- no host round trip between collectives (the two-step form above needs the counts on the host before
 * it can size the payload).  slamx_mapgather_reserve(g, per_rank_cap): local, allocates the blocks, to be called by every
 * rank with the SAME capacity before the first exchange (a rank that fails here must not enter the exchange alone: treat
 * it like a failed slamx_mapgather_create).  _allgather_fixed: as slamx_mapgather_allgather; a rank with more points than
 * the capacity, or any other local problem, still enters the collective (with no points) and every rank returns an error
 * afterwards. */
int slamx_mapgather_reserve(slamx_mapgather* g, int per_rank_cap);
int slamx_mapgather_allgather_fixed(slamx_mapgather* g, const double* xyz, int n_points, int32_t* counts /* [world] */,
                                    double* out, int cap_points, int* total);
/* The host-side bookkeeping of the exchange on its own (no GPU, no RCCL): the padded per-rank capacity (largest
 * count, at least 1) and the total for a list of counts; and the unpacking of a gathered padded payload
 * [world][padded_cap][3] into `out`.  slamx_mapgather_allgather uses exactly these. */
int slamx_mapgather_plan(const int32_t* counts, int world, int* padded_cap, int* total);
int slamx_mapgather_unpack(const double* padded, const int32_t* counts, int world, int padded_cap, double* out,
                           int cap_points);

/* ---- deterministic synthetic frames, host side, for tests and bench ---- */
/* Frame k of the sequence with the given seed: background 96, n_rect filled rectangles with
 * sides in [rect_min, rect_max], per-pixel noise in [-4,4], translated by (2k mod 64, k mod 32). */
int slamx_synth_frame(uint64_t seed, int k, int rows, int cols, int n_rect, int rect_min, int rect_max,
                      uint8_t* out /* rows*cols dense */);

#ifdef __cplusplus
}
#endif
#endif /* SLAMX_H */
