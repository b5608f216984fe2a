// slamx_internal.h -- shared between the HIP kernels and the C-ABI host code.
// gfx950 only: wave = 64 lanes everywhere in this code base.
#pragma once

#include <atomic>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/slamx.h"

namespace slamx {

constexpr int WAVE = 64;
constexpr int SCORE_SHIFT = 20;            // key = score << 20 | raster index
constexpr uint32_t INDEX_MASK = (1u << SCORE_SHIFT) - 1;
constexpr int MAX_CANDIDATES_HARD = 1 << SCORE_SHIFT;
constexpr int SORT_LEAF = 16;              // libstdc++ _S_threshold (bits/stl_algo.h)
constexpr int MAX_PATTERN = 1024;          // NumBRIEFPairs upper bound held in constant memory

// NMS states (by sorted rank)
constexpr uint8_t NMS_UNDECIDED = 0;
constexpr uint8_t NMS_KEPT = 1;
constexpr uint8_t NMS_SUPPRESSED = 2;
constexpr int MATCH_MAX_SPLIT = 8;
// Train sets of at least this many keypoints take the matrix-core scan (match.hip), smaller ones the xor + popcount
// scan (640x480, per 1024 frames: 0.258 vs 0.269 ms at 844 keypoints, 0.467 vs 0.397 at 1.3k, 0.81 vs 0.66 at 1.9k,
// 1.13 vs 0.85 at 2.4k; until the end of round 3 the value was 3072 -- the scan's start-up and slow path were heavier
// then and narrow frames tied up to 2.6k) -- and from fewer when the frame is WIDE: when its keypoints spread over
// more than 1000 px, most pairs carry the jump penalty, the spatial scan sees them through a halved bound, and it wins
// from ~800 keypoints on (1080p: 0.204 vs 0.212 ms per 512 frames of 835 keypoints, 0.26 vs 0.35 at 1.3k).
constexpr int MATCH_MFMA_MIN_KEYPOINTS = 1024;
constexpr int MATCH_MFMA_MIN_KEYPOINTS_WIDE = 768;
constexpr float MATCH_MFMA_WIDE_DIAGONAL = 1000.0f;
constexpr int NBR_CAP = 16;            // cached earlier-ranked neighbours per candidate
constexpr uint8_t NBR_OVERFLOW = 255;  // more than NBR_CAP: re-enumerate every round

// per-frame status bits
constexpr int STATUS_CAND_OVERFLOW = 1;
constexpr int STATUS_KP_OVERFLOW = 2;
constexpr int STATUS_SEG_OVERFLOW = 4;

// FAST tile geometry, shared by the detector (fast.hip) and the buffers sized for it (slamx_api.hip)
constexpr int FAST_TILE_W = 256;
constexpr int FAST_TILE_H = 32;
constexpr int CORNER_LIST_CAP = 1024;

// Device view of a frame batch in HBM.  All per-frame arrays are [frame][capacity].
struct BatchView {
    const uint8_t* frames;   // [F][rows][pitch] uint8, pitch % 64 == 0
    uint8_t* blurred;        // same layout
    int rows, cols, pitch;
    size_t frame_stride;     // bytes between frames
    int nw;                  // 64-pixel words per row = ceil(cols/64)
    unsigned long long* masks;  // [F][rows*nw]  FAST corner bitmap, bit l of word w = pixel 64w+l
    uint32_t* wordoff;       // [F][rows*nw]  exclusive prefix of popcount(masks) in raster order
    uint8_t* wordcnt;        // [F][rows*nw]  popcount(masks), written with the bitmap: all the prefix scan reads
    uint32_t* tile_count;    // [F][fast_tiles]  corners FAST found in each of its 256 x 32 tiles
    uint32_t* tile_list;     // [F][fast_tiles][CORNER_LIST_CAP]  score<<16 | row-in-tile<<8 | column-in-tile;
                             // not written for a tile whose count exceeds the capacity
    int fast_tiles;          // tiles per frame
    int* n_cand;             // [F]  M
    uint32_t* cand_xy;       // [F][maxM]  y<<16|x, raster order
    uint32_t* keys;          // [F][maxM]  score<<20|raster idx; permuted in place by the sort
    uint32_t* sorted;        // [F][maxM]  keys in final std::sort order
    uint32_t* rank_of;       // [F][maxM]  raster idx -> sorted position
    uint8_t* state;          // [F][maxM]  NMS state by raster index
    uint8_t* nbr_cnt;        // [F][maxM]  number of cached earlier neighbours (NBR_OVERFLOW = too many)
    uint32_t* nbr;           // [F][maxM][NBR_CAP] raster indices of the earlier-ranked neighbours in radius
    uint8_t* segstart;       // [F][segstart_stride] leaf-start bitmask scratch (used when it does not fit LDS)
    size_t segstart_stride;  // bytes per frame, multiple of 4, >= (maxM+32)/8
    int2* seglist;           // [F][3][maxSeg] two level lists + the list of LDS-sized segments
    slamx_keypoint* kps;     // [F][maxK]
    int* n_kp;               // [F]
    uint8_t* desc;           // [F][maxK][width]
    uint32_t* visit;         // [F][maxK] keypoint slots in raster order (written by the NMS finisher)
    int* region_first;       // [F][regions + 1] first entry of every 240 x 64-pixel region in region_list (describe.hip), or nullptr
    int2* region_list;       // [F][maxK] {keypoint slot, y << 16 | x} grouped by region
    int nbands;              // ceil(rows / 64): regions = ceil(cols / 240) * nbands
    int2* moments;           // [F][maxK] {m01, m10} scratch of the describe stage
    float2* cos_sin;         // [F][maxK] {cos, sin} scratch of the describe stage
    int* status;             // [F]
    int maxM, maxK, maxSeg, width;
};

struct DetectorParams {
    int threshold, arc, nms, window, patch, npairs /*NumBRIEFPairs*/, pattern_len;
    const uint2* moment_weights;  // device table of moment_weight_table() or nullptr (radius > 15)
};

// Orientation moments over dword-aligned tiles (describe.hip, moments_kernel): a keypoint's disc sits
// in a 31-row x 36-byte tile whose first column is (x - 15) & ~3; lane l of a wave owns the tile
// dwords e = l + 64 k (k < 5, e = row * 9 + dword column).  Entry ((a * 5 + k) * 64 + l) of the
// table holds, for alignment a = (x - 15) & 3, the four per-byte weights u + 15 (.x) and the four
// disc-membership flags (.y) of that dword, zero outside the disc of the given radius.
// Behind those 4 * 5 * 64 entries: the same pair for the 31-row x 8-dword patch that starts exactly at
// column x - 15 (entry k * 64 + l = patch dword l + 64 k), used by describe_fused_kernel.
constexpr int MOMENT_ALIGNED_SLOTS = 4;
constexpr int MOMENT_TABLE_ENTRIES = 4 * 5 * 64 + MOMENT_ALIGNED_SLOTS * 64;
void moment_weight_table(int radius, uint32_t* out_xy /* 2 * MOMENT_TABLE_ENTRIES words */);

// Correctly rounded float sqrt.  NOTE: HIP's __fsqrt_rn() is __ocml_native_sqrt_f32 (1 ulp,
// v_sqrt_f32 alone) unless OCML_BASIC_ROUNDED_OPERATIONS is defined; plain sqrtf is expanded to the
// IEEE-correct sequence under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt, which is
// what the reference's std::sqrt(float) needs.
#if defined(__HIPCC__)
__device__ __forceinline__ float sqrtf_rn(float x) { return __builtin_sqrtf(x); }

// findBestMatchesHamming's jump-radius penalty, src/frontend/feature_matcher.cpp:161-170, with the
// reference's float operations (MAX_JUMP_RADIUS = 500 converted to float in both places).
__device__ __forceinline__ int penalised_distance_exact(int dist, float dx, float dy) {
    const float d = sqrtf_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
    if (d > 500.0f) {
        const float penalty = __fadd_rn(1.0f, __fdiv_rn(d, 500.0f));
        dist = static_cast<int>(__fmul_rn(static_cast<float>(dist), penalty));
    }
    return dist;
}
// Same value, usually without the correctly-rounded sqrt and divide (~40 instructions): evaluate
// the expression with the 1-ulp hardware sqrt and a multiply by 0.002f (total relative error
// < 1e-6), and accept its integer part only when the value is farther than 4e-6*x (+1e-6) from
// an integer boundary and d is farther than 0.01 from the 500 px threshold; otherwise (a few in
// 1e5 pairs) fall back to the exact expression.  The result is ALWAYS the reference's value.
__device__ __forceinline__ int penalised_distance(int dist, float qx, float qy, float tx, float ty) {
    const float dx = __fsub_rn(qx, tx);
    const float dy = __fsub_rn(qy, ty);
    const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
    const float da = __builtin_amdgcn_sqrtf(d2);  // v_sqrt_f32, 1 ulp
    if (fabsf(da - 500.0f) < 0.01f || !(da < 3.0e37f)) {
        return penalised_distance_exact(dist, dx, dy);
    }
    if (da <= 500.0f || dist == 0) {
        return dist;
    }
    const float x = static_cast<float>(dist) * (1.0f + da * 0.002f);
    const int k = static_cast<int>(x);
    const float fr = x - static_cast<float>(k);
    const float tol = x * 4.0e-6f + 1.0e-6f;
    if (fr > tol && fr < 1.0f - tol && x < 1.0e9f) {
        return k;
    }
    return penalised_distance_exact(dist, dx, dy);
}
#endif

// hipFuncSetAttribute(hipFuncAttributeMaxDynamicSharedMemorySize) applies to the CURRENT device, so "once per process" is
// wrong for a host that drives several GPUs from one process (tools/cli -g N: one thread + one context per GPU): once per
// (call site, device).  `done` is the call site's bit mask of devices that have been set up.
inline void allow_dynamic_lds(const void* kernel, size_t bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
        return;
    }
    const unsigned long long bit = 1ull << dev;
    if ((done.load(std::memory_order_relaxed) & bit) == 0) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
        done.fetch_or(bit, std::memory_order_relaxed);
    }
}

// blocks per frame for a grid-stride kernel: enough to cover `items`, capped so that a large
// batch does not launch millions of empty workgroups (>= 8 per frame, ~8192 in total)
inline int grid_x(int items, int per_block, int nframes) {
    int blocks = (items + per_block - 1) / per_block;
    int cap = 8192 / (nframes > 0 ? nframes : 1);
    if (cap < 8) {
        cap = 8;
    }
    if (blocks > cap) {
        blocks = cap;
    }
    return blocks < 1 ? 1 : blocks;
}

// 3-D grid -> (frame, block-in-frame).  The hardware deals consecutive workgroup ids (x fastest) round-robin
// over the 8 XCDs, so a grid whose x extent is a multiple of 8 pins block column k to XCD k
// (uneven columns then idle whole XCDs).  With 8 or more frames all blocks of a frame land on ONE
// XCD, frame after frame: its 4 MB L2 then only has to hold the active row bands of the few frames
// in flight, and neighbouring tiles / keypoint patches of a frame find each other's lines there.
// With fewer frames the blocks are simply dealt round-robin.
// Placement is a speed matter only.
inline dim3 frame_grid(int nframes, int blocks_per_frame) {
    // x: the frame's slot among the 8 XCDs (workgroups are dealt to XCDs round robin in x-fastest order),
    // y: block of the frame, z: group of 8 frames -- the kernel reads all three off blockIdx, no division.
    // More blocks per frame than grid.y holds (frames beyond ~500 Mpixel) spill into z.
    // (SLAMX_GRID_Y_LIMIT: a test hook that lowers the limit, so that ordinary frames exercise the spill into z)
    static const int y_limit = [] {
        const char* e = std::getenv("SLAMX_GRID_Y_LIMIT");
        const int v = e != nullptr ? std::atoi(e) : 0;
        return v >= 1 && v <= 65535 ? v : 65535;
    }();
    const unsigned y = static_cast<unsigned>(blocks_per_frame < y_limit ? (blocks_per_frame > 0 ? blocks_per_frame : 1) : y_limit);
    const unsigned parts = (static_cast<unsigned>(blocks_per_frame) + y - 1) / y;
    const unsigned groups = nframes >= 8 ? static_cast<unsigned>((nframes + 7) / 8) : 1u;
    return dim3(nframes >= 8 ? 8u : static_cast<unsigned>(nframes), y, groups * (parts > 0 ? parts : 1u));
}
#if defined(__HIPCC__)
__device__ __forceinline__ bool frame_block(int nframes, int blocks_per_frame, int* f, int* bx) {
    unsigned group = blockIdx.z, part = 0;
    if (gridDim.y < static_cast<unsigned>(blocks_per_frame)) {
        const unsigned parts = (static_cast<unsigned>(blocks_per_frame) + gridDim.y - 1) / gridDim.y;
        group = blockIdx.z / parts;
        part = blockIdx.z - group * parts;
    }
    *f = static_cast<int>(blockIdx.x + 8 * group);  // an XCD takes its frames one after another
    *bx = static_cast<int>(blockIdx.y + part * gridDim.y);
    return *f < nframes && *bx < blocks_per_frame;
}
#endif

// ---- launchers (each enqueues on `stream`, no synchronization) ----
void launch_fast_detect(const BatchView& v, const DetectorParams& p, int nframes, hipStream_t stream);
void launch_scan(const BatchView& v, int nframes, hipStream_t stream);
void launch_expand(const BatchView& v, const DetectorParams& p, int nframes, hipStream_t stream);
// need_rank_of: also write the inverse permutation (only the generic NMS path reads it)
void launch_sort(const BatchView& v, int nframes, hipStream_t stream, bool need_rank_of);
// true when launch_nms will take the greedy one-wave-per-frame path (which needs no rank_of)
bool nms_uses_greedy_path(const BatchView& v, const DetectorParams& p);
// ev_mid (optional) is recorded after the first kernel of a multi-kernel stage
void launch_nms(const BatchView& v, const DetectorParams& p, int nframes, hipStream_t stream, hipEvent_t ev_mid);
void launch_passthrough_keypoints(const BatchView& v, int nframes, hipStream_t stream);  // NonMaxSuppression: 0
void launch_blur(const BatchView& v, const double* weights_dev, int nframes, hipStream_t stream);
// use_visit: walk the keypoints in the raster order left by launch_nms (v.visit) instead of slot order
// allow_tiles: false when the caller KNOWS that no frame is dense enough for the shared region tiles (their two launches
// are then skipped; the result is the same either way)
void launch_describe(const BatchView& v, const DetectorParams& p, const int8_t* pattern_dev, int nframes, bool use_visit,
                     hipStream_t stream, hipEvent_t ev_after_moments, hipEvent_t ev_after_trig, bool allow_tiles = true);
// keypoints per frame from which describe.hip sends a frame through the shared region tiles (0x7fffffff: never)
int describe_tile_min_keypoints(const BatchView& v);

struct MatchView {
    // pair q: query = frame qf[q], train = frame tf[q] of the same batch layout
    const uint8_t* desc;           // [F][maxK][width]
    const slamx_keypoint* kps;     // [F][maxK]  (or nullptr: no jump penalty)
    const int* n_kp;               // [F]
    slamx_match* matches;          // [F][maxK]  matches of query frame f (vs f-1)
    int* n_matches;                // [F]
    int2* best;                    // [F][nsplit][maxK] scratch: {best_dist, best_idx} per train chunk
    int* second;                   // [F][nsplit][maxK] scratch: second best per train chunk
    int nsplit;                    // train set split into this many chunks (more workgroups)
    int maxK, width;
    int live_words;  // 32-bit words that can be non-zero in any descriptor (0 = unknown: all)
    int live_bits;   // highest bit that can be set in any descriptor, plus one (0 = unknown)
    uint8_t* xdesc;  // scratch of the matrix-core path (match_scratch_bytes(): expanded descriptors in SPATIAL order and
                     // the tables below, carved out of the one allocation by match_scratch_carve()), or nullptr
    uint4* sbits;    // [F][maxKp]  {compact bits (first two words), original index, 0} of the keypoint at each sorted position
    float4* sxyi;    // [F][maxKp]  {x, y, original index (int bits), 0} of the keypoint at each sorted position
    float4* sbox;    // [F][maxKp / 64]  bounding box {min x, min y, max x, max y} of every stage (64 consecutive sorted
                     // positions); min x = +inf when a coordinate of the stage is not finite or beyond +-1e6
    int maxKp;       // sorted arrays' stride: maxK rounded up to 64, plus one padding stage
    int* mfma_flag;  // [F]  1 = pairs with this frame as TRAIN set take the matrix-core scan (scan_choice_kernel decides)
    int mfma_min_wide;   // ... or at least this many when the frame's keypoints spread over MATCH_MFMA_WIDE_DIAGONAL px
    int mfma_min_train;  // with xdesc: pairs whose TRAIN set has at least this many keypoints take the matrix-core
                         // scan, the others the popcount scan (both kernels are launched, each skips the other's pairs)
    int use_ratio;
    float ratio;
};
// query frames first_query_frame .. first_query_frame+npairs-1; train = query-1, except that
// query frame 0 uses first_train_frame.  ev_mid (optional) is recorded between the all-pairs
// kernel and the emit kernel.
inline int match_maxKp(int maxK) { return ((maxK + 63) / 64 + 1) * 64; }
inline size_t match_scratch_bytes(size_t frames, int maxK) {
    const size_t kp = static_cast<size_t>(match_maxKp(maxK));
    return frames * kp * (64 + sizeof(uint4) + sizeof(float4)) + frames * (kp / 64) * sizeof(float4) + frames * sizeof(int) + 256;
}
inline void match_scratch_carve(uint8_t* blob, size_t frames, int maxK, MatchView* mv) {
    const size_t kp = static_cast<size_t>(match_maxKp(maxK));
    mv->maxKp = static_cast<int>(kp);
    mv->xdesc = blob;
    mv->sxyi = blob == nullptr ? nullptr : reinterpret_cast<float4*>(blob + frames * kp * 64);
    mv->sbits = blob == nullptr ? nullptr : reinterpret_cast<uint4*>(blob + frames * kp * (64 + sizeof(float4)));
    mv->sbox = blob == nullptr ? nullptr : reinterpret_cast<float4*>(blob + frames * kp * (64 + sizeof(float4) + sizeof(uint4)));
    mv->mfma_flag = blob == nullptr ? nullptr : reinterpret_cast<int*>(blob + frames * kp * (64 + sizeof(float4) + sizeof(uint4)) +
                                                                       frames * (kp / 64) * sizeof(float4));
}
void launch_match_consecutive(const MatchView& mv, int first_query_frame, int npairs, int first_train_frame,
                              hipStream_t stream, hipEvent_t ev_mid);

void launch_pnp_dlt(const double* p3d, const double* p2d, int n_points, int n_hyp, double* R, double* t, double* nullvec,
                    hipStream_t stream);
void launch_inliers(const double* p3d, const double* p2d, int n, const double* K, const double* R, const double* t,
                    double threshold, int n_hyp, int* counts, hipStream_t stream);
void launch_recover_pose(const double* E, const double* K, const float* pts1, const float* pts2, int n, double* poses,
                         double* KP, double* KP0, int* front, hipStream_t stream);
void launch_triangulate(const double* P1, const double* P2, const float* pts1, const float* pts2, int n, float* x4,
                        double* xyz, hipStream_t stream);

// ---- the pose stage of a batch (pose.hip; host side in pose_api.hip) ----
// Counter-based sample generator of the batched RANSAC loops: draw d of hypothesis h of frame slot s is
// splitmix64-mix(key(seed, stream) + counter(s, h, d)), reduced to [0, n) by multiply-high; indices are redrawn until
// distinct (loop_closure.cpp:185-193's rule).  Host and device run the same function.
#if defined(__HIPCC__)
#define SLAMX_HD __host__ __device__
#else
#define SLAMX_HD
#endif
SLAMX_HD inline unsigned long long pose_mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// k <= 8 distinct indices out of [0, n), n >= k
SLAMX_HD inline void pose_draw_sample(unsigned seed, int stream, int slot, int hyp, int n, int k, int* out) {
    const unsigned long long key = pose_mix64((static_cast<unsigned long long>(seed) << 1 | 1ull) +
                                              static_cast<unsigned long long>(stream) * 0xD1B54A32D192ED03ull);
    const unsigned long long ctr0 = (static_cast<unsigned long long>(static_cast<unsigned>(slot)) << 40) |
                                    (static_cast<unsigned long long>(static_cast<unsigned>(hyp)) << 16);
    int have = 0;
    for (unsigned draw = 0; have < k && draw < 4096u; ++draw) {
        const unsigned long long z = pose_mix64(key + (ctr0 | draw) * 0x9E3779B97F4A7C15ull);
        const int c = static_cast<int>(((z >> 32) * static_cast<unsigned long long>(static_cast<unsigned>(n))) >> 32);
        bool seen = false;
        for (int j = 0; j < have; ++j) {
            seen = seen || out[j] == c;
        }
        if (!seen) {
            out[have++] = c;
        }
    }
    for (int c = 0; have < k && c < n; ++c) {  // (4096 draws without k distinct values: never in practice; still total)
        bool seen = false;
        for (int j = 0; j < have; ++j) {
            seen = seen || out[j] == c;
        }
        if (!seen) {
            out[have++] = c;
        }
    }
}

// (refit on the current model's inliers, keep it iff its MSAC cost is not higher) rounds after the RANSAC winner is known;
// oracle/pose_oracle.cpp restates the same number.  One: with a 1 px threshold nearly every good sample already flags
// the complete inlier set, and what limits the refit afterwards are the few wrong matches that lie within the threshold
// of the epipolar lines, which further rounds keep (measured on bench.py's scene with 3 rounds: median translation
// direction error 0.80 -> 0.70 deg for + 0.6 ms per 256 pairs; the cost rule alone brought it from 1.6)
constexpr int POSE_REFIT_ROUNDS = 1;
struct PoseParams {
    double K[9];
    double fx, fy, cx, cy;
    double thr2;           // (threshold_px / mean focal length)^2: Sampson threshold in normalised units
    double pnp_threshold;  // pixels
    int iterations, pnp_iterations, pnp_min;
    unsigned seed;
    int cap;               // slots per frame pair in every correspondence-parallel array
    int device_sampling;   // 1: the hypothesis kernels draw their samples (and record them in idx / pidx); 0: read them
    int geometry_mode;          // SLAMX_GEOMETRY_REFERENCE / SLAMX_GEOMETRY_CALIBRATED (slamx.h)
};
// Device arrays of the stage; [P] = frame slot (the QUERY frame of the pair), H = iterations, Hp = pnp_iterations.
struct PoseWork {
    int* n;              // [P] correspondences of the pair (matches of the frame, clamped to cap; 0 without a train frame)
    double2* x1;         // [P][cap] K^-1-normalised point in the TRAIN frame ("frame 1" of PoseEstimator::estimate)
    double2* x2;         // [P][cap] ... in the QUERY frame
    float2* p1;          // [P][cap] the same points in pixels (triangulatePoints takes those)
    float2* p2;
    int* idx;            // [P][H][8] sample table
    double* Eh;          // [P][H][9] hypotheses
    int* counts;         // [P][H] Sampson inlier counts
    uint8_t* flags_a;    // [P][cap] inlier flags of the winning hypothesis
    uint8_t* flags_b;    // [P][cap] ... of the refit, then of whichever of the two was kept (the result's mask)
    double* Ebest;       // [P][9]
    double* Efit;        // [P][9]
    int* sel;            // [P][4] winning hypothesis, its count, the refit's count, refit kept
    double* poses4;      // [P][4][12] the four (R, t) candidates of decomposeEssential
    double* KP;          // [P][4][12] K [R|t] of each
    int* front;          // [P][4] cheirality votes
    slamx_pose_result* res;  // [P]
    double* xyz;         // [P][cap][3] map points, parallel to the pair's matches
    float* x4;           // [P][cap][4] homogeneous float points (cv::triangulatePoints' output) or nullptr
    int* m;              // [P] 3D-2D correspondences of the PnP step
    double* c3;          // [P][cap][3]
    double* c2;          // [P][cap][2]
    int* pidx;           // [P][Hp][6]
    double* pRt;         // [P][Hp][12] R row-major, t
    double* pnull;       // [P][Hp][12] the DLT null vector of every hypothesis (before the column-major re-map) or nullptr
    int* pcounts;        // [P][Hp]
};
struct PoseSource {  // where the keypoints and matches of the batch are
    const slamx_keypoint* kps;  // [F][maxK]
    const slamx_match* matches;  // [F][maxK], queryIdx ascending
    const int* n_matches;        // [F]
    int maxK;
};
void launch_pose_gather(const PoseSource& src, const PoseWork& w, const PoseParams& prm, int nframes, int first_train_frame,
                        hipStream_t stream);
void launch_pose_load_points(const PoseWork& w, const PoseParams& prm, hipStream_t stream);  // one pair: p1, p2, n[0] uploaded
void launch_essential_hypotheses(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream);
void launch_sampson_score(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream);
void launch_pose_select(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream);
void launch_essential_refit_pairs(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream);
void launch_pose_final(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream);
void launch_cheirality_pairs(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream);
void launch_triangulate_pairs(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream);
void launch_pnp_gather(const PoseSource& src, const PoseWork& w, const PoseParams& prm, int nframes, int first_train_frame,
                       hipStream_t stream);
void launch_pnp_hypotheses(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream);
void launch_pnp_score(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream);
void launch_pnp_select(const PoseWork& w, const PoseParams& prm, int npairs, hipStream_t stream);

// image pyramid (pyramid.hip): dst level frames from src level frames (tables built by the host), and the
// append of a level's results, scaled to level-0 coordinates, to level 0's keypoint / descriptor arrays
void launch_downscale(const BatchView& src, const BatchView& dst, const uint32_t* xtab, const uint32_t* ytab, int nframes,
                      hipStream_t stream);
size_t packed_bound_bytes(int nframes, int maxK, int width, bool with_matches);
void launch_pack_results(const BatchView& v, const slamx_match* matches, const int* n_matches, int nframes, uint8_t* pack,
                         hipStream_t stream, uint8_t* mirror = nullptr, size_t mirror_bytes = 0);
void launch_merge_level(const BatchView& src, const BatchView& dst, float scale, int nframes, hipStream_t stream);

// sliding-window Gauss-Newton (gn.hip): device pointers of a batch of windows
struct GnLaunch {
    const double* poses;     // [n_windows][n_cams][12]  linearisation point
    double* landmarks;       // [n_windows][max_landmarks][3]  updated in place by the back-substitution
    const float* obs;        // [n_windows][max_landmarks][n_cams][2]
    const int* n_landmarks;  // [n_windows]
    double* partial;         // [n_windows][nslices][15][256]  per-wave accumulator tiles
    double* delta;           // [n_windows][64]  camera step
    double* cost;            // [n_windows]
    int* status;             // [n_windows]
    int n_windows, n_cams, max_landmarks, nslices, n_fixed;
    double K[4];             // fx fy cx cy
    double lambda;
};
void launch_gn_accumulate(const GnLaunch& a, hipStream_t stream);
// writes delta, cost, status and (unless cost_only) the stepped poses into poses_next
void launch_gn_solve(const GnLaunch& a, double* poses_next, int cost_only, double* S_out, double* g_out, hipStream_t stream);
void launch_gn_backsub(const GnLaunch& a, hipStream_t stream);

void launch_undistort(const uint8_t* img, int rows, int cols, int pitch, const slamx_camera& cam, double* out,
                      hipStream_t stream);

void launch_debug_sort(uint32_t* keys, uint32_t* sorted, uint8_t* segstart, int2* seglist, int maxSeg, int n,
                       int* status, hipStream_t stream);
// the two-launch form of the sort (sort_split_kernel + sort_pieces_kernel) on one array: n_cand = device int holding n,
// rank_of = n words of scratch
void launch_debug_sort_split(uint32_t* keys, uint32_t* sorted, uint8_t* segstart, int2* seglist, int maxSeg, int n, int* status,
                             int* n_cand, uint32_t* rank_of, int groups, hipStream_t stream);
void launch_debug_penalty(const int* dist, const float* qx, const float* qy, const float* tx, const float* ty, int n,
                          int* out, hipStream_t stream);
void launch_debug_trig(const float* m01, const float* m10, int n, float* deg, float* c, float* s, hipStream_t stream);

}  // namespace slamx
