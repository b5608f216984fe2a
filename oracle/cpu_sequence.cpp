// cpu_sequence.cpp -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (part of the oracle; see slam_oracle.cpp's header).
//
// The "restated reference CPU path" of SURVEY.md 8d run over a SEQUENCE of frames on T host threads:
// detectAndCompute of every frame (src/frontend/feature_detector.cpp:49-54) and match(frame k, frame k-1) with
// keypoints (src/frontend/feature_matcher.cpp:71-95, as test/frontend/test_feature_matcher.cpp:42-62 calls them).
// The reference has NO threading on this path (no #pragma omp, no std::thread use: SURVEY.md section 5); the
// all-cores number SURVEY.md 8d asks for is obtained the only way the path allows -- frames are independent, so
// contiguous chunks of frames go to std::threads, each of which first recomputes the predecessor of its chunk's
// first frame (a 1-frame halo, not counted).  T = 1 is the reference's own execution model.
//
// Only bench.py's cpu_baseline leg and tests/ call this.  Nothing here is restated arithmetic: it calls the
// extern "C" entry points of slam_oracle.cpp (oracle_detect, oracle_compute, oracle_match).
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {
struct Keypoint {
    float x, y, size, angle, response;
};
struct Match {
    int queryIdx, trainIdx;
    float distance;
};
struct DetectorParams {
    int intensityThreshold, contiguousPixelsThreshold, nonMaxSuppression, suppressionWindowSize, patchSize, numBRIEFPairs;
};
struct MatcherParams {
    int filterMatches, goodMatchesCount, useRatioTest;
    float ratioTestThreshold;
};
}  // namespace

extern "C" {
int oracle_detect(const uint8_t* img, int rows, int cols, int stride, const DetectorParams* P, Keypoint* kps, int cap);
void oracle_compute(const uint8_t* img, int rows, int cols, int stride, const DetectorParams* P, Keypoint* kps, int n,
                    uint8_t* desc, float* moments);
int oracle_match(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int width, const Keypoint* kp1, int nkp1,
                 const Keypoint* kp2, int nkp2, const MatcherParams* P, Match* out, int cap);
}

namespace {

struct Features {
    std::vector<Keypoint> kps;
    std::vector<uint8_t> desc;
};

void detect_and_compute(const uint8_t* img, int rows, int cols, const DetectorParams& P, Features* f) {
    f->kps.resize(4096);
    int n = oracle_detect(img, rows, cols, cols, &P, f->kps.data(), static_cast<int>(f->kps.size()));
    if (n > static_cast<int>(f->kps.size())) {
        f->kps.resize(static_cast<size_t>(n));
        n = oracle_detect(img, rows, cols, cols, &P, f->kps.data(), n);
    }
    f->kps.resize(static_cast<size_t>(n));
    const int width = P.numBRIEFPairs / 8;
    f->desc.assign(static_cast<size_t>(n) * width, 0);
    oracle_compute(img, rows, cols, cols, &P, f->kps.data(), n, f->desc.data(), nullptr);
}

uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    for (size_t i = 0; i < n; ++i) {
        h = (h ^ b[i]) * 0x100000001B3ULL;
    }
    return h;
}

}  // namespace

extern "C" {

// frames: [nframes][rows][cols] dense.  Frame k is matched against frame k-1, frame 0 against frame nframes-1 (the
// convention of bench.py's step).  per_frame (optional, [nframes][3]) receives keypoints, matches and an FNV-1a hash of
// the frame's keypoints + descriptors + matches: identical for every thread count.  reps >= 1: every thread goes over
// its chunk that many times (a long enough timed region on many cores without holding thousands of frames in memory;
// nframes * reps frames are processed, the halo once per thread).  Returns the wall seconds of the threaded region
// (thread start to last join).
double oracle_sequence_mt(const uint8_t* frames, int nframes, int rows, int cols, const DetectorParams* P,
                          const MatcherParams* M, int use_keypoints, int n_threads, int reps, uint64_t* per_frame) {
    if (nframes <= 0) {
        return 0.0;
    }
    if (n_threads < 1) {
        n_threads = 1;
    }
    if (n_threads > nframes) {
        n_threads = nframes;
    }
    const size_t fsz = static_cast<size_t>(rows) * cols;
    const int width = P->numBRIEFPairs / 8;
    auto work = [&](int first, int last) {
        Features prev, cur;
        detect_and_compute(frames + fsz * static_cast<size_t>((first + nframes - 1) % nframes), rows, cols, *P, &prev);  // halo
        std::vector<Match> m;
        for (int rep = 0; rep < (reps < 1 ? 1 : reps); ++rep)
        for (int k = first; k < last; ++k) {
            detect_and_compute(frames + fsz * static_cast<size_t>(k), rows, cols, *P, &cur);
            const int n1 = static_cast<int>(cur.kps.size()), n2 = static_cast<int>(prev.kps.size());
            m.resize(static_cast<size_t>(n1 > 0 ? n1 : 1));
            int nm = 0;
            if (n1 > 0 && n2 > 0) {
                nm = oracle_match(cur.desc.data(), n1, prev.desc.data(), n2, width, use_keypoints ? cur.kps.data() : nullptr,
                                  use_keypoints ? n1 : 0, use_keypoints ? prev.kps.data() : nullptr, use_keypoints ? n2 : 0, M,
                                  m.data(), static_cast<int>(m.size()));
            }
            if (per_frame != nullptr && rep == 0) {  // later passes pair a chunk's first frame with its own last
                uint64_t h = 0xCBF29CE484222325ULL;
                h = fnv(h, cur.kps.data(), cur.kps.size() * sizeof(Keypoint));
                h = fnv(h, cur.desc.data(), cur.desc.size());
                h = fnv(h, m.data(), static_cast<size_t>(nm) * sizeof(Match));
                per_frame[3 * k] = static_cast<uint64_t>(n1);
                per_frame[3 * k + 1] = static_cast<uint64_t>(nm);
                per_frame[3 * k + 2] = h;
            }
            std::swap(prev, cur);
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    if (n_threads == 1) {
        work(0, nframes);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) {
            const int first = static_cast<int>(static_cast<long long>(nframes) * t / n_threads);
            const int last = static_cast<int>(static_cast<long long>(nframes) * (t + 1) / n_threads);
            pool.emplace_back(work, first, last);
        }
        for (auto& th : pool) {
            th.join();
        }
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
