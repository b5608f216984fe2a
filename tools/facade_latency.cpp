// Per-call latency of the reference-signature API: what the reference's own programs print
// (test/frontend/test_feature_detector.cpp:34-50 times detect and compute, test_feature_matcher.cpp:41-68 times
// detectAndCompute and match, once each, cold) measured properly -- a few warm-up calls, then the median, minimum and
// maximum of `reps` calls of slam::FeatureDetector::detect / compute / detectAndCompute and slam::FeatureMatcher::match
// on the reference's own frames, through the header-only facade over the C ABI, host buffers in, host buffers out.
// bench.py runs it and prints the numbers as `facade_latency`, next to the CPU restatement's time for the same calls.
//   facade_latency <feature_detector.yml> <feature_matcher.yml> <img1.png> <img2.png> [reps]
// Prints one JSON object.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <slam/frontend/feature_detector.hpp>
#include <slam/frontend/feature_matcher.hpp>
#include <slam/preprocessing/png_gray.hpp>

namespace {
struct Stat {
    double median, min, max;
};
template <typename F>
Stat time_us(int reps, F&& f) {
    for (int i = 0; i < 3; ++i) {
        f();  // warm-up: lazily created batches and scratch, first-launch code upload
    }
    std::vector<double> us(static_cast<size_t>(reps));
    for (int i = 0; i < reps; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        f();
        us[static_cast<size_t>(i)] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    std::sort(us.begin(), us.end());
    return {us[us.size() / 2], us.front(), us.back()};
}
void print(const char* name, const Stat& s, bool last = false) {
    std::printf("\"%s\": {\"median_us\": %.1f, \"min_us\": %.1f, \"max_us\": %.1f}%s", name, s.median, s.min, s.max, last ? "" : ", ");
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s <feature_detector.yml> <feature_matcher.yml> <img1.png> <img2.png> [reps]\n", argv[0]);
        return -1;
    }
    try {
        const int reps = argc > 5 ? std::max(3, std::atoi(argv[5])) : 50;
        slam::FeatureDetector detector(argv[1]);
        slam::FeatureMatcher matcher(argv[2]);
        const slam::EigenGrayMatrix image1 = slamx_tools::read_png_gray(argv[3]);
        const slam::EigenGrayMatrix image2 = slamx_tools::read_png_gray(argv[4]);
        std::vector<slam::Keypoint> kp1, kp2, kps;
        slam::DescriptorMatrix d1, d2, desc;
        std::vector<slam::Match> matches;
        detector.detectAndCompute(image1, kp1, d1);
        detector.detectAndCompute(image2, kp2, d2);
        const Stat detect = time_us(reps, [&] { detector.detect(image1, kps); });
        const Stat compute = time_us(reps, [&] { detector.compute(image1, kps, desc); });
        const Stat dac = time_us(reps, [&] { detector.detectAndCompute(image1, kps, desc); });
        const Stat match = time_us(reps, [&] { matcher.match(d1, d2, matches, kp1, kp2); });
        const Stat match_nokp = time_us(reps, [&] { matcher.match(d1, d2, matches); });
        std::printf("{\"rows\": %td, \"cols\": %td, \"keypoints\": [%zu, %zu], \"matches\": %zu, \"reps\": %d, ", image1.rows(), image1.cols(),
                    kp1.size(), kp2.size(), matches.size(), reps);
        print("detect", detect);
        print("compute", compute);
        print("detectAndCompute", dac);
        print("match", match);
        print("match_without_keypoints", match_nokp, true);
        std::printf("}\n");
    } catch (const std::exception& e) {
        std::fprintf(stderr, "An error occurred: %s\n", e.what());
        return -1;
    }
    return 0;
}
