// Counterpart of the reference's test/frontend/test_feature_matcher.cpp (:14-135): two frames ->
// detectAndCompute x2 -> match WITH keypoints, distance statistics; dumps the matches.
// Usage: test_feature_matcher <feature_detector.yml> <feature_matcher.yml> <img1.png> <img2.png> <out.bin>
#include <cstdio>
#include <fstream>
#include <limits>

#include <slam/frontend/feature_detector.hpp>
#include <slam/frontend/feature_matcher.hpp>

#include <slam/preprocessing/png_gray.hpp>

int main(int argc, char** argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s <detector.yml> <matcher.yml> <img1.png> <img2.png> <out.bin>\n", argv[0]);
        return -1;
    }
    try {
        slam::FeatureDetector featureDetector(argv[1]);
        slam::FeatureMatcher featureMatcher(argv[2]);
        slam::EigenGrayMatrix eigenImage1 = slamx_tools::read_png_gray(argv[3]);
        slam::EigenGrayMatrix eigenImage2 = slamx_tools::read_png_gray(argv[4]);

        std::vector<slam::Keypoint> keypoints1, keypoints2;
        slam::DescriptorMatrix descriptors1, descriptors2;
        featureDetector.detectAndCompute(eigenImage1, keypoints1, descriptors1);
        featureDetector.detectAndCompute(eigenImage2, keypoints2, descriptors2);
        std::printf("Image 1: Detected %zu keypoints with %td descriptors\n", keypoints1.size(), descriptors1.rows());
        std::printf("Image 2: Detected %zu keypoints with %td descriptors\n", keypoints2.size(), descriptors2.rows());

        std::vector<slam::Match> matches;
        featureMatcher.match(descriptors1, descriptors2, matches, keypoints1, keypoints2);
        std::printf("Found %zu matches\n", matches.size());

        std::vector<slam::Match> matchesNoKp;  // test_pose_estimator.cpp:138 calls match without keypoints
        featureMatcher.match(descriptors1, descriptors2, matchesNoKp);

        if (!matches.empty()) {
            float minDist = std::numeric_limits<float>::max();
            float maxDist = 0.0F;
            float avgDist = 0.0F;
            for (const auto& match : matches) {
                minDist = std::min(minDist, match.distance);
                maxDist = std::max(maxDist, match.distance);
                avgDist += match.distance;
            }
            avgDist /= static_cast<float>(matches.size());
            std::printf("Match distances: min=%.2f, max=%.2f, avg=%.2f\n", minDist, maxDist, avgDist);
        }

        // exception behaviour of validateInputs (feature_matcher.cpp:97-111)
        int exceptions = 0;
        try {
            slam::DescriptorMatrix empty;
            featureMatcher.match(empty, descriptors2, matches);
        } catch (const std::invalid_argument&) {
            exceptions |= 1;
        }
        try {
            slam::DescriptorMatrix narrow(4, 16);
            std::vector<slam::Match> tmp;
            featureMatcher.match(narrow, descriptors2, tmp);
        } catch (const std::runtime_error&) {
            exceptions |= 2;
        }
        featureMatcher.match(descriptors1, descriptors2, matches, keypoints1, keypoints2);

        std::ofstream out(argv[5], std::ios::binary);
        const int32_t ex = exceptions;
        out.write(reinterpret_cast<const char*>(&ex), 4);
        for (const auto* m : {&matches, &matchesNoKp}) {
            const int32_t n = static_cast<int32_t>(m->size());
            out.write(reinterpret_cast<const char*>(&n), 4);
            out.write(reinterpret_cast<const char*>(m->data()), sizeof(slam::Match) * m->size());
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "An error occurred: %s\n", e.what());
        return -1;
    }
    return 0;
}
