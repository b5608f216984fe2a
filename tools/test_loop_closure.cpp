// test_loop_closure -- mirrors the reference's test/backend/test_loop_closure.cpp:19-109: build a
// keyframe database from an image sequence (map points = (x, y, 1) of the keypoints, as the
// reference's test fakes them), then ask whether the last frame closes a loop.
//   test_loop_closure <loop_closure.yml> <feature_detector.yml> <feature_matcher.yml> <camera.yml> <image dir> <n images> [seed]
// Prints the voting score of every keyframe and the verdict; exit code 0 like the reference's test.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <slam/backend/loop_closure.hpp>
#include <slam/common/camera.hpp>
#include <slam/frontend/feature_detector.hpp>
#include <slam/frontend/feature_matcher.hpp>
#include <slam/preprocessing/png_gray.hpp>

int main(int argc, char** argv) {
    if (argc < 7) {
        std::fprintf(stderr, "usage: test_loop_closure <loop.yml> <detector.yml> <matcher.yml> <camera.yml> <image dir> <n> [seed]\n");
        return -1;
    }
    try {
        slam::Camera camera(argv[4]);
        slam::FeatureDetector featureDetector(argv[2]);
        slam::FeatureMatcher matcher(argv[3]);
        const uint32_t seed = argc > 7 ? static_cast<uint32_t>(std::strtoul(argv[7], nullptr, 10)) : 0u;
        slam::LoopClosure detector("", argv[1], matcher, seed);
        std::vector<slam::Keypoint> lastKeypoints;
        slam::DescriptorMatrix lastDescriptors;
        const int nImages = std::atoi(argv[6]);
        for (int i = 0; i < nImages; ++i) {
            const std::string imagePath = std::string(argv[5]) + "/" + std::to_string(i) + ".png";
            const slam::EigenGrayMatrix eigenImage = slamx_tools::read_png_gray(imagePath);
            std::vector<slam::Keypoint> kpsSlam;
            slam::DescriptorMatrix descSlam;
            featureDetector.detectAndCompute(eigenImage, kpsSlam, descSlam);
            std::vector<slam::Vec3d> mapPoints;
            mapPoints.reserve(kpsSlam.size());
            for (const auto& kp : kpsSlam) {
                mapPoints.push_back({static_cast<double>(kp.x), static_cast<double>(kp.y), 1.0});
            }
            detector.addKeyframe(i, descSlam, kpsSlam, mapPoints);
            if (i == nImages - 1) {
                lastKeypoints = kpsSlam;
                lastDescriptors = descSlam;
            }
            std::printf("keyframe %d: %zu keypoints\n", i, kpsSlam.size());
        }
        const auto result = detector.detect(lastDescriptors, lastKeypoints, camera);
        if (result) {
            std::printf("loop with keyframe %d\n", result->matchedKeyframeId);
            for (int r = 0; r < 4; ++r) {
                std::printf("T %.12g %.12g %.12g %.12g\n", result->relativeTransform[static_cast<size_t>(r * 4)],
                            result->relativeTransform[static_cast<size_t>(r * 4 + 1)],
                            result->relativeTransform[static_cast<size_t>(r * 4 + 2)],
                            result->relativeTransform[static_cast<size_t>(r * 4 + 3)]);
            }
        } else {
            std::printf("no loop\n");
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "An unhandled exception occurred: %s\n", e.what());
        return 0;
    }
    return 0;
}
