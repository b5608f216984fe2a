// cli -- keeps the reference's flag surface (tools/cli/cli.cpp:20: getopt "hv:c:") but, unlike the
// reference's stub SLAMModel (src/model/model.cpp:5-7 is empty, run() is never defined), actually
// drives the hot path: frame directory -> detectAndCompute -> match(frame k, frame k-1) and, when the
// config directory also holds a camera.yml, PoseEstimator::estimate + triangulatePoints on the matches.
//   cli -c <config dir with feature_detector.yml + feature_matcher.yml [+ camera.yml]> -v <directory of PNG frames> [-p <levels>]
// -p (an extension: the reference has no pyramid) runs the detector on an image pyramid of that many levels.
#include <getopt.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <string>
#include <vector>

#include <slam/frontend/feature_detector.hpp>
#include <slam/frontend/feature_matcher.hpp>
#include <slam/frontend/pose_estimator.hpp>

#include <slam/preprocessing/png_gray.hpp>

static void printHelp() {
    std::printf("cli -c <config_dir> -v <stream_path> [-p <levels>]\n"
                "  -c  directory holding feature_detector.yml and feature_matcher.yml\n"
                "  -v  directory of 8-bit PNG frames (processed in lexical order)\n"
                "  -p  detect on an image pyramid of this many levels (scale 1.2); default: full resolution only\n"
                "  -h  this help\n");
}

int main(int argc, char** argv) {
    int c{};
    std::string streamPath{};
    std::string configPath{};
    int pyramidLevels = 0;
    while ((c = getopt(argc, argv, "hv:c:p:")) != -1) {
        switch (c) {
            case 'h':
                printHelp();
                return 0;
            case 'v':
                streamPath = optarg;
                break;
            case 'c':
                configPath = optarg;
                break;
            case 'p':
                pyramidLevels = std::atoi(optarg);
                break;
            default:
                printHelp();
                return 0;
        }
    }
    if (streamPath.empty() || configPath.empty()) {
        printHelp();
        return 0;
    }
    try {
        namespace fs = std::filesystem;
        slam::FeatureDetector detector(fs::path(configPath) / "feature_detector.yml");
        slam::FeatureMatcher matcher(fs::path(configPath) / "feature_matcher.yml");
        // Preprocessor::prepareDirectory, src/preprocessing/preprocessor.cpp:24-41
        std::vector<fs::path> files;
        for (const auto& entry : fs::directory_iterator(streamPath)) {
            if (entry.is_regular_file() && entry.path().extension() == ".png") {
                files.push_back(entry.path());
            }
        }
        std::sort(files.begin(), files.end());
        const bool havePose = fs::exists(fs::path(configPath) / "camera.yml");
        slam::Camera camera = havePose ? slam::Camera(fs::path(configPath) / "camera.yml") : slam::Camera();
        slam::PoseEstimator poseEstimator(camera);
        std::vector<slam::Keypoint> prevKps;
        slam::DescriptorMatrix prevDesc;
        const auto t0 = std::chrono::steady_clock::now();
        size_t totalKps = 0, totalMatches = 0;
        for (size_t i = 0; i < files.size(); ++i) {
            slam::EigenGrayMatrix image = slamx_tools::read_png_gray(files[i].string());
            std::vector<slam::Keypoint> kps;
            slam::DescriptorMatrix desc;
            if (pyramidLevels > 1) {
                detector.detectAndComputePyramid(image, kps, desc, pyramidLevels, 1.2);
            } else {
                detector.detectAndCompute(image, kps, desc);
            }
            std::vector<slam::Match> matches;
            if (i > 0 && !kps.empty() && !prevKps.empty()) {
                matcher.match(desc, prevDesc, matches, kps, prevKps);
            }
            std::printf("frame %zu %s: %zu keypoints, %zu matches to previous\n", i, files[i].filename().c_str(),
                        kps.size(), matches.size());
            if (havePose && matches.size() >= 8) {
                std::vector<std::pair<int, int>> pairs;
                for (const auto& m : matches) {
                    pairs.emplace_back(m.trainIdx, m.queryIdx);  // (previous frame, this frame)
                }
                slam::Mat3d R{1, 0, 0, 0, 1, 0, 0, 0, 1};
                slam::Vec3d t{0, 0, 0};
                if (poseEstimator.estimate(prevKps, kps, pairs, R, t)) {
                    std::vector<slam::Match> swapped;
                    for (const auto& m : matches) {
                        swapped.push_back({m.trainIdx, m.queryIdx, m.distance});
                    }
                    const auto pts = poseEstimator.triangulatePoints(prevKps, kps, swapped, R, t);
                    std::printf("  pose: t = (%.4f %.4f %.4f), trace(R) = %.6f, %zu map points\n", t[0], t[1], t[2],
                                R[0] + R[4] + R[8], pts.size());
                }
            }
            totalKps += kps.size();
            totalMatches += matches.size();
            prevKps = std::move(kps);
            prevDesc = std::move(desc);
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%zu frames, %zu keypoints, %zu matches in %.3f s\n", files.size(), totalKps, totalMatches, dt);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "An error occurred: %s\n", e.what());
        return -1;
    }
    return 0;
}
