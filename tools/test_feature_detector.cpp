// Counterpart of the reference's test/frontend/test_feature_detector.cpp (:12-98): same call
// sequence on the same image through the slam:: facade (detect, compute, detectAndCompute), with
// the OpenCV drawing replaced by a binary dump that tests/test_facade_gpu.py compares with the
// oracle.  Usage: test_feature_detector <feature_detector.yml> <image.png> <out.bin>
#include <chrono>
#include <cstdio>
#include <fstream>

#include <slam/frontend/feature_detector.hpp>

#include <slam/preprocessing/png_gray.hpp>

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <feature_detector.yml> <image.png> <out.bin>\n", argv[0]);
        return -1;
    }
    try {
        slam::FeatureDetector featureDetector(argv[1]);
        slam::EigenGrayMatrix eigenImage = slamx_tools::read_png_gray(argv[2]);

        std::vector<slam::Keypoint> keypoints;
        slam::DescriptorMatrix descriptors;

        auto detectStart = std::chrono::high_resolution_clock::now();
        featureDetector.detect(eigenImage, keypoints);
        auto detectEnd = std::chrono::high_resolution_clock::now();
        auto computeStart = std::chrono::high_resolution_clock::now();
        featureDetector.compute(eigenImage, keypoints, descriptors);
        auto computeEnd = std::chrono::high_resolution_clock::now();

        std::printf("Detected %zu keypoints\n", keypoints.size());
        std::printf("Computed %td descriptors\n", descriptors.rows());
        std::printf("Keypoint detection: %lld us\n",
                    (long long)std::chrono::duration_cast<std::chrono::microseconds>(detectEnd - detectStart).count());
        std::printf("Descriptor computation: %lld us\n",
                    (long long)std::chrono::duration_cast<std::chrono::microseconds>(computeEnd - computeStart).count());

        std::vector<slam::Keypoint> keypoints2;
        slam::DescriptorMatrix descriptors2;
        featureDetector.detectAndCompute(eigenImage, keypoints2, descriptors2);
        std::printf("DetectAndCompute: %zu keypoints, %td descriptors\n", keypoints2.size(), descriptors2.rows());

        std::ofstream out(argv[3], std::ios::binary);
        for (const auto* kd : {&keypoints, &keypoints2}) {
            const int32_t n = static_cast<int32_t>(kd->size());
            out.write(reinterpret_cast<const char*>(&n), 4);
            out.write(reinterpret_cast<const char*>(kd->data()), sizeof(slam::Keypoint) * kd->size());
        }
        for (const auto* d : {&descriptors, &descriptors2}) {
            const int32_t r = static_cast<int32_t>(d->rows()), c = static_cast<int32_t>(d->cols());
            out.write(reinterpret_cast<const char*>(&r), 4);
            out.write(reinterpret_cast<const char*>(&c), 4);
            out.write(reinterpret_cast<const char*>(d->data()), static_cast<std::streamsize>(d->size()));
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "An error occurred: %s\n", e.what());
        return -1;
    }
    return 0;
}
