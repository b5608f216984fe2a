// test_preprocessing -- mirrors the reference's test/preprocessing/test_preprocessing.cpp:12-65:
// undistort one image through slam::Camera, then pull frames from slam::Preprocessor.
//   test_preprocessing <camera.yml> <image.png> <frame directory> <n frames> [dump prefix]
// Prints one line per yielded frame (rows cols timestamp_ms checksum); with a dump prefix the
// undistorted first image is written as raw doubles for the parity test.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include <slam/common/camera.hpp>
#include <slam/preprocessing/preprocessor.hpp>

int main(int argc, char** argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: test_preprocessing <camera.yml> <image.png> <frame dir> <n> [dump prefix]\n");
        return -1;
    }
    try {
        slam::Camera camera(argv[1]);
        const slam::EigenGrayMatrix image = slamx_tools::read_png_gray(argv[2]);
        const slam::RowMajorMatrix<double> undistorted = camera.undistortImage(image);
        if (argc > 5) {
            std::ofstream out(std::string(argv[5]) + "_undistorted.f64", std::ios::binary);
            out.write(reinterpret_cast<const char*>(undistorted.data()),
                      static_cast<std::streamsize>(sizeof(double) * undistorted.rows() * undistorted.cols()));
        }
        std::printf("undistorted %ld x %ld\n", static_cast<long>(undistorted.rows()), static_cast<long>(undistorted.cols()));
        slam::Preprocessor preprocessor(argv[3], camera);
        const int nFrames = std::atoi(argv[4]);
        for (int i = 0; i < nFrames; ++i) {
            slam::MatrixTimePair pair{preprocessor.yield()};
            if (pair.first.rows() == 0 && pair.first.cols() == 0) {
                std::fprintf(stderr, "No frames yielded from preprocessor.\n");
                return -1;
            }
            double sum = 0.0;
            for (std::ptrdiff_t k = 0; k < pair.first.rows() * pair.first.cols(); ++k) {
                sum += pair.first.data()[k];
            }
            const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(pair.second.time_since_epoch()).count();
            std::printf("frame %d: %ld x %ld t_ms %lld sum %.9f\n", i, static_cast<long>(pair.first.rows()),
                        static_cast<long>(pair.first.cols()), static_cast<long long>(ms), sum);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "An error occurred: %s\n", e.what());
        return -1;
    }
    return 0;
}
