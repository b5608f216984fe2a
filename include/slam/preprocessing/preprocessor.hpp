// slam/preprocessing/preprocessor.hpp -- facade of slam::Preprocessor
// (reference: include/slam/preprocessing/preprocessor.hpp:16-55, src/preprocessing/preprocessor.cpp).
// Directory streams only: the frames are the .png files of the directory in lexical order with the
// capture times of its timestamps.txt; every yield() reads the next frame as gray and undistorts it
// on the GPU (Camera::undistortImage -> slamx_undistort).  Video files need cv::VideoCapture, which
// this build does not have: the constructor throws for them as the reference does when the capture
// cannot be opened.  .jpg files (accepted by the reference's directory scan) are rejected loudly.
#pragma once

#include <algorithm>
#include <chrono>
#include <ctime>
#include <filesystem>
#include <fstream>
#include <functional>
#include <iomanip>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include <slam/common/camera.hpp>
#include <slam/common/common.hpp>
#include <slam/preprocessing/png_gray.hpp>

namespace slam {

using MatrixTimePair = std::pair<RowMajorMatrix<double>, std::chrono::system_clock::time_point>;

class Preprocessor {
public:
    Preprocessor(std::filesystem::path streamPath, const Camera& camera, int frameSkip = 0)
        : m_frameSkip(frameSkip), m_streamPath(std::move(streamPath)), m_camera(camera) {
        if (std::filesystem::is_directory(m_streamPath)) {
            prepareDirectory();
        } else if (std::filesystem::is_regular_file(m_streamPath)) {
            throw std::runtime_error("Could not open video file: " + m_streamPath.string());  // preprocessor.cpp:85-87
        } else {
            throw std::runtime_error("Unsupported stream type: " + m_streamPath.string());  // :19-21
        }
    }

    // The next undistorted frame ([0,1] doubles) and its timestamp; an empty matrix at the end of
    // the stream (preprocessor.cpp:99-102).
    MatrixTimePair yield() {
        MatrixTimePair pair;
        pair.second = std::chrono::system_clock::now();
        if (m_frameNumber >= m_totalFrames) {
            return pair;
        }
        const std::filesystem::path& file = m_files[static_cast<size_t>(m_frameNumber)];
        if (file.extension() != ".png") {
            throw std::runtime_error("Failed to read image from file: " + file.string() + " (only PNG is supported)");
        }
        // cv::imread(IMREAD_COLOR) + cvtColor(BGR2GRAY) (:112-113, :137-138): identity on gray files,
        // OpenCV's fixed-point weights on colour files (png_gray.hpp)
        const EigenGrayMatrix gray = slamx_tools::read_png_gray(file.string());
        // :117-122: the capture time, truncated to whole milliseconds
        constexpr double NANOSECONDS_PER_MILLISECOND = 1.0e6;
        const auto ns = std::chrono::duration_cast<std::chrono::nanoseconds>(
                            m_timestamps[static_cast<size_t>(m_frameNumber)].time_since_epoch())
                            .count();
        const double timestampMs = static_cast<double>(ns) / NANOSECONDS_PER_MILLISECOND;
        pair.second = std::chrono::system_clock::time_point(std::chrono::milliseconds(static_cast<int64_t>(timestampMs)));
        pair.first = m_camera.get().undistortImage(gray);
        m_frameNumber += 1 + m_frameSkip;
        return pair;
    }

    [[nodiscard]] int totalFrames() const { return m_totalFrames; }

private:
    // preprocessor.cpp:24-82
    void prepareDirectory() {
        for (const auto& entry : std::filesystem::directory_iterator(m_streamPath)) {
            if ((entry.is_regular_file() && entry.path().extension() == ".jpg") || entry.path().extension() == ".png") {
                m_totalFrames++;
                m_files.push_back(entry.path());
            }
        }
        std::sort(m_files.begin(), m_files.end());
        std::ifstream file(m_streamPath / "timestamps.txt");
        if (!file) {
            throw std::runtime_error("Could not open timestamps.txt in directory: " + m_streamPath.string());
        }
        std::string line;
        while (std::getline(file, line)) {
            const auto decimalPos = line.find('.');
            if (decimalPos == std::string::npos) {
                continue;  // "Invalid Format" (logged and skipped in the reference)
            }
            const std::string mainPart = line.substr(0, decimalPos);
            const std::string nanoPart = line.substr(decimalPos + 1);
            std::tm timeStruct = {};
            std::stringstream ss(mainPart);
            ss >> std::get_time(&timeStruct, "%Y-%m-%d %H:%M:%S");
            if (ss.fail()) {
                continue;
            }
            auto timePoint = std::chrono::system_clock::from_time_t(std::mktime(&timeStruct));
            const int64_t nanoseconds = std::stoll(nanoPart);
            timePoint += std::chrono::duration_cast<std::chrono::system_clock::duration>(std::chrono::nanoseconds(nanoseconds));
            m_timestamps.push_back(timePoint);
        }
        if (m_timestamps.size() != static_cast<size_t>(m_totalFrames)) {
            throw std::runtime_error("Number of timestamps does not match number of frames.");
        }
    }

    int m_frameNumber{0};
    int m_totalFrames{0};
    int m_frameSkip{0};
    std::vector<std::chrono::system_clock::time_point> m_timestamps;
    std::vector<std::filesystem::path> m_files;
    std::filesystem::path m_streamPath;
    std::reference_wrapper<const Camera> m_camera;
};

}  // namespace slam
