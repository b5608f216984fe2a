// slam/preprocessing/preprocessor.hpp -- facade of slam::Preprocessor
// (reference: include/slam/preprocessing/preprocessor.hpp:16-55, src/preprocessing/preprocessor.cpp).
// Directory streams: the frames are the .png files of the directory in lexical order with the
// capture times of its timestamps.txt; every yield() reads the next frame as gray and undistorts it
// on the GPU (Camera::undistortImage -> slamx_undistort).  .jpg files (accepted by the reference's
// directory scan) are rejected loudly.
// Video files (preprocessor.cpp:84-93, :124-131) go through cv::VideoCapture in the reference, i.e. through
// whatever codecs its OpenCV was built with; this image has no decoder, so the one container that needs none is
// read here -- UNCOMPRESSED YUV4MPEG2 (.y4m: a text header, then per frame "FRAME\n" and the planes; what
// `ffmpeg -i in.mp4 out.y4m` writes): frame k is the file's k-th LUMA plane (the gray image itself; OpenCV would
// go YUV -> BGR -> gray, third-party rounding that nothing pins), random access by frame number as the reference
// seeks (CAP_PROP_POS_FRAMES), timestamp = k / frame rate in whole milliseconds.  Every other regular file is
// refused with the reference's own message for a capture that cannot be opened.
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
            prepareVideo();
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
        if (m_isVideo) {
            // :124-131: seek to the frame, read it, its position in milliseconds
            EigenGrayMatrix gray(m_videoRows, m_videoCols);
            m_video.clear();
            m_video.seekg(m_videoData + static_cast<std::streamoff>(m_frameNumber) * (m_videoFrameBytes + 6) + 6);
            m_video.read(reinterpret_cast<char*>(gray.data()), static_cast<std::streamsize>(m_videoRows) * m_videoCols);
            if (!m_video) {
                throw std::runtime_error("Failed to read frame from video.");
            }
            const double timestampMs = 1000.0 * static_cast<double>(m_frameNumber) * m_videoRateDen / m_videoRateNum;
            pair.second = std::chrono::system_clock::time_point(std::chrono::milliseconds(static_cast<int64_t>(timestampMs)));
            pair.first = m_camera.get().undistortImage(gray);
            m_frameNumber += 1 + m_frameSkip;
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

    // preprocessor.cpp:84-93 for the container that needs no codec (see the header comment)
    void prepareVideo() {
        const std::string cannot = "Could not open video file: " + m_streamPath.string();  // :85-87
        m_video.open(m_streamPath, std::ios::binary);
        std::string header;
        if (!m_video || !std::getline(m_video, header) || header.rfind("YUV4MPEG2 ", 0) != 0) {
            throw std::runtime_error(cannot + " (no decoder in this build: only uncompressed YUV4MPEG2 / .y4m files are read)");
        }
        std::string colour = "420";
        std::istringstream fields(header.substr(10));
        for (std::string f; fields >> f;) {
            if (f[0] == 'W') {
                m_videoCols = std::atoi(f.c_str() + 1);
            } else if (f[0] == 'H') {
                m_videoRows = std::atoi(f.c_str() + 1);
            } else if (f[0] == 'F') {
                const auto colon = f.find(':');
                m_videoRateNum = std::atof(f.c_str() + 1);
                m_videoRateDen = colon == std::string::npos ? 1.0 : std::atof(f.c_str() + colon + 1);
            } else if (f[0] == 'C') {
                colour = f.substr(1);
            }
        }
        const size_t luma = static_cast<size_t>(m_videoRows) * static_cast<size_t>(m_videoCols);
        size_t chroma = 0;  // bytes of the two chroma planes of a frame (8-bit formats only)
        // (high bit depths are spelled 420p10, 444p12, mono16 ...: a 'p' or "mono" followed by digits)
        const bool deep = colour.find("p1") != std::string::npos || colour.find("p9") != std::string::npos;
        if (colour == "mono") {
            chroma = 0;
        } else if (colour.rfind("420", 0) == 0 && !deep) {
            chroma = 2 * (static_cast<size_t>((m_videoRows + 1) / 2) * static_cast<size_t>((m_videoCols + 1) / 2));
        } else if (colour == "422") {
            chroma = 2 * (static_cast<size_t>(m_videoRows) * static_cast<size_t>((m_videoCols + 1) / 2));
        } else if (colour == "444") {
            chroma = 2 * luma;
        } else {
            throw std::runtime_error(cannot + " (YUV4MPEG2 colour space C" + colour + " is not an 8-bit planar format)");
        }
        if (m_videoRows <= 0 || m_videoCols <= 0 || !(m_videoRateNum > 0.0) || !(m_videoRateDen > 0.0)) {
            throw std::runtime_error(cannot + " (YUV4MPEG2 header without a size or frame rate)");
        }
        m_videoFrameBytes = static_cast<std::streamoff>(luma + chroma);
        m_videoData = m_video.tellg();
        m_video.seekg(0, std::ios::end);
        // every frame is "FRAME\n" (no per-frame parameters: they would make frames of different lengths) + its planes
        m_totalFrames = static_cast<int>((m_video.tellg() - m_videoData) / (m_videoFrameBytes + 6));  // CAP_PROP_FRAME_COUNT
        for (int k = 0; k < m_totalFrames; ++k) {
            char tag[6] = {};
            m_video.seekg(m_videoData + static_cast<std::streamoff>(k) * (m_videoFrameBytes + 6));
            m_video.read(tag, 6);
            if (!m_video || std::string(tag, 6) != "FRAME\n") {
                throw std::runtime_error(cannot + " (YUV4MPEG2 frame " + std::to_string(k) + " carries parameters or is cut short)");
            }
        }
        m_isVideo = true;
    }

    bool m_isVideo{false};
    std::ifstream m_video;
    std::streamoff m_videoData{0}, m_videoFrameBytes{0};
    int m_videoRows{0}, m_videoCols{0};
    double m_videoRateNum{0.0}, m_videoRateDen{1.0};
    int m_frameNumber{0};
    int m_totalFrames{0};
    int m_frameSkip{0};
    std::vector<std::chrono::system_clock::time_point> m_timestamps;
    std::vector<std::filesystem::path> m_files;
    std::filesystem::path m_streamPath;
    std::reference_wrapper<const Camera> m_camera;
};

}  // namespace slam
