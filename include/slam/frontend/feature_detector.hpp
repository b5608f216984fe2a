// slam/frontend/feature_detector.hpp -- drop-in facade for the reference's
// include/slam/frontend/feature_detector.hpp: same namespace, type and method names, argument
// meaning, output semantics and exception types; every method forwards to the C ABI (slamx.h),
// which runs the HIP kernels.  No CPU implementation exists behind this class.
#pragma once

#include <cstdlib>
#include <filesystem>
#include <utility>
#include <vector>

#include <slam/common/common.hpp>
#include <slam/frontend/slamx_runtime.hpp>

namespace slam {

namespace constants {
constexpr int CIRCLE_PERIMETER = 16;
constexpr int BRIEF_PAIRS = 8;
constexpr int BLUR_KERNEL_SIZE = 5;
constexpr float DEFAULT_KEYPOINT_SIZE = 6.0F;
}  // namespace constants

// feature_detector.hpp:28-38 -- identical layout to slamx_keypoint (5 floats).
struct Keypoint {
    float x{};
    float y{};
    float size{};
    float angle{};
    float response{};

    Keypoint() = default;
    Keypoint(float X, float Y, float size = constants::DEFAULT_KEYPOINT_SIZE) : x(X), y(Y), size(size) {}
};
static_assert(sizeof(Keypoint) == sizeof(slamx_keypoint), "Keypoint must match the C ABI layout");

class FeatureDetector {
public:
    // feature_detector.hpp:53-107: reads the six YAML keys, validates, generates the BRIEF pattern.
    explicit FeatureDetector(const std::filesystem::path& configPath) {
        slamx_detector_config cfg{};
        detail::check(slamx_detector_config_load(configPath.string().c_str(), &cfg));
        detail::check(slamx_detector_create(detail::default_context(), &cfg, &m_handle));
    }
    ~FeatureDetector() {
        slamx_pyramid_destroy(m_pyramid);
        slamx_detector_destroy(m_handle);
    }
    FeatureDetector(const FeatureDetector&) = delete;
    FeatureDetector& operator=(const FeatureDetector&) = delete;

    // (extension) the C-ABI handle, for the batch interface (slam::FrameBatch)
    slamx_detector* handle() const { return m_handle; }

    // feature_detector.cpp:8-18
    void detect(const EigenGrayMatrix& image, std::vector<Keypoint>& keypoints) {
        keypoints.clear();
        run_sized(image, keypoints, nullptr);
    }

    // feature_detector.cpp:20-47 (angles are written back into `keypoints`)
    void compute(const EigenGrayMatrix& image, std::vector<Keypoint>& keypoints, DescriptorMatrix& descriptors) {
        if (keypoints.empty()) {
            descriptors = DescriptorMatrix(0, 0);
            return;
        }
        const int width = slamx_detector_descriptor_width(m_handle);
        descriptors.resize(static_cast<std::ptrdiff_t>(keypoints.size()), width);
        detail::check(slamx_compute(m_handle, image.data(), static_cast<int>(image.rows()), static_cast<int>(image.cols()),
                                    static_cast<std::ptrdiff_t>(image.cols()),
                                    reinterpret_cast<slamx_keypoint*>(keypoints.data()),
                                    static_cast<int>(keypoints.size()), descriptors.data()));
    }

    // feature_detector.cpp:49-54
    void detectAndCompute(const EigenGrayMatrix& image, std::vector<Keypoint>& keypoints,
                          DescriptorMatrix& descriptors) {
        keypoints.clear();
        run_sized(image, keypoints, &descriptors);
    }

    // EXTENSION, no reference counterpart (the reference detects at full resolution only, feature_detector.cpp:8-18;
    // BASELINE.json configs[2] asks for an 8-level pyramid): the same detector on every level of an image pyramid
    // (level l = cvRound(size / scaleFactor^l), integer bilinear downscale), results of all levels in one list with
    // x, y and size scaled to the full-resolution frame (Keypoint::size = 6 * scaleFactor^l tells the level).
    void detectAndComputePyramid(const EigenGrayMatrix& image, std::vector<Keypoint>& keypoints, DescriptorMatrix& descriptors,
                                 int nLevels = 8, double scaleFactor = 1.2) {
        const int rows = static_cast<int>(image.rows());
        const int cols = static_cast<int>(image.cols());
        const int width = slamx_detector_descriptor_width(m_handle);
        // Capacities: a quarter of the pixels as corners, an eighth as keypoints; a frame that overflows either (or the
        // merged list) is run once more with "every pixel" sizing, as the single-frame path does -- an overflow must
        // never come back as an empty keypoint list.
        for (int attempt = 0; attempt < 2; ++attempt) {
            const bool big = attempt == 1 || m_pyrBig;
            if (m_pyramid == nullptr || rows != m_pyrRows || cols != m_pyrCols || nLevels != m_pyrLevels ||
                scaleFactor != m_pyrScale || big != m_pyrBig) {
                slamx_pyramid_destroy(m_pyramid);
                m_pyramid = nullptr;
                slamx_batch_shape shape{};
                shape.rows = rows;
                shape.cols = cols;
                shape.max_frames = 1;
                shape.max_candidates = big ? std::min(rows * cols, 1 << 20) : std::max(1024, rows * cols / 4);
                shape.max_keypoints = big ? std::min(nLevels * rows * cols, 1 << 21) : std::max(1024, rows * cols / 8);
                detail::check(slamx_pyramid_create(m_handle, &shape, nLevels, scaleFactor, &m_pyramid));
                m_pyrRows = rows;
                m_pyrCols = cols;
                m_pyrLevels = nLevels;
                m_pyrScale = scaleFactor;
                m_pyrBig = big;
            }
            slamx_batch* level0 = slamx_pyramid_batch(m_pyramid, 0);
            detail::check(slamx_batch_upload(level0, 0, 1, image.data(), cols, 0));
            detail::check(slamx_pyramid_detect_and_compute(m_pyramid, 1));
            // the count through slamx_batch_counts: it reports a candidate / keypoint / merge overflow of the frame as
            // SLAMX_ERR_CAPACITY (slamx_batch_download(cap = 0) uses that same code for "buffer too small")
            int32_t n32 = 0;
            const int rc = slamx_batch_counts(level0, 1, nullptr, &n32, nullptr);
            if (rc == SLAMX_ERR_CAPACITY && !big) {
                continue;
            }
            detail::check(rc);
            int n = n32;
            keypoints.resize(static_cast<size_t>(n));
            if (n == 0) {
                descriptors = DescriptorMatrix(0, 0);
                return;
            }
            descriptors.resize(n, width);
            detail::check(slamx_batch_download(level0, 0, reinterpret_cast<slamx_keypoint*>(keypoints.data()), descriptors.data(), n, &n));
            return;
        }
    }

private:
    // Capacity-checked output sizing over the two-call C ABI.  The library writes into scratch vectors that live with
    // the detector (grown, never shrunk): sizing the CALLER's vectors to the capacity first would value-initialise
    // rows * cols / 16 keypoints and as many descriptor rows on every call (2.3 MB of stores, ~80 us, on a 1392 x 512
    // frame that yields 1.1k keypoints).
    void run_sized(const EigenGrayMatrix& image, std::vector<Keypoint>& keypoints, DescriptorMatrix* descriptors) {
        const int rows = static_cast<int>(image.rows());
        const int cols = static_cast<int>(image.cols());
        const int width = slamx_detector_descriptor_width(m_handle);
        int cap = std::max(1024, rows * cols / 16);
        for (int attempt = 0; attempt < 3; ++attempt) {
            if (m_kpScratch.size() < static_cast<size_t>(cap)) {
                m_kpScratch.resize(static_cast<size_t>(cap));
            }
            int n = 0;
            int rc;
            if (descriptors != nullptr) {
                if (m_descScratch.size() < static_cast<size_t>(cap) * static_cast<size_t>(width)) {
                    m_descScratch.resize(static_cast<size_t>(cap) * static_cast<size_t>(width));
                }
                rc = slamx_detect_and_compute(m_handle, image.data(), rows, cols, cols,
                                              reinterpret_cast<slamx_keypoint*>(m_kpScratch.data()), m_descScratch.data(), cap, &n);
            } else {
                rc = slamx_detect(m_handle, image.data(), rows, cols, cols, reinterpret_cast<slamx_keypoint*>(m_kpScratch.data()),
                                  cap, &n);
            }
            if (rc == SLAMX_ERR_CAPACITY && n > cap) {
                cap = n;
                continue;
            }
            detail::check(rc);
            keypoints.assign(m_kpScratch.begin(), m_kpScratch.begin() + n);
            if (descriptors != nullptr) {
                if (n == 0) {
                    *descriptors = DescriptorMatrix(0, 0);
                } else {
                    descriptors->resize(n, width);
                    std::copy(m_descScratch.begin(), m_descScratch.begin() + static_cast<std::ptrdiff_t>(n) * width,
                              descriptors->data());
                }
            }
            return;
        }
        throw std::runtime_error("keypoint capacity negotiation failed");
    }

    slamx_detector* m_handle{};
    std::vector<Keypoint> m_kpScratch;    // run_sized: what the C ABI writes into
    std::vector<uint8_t> m_descScratch;
    slamx_pyramid* m_pyramid{};  // lazily created by detectAndComputePyramid, one frame deep
    int m_pyrRows{}, m_pyrCols{}, m_pyrLevels{};
    bool m_pyrBig{};  // "every pixel" capacities after an overflow
    double m_pyrScale{};
};

}  // namespace slam
