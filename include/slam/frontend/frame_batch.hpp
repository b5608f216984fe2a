// slam/frontend/frame_batch.hpp -- EXTENSION, no reference counterpart: the reference's FeatureDetector /
// FeatureMatcher take ONE frame (pair) per call (include/slam/frontend/feature_detector.hpp:109-119,
// feature_matcher.hpp:33-37), which on a GPU leaves 255 of 256 CUs idle (a frame's sort is one workgroup).
// A FrameBatch runs the SAME detector and matcher objects over many equally sized frames per call -- frames,
// keypoints, descriptors and matches stay in HBM between the calls -- and hands the results back frame by frame in
// the reference's types: frame k of a batch holds exactly what detectAndCompute(image k) and
// match(desc k, desc k-1, matches, kps k, kps k-1) return (tools/test_frame_batch.cpp compares them).
//   slam::FrameBatch batch(detector, rows, cols, 256);
//   for (i...) batch.setFrame(i, image[i]);
//   batch.detectAndCompute(n);
//   batch.matchConsecutive(matcher, n);        // frame k against frame k - 1, k = 1 .. n - 1
//   batch.fetch(n);                            // ONE copy of every result into page-locked memory
//   batch.keypoints(k, kps); batch.descriptors(k, desc); batch.matches(matcher, k, m);
#pragma once

#include <algorithm>
#include <cstring>
#include <vector>

#include <slam/frontend/feature_detector.hpp>
#include <slam/frontend/feature_matcher.hpp>

namespace slam {

class FrameBatch {
public:
    // maxCandidates / maxKeypoints: per-frame capacities (0 = the library's defaults: a quarter / a sixteenth of the
    // pixels); a frame that exceeds them makes fetch() throw std::runtime_error, as an overflow of the C ABI does.
    FrameBatch(FeatureDetector& detector, int rows, int cols, int maxFrames, int maxCandidates = 0, int maxKeypoints = 0)
        : m_rows(rows), m_cols(cols), m_maxFrames(maxFrames), m_ctx(slamx_detector_context(detector.handle())) {
        slamx_batch_shape shape{};
        shape.rows = rows;
        shape.cols = cols;
        shape.max_frames = maxFrames;
        shape.max_candidates = maxCandidates > 0 ? maxCandidates : std::max(1024, rows * cols / 4);
        shape.max_keypoints = maxKeypoints > 0 ? maxKeypoints : std::max(1024, rows * cols / 16);
        detail::check(slamx_batch_create(detector.handle(), &shape, &m_batch));
        m_width = slamx_detector_descriptor_width(detector.handle());
        const size_t frameBytes = static_cast<size_t>(rows) * static_cast<size_t>(cols);
        void* p = nullptr;
        int rc = slamx_host_alloc(frameBytes * static_cast<size_t>(maxFrames), &p);
        if (rc == SLAMX_OK) {
            m_frames = static_cast<uint8_t*>(p);
            m_resultBytes = slamx_batch_packed_bound(m_batch, maxFrames);
            rc = slamx_host_alloc(m_resultBytes, &p);
            if (rc == SLAMX_OK) {
                m_results = static_cast<uint8_t*>(p);
            }
        }
        if (rc != SLAMX_OK) {
            release();
            detail::check(rc);
        }
    }
    ~FrameBatch() { release(); }
    FrameBatch(const FrameBatch&) = delete;
    FrameBatch& operator=(const FrameBatch&) = delete;

    int rows() const { return m_rows; }
    int cols() const { return m_cols; }
    int capacity() const { return m_maxFrames; }
    slamx_batch* handle() const { return m_batch; }

    // The image of slot `slot` (copied into the batch's page-locked block; uploaded by detectAndCompute).
    void setFrame(int slot, const EigenGrayMatrix& image) {
        if (slot < 0 || slot >= m_maxFrames || image.rows() != m_rows || image.cols() != m_cols) {
            throw std::invalid_argument("FrameBatch::setFrame: slot out of range or image of another size");
        }
        std::memcpy(m_frames + static_cast<size_t>(slot) * m_rows * m_cols, image.data(), static_cast<size_t>(m_rows) * m_cols);
    }

    // detectAndCompute on frames 0 .. nframes - 1 (asynchronous: returns when the work is queued)
    void detectAndCompute(int nframes) {
        detail::check(slamx_batch_upload_async(m_batch, 0, nframes, m_frames, m_cols, static_cast<std::ptrdiff_t>(m_rows) * m_cols));
        detail::check(slamx_batch_detect_and_compute(m_batch, nframes));
        m_matched = false;
    }

    // match(frame k, frame k - 1) for k = 1 .. nframes - 1, with the jump-radius penalty when useKeypoints is set (the
    // five-argument form of FeatureMatcher::match); frame 0 is matched against frame firstTrainFrame, or not at all (-1).
    void matchConsecutive(const FeatureMatcher& matcher, int nframes, bool useKeypoints = true, int firstTrainFrame = -1) {
        detail::check(slamx_batch_match_consecutive(m_batch, matcher.handle(), nframes, useKeypoints ? 1 : 0, firstTrainFrame));
        m_matched = true;
    }

    // Every result of frames 0 .. nframes - 1 in one copy; waits for the batch.  Throws on a capacity overflow.
    void fetch(int nframes) {
        detail::check(slamx_batch_download_packed_async(m_batch, nframes, m_results, m_resultBytes));
        detail::check(slamx_ctx_synchronize(m_ctx));  // (the detector's context)
        std::memcpy(&m_header, m_results, sizeof(m_header));
        if (static_cast<size_t>(m_header.total_bytes) > m_resultBytes) {
            throw std::runtime_error("FrameBatch::fetch: packed results larger than their bound");
        }
        const int32_t* status = reinterpret_cast<const int32_t*>(m_results + m_header.status_off);
        for (int f = 0; f < nframes; ++f) {
            if (status[f] != 0) {
                throw std::runtime_error("FrameBatch::fetch: frame " + std::to_string(f) + " exceeded the batch's capacities (status " +
                                         std::to_string(status[f]) + ")");
            }
        }
        m_fetched = nframes;
    }

    int keypointCount(int frame) const {
        const int32_t* first = kpFirst(frame);
        return first[1] - first[0];
    }
    void keypoints(int frame, std::vector<Keypoint>& out) const {
        const int32_t* first = kpFirst(frame);
        const Keypoint* src = reinterpret_cast<const Keypoint*>(m_results + m_header.kps_off) + first[0];
        out.assign(src, src + (first[1] - first[0]));
    }
    void descriptors(int frame, DescriptorMatrix& out) const {
        const int32_t* first = kpFirst(frame);
        const int n = first[1] - first[0];
        if (n == 0) {
            out = DescriptorMatrix(0, 0);
            return;
        }
        out.resize(n, m_width);
        std::memcpy(out.data(), m_results + m_header.desc_off + static_cast<size_t>(first[0]) * m_width, static_cast<size_t>(n) * m_width);
    }
    // What FeatureMatcher::match(desc[frame], desc[frame - 1], ...) returns: the raw matches of the frame with the
    // matcher's filterAndSortMatches step (feature_matcher.cpp:191-204) applied by the library.
    void matches(const FeatureMatcher& matcher, int frame, std::vector<Match>& out) const {
        if (!m_matched) {
            throw std::runtime_error("FrameBatch::matches: matchConsecutive has not run on this batch");
        }
        const int cap = std::max(1, keypointCount(frame));
        out.resize(static_cast<size_t>(cap));
        int n = 0;
        detail::check(slamx_batch_download_matches(m_batch, matcher.handle(), frame, 1, reinterpret_cast<slamx_match*>(out.data()), cap, &n));
        out.resize(static_cast<size_t>(n));
    }

private:
    const int32_t* kpFirst(int frame) const {
        if (frame < 0 || frame >= m_fetched) {
            throw std::invalid_argument("FrameBatch: frame not fetched");
        }
        return reinterpret_cast<const int32_t*>(m_results + m_header.first_off) + frame;
    }
    void release() {
        slamx_batch_destroy(m_batch);
        m_batch = nullptr;
        slamx_host_free(m_frames);
        slamx_host_free(m_results);
        m_frames = nullptr;
        m_results = nullptr;
    }

    int m_rows, m_cols, m_maxFrames;
    slamx_ctx* m_ctx;
    int m_width{};
    slamx_batch* m_batch{};
    uint8_t* m_frames{};   // page-locked: [maxFrames][rows][cols]
    uint8_t* m_results{};  // page-locked: the packed results of the last fetch (slamx_packed_header + arrays)
    size_t m_resultBytes{};
    slamx_packed_header m_header{};
    int m_fetched{};
    bool m_matched{};
};

}  // namespace slam
