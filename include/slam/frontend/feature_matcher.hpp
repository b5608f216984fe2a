// slam/frontend/feature_matcher.hpp -- drop-in facade for the reference's
// include/slam/frontend/feature_matcher.hpp (+ src/frontend/feature_matcher.cpp): same names,
// signature, defaults and exceptions; forwards to the C ABI.
#pragma once

#include <filesystem>
#include <vector>

#include <slam/frontend/feature_detector.hpp>

namespace slam {
namespace constants {
constexpr uint32_t MAX_JUMP_RADIUS = 500;  // feature_matcher.hpp:12
}  // namespace constants

// feature_matcher.hpp:18-25 -- identical layout to slamx_match.
struct Match {
    int queryIdx;
    int trainIdx;
    float distance;

    Match() : queryIdx(0), trainIdx(0), distance(0.0F) {}
    Match(int qIdx, int tIdx, float dist) : queryIdx(qIdx), trainIdx(tIdx), distance(dist) {}
};
static_assert(sizeof(Match) == sizeof(slamx_match), "Match must match the C ABI layout");

class FeatureMatcher {
public:
    enum class DistanceType : uint8_t { HAMMING, L2 };

    // feature_matcher.cpp:18-69
    explicit FeatureMatcher(const std::filesystem::path& configPath) {
        slamx_matcher_config cfg{};
        detail::check(slamx_matcher_config_load(configPath.string().c_str(), &cfg));
        detail::check(slamx_matcher_create(detail::default_context(), &cfg, &m_handle));
    }
    ~FeatureMatcher() { slamx_matcher_destroy(m_handle); }
    FeatureMatcher(const FeatureMatcher&) = delete;
    FeatureMatcher& operator=(const FeatureMatcher&) = delete;

    // (extension) the C-ABI handle, for the batch interface (slam::FrameBatch)
    slamx_matcher* handle() const { return m_handle; }

    // feature_matcher.cpp:71-95.  Throws std::invalid_argument on empty descriptor sets and
    // std::runtime_error on width mismatch / L2, exactly as validateInputs does (:97-111).
    void match(const DescriptorMatrix& descriptors1, const DescriptorMatrix& descriptors2, std::vector<Match>& matches,
               const std::vector<Keypoint>& keypoints1 = {}, const std::vector<Keypoint>& keypoints2 = {}) const {
        matches.clear();
        const int n1 = static_cast<int>(descriptors1.rows());
        const int n2 = static_cast<int>(descriptors2.rows());
        std::vector<Match> out(static_cast<size_t>(n1 > 0 ? n1 : 1));
        int n = 0;
        detail::check(slamx_matcher_match(
            m_handle, descriptors1.data(), n1, static_cast<int>(descriptors1.cols()), descriptors2.data(), n2,
            static_cast<int>(descriptors2.cols()), reinterpret_cast<const slamx_keypoint*>(keypoints1.data()),
            static_cast<int>(keypoints1.size()), reinterpret_cast<const slamx_keypoint*>(keypoints2.data()),
            static_cast<int>(keypoints2.size()), reinterpret_cast<slamx_match*>(out.data()), static_cast<int>(out.size()),
            &n));
        out.resize(static_cast<size_t>(n));
        matches = std::move(out);
    }

private:
    slamx_matcher* m_handle{};
};

}  // namespace slam
