// slam/backend/loop_closure.hpp -- facade of slam::LoopClosure
// (reference: include/slam/backend/loop_closure.hpp:17-73, src/backend/loop_closure.cpp).
// Everything after the candidate search is the reference's: parameters and their checks, the
// keyframe database, match() against the candidate, the 6-point DLT PnP inside a RANSAC loop
// (all hypotheses solved and scored on the GPU), the inlier threshold, the 4x4 result.  The
// candidate search itself scores keyframes with fbow bag-of-words vectors in the reference; fbow
// and its vocabulary file are not part of the reference tree (.MISSING_LARGE_BLOBS), so this build
// scores a keyframe by descriptor voting: the fraction of the query's descriptors that find a match
// in the keyframe under the matcher's own filters (no keypoints, hence no jump penalty).  The
// best / second-best decision rule on those scores is the reference's (loop_closure.cpp:128-146).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <filesystem>
#include <functional>
#include <map>
#include <optional>
#include <string>
#include <vector>

#include <slam/common/camera.hpp>
#include <slam/common/common.hpp>
#include <slam/frontend/feature_detector.hpp>
#include <slam/frontend/feature_matcher.hpp>
#include <slam/frontend/slamx_runtime.hpp>

namespace slam {

// LoopClosure::solvePnP, loop_closure.cpp:238-274 (private static in the reference).
inline bool solvePnP(const std::vector<Vec3d>& points3d, const std::vector<std::array<double, 2>>& points2d,
                     Mat3d& rotation, Vec3d& translation) {
    if (points3d.size() < 6) {
        return false;  // :245-247
    }
    detail::check(slamx_pnp_dlt_batch(detail::default_context(), points3d[0].data(), points2d[0].data(),
                                      static_cast<int>(points3d.size()), 1, rotation.data(), translation.data(), nullptr));
    return true;
}

struct PnPRansacResult {
    Mat3d rotation{1, 0, 0, 0, 1, 0, 0, 0, 1};
    Vec3d translation{0, 0, 0};
    int inliers = 0;
};

// The RANSAC loop of verifyGeometricConsistency, loop_closure.cpp:173-222, with an explicit seed
// (the reference draws it from std::random_device).
inline PnPRansacResult pnpRansac(const std::vector<Vec3d>& points3d, const std::vector<std::array<double, 2>>& points2d,
                                 const Mat3d& K, int maxIterations, double reprojectionThreshold, uint32_t seed) {
    PnPRansacResult r;
    detail::check(slamx_pnp_ransac(detail::default_context(), points3d[0].data(), points2d[0].data(),
                                   static_cast<int>(points3d.size()), K.data(), maxIterations, reprojectionThreshold, seed,
                                   r.rotation.data(), r.translation.data(), &r.inliers));
    return r;
}

#ifdef SLAMX_HAVE_EIGEN
// include/slam/backend/loop_closure.hpp:17-20 of the reference
struct LoopResult {
    int matchedKeyframeId{};
    Eigen::Matrix4d relativeTransform = Eigen::Matrix4d::Identity();
};
#else
struct LoopResult {
    int matchedKeyframeId{};
    std::array<double, 16> relativeTransform{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // row-major 4x4
};
#endif

namespace detail {
inline void set_transform(LoopResult& r, int i, int j, double v) {
#ifdef SLAMX_HAVE_EIGEN
    r.relativeTransform(i, j) = v;
#else
    r.relativeTransform[static_cast<size_t>(i * 4 + j)] = v;
#endif
}
}  // namespace detail

class LoopClosure {
public:
    // vocabPath is accepted for signature compatibility and ignored (no fbow in this build); seed
    // feeds the RANSAC sampler, which the reference seeds from std::random_device.
    LoopClosure(const std::string& /*vocabPath*/, const std::string& configPath, const FeatureMatcher& matcher,
                uint32_t seed = 0)
        : m_matcher(matcher), m_seed(seed) {
        detail::check(slamx_loop_closure_config_load(configPath.c_str(), &m_params));
    }

    // loop_closure.cpp:97-109
    void addKeyframe(int keyframeId, const DescriptorMatrix& descriptors, const std::vector<Keypoint>& keypoints,
                     const std::vector<Vec3d>& mapPoints) {
        KeyframeData data;
        data.keypoints = keypoints;
        data.mapPoints = mapPoints;
        m_keyframeDatabase[keyframeId] = data;
        m_keyframeDescriptors[keyframeId] = descriptors;
        m_lastKeyframeId = keyframeId;
    }

#ifdef SLAMX_HAVE_EIGEN
    // the reference's own signature (include/slam/backend/loop_closure.hpp:26-28)
    void addKeyframe(int keyframeId, const DescriptorMatrix& descriptors, const std::vector<Keypoint>& keypoints,
                     const std::vector<Eigen::Vector3d>& mapPoints) {
        std::vector<Vec3d> pts(mapPoints.size());
        for (size_t i = 0; i < mapPoints.size(); ++i) {
            pts[i] = {mapPoints[i](0), mapPoints[i](1), mapPoints[i](2)};
        }
        addKeyframe(keyframeId, descriptors, keypoints, pts);
    }
#endif

    // loop_closure.cpp:111-151
    [[nodiscard]] std::optional<LoopResult> detect(const DescriptorMatrix& descriptors, const std::vector<Keypoint>& keypoints,
                                                   const Camera& camera) {
        if (m_keyframeDatabase.size() < static_cast<size_t>(m_params.min_db_size)) {
            return std::nullopt;
        }
        if (descriptors.rows() == 0) {
            return std::nullopt;  // an empty bag-of-words vector in the reference
        }
        int bestCandidateId = -1;
        double maxScore = 0.0;
        double secondMaxScore = 0.0;
        for (const auto& [id, data] : m_keyframeDatabase) {
            (void)data;
            if (std::abs(m_lastKeyframeId - id) < m_params.min_frames_difference) {
                continue;
            }
            const double score = votingScore(descriptors, m_keyframeDescriptors.at(id));
            if (score > maxScore) {
                secondMaxScore = maxScore;
                maxScore = score;
                bestCandidateId = id;
            } else if (score > secondMaxScore) {
                secondMaxScore = score;
            }
        }
        if (bestCandidateId == -1 || maxScore < m_params.min_absolute_score ||
            maxScore < m_params.relative_score_factor * secondMaxScore) {
            return std::nullopt;
        }
        return verifyGeometricConsistency(descriptors, keypoints, bestCandidateId, camera);
    }

    // the score that replaces fbow::BoWVector::score (see the header comment)
    [[nodiscard]] double votingScore(const DescriptorMatrix& query, const DescriptorMatrix& keyframe) const {
        if (query.rows() == 0 || keyframe.rows() == 0) {
            return 0.0;
        }
        std::vector<Match> matches;
        m_matcher.get().match(query, keyframe, matches);
        return static_cast<double>(matches.size()) / static_cast<double>(query.rows());
    }

private:
    struct KeyframeData {
        std::vector<Keypoint> keypoints;
        std::vector<Vec3d> mapPoints;
    };

    // loop_closure.cpp:153-236
    std::optional<LoopResult> verifyGeometricConsistency(const DescriptorMatrix& queryDescriptors,
                                                         const std::vector<Keypoint>& queryKeypoints, int candidateId,
                                                         const Camera& camera) {
        std::vector<Match> matches;
        const KeyframeData& kf = m_keyframeDatabase.at(candidateId);
        m_matcher.get().match(queryDescriptors, m_keyframeDescriptors.at(candidateId), matches, queryKeypoints, kf.keypoints);
        if (matches.size() < static_cast<size_t>(m_params.min_matches_for_pnp)) {
            return std::nullopt;
        }
        std::vector<Vec3d> points3d;
        std::vector<std::array<double, 2>> points2d;
        points3d.reserve(matches.size());
        points2d.reserve(matches.size());
        for (const auto& match : matches) {
            points2d.push_back({static_cast<double>(queryKeypoints[static_cast<size_t>(match.queryIdx)].x),
                                static_cast<double>(queryKeypoints[static_cast<size_t>(match.queryIdx)].y)});
            points3d.push_back(kf.mapPoints[static_cast<size_t>(match.trainIdx)]);
        }
        const PnPRansacResult r = pnpRansac(points3d, points2d, camera.intrinsics(), m_params.ransac_max_iterations,
                                            m_params.ransac_reprojection_threshold, m_seed);
        if (r.inliers >= m_params.min_inliers_for_pnp) {
            LoopResult result;
            result.matchedKeyframeId = candidateId;
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) {
                    detail::set_transform(result, i, j, r.rotation[static_cast<size_t>(i * 3 + j)]);
                }
                detail::set_transform(result, i, 3, r.translation[static_cast<size_t>(i)]);
            }
            return result;
        }
        return std::nullopt;
    }

    slamx_loop_closure_config m_params{};
    std::reference_wrapper<const FeatureMatcher> m_matcher;
    uint32_t m_seed{0};
    std::map<int, KeyframeData> m_keyframeDatabase;
    std::map<int, DescriptorMatrix> m_keyframeDescriptors;
    int m_lastKeyframeId{-1};
};

}  // namespace slam
