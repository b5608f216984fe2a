// slam/frontend/pose_estimator.hpp -- facade of slam::PoseEstimator
// (reference: include/slam/frontend/pose_estimator.hpp:12-32, src/frontend/pose_estimator.cpp:14-104)
// with plain value types instead of cv::KeyPoint / cv::Mat / cv::DMatch.
//
// estimate(): the reference delegates the essential matrix to cv::findEssentialMat(..., cv::RANSAC)
// (pose_estimator.cpp:44, third-party); here it is slamx_essential_ransac -- this build's own
// estimator (slamx.h) -- followed, as in the reference, by simpleRecoverPose on the K^-1-normalised
// points.  triangulatePoints(): K [I|0] and K [R|t] through slam::triangulate.
#pragma once

#include <cstdint>
#include <functional>
#include <utility>
#include <vector>

#include <slam/common/camera.hpp>
#include <slam/common/common.hpp>
#include <slam/frontend/feature_detector.hpp>
#include <slam/frontend/feature_matcher.hpp>
#include <slam/frontend/simple_pose_recover.hpp>
#include <slam/frontend/slamx_runtime.hpp>

namespace slam {

class PoseEstimator {
public:
    explicit PoseEstimator(const Camera& camera) : m_camera(camera) {}

    // RANSAC settings of this build's estimator (cv::findEssentialMat's defaults are prob 0.999 /
    // threshold 1.0 px with 5-point samples; 8-point samples need more hypotheses)
    int ransacIterations = 2048;
    double ransacThresholdPx = 1.0;
    uint32_t ransacSeed = 0;

    // matches: (index into keypoints1, index into keypoints2).  Fewer than 8 matches: R and t are
    // left untouched, as in the reference (pose_estimator.cpp:21-25); returns whether a pose was set.
    bool estimate(const std::vector<Keypoint>& keypoints1, const std::vector<Keypoint>& keypoints2,
                  const std::vector<std::pair<int, int>>& matches, Mat3d& R, Vec3d& t) const {
        if (matches.size() < 8) {
            return false;
        }
        std::vector<Point2f> points1, points2;
        points1.reserve(matches.size());
        points2.reserve(matches.size());
        for (const auto& m : matches) {
            points1.push_back({keypoints1[static_cast<size_t>(m.first)].x, keypoints1[static_cast<size_t>(m.first)].y});
            points2.push_back({keypoints2[static_cast<size_t>(m.second)].x, keypoints2[static_cast<size_t>(m.second)].y});
        }
        const Mat3d K = m_camera.get().intrinsics();
        Mat3d E{};
        int inliers = 0;
        detail::check(slamx_essential_ransac(detail::default_context(), reinterpret_cast<const float*>(points1.data()),
                                             reinterpret_cast<const float*>(points2.data()), static_cast<int>(points1.size()),
                                             K.data(), ransacIterations, ransacThresholdPx, ransacSeed, E.data(), &inliers,
                                             nullptr));
        // pose_estimator.cpp:55-66: normalised points for the cheirality vote
        const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
        std::vector<Point2f> norm1, norm2;
        norm1.reserve(points1.size());
        norm2.reserve(points2.size());
        for (const auto& p : points1) {
            norm1.push_back({static_cast<float>((p.x - cx) / fx), static_cast<float>((p.y - cy) / fy)});
        }
        for (const auto& p : points2) {
            norm2.push_back({static_cast<float>((p.x - cx) / fx), static_cast<float>((p.y - cy) / fy)});
        }
        simpleRecoverPose(E, norm1, norm2, K, R, t);
        return true;
    }

    // pose_estimator.cpp:70-104
    std::vector<Point3d> triangulatePoints(const std::vector<Keypoint>& keypoints1, const std::vector<Keypoint>& keypoints2,
                                           const std::vector<Match>& matches, const Mat3d& R, const Vec3d& t) const {
        const Mat3d K = m_camera.get().intrinsics();
        const Mat34d T1{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        const Mat34d T2{R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2]};
        auto mul = [&](const Mat34d& T) {
            Mat34d P{};
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 4; ++j) {
                    double s = 0.0;
                    for (int k = 0; k < 3; ++k) {
                        s += K[static_cast<size_t>(i * 3 + k)] * T[static_cast<size_t>(k * 4 + j)];
                    }
                    P[static_cast<size_t>(i * 4 + j)] = s;
                }
            }
            return P;
        };
        std::vector<Point2f> points1, points2;
        points1.reserve(matches.size());
        points2.reserve(matches.size());
        for (const auto& m : matches) {
            points1.push_back({keypoints1[static_cast<size_t>(m.queryIdx)].x, keypoints1[static_cast<size_t>(m.queryIdx)].y});
            points2.push_back({keypoints2[static_cast<size_t>(m.trainIdx)].x, keypoints2[static_cast<size_t>(m.trainIdx)].y});
        }
        return triangulate(mul(T1), mul(T2), points1, points2);
    }

#ifdef SLAMX_HAVE_OPENCV
    // ---- the reference's own signatures (include/slam/frontend/pose_estimator.hpp:15-32) ----
    // src/frontend/pose_estimator.cpp:18-67: fewer than 8 matches, or no essential matrix, leave R and t untouched.
    void estimate(const std::vector<KeyDescriptorPair>& pairs1, const std::vector<KeyDescriptorPair>& pairs2,
                  const std::vector<std::pair<int, int>>& matches, cv::Mat& R, cv::Mat& t) const {
        if (matches.size() < 8) {
            return;  // pose_estimator.cpp:22-26
        }
        std::vector<Keypoint> k1(pairs1.size()), k2(pairs2.size());
        for (size_t i = 0; i < pairs1.size(); ++i) {
            k1[i].x = pairs1[i].first.pt.x;
            k1[i].y = pairs1[i].first.pt.y;
        }
        for (size_t i = 0; i < pairs2.size(); ++i) {
            k2[i].x = pairs2[i].first.pt.x;
            k2[i].y = pairs2[i].first.pt.y;
        }
        Mat3d r{};
        Vec3d tv{};
        if (estimate(k1, k2, matches, r, tv)) {
            detail::to_cv(r, R);
            detail::to_cv(tv, t);
        }
    }

    // src/frontend/pose_estimator.cpp:69-104
    std::vector<cv::Point3d> triangulatePoints(const std::vector<cv::KeyPoint>& keypoints1,
                                               const std::vector<cv::KeyPoint>& keypoints2,
                                               const std::vector<cv::DMatch>& matches, const cv::Mat& R,
                                               const cv::Mat& t) const {
        const Mat3d r = detail::mat3_of(R, "triangulatePoints: R");
        detail::require64(t, 3, 1, "triangulatePoints: t");
        const Vec3d tv{t.at<double>(0, 0), t.at<double>(1, 0), t.at<double>(2, 0)};
        std::vector<Keypoint> k1(keypoints1.size()), k2(keypoints2.size());
        for (size_t i = 0; i < keypoints1.size(); ++i) {
            k1[i].x = keypoints1[i].pt.x;
            k1[i].y = keypoints1[i].pt.y;
        }
        for (size_t i = 0; i < keypoints2.size(); ++i) {
            k2[i].x = keypoints2[i].pt.x;
            k2[i].y = keypoints2[i].pt.y;
        }
        std::vector<Match> m(matches.size());
        for (size_t i = 0; i < matches.size(); ++i) {
            m[i].queryIdx = matches[i].queryIdx;
            m[i].trainIdx = matches[i].trainIdx;
            m[i].distance = matches[i].distance;
        }
        std::vector<cv::Point3d> out;
        out.reserve(m.size());
        for (const Point3d& p : triangulatePoints(k1, k2, m, r, tv)) {
            out.emplace_back(p.x, p.y, p.z);
        }
        return out;
    }
#endif

private:
    std::reference_wrapper<const Camera> m_camera;
};

}  // namespace slam
