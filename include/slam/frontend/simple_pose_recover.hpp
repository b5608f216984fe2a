// slam/frontend/simple_pose_recover.hpp -- facade of the reference's free function
// simpleRecoverPose (include/slam/frontend/simple_pose_recover.hpp:17-19,
// src/frontend/simple_pose_recover.cpp:35-98), in the global namespace like the reference, with
// plain value types instead of cv::Mat / cv::Point2f.
#pragma once

#include <vector>

#include <slam/common/common.hpp>
#include <slam/frontend/slamx_runtime.hpp>

// E: essential matrix, points: NORMALISED image points, K: intrinsics -> R, t of the candidate
// with the most points in front of both cameras (first maximum wins).
inline void simpleRecoverPose(const slam::Mat3d& E, const std::vector<slam::Point2f>& points1,
                              const std::vector<slam::Point2f>& points2, const slam::Mat3d& K, slam::Mat3d& R,
                              slam::Vec3d& t) {
    slam::detail::check(slamx_recover_pose(slam::detail::default_context(), E.data(),
                                           reinterpret_cast<const float*>(points1.data()),
                                           reinterpret_cast<const float*>(points2.data()),
                                           static_cast<int>(points1.size()), K.data(), R.data(), t.data(), nullptr));
}

#ifdef SLAMX_HAVE_OPENCV
// The reference's own signature (include/slam/frontend/simple_pose_recover.hpp:17-19): E and K 3x3 CV_64F, points
// normalised; R (3x3) and t (3x1) are (re)created as CV_64F, as cv::Mat_<double> results are in the reference
// (src/frontend/simple_pose_recover.cpp:84-97).
inline void simpleRecoverPose(const cv::Mat& E, const std::vector<cv::Point2f>& points1, const std::vector<cv::Point2f>& points2,
                              const cv::Mat& K, cv::Mat& R, cv::Mat& t) {
    const slam::Mat3d e = slam::detail::mat3_of(E, "simpleRecoverPose: E");
    const slam::Mat3d k = slam::detail::mat3_of(K, "simpleRecoverPose: K");
    slam::Mat3d r{};
    slam::Vec3d tv{};
    slam::detail::check(slamx_recover_pose(slam::detail::default_context(), e.data(),
                                           reinterpret_cast<const float*>(points1.data()),
                                           reinterpret_cast<const float*>(points2.data()),
                                           static_cast<int>(points1.size()), k.data(), r.data(), tv.data(), nullptr));
    slam::detail::to_cv(r, R);
    slam::detail::to_cv(tv, t);
}
#endif
