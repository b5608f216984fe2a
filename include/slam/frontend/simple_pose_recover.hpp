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
