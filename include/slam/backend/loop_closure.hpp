// slam/backend/loop_closure.hpp -- facade of the geometric-verification core of
// slam::LoopClosure (reference: src/backend/loop_closure.cpp:153-274).  The bag-of-words candidate
// search (fbow, vocabulary file absent from the reference tree) is out of scope; what is provided
// is what runs after a candidate is chosen: the 6-point DLT PnP and its RANSAC loop.
#pragma once

#include <cstdint>
#include <vector>

#include <slam/common/common.hpp>
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

}  // namespace slam
