// slam/common/camera.hpp -- facade of slam::Camera and slam::triangulate
// (reference: include/slam/common/common.hpp:67-190 and :201-221) over the C ABI.
#pragma once

#include <memory>
#include <vector>

#include <slam/common/common.hpp>
#include <slam/frontend/slamx_runtime.hpp>

namespace slam {

class Camera {
public:
    Camera() = default;
    // common.hpp:76-119: K<i>, D<i>, ImageSize from an OpenCV-style YAML file
    explicit Camera(const std::filesystem::path& configPath, int cameraIndex = 0) {
        detail::check(slamx_camera_load(configPath.string().c_str(), cameraIndex, &m_cam));
    }
    // common.hpp:127-173.  The reference returns a column-major Eigen::MatrixXd; this facade
    // returns the same values row-major.
    RowMajorMatrix<double> undistortImage(const EigenGrayMatrix& rawImage) const {
        RowMajorMatrix<double> out(rawImage.rows(), rawImage.cols());
        detail::check(slamx_undistort(detail::default_context(), &m_cam, rawImage.data(), static_cast<int>(rawImage.rows()),
                                      static_cast<int>(rawImage.cols()), static_cast<std::ptrdiff_t>(rawImage.cols()),
                                      out.data()));
        return out;
    }
    // K as the row-major array the C ABI takes (what the facade's own code uses, with or without Eigen)
    [[nodiscard]] Mat3d intrinsics() const {
        Mat3d K{};
        for (int i = 0; i < 9; ++i) {
            K[static_cast<size_t>(i)] = m_cam.K[i];
        }
        return K;
    }
#ifdef SLAMX_HAVE_EIGEN
    // the reference's own signature (common.hpp:175): callers hand it to cv::eigen2cv
    [[nodiscard]] const Eigen::Matrix3d& getIntrinsicMatrix() const {
        if (!m_eigenK) {
            m_eigenK = std::make_shared<Eigen::Matrix3d>();
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) {
                    (*m_eigenK)(i, j) = m_cam.K[i * 3 + j];
                }
            }
        }
        return *m_eigenK;
    }
#else
    [[nodiscard]] Mat3d getIntrinsicMatrix() const { return intrinsics(); }
#endif
    [[nodiscard]] std::array<double, 5> getDistortionCoefficients() const {
        return {m_cam.k1, m_cam.k2, m_cam.p1, m_cam.p2, m_cam.k3};
    }

private:
    slamx_camera m_cam{};
#ifdef SLAMX_HAVE_EIGEN
    mutable std::shared_ptr<Eigen::Matrix3d> m_eigenK;  // built on first use
#endif
};

// common.hpp:201-221 + the conversion loop of PoseEstimator::triangulatePoints
// (pose_estimator.cpp:94-101): homogeneous float32 points divided by w.
inline std::vector<Point3d> triangulate(const Mat34d& P1, const Mat34d& P2, const std::vector<Point2f>& points1,
                                        const std::vector<Point2f>& points2) {
    const int n = static_cast<int>(points1.size());
    std::vector<double> xyz(static_cast<size_t>(n) * 3);
    detail::check(slamx_triangulate(detail::default_context(), P1.data(), P2.data(),
                                    reinterpret_cast<const float*>(points1.data()),
                                    reinterpret_cast<const float*>(points2.data()), n, nullptr, xyz.data()));
    std::vector<Point3d> out(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
        out[static_cast<size_t>(i)] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    }
    return out;
}

}  // namespace slam
