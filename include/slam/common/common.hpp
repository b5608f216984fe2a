// slam/common/common.hpp -- facade counterpart of the reference's include/slam/common/common.hpp
// for the hot path: the row-major uint8 matrix type the detector and matcher exchange.
//
// The reference aliases Eigen::Matrix<unsigned char, Dynamic, Dynamic, RowMajor> as
// slam::EigenGrayMatrix (common.hpp:61-62) and uses only rows()/cols()/data()/operator()/
// resize() of it on this path.  This header provides exactly that subset without Eigen (Eigen is
// a Conan dependency of the reference and is absent here); when <Eigen/Dense> is available and
// SLAMX_USE_EIGEN is defined the real Eigen types are aliased instead, so reference call sites
// compile unchanged.  (Compiled in this image only against tests/compile_check/Eigen/Dense, an API-subset
// header written for that purpose: INTEGRATION.md section 2.)
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <vector>

// Reference-typed overloads of the pose / backend facade (exact signatures of the reference's headers, next to the
// plain-value-type versions this image compiles): switched on per dependency, and only when its headers exist.
//   -DSLAMX_USE_OPENCV + <opencv2/core.hpp>  -> cv::Mat / cv::Point2f / cv::KeyPoint / cv::DMatch overloads
//   -DSLAMX_USE_EIGEN  + <Eigen/Dense>       -> Eigen::Vector3d / Matrix4d overloads (and the matrix aliases above)
// Neither library is in this image (Conan packages of the reference, no network): INTEGRATION.md section 2 says
// which branches have only ever been compiled against tests/compile_check/'s API-subset headers.
#if defined(SLAMX_USE_OPENCV) && __has_include(<opencv2/core.hpp>)
#define SLAMX_HAVE_OPENCV 1
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#endif
#if defined(SLAMX_USE_EIGEN) && __has_include(<Eigen/Dense>)
#define SLAMX_HAVE_EIGEN 1
#include <Eigen/Dense>
#endif

namespace slam {

// Minimal row-major dense matrix (the Eigen subset used on the hot path).
template <typename T>
class RowMajorMatrix {
public:
    using Index = std::ptrdiff_t;
    RowMajorMatrix() = default;
    RowMajorMatrix(Index rows, Index cols) { resize(rows, cols); }
    void resize(Index rows, Index cols) {
        if (rows < 0 || cols < 0) {
            throw std::invalid_argument("negative matrix dimension");
        }
        rows_ = rows;
        cols_ = cols;
        data_.assign(static_cast<size_t>(rows) * static_cast<size_t>(cols), T());
    }
    Index rows() const { return rows_; }
    Index cols() const { return cols_; }
    Index size() const { return rows_ * cols_; }
    T* data() { return data_.data(); }
    const T* data() const { return data_.data(); }
    T& operator()(Index r, Index c) { return data_[static_cast<size_t>(r) * static_cast<size_t>(cols_) + static_cast<size_t>(c)]; }
    const T& operator()(Index r, Index c) const {
        return data_[static_cast<size_t>(r) * static_cast<size_t>(cols_) + static_cast<size_t>(c)];
    }
    static RowMajorMatrix Zero(Index rows, Index cols) { return RowMajorMatrix(rows, cols); }

private:
    Index rows_ = 0, cols_ = 0;
    std::vector<T> data_;
};

#ifdef SLAMX_HAVE_EIGEN
// the reference's own types (common.hpp:61-62, feature_detector.hpp:45): row-major, contiguous -- .data() is the
// pointer the C ABI takes
using EigenGrayMatrix = Eigen::Matrix<unsigned char, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
using DescriptorMatrix = Eigen::Matrix<uint8_t, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
#else
using EigenGrayMatrix = RowMajorMatrix<unsigned char>;  // common.hpp:61-62
using DescriptorMatrix = RowMajorMatrix<uint8_t>;       // feature_detector.hpp:45
#endif

}  // namespace slam

#include <array>
#include <filesystem>
#include <string>

#include <slamx.h>


namespace slam {
namespace constants {
constexpr double COLOR_RANGE = 255.0;  // common.hpp:28
}  // namespace constants

// Plain stand-ins for the OpenCV / Eigen value types the reference's pose interfaces exchange
// (cv::Point2f, cv::Point3d, 3x3 / 3x4 double matrices); all row-major.
struct Point2f {
    float x{}, y{};
};
struct Point3d {
    double x{}, y{}, z{};
};
using Mat3d = std::array<double, 9>;
using Mat34d = std::array<double, 12>;
using Vec3d = std::array<double, 3>;

namespace detail {
void check(int rc);
slamx_ctx* default_context();

#ifdef SLAMX_HAVE_OPENCV
static_assert(sizeof(cv::Point2f) == sizeof(Point2f), "cv::Point2f is two floats: vectors of it go to the C ABI as they are");
// a 3x3 (or 3x1) CV_64F cv::Mat <-> the row-major arrays of the C ABI; other element types are refused loudly
inline void require64(const cv::Mat& m, int rows, int cols, const char* what) {
    if (m.rows != rows || m.cols != cols || m.type() != CV_64F) {
        throw std::invalid_argument(std::string(what) + ": expected a " + std::to_string(rows) + "x" + std::to_string(cols) +
                                    " CV_64F matrix");
    }
}
inline Mat3d mat3_of(const cv::Mat& m, const char* what) {
    require64(m, 3, 3, what);
    Mat3d out{};
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            out[static_cast<size_t>(i * 3 + j)] = m.at<double>(i, j);
        }
    }
    return out;
}
inline void to_cv(const Mat3d& a, cv::Mat& m) {
    m.create(3, 3, CV_64F);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            m.at<double>(i, j) = a[static_cast<size_t>(i * 3 + j)];
        }
    }
}
inline void to_cv(const Vec3d& a, cv::Mat& m) {
    m.create(3, 1, CV_64F);
    for (int i = 0; i < 3; ++i) {
        m.at<double>(i, 0) = a[static_cast<size_t>(i)];
    }
}
#endif
}  // namespace detail

#ifdef SLAMX_HAVE_OPENCV
// include/slam/common/common.hpp:56 of the reference
using KeyDescriptorPair = std::pair<cv::KeyPoint, cv::Mat>;
#endif

}  // namespace slam
