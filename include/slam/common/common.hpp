// slam/common/common.hpp -- facade counterpart of the reference's include/slam/common/common.hpp
// for the hot path: the row-major uint8 matrix type the detector and matcher exchange.
//
// The reference aliases Eigen::Matrix<unsigned char, Dynamic, Dynamic, RowMajor> as
// slam::EigenGrayMatrix (common.hpp:61-62) and uses only rows()/cols()/data()/operator()/
// resize() of it on this path.  This header provides exactly that subset without Eigen (Eigen is
// a Conan dependency of the reference and is absent here); when <Eigen/Dense> is available and
// SLAMX_USE_EIGEN is defined the real Eigen types are used instead, so reference call sites
// compile unchanged.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <vector>

#if defined(SLAMX_USE_EIGEN) && __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
namespace slam {
using EigenGrayMatrix = Eigen::Matrix<unsigned char, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
using DescriptorMatrix = Eigen::Matrix<uint8_t, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
}  // namespace slam
#else
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

using EigenGrayMatrix = RowMajorMatrix<unsigned char>;  // common.hpp:61-62
using DescriptorMatrix = RowMajorMatrix<uint8_t>;       // feature_detector.hpp:45

}  // namespace slam
#endif

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
}  // namespace detail

}  // namespace slam
