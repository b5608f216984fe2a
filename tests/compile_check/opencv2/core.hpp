// NOT OpenCV: the API subset the facade's SLAMX_USE_OPENCV branches use (see ../README.md).  Test infrastructure.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6

namespace cv {

template <typename T>
struct Point_ {
    T x{}, y{};
    Point_() = default;
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
using Point2f = Point_<float>;
using Point2d = Point_<double>;

template <typename T>
struct Point3_ {
    T x{}, y{}, z{};
    Point3_() = default;
    Point3_(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
};
using Point3d = Point3_<double>;
using Point3f = Point3_<float>;

// dense, continuous, single channel; copies share the buffer (as cv::Mat headers do)
class Mat {
public:
    int rows = 0, cols = 0;
    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && buf_) {
            return;
        }
        rows = r;
        cols = c;
        type_ = type;
        buf_ = std::make_shared<std::vector<uint8_t>>(static_cast<size_t>(r) * c * elemSize(), uint8_t(0));
    }
    int type() const { return type_; }
    bool empty() const { return rows == 0 || cols == 0 || !buf_; }
    size_t elemSize() const { return type_ == CV_64F ? 8 : (type_ == CV_32F ? 4 : 1); }
    template <typename T>
    T& at(int i, int j) { return reinterpret_cast<T*>(buf_->data())[static_cast<size_t>(i) * cols + j]; }
    template <typename T>
    const T& at(int i, int j) const { return reinterpret_cast<const T*>(buf_->data())[static_cast<size_t>(i) * cols + j]; }
    template <typename T>
    T& at(int i) { return reinterpret_cast<T*>(buf_->data())[i]; }
    template <typename T>
    const T& at(int i) const { return reinterpret_cast<const T*>(buf_->data())[i]; }

private:
    int type_ = CV_8U;
    std::shared_ptr<std::vector<uint8_t>> buf_;
};

struct KeyPoint {
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
    KeyPoint() = default;
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};

struct DMatch {
    int queryIdx = -1, trainIdx = -1, imgIdx = -1;
    float distance = 0;
    DMatch() = default;
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
};

}  // namespace cv
