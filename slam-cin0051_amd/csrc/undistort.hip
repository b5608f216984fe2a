// undistort.hip -- Camera::undistortImage as a gather kernel.
//
// Reference: include/slam/common/common.hpp:127-173.  For every output pixel (i, j) the distorted
// source position is evaluated in double with the reference's expression order (k1, k2, p1, p2;
// k3 unused), rounded with std::round, and the [0,1]-scaled source pixel (u8 / 255.0) is fetched,
// 0 outside.  One lane per output pixel; the output stream (8 B/pixel) dominates the traffic.
// r.pow(4) is evaluated as (r*r)*(r*r) here and in the oracle (Eigen's vectorised pow is
// third-party arithmetic: parity unpinned by <= 1 ulp of r^4).
#include "slamx_internal.h"

#pragma clang fp contract(off)

namespace slamx {

namespace {

__global__ __launch_bounds__(256) void undistort_kernel(const uint8_t* __restrict__ img, int rows, int cols, int pitch,
                                                        slamx_camera cam, double* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= cols) {
        return;
    }
    const double u = static_cast<double>(j);
    const double v = static_cast<double>(i);
    const double x = (u - cam.cx) / cam.fx;
    const double y = (v - cam.cy) / cam.fy;
    const double r = sqrt(x * x + y * y);
    const double r2 = r * r;
    const double r4 = r2 * r2;
    const double radial = (1 + cam.k1 * r2) + cam.k2 * r4;
    const double xDist = (x * radial + ((2 * cam.p1) * x) * y) + cam.p2 * (r2 + 2 * (x * x));
    const double yDist = (y * radial + ((2 * cam.p2) * x) * y) + cam.p1 * (r2 + 2 * (y * y));
    const double uD = cam.fx * xDist + cam.cx;
    const double vD = cam.fy * yDist + cam.cy;
    const int ui = static_cast<int>(round(uD));
    const int vi = static_cast<int>(round(vD));
    double val = 0.0;
    if (ui >= 0 && vi >= 0 && ui < cols && vi < rows) {
        val = static_cast<double>(img[static_cast<size_t>(vi) * pitch + ui]) / 255.0;
    }
    out[static_cast<size_t>(i) * cols + j] = val;
}

}  // namespace

void launch_undistort(const uint8_t* img, int rows, int cols, int pitch, const slamx_camera& cam, double* out,
                      hipStream_t stream) {
    hipLaunchKernelGGL(undistort_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, stream, img, rows, cols, pitch, cam,
                       out);
}

}  // namespace slamx
