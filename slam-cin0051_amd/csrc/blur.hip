// blur.hip -- the 5x5 Gaussian blur that precedes orientation / BRIEF.
//
// Reference: gaussianBlur, src/frontend/feature_detector.cpp:315-364 (called with kernelSize 5,
// sigma 1.0 at :33-34): double-precision kernel, direct 25-tap accumulation in (ky outer, kx
// inner) order, one multiply and one add per tap (the reference's Release build has no FMA),
// uint8(std::round(v)); the 2-pixel border is copied from the source image.
//
// Bit-exactness: same 25 IEEE double multiplies and adds in the same order (__dmul_rn /
// __dadd_rn are never contracted), weights computed on the HOST with the same std::exp and
// division as the reference and passed in.  One HBM read + one HBM write of the frame; the
// 64x16 tile (+2 halo) is staged through LDS as aligned dwords.
#include "slamx_internal.h"

namespace slamx {

namespace {

constexpr int TILE_W = 64;
constexpr int TILE_H = 16;
constexpr int HALO = 2;
constexpr int LDS_ROW_DW = 18;  // 72 bytes: columns [64wx-4, 64wx+68)
constexpr int LDS_ROWS = TILE_H + 2 * HALO;

__global__ __launch_bounds__(256) void blur_kernel(BatchView v, const double* __restrict__ weights) {
    __shared__ uint32_t tile[LDS_ROWS * LDS_ROW_DW];
    __shared__ double wgt[25];
    const int wx = blockIdx.x;
    const int ty = blockIdx.y;
    const int f = blockIdx.z;
    const uint8_t* img = v.frames + static_cast<size_t>(f) * v.frame_stride;
    uint8_t* out = v.blurred + static_cast<size_t>(f) * v.frame_stride;
    const int x0 = wx * TILE_W - 4;
    const int y0 = ty * TILE_H - HALO;
    if (threadIdx.x < 25) {
        wgt[threadIdx.x] = weights[threadIdx.x];
    }
    for (int i = threadIdx.x; i < LDS_ROWS * LDS_ROW_DW; i += 256) {
        const int r = i / LDS_ROW_DW;
        const int c = i - r * LDS_ROW_DW;
        const int yy = y0 + r;
        const int xx = x0 + c * 4;
        uint32_t val = 0;
        if (yy >= 0 && yy < v.rows && xx >= 0 && xx + 4 <= v.pitch) {
            val = *reinterpret_cast<const uint32_t*>(img + static_cast<size_t>(yy) * v.pitch + xx);
        }
        tile[i] = val;
    }
    __syncthreads();
    const uint8_t* t8 = reinterpret_cast<const uint8_t*>(tile);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int x = wx * TILE_W + lane;
    if (x >= v.cols) {
        return;
    }
#pragma unroll
    for (int k = 0; k < TILE_H / 4; ++k) {
        const int ry = wave * (TILE_H / 4) + k;
        const int y = ty * TILE_H + ry;
        if (y >= v.rows) {
            break;
        }
        const uint8_t* p = t8 + (ry + HALO) * (LDS_ROW_DW * 4) + (lane + 4);
        uint8_t res;
        if (x >= HALO && x < v.cols - HALO && y >= HALO && y < v.rows - HALO) {
            double acc = 0.0;
#pragma unroll
            for (int ky = -2; ky <= 2; ++ky) {
#pragma unroll
                for (int kx = -2; kx <= 2; ++kx) {
                    const double px = static_cast<double>(p[ky * (LDS_ROW_DW * 4) + kx]);
                    acc = __dadd_rn(acc, __dmul_rn(px, wgt[(ky + 2) * 5 + (kx + 2)]));
                }
            }
            res = static_cast<uint8_t>(static_cast<int>(round(acc)));
        } else {
            res = p[0];  // border copied from the source (:355-361)
        }
        out[static_cast<size_t>(y) * v.pitch + x] = res;
    }
}

}  // namespace

void launch_blur(const BatchView& v, const double* weights_dev, int nframes, hipStream_t stream) {
    dim3 grid(v.nw, (v.rows + TILE_H - 1) / TILE_H, nframes);
    hipLaunchKernelGGL(blur_kernel, grid, dim3(256), 0, stream, v, weights_dev);
}

}  // namespace slamx
