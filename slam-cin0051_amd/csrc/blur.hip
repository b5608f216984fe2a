// blur.hip -- the 5x5 Gaussian blur that precedes orientation / BRIEF.
//
// Reference: gaussianBlur, src/frontend/feature_detector.cpp:315-364 (called with kernelSize 5,
// sigma 1.0 at :33-34): double-precision kernel, direct 25-tap accumulation in (ky outer, kx
// inner) order, one multiply and one add per tap (the reference's Release build has no FMA),
// uint8(std::round(v)); the 2-pixel border is copied from the source image.
//
// Bit-exactness without paying for 25 double MACs per pixel (that version ran at 7 % of HBM
// bandwidth, FP64-issue bound):
//   * fast path: the kernel is separable (w_ij = g_i g_j, g = exp(-i^2/2)/sum), so every pixel is
//     first evaluated in float as 5 vertical + 5 horizontal FMAs.  That estimate is within 3e-4
//     of the reference's double sum (255 * (10 float roundings + the g_i g_j vs w_ij model error),
//     error budget in DESIGN.md), so whenever its fractional part is farther than 6e-4 from .5 the rounded byte
//     is already determined: result = floor(v + 0.5).
//   * exact path (about 0.12 % of the pixels): the reference's 25 IEEE double multiplies and adds
//     in the reference's order (__dmul_rn / __dadd_rn, never contracted), weights computed on
//     the HOST with the same std::exp and division as the reference.
// Either way the byte written is the reference's byte.
//
// Layout: a workgroup owns a 256 x 16 pixel tile staged through LDS as aligned dwords (2-px
// halo); a lane produces 4 horizontally adjacent pixels = one dword store, a wave a 256-pixel row
// segment, so loads and stores are full-width coalesced.  One HBM read + one HBM write per frame.
#include "slamx_internal.h"

#pragma clang fp contract(off)

namespace slamx {

namespace {

constexpr int TILE_W = 256;
constexpr int TILE_H = 16;
constexpr int HALO = 2;
constexpr int LDS_ROW_DW = TILE_W / 4 + 2;   // columns [x0-4, x0+260) as dwords
constexpr int LDS_ROWS = TILE_H + 2 * HALO;  // 20

__device__ __forceinline__ float ub(uint32_t w, int k) {  // byte k of w as float (v_cvt_f32_ubyteK)
    return static_cast<float>((w >> (8 * k)) & 0xffu);
}

__global__ __launch_bounds__(256) void blur_kernel(BatchView v, const double* __restrict__ weights) {
    __shared__ uint32_t tile[LDS_ROWS * LDS_ROW_DW];
    __shared__ double wgt[25];
    __shared__ float g[5];
    const int tx = blockIdx.x;
    const int ty = blockIdx.y;
    const int f = blockIdx.z;
    const uint8_t* img = v.frames + static_cast<size_t>(f) * v.frame_stride;
    uint8_t* out = v.blurred + static_cast<size_t>(f) * v.frame_stride;
    const int x0 = tx * TILE_W - 4;
    const int y0 = ty * TILE_H - HALO;
    if (threadIdx.x < 25) {
        wgt[threadIdx.x] = weights[threadIdx.x];
    }
    if (threadIdx.x < 5) {
        // separable factor: w[i][2] / sqrt(w[2][2]) == exp(-i^2/2) / sum(exp(-k^2/2))
        g[threadIdx.x] = static_cast<float>(weights[threadIdx.x * 5 + 2] / sqrt(weights[12]));
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
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int x = tx * TILE_W + lane * 4;  // first of this lane's 4 pixels
    if (x >= v.cols) {
        return;
    }
    const float g0 = g[0], g1 = g[1], g2 = g[2];  // g[3] == g[1], g[4] == g[0]
    const uint8_t* t8 = reinterpret_cast<const uint8_t*>(tile);
    // rolling window of 5 input rows x 8 columns (x-2 .. x+5), converted to float once per row
    float win[5][8];
    auto load_row = [&](int lds_row, float* dst) {
        const uint32_t* rp = tile + lds_row * LDS_ROW_DW + lane;  // dwords covering x-4 .. x+7
        const uint32_t a = rp[0], b = rp[1], c = rp[2];
        dst[0] = ub(a, 2);
        dst[1] = ub(a, 3);
        dst[2] = ub(b, 0);
        dst[3] = ub(b, 1);
        dst[4] = ub(b, 2);
        dst[5] = ub(b, 3);
        dst[6] = ub(c, 0);
        dst[7] = ub(c, 1);
    };
    const int ry0 = wave * (TILE_H / 4);  // this wave's 4 output rows: ry0 .. ry0+3 (tile-relative)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        load_row(ry0 + r, win[r]);  // LDS rows ry0 .. ry0+3 are image rows y-2 .. y+1 of output row ry0
    }
#pragma unroll
    for (int k = 0; k < TILE_H / 4; ++k) {
        const int ry = ry0 + k;
        const int y = ty * TILE_H + ry;
        load_row(ry + 4, win[(k + 4) % 5]);
        if (y >= v.rows) {
            break;  // wave-uniform
        }
        const float* rm2 = win[(k + 0) % 5];
        const float* rm1 = win[(k + 1) % 5];
        const float* rc = win[(k + 2) % 5];
        const float* rp1 = win[(k + 3) % 5];
        const float* rp2 = win[(k + 4) % 5];
        float h[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            h[c] = fmaf(g0, rm2[c] + rp2[c], fmaf(g1, rm1[c] + rp1[c], g2 * rc[c]));
        }
        uint32_t packed = 0;
        const bool row_inside = y >= HALO && y < v.rows - HALO;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int xp = x + p;
            uint32_t res;
            if (row_inside && xp >= HALO && xp < v.cols - HALO) {
                const float est = fmaf(g0, h[p] + h[p + 4], fmaf(g1, h[p + 1] + h[p + 3], g2 * h[p + 2]));
                const float fl = floorf(est);
                const float fr = est - fl;
                if (fabsf(fr - 0.5f) > 6.0e-4f) {
                    res = static_cast<uint32_t>(static_cast<int>(floorf(est + 0.5f)));
                } else {
                    // too close to a rounding boundary for the float estimate: the reference's sum
                    const uint8_t* pc = t8 + (ry + HALO) * (LDS_ROW_DW * 4) + (lane * 4 + p + 4);
                    double acc = 0.0;
#pragma unroll
                    for (int ky = -2; ky <= 2; ++ky) {
#pragma unroll
                        for (int kx = -2; kx <= 2; ++kx) {
                            const double px = static_cast<double>(pc[ky * (LDS_ROW_DW * 4) + kx]);
                            acc = __dadd_rn(acc, __dmul_rn(px, wgt[(ky + 2) * 5 + (kx + 2)]));
                        }
                    }
                    res = static_cast<uint32_t>(static_cast<int>(round(acc)));
                }
            } else {
                res = static_cast<uint32_t>(rc[p + 2]);  // border copied from the source (:355-361)
            }
            packed |= (res & 0xffu) << (8 * p);
        }
        uint8_t* dst = out + static_cast<size_t>(y) * v.pitch + x;
        if (x + 4 <= v.cols) {
            *reinterpret_cast<uint32_t*>(dst) = packed;
        } else {
            for (int p = 0; p < v.cols - x; ++p) {
                dst[p] = static_cast<uint8_t>(packed >> (8 * p));
            }
        }
    }
}

}  // namespace

void launch_blur(const BatchView& v, const double* weights_dev, int nframes, hipStream_t stream) {
    dim3 grid((v.cols + TILE_W - 1) / TILE_W, (v.rows + TILE_H - 1) / TILE_H, nframes);
    hipLaunchKernelGGL(blur_kernel, grid, dim3(256), 0, stream, v, weights_dev);
}

}  // namespace slamx
