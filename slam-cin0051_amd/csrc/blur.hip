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
//     first evaluated in float as 5 vertical + 5 horizontal FMAs.  That estimate is within 1.3e-4
//     of the reference's double sum (error budget in DESIGN.md), so whenever its fractional part
//     is farther than 6e-4 from .5 the rounded byte is already determined: round-to-nearest(v).
//   * exact path (about 0.12 % of the pixels): the reference's 25 IEEE double multiplies and adds
//     in the reference's order (__dmul_rn / __dadd_rn, never contracted), weights computed on
//     the HOST with the same std::exp and division as the reference.  These pixels are collected
//     in an LDS list and patched by a dense second pass, so the main loop never diverges.
// Either way the byte written is the reference's byte.
//
// Layout: a workgroup owns a 248 x 32 pixel tile: lane l of a wave owns the dword of columns
// X0-4+4l .. X0-1+4l, so lanes 1..62 produce output and lanes 0 / 63 are the halo columns.  A lane
// runs the vertical pass on its own 4 columns only and receives the two neighbouring column pairs
// it needs for the horizontal pass from lanes l-1 / l+1 through DPP wave shifts -- no column is
// filtered twice.  Results go through an LDS output tile (patched there by the exact pass) and
// leave as coalesced dword stores.  One HBM read + one HBM write per frame.
#include "slamx_internal.h"

#pragma clang fp contract(off)

namespace slamx {

namespace {

constexpr int TILE_W = 248;                  // output columns per tile (62 lanes x 4)
constexpr int TILE_H = 32;
constexpr int HALO = 2;
constexpr int LDS_ROW_DW = 64;               // columns [X0-4, X0+252) as dwords, one per lane
constexpr int LDS_ROWS = TILE_H + 2 * HALO;  // 36
constexpr int ROWS_PER_WAVE = TILE_H / 4;
constexpr int FIX_PER_WAVE = 512;            // uncertain dwords listed per wave: at most ROWS_PER_WAVE x 62 = 496, a list cannot overflow
static_assert(ROWS_PER_WAVE * 62 <= FIX_PER_WAVE, "every output dword of a wave must fit its list");
// |frac(estimate) - 0.5| <= 6e-4 sends a pixel to the exact pass; tested as
// frac(estimate - UNC_LO) <= UNC_SPAN, widened by 5e-5 for the rounding of that subtraction
constexpr float UNC_LO = 0.5f - 6.5e-4f;
constexpr float UNC_SPAN = 1.3e-3f;

// the reference's sum for the pixel whose centre byte is at pc in the LDS tile (wgt: the kernel argument, so the 25
// weights are scalar loads -- as an LDS array the compiler kept all of them in 50 vector registers across the pass)
__device__ __forceinline__ uint32_t exact_pixel(const uint8_t* pc, const double* wgt) {
    double acc = 0.0;
#pragma unroll
    for (int ky = -2; ky <= 2; ++ky) {
#pragma unroll
        for (int kx = -2; kx <= 2; ++kx) {
            const double px = static_cast<double>(pc[ky * (LDS_ROW_DW * 4) + kx]);
            acc = __dadd_rn(acc, __dmul_rn(px, wgt[(ky + 2) * 5 + (kx + 2)]));
        }
    }
    return static_cast<uint32_t>(static_cast<int>(round(acc)));
}

typedef float f2 __attribute__((ext_vector_type(2)));

// lane i <- lane i-1 / lane i+1 of the wave (lanes 0 / 63 receive 0)
__device__ __forceinline__ float from_left(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x138, 0xf, 0xf, true));  // bound_ctrl: 0 shifted in
}
__device__ __forceinline__ float from_right(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x130, 0xf, 0xf, true));
}

__global__ __launch_bounds__(256) void blur_kernel(BatchView v, const double* __restrict__ weights, int nframes, int tiles_x,
                                                    int tiles_y, uint32_t tiles_x_recip) {
    __shared__ uint32_t tile[LDS_ROWS * LDS_ROW_DW];
    __shared__ uint32_t outt[TILE_H * LDS_ROW_DW];
    // Uncertain pixels, one entry per output dword that has any: row << 10 | lane << 4 | mask of its bytes.  Every wave
    // keeps a list of its own with the count in a scalar register -- no LDS atomic, no round trip in the main loop.
    __shared__ uint16_t fix[4][FIX_PER_WAVE];
    __shared__ int nfix[4];
    __shared__ float g[5];
    // all tiles of a frame on ONE XCD (frame_block): its L2 serves the halo rows shared by vertically adjacent
    // tiles AND the 128-byte lines that horizontally adjacent tiles share (a tile row starts at x0 - 4: with a tile
    // column per XCD every L2 fetched its own copy of those lines, 1.43x the frame bytes)
    int f, tile_id;
    if (!frame_block(nframes, tiles_x * tiles_y, &f, &tile_id)) {
        return;
    }
    const int ty = tiles_x_recip != 0 ? static_cast<int>(__umulhi(static_cast<uint32_t>(tile_id), tiles_x_recip)) : tile_id;
    const int tx = tile_id - ty * tiles_x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));  // scalar: row addresses on the SALU
    const uint8_t* img = v.frames + static_cast<size_t>(f) * v.frame_stride;
    uint8_t* out = v.blurred + static_cast<size_t>(f) * v.frame_stride;
    const int x = tx * TILE_W - 4 + lane * 4;  // first of this lane's 4 columns
    const int y0 = ty * TILE_H - HALO;
    if (threadIdx.x < 5) {
        // separable factor: w[i][2] / sqrt(w[2][2]) == exp(-i^2/2) / sum(exp(-k^2/2))
        g[threadIdx.x] = static_cast<float>(weights[threadIdx.x * 5 + 2] / sqrt(weights[12]));
    }
    {
        // a wave stages whole rows, lane -> dword column; all loads are issued before the first LDS store
        constexpr int NLOAD = LDS_ROWS / 4;
        static_assert(LDS_ROWS % 4 == 0, "rows are dealt to 4 waves");
        // Buffer loads: row offset in a scalar register, column offset in the lane (no 64-bit vector address
        // math); what lies outside the frame is zero-filled by the range check -- lanes left / right of it carry
        // an offset beyond every buffer, rows above / below it go through a descriptor of 0 bytes.
        const int frame_bytes = v.rows * v.pitch;
        const uint32_t xoff = x >= 0 && x + 4 <= v.pitch ? static_cast<uint32_t>(x) : 0xffffffffu;
        uint32_t va[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int yy = y0 + wave + 4 * i;  // wave-uniform
            const bool row_ok = yy >= 0 && yy < v.rows;
            const __amdgpu_buffer_rsrc_t rows_of_frame =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, row_ok ? frame_bytes : 0, 0x00020000);
            va[i] = __builtin_amdgcn_raw_buffer_load_b32(rows_of_frame, xoff, (row_ok ? yy : 0) * v.pitch, 0);
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            tile[(wave + 4 * i) * LDS_ROW_DW + lane] = va[i];
        }
    }
    __syncthreads();
    const float g0 = g[0], g1 = g[1], g2 = g[2];  // g[3] == g[1], g[4] == g[0]
    // bytes of the output dword that are image border (columns < 2 or >= cols - 2): copied (:355-361)
    uint32_t keep = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (x + p < HALO || x + p >= v.cols - HALO) {
            keep |= 0xffu << (8 * p);
        }
    }
    const bool writer = lane >= 1 && lane <= 62 && x < v.cols;
    const uint32_t keep4 = (keep & 1u) | (keep >> 7 & 2u) | (keep >> 14 & 4u) | (keep >> 21 & 8u);
    int listed_here = 0;  // (wave-uniform) entries in this wave's list
    {
        // rolling window of 5 input rows x this lane's 4 columns, converted once per row
        // (v_cvt_f32_ubyte0..3).  Scalar VOP2 float math on purpose: v_add_f32 / v_fmac_f32 issue at
        // twice the rate of the packed VOP3P forms, which makes them cheaper than v_pk_* even per lane
        // (tools/microbench/op_rates.hip).
        float win[5][4];
        uint32_t centre[5];
        auto load_row = [&](int lds_row, float* dst, uint32_t* mid) {
            const uint32_t w = tile[lds_row * LDS_ROW_DW + lane];
            asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(dst[0]) : "v"(w));
            asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(dst[1]) : "v"(w));
            asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(dst[2]) : "v"(w));
            asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(dst[3]) : "v"(w));
            *mid = w;
        };
        const int ry0 = wave * ROWS_PER_WAVE;  // this wave's output rows ry0 .. ry0+RPW-1 (tile-relative)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            load_row(ry0 + r, win[r], &centre[r]);  // LDS rows ry0 .. ry0+3 are image rows y-2 .. y+1 of row ry0
        }
#pragma unroll
        for (int k = 0; k < ROWS_PER_WAVE; ++k) {
            const int ry = ry0 + k;
            const int y = ty * TILE_H + ry;
            load_row(ry + 4, win[(k + 4) % 5], &centre[(k + 4) % 5]);
            if (y >= v.rows) {
                break;  // wave-uniform
            }
            const uint32_t src = centre[(k + 2) % 5];
            uint32_t packed = src;
            if (y >= HALO && y < v.rows - HALO) {  // wave-uniform
                const float* rm2 = win[(k + 0) % 5];
                const float* rm1 = win[(k + 1) % 5];
                const float* rc = win[(k + 2) % 5];
                const float* rp1 = win[(k + 3) % 5];
                const float* rp2 = win[(k + 4) % 5];
                // Column pairs as 64-bit operands: v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 do two columns per
                // instruction with the roundings of the scalar forms.  Vertical pass of columns x-2 .. x+5 as the
                // pairs h01, h23, h45, h67; h23 and h45 are this lane's own.
                const f2 G0 = {g0, g0}, G1 = {g1, g1}, G2 = {g2, g2};
                const f2 h23 = __builtin_elementwise_fma(
                    G0, f2{rm2[0], rm2[1]} + f2{rp2[0], rp2[1]},
                    __builtin_elementwise_fma(G1, f2{rm1[0], rm1[1]} + f2{rp1[0], rp1[1]}, G2 * f2{rc[0], rc[1]}));
                const f2 h45 = __builtin_elementwise_fma(
                    G0, f2{rm2[2], rm2[3]} + f2{rp2[2], rp2[3]},
                    __builtin_elementwise_fma(G1, f2{rm1[2], rm1[3]} + f2{rp1[2], rp1[3]}, G2 * f2{rc[2], rc[3]}));
                const f2 h01 = {from_left(h45.x), from_left(h45.y)};
                const f2 h67 = {from_right(h23.x), from_right(h23.y)};
                // horizontal pass: outputs (0, 1) and (2, 3); the odd-distance sums straddle the pairs and stay scalar
                const f2 odd01 = {h01.y + h23.y, h23.x + h45.x};
                const f2 odd23 = {h23.y + h45.y, h45.x + h67.x};
                const f2 est01 = __builtin_elementwise_fma(G0, h01 + h45, __builtin_elementwise_fma(G1, odd01, G2 * h23));
                const f2 est23 = __builtin_elementwise_fma(G0, h23 + h67, __builtin_elementwise_fma(G1, odd23, G2 * h45));
                // est >= 0; round-to-nearest equals floor(est + 0.5) away from the .5 boundary
                uint32_t acc = 0;
                acc = __builtin_amdgcn_cvt_pk_u8_f32(est01.x, 0, acc);
                acc = __builtin_amdgcn_cvt_pk_u8_f32(est01.y, 1, acc);
                acc = __builtin_amdgcn_cvt_pk_u8_f32(est23.x, 2, acc);
                acc = __builtin_amdgcn_cvt_pk_u8_f32(est23.y, 3, acc);
                const f2 UNC = {UNC_LO, UNC_LO};
                const f2 d01 = est01 - UNC, d23 = est23 - UNC;
                float w[4];
                w[0] = __builtin_amdgcn_fractf(d01.x);
                w[1] = __builtin_amdgcn_fractf(d01.y);
                w[2] = __builtin_amdgcn_fractf(d23.x);
                w[3] = __builtin_amdgcn_fractf(d23.y);
                packed = (src & keep) | (acc & ~keep);
                const bool some = writer && fminf(fminf(w[0], w[1]), fminf(w[2], w[3])) <= UNC_SPAN;
                if (__ballot(some) != 0) {  // about a quarter of the wave-rows
                    uint32_t m = 0;
                    if (some) {
                        m = ((w[0] <= UNC_SPAN ? 1u : 0u) | (w[1] <= UNC_SPAN ? 2u : 0u) | (w[2] <= UNC_SPAN ? 4u : 0u) |
                             (w[3] <= UNC_SPAN ? 8u : 0u)) & ~keep4;
                    }
                    const unsigned long long b = __ballot(m != 0);
                    if (m != 0) {
                        const int pos = listed_here + static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(b >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(b), 0u)));
                        fix[wave][pos] = static_cast<uint16_t>(ry << 10 | lane << 4 | static_cast<int>(m));
                    }
                    listed_here += __popcll(b);
                }
            }
            outt[ry * LDS_ROW_DW + lane] = packed;
        }
    }
    if (lane == 0) {
        nfix[wave] = listed_here;
    }
    __syncthreads();
    // ---- exact pass on the listed pixels (every pixel of the tile when all of them are uncertain: two-level images)
    const uint8_t* t8 = reinterpret_cast<const uint8_t*>(tile);
    uint8_t* o8 = reinterpret_cast<uint8_t*>(outt);
    const int n0 = nfix[0], n1 = nfix[1], n2 = nfix[2], n3 = nfix[3];
    const int total = n0 + n1 + n2 + n3;
    for (int s = threadIdx.x; s < total; s += 256) {
        const int e = s < n0 ? fix[0][s] : s < n0 + n1 ? fix[1][s - n0] : s < n0 + n1 + n2 ? fix[2][s - n0 - n1] : fix[3][s - n0 - n1 - n2];
        const int ry = e >> 10;
        const int col = (e >> 4 & 63) * 4;  // byte column of the dword inside the LDS row
#pragma nounroll
        for (int m = e & 15; m != 0; m &= m - 1) {
            const int c = col + __builtin_ctz(m);
            o8[ry * (LDS_ROW_DW * 4) + c] = static_cast<uint8_t>(exact_pixel(t8 + (ry + HALO) * (LDS_ROW_DW * 4) + c, weights));
        }
    }
    __syncthreads();
    if (writer) {
        const __amdgpu_buffer_rsrc_t blurred = __builtin_amdgcn_make_buffer_rsrc(out, 0, v.rows * v.pitch, 0x00020000);
#pragma unroll
        for (int k = 0; k < ROWS_PER_WAVE; ++k) {
            const int ry = wave * ROWS_PER_WAVE + k;
            const int y = ty * TILE_H + ry;
            if (y >= v.rows) {
                break;
            }
            const uint32_t packed = outt[ry * LDS_ROW_DW + lane];
            uint8_t* dst = out + static_cast<size_t>(y) * v.pitch + x;
            if (x + 4 <= v.cols) {
                __builtin_amdgcn_raw_buffer_store_b32(packed, blurred, x, y * v.pitch, 0);  // scalar row offset
            } else {
                for (int p = 0; p < v.cols - x; ++p) {
                    dst[p] = static_cast<uint8_t>(packed >> (8 * p));
                }
            }
        }
    }
}

}  // namespace

void launch_blur(const BatchView& v, const double* weights_dev, int nframes, hipStream_t stream) {
    const int tiles_x = (v.cols + TILE_W - 1) / TILE_W;
    const int tiles_y = (v.rows + TILE_H - 1) / TILE_H;
    // tile_id / tiles_x as a multiply-high (exact while tile_id * tiles_x < 2^32; 0 stands for tiles_x == 1)
    const uint32_t tiles_x_recip = tiles_x > 1 ? 0xffffffffu / static_cast<uint32_t>(tiles_x) + 1u : 0u;
    hipLaunchKernelGGL(blur_kernel, frame_grid(nframes, tiles_x * tiles_y), dim3(256), 0, stream, v, weights_dev, nframes, tiles_x,
                       tiles_y, tiles_x_recip);
}

}  // namespace slamx
