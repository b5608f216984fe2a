// pyramid.hip -- image pyramid for multi-scale detection: integer bilinear downscale of a whole
// batch of frames, and the merge of the per-level keypoints / descriptors into level 0's result
// arrays.
//
// NO REFERENCE COUNTERPART: the reference detects on the full-resolution image only
// (src/frontend/feature_detector.cpp:8-18, :56-68; Keypoint::size is a constant 6.0,
// include/slam/frontend/feature_detector.hpp:18,35).  BASELINE.json's
// north_star ("image pyramids ...") and configs[2] ("8-level pyramid") ask for one, so it is a
// build-side extension: level 0 is the reference's detector unchanged (parity-pinned as before),
// levels >= 1 run the SAME detector on downscaled frames.  PARITY UNPINNED for levels >= 1; the
// arithmetic is integer-only and defined here, so the CPU restatement in
// oracle/pyramid_oracle.cpp is compared bit for bit.
//
// Downscale (level l from level l-1, as ORB-style pyramids do): bilinear with 11-bit weights,
//   centre-aligned source position  s = ((2 d + 1) Ws - Wd) / (2 Wd)   (exact rational, per axis)
//   i0 = floor(s) clamped to [0, Ws - 1], f = floor(frac(s) * 2048) (0 at the clamps), i1 = min(i0 + 1, Ws - 1)
//   v = ((a (2048 - fx) + b fx) (2048 - fy) + (c (2048 - fx) + d fx) fy + 2^21) >> 22
// The per-axis tables (i0 | f << 16) are built once on the host with 64-bit integers.
// HBM-bound: every source byte read once (neighbouring taps hit L2 / TA), every destination byte written once.
#include "slamx_internal.h"

namespace slamx {

namespace {

// one lane = 4 horizontally adjacent destination pixels (one dword store)
__global__ __launch_bounds__(256) void downscale_kernel(const uint8_t* __restrict__ src, size_t src_frame_stride, int src_pitch,
                                                        int src_cols, uint8_t* __restrict__ dst, size_t dst_frame_stride,
                                                        int dst_pitch, int dst_rows, int dst_cols,
                                                        const uint32_t* __restrict__ xtab, const uint32_t* __restrict__ ytab,
                                                        int nframes, int blocks_per_frame) {
    int f, bx;
    if (!frame_block(nframes, blocks_per_frame, &f, &bx)) {
        return;
    }
    const int dwords_per_row = dst_pitch / 4;
    const int total = dst_rows * dwords_per_row;
    const uint8_t* s = src + static_cast<size_t>(f) * src_frame_stride;
    uint8_t* d = dst + static_cast<size_t>(f) * dst_frame_stride;
    for (int e = bx * 256 + static_cast<int>(threadIdx.x); e < total; e += blocks_per_frame * 256) {
        const int y = e / dwords_per_row;
        const int x4 = (e - y * dwords_per_row) * 4;
        const uint32_t ye = ytab[y];
        const int y0 = static_cast<int>(ye & 0xffffu);
        const uint32_t fy = ye >> 16;
        const uint8_t* r0 = s + static_cast<size_t>(y0) * src_pitch;
        const uint8_t* r1 = s + static_cast<size_t>(fy != 0 ? y0 + 1 : y0) * src_pitch;  // f != 0 => y0 + 1 exists
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x4 + k;
            uint32_t v = 0;
            if (x < dst_cols) {
                const uint32_t xe = xtab[x];
                const int x0 = static_cast<int>(xe & 0xffffu);
                const uint32_t fx = xe >> 16;
                const int x1 = fx != 0 ? x0 + 1 : x0;  // f != 0 => x0 + 1 < src_cols
                const uint32_t a = r0[x0], b = r0[x1], c = r1[x0], dd = r1[x1];
                const uint32_t top = a * (2048u - fx) + b * fx;
                const uint32_t bot = c * (2048u - fx) + dd * fx;
                v = (top * (2048u - fy) + bot * fy + (1u << 21)) >> 22;
            }
            out |= v << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(d + static_cast<size_t>(y) * dst_pitch + x4) = out;  // pitch padding gets zeros
    }
    (void)src_cols;
}

// Append level l's keypoints (coordinates and size scaled to level 0) and descriptors to level 0's arrays.
// One workgroup per frame.  dst_count is level 0's n_kp: read as the offset, then increased.
__global__ __launch_bounds__(256) void merge_level_kernel(const slamx_keypoint* __restrict__ src_kps,
                                                          const uint8_t* __restrict__ src_desc, const int* __restrict__ src_count,
                                                          const int* __restrict__ src_status, int src_maxK,
                                                          slamx_keypoint* __restrict__ dst_kps, uint8_t* __restrict__ dst_desc,
                                                          int* __restrict__ dst_count, int* __restrict__ dst_status, int dst_maxK,
                                                          int width, float scale) {
    const int f = blockIdx.x;
    const int n = src_count[f];
    const int off = dst_count[f];
    __syncthreads();  // everyone has read the offset before thread 0 updates it
    const int st = src_status[f];
    if (off + n > dst_maxK || st != 0) {
        if (threadIdx.x == 0) {
            atomicOr(&dst_status[f], st != 0 ? st : STATUS_KP_OVERFLOW);
        }
        return;
    }
    const slamx_keypoint* sk = src_kps + static_cast<size_t>(f) * src_maxK;
    slamx_keypoint* dk = dst_kps + static_cast<size_t>(f) * dst_maxK + off;
    for (int i = threadIdx.x; i < n; i += 256) {
        slamx_keypoint k = sk[i];
        k.x = __fmul_rn(k.x, scale);
        k.y = __fmul_rn(k.y, scale);
        k.size = __fmul_rn(k.size, scale);
        dk[i] = k;
    }
    if ((width & 3) == 0) {
        const uint32_t* sd = reinterpret_cast<const uint32_t*>(src_desc + static_cast<size_t>(f) * src_maxK * width);
        uint32_t* dd = reinterpret_cast<uint32_t*>(dst_desc + (static_cast<size_t>(f) * dst_maxK + off) * width);
        const int words = n * (width / 4);
        for (int i = threadIdx.x; i < words; i += 256) {
            dd[i] = sd[i];
        }
    } else {
        // descriptor widths that are not a multiple of 4 bytes: byte copy, in this same kernel, so that a level that
        // was NOT merged (the early return above) can never be copied over the descriptors of earlier levels
        const uint8_t* sd = src_desc + static_cast<size_t>(f) * src_maxK * width;
        uint8_t* dd = dst_desc + (static_cast<size_t>(f) * dst_maxK + off) * width;
        for (int i = threadIdx.x; i < n * width; i += 256) {
            dd[i] = sd[i];
        }
    }
    if (threadIdx.x == 0) {
        dst_count[f] = off + n;
    }
}

}  // namespace

void launch_downscale(const BatchView& src, const BatchView& dst, const uint32_t* xtab, const uint32_t* ytab, int nframes,
                      hipStream_t stream) {
    const int total = dst.rows * (dst.pitch / 4);
    const int bpf = grid_x(total, 256 * 4, nframes);
    hipLaunchKernelGGL(downscale_kernel, dim3(frame_grid(nframes, bpf)), dim3(256), 0, stream, src.frames, src.frame_stride,
                       src.pitch, src.cols, const_cast<uint8_t*>(dst.frames), dst.frame_stride, dst.pitch, dst.rows, dst.cols,
                       xtab, ytab, nframes, bpf);
}

void launch_merge_level(const BatchView& src, const BatchView& dst, float scale, int nframes, hipStream_t stream) {
    hipLaunchKernelGGL(merge_level_kernel, dim3(nframes), dim3(256), 0, stream, src.kps, src.desc, src.n_kp, src.status,
                       src.maxK, dst.kps, dst.desc, dst.n_kp, dst.status, dst.maxK, src.width, scale);
}

}  // namespace slamx
