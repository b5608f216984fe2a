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

// One lane = 4 horizontally adjacent destination pixels (one dword store); a workgroup = 1024 pixels of ONE destination
// row, so the source rows and the vertical weight are scalars.  The lane's source pixels lie within 12 bytes of the
// dword that holds its first one (scale factors up to 2.5: 3 + 3 * 2.5 + 1 < 12), so the two source rows are read as
// ONE 12-byte load each and the taps are cut out of registers (v_alignbyte); the first version issued 16 byte loads
// and 4 table loads per lane and ran at a quarter of the HBM rate (3.2 of the pyramid workload's 15.6 ms).
constexpr int DS_ROWS = 4;  // destination rows per thread: the column table is decoded once for all of them

__global__ __launch_bounds__(256) void downscale_kernel(const uint8_t* __restrict__ src, size_t src_frame_stride, int src_pitch,
                                                        int src_cols, uint8_t* __restrict__ dst, size_t dst_frame_stride,
                                                        int dst_pitch, int dst_rows, int dst_cols,
                                                        const uint32_t* __restrict__ xtab, const uint32_t* __restrict__ ytab,
                                                        int nframes, int blocks_per_frame, int segs) {
    int f, bx;
    if (!frame_block(nframes, blocks_per_frame, &f, &bx)) {
        return;
    }
    const int yb = (bx / segs) * DS_ROWS;  // (scalar) first of this block's destination rows
    const int xd = (bx - (bx / segs) * segs) * 256 + static_cast<int>(threadIdx.x);
    if (yb >= dst_rows || xd * 4 >= dst_pitch) {
        return;
    }
    const uint8_t* s = src + static_cast<size_t>(f) * src_frame_stride;
    uint32_t* out = reinterpret_cast<uint32_t*>(dst + static_cast<size_t>(f) * dst_frame_stride + static_cast<size_t>(yb) * dst_pitch) + xd;
    const int out_step = dst_pitch / 4;  // (the pitch is a multiple of 64)
    const int nrows = min(DS_ROWS, dst_rows - yb);
    const int x4 = xd * 4;
    if (x4 >= dst_cols) {
        for (int r = 0; r < nrows; ++r) {
            out[r * out_step] = 0u;  // pitch padding gets zeros
        }
        return;
    }
    const uint4 xe4 = *reinterpret_cast<const uint4*>(xtab + x4);  // (the table is padded to a multiple of 4 entries)
    const uint32_t xe[4] = {xe4.x, xe4.y, xe4.z, xe4.w};
    const int base = static_cast<int>(xe[0] & 0xfffcu);
    uint32_t ye[DS_ROWS];
#pragma unroll
    for (int r = 0; r < DS_ROWS; ++r) {
        ye[r] = ytab[min(yb + r, dst_rows - 1)];
    }
    if (static_cast<int>(xe[3] & 0xffffu) - base <= 10) {
        // all row pairs are requested before the first is used
        // (frames end with 64 bytes of slack: the 12 bytes may reach past the last row's pitch)
        uint3 a3[DS_ROWS], b3[DS_ROWS];
#pragma unroll
        for (int r = 0; r < DS_ROWS; ++r) {
            const int y0 = static_cast<int>(ye[r] & 0xffffu);
            const int fy = static_cast<int>(ye[r] >> 16);
            a3[r] = *reinterpret_cast<const uint3*>(s + static_cast<size_t>(y0) * src_pitch + base);
            b3[r] = *reinterpret_cast<const uint3*>(s + static_cast<size_t>(fy != 0 ? y0 + 1 : y0) * src_pitch + base);  // f != 0 => y0 + 1 exists
        }
        uint32_t result[DS_ROWS] = {};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = static_cast<int>(xe[k] & 0xffffu) - base;  // 0 .. 10: this tap and its right neighbour are bytes o, o + 1
            const int fx = static_cast<int>(xe[k] >> 16);
            const bool hi = o >= 4;
            const bool hi2 = o >= 8;
            const bool inside = x4 + k < dst_cols;
#pragma unroll
            for (int r = 0; r < DS_ROWS; ++r) {
                const int fy = static_cast<int>(ye[r] >> 16);
                // the 8 bytes that hold byte o and o + 1, then the 4 bytes from byte o on
                const uint32_t ta = __builtin_amdgcn_alignbyte(hi2 ? 0u : (hi ? a3[r].z : a3[r].y), hi2 ? a3[r].z : (hi ? a3[r].y : a3[r].x),
                                                               static_cast<uint32_t>(o));
                const uint32_t tb = __builtin_amdgcn_alignbyte(hi2 ? 0u : (hi ? b3[r].z : b3[r].y), hi2 ? b3[r].z : (hi ? b3[r].y : b3[r].x),
                                                               static_cast<uint32_t>(o));
                const int a = static_cast<int>(ta & 0xffu), b = static_cast<int>(ta >> 8 & 0xffu);
                const int c = static_cast<int>(tb & 0xffu), d = static_cast<int>(tb >> 8 & 0xffu);
                // a (2048 - fx) + b fx == 2048 a + (b - a) fx: the same integers as the defining formula (fx == 0: b is
                // not used, whatever byte it is)
                const int top = (a << 11) + (b - a) * fx;
                const int bot = (c << 11) + (d - c) * fx;
                const int v = ((top << 11) + (bot - top) * fy + (1 << 21)) >> 22;
                result[r] |= (inside ? static_cast<uint32_t>(v) : 0u) << (8 * k);
            }
        }
#pragma unroll
        for (int r = 0; r < DS_ROWS; ++r) {
            if (r < nrows) {
                out[r * out_step] = result[r];
            }
        }
    } else {  // very large scale factors: the taps are read one by one
        for (int r = 0; r < nrows; ++r) {
            const int y0 = static_cast<int>(ye[r] & 0xffffu);
            const int fy = static_cast<int>(ye[r] >> 16);
            const uint8_t* r0 = s + static_cast<size_t>(y0) * src_pitch;
            const uint8_t* r1 = s + static_cast<size_t>(fy != 0 ? y0 + 1 : y0) * src_pitch;
            uint32_t result = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x0 = static_cast<int>(xe[k] & 0xffffu);
                const int fx = static_cast<int>(xe[k] >> 16);
                const int x1 = fx != 0 ? x0 + 1 : x0;  // f != 0 => x0 + 1 < src_cols
                const int a = r0[x0], b = r0[x1], c = r1[x0], d = r1[x1];
                const int top = (a << 11) + (b - a) * fx;
                const int bot = (c << 11) + (d - c) * fx;
                const int v = ((top << 11) + (bot - top) * fy + (1 << 21)) >> 22;
                result |= (x4 + k < dst_cols ? static_cast<uint32_t>(v) : 0u) << (8 * k);
            }
            out[r * out_step] = result;
        }
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
    const int segs = (dst.pitch / 4 + 255) / 256;  // workgroups per group of DS_ROWS destination rows
    const int bpf = (dst.rows + DS_ROWS - 1) / DS_ROWS * segs;
    hipLaunchKernelGGL(downscale_kernel, dim3(frame_grid(nframes, bpf)), dim3(256), 0, stream, src.frames, src.frame_stride,
                       src.pitch, src.cols, const_cast<uint8_t*>(dst.frames), dst.frame_stride, dst.pitch, dst.rows, dst.cols,
                       xtab, ytab, nframes, bpf, segs);
}

void launch_merge_level(const BatchView& src, const BatchView& dst, float scale, int nframes, hipStream_t stream) {
    hipLaunchKernelGGL(merge_level_kernel, dim3(nframes), dim3(256), 0, stream, src.kps, src.desc, src.n_kp, src.status,
                       src.maxK, dst.kps, dst.desc, dst.n_kp, dst.status, dst.maxK, src.width, scale);
}

}  // namespace slamx
