// fast.hip -- FAST-N segment test, raster-order compaction and corner score.
//
// Reference behaviour (paths relative to the reference root):
//   detectFASTKeypoints   src/frontend/feature_detector.cpp:56-68   raster scan, border 3
//   isFASTCorner          src/frontend/feature_detector.cpp:70-145  two pretests + 32-step walk
//   computeFASTScore      src/frontend/feature_detector.cpp:190-203 sum of 16 |I_k - I_c|
//   M_PIXEL_OFFSETS       include/slam/frontend/feature_detector.hpp:138-153
//
// MI355X design: integer-only, one image read.
//   K1  fast_detect : one workgroup = 256 x 16 pixel tile, staged through LDS with a 3-px halo as
//                     aligned dword loads; pretest A on 4 pixels per lane with packed 16-bit math,
//                     pretest B + segment test only on the compacted survivors; the corner
//                     predicate of a 64-pixel row segment is one 64-bit word of a corner bitmap
//                     (W*H/8 bytes), the only thing written.
//   K1b scan        : per frame, exclusive prefix sum of the bitmap popcounts in raster order
//                     (row-major, word-major) -> the raster index of every corner is
//                     wordoff[word] + popcount(bits below it).  This bitmap+prefix pair is also
//                     the spatial index the NMS kernel uses, so no cell grid is ever built.
//   K1c expand      : one lane per bitmap word walks its set bits; writes (x,y) and the
//                     sort key score<<20|rasterIndex in raster order.
#include "slamx_internal.h"

namespace slamx {

namespace {

// circle offsets {dx,dy}, clockwise from 12 o'clock (feature_detector.hpp:138-153)
__device__ __constant__ int8_t CIRC_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__device__ __constant__ int8_t CIRC_DY[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};

constexpr int TILE_W = 256;
constexpr int TILE_H = 16;
constexpr int HALO = 3;
constexpr int LDS_ROW_DW = TILE_W / 4 + 2;   // 264 bytes: columns [x0-4, x0+260)
constexpr int LDS_ROWS = TILE_H + 2 * HALO;  // 22

// The segment test of isFASTCorner on brighter/darker bit masks of the 16 circle pixels.
__device__ __forceinline__ bool segment_test(uint32_t B, uint32_t D, int arc) {
    // pretest A (feature_detector.cpp:81-96): pixels 0 and 8
    if (((B | D) & 0x0101u) == 0) {
        return false;
    }
    // pretest B (:98-113): pixels 0,4,8,12 -- applied for every arc length, as the reference does
    if (__popc(B & 0x1111u) < 3 && __popc(D & 0x1111u) < 3) {
        return false;
    }
    // 32-step circular walk (:118-142): true iff the doubled sequence has a run >= arc.
    if (arc <= 0) {
        return true;  // "0 >= 0" fires on the first step
    }
    uint32_t mb = B | (B << 16);
    uint32_t md = D | (D << 16);
    uint32_t rb = mb;
    uint32_t rd = md;
    for (int s = 1; s < arc; ++s) {
        rb &= mb >> s;
        rd &= md >> s;
    }
    return (rb | rd) != 0;
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// bytes (0,1) / (2,3) of w zero-extended to two u16 lanes (v_perm_b32)
__device__ __forceinline__ us2 unpack_lo(uint32_t w) {
    return __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, w, 0x0c010c00u));
}
__device__ __forceinline__ us2 unpack_hi(uint32_t w) {
    return __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, w, 0x0c030c02u));
}
// per u16 lane: non-zero iff |n - c| > t, i.e. n is "brighter" or "darker" than c
// (feature_detector.cpp:87-91: n > c + t  or  n < c - t, in int arithmetic)
__device__ __forceinline__ us2 differs(us2 n, us2 c, us2 t) {
    const us2 up = __builtin_elementwise_sub_sat(__builtin_elementwise_sub_sat(n, c), t);
    const us2 dn = __builtin_elementwise_sub_sat(__builtin_elementwise_sub_sat(c, n), t);
    return up | dn;
}

// Two phases per 256 x 16 tile (a lane owns 4 horizontally adjacent pixels = one dword).
//  phase 1: pretest A of the reference (feature_detector.cpp:81-96: circle pixels 0 and 8, the
//           pixels 3 rows above / below, i.e. the same dword of two other tile rows) on 4 pixels at
//           once with packed 16-bit saturating arithmetic; survivors go to an LDS list.
//  phase 2: pretest B on the survivors, compacted again, then the 16-pixel segment test on what
//           is left (about 1-5 % of the pixels), so every stage runs on dense waves.
// Pixels that fail A can never be corners, so the bitmap is the reference's.
__global__ __launch_bounds__(256) void fast_detect_kernel(BatchView v, int threshold, int arc) {
    __shared__ uint32_t tile[LDS_ROWS * LDS_ROW_DW];
    __shared__ uint16_t survivors[TILE_W * TILE_H];
    __shared__ uint32_t row_mask[TILE_H][TILE_W / 32];
    __shared__ uint16_t survivors2[TILE_W * TILE_H];
    __shared__ int n_survivors;
    __shared__ int n_survivors2;
    const int tx = blockIdx.x;
    const int ty = blockIdx.y;
    const int f = blockIdx.z;
    const uint8_t* img = v.frames + static_cast<size_t>(f) * v.frame_stride;
    const int x0 = tx * TILE_W - 4;
    const int y0 = ty * TILE_H - HALO;
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
    if (threadIdx.x < TILE_H * (TILE_W / 32)) {
        row_mask[threadIdx.x / (TILE_W / 32)][threadIdx.x % (TILE_W / 32)] = 0;
    }
    if (threadIdx.x == 0) {
        n_survivors = 0;
        n_survivors2 = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int x = tx * TILE_W + lane * 4;
    const us2 t2 = {static_cast<unsigned short>(threshold), static_cast<unsigned short>(threshold)};
    // ---- phase 1
#pragma unroll
    for (int k = 0; k < TILE_H / 4; ++k) {
        const int ry = wave * (TILE_H / 4) + k;
        const int y = ty * TILE_H + ry;
        uint32_t flags = 0;  // bit p: pixel x+p survives pretest A
        if (y >= HALO && y < v.rows - HALO) {
            const uint32_t* rp = tile + (ry + HALO) * LDS_ROW_DW + lane + 1;
            const uint32_t C = rp[0];
            const uint32_t U = rp[-HALO * LDS_ROW_DW];
            const uint32_t D = rp[HALO * LDS_ROW_DW];
            const us2 c01 = unpack_lo(C), c23 = unpack_hi(C);
            const us2 f01 = differs(unpack_lo(U), c01, t2) | differs(unpack_lo(D), c01, t2);
            const us2 f23 = differs(unpack_hi(U), c23, t2) | differs(unpack_hi(D), c23, t2);
            flags = (f01.x != 0 ? 1u : 0u) | (f01.y != 0 ? 2u : 0u) | (f23.x != 0 ? 4u : 0u) | (f23.y != 0 ? 8u : 0u);
            // image border: columns < 3 or >= cols - 3 are never tested
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (x + p < HALO || x + p >= v.cols - HALO) {
                    flags &= ~(1u << p);
                }
            }
        }
        const unsigned long long any = __ballot(flags != 0);
        if (any != 0) {
            // wave-wide exclusive prefix of popcount(flags) through the four bit planes
            uint32_t before = 0;
            uint32_t total = 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const unsigned long long mp = __ballot((flags >> p) & 1u);
                before += static_cast<uint32_t>(__popcll(mp & ((1ull << lane) - 1ull)));
                total += static_cast<uint32_t>(__popcll(mp));
            }
            int base = 0;
            if (lane == 0) {
                base = atomicAdd(&n_survivors, static_cast<int>(total));
            }
            base = __shfl(base, 0);
            uint32_t pos = static_cast<uint32_t>(base) + before;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if ((flags >> p) & 1u) {
                    survivors[pos++] = static_cast<uint16_t>((ry << 8) | (lane * 4 + p));
                }
            }
        }
    }
    __syncthreads();
    // ---- phase 2a: pretest B (needs the two horizontal cardinal pixels too) on the A-survivors,
    // compacted again, so that the segment test below runs on dense waves
    const uint8_t* t8 = reinterpret_cast<const uint8_t*>(tile);
    constexpr int RS = LDS_ROW_DW * 4;  // LDS row stride in bytes
    const int total = n_survivors;
    for (int s0 = 0; s0 < total; s0 += 256) {  // block-uniform trip count
        const int s = s0 + static_cast<int>(threadIdx.x);
        bool pass = false;
        int e = 0;
        if (s < total) {
            e = survivors[s];
            const uint8_t* p = t8 + ((e >> 8) + HALO) * RS + ((e & 255) + 4);
            const int c = p[0];
            const int hi = c + threshold;
            const int lo = c - threshold;
            const int n0 = p[-3 * RS], n8 = p[3 * RS], n4 = p[3], n12 = p[-3];
            const int nb = (n0 > hi) + (n8 > hi) + (n4 > hi) + (n12 > hi);
            const int nd = (n0 < lo) + (n8 < lo) + (n4 < lo) + (n12 < lo);
            pass = nb >= 3 || nd >= 3;
        }
        const unsigned long long m = __ballot(pass);
        if (m != 0) {
            int base = 0;
            if (lane == 0) {
                base = atomicAdd(&n_survivors2, __popcll(m));
            }
            base = __shfl(base, 0);
            if (pass) {
                survivors2[base + __popcll(m & ((1ull << lane) - 1ull))] = static_cast<uint16_t>(e);
            }
        }
    }
    __syncthreads();
    // ---- phase 2b: the 16-pixel segment test
    const int total2 = n_survivors2;
    for (int s = threadIdx.x; s < total2; s += 256) {
        const int e = survivors2[s];
        const int ry = e >> 8;
        const int col = e & 255;
        const uint8_t* p = t8 + (ry + HALO) * RS + (col + 4);
        const int c = p[0];
        const int hi = c + threshold;
        const int lo = c - threshold;
        uint32_t B = 0;
        uint32_t D = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int n = p[CIRC_DY[q] * RS + CIRC_DX[q]];
            B |= static_cast<uint32_t>(n > hi) << q;
            D |= static_cast<uint32_t>(n < lo) << q;
        }
        if (segment_test(B, D, arc)) {
            atomicOr(&row_mask[ry][col >> 5], 1u << (col & 31));
        }
    }
    __syncthreads();
    // one 64-bit bitmap word per (row, 64-column group): 16 rows x 4 words per tile
    if (threadIdx.x < TILE_H * (TILE_W / 64)) {
        const int ry = threadIdx.x / (TILE_W / 64);
        const int wq = threadIdx.x % (TILE_W / 64);
        const int y = ty * TILE_H + ry;
        const int wx = tx * (TILE_W / 64) + wq;
        if (y < v.rows && wx < v.nw) {
            unsigned long long* masks = v.masks + static_cast<size_t>(f) * v.rows * v.nw;
            masks[static_cast<size_t>(y) * v.nw + wx] = static_cast<unsigned long long>(row_mask[ry][2 * wq]) |
                                                        (static_cast<unsigned long long>(row_mask[ry][2 * wq + 1]) << 32);
        }
    }
}

// One workgroup per frame: exclusive scan of popcount(masks) over rows*nw words.
__global__ __launch_bounds__(1024) void scan_kernel(BatchView v) {
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t carry;
    const int f = blockIdx.x;
    const int n = v.rows * v.nw;
    const unsigned long long* masks = v.masks + static_cast<size_t>(f) * n;
    uint32_t* off = v.wordoff + static_cast<size_t>(f) * n;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        carry = 0;
    }
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t c = i < n ? static_cast<uint32_t>(__popcll(masks[i])) : 0u;
        // inclusive wave scan
        uint32_t s = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(s, d);
            if (lane >= d) {
                s += t;
            }
        }
        if (lane == 63) {
            wave_sums[wave] = s;
        }
        __syncthreads();
        uint32_t wbase = carry;
        for (int w = 0; w < wave; ++w) {
            wbase += wave_sums[w];
        }
        if (i < n) {
            off[i] = wbase + s - c;
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
            carry = wbase + s;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int m = static_cast<int>(carry);
        int st = 0;
        if (m > v.maxM) {
            st = STATUS_CAND_OVERFLOW;  // fail loudly on the host; downstream kernels see M = 0
            m = 0;
        }
        v.n_cand[f] = m;
        v.status[f] = st;
        v.n_kp[f] = 0;
    }
}

// One lane per bitmap word (coalesced 8-byte loads; ~half of the words hold a corner, usually
// one): the lane walks its set bits and writes (x,y) and the sort key of each corner.
__global__ __launch_bounds__(256) void expand_kernel(BatchView v) {
    const int f = blockIdx.y;
    const int m_total = v.n_cand[f];
    if (m_total == 0) {
        return;
    }
    const int n = v.rows * v.nw;
    const unsigned long long* __restrict__ masks = v.masks + static_cast<size_t>(f) * n;
    const uint32_t* __restrict__ off = v.wordoff + static_cast<size_t>(f) * n;
    const uint8_t* __restrict__ img = v.frames + static_cast<size_t>(f) * v.frame_stride;
    uint32_t* xy = v.cand_xy + static_cast<size_t>(f) * v.maxM;
    uint32_t* keys = v.keys + static_cast<size_t>(f) * v.maxM;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < n; w += gridDim.x * blockDim.x) {
        unsigned long long m = masks[w];
        if (m == 0) {
            continue;
        }
        const int row = w / v.nw;
        const int x0 = (w - row * v.nw) * 64;
        uint32_t idx = off[w];
        const uint8_t* prow = img + static_cast<size_t>(row) * v.pitch + x0;
        while (m != 0) {
            const int b = __ffsll(static_cast<long long>(m)) - 1;
            m &= m - 1;
            const uint8_t* p = prow + b;
            const int c = p[0];
            int score = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int nb = p[CIRC_DY[q] * v.pitch + CIRC_DX[q]];
                score += abs(nb - c);
            }
            xy[idx] = (static_cast<uint32_t>(row) << 16) | static_cast<uint32_t>(x0 + b);
            keys[idx] = (static_cast<uint32_t>(score) << SCORE_SHIFT) | idx;
            ++idx;
        }
    }
}

}  // namespace

void launch_fast_detect(const BatchView& v, const DetectorParams& p, int nframes, hipStream_t stream) {
    dim3 grid((v.cols + TILE_W - 1) / TILE_W, (v.rows + TILE_H - 1) / TILE_H, nframes);
    hipLaunchKernelGGL(fast_detect_kernel, grid, dim3(256), 0, stream, v, p.threshold, p.arc);
}

void launch_scan(const BatchView& v, int nframes, hipStream_t stream) {
    hipLaunchKernelGGL(scan_kernel, dim3(nframes), dim3(1024), 0, stream, v);
}

void launch_expand(const BatchView& v, const DetectorParams& p, int nframes, hipStream_t stream) {
    (void)p;
    const int n = v.rows * v.nw;
    hipLaunchKernelGGL(expand_kernel, dim3(grid_x(n, 256, nframes), nframes), dim3(256), 0, stream, v);
}

}  // namespace slamx
