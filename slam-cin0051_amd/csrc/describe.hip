// describe.hip -- intensity-centroid orientation + steered BRIEF, one wave per keypoint.
//
// Reference (paths relative to the reference root):
//   computeOrientation      src/frontend/feature_detector.cpp:205-231
//   computeBRIEFDescriptor  src/frontend/feature_detector.cpp:233-284
//   compute (driver)        src/frontend/feature_detector.cpp:20-47
//
// Both run on the BLURRED image (feature_detector.cpp:33-41).
//  * Moments: the reference accumulates v*I and u*I in float over the disc u^2+v^2 <= r^2.  Every
//    partial sum is an integer below 2^24 for the shipped patch size (255 * sum|v| = 1.15e6), so
//    float accumulation is exact and order-free: lanes accumulate int32 partials and the wave
//    reduces them.  For patch sizes where that bound fails (host decides) one lane replays the
//    reference's float loop in its (v outer, u inner) order.
//  * atan2f / cosf / sinf: glibc's algorithms restated in glibc_trig.h (third-party arithmetic).
//  * Rotation: the same four float multiplies and two adds/subs per sample, truncation toward zero.
//  * Bit packing is data dependent (a pair whose samples leave the image is skipped WITHOUT
//    advancing bitIndex, :268-280): per 64-pair chunk the wave ballots `valid` and `bit`, lane k
//    picks the k-th valid pair's bit (select on the ballot) and a second ballot is the packed
//    run, appended at the running bit offset.
#include <cstdlib>

#include "glibc_trig.h"
#include "slamx_internal.h"

#pragma clang fp contract(off)

namespace slamx {

namespace {

__device__ __forceinline__ int select64d(unsigned long long m, int k) {
    int pos = 0;
    uint32_t v = static_cast<uint32_t>(m);
    int c = __popc(v);
    if (k >= c) {
        k -= c;
        pos = 32;
        v = static_cast<uint32_t>(m >> 32);
    }
    c = __popc(v & 0xffffu);
    if (k >= c) {
        k -= c;
        pos += 16;
        v >>= 16;
    }
    c = __popc(v & 0xffu);
    if (k >= c) {
        k -= c;
        pos += 8;
        v >>= 8;
    }
    c = __popc(v & 0xfu);
    if (k >= c) {
        k -= c;
        pos += 4;
        v >>= 4;
    }
    c = __popc(v & 0x3u);
    if (k >= c) {
        k -= c;
        pos += 2;
        v >>= 2;
    }
    if (k >= static_cast<int>(v & 1u)) {
        pos += 1;
    }
    return pos;
}

// Keypoint coordinates are caller data (slamx_compute).  The reference truncates them with
// static_cast<int>(kp.x) (feature_detector.cpp:207-208, :235-236), undefined for values an int
// cannot hold; here they are clamped to +-2^20 first (NaN -> -2^20), so that x +- radius cannot wrap
// and such a keypoint is simply "outside the image": angle 0, zero descriptor.  Identical to the
// reference for every coordinate an image (<= 65535 per side) can contain.
__device__ __forceinline__ int kp_coord(float c) {
    return static_cast<int>(__builtin_amdgcn_fmed3f(c, -1048576.0f, 1048576.0f));
}

__device__ __forceinline__ unsigned long long low_mask_d(int nbits) {  // bits [0, nbits)
    return nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        v += __shfl_xor(v, d);
    }
    return v;
}

constexpr int DESC_WORDS = MAX_PATTERN / 64;  // 16 x u64 = 128 bytes max descriptor

__global__ __launch_bounds__(256) void describe_kernel(BatchView v, DetectorParams P, const int8_t* __restrict__ pattern,
                                                       int exact_int_moments) {
    __shared__ unsigned long long dbuf[4][DESC_WORDS];
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int n = v.n_kp[f];
    const uint8_t* img = v.blurred + static_cast<size_t>(f) * v.frame_stride;
    for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {  // wave-uniform trip count
    slamx_keypoint* kp = v.kps + static_cast<size_t>(f) * v.maxK + i;
    uint8_t* desc = v.desc + (static_cast<size_t>(f) * v.maxK + i) * v.width;
    const int x = kp_coord(kp->x);
    const int y = kp_coord(kp->y);
    const int radius = P.patch / 2;
    const bool inside = !(x - radius < 0 || x + radius >= v.cols || y - radius < 0 || y + radius >= v.rows);

    // ---- computeOrientation ----
    float angle = 0.0f;
    if (inside) {
        float m01f, m10f;
        if (exact_int_moments) {
            const int side = 2 * radius + 1;
            const int total = side * side;
            int m01 = 0;
            int m10 = 0;
            for (int t = lane; t < total; t += 64) {
                const int vv = t / side - radius;
                const int uu = t - (vv + radius) * side - radius;
                if (uu * uu + vv * vv <= radius * radius) {
                    const int I = img[static_cast<size_t>(y + vv) * v.pitch + (x + uu)];
                    m01 += vv * I;
                    m10 += uu * I;
                }
            }
            m01f = static_cast<float>(wave_sum(m01));
            m10f = static_cast<float>(wave_sum(m10));
        } else {
            // partial sums may exceed 2^24: replay the reference's float loop order
            float a01 = 0.0f;
            float a10 = 0.0f;
            if (lane == 0) {
                for (int vv = -radius; vv <= radius; ++vv) {
                    for (int uu = -radius; uu <= radius; ++uu) {
                        if (uu * uu + vv * vv <= radius * radius) {
                            const float I = static_cast<float>(img[static_cast<size_t>(y + vv) * v.pitch + (x + uu)]);
                            a01 = __fadd_rn(a01, __fmul_rn(static_cast<float>(vv), I));
                            a10 = __fadd_rn(a10, __fmul_rn(static_cast<float>(uu), I));
                        }
                    }
                }
            }
            m01f = __shfl(a01, 0);
            m10f = __shfl(a10, 0);
        }
        angle = __fmul_rn(glibc_atan2f(m01f, m10f), rad_to_deg());
    }
    if (lane == 0) {
        kp->angle = angle;  // feature_detector.cpp:37-38
    }

    // ---- computeBRIEFDescriptor ----
    const int words = (v.width + 7) / 8;
    if (lane < words) {
        dbuf[wave][lane] = 0ull;
    }
    if (inside) {
        const float a = __fmul_rn(angle, deg_to_rad());
        const float ca = glibc_cosf(a);
        const float sa = glibc_sinf(a);
        int bitIndex = 0;
        for (int base = 0; base < P.pattern_len; base += 64) {
            const int pi = base + lane;
            bool valid = false;
            bool bit = false;
            if (pi < P.pattern_len) {
                const float p1x = static_cast<float>(pattern[4 * pi + 0]);
                const float p1y = static_cast<float>(pattern[4 * pi + 1]);
                const float p2x = static_cast<float>(pattern[4 * pi + 2]);
                const float p2y = static_cast<float>(pattern[4 * pi + 3]);
                const int x1 = static_cast<int>(__fsub_rn(__fmul_rn(p1x, ca), __fmul_rn(p1y, sa))) + x;
                const int y1 = static_cast<int>(__fadd_rn(__fmul_rn(p1x, sa), __fmul_rn(p1y, ca))) + y;
                const int x2 = static_cast<int>(__fsub_rn(__fmul_rn(p2x, ca), __fmul_rn(p2y, sa))) + x;
                const int y2 = static_cast<int>(__fadd_rn(__fmul_rn(p2x, sa), __fmul_rn(p2y, ca))) + y;
                valid = x1 >= 0 && x1 < v.cols && y1 >= 0 && y1 < v.rows && x2 >= 0 && x2 < v.cols && y2 >= 0 &&
                        y2 < v.rows;
                if (valid) {
                    const uint8_t I1 = img[static_cast<size_t>(y1) * v.pitch + x1];
                    const uint8_t I2 = img[static_cast<size_t>(y2) * v.pitch + x2];
                    bit = I1 < I2;
                }
            }
            const unsigned long long vm = __ballot(valid);
            const unsigned long long bm = __ballot(valid && bit);
            const int cnt = __popcll(vm);
            bool outbit = false;
            if (lane < cnt) {
                outbit = ((bm >> select64d(vm, lane)) & 1ull) != 0;
            }
            const unsigned long long packed = __ballot(outbit);
            if (lane == 0 && cnt > 0) {
                const int w = bitIndex >> 6;
                const int sh = bitIndex & 63;
                dbuf[wave][w] |= packed << sh;
                if (sh != 0 && sh + cnt > 64) {
                    dbuf[wave][w + 1] |= packed >> (64 - sh);
                }
            }
            bitIndex += cnt;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint8_t* d8 = reinterpret_cast<const uint8_t*>(dbuf[wave]);
    for (int b = lane; b < v.width; b += 64) {
        desc[b] = d8[b];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    }  // keypoint loop
}

// ---- fast path for PatchSize <= 31 (radius <= 15, rotated BRIEF offsets within +-21) ----
// Three kernels so that the serial glibc trig chain (~300 dependent instructions) runs one
// keypoint per LANE instead of once per 64-lane wave:
//   moments_kernel : wave per keypoint; the 31-row neighbourhood is staged into LDS with coalesced
//                    aligned dword loads, the 709 disc offsets come from a table built once per
//                    workgroup (no divide per pixel); exact integer m01, m10.
//   trig_kernel    : lane per keypoint; atan2f -> degrees (written to Keypoint::angle) -> cosf,
//                    sinf (glibc_trig.h).
//   brief_kernel   : wave per keypoint; the 43 x 48-byte neighbourhood staged into LDS, lane per
//                    pattern pair, ballot packing.
constexpr int TILE_HALF = 21;
constexpr int TILE_ROWS = 2 * TILE_HALF + 1;  // 43
constexpr int TILE_DW = 12;                   // 48 bytes: columns [x0a, x0a+48), x0a = (x-21) & ~3
constexpr int MOM_HALF = 15;
constexpr int MOM_ROWS = 2 * MOM_HALF + 1;    // 31
constexpr int MOM_DW = 9;                     // 36 bytes: columns [x0a, x0a+36), x0a = (x-15) & ~3

// Both wave-per-keypoint kernels below are latency chains (keypoint -> tile -> result), so each
// wave (1) fetches the coordinates of up to 64 of its keypoints with ONE coalesced load and hands
// them out with readlane, and (2) issues the global loads of the NEXT keypoint's tile into
// registers before it computes on the current one, committing them to LDS afterwards.
template <int ROWS, int DW, int HALF>
struct TilePrefetch {
    static constexpr int PER_LANE = (ROWS * DW + 63) / 64;
    uint32_t reg[PER_LANE];
    __device__ __forceinline__ void issue(const BatchView& v, const uint8_t* img, int x, int y, int lane) {
        const int x0a = (x - HALF) & ~3;
        const int y0 = y - HALF;
#pragma unroll
        for (int k = 0; k < PER_LANE; ++k) {
            const int e = lane + 64 * k;
            const int r = e / DW;
            const int c = e - r * DW;
            const int yy = y0 + r;
            const int xx = x0a + c * 4;
            uint32_t val = 0;
            if (e < ROWS * DW && yy >= 0 && yy < v.rows && xx >= 0 && xx + 4 <= v.pitch) {
                val = *reinterpret_cast<const uint32_t*>(img + static_cast<size_t>(yy) * v.pitch + xx);
            }
            reg[k] = val;
        }
    }
    // same for a tile that lies entirely inside the padded frame (the caller checks): off[k] is the
    // byte offset of this lane's k-th dword from the tile's first byte, computed once per wave
    __device__ __forceinline__ void init_offsets(int pitch, int lane, int* off) const {
#pragma unroll
        for (int k = 0; k < PER_LANE; ++k) {
            const int e = lane + 64 * k;
            const int ee = e < ROWS * DW ? e : 0;
            const int r = ee / DW;
            off[k] = r * pitch + (ee - r * DW) * 4;
        }
    }
    // (buffer loads: the tile's first byte is a scalar offset, off[k] the lane's -- no 64-bit vector address math)
    __device__ __forceinline__ void issue_inside(__amdgpu_buffer_rsrc_t frame, int pitch, int x, int y, const int* off) {
        const int base = (y - HALF) * pitch + ((x - HALF) & ~3);
#pragma unroll
        for (int k = 0; k < PER_LANE; ++k) {
            reg[k] = __builtin_amdgcn_raw_buffer_load_b32(frame, off[k], base, 0);
        }
    }
    __device__ __forceinline__ void commit(uint32_t* tile, int lane) const {
#pragma unroll
        for (int k = 0; k < PER_LANE; ++k) {
            const int e = lane + 64 * k;
            if (e < ROWS * DW) {
                tile[e] = reg[k];
            }
        }
    }
};

// The same tile moved in 12-byte pieces (a 48-byte tile row = 4 pieces): the texture-address unit spends
// ~16 cycles on a wave's load instruction whatever its width, and it is the busiest unit of the fused
// describe kernel -- 3 instructions per tile instead of 9.  Piece e = lane + 64 k covers tile dwords
// 3 e .. 3 e + 2 (row e / 4), so the LDS image is the same row-major tile.
template <int ROWS, int HALF>
struct TilePrefetch12 {
    static constexpr int DW = 12;
    static constexpr int PIECES = ROWS * 4;
    static constexpr int PER_LANE = (PIECES + 63) / 64;
    uint32_t reg[PER_LANE][3];
    __device__ __forceinline__ void init_offsets(int pitch, int lane, int* off) const {
#pragma unroll
        for (int k = 0; k < PER_LANE; ++k) {
            const int e = lane + 64 * k;
            const int ee = e < PIECES ? e : 0;
            off[k] = (ee >> 2) * pitch + (ee & 3) * 12;
        }
    }
    // tile entirely inside the padded frame (the caller checks)
    __device__ __forceinline__ void issue_inside(__amdgpu_buffer_rsrc_t frame, int pitch, int x, int y, const int* off) {
        const int base = (y - HALF) * pitch + ((x - HALF) & ~3);  // scalar: x, y come from readlane
#pragma unroll
        for (int k = 0; k < PER_LANE; ++k) {
            const auto p = __builtin_amdgcn_raw_buffer_load_b96(frame, off[k], base, 0);
            reg[k][0] = p[0];
            reg[k][1] = p[1];
            reg[k][2] = p[2];
        }
    }
    // any tile: dwords outside the padded frame read as 0
    __device__ __forceinline__ void issue(const BatchView& v, const uint8_t* img, int x, int y, int lane) {
        const int x0a = (x - HALF) & ~3;
        const int y0 = y - HALF;
#pragma unroll
        for (int k = 0; k < PER_LANE; ++k) {
            const int e = lane + 64 * k;
            const int yy = y0 + (e >> 2);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int xx = x0a + (e & 3) * 12 + j * 4;
                uint32_t val = 0;
                if (e < PIECES && yy >= 0 && yy < v.rows && xx >= 0 && xx + 4 <= v.pitch) {
                    val = *reinterpret_cast<const uint32_t*>(img + static_cast<size_t>(yy) * v.pitch + xx);
                }
                reg[k][j] = val;
            }
        }
    }
    __device__ __forceinline__ void commit(uint32_t* tile, int lane) const {
#pragma unroll
        for (int k = 0; k < PER_LANE; ++k) {
            const int e = lane + 64 * k;
            if (e < PIECES) {
                tile[3 * e + 0] = reg[k][0];
                tile[3 * e + 1] = reg[k][1];
                tile[3 * e + 2] = reg[k][2];
            }
        }
    }
};

// value of lane `src` (wave-uniform index) as a scalar: v_readlane_b32, so that everything derived
// from it (tile base address, alignment, bounds tests) stays on the SALU
__device__ __forceinline__ int lane_value(int v, int src) {
    return __builtin_amdgcn_readlane(v, src);
}
__device__ __forceinline__ float lane_value(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

// sum over the 64 lanes (DPP row reductions + row broadcasts; the total lands in lane 63)
__device__ __forceinline__ int wave_sum_dpp(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);  // row_half_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false);  // row_mirror: every lane holds its row's sum
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(x, 63);
}

// the same for N independent values at once, one reduction step of all of them after the other (the sums
// land in lane 63 of x[i])
template <int N>
__device__ __forceinline__ void wave_sums_dpp(int* x) {
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0xB1, 0xf, 0xf, false);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x4E, 0xf, 0xf, false);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x141, 0xf, 0xf, false);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x140, 0xf, 0xf, false);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x142, 0xa, 0xf, false);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] += __builtin_amdgcn_update_dpp(0, x[i], 0x143, 0xc, 0xf, false);
}

// Wave per keypoint, no LDS: lane l loads the tile dwords e = l + 64 k (k < 5) of the 31 x 36-byte
// tile straight into registers -- for the NEXT keypoint while the current one is reduced -- and
// multiplies them with per-lane constant weights: v_dot4_u32_u8 with the per-byte weights u + 15
// gives sum (u + 15) I, with the disc flags it gives sum I, and m10 = sum (u+15) I - 15 sum I,
// m01 = sum over dwords of v * (sum I of the dword).  All integer, exact.  The weights depend on
// the alignment (x - 15) & 3 of the tile: four register sets, picked by a wave-uniform switch.
__global__ __launch_bounds__(256) void moments_kernel(BatchView v, DetectorParams P, int2* __restrict__ moments, int nframes,
                                                      int blocks_per_frame, int use_visit) {
    int f, bx;
    if (!frame_block(nframes, blocks_per_frame, &f, &bx)) {
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int n = v.n_kp[f];
    const int radius = P.patch / 2;
    const int stride = blocks_per_frame * 4;
    const int i0 = bx * 4 + wave;
    if (i0 >= n) {
        return;
    }
    constexpr int SLOTS = 5;
    int off[SLOTS], vrow[SLOTS];
    uint2 wgt[4][SLOTS];
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int e = lane + 64 * k;
        const int ee = e < MOM_ROWS * MOM_DW ? e : 0;  // surplus slots re-read dword 0 with zero weights
        const int r = ee / MOM_DW;
        off[k] = r * v.pitch + (ee - r * MOM_DW) * 4;
        vrow[k] = r - MOM_HALF;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            wgt[a][k] = P.moment_weights[(a * SLOTS + k) * 64 + lane];
        }
    }
    const uint8_t* img = v.blurred + static_cast<size_t>(f) * v.frame_stride;
    const slamx_keypoint* kps = v.kps + static_cast<size_t>(f) * v.maxK;
    const uint32_t* visit = v.visit + static_cast<size_t>(f) * v.maxK;
    auto is_inside = [&](int x, int y) {
        return !(x - radius < 0 || x + radius >= v.cols || y - radius < 0 || y + radius >= v.rows);
    };
    // The tile is ALWAYS 31 rows x 36 bytes around the keypoint (the weights outside the patch's disc are zero), whatever
    // PatchSize is: "inside" only keeps the PATCH in the image, so with a small patch (PatchSize 5: keypoints 3 pixels from
    // the border are inside) the tile's first rows lie ABOVE the frame and its last ones below it.  Until round 4 the loads
    // were plain pointer reads -- before the first frame of a batch or behind its last one that is memory the batch does
    // not own (found by tests/fuzz/fuzz_detector.py as a GPU memory fault: 72 x 525, PatchSize 5, NumBRIEFPairs 24).  Buffer
    // loads with the whole offset in the lane: what lies before the frame (a negative offset, i.e. a huge unsigned one)
    // or behind it reads as zero.
    const __amdgpu_buffer_rsrc_t frame_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, v.rows * v.pitch, 0x00020000);
    auto issue = [&](int x, int y, uint32_t* px) {
        const int base = (y - MOM_HALF) * v.pitch + ((x - MOM_HALF) & ~3);  // may be negative
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            px[k] = __builtin_amdgcn_raw_buffer_load_b32(frame_rsrc, base + off[k], 0, 0);
        }
    };
    for (int chunk = 0; i0 + chunk * stride < n; chunk += 64) {  // wave-uniform
        // lane j holds the slot and coordinates of this wave's (chunk + j)-th keypoint
        const int mine = i0 + (chunk + lane) * stride;
        int kx = 0, ky = 0, kslot = 0;
        if (mine < n) {
            kslot = use_visit ? static_cast<int>(visit[mine]) : mine;
            kx = kp_coord(kps[kslot].x);
            ky = kp_coord(kps[kslot].y);
        }
        const int cnt = min(64, (n - 1 - (i0 + chunk * stride)) / stride + 1);
        int x = __builtin_amdgcn_readlane(kx, 0);
        int y = __builtin_amdgcn_readlane(ky, 0);
        uint32_t nxt[SLOTS] = {0, 0, 0, 0, 0};
        if (is_inside(x, y)) {
            issue(x, y, nxt);
        }
        for (int k = 0; k < cnt; ++k) {
            const int slot = lane_value(kslot, k);
            const bool inside = is_inside(x, y);  // wave-uniform
            uint32_t cur[SLOTS];
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) {
                cur[q] = nxt[q];
            }
            const int xn = lane_value(kx, (k + 1) & 63);
            const int yn = lane_value(ky, (k + 1) & 63);
            if (k + 1 < cnt && is_inside(xn, yn)) {
                issue(xn, yn, nxt);
            }
            int m01 = 0;
            int m10 = 0x7fffffff;  // marks "patch leaves the image"
            if (inside) {
                uint32_t su = 0, s = 0;
                int sv = 0;
                auto reduce = [&](const uint2* w) {
#pragma unroll
                    for (int q = 0; q < SLOTS; ++q) {
                        su = __builtin_amdgcn_udot4(cur[q], w[q].x, su, false);
                        const uint32_t row = __builtin_amdgcn_udot4(cur[q], w[q].y, 0u, false);
                        s += row;
                        sv += vrow[q] * static_cast<int>(row);
                    }
                };
                switch ((x - MOM_HALF) & 3) {  // wave-uniform
                    case 0: reduce(wgt[0]); break;
                    case 1: reduce(wgt[1]); break;
                    case 2: reduce(wgt[2]); break;
                    default: reduce(wgt[3]); break;
                }
                m10 = wave_sum_dpp(static_cast<int>(su) - MOM_HALF * static_cast<int>(s));
                m01 = wave_sum_dpp(sv);
            }
            if (lane == 0) {
                moments[static_cast<size_t>(f) * v.maxK + slot] = make_int2(m01, m10);
            }
            x = xn;
            y = yn;
        }
    }
}

// host side of the table described in slamx_internal.h
}  // namespace

void moment_weight_table(int radius, uint32_t* out_xy) {
    for (int a = 0; a < 4; ++a) {
        for (int k = 0; k < 5; ++k) {
            for (int lane = 0; lane < 64; ++lane) {
                const int e = lane + 64 * k;
                uint32_t wu = 0, wm = 0;
                if (e < MOM_ROWS * MOM_DW) {
                    const int r = e / MOM_DW;
                    const int c = e - r * MOM_DW;
                    const int vv = r - MOM_HALF;
                    for (int b = 0; b < 4; ++b) {
                        const int uu = 4 * c + b - (MOM_HALF + a);
                        if (uu >= -MOM_HALF && uu <= MOM_HALF && uu * uu + vv * vv <= radius * radius) {
                            wu |= static_cast<uint32_t>(uu + MOM_HALF) << (8 * b);
                            wm |= 1u << (8 * b);
                        }
                    }
                }
                const size_t idx = (static_cast<size_t>(a) * 5 + k) * 64 + lane;
                out_xy[2 * idx] = wu;
                out_xy[2 * idx + 1] = wm;
            }
        }
    }
    // the patch itself, 31 rows x 8 dwords starting AT column x - 15 (describe_fused_kernel shifts the
    // data into place with v_alignbyte instead of choosing among four weight sets)
    for (int k = 0; k < MOMENT_ALIGNED_SLOTS; ++k) {
        for (int lane = 0; lane < 64; ++lane) {
            const int e = lane + 64 * k;
            uint32_t wu = 0, wm = 0;
            if (e < MOM_ROWS * 8) {
                const int vv = e / 8 - MOM_HALF;
                for (int b = 0; b < 4; ++b) {
                    const int uu = 4 * (e % 8) + b - MOM_HALF;
                    if (uu <= MOM_HALF && uu * uu + vv * vv <= radius * radius) {
                        wu |= static_cast<uint32_t>(uu + MOM_HALF) << (8 * b);
                        wm |= 1u << (8 * b);
                    }
                }
            }
            const size_t idx = static_cast<size_t>(4 * 5 * 64) + static_cast<size_t>(k) * 64 + lane;
            out_xy[2 * idx] = wu;
            out_xy[2 * idx + 1] = wm;
        }
    }
}

namespace {

// lane per keypoint: (m01, m10) -> angle -> cos, sin.  m10 == INT_MAX marks "patch leaves the
// image": angle 0 and an all-zero descriptor (feature_detector.cpp:211-214, :242-245).
__global__ __launch_bounds__(256) void trig_kernel(BatchView v, const int2* __restrict__ moments,
                                                   float2* __restrict__ cs) {
    const int f = blockIdx.y;
    const int n = v.n_kp[f];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int2 m = moments[static_cast<size_t>(f) * v.maxK + i];
        float angle = 0.0f;
        float2 r = make_float2(0.0f, 0.0f);
        if (m.y != 0x7fffffff) {
            angle = __fmul_rn(glibc_atan2f(static_cast<float>(m.x), static_cast<float>(m.y)), rad_to_deg());
            const float a = __fmul_rn(angle, deg_to_rad());
            glibc_sincosf(a, &r.y, &r.x);
        } else {
            r.x = __uint_as_float(0x7fc00000u);  // NaN cos marks "outside" for brief_kernel
        }
        v.kps[static_cast<size_t>(f) * v.maxK + i].angle = angle;  // feature_detector.cpp:37-38
        cs[static_cast<size_t>(f) * v.maxK + i] = r;
    }
}

__global__ __launch_bounds__(256) void brief_kernel(BatchView v, DetectorParams P, const int8_t* __restrict__ pattern,
                                                    const float2* __restrict__ cs, int nframes, int blocks_per_frame,
                                                    int use_visit) {
    __shared__ uint32_t tile[4][TILE_ROWS * TILE_DW];
    __shared__ unsigned long long dbuf[4][DESC_WORDS];
    int f, bx;
    if (!frame_block(nframes, blocks_per_frame, &f, &bx)) {
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int n = v.n_kp[f];
    const uint8_t* img = v.blurred + static_cast<size_t>(f) * v.frame_stride;
    const uint8_t* t8 = reinterpret_cast<const uint8_t*>(tile[wave]);
    const slamx_keypoint* kps = v.kps + static_cast<size_t>(f) * v.maxK;
    const float2* csf = cs + static_cast<size_t>(f) * v.maxK;
    const uint32_t* visit = v.visit + static_cast<size_t>(f) * v.maxK;
    const int stride = blocks_per_frame * 4;
    const int i0 = bx * 4 + wave;
    const int words = (v.width + 7) / 8;
    // this lane's pattern pair of the first 64-pair chunk (loop invariant)
    float q1x = 0.0f, q1y = 0.0f, q2x = 0.0f, q2y = 0.0f;
    if (lane < P.pattern_len) {
        q1x = static_cast<float>(pattern[4 * lane + 0]);
        q1y = static_cast<float>(pattern[4 * lane + 1]);
        q2x = static_cast<float>(pattern[4 * lane + 2]);
        q2y = static_cast<float>(pattern[4 * lane + 3]);
    }
    TilePrefetch<TILE_ROWS, TILE_DW, TILE_HALF> pf;
    int tile_off[TilePrefetch<TILE_ROWS, TILE_DW, TILE_HALF>::PER_LANE];
    pf.init_offsets(v.pitch, lane, tile_off);
    const __amdgpu_buffer_rsrc_t frame_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, v.rows * v.pitch, 0x00020000);
    auto fetch = [&](int x, int y) {  // wave-uniform choice: no per-element bounds checks when the tile is inside
        const int xa = (x - TILE_HALF) & ~3;
        if (x - TILE_HALF >= 0 && xa + TILE_DW * 4 <= v.pitch && y - TILE_HALF >= 0 && y + TILE_HALF < v.rows) {
            pf.issue_inside(frame_rsrc, v.pitch, x, y, tile_off);
        } else {
            pf.issue(v, img, x, y, lane);
        }
    };
    const bool one_chunk = P.pattern_len <= 64 && (v.width & 7) == 0;
    const unsigned long long all_pairs = low_mask_d(P.pattern_len < 64 ? P.pattern_len : 64);
    for (int chunk = 0; i0 + chunk * stride < n; chunk += 64) {  // wave-uniform
        const int mine = i0 + (chunk + lane) * stride;
        int kx = 0, ky = 0, kslot = 0;
        float kc = 0.0f, ks = 0.0f;
        if (mine < n) {
            kslot = use_visit ? static_cast<int>(visit[mine]) : mine;
            kx = kp_coord(kps[kslot].x);
            ky = kp_coord(kps[kslot].y);
            const float2 r = csf[kslot];
            kc = r.x;
            ks = r.y;
        }
        const int cnt = min(64, (n - 1 - (i0 + chunk * stride)) / stride + 1);
        int x = lane_value(kx, 0);
        int y = lane_value(ky, 0);
        float ca = lane_value(kc, 0);
        float sa = lane_value(ks, 0);
        if (ca == ca) {
            fetch(x, y);
        }
        for (int k = 0; k < cnt; ++k) {
            const int i = lane_value(kslot, k);
            uint8_t* desc = v.desc + (static_cast<size_t>(f) * v.maxK + i) * v.width;
            const bool inside = ca == ca;  // NaN cos marks "patch leaves the image" (wave-uniform)
            if (!one_chunk && lane < words) {
                dbuf[wave][lane] = 0ull;
            }
            unsigned long long first_word = 0;  // the descriptor itself when the pattern fits one chunk
            if (inside) {
                pf.commit(tile[wave], lane);
            }
            const int xn = lane_value(kx, (k + 1) & 63);
            const int yn = lane_value(ky, (k + 1) & 63);
            const float can = lane_value(kc, (k + 1) & 63);
            const float san = lane_value(ks, (k + 1) & 63);
            if (k + 1 < cnt && can == can) {
                fetch(xn, yn);
            }
            if (inside) {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int cx = x - ((x - TILE_HALF) & ~3);  // pixel (x+du, y+dv) at t8[(dv+21)*48 + cx + du]
                int bitIndex = 0;
                for (int base = 0; base < P.pattern_len; base += 64) {
                    const int pi = base + lane;
                    bool valid = false;
                    bool bit = false;
                    if (pi < P.pattern_len) {
                        float p1x = q1x, p1y = q1y, p2x = q2x, p2y = q2y;
                        if (base != 0) {
                            p1x = static_cast<float>(pattern[4 * pi + 0]);
                            p1y = static_cast<float>(pattern[4 * pi + 1]);
                            p2x = static_cast<float>(pattern[4 * pi + 2]);
                            p2y = static_cast<float>(pattern[4 * pi + 3]);
                        }
                        const int dx1 = static_cast<int>(__fsub_rn(__fmul_rn(p1x, ca), __fmul_rn(p1y, sa)));
                        const int dy1 = static_cast<int>(__fadd_rn(__fmul_rn(p1x, sa), __fmul_rn(p1y, ca)));
                        const int dx2 = static_cast<int>(__fsub_rn(__fmul_rn(p2x, ca), __fmul_rn(p2y, sa)));
                        const int dy2 = static_cast<int>(__fadd_rn(__fmul_rn(p2x, sa), __fmul_rn(p2y, ca)));
                        const int x1 = dx1 + x, y1 = dy1 + y, x2 = dx2 + x, y2 = dy2 + y;
                        valid = x1 >= 0 && x1 < v.cols && y1 >= 0 && y1 < v.rows && x2 >= 0 && x2 < v.cols &&
                                y2 >= 0 && y2 < v.rows;
                        if (valid) {
                            const uint8_t I1 = t8[(dy1 + TILE_HALF) * (TILE_DW * 4) + cx + dx1];
                            const uint8_t I2 = t8[(dy2 + TILE_HALF) * (TILE_DW * 4) + cx + dx2];
                            bit = I1 < I2;
                        }
                    }
                    const unsigned long long vm = __ballot(valid);
                    const unsigned long long bm = __ballot(valid && bit);
                    const int c2 = __popcll(vm);
                    unsigned long long packed = bm;  // every pair of the chunk valid: nothing to close up
                    if (vm != (base == 0 ? all_pairs : low_mask_d(min(64, P.pattern_len - base)))) {
                        bool outbit = false;
                        if (lane < c2) {
                            outbit = ((bm >> select64d(vm, lane)) & 1ull) != 0;
                        }
                        packed = __ballot(outbit);
                    }
                    if (one_chunk) {
                        first_word = packed;
                    } else if (lane == 0 && c2 > 0) {
                        const int w = bitIndex >> 6;
                        const int sh = bitIndex & 63;
                        dbuf[wave][w] |= packed << sh;
                        if (sh != 0 && sh + c2 > 64) {
                            dbuf[wave][w + 1] |= packed >> (64 - sh);
                        }
                    }
                    bitIndex += c2;
                }
            }
            if (one_chunk) {
                if (lane < words) {  // descriptor rows are 8-byte aligned (width % 8 == 0)
                    reinterpret_cast<unsigned long long*>(desc)[lane] = lane == 0 ? first_word : 0ull;
                }
            } else {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint8_t* d8 = reinterpret_cast<const uint8_t*>(dbuf[wave]);
                for (int b = lane; b < v.width; b += 64) {
                    desc[b] = d8[b];
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            x = xn;
            y = yn;
            ca = can;
            sa = san;
        }
    }
}

// Which kernel describes a frame is decided per frame, on the device: the shared tiles cost ~4 us of latency per
// region (tile load with the list fetch under it, barriers; 4 workgroups per CU) and 0.22 ns per keypoint, the
// per-keypoint tiles of describe_fused_kernel 0.34-0.50 ns per keypoint -- the tiles win from ~14 keypoints per region on
// (measured: 1080p with 2.1k keypoints = 13 per region 0.53 against 0.51 ms per 512 frames; 10.4k keypoints = 77 per
// region 0.74 against 0.97 ms per 256; 640 x 480 with 2.0k = 83 per region 0.54 against 0.70 ms per 1024); the threshold
// keeps a margin above the break-even.  Both kernels are launched, each leaves the other's frames alone.
constexpr int RT_MIN_KEYPOINTS_PER_REGION = 20;
__device__ __forceinline__ bool frame_takes_tiles(int n_kp, int tile_min) { return n_kp >= tile_min; }

// ---- orientation + BRIEF in ONE pass over the blurred frame (PatchSize <= 31, <= 64 surviving pairs) ----
// moments_kernel and brief_kernel each read every blurred frame from HBM once (the patches of ~2k
// keypoints cover a 1080p frame), which is all their time.  Here a wave takes 4 keypoints per round and
// keeps their 43 x 48-byte tiles in LDS across three phases (no workgroup barriers):
//   A  per keypoint: tile -> LDS; the moments patch is dword-aligned inside that tile (its first
//      column (x-15) & ~3 is 4 or 8 bytes right of the tile's (x-21) & ~3), so the moments are the
//      same v_dot4 sums as in moments_kernel, read from LDS
//   B  atan2f -> degrees -> cosf, sinf, one keypoint per lane (the serial glibc chain)
//   C  per keypoint: the BRIEF samples from the tile that is still in LDS
// One HBM read of the frame instead of two; results identical to the three-kernel path.
// FUSED_PER_WAVE keypoints (= LDS tiles) travel with a wave through the three phases; WPB waves per
// workgroup; OCC = waves per SIMD the register allocation must leave room for.
#ifndef FUSED_EARLY_TILES
#define FUSED_EARLY_TILES 1
#endif
template <int FUSED_PER_WAVE, int WPB, int OCC>
__global__ __launch_bounds__(64 * WPB, OCC) void describe_fused_kernel(BatchView v, DetectorParams P, const int8_t* __restrict__ pattern,
                                                                      int nframes, int blocks_per_frame, int use_visit, int tile_min) {
    __shared__ uint32_t tiles[WPB][FUSED_PER_WAVE][TILE_ROWS * TILE_DW];
    int f, bx;
    if (!frame_block(nframes, blocks_per_frame, &f, &bx)) {
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int n = v.n_kp[f];
    if (frame_takes_tiles(n, tile_min)) {
        return;  // a dense frame: describe_tile_kernel takes it (uniform)
    }
    const int radius = P.patch / 2;
    const uint8_t* img = v.blurred + static_cast<size_t>(f) * v.frame_stride;
    slamx_keypoint* kps = v.kps + static_cast<size_t>(f) * v.maxK;
    const uint32_t* visit = v.visit + static_cast<size_t>(f) * v.maxK;
    const int words = v.width / 8;
    // per-lane constants: tile dword offsets, moments slots and weights, this lane's pattern pair
    using Prefetch = TilePrefetch12<TILE_ROWS, TILE_HALF>;
    static_assert(TILE_DW == Prefetch::DW, "tile rows are four 12-byte pieces");
    Prefetch pf[FUSED_PER_WAVE];  // the tiles of the NEXT group travel in registers while this one is worked on
    int tile_off[Prefetch::PER_LANE];
    pf[0].init_offsets(v.pitch, lane, tile_off);
    const __amdgpu_buffer_rsrc_t frame_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, v.rows * v.pitch, 0x00020000);
    // orientation patch = 31 rows x 8 dwords that start AT column x - 15: slot k of a lane is patch dword
    // e = lane + 64 k (row e / 8, dword e % 8), whose bytes straddle two tile dwords
    constexpr int SLOTS = MOMENT_ALIGNED_SLOTS;
    int mom_idx[SLOTS], vrow[SLOTS];
    uint2 wal[SLOTS];
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int e = lane + 64 * k;
        const int ee = e < MOM_ROWS * 8 ? e : 0;  // surplus slots carry zero weights
        const int r = ee / 8;
        mom_idx[k] = (r + (TILE_HALF - MOM_HALF)) * TILE_DW + (ee - r * 8);
        vrow[k] = r - MOM_HALF;
        wal[k] = P.moment_weights[4 * 5 * 64 + k * 64 + lane];
    }
    float q1x = 0.0f, q1y = 0.0f, q2x = 0.0f, q2y = 0.0f;
    if (lane < P.pattern_len) {
        q1x = static_cast<float>(pattern[4 * lane + 0]);
        q1y = static_cast<float>(pattern[4 * lane + 1]);
        q2x = static_cast<float>(pattern[4 * lane + 2]);
        q2y = static_cast<float>(pattern[4 * lane + 3]);
    }
    const unsigned long long all_pairs = low_mask_d(P.pattern_len);
    auto patch_inside = [&](int x, int y) {
        return !(x - radius < 0 || x + radius >= v.cols || y - radius < 0 || y + radius >= v.rows);
    };
    auto wave_fence = [] {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // lane q < FUSED_PER_WAVE of a group's (slot, x, y) holds keypoint q; slot < 0 = no keypoint
    auto load_group = [&](int g, int& kslot, int& kx, int& ky) {
        kslot = -1;
        kx = 0;
        ky = 0;
        const int idx = g * FUSED_PER_WAVE + lane;
        if (lane < FUSED_PER_WAVE && idx < n) {
            kslot = use_visit ? static_cast<int>(visit[idx]) : idx;
            kx = kp_coord(kps[kslot].x);
            ky = kp_coord(kps[kslot].y);
        }
    };
    // Two wave-uniform bit masks of a group, evaluated once by the lanes that hold its keypoints: bit q =
    // keypoint q exists and its orientation patch is inside the image / bit 16 + q = its whole tile lies
    // inside the padded frame (no per-element bounds checks when loading it)
    auto group_masks = [&](int kslot, int kx, int ky) {
        const bool in = kslot >= 0 && patch_inside(kx, ky);
        const int xa = (kx - TILE_HALF) & ~3;
        const bool whole = kx - TILE_HALF >= 0 && xa + TILE_DW * 4 <= v.pitch && ky - TILE_HALF >= 0 && ky + TILE_HALF < v.rows;
        const uint32_t a = static_cast<uint32_t>(__ballot(in)) & ((1u << FUSED_PER_WAVE) - 1u);
        const uint32_t b = static_cast<uint32_t>(__ballot(in && whole)) & ((1u << FUSED_PER_WAVE) - 1u);
        return a | (b << 16);
    };
    auto issue_group = [&](uint32_t masks, int kx, int ky, int q_first, int q_end) {  // global loads of tiles, no waiting
#pragma unroll
        for (int q = 0; q < FUSED_PER_WAVE; ++q) {
            if (q < q_first || q >= q_end) {
                continue;
            }
            if ((masks >> q) & 1u) {  // wave-uniform
                const int x = lane_value(kx, q), y = lane_value(ky, q);
                if ((masks >> (16 + q)) & 1u) {
                    pf[q].issue_inside(frame_rsrc, v.pitch, x, y, tile_off);
                } else {
                    pf[q].issue(v, img, x, y, lane);
                }
            }
        }
    };
    // The waves are independent: wave `gw` of the frame takes keypoint groups gw, gw + nwaves, ...  The
    // coordinates of the next group are loaded a round ahead and its tiles are requested -- all at once,
    // into registers -- before the BRIEF phase of the current group (the phase with the fewest live
    // registers), so that no chain of coordinate and tile loads is left on a wave's critical path.
    constexpr int EARLY_TILES = FUSED_EARLY_TILES;  // tiles requested a phase early (register budget)
    const int nwaves = blocks_per_frame * WPB;
    int g = bx * WPB + wave;
    int kslot, kx, ky;
    load_group(g, kslot, kx, ky);
    uint32_t masks = group_masks(kslot, kx, ky);
    issue_group(masks, kx, ky, 0, EARLY_TILES);
    for (; g * FUSED_PER_WAVE < n; g += nwaves) {  // wave-uniform
        issue_group(masks, kx, ky, EARLY_TILES, FUSED_PER_WAVE);
        int kslot1, kx1, ky1;
        load_group(g + nwaves, kslot1, kx1, ky1);
        // ---- phase A: tiles registers -> LDS, moments (lane q keeps m01, m10 of keypoint q)
        int my_m01 = 0, my_m10 = 0x7fffffff;
        int part[2 * FUSED_PER_WAVE];  // per-lane partial sums: m10 of keypoint q, then m01 of keypoint q
        const uint32_t inside_mask = masks;
#pragma unroll
        for (int q = 0; q < FUSED_PER_WAVE; ++q) {
            part[q] = 0;
            part[FUSED_PER_WAVE + q] = 0;
            if ((masks >> q) & 1u) {  // wave-uniform
                const int x = lane_value(kx, q);
                pf[q].commit(tiles[wave][q], lane);
                wave_fence();
                const int po = (x - MOM_HALF) - ((x - TILE_HALF) & ~3);  // patch column 0 inside the tile: 6..9
                const uint32_t* t = tiles[wave][q] + (po >> 2);
                const uint32_t sh = static_cast<uint32_t>(po & 3);
                uint32_t su = 0, srow = 0;
                int sv = 0;
#pragma unroll
                for (int k = 0; k < SLOTS; ++k) {
                    const uint32_t lo = t[mom_idx[k]], hi = t[mom_idx[k] + 1];
                    const uint32_t cur = __builtin_amdgcn_alignbyte(hi, lo, sh);
                    su = __builtin_amdgcn_udot4(cur, wal[k].x, su, false);
                    const uint32_t row = __builtin_amdgcn_udot4(cur, wal[k].y, 0u, false);
                    srow += row;
                    sv += vrow[k] * static_cast<int>(row);
                }
                part[q] = static_cast<int>(su) - MOM_HALF * static_cast<int>(srow);
                part[FUSED_PER_WAVE + q] = sv;
            }
        }
        // the 2 x FUSED_PER_WAVE wave sums together, step by step: a DPP add that reads the result of the
        // previous one needs two idle issue slots, which the other chains fill
        wave_sums_dpp<2 * FUSED_PER_WAVE>(part);
#pragma unroll
        for (int q = 0; q < FUSED_PER_WAVE; ++q) {
            // readlane in uniform control flow: under `lane == q` the compiler may sink the last add into
            // the branch, where lane 63 is switched off, and readlane would return a stale value
            const int m10 = __builtin_amdgcn_readlane(part[q], 63);
            const int m01 = __builtin_amdgcn_readlane(part[FUSED_PER_WAVE + q], 63);
            if (((inside_mask >> q) & 1u) != 0 && lane == q) {
                my_m10 = m10;
                my_m01 = m01;
            }
        }
        // ---- phase B: angle, cos, sin -- lane q for keypoint q (the serial glibc chain)
        float my_cos = __uint_as_float(0x7fc00000u);  // NaN cos marks "outside"
        float my_sin = 0.0f;
        if (kslot >= 0) {
            float angle = 0.0f;
            if (my_m10 != 0x7fffffff) {
                angle = __fmul_rn(glibc_atan2f(static_cast<float>(my_m01), static_cast<float>(my_m10)), rad_to_deg());
                const float ar = __fmul_rn(angle, deg_to_rad());
                glibc_sincosf(ar, &my_sin, &my_cos);
            }
            kps[kslot].angle = angle;  // feature_detector.cpp:37-38
        }
        // the next group's tiles: in flight (in registers) while this group's descriptors are sampled
        const uint32_t masks1 = group_masks(kslot1, kx1, ky1);
        issue_group(masks1, kx1, ky1, 0, EARLY_TILES);
        // ---- phase C: BRIEF from the tiles still in LDS
#pragma unroll
        for (int q = 0; q < FUSED_PER_WAVE; ++q) {
            const int slot = lane_value(kslot, q);
            if (slot < 0) {
                continue;  // wave-uniform
            }
            const int x = lane_value(kx, q);
            const int y = lane_value(ky, q);
            const float ca = lane_value(my_cos, q);
            const float sa = lane_value(my_sin, q);
            unsigned long long packed = 0;
            if (ca == ca) {
                const uint8_t* t8 = reinterpret_cast<const uint8_t*>(tiles[wave][q]);
                const int cx = x - ((x - TILE_HALF) & ~3);  // pixel (x+du, y+dv) at t8[(dv+21)*48 + cx + du]
                // Straight-line code: a rotated offset stays within +-21 of the keypoint, i.e. inside the tile,
                // whether or not the pixel is inside the image, so both samples are read unconditionally and
                // the image test only masks the result (written with short-circuit && / nested ifs this body
                // compiled to four levels of exec-mask branches; lanes >= pattern_len carry zero offsets).
                const int dx1 = static_cast<int>(__fsub_rn(__fmul_rn(q1x, ca), __fmul_rn(q1y, sa)));
                const int dy1 = static_cast<int>(__fadd_rn(__fmul_rn(q1x, sa), __fmul_rn(q1y, ca)));
                const int dx2 = static_cast<int>(__fsub_rn(__fmul_rn(q2x, ca), __fmul_rn(q2y, sa)));
                const int dy2 = static_cast<int>(__fadd_rn(__fmul_rn(q2x, sa), __fmul_rn(q2y, ca)));
                const uint32_t I1 = t8[(dy1 + TILE_HALF) * (TILE_DW * 4) + cx + dx1];
                const uint32_t I2 = t8[(dy2 + TILE_HALF) * (TILE_DW * 4) + cx + dx2];
                const bool valid = static_cast<int>(static_cast<uint32_t>(dx1 + x) < static_cast<uint32_t>(v.cols)) &
                                   static_cast<int>(static_cast<uint32_t>(dy1 + y) < static_cast<uint32_t>(v.rows)) &
                                   static_cast<int>(static_cast<uint32_t>(dx2 + x) < static_cast<uint32_t>(v.cols)) &
                                   static_cast<int>(static_cast<uint32_t>(dy2 + y) < static_cast<uint32_t>(v.rows)) &
                                   static_cast<int>(lane < P.pattern_len);
                const bool bit = I1 < I2;
                const unsigned long long vm = __ballot(valid);
                const unsigned long long bm = __ballot(valid && bit);
                packed = bm;  // every pair valid: nothing to close up
                if (vm != all_pairs) {
                    const int c2 = __popcll(vm);
                    bool outbit = false;
                    if (lane < c2) {
                        outbit = ((bm >> select64d(vm, lane)) & 1ull) != 0;
                    }
                    packed = __ballot(outbit);
                }
            }
            if (lane < words) {  // descriptor rows are 8-byte aligned (width % 8 == 0)
                uint8_t* desc = v.desc + (static_cast<size_t>(f) * v.maxK + slot) * v.width;
                reinterpret_cast<unsigned long long*>(desc)[lane] = lane == 0 ? packed : 0ull;
            }
        }
        wave_fence();  // the tiles are overwritten by the next group
        kslot = kslot1, kx = kx1, ky = ky1;
        masks = masks1;
    }
}

// ---- orientation + BRIEF from SHARED image tiles (keypoints in detection order: the visiting list of the NMS) ----
// describe_fused_kernel gives every keypoint its own 43 x 48-byte tile: three gathered loads and three LDS stores per
// keypoint, a trig chain on 4 of 64 lanes per group, and ~80 scalar instructions per keypoint of pipelining around
// it (255 issue slots per keypoint in all, which is its duration).  Here a workgroup owns a REGION of the frame --
// 240 x 64 pixels -- stages the blurred region with its 21-pixel halo once (288 bytes x 106 rows = 30.5 KB, wide
// coalesced buffer loads, zero fill above / below the frame) and serves every keypoint whose centre lies in the region
// from that one tile in three phases separated by workgroup barriers:
//   A  moments: a wave takes 4 keypoints per step, the 31 x 32-byte patch read straight from the shared tile
//      (ds_read2 + v_alignbyte + v_dot4 as before; the 8 wave sums reduced together)
//   B  atan2f -> degrees -> cosf, sinf: ONE KEYPOINT PER LANE over all keypoints of the region
//   C  BRIEF: a wave per keypoint, lane = pattern pair, byte reads from the tile
// The keypoints of a region come from region_index_kernel (a counting sort of the frame's keypoints by region, one
// workgroup per frame).  A tile row is 72 dwords = 8 mod 64 banks: the 8 rows x 8 dwords a wave reads per moment slot
// cover the 64 banks exactly once.
constexpr int RT_W = 240;
constexpr int RT_H = 64;
constexpr int RT_DW = 72;                       // dwords per tile row (288 bytes: columns x0 - 24 .. x0 + 263)
constexpr int RT_XOFF = 24;
constexpr int RT_ROWS = RT_H + 2 * TILE_HALF;   // 106: rows y0 - 21 .. y0 + 84
constexpr int RT_PIECES = RT_ROWS * (RT_DW / 4);
constexpr int RT_LIST = 512;                    // keypoints held per pass (a pass starts with fewer than 256)

// Keypoints of a frame binned by region: region_first[f][r] .. region_first[f][r + 1] delimit region r's slots in
// region_list[f] (any order inside a region: every keypoint writes only its own angle and descriptor row).  One
// workgroup per frame: histogram in LDS, prefix, scatter.  regions <= RT_MAX_REGIONS (else the caller keeps the
// per-keypoint tiles).
constexpr int RT_MAX_REGIONS = 4096;
__global__ __launch_bounds__(1024) void region_index_kernel(BatchView v, int regions_x, int regions, int tile_min) {
    __shared__ int hist[RT_MAX_REGIONS];
    __shared__ int wave_tot[16];
    const int f = blockIdx.x;
    const int n = v.n_kp[f];
    if (!frame_takes_tiles(n, tile_min)) {
        return;  // describe_fused_kernel takes this frame
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const slamx_keypoint* kps = v.kps + static_cast<size_t>(f) * v.maxK;
    const uint32_t* visit = v.visit + static_cast<size_t>(f) * v.maxK;
    int* first = v.region_first + static_cast<size_t>(f) * (regions + 1);
    int2* list = v.region_list + static_cast<size_t>(f) * v.maxK;  // {slot, y << 16 | x}
    for (int r = tid; r < regions; r += 1024) {
        hist[r] = 0;
    }
    __syncthreads();
    auto coords_of = [&](int slot) {  // clamped into the image: y << 16 | x
        const int x = min(max(kp_coord(kps[slot].x), 0), v.cols - 1);
        const int y = min(max(kp_coord(kps[slot].y), 0), v.rows - 1);
        return y << 16 | x;
    };
    auto region_of = [&](int xy) { return ((xy >> 16) >> 6) * regions_x + (xy & 0xffff) / RT_W; };
    for (int i = tid; i < n; i += 1024) {
        atomicAdd(&hist[region_of(coords_of(static_cast<int>(visit[i])))], 1);
    }
    __syncthreads();
    // exclusive prefix over the regions (block scan in chunks of 1024), in place: hist[r] becomes the region's cursor
    int carry = 0;
    for (int base = 0; base < regions; base += 1024) {
        const int r = base + tid;
        const int c = r < regions ? hist[r] : 0;
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            incl += lane >= d ? o : 0;
        }
        if (lane == 63) {
            wave_tot[wv] = incl;
        }
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            woff += k < wv ? wave_tot[k] : 0;
            tot += wave_tot[k];
        }
        if (r < regions) {
            const int excl = carry + woff + incl - c;
            hist[r] = excl;
            first[r] = excl;
        }
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) {
        first[regions] = n;
    }
    for (int i = tid; i < n; i += 1024) {
        const int slot = static_cast<int>(visit[i]);
        const int xy = coords_of(slot);
        list[atomicAdd(&hist[region_of(xy)], 1)] = make_int2(slot, xy);
    }
}

__global__ __launch_bounds__(256) void describe_tile_kernel(BatchView v, DetectorParams P, const int8_t* __restrict__ pattern,
                                                            int nframes, int regions_x, int regions_per_frame, int tile_min) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[RT_ROWS * RT_DW];
    __shared__ int l_slot[RT_LIST];
    __shared__ uint32_t l_xy[RT_LIST];   // (y - y0 + 21) << 16 | (x - x0 + 24): tile coordinates of the centre
    __shared__ int l_a[RT_LIST];         // m01, then cos bits
    __shared__ int l_b[RT_LIST];         // m10 (0x7fffffff = patch leaves the image), then sin bits
    int f, r;
    if (!frame_block(nframes, regions_per_frame, &f, &r)) {
        return;
    }
    const int by = r / regions_x;
    const int bx = r - by * regions_x;
    const int x0 = bx * RT_W, y0 = by * RT_H;
#ifdef SLAMX_DESC_CLOCKS
    const unsigned long long dc0 = clock64();  // development: phase stamps of a few regions of frame 0, printed at the end
    unsigned long long dc1 = 0, dc2 = 0, dc3 = 0, dc4 = 0;
    int dc_n = 0;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (!frame_takes_tiles(v.n_kp[f], tile_min)) {
        return;  // describe_fused_kernel takes this frame (uniform)
    }
    // The tile's loads go out FIRST (they need only the frame and the region; an empty region of a dense frame wastes
    // them, which is rare) and are in flight while the region's list is fetched: 16-byte pieces, rows above / below the
    // frame through an offset beyond the buffer (zero fill by the range check).
    const uint8_t* img = v.blurred + static_cast<size_t>(f) * v.frame_stride;
    constexpr int TILE_LOADS = (RT_PIECES + 255) / 256;
    uint4 tq[TILE_LOADS];
    {
        const __amdgpu_buffer_rsrc_t frame_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, v.rows * v.pitch, 0x00020000);
        const int soff = (y0 - TILE_HALF) * v.pitch + (x0 - RT_XOFF);  // may be negative: such rows read as zero
#pragma unroll
        for (int k = 0; k < TILE_LOADS; ++k) {
            const int p = tid + 256 * k;
            const int rr = p / (RT_DW / 4);
            const int cc = p - rr * (RT_DW / 4);
            const int yy = y0 - TILE_HALF + rr;
            const int voff = (p < RT_PIECES && yy >= 0 && yy < v.rows) ? rr * v.pitch + cc * 16 + soff : 0x7ffffff0;
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(frame_rsrc, voff, 0, 0);
            tq[k] = make_uint4(q[0], q[1], q[2], q[3]);
        }
        // A tile row starts RT_XOFF = 24 bytes left of the region, so in the frame's leftmost regions the 16-byte piece
        // of columns -8 .. 7 straddles x = 0.  In image rows >= 1 its offset is still inside the buffer (the bytes for
        // x < 0 come from the end of the row above: never sampled), but in IMAGE ROW 0 the offset is negative, the range
        // check zero-fills the whole piece, and columns 0 .. 7 of row 0 -- which BRIEF pairs and the moments of keypoints
        // near the top-left corner do sample -- read as 0.  Found by tests/fuzz/fuzz_detector.py (one descriptor bit of one
        // keypoint at (17, 18) on a 37 x 553 frame).  The thread that owns that piece of region 0 loads the eight valid
        // bytes on their own.
        constexpr int STRADDLE_PIECE = TILE_HALF * (RT_DW / 4) + RT_XOFF / 16;  // tile row of image row 0, piece of x = -8 .. 7
        static_assert(RT_XOFF % 16 == 8 && RT_XOFF / 16 == 1, "the straddling piece is the second one, half inside the frame");
        if (r == 0 && tid == STRADDLE_PIECE % 256) {  // (region 0: x0 == 0 and y0 == 0)
            const uint2 head = *reinterpret_cast<const uint2*>(img);
            tq[STRADDLE_PIECE / 256] = make_uint4(0u, 0u, head.x, head.y);
        }
    }
    const int* first = v.region_first + static_cast<size_t>(f) * (regions_per_frame + 1);
    const int lo = first[r], hi = first[r + 1];
    if (lo >= hi) {
        return;  // no keypoint in the region (uniform)
    }
    slamx_keypoint* kps = v.kps + static_cast<size_t>(f) * v.maxK;
    const int2* list = v.region_list + static_cast<size_t>(f) * v.maxK;
    const int radius = P.patch / 2;
    const int words = v.width / 8;
    // per-lane constants of the phases
    constexpr int SLOTS = MOMENT_ALIGNED_SLOTS;
    int mom_idx[SLOTS], vrow[SLOTS];
    uint2 wal[SLOTS];
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int e = lane + 64 * k;
        const int ee = e < MOM_ROWS * 8 ? e : 0;  // surplus slots carry zero weights
        const int rr = ee / 8;
        mom_idx[k] = rr * RT_DW + (ee - rr * 8);
        vrow[k] = rr - MOM_HALF;
        wal[k] = P.moment_weights[4 * 5 * 64 + k * 64 + lane];
    }
    float q1x = 0.0f, q1y = 0.0f, q2x = 0.0f, q2y = 0.0f;
    if (lane < P.pattern_len) {
        q1x = static_cast<float>(pattern[4 * lane + 0]);
        q1y = static_cast<float>(pattern[4 * lane + 1]);
        q2x = static_cast<float>(pattern[4 * lane + 2]);
        q2y = static_cast<float>(pattern[4 * lane + 3]);
    }
    const unsigned long long all_pairs = low_mask_d(P.pattern_len);
    bool tile_loaded = false;
    for (int base = lo; base < hi; base += RT_LIST) {  // uniform; one pass unless a region holds more than RT_LIST keypoints
        const int nl = min(RT_LIST, hi - base);
        for (int k = tid; k < nl; k += 256) {
            const int2 e = list[base + k];
            l_slot[k] = e.x;
            l_xy[k] = static_cast<uint32_t>((e.y >> 16) - y0 + TILE_HALF) << 16 | static_cast<uint32_t>((e.y & 0xffff) - x0 + RT_XOFF);
        }
        // ---- the tile, once: registers -> LDS
        if (!tile_loaded) {
            tile_loaded = true;
#pragma unroll
            for (int k = 0; k < TILE_LOADS; ++k) {
                const int p = tid + 256 * k;
                if (p < RT_PIECES) {
                    const int rr = p / (RT_DW / 4);
                    const int cc = p - rr * (RT_DW / 4);
                    *reinterpret_cast<uint4*>(&tile[rr * RT_DW + cc * 4]) = tq[k];
                }
            }
        }
        __syncthreads();
#ifdef SLAMX_DESC_CLOCKS
        dc1 = clock64();
        dc_n = nl;
#endif
        // ---- phase A: moments, a wave takes 4 keypoints per step
        for (int g = wave; g * 4 < nl; g += 4) {  // wave-uniform
            int part[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                part[q] = 0;
                part[4 + q] = 0;
                const int k = g * 4 + q;
                if (k < nl) {  // wave-uniform
                    const uint32_t xy = l_xy[k];
                    const int tx = static_cast<int>(xy & 0xffffu), ty = static_cast<int>(xy >> 16);
                    const int po = tx - MOM_HALF;  // first patch column inside the tile row
                    const uint32_t* t = tile + (ty - MOM_HALF) * RT_DW + (po >> 2);
                    const uint32_t sh = static_cast<uint32_t>(po & 3);
                    uint32_t su = 0, srow = 0;
                    int sv = 0;
#pragma unroll
                    for (int kk = 0; kk < SLOTS; ++kk) {
                        const uint32_t lo32 = t[mom_idx[kk]], hi32 = t[mom_idx[kk] + 1];
                        const uint32_t cur = __builtin_amdgcn_alignbyte(hi32, lo32, sh);
                        su = __builtin_amdgcn_udot4(cur, wal[kk].x, su, false);
                        const uint32_t row = __builtin_amdgcn_udot4(cur, wal[kk].y, 0u, false);
                        srow += row;
                        sv += vrow[kk] * static_cast<int>(row);
                    }
                    part[q] = static_cast<int>(su) - MOM_HALF * static_cast<int>(srow);
                    part[4 + q] = sv;
                }
            }
            wave_sums_dpp<8>(part);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m10 = __builtin_amdgcn_readlane(part[q], 63);
                const int m01 = __builtin_amdgcn_readlane(part[4 + q], 63);
                const int k = g * 4 + q;
                if (lane == q && k < nl) {
                    const uint32_t xy = l_xy[k];
                    const int x = static_cast<int>(xy & 0xffffu) - RT_XOFF + x0, y = static_cast<int>(xy >> 16) - TILE_HALF + y0;
                    const bool inside = !(x - radius < 0 || x + radius >= v.cols || y - radius < 0 || y + radius >= v.rows);
                    l_a[k] = m01;
                    l_b[k] = inside ? m10 : 0x7fffffff;
                }
            }
        }
        __syncthreads();
#ifdef SLAMX_DESC_CLOCKS
        dc2 = clock64();
#endif
        // ---- phase B: angle, cos, sin -- one keypoint per lane
        for (int k = tid; k < nl; k += 256) {
            const int m01 = l_a[k], m10 = l_b[k];
            float angle = 0.0f, c = __uint_as_float(0x7fc00000u), sn = 0.0f;  // NaN cos marks "outside"
            if (m10 != 0x7fffffff) {
                angle = __fmul_rn(glibc_atan2f(static_cast<float>(m01), static_cast<float>(m10)), rad_to_deg());
                const float ar = __fmul_rn(angle, deg_to_rad());
                glibc_sincosf(ar, &sn, &c);
            }
            kps[l_slot[k]].angle = angle;  // feature_detector.cpp:37-38
            l_a[k] = __float_as_int(c);
            l_b[k] = __float_as_int(sn);
        }
        __syncthreads();
#ifdef SLAMX_DESC_CLOCKS
        dc3 = clock64();
#endif
        // ---- phase C: BRIEF, a wave per keypoint, lane = pattern pair
        const uint8_t* t8 = reinterpret_cast<const uint8_t*>(tile);
        for (int k = wave; k < nl; k += 4) {  // wave-uniform
            const uint32_t xy = l_xy[k];
            const int tx = static_cast<int>(xy & 0xffffu), ty = static_cast<int>(xy >> 16);
            const int x = tx - RT_XOFF + x0, y = ty - TILE_HALF + y0;
            const float ca = __int_as_float(l_a[k]);
            const float sa = __int_as_float(l_b[k]);
            unsigned long long packed = 0;
            if (ca == ca) {  // wave-uniform
                const int dx1 = static_cast<int>(__fsub_rn(__fmul_rn(q1x, ca), __fmul_rn(q1y, sa)));
                const int dy1 = static_cast<int>(__fadd_rn(__fmul_rn(q1x, sa), __fmul_rn(q1y, ca)));
                const int dx2 = static_cast<int>(__fsub_rn(__fmul_rn(q2x, ca), __fmul_rn(q2y, sa)));
                const int dy2 = static_cast<int>(__fadd_rn(__fmul_rn(q2x, sa), __fmul_rn(q2y, ca)));
                const uint32_t I1 = t8[(dy1 + ty) * (RT_DW * 4) + tx + dx1];
                const uint32_t I2 = t8[(dy2 + ty) * (RT_DW * 4) + tx + dx2];
                const bool valid = static_cast<int>(static_cast<uint32_t>(dx1 + x) < static_cast<uint32_t>(v.cols)) &
                                   static_cast<int>(static_cast<uint32_t>(dy1 + y) < static_cast<uint32_t>(v.rows)) &
                                   static_cast<int>(static_cast<uint32_t>(dx2 + x) < static_cast<uint32_t>(v.cols)) &
                                   static_cast<int>(static_cast<uint32_t>(dy2 + y) < static_cast<uint32_t>(v.rows)) &
                                   static_cast<int>(lane < P.pattern_len);
                const bool bit = I1 < I2;
                const unsigned long long vm = __ballot(valid);
                const unsigned long long bm = __ballot(valid && bit);
                packed = bm;  // every pair valid: nothing to close up
                if (vm != all_pairs) {
                    const int c2 = __popcll(vm);
                    bool outbit = false;
                    if (lane < c2) {
                        outbit = ((bm >> select64d(vm, lane)) & 1ull) != 0;
                    }
                    packed = __ballot(outbit);
                }
            }
            if (lane < words) {  // descriptor rows are 8-byte aligned (width % 8 == 0)
                uint8_t* desc = v.desc + (static_cast<size_t>(f) * v.maxK + l_slot[k]) * v.width;
                reinterpret_cast<unsigned long long*>(desc)[lane] = lane == 0 ? packed : 0ull;
            }
        }
        __syncthreads();  // the lists are refilled by the next pass
#ifdef SLAMX_DESC_CLOCKS
        dc4 = clock64();
#endif
    }
#ifdef SLAMX_DESC_CLOCKS
    if (tid == 0 && f == 0 && r % 29 == 7) {
        printf("describe tile region %d: %d keypoints; cycles: tile + list %llu, moments %llu, trig %llu, brief %llu\n", r, dc_n,
               dc1 - dc0, dc2 - dc1, dc3 - dc2, dc4 - dc3);
    }
#endif
}

}  // namespace

int describe_tile_min_keypoints(const BatchView& v) {
    const int regions = ((v.cols + RT_W - 1) / RT_W) * v.nbands;
    return v.region_first != nullptr && regions <= RT_MAX_REGIONS ? RT_MIN_KEYPOINTS_PER_REGION * regions : 0x7fffffff;
}

void launch_describe(const BatchView& v, const DetectorParams& p, const int8_t* pattern_dev, int nframes, bool use_visit,
                     hipStream_t stream, hipEvent_t ev_after_moments, hipEvent_t ev_after_trig, bool allow_tiles) {
    // float accumulation of integer moments is exact (order-free) iff 255 * sum over the disc
    // of |v| stays below 2^24
    const int radius = p.patch / 2;
    long long sum_abs = 0;
    for (int vv = -radius; vv <= radius; ++vv) {
        for (int uu = -radius; uu <= radius; ++uu) {
            if (uu * uu + vv * vv <= radius * radius) {
                sum_abs += vv < 0 ? -vv : vv;
            }
        }
    }
    const int exact_int = (255LL * sum_abs < (1LL << 24)) ? 1 : 0;
    dim3 grid(grid_x(v.maxK, 4, nframes), nframes);
    // SLAMX_NO_FUSED_DESCRIBE=1 keeps the three-kernel path (used by the tests to cover it)
    static const bool fused_allowed = getenv("SLAMX_NO_FUSED_DESCRIBE") == nullptr;
    // SLAMX_DESCRIBE_TILES=0: per-keypoint tiles for every frame (describe_fused_kernel only); =2: shared region tiles for
    // every frame that has a keypoint (the parity tests run sparse frames through that kernel this way)
    static const int tiles_mode = getenv("SLAMX_DESCRIBE_TILES") == nullptr ? 1 : atoi(getenv("SLAMX_DESCRIBE_TILES"));
    if (fused_allowed && radius <= 15 && exact_int != 0 && p.moment_weights != nullptr && p.pattern_len <= 64 &&
        v.width % 8 == 0) {
        // keypoints in detection order: dense frames from shared region tiles, the others from per-keypoint tiles -- each
        // kernel checks the frame's keypoint count and leaves the other's frames alone (frame_takes_tiles)
        int tile_min = 0x7fffffff;  // keypoints from which a frame takes the shared tiles
        const int regions_x = (v.cols + RT_W - 1) / RT_W;
        const int tile_regions = regions_x * v.nbands;
        const int auto_min = RT_MIN_KEYPOINTS_PER_REGION * tile_regions;
        if (tiles_mode != 0 && use_visit && v.region_first != nullptr && tile_regions <= RT_MAX_REGIONS &&
            (tiles_mode == 2 || (v.maxK >= auto_min && allow_tiles))) {
            tile_min = tiles_mode == 2 ? 1 : auto_min;
            hipLaunchKernelGGL(region_index_kernel, dim3(nframes), dim3(1024), 0, stream, v, regions_x, tile_regions, tile_min);
            hipLaunchKernelGGL(describe_tile_kernel, dim3(frame_grid(nframes, tile_regions)), dim3(256), 0, stream, v, p, pattern_dev,
                               nframes, regions_x, tile_regions, tile_min);
        }
        const int waves_per_frame = static_cast<int>(grid.x) * 4;
        // Measured alternatives (more waves per SIMD by capping the registers at 96 / 80 / 64 with 4, 3 or 2
        // tiles per wave and 1-, 2- or 4-wave workgroups) all spill and ran 25 % to 2.4x slower.
        const int wpb = 4;
        const int bpf = (waves_per_frame + wpb - 1) / wpb;
        hipLaunchKernelGGL((describe_fused_kernel<4, 4, 4>), dim3(frame_grid(nframes, bpf)), dim3(64 * wpb), 0, stream, v, p,
                           pattern_dev, nframes, bpf, use_visit ? 1 : 0, tile_min);
        if (ev_after_moments != nullptr) {
            (void)hipEventRecord(ev_after_moments, stream);
        }
        if (ev_after_trig != nullptr) {
            (void)hipEventRecord(ev_after_trig, stream);
        }
        return;
    }
    if (radius <= 15 && exact_int != 0 && v.moments != nullptr) {
        const int bpf = static_cast<int>(grid.x);
        const dim3 grid1d(frame_grid(nframes, bpf));
        hipLaunchKernelGGL(moments_kernel, grid1d, dim3(256), 0, stream, v, p, v.moments, nframes, bpf, use_visit ? 1 : 0);
        if (ev_after_moments != nullptr) {
            (void)hipEventRecord(ev_after_moments, stream);
        }
        hipLaunchKernelGGL(trig_kernel, dim3(grid_x(v.maxK, 256, nframes), nframes), dim3(256), 0, stream, v, v.moments,
                           v.cos_sin);
        if (ev_after_trig != nullptr) {
            (void)hipEventRecord(ev_after_trig, stream);
        }
        hipLaunchKernelGGL(brief_kernel, grid1d, dim3(256), 0, stream, v, p, pattern_dev, v.cos_sin, nframes, bpf,
                           use_visit ? 1 : 0);
    } else {
        hipLaunchKernelGGL(describe_kernel, grid, dim3(256), 0, stream, v, p, pattern_dev, exact_int);
        if (ev_after_moments != nullptr) {
            (void)hipEventRecord(ev_after_moments, stream);
        }
        if (ev_after_trig != nullptr) {
            (void)hipEventRecord(ev_after_trig, stream);
        }
    }
}

}  // namespace slamx
