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
    const int wave = threadIdx.x >> 6;
    const int n = v.n_kp[f];
    const uint8_t* img = v.blurred + static_cast<size_t>(f) * v.frame_stride;
    for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {  // wave-uniform trip count
    slamx_keypoint* kp = v.kps + static_cast<size_t>(f) * v.maxK + i;
    uint8_t* desc = v.desc + (static_cast<size_t>(f) * v.maxK + i) * v.width;
    const int x = static_cast<int>(kp->x);
    const int y = static_cast<int>(kp->y);
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
constexpr int MAX_DISC = 716;                 // pixels of the radius-15 disc (709) rounded up

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

__global__ __launch_bounds__(256) void moments_kernel(BatchView v, DetectorParams P, int2* __restrict__ moments, int nframes,
                                                      int blocks_per_frame, int use_visit) {
    __shared__ uint32_t tile[4][MOM_ROWS * MOM_DW];
    __shared__ uint16_t disc[MAX_DISC];
    __shared__ int disc_count;
    int f, bx;
    if (!frame_block(nframes, blocks_per_frame, &f, &bx)) {
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = v.n_kp[f];
    const int radius = P.patch / 2;
    if (threadIdx.x == 0) {
        disc_count = 0;
    }
    __syncthreads();
    {
        // disc offsets (u,v) with u^2+v^2 <= r^2, any order (the integer moments are order-free)
        const int side = 2 * radius + 1;
        for (int t = threadIdx.x; t < side * side; t += 256) {
            const int vv = t / side - radius;
            const int uu = t - (vv + radius) * side - radius;
            if (uu * uu + vv * vv <= radius * radius) {
                const int slot = atomicAdd(&disc_count, 1);
                disc[slot] = static_cast<uint16_t>(((vv + 128) << 8) | (uu + 128));
            }
        }
    }
    __syncthreads();
    const int ndisc = disc_count;
    const uint8_t* img = v.blurred + static_cast<size_t>(f) * v.frame_stride;
    const uint8_t* t8 = reinterpret_cast<const uint8_t*>(tile[wave]);
    const slamx_keypoint* kps = v.kps + static_cast<size_t>(f) * v.maxK;
    const uint32_t* visit = v.visit + static_cast<size_t>(f) * v.maxK;
    const int stride = blocks_per_frame * 4;
    const int i0 = bx * 4 + wave;
    TilePrefetch<MOM_ROWS, MOM_DW, MOM_HALF> pf;
    for (int chunk = 0; i0 + chunk * stride < n; chunk += 64) {  // wave-uniform
        // lane j holds the slot and coordinates of this wave's (chunk + j)-th keypoint
        const int mine = i0 + (chunk + lane) * stride;
        int kx = 0, ky = 0, kslot = 0;
        if (mine < n) {
            kslot = use_visit ? static_cast<int>(visit[mine]) : mine;
            kx = static_cast<int>(kps[kslot].x);
            ky = static_cast<int>(kps[kslot].y);
        }
        const int cnt = min(64, (n - 1 - (i0 + chunk * stride)) / stride + 1);
        auto is_inside = [&](int x, int y) {
            return !(x - radius < 0 || x + radius >= v.cols || y - radius < 0 || y + radius >= v.rows);
        };
        int x = __shfl(kx, 0);
        int y = __shfl(ky, 0);
        if (is_inside(x, y)) {
            pf.issue(v, img, x, y, lane);
        }
        for (int k = 0; k < cnt; ++k) {
            const int i = __shfl(kslot, k);
            const bool inside = is_inside(x, y);  // wave-uniform
            if (inside) {
                pf.commit(tile[wave], lane);
            }
            const int cx = x - ((x - MOM_HALF) & ~3);
            // prefetch the next keypoint's tile while this one is reduced
            const int xn = __shfl(kx, (k + 1) & 63);
            const int yn = __shfl(ky, (k + 1) & 63);
            if (k + 1 < cnt && is_inside(xn, yn)) {
                pf.issue(v, img, xn, yn, lane);
            }
            int m01 = 0;
            int m10 = 0;
            if (inside) {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int t = lane; t < ndisc; t += 64) {
                    const int d = disc[t];
                    const int vv = (d >> 8) - 128;
                    const int uu = (d & 0xff) - 128;
                    const int I = t8[(vv + MOM_HALF) * (MOM_DW * 4) + cx + uu];
                    m01 += vv * I;
                    m10 += uu * I;
                }
                m01 = wave_sum(m01);
                m10 = wave_sum(m10);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (lane == 0) {
                moments[static_cast<size_t>(f) * v.maxK + i] = make_int2(m01, inside ? m10 : 0x7fffffff);
            }
            x = xn;
            y = yn;
        }
    }
}

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
            r = make_float2(glibc_cosf(a), glibc_sinf(a));
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
    const int wave = threadIdx.x >> 6;
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
    for (int chunk = 0; i0 + chunk * stride < n; chunk += 64) {  // wave-uniform
        const int mine = i0 + (chunk + lane) * stride;
        int kx = 0, ky = 0, kslot = 0;
        float kc = 0.0f, ks = 0.0f;
        if (mine < n) {
            kslot = use_visit ? static_cast<int>(visit[mine]) : mine;
            kx = static_cast<int>(kps[kslot].x);
            ky = static_cast<int>(kps[kslot].y);
            const float2 r = csf[kslot];
            kc = r.x;
            ks = r.y;
        }
        const int cnt = min(64, (n - 1 - (i0 + chunk * stride)) / stride + 1);
        int x = __shfl(kx, 0);
        int y = __shfl(ky, 0);
        float ca = __shfl(kc, 0);
        float sa = __shfl(ks, 0);
        if (ca == ca) {
            pf.issue(v, img, x, y, lane);
        }
        for (int k = 0; k < cnt; ++k) {
            const int i = __shfl(kslot, k);
            uint8_t* desc = v.desc + (static_cast<size_t>(f) * v.maxK + i) * v.width;
            const bool inside = ca == ca;  // NaN cos marks "patch leaves the image" (wave-uniform)
            if (lane < words) {
                dbuf[wave][lane] = 0ull;
            }
            if (inside) {
                pf.commit(tile[wave], lane);
            }
            const int xn = __shfl(kx, (k + 1) & 63);
            const int yn = __shfl(ky, (k + 1) & 63);
            const float can = __shfl(kc, (k + 1) & 63);
            const float san = __shfl(ks, (k + 1) & 63);
            if (k + 1 < cnt && can == can) {
                pf.issue(v, img, xn, yn, lane);
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
                    bool outbit = false;
                    if (lane < c2) {
                        outbit = ((bm >> select64d(vm, lane)) & 1ull) != 0;
                    }
                    const unsigned long long packed = __ballot(outbit);
                    if (lane == 0 && c2 > 0) {
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
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint8_t* d8 = reinterpret_cast<const uint8_t*>(dbuf[wave]);
            for (int b = lane; b < v.width; b += 64) {
                desc[b] = d8[b];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            x = xn;
            y = yn;
            ca = can;
            sa = san;
        }
    }
}

}  // namespace

void launch_describe(const BatchView& v, const DetectorParams& p, const int8_t* pattern_dev, int nframes, bool use_visit,
                     hipStream_t stream, hipEvent_t ev_after_moments, hipEvent_t ev_after_trig) {
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
