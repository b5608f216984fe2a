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

}  // namespace

void launch_describe(const BatchView& v, const DetectorParams& p, const int8_t* pattern_dev, int nframes,
                     hipStream_t stream) {
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
    hipLaunchKernelGGL(describe_kernel, grid, dim3(256), 0, stream, v, p, pattern_dev, exact_int);
}

}  // namespace slamx
