// match.hip -- brute-force Hamming 1-NN with second-best tracking, jump-radius penalty and
// Lowe ratio test.
//
// Reference (paths relative to the reference root):
//   calculateHammingDistance   include/slam/common/common.hpp:40-50   (byte LUT popcount)
//   updateBestMatches          src/frontend/feature_matcher.cpp:132-141 (strict '<': lowest j wins ties)
//   findBestMatchesHamming     src/frontend/feature_matcher.cpp:143-189
//   MAX_JUMP_RADIUS = 500      include/slam/frontend/feature_matcher.hpp:12
//
// MI355X design: VALU-integer bound (16 int ops per 256-bit pair: 8 xor + 8 v_bcnt with fused
// accumulate), not HBM or MFMA bound.  One lane owns one query descriptor in registers; train
// descriptors are staged 256 at a time through LDS and read back as wave-uniform broadcasts, so
// every lane walks the train set in ascending j -- the reference's scan order, which is what
// makes "lowest j wins ties" fall out of a strict '<' with no tie-break logic.  The penalty
// uses the same float operations as the reference (correctly rounded sqrt and divide).
#include "slamx_internal.h"

#pragma clang fp contract(off)

namespace slamx {

namespace {

constexpr int TILE = 256;
constexpr float MAX_JUMP_RADIUS = 500.0f;  // uint32 500 converted to float in both comparisons

template <int WORDS>
__global__ __launch_bounds__(256) void match_kernel(MatchView mv, int first_query_frame, int first_train_frame) {
    __shared__ uint32_t tdesc[TILE * WORDS];
    __shared__ float2 txy[TILE];
    const int qf = first_query_frame + blockIdx.y;
    const int tf = qf == 0 ? first_train_frame : qf - 1;
    const int n1 = mv.n_kp[qf];
    const int n2 = mv.n_kp[tf];
    const int i = static_cast<int>(blockIdx.x) * TILE + static_cast<int>(threadIdx.x);
    if (static_cast<int>(blockIdx.x) * TILE >= n1) {
        return;  // whole workgroup idle (uniform)
    }
    const bool use_kp = mv.kps != nullptr;
    const uint32_t* qd = reinterpret_cast<const uint32_t*>(mv.desc + static_cast<size_t>(qf) * mv.maxK * mv.width);
    const uint32_t* td = reinterpret_cast<const uint32_t*>(mv.desc + static_cast<size_t>(tf) * mv.maxK * mv.width);
    uint32_t q[WORDS];
    float qx = 0.0f, qy = 0.0f;
    if (i < n1) {
#pragma unroll
        for (int w = 0; w < WORDS; ++w) {
            q[w] = qd[static_cast<size_t>(i) * WORDS + w];
        }
        if (use_kp) {
            const slamx_keypoint k = mv.kps[static_cast<size_t>(qf) * mv.maxK + i];
            qx = k.x;
            qy = k.y;
        }
    } else {
#pragma unroll
        for (int w = 0; w < WORDS; ++w) {
            q[w] = 0;
        }
    }
    int best = 0x7fffffff;
    int second = 0x7fffffff;
    int best_idx = -1;
    for (int t0 = 0; t0 < n2; t0 += TILE) {
        const int tn = min(TILE, n2 - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < tn * WORDS; e += 256) {
            tdesc[e] = td[static_cast<size_t>(t0) * WORDS + e];
        }
        if (use_kp && static_cast<int>(threadIdx.x) < tn) {
            const slamx_keypoint k = mv.kps[static_cast<size_t>(tf) * mv.maxK + t0 + threadIdx.x];
            txy[threadIdx.x] = make_float2(k.x, k.y);
        }
        __syncthreads();
        for (int j = 0; j < tn; ++j) {
            int dist = 0;
#pragma unroll
            for (int w = 0; w < WORDS; ++w) {
                dist += __popc(q[w] ^ tdesc[j * WORDS + w]);
            }
            if (use_kp) {
                const float2 t = txy[j];
                const float dx = __fsub_rn(qx, t.x);
                const float dy = __fsub_rn(qy, t.y);
                const float d = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
                if (d > MAX_JUMP_RADIUS) {
                    const float penalty = __fadd_rn(1.0f, __fdiv_rn(d, MAX_JUMP_RADIUS));
                    dist = static_cast<int>(__fmul_rn(static_cast<float>(dist), penalty));
                }
            }
            const bool lt = dist < best;
            second = lt ? best : min(second, dist);
            best_idx = lt ? (t0 + j) : best_idx;
            best = lt ? dist : best;
        }
    }
    if (i < n1) {
        mv.best[static_cast<size_t>(qf) * mv.maxK + i] = make_int2(best, best_idx);
        mv.second[static_cast<size_t>(qf) * mv.maxK + i] = second;
    }
}

// Any descriptor width (bytes): byte-wise popcount straight from global memory.  Correctness
// path for widths that are not 8/16/32/64 bytes.
__global__ __launch_bounds__(256) void match_generic_kernel(MatchView mv, int first_query_frame, int first_train_frame) {
    const int qf = first_query_frame + blockIdx.y;
    const int tf = qf == 0 ? first_train_frame : qf - 1;
    const int n1 = mv.n_kp[qf];
    const int n2 = mv.n_kp[tf];
    const bool use_kp = mv.kps != nullptr;
    const uint8_t* qd = mv.desc + static_cast<size_t>(qf) * mv.maxK * mv.width;
    const uint8_t* td = mv.desc + static_cast<size_t>(tf) * mv.maxK * mv.width;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += gridDim.x * blockDim.x) {
        float qx = 0.0f, qy = 0.0f;
        if (use_kp) {
            const slamx_keypoint k = mv.kps[static_cast<size_t>(qf) * mv.maxK + i];
            qx = k.x;
            qy = k.y;
        }
        int best = 0x7fffffff;
        int second = 0x7fffffff;
        int best_idx = -1;
        for (int j = 0; j < n2; ++j) {
            int dist = 0;
            for (int b = 0; b < mv.width; ++b) {
                dist += __popc(static_cast<uint32_t>(qd[static_cast<size_t>(i) * mv.width + b] ^
                                                     td[static_cast<size_t>(j) * mv.width + b]));
            }
            if (use_kp) {
                const slamx_keypoint k = mv.kps[static_cast<size_t>(tf) * mv.maxK + j];
                const float dx = __fsub_rn(qx, k.x);
                const float dy = __fsub_rn(qy, k.y);
                const float d = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
                if (d > MAX_JUMP_RADIUS) {
                    const float penalty = __fadd_rn(1.0f, __fdiv_rn(d, MAX_JUMP_RADIUS));
                    dist = static_cast<int>(__fmul_rn(static_cast<float>(dist), penalty));
                }
            }
            const bool lt = dist < best;
            second = lt ? best : min(second, dist);
            best_idx = lt ? j : best_idx;
            best = lt ? dist : best;
        }
        mv.best[static_cast<size_t>(qf) * mv.maxK + i] = make_int2(best, best_idx);
        mv.second[static_cast<size_t>(qf) * mv.maxK + i] = second;
    }
}

// Ratio test (feature_matcher.cpp:175-187) + emission in query order.  One workgroup per pair.
__global__ __launch_bounds__(1024) void emit_kernel(MatchView mv, int first_query_frame, int first_train_frame) {
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t carry;
    const int qf = first_query_frame + blockIdx.x;
    const int n1 = mv.n_kp[qf];
    const int n2 = mv.n_kp[qf == 0 ? first_train_frame : qf - 1];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        carry = 0;
    }
    __syncthreads();
    slamx_match* out = mv.matches + static_cast<size_t>(qf) * mv.maxK;
    if (n2 > 0) {
        for (int base = 0; base < n1; base += 1024) {
            const int i = base + threadIdx.x;
            bool good = false;
            int2 b = make_int2(0, -1);
            if (i < n1) {
                b = mv.best[static_cast<size_t>(qf) * mv.maxK + i];
                const int second = mv.second[static_cast<size_t>(qf) * mv.maxK + i];
                good = true;
                if (mv.use_ratio != 0) {
                    if (static_cast<float>(b.x) >= __fmul_rn(mv.ratio, static_cast<float>(second))) {
                        good = false;
                    }
                }
                good = good && b.y != -1;
            }
            const unsigned long long bm = __ballot(good);
            const uint32_t c = static_cast<uint32_t>(__popcll(bm));
            if (lane == 0) {
                wave_sums[wave] = c;
            }
            __syncthreads();
            uint32_t wbase = carry;
            for (int w = 0; w < wave; ++w) {
                wbase += wave_sums[w];
            }
            if (good) {
                const uint32_t pos = wbase + static_cast<uint32_t>(__popcll(bm & ((1ull << lane) - 1ull)));
                slamx_match m;
                m.queryIdx = i;
                m.trainIdx = b.y;
                m.distance = static_cast<float>(b.x);
                out[pos] = m;
            }
            __syncthreads();
            if (threadIdx.x == 1023) {
                carry = wbase + c;
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        mv.n_matches[qf] = static_cast<int>(carry);
    }
}

}  // namespace

void launch_match_consecutive(const MatchView& mv, int first_query_frame, int npairs, int first_train_frame,
                              hipStream_t stream, hipEvent_t ev_mid) {
    if (npairs <= 0) {
        return;
    }
    dim3 grid((mv.maxK + TILE - 1) / TILE, npairs);
    switch (mv.width) {
        case 8:
            hipLaunchKernelGGL(match_kernel<2>, grid, dim3(256), 0, stream, mv, first_query_frame, first_train_frame);
            break;
        case 16:
            hipLaunchKernelGGL(match_kernel<4>, grid, dim3(256), 0, stream, mv, first_query_frame, first_train_frame);
            break;
        case 32:
            hipLaunchKernelGGL(match_kernel<8>, grid, dim3(256), 0, stream, mv, first_query_frame, first_train_frame);
            break;
        case 64:
            hipLaunchKernelGGL(match_kernel<16>, grid, dim3(256), 0, stream, mv, first_query_frame, first_train_frame);
            break;
        default: {
            dim3 g(grid_x(mv.maxK, 256, npairs), npairs);
            hipLaunchKernelGGL(match_generic_kernel, g, dim3(256), 0, stream, mv, first_query_frame, first_train_frame);
        }
    }
    if (ev_mid != nullptr) {
        (void)hipEventRecord(ev_mid, stream);
    }
    hipLaunchKernelGGL(emit_kernel, dim3(npairs), dim3(1024), 0, stream, mv, first_query_frame, first_train_frame);
}

}  // namespace slamx
