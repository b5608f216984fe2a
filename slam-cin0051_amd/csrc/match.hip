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

constexpr int TILE = 256;        // queries per workgroup and query slot (= threads)
constexpr int TRAIN_TILE = 256;  // train descriptors staged in LDS at a time; the queues are drained at its end
constexpr int EVQ_DEPTH = 32;  // deferred-update queue entries per lane

// WORDS = 32-bit words compared per descriptor (the "live" prefix: words that can be non-zero in
// either set; the all-zero tail contributes popcount(0^0) = 0 and is skipped -- exact).
// STRIDE words separate consecutive descriptors in memory.  QPL = queries per lane (2 or 4): a
// train descriptor read from LDS is compared with QPL queries, which spreads the loop, filter and
// queue overhead (more than half of the instructions at QPL = 2) over more pairs.
//
// A pair can only change (best, second) if its distance is < second, and the jump penalty only
// ever increases the distance, so the Hamming distance alone filters: raw >= second => the pair
// is dropped.  About 2 % of the pairs pass (`second` tightens like a record sequence), i.e. almost
// every 64-lane wave has SOME lane that passes at every step, so handling a pass on the spot would
// run the penalty arithmetic at 1/64 lane utilisation.  Instead every lane appends the train index
// of a step in which one of its queries passed to a private LDS queue, and the wave drains all
// queues together when one fills up or the train tile ends: lane utilisation of the expensive part
// rises from ~2 % to ~40 %.  The filter then uses a slightly stale `second` (an upper bound of the
// current one: a superset passes) and the drain recomputes the distances and re-applies the exact
// update rule in ascending train order per query, so the result is the sequential scan's.
template <int WORDS, int QPL>
__global__ __launch_bounds__(256) void match_kernel(MatchView mv, int first_query_frame, int first_train_frame,
                                                    int stride, int npairs, int qblocks) {
    static_assert(QPL == 2 || QPL == 4, "queries are filtered in packed pairs");
    __shared__ uint32_t tdesc[TRAIN_TILE * WORDS];
    __shared__ float2 txy[TRAIN_TILE];
    __shared__ uint16_t evq[4][EVQ_DEPTH * 64];  // per wave: entry k of lane l at [k * 64 + l]
    // 1-D grid, pair index fastest: workgroups are dealt to the 8 XCDs round-robin in launch order,
    // so a grid with the query-block index fastest (8 blocks per pair, the last ones empty) would
    // leave whole XCDs idle; this way every XCD gets the same mix and the blocks of one pair share
    // an XCD (and its L2) when npairs is a multiple of 8.
    const int pair = static_cast<int>(blockIdx.x) % npairs;
    const int rest = static_cast<int>(blockIdx.x) / npairs;
    const int qblock = rest % qblocks;
    const int split = rest / qblocks;
    const int qf = first_query_frame + pair;
    const int tf = qf == 0 ? first_train_frame : qf - 1;
    const int n1 = mv.n_kp[qf];
    const int n2 = mv.n_kp[tf];
    const int block_first = qblock * TILE * QPL;
    if (block_first >= n1) {
        return;  // whole workgroup idle (uniform)
    }
    const bool use_kp = mv.kps != nullptr;
    const uint32_t* qd = reinterpret_cast<const uint32_t*>(mv.desc + static_cast<size_t>(qf) * mv.maxK * mv.width);
    const uint32_t* td = reinterpret_cast<const uint32_t*>(mv.desc + static_cast<size_t>(tf) * mv.maxK * mv.width);
    uint32_t q[QPL][WORDS];
    float qx[QPL], qy[QPL];
    int best[QPL], second[QPL], best_idx[QPL];
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
        const int i = block_first + u * TILE + static_cast<int>(threadIdx.x);
        qx[u] = 0.0f;
        qy[u] = 0.0f;
        best[u] = 0x7fffffff;
        second[u] = 0x7fffffff;
        best_idx[u] = -1;
#pragma unroll
        for (int w = 0; w < WORDS; ++w) {
            // WORDS can exceed the descriptor's own word count (5 -> 6, 7 -> 8, 9..15 -> 16): the
            // words past `stride` belong to the NEXT descriptor and are read as zero here
            q[u][w] = (i < n1 && w < stride) ? qd[static_cast<size_t>(i) * stride + w] : 0u;
        }
        if (use_kp && i < n1) {
            const slamx_keypoint k = mv.kps[static_cast<size_t>(qf) * mv.maxK + i];
            qx[u] = k.x;
            qy[u] = k.y;
        }
    }
    constexpr int JB = 8;
    const int lane = threadIdx.x & 63;
    // the last workgroup of a frame is mostly empty (2087 queries = 4 x 512 + 39): its idle waves skip the scan
    const bool wave_has_queries = block_first + static_cast<int>(threadIdx.x & ~63u) < n1;
    // LDS address space spelled out: the queue cursor then is a 32-bit offset, not a 64-bit flat pointer
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    lds_u16* const myq = (lds_u16*)(evq[__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6))] + lane);
    lds_u16* qend = myq;  // one past this lane's last queue entry (entries are 64 elements apart)
    // one (query u, train t) pair through the reference's update (feature_matcher.cpp:132-141,
    // penalty :161-170); t is tile-relative
    auto apply = [&](int dist, int t, int u, int t0) {
        if (dist < second[u]) {
            if (use_kp) {
                const float2 p = txy[t];
                // Cheap exact rejection: evaluate the penalised value with the 1-ulp hardware sqrt
                // (relative error < 1e-6 against the reference expression); if even that value
                // shrunk by 1e-5 reaches `second`, the pair cannot change (best, second).
                const float dx = __fsub_rn(qx[u], p.x);
                const float dy = __fsub_rn(qy[u], p.y);
                const float da = __builtin_amdgcn_sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
                if (da > 500.01f) {
                    const float lower = static_cast<float>(dist) * (1.0f + da * 0.002f) * 0.99999f;
                    if (lower >= static_cast<float>(second[u])) {
                        return;
                    }
                }
                dist = penalised_distance(dist, qx[u], qy[u], p.x, p.y);
            }
            const bool lt = dist < best[u];
            second[u] = lt ? best[u] : min(second[u], dist);
            best_idx[u] = lt ? (t0 + t) : best_idx[u];
            best[u] = lt ? dist : best[u];
        }
    };
    // the (stale) filter bounds: `second` as of the last drain
    int bound[QPL];
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
        bound[u] = 0x7fffffff;
    }
    // every lane replays its own queue in push order (= ascending train index); an entry is the
    // tile-relative train index, the distances to the lane's queries are recomputed from the LDS
    // tile and apply() re-tests each against the now-current `second`
    auto drain = [&](int t0) {
        for (const lds_u16* e = myq; __ballot(e < qend) != 0; e += 64) {
            if (e < qend) {
                const int t = *e;
                uint32_t tw[WORDS];
#pragma unroll
                for (int w = 0; w < WORDS; ++w) {
                    tw[w] = tdesc[t * WORDS + w];
                }
#pragma unroll
                for (int u = 0; u < QPL; ++u) {
                    int dd = 0;
#pragma unroll
                    for (int w = 0; w < WORDS; ++w) {
                        dd += __popc(q[u][w] ^ tw[w]);
                    }
                    apply(dd, t, u, t0);
                }
            }
        }
        qend = myq;
#pragma unroll
        for (int u = 0; u < QPL; ++u) {
            bound[u] = second[u];
        }
    };
    // some d[u] < bound[u]: one VOPC compare per query, combined on the scalar side (a packed 16-bit
    // subtract of two queries at once costs two VOP3 instructions, i.e. more issue cycles)
    auto passes = [&](const int* dq) {
        bool any = false;
#pragma unroll
        for (int u = 0; u < QPL; ++u) {
            any = any | (dq[u] < bound[u]);
        }
        return any;
    };
    auto step = [&](const uint32_t* tw, int j) {  // one train descriptor against the lane's queries
        int dq[QPL];
#pragma unroll
        for (int u = 0; u < QPL; ++u) {
            dq[u] = 0;
#pragma unroll
            for (int w = 0; w < WORDS; ++w) {
                dq[u] += __popc(q[u][w] ^ tw[w]);
            }
        }
        if (passes(dq)) {  // against a stale upper bound of the true `second`
            *qend = static_cast<uint16_t>(j);
            qend += 64;
        }
    };
    // this workgroup scans train chunk `split` of mv.nsplit (merged in order by emit_kernel)
    const int chunk = (n2 + mv.nsplit - 1) / mv.nsplit;
    const int t_begin = split * chunk;
    const int t_end = min(n2, t_begin + chunk);
    for (int t0 = t_begin; t0 < t_end; t0 += TRAIN_TILE) {
        const int tn = min(TRAIN_TILE, t_end - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < tn * WORDS; e += 256) {
            const int j = e / WORDS;
            const int w = e - j * WORDS;
            tdesc[e] = w < stride ? td[static_cast<size_t>(t0 + j) * stride + w] : 0u;
        }
        if (use_kp) {
            for (int e = threadIdx.x; e < tn; e += 256) {
                const slamx_keypoint k = mv.kps[static_cast<size_t>(tf) * mv.maxK + t0 + e];
                txy[e] = make_float2(k.x, k.y);
            }
        }
        __syncthreads();
        if (!wave_has_queries) {
            continue;  // wave-uniform: this wave only helps staging the train tiles
        }
        int j = 0;
        for (; j + JB <= tn; j += JB) {
            // all LDS reads of the group first, then the popcounts
            uint32_t t[JB][WORDS];
#pragma unroll
            for (int k = 0; k < JB; ++k) {
#pragma unroll
                for (int w = 0; w < WORDS; ++w) {
                    t[k][w] = tdesc[(j + k) * WORDS + w];
                }
            }
#pragma unroll
            for (int k = 0; k < JB; ++k) {
                step(t[k], j + k);
            }
            if (__ballot(qend > myq + (EVQ_DEPTH - JB) * 64) != 0) {  // wave-uniform
                drain(t0);
            }
        }
        for (; j < tn; ++j) {
            uint32_t t[WORDS];
#pragma unroll
            for (int w = 0; w < WORDS; ++w) {
                t[w] = tdesc[j * WORDS + w];
            }
            step(t, j);
            if (__ballot(qend > myq + (EVQ_DEPTH - JB) * 64) != 0) {
                drain(t0);
            }
        }
        drain(t0);  // the queue holds tile-relative indices and txy[] is about to be replaced
    }
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
        const int i = block_first + u * TILE + static_cast<int>(threadIdx.x);
        if (i < n1) {
            const size_t slot = (static_cast<size_t>(qf) * mv.nsplit + split) * mv.maxK + i;
            mv.best[slot] = make_int2(best[u], best_idx[u]);
            mv.second[slot] = second[u];
        }
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
                dist = penalised_distance(dist, qx, qy, k.x, k.y);
            }
            const bool lt = dist < best;
            second = lt ? best : min(second, dist);
            best_idx = lt ? j : best_idx;
            best = lt ? dist : best;
        }
        for (int c = 0; c < mv.nsplit; ++c) {  // the generic path scans everything as chunk 0
            const size_t slot = (static_cast<size_t>(qf) * mv.nsplit + c) * mv.maxK + i;
            mv.best[slot] = c == 0 ? make_int2(best, best_idx) : make_int2(0x7fffffff, -1);
            mv.second[slot] = c == 0 ? second : 0x7fffffff;
        }
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
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
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
                // merge the per-chunk (best, idx, second) in ascending train order through the
                // same update rule: the two smallest of the union, lowest index on ties
                int best = 0x7fffffff;
                int second = 0x7fffffff;
                int best_idx = -1;
                for (int c = 0; c < mv.nsplit; ++c) {
                    const size_t slot = (static_cast<size_t>(qf) * mv.nsplit + c) * mv.maxK + i;
                    const int2 pb = mv.best[slot];
                    const int ps = mv.second[slot];
                    if (pb.x < best) {
                        second = best;
                        best = pb.x;
                        best_idx = pb.y;
                    } else if (pb.x < second) {
                        second = pb.x;
                    }
                    if (ps < second) {
                        second = ps;
                    }
                }
                b = make_int2(best, best_idx);
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

template <int WORDS>
void launch_words(const MatchView& mv, int first_query_frame, int npairs, int first_train_frame, hipStream_t stream) {
    constexpr int QPL = 2;  // 4 per lane measured slower (0.55 vs 0.40 ms): longer drains, fewer waves
    const int qblocks = (mv.maxK + TILE * QPL - 1) / (TILE * QPL);
    dim3 grid(static_cast<unsigned>(qblocks) * npairs * mv.nsplit);
    hipLaunchKernelGGL((match_kernel<WORDS, QPL>), grid, dim3(256), 0, stream, mv, first_query_frame, first_train_frame,
                       mv.width / 4, npairs, qblocks);
}

}  // namespace

void launch_match_consecutive(const MatchView& mv, int first_query_frame, int npairs, int first_train_frame,
                              hipStream_t stream, hipEvent_t ev_mid) {
    if (npairs <= 0) {
        return;
    }
    const int stride = mv.width / 4;
    if (mv.width % 4 == 0 && stride <= 16) {
        int live = mv.live_words > 0 && mv.live_words < stride ? mv.live_words : stride;
        switch (live) {
            case 1: launch_words<1>(mv, first_query_frame, npairs, first_train_frame, stream); break;
            case 2: launch_words<2>(mv, first_query_frame, npairs, first_train_frame, stream); break;
            case 3: launch_words<3>(mv, first_query_frame, npairs, first_train_frame, stream); break;
            case 4: launch_words<4>(mv, first_query_frame, npairs, first_train_frame, stream); break;
            case 5:
            case 6: launch_words<6>(mv, first_query_frame, npairs, first_train_frame, stream); break;
            case 7:
            case 8: launch_words<8>(mv, first_query_frame, npairs, first_train_frame, stream); break;
            default: launch_words<16>(mv, first_query_frame, npairs, first_train_frame, stream); break;
        }
    } else {
        dim3 g(grid_x(mv.maxK, 256, npairs), npairs);
        hipLaunchKernelGGL(match_generic_kernel, g, dim3(256), 0, stream, mv, first_query_frame, first_train_frame);
    }
    if (ev_mid != nullptr) {
        (void)hipEventRecord(ev_mid, stream);
    }
    hipLaunchKernelGGL(emit_kernel, dim3(npairs), dim3(1024), 0, stream, mv, first_query_frame, first_train_frame);
}

}  // namespace slamx
