// nms.hip -- the greedy radius suppression of applyNonMaxSuppression, bit-exact and parallel.
//
// Reference: src/frontend/feature_detector.cpp:163-187.  Walking the std::sort-ed candidates, the
// reference keeps a candidate unless an EARLIER KEPT one lies closer than SuppressionWindowSize
// (sqrtf(dx^2+dy^2) < float(window), :177-181): the lexicographically-first maximal independent
// set of the "closer than window" graph in sorted order.
//
// Parallel form (same set, same order): a candidate is KEPT once every earlier-ranked neighbour
// is SUPPRESSED and SUPPRESSED once one of them is KEPT.  States only move UNDECIDED -> final, so
// racy reads are harmless (a stale read only delays a decision).
//   nms_build_kernel : one lane per candidate (in RASTER order, so a wave walks neighbouring
//                      pixels and shares bitmap rows in L1).  Neighbours are enumerated from the
//                      FAST corner bitmap rows y-(w-1)..y+(w-1) and its prefix sums (fast.hip)
//                      -- no cell grid -- and the earlier-ranked ones inside the radius are
//                      cached (up to NBR_CAP raster indices); no earlier neighbour => KEPT.
//   nms_round_kernel : two fully parallel rounds from the cached lists (stored transposed, so
//                      the reads are coalesced).
//   nms_finish_kernel: one workgroup per frame copies the states into LDS and loops rounds to the
//                      fix-point from the cached lists (a list that overflowed falls back to
//                      re-enumeration; dependency chains can be long on straight edges), then
//                      emits the kept keypoints in rank order (feature_detector.cpp:170, :187).
#include "slamx_internal.h"

#pragma clang fp contract(off)

namespace slamx {

namespace {

__device__ __forceinline__ unsigned long long low_mask(int nbits) {  // bits [0, nbits)
    return nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
}

struct Cand {
    int x, y;
    uint32_t rank;
};

// Calls visit(nidx) for every candidate that is ranked before `me` and lies closer than `window`
// (the reference's float test).  visit returns false to stop early.
template <typename Visit>
__device__ __forceinline__ void for_each_earlier_neighbour(const BatchView& v, int f, const Cand& me, int window,
                                                            const uint32_t* __restrict__ rank_of, Visit&& visit) {
    const int n = v.rows * v.nw;
    const unsigned long long* __restrict__ masks = v.masks + static_cast<size_t>(f) * n;
    const uint32_t* __restrict__ off = v.wordoff + static_cast<size_t>(f) * n;
    const float fx = static_cast<float>(me.x);
    const float fy = static_cast<float>(me.y);
    const float fw = static_cast<float>(window);
    const int reach = window - 1;  // |d| >= window can never be < window
    const int ylo = max(0, me.y - reach);
    const int yhi = min(v.rows - 1, me.y + reach);
    const int xlo = max(0, me.x - reach);
    const int xhi = min(v.cols - 1, me.x + reach);
    const int w0 = xlo >> 6;
    const int w1 = xhi >> 6;
    // Fast path (window <= 32, so the probed column range spans <= 63 pixels and <= 2 bitmap
    // words): per row the bits of columns [xlo, xhi] are funnel-shifted into one 64-bit window;
    // candidates of a row are consecutive in raster order, so the t-th set bit of the window is
    // candidate base + t with base = wordoff + popcount(bits left of xlo): one prefix lookup per
    // non-empty row instead of one per neighbour.  Coordinates are integers, so the reference's
    // sqrtf(dx^2+dy^2) < float(w) is exactly dx^2+dy^2 < w^2 (both sides exact in float for
    // w < 2896, sqrt correctly rounded and monotone).  Rows are probed in chunks of 8 with all
    // bitmap loads of a chunk issued before any is used.
    constexpr int CH = 8;
    if (window <= 32) {
        const bool two_words = w1 != w0;
        const int sh = xlo - w0 * 64;          // bit of column xlo inside word w0
        const int width = xhi - xlo + 1;       // <= 63
        const unsigned long long wmask = low_mask(width);
        const int w2 = window * window;
        for (int yc = ylo; yc <= yhi; yc += CH) {
            unsigned long long ma[CH], mb[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int yy = min(yc + k, yhi);
                ma[k] = masks[static_cast<size_t>(yy) * v.nw + w0];
                mb[k] = two_words ? masks[static_cast<size_t>(yy) * v.nw + w1] : 0ull;
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int yy = yc + k;
                if (yy > yhi) {
                    break;
                }
                unsigned long long win = ma[k] >> sh;
                if (sh != 0) {
                    win |= mb[k] << (64 - sh);
                }
                win &= wmask;
                if (win == 0) {
                    continue;
                }
                const int dyi = me.y - yy;
                const int rem = w2 - dyi * dyi;  // need dx^2 < rem
                uint32_t nidx = off[static_cast<size_t>(yy) * v.nw + w0] +
                                static_cast<uint32_t>(__popcll(ma[k] & low_mask(sh)));
                while (win != 0) {
                    const int b = __ffsll(static_cast<long long>(win)) - 1;
                    win &= win - 1;
                    const uint32_t cur = nidx++;
                    const int dxi = me.x - (xlo + b);
                    if (dxi * dxi < rem && rank_of[cur] < me.rank) {
                        if (!visit(cur)) {
                            return;
                        }
                    }
                }
            }
        }
        return;
    }
    // windows wider than 64 pixels: plain walk over every word of every row
    for (int yy = ylo; yy <= yhi; ++yy) {
        const float dy = fy - static_cast<float>(yy);
        for (int w = w0; w <= w1; ++w) {
            const unsigned long long full = masks[static_cast<size_t>(yy) * v.nw + w];
            const int bl = max(xlo - w * 64, 0);
            const int bh = min(xhi - w * 64, 63);
            unsigned long long m = full & (low_mask(bh + 1) & ~low_mask(bl));
            if (m == 0) {
                continue;
            }
            const uint32_t base = off[static_cast<size_t>(yy) * v.nw + w];
            while (m != 0) {
                const int b = __ffsll(static_cast<long long>(m)) - 1;
                m &= m - 1;
                const uint32_t nidx = base + static_cast<uint32_t>(__popcll(full & low_mask(b)));
                if (rank_of[nidx] >= me.rank) {
                    continue;  // later in sorted order (or myself)
                }
                const float dx = fx - static_cast<float>(w * 64 + b);
                const float dist = sqrtf_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
                if (dist < fw) {
                    if (!visit(nidx)) {
                        return;
                    }
                }
            }
        }
    }
}

// One lane per candidate, candidates in RASTER order: neighbouring lanes probe the same bitmap
// rows, so the loads of a wave hit a handful of cache lines.  The cached list is stored
// TRANSPOSED (entry k of candidate i at nbr[k * maxM + i]) so that later rounds read it coalesced.
__global__ __launch_bounds__(256) void nms_build_kernel(BatchView v, int window) {
    const int f = blockIdx.y;
    const int n = v.n_cand[f];
    const uint32_t* __restrict__ xy = v.cand_xy + static_cast<size_t>(f) * v.maxM;
    const uint32_t* __restrict__ rank_of = v.rank_of + static_cast<size_t>(f) * v.maxM;
    uint8_t* state = v.state + static_cast<size_t>(f) * v.maxM;
    uint8_t* cnt_out = v.nbr_cnt + static_cast<size_t>(f) * v.maxM;
    uint32_t* nbr = v.nbr + static_cast<size_t>(f) * v.maxM * NBR_CAP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t pxy = xy[i];
        Cand me;
        me.x = static_cast<int>(pxy & 0xffffu);
        me.y = static_cast<int>(pxy >> 16);
        me.rank = rank_of[i];
        int cnt = 0;
        for_each_earlier_neighbour(v, f, me, window, rank_of, [&](uint32_t nidx) {
            if (cnt < NBR_CAP) {
                nbr[static_cast<size_t>(cnt) * v.maxM + i] = nidx;
            }
            ++cnt;
            return true;
        });
        cnt_out[i] = cnt > NBR_CAP ? NBR_OVERFLOW : static_cast<uint8_t>(cnt);
        state[i] = cnt == 0 ? NMS_KEPT : NMS_UNDECIDED;
    }
}

// new state of candidate i (raster index) from the current states of its earlier neighbours
__device__ __forceinline__ uint8_t nms_decide(const BatchView& v, int f, int i, int window, const uint32_t* __restrict__ xy,
                                              const uint32_t* __restrict__ rank_of, const volatile uint8_t* state,
                                              const uint8_t* __restrict__ nbr_cnt, const uint32_t* __restrict__ nbr) {
    const int cnt = nbr_cnt[i];
    bool all_decided = true;
    if (cnt != NBR_OVERFLOW) {
        // four list entries, then their four states, as independent loads per step
        bool any_kept = false;
        for (int k0 = 0; k0 < cnt; k0 += 4) {
            uint32_t ids[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ids[k] = k0 + k < cnt ? nbr[static_cast<size_t>(k0 + k) * v.maxM + i] : static_cast<uint32_t>(i);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint8_t s = state[ids[k]];
                any_kept = any_kept || (k0 + k < cnt && s == NMS_KEPT);
                all_decided = all_decided && (k0 + k >= cnt || s != NMS_UNDECIDED);
            }
            if (any_kept) {
                break;
            }
        }
        return any_kept ? NMS_SUPPRESSED : (all_decided ? NMS_KEPT : NMS_UNDECIDED);
    }
    const uint32_t pxy = xy[i];
    Cand me;
    me.x = static_cast<int>(pxy & 0xffffu);
    me.y = static_cast<int>(pxy >> 16);
    me.rank = rank_of[i];
    bool suppressed = false;
    for_each_earlier_neighbour(v, f, me, window, rank_of, [&](uint32_t nidx) {
        const uint8_t s = state[nidx];
        if (s == NMS_KEPT) {
            suppressed = true;
            return false;
        }
        all_decided = all_decided && s != NMS_UNDECIDED;
        return true;
    });
    if (suppressed) {
        return NMS_SUPPRESSED;
    }
    return all_decided ? NMS_KEPT : NMS_UNDECIDED;
}

// One fully parallel round on the global states (most candidates are decided in the first two
// rounds: "has a KEPT earlier neighbour"); the per-frame finisher below runs the long tail.
__global__ __launch_bounds__(256) void nms_round_kernel(BatchView v, int window) {
    const int f = blockIdx.y;
    const int n = v.n_cand[f];
    uint8_t* state = v.state + static_cast<size_t>(f) * v.maxM;
    const uint32_t* xy = v.cand_xy + static_cast<size_t>(f) * v.maxM;
    const uint32_t* rank_of = v.rank_of + static_cast<size_t>(f) * v.maxM;
    const uint8_t* nbr_cnt = v.nbr_cnt + static_cast<size_t>(f) * v.maxM;
    const uint32_t* nbr = v.nbr + static_cast<size_t>(f) * v.maxM * NBR_CAP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (state[i] != NMS_UNDECIDED) {
            continue;
        }
        const uint8_t s = nms_decide(v, f, i, window, xy, rank_of, state, nbr_cnt, nbr);
        if (s != NMS_UNDECIDED) {
            state[i] = s;
        }
    }
}

extern __shared__ uint8_t lds_state[];

__global__ __launch_bounds__(1024) void nms_finish_kernel(BatchView v, int window, int lds_cap) {
    __shared__ int pending;
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t carry;
    const int f = blockIdx.x;
    const int n = v.n_cand[f];
    if (n == 0) {
        return;
    }
    uint8_t* gstate = v.state + static_cast<size_t>(f) * v.maxM;
    const uint32_t* sorted = v.sorted + static_cast<size_t>(f) * v.maxM;
    const uint32_t* xy = v.cand_xy + static_cast<size_t>(f) * v.maxM;
    uint32_t* slot_of = v.rank_of + static_cast<size_t>(f) * v.maxM;  // reused after the rounds
    const uint32_t* rank_of = slot_of;
    const uint8_t* nbr_cnt = v.nbr_cnt + static_cast<size_t>(f) * v.maxM;
    const uint32_t* nbr = v.nbr + static_cast<size_t>(f) * v.maxM * NBR_CAP;
    // all rounds run on a copy of the state in LDS when it fits (it always does below 128 Ki
    // candidates); the fix-point is the same wherever the bytes live
    uint8_t* state = gstate;
    if (n <= lds_cap) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            lds_state[i] = gstate[i];
        }
        state = lds_state;
        __syncthreads();
    }
    // every round decides at least the lowest-ranked undecided candidate, so n rounds bound the
    // loop; the bound only guards the GPU against a corrupted rank table
    for (int iter = 0; iter <= n; ++iter) {
        if (threadIdx.x == 0) {
            pending = 0;
        }
        __syncthreads();
        int mine = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            if (state[i] == NMS_UNDECIDED) {
                const uint8_t s = nms_decide(v, f, i, window, xy, rank_of, state, nbr_cnt, nbr);
                if (s != NMS_UNDECIDED) {
                    state[i] = s;
                } else {
                    mine++;
                }
            }
        }
        if (mine != 0) {
            atomicAdd(&pending, mine);
        }
        __threadfence_block();
        __syncthreads();
        const int p = pending;
        __syncthreads();
        if (p == 0) {
            break;
        }
    }
    // compaction of KEPT in rank order
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        carry = 0;
    }
    __syncthreads();
    slamx_keypoint* kps = v.kps + static_cast<size_t>(f) * v.maxK;
    for (int base = 0; base < n; base += 1024) {
        const int r = base + threadIdx.x;
        uint32_t key = 0;
        bool keep = false;
        if (r < n) {
            key = sorted[r];
            keep = state[key & INDEX_MASK] == NMS_KEPT;
        }
        const unsigned long long bm = __ballot(keep);
        const uint32_t c = static_cast<uint32_t>(__popcll(bm));
        if (lane == 0) {
            wave_sums[wave] = c;
        }
        __syncthreads();
        uint32_t wbase = carry;
        for (int w = 0; w < wave; ++w) {
            wbase += wave_sums[w];
        }
        if (keep) {
            const uint32_t pos = wbase + static_cast<uint32_t>(__popcll(bm & low_mask(lane)));
            if (pos < static_cast<uint32_t>(v.maxK)) {
                const uint32_t pxy = xy[key & INDEX_MASK];
                slamx_keypoint kp;
                kp.x = static_cast<float>(pxy & 0xffffu);
                kp.y = static_cast<float>(pxy >> 16);
                kp.size = 6.0f;  // constants::DEFAULT_KEYPOINT_SIZE
                kp.angle = 0.0f;
                kp.response = static_cast<float>(key >> SCORE_SHIFT);
                kps[pos] = kp;
                slot_of[key & INDEX_MASK] = pos;  // rank_of is dead from here on: reuse it
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
            carry = wbase + c;
        }
        __syncthreads();
    }
    const int total_kept = static_cast<int>(carry);
    __syncthreads();
    // second compaction, in RASTER order: visit[q] = output slot of the q-th kept candidate in
    // raster order.  The describe kernels walk keypoints in this order so that consecutive waves
    // read neighbouring image rows (the rank order is spatially random).
    if (total_kept <= v.maxK) {
        uint32_t* visit = v.visit + static_cast<size_t>(f) * v.maxK;
        if (threadIdx.x == 0) {
            carry = 0;
        }
        __syncthreads();
        for (int base = 0; base < n; base += 1024) {
            const int i = base + threadIdx.x;
            const bool keep = i < n && state[i] == NMS_KEPT;
            const unsigned long long bm = __ballot(keep);
            const uint32_t c = static_cast<uint32_t>(__popcll(bm));
            if (lane == 0) {
                wave_sums[wave] = c;
            }
            __syncthreads();
            uint32_t wbase = carry;
            for (int w = 0; w < wave; ++w) {
                wbase += wave_sums[w];
            }
            if (keep) {
                visit[wbase + static_cast<uint32_t>(__popcll(bm & low_mask(lane)))] = slot_of[i];
            }
            __syncthreads();
            if (threadIdx.x == 1023) {
                carry = wbase + c;
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        int total = total_kept;
        if (total > v.maxK) {
            atomicOr(&v.status[f], STATUS_KP_OVERFLOW);
            total = 0;
        }
        v.n_kp[f] = total;
    }
}

// NonMaxSuppression: 0 -> detect() returns the raster-ordered candidates with response 0
// (feature_detector.cpp:13-15, :64).
__global__ __launch_bounds__(256) void passthrough_kernel(BatchView v) {
    const int f = blockIdx.y;
    const int n = v.n_cand[f];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (n > v.maxK) {
            atomicOr(&v.status[f], STATUS_KP_OVERFLOW);
            v.n_kp[f] = 0;
        } else {
            v.n_kp[f] = n;
        }
    }
    if (n > v.maxK) {
        return;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t pxy = v.cand_xy[static_cast<size_t>(f) * v.maxM + i];
        slamx_keypoint kp;
        kp.x = static_cast<float>(pxy & 0xffffu);
        kp.y = static_cast<float>(pxy >> 16);
        kp.size = 6.0f;
        kp.angle = 0.0f;
        kp.response = 0.0f;
        v.kps[static_cast<size_t>(f) * v.maxK + i] = kp;
    }
}

constexpr int NMS_PARALLEL_ROUNDS = 2;

}  // namespace

void launch_nms(const BatchView& v, const DetectorParams& p, int nframes, hipStream_t stream, hipEvent_t ev_mid) {
    static const bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_finish_kernel),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 131072),
                              true);
    (void)once;
    dim3 grid(grid_x(v.maxM, 256, nframes), nframes);
    hipLaunchKernelGGL(nms_build_kernel, grid, dim3(256), 0, stream, v, p.window);
    if (ev_mid != nullptr) {
        (void)hipEventRecord(ev_mid, stream);
    }
    for (int r = 0; r < NMS_PARALLEL_ROUNDS; ++r) {
        hipLaunchKernelGGL(nms_round_kernel, grid, dim3(256), 0, stream, v, p.window);
    }
    const int lds_cap = v.maxM < 131072 ? (v.maxM + 15) / 16 * 16 : 131072;
    hipLaunchKernelGGL(nms_finish_kernel, dim3(nframes), dim3(1024), static_cast<size_t>(lds_cap), stream, v, p.window,
                       lds_cap);
}

void launch_passthrough_keypoints(const BatchView& v, int nframes, hipStream_t stream) {
    dim3 grid(grid_x(v.maxM, 256, nframes), nframes);
    hipLaunchKernelGGL(passthrough_kernel, grid, dim3(256), 0, stream, v);
}

}  // namespace slamx
