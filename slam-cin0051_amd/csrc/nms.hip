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
#include <cstdlib>

#include <algorithm>

#pragma clang fp contract(off)

namespace slamx {

namespace {

__device__ __forceinline__ void wave_mem_fence() {  // LDS traffic of one wave: order it, keep the compiler from moving it
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int rank_below_mask(unsigned long long m) {  // set bits of m below this lane
    return static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                      __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u)));
}

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
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
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

// ---------------------------------------------------------------------------------------------
// Fast path (2 <= window <= GREEDY_MAX_WINDOW and the cell grid fits LDS): the reference's loop
// itself, one workgroup of 1 + LOOKAHEAD_MAX waves per frame (one serial wave, 4 or 7 filter waves at work).
//
// The kept keypoints live in a grid of window x window pixel cells in LDS.  Kept points are at
// least `window` apart, so a cell holds at most two of them (three points pairwise >= w apart
// need an enclosing square of side 0.966 w > w - 1 for w <= 29), and everything closer than
// `window` to a pixel lies in the 3 x 3 cells around it.  A cell is one dword: two entries
// (oy << 8 | ox, offsets inside the cell), 0x7f7f = empty -- chosen so that an empty entry fails
// the distance test by itself (|127 - offset| - 2w >= w for w <= 24) and both entries of a cell are
// tested at once in packed 16-bit arithmetic with no validity branch.  The test is the reference's:
// for integer coordinates sqrtf(dx^2+dy^2) < float(w) <=> dx^2+dy^2 < w^2.
//
// The candidates are taken in std::sort order, 64 (one chunk) at a time.  Kept points never leave
// the grid, so a test against ANY earlier state of the grid is a valid filter: the LOOKAHEAD filter
// waves run a group of chunks ahead and drop what the grid already suppresses; wave 0 collects the
// rest, in rank order, in a 64-entry batch, tests a full batch against the grid as it is then,
// resolves its members among themselves in rank order -- the lowest surviving lane is kept, it
// knocks out the later lanes closer than `window`, repeat -- and inserts the kept ones.  That is
// the sequential loop of feature_detector.cpp:165-185: a candidate is kept iff no earlier kept
// point is near, and every earlier kept point is either in the grid at the batch test or in the
// same batch.  (The dependency-round form above needs thousands of gathered loads per frame.)
constexpr int GREEDY_MAX_WINDOW = 24;
constexpr int GREEDY_LDS_BUDGET = 150 * 1024;
constexpr uint32_t CELL_EMPTY = 0x7f7f7f7fu;
#ifndef SLAMX_NMS_LOOKAHEAD
#define SLAMX_NMS_LOOKAHEAD 4
#endif
#ifndef SLAMX_NMS_LOOKAHEAD_DENSE
#define SLAMX_NMS_LOOKAHEAD_DENSE 7
#endif
#ifndef SLAMX_NMS_DENSE_CANDIDATES
#define SLAMX_NMS_DENSE_CANDIDATES 32768
#endif
// Filter waves of the greedy kernel.  Two instantiations: 1 + LOOKAHEAD waves, and -- for batches whose candidate
// capacity is at least twice NMS_DENSE_CANDIDATES -- 1 + LOOKAHEAD_MAX waves, of which a frame uses LOOKAHEAD filter
// waves, or all of them when it has NMS_DENSE_CANDIDATES candidates or more (the others only join the barriers:
// +8 % on the frames that do not need them, hence the small instantiation).
// Measured per launch: 80k candidates per frame, window 8: 0.94 / 0.77 / 0.72 ms with 3 / 5 / 7 filter waves; 14.5k
// candidates, window 12: 0.175 / 0.186 / 0.195 ms; 13k candidates on 640 x 480, window 6: 0.261 / 0.276 / 0.287 ms --
// where nine candidates in ten die in the filter the serial wave flushes once per ~10 chunks and longer groups only
// add to what each barrier waits for; at 80k candidates it flushes about once per three chunks and a group of seven
// chunks takes the barrier out of its way.
constexpr int LOOKAHEAD = SLAMX_NMS_LOOKAHEAD;
constexpr int LOOKAHEAD_MAX = SLAMX_NMS_LOOKAHEAD_DENSE > SLAMX_NMS_LOOKAHEAD ? SLAMX_NMS_LOOKAHEAD_DENSE : SLAMX_NMS_LOOKAHEAD;
constexpr int NMS_DENSE_CANDIDATES = SLAMX_NMS_DENSE_CANDIDATES;
constexpr int GREEDY_STATIC_LDS = 1024 + 2 * LOOKAHEAD_MAX * 64 * 8;  // stage + look + counters, rounded up

typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short u2 __attribute__((ext_vector_type(2)));

extern __shared__ uint32_t nms_grid[];

template <int LMAX>
__global__ __launch_bounds__(64 * (1 + LMAX)) void nms_greedy_kernel(BatchView v, int window, int gx2, int gy2, uint32_t magic,
                                                                    int tab_mask, int scratch_words) {
    constexpr int GREEDY_WAVES = 1 + LMAX;
    constexpr int GREEDY_THREADS = 64 * GREEDY_WAVES;
    __shared__ uint32_t wave_sums[2][GREEDY_WAVES];
    __shared__ uint32_t carry;
    __shared__ int kept_total;
    __shared__ uint2 stage[64];  // {y << 16 | x, key} of the candidates that passed the grid test, in rank order
    __shared__ uint2 look[2][LMAX][64];  // chunks filtered ahead of wave 0 (double-buffered by group)
    __shared__ unsigned long long look_mask[2][LMAX];
#ifdef SLAMX_NMS_CLOCKS
    __shared__ unsigned long long nclk[16];  // development: phase stamps and per-role sums (cycles), printed by block 0
    const unsigned long long nc_t0 = clock64();
    unsigned long long nc_flush = 0, nc_filter = 0, nc_barrier = 0, nc_grid = 0, nc_tab = 0, nc_loop = 0;
    int nc_nflush = 0;
#endif
    const int f = blockIdx.x;
    const int n = v.n_cand[f];
    if (n == 0) {
        return;  // scan_kernel already set n_kp = 0
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const uint32_t* __restrict__ sorted = v.sorted + static_cast<size_t>(f) * v.maxM;
    const uint32_t* __restrict__ sorted_xy = v.keys + static_cast<size_t>(f) * v.maxM;  // left there by sort_kernel
    uint8_t* state = v.state + static_cast<size_t>(f) * v.maxM;
    uint32_t* slot_of = v.rank_of + static_cast<size_t>(f) * v.maxM;  // rank_of is not needed on this path
    slamx_keypoint* kps = v.kps + static_cast<size_t>(f) * v.maxK;
    for (int i = threadIdx.x; i < gx2 * gy2; i += GREEDY_THREADS) {
        nms_grid[i] = CELL_EMPTY;
    }
    // occupancy counters of the survivors of the current step, hashed by cell index (see below)
    uint32_t* tab = nms_grid + gx2 * gy2;
    for (int i = threadIdx.x; i <= tab_mask; i += GREEDY_THREADS) {
        tab[i] = 0;
    }
    // The visiting list at the end needs the kept candidates as a bitmap over their raster index and the prefix of its
    // word counts: both are built AFTER the loop in the grid's and the table's own words, which are dead by then (round 5,
    // late: until then the bitmap had LDS of its own, was filled batch by batch, and at 80k candidates with window 8 --
    // a 133 KB grid -- there was no room for it: those frames went through `state` in global memory, a byte store per
    // suppressed candidate and 156 barrier-separated steps at the end, a quarter of the kernel).  Candidate sets too large
    // even for that (tiny frames full of corners) still take the route through `state`.
    const int n_words = (n + 31) >> 5;
    const bool use_bits = 2 * n_words <= scratch_words;  // block-uniform
    __syncthreads();
    {
        const int w = window;
        const int w2 = w * w;
        const u2 W2 = {static_cast<unsigned short>(w2), static_cast<unsigned short>(w2)};
        int count = 0;
        bool grid_overflow = false;
        const int nchunks = (n + 63) / 64;
        // Waves 1..3 run ahead of wave 0, one chunk of 64 candidates (in std::sort order) each: a candidate
        // closer than `window` to a kept point already in the grid can be dropped at once -- kept points
        // never leave the grid, so a test against an older state of it is a valid filter.  Wave 0 collects
        // what passes, in rank order, in `stage`; when 64 are together or nearly so it tests them against
        // the grid as it is now (which holds every kept point of the earlier batches), resolves them among
        // themselves in rank order and inserts the kept ones.  A candidate is kept iff no earlier kept point
        // is near, and every earlier kept point is either in the grid at that moment or in the same batch.
        int pend = 0;  // wave-uniform
        auto near_kept = [&](uint32_t xy) {  // the grid test: true when a kept point is closer than w
            const int x = static_cast<int>(xy & 0xffffu);
            const int y = static_cast<int>(xy >> 16);
            const int cx = static_cast<int>(__umulhi(static_cast<uint32_t>(x), magic));  // x / w (exact below 65536)
            const int cy = static_cast<int>(__umulhi(static_cast<uint32_t>(y), magic));
            const int xr = x - cx * w;
            const int yr = y - cy * w;
            const uint32_t* cell = nms_grid + (cy + 1) * gx2 + (cx + 1);  // the grid has a one-cell empty rim
            // 3 x 3 cells, two entries per cell at once.  All nine reads are issued before the first is
            // used (the wave is alone on its SIMD: every separate wait is a full LDS round trip).
            uint32_t cw[9];
#pragma unroll
            for (int j = -1; j <= 1; ++j) {
#pragma unroll
                for (int i = -1; i <= 1; ++i) {
                    cw[(j + 1) * 3 + (i + 1)] = cell[j * gx2 + i];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            u2 nearest = {0xffff, 0xffff};  // smallest squared distance seen, per entry slot
#pragma unroll
            for (int j = -1; j <= 1; ++j) {
                const short yj = static_cast<short>(yr - j * w);
                const s2 YR = {yj, yj};
#pragma unroll
                for (int i = -1; i <= 1; ++i) {
                    const uint32_t word = cw[(j + 1) * 3 + (i + 1)];
                    const short xi = static_cast<short>(xr - i * w);
                    const s2 XR = {xi, xi};
                    const s2 OX = __builtin_bit_cast(s2, word & 0x00ff00ffu);
                    const s2 OY = __builtin_bit_cast(s2, __builtin_amdgcn_perm(0u, word, 0x0c030c01u));
                    const s2 DX = XR - OX;
                    const s2 DY = YR - OY;
                    const u2 D2 = __builtin_bit_cast(u2, DX * DX + DY * DY);  // <= 2 * 175^2 < 65536
                    nearest = __builtin_elementwise_min(nearest, D2);
                }
            }
            return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(W2, nearest)) != 0u;  // d^2 < w^2
        };
        auto flush = [&]() {
            if (pend == 0) {
                return;
            }
#ifdef SLAMX_NMS_CLOCKS
            const unsigned long long nc_f0 = clock64();
            nc_nflush++;
#endif
            wave_mem_fence();
            const bool have = lane < pend;
            const uint2 e = have ? stage[lane] : make_uint2(0u, 0u);  // idle lanes: a valid cell
            const uint32_t key = e.y;
            const int x = static_cast<int>(e.x & 0xffffu);
            const int y = static_cast<int>(e.x >> 16);
            const int cx = static_cast<int>(__umulhi(static_cast<uint32_t>(x), magic));
            const int cy = static_cast<int>(__umulhi(static_cast<uint32_t>(y), magic));
            const int xr = x - cx * w;
            const int yr = y - cy * w;
            bool alive = have && !near_kept(e.x);  // the grid has grown since the filter ran
#ifdef SLAMX_NMS_CLOCKS
            const unsigned long long nc_f1 = clock64();
            nc_grid += nc_f1 - nc_f0;
#endif
            // The batch among itself.  Two of its members can only interact when they sit in
            // neighbouring cells, which is rare for 64 candidates on ~10^4 cells: every member
            // counts itself into a table slot hashed from its cell and sums the slots of its
            // 3 x 3 cells; a sum of 1 means no other member is near (a hash collision only makes
            // a lonely one look crowded), so it is kept without entering the rank-order loop.
            const int cidx = (cy + 1) * gx2 + (cx + 1);
            if (alive) {
                atomicAdd(&tab[cidx & tab_mask], 1u);
            }
            uint32_t tw[9];
#pragma unroll
            for (int j = -1; j <= 1; ++j) {
#pragma unroll
                for (int i = -1; i <= 1; ++i) {
                    tw[(j + 1) * 3 + (i + 1)] = tab[(cidx + j * gx2 + i) & tab_mask];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            uint32_t crowd = 0;
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                crowd += tw[q];
            }
            const bool lonely = alive && crowd == 1u;
            if (alive) {
                tab[cidx & tab_mask] = 0;
            }
            unsigned long long kept = __ballot(lonely);
            alive = alive && !lonely;
#ifdef SLAMX_NMS_CLOCKS
            const unsigned long long nc_f2 = clock64();
            nc_tab += nc_f2 - nc_f1;
#endif
            // the rest in rank (= lane) order
            // Two members per step: the lowest surviving lane l0 is kept; the next one, l1, is kept as well unless l0
            // suppresses it -- both neighbourhoods are evaluated side by side and the scalar side decides which to apply
            // (the step's length is its chain of vector -> scalar -> vector hops, not its arithmetic: one member per step
            // 0.171 ms per 512 default frames, two 0.156; a general form for three, four or six per step pays more for its
            // scalar bookkeeping than it saves: 0.168 / 0.174 / 0.180).
            unsigned long long a = __ballot(alive);
            while (a != 0) {
                const int l0 = __ffsll(static_cast<long long>(a)) - 1;
                const unsigned long long rest = a & (a - 1);  // without l0
                const int l1 = rest != 0 ? __ffsll(static_cast<long long>(rest)) - 1 : l0;
                const int dx0 = x - __builtin_amdgcn_readlane(x, l0);
                const int dy0 = y - __builtin_amdgcn_readlane(y, l0);
                const int dx1 = x - __builtin_amdgcn_readlane(x, l1);
                const int dy1 = y - __builtin_amdgcn_readlane(y, l1);
                const unsigned long long near0 = __ballot(dx0 * dx0 + dy0 * dy0 < w2);  // includes lane l0 itself
                const unsigned long long near1 = __ballot(dx1 * dx1 + dy1 * dy1 < w2);
                kept |= 1ull << l0;
                a &= ~near0;
                if (((near0 >> l1) & 1ull) == 0) {  // l1 survives l0 (l1 == l0, nothing else left: near1 == near0)
                    kept |= 1ull << l1;
                    a &= ~near1;
                }
            }
#ifdef SLAMX_NMS_CLOCKS
            nc_loop += clock64() - nc_f2;
#endif
            const bool mine = (kept >> lane) & 1ull;
            if (mine) {
                const uint32_t entry = static_cast<uint32_t>((yr << 8) | xr);
                uint32_t* c = nms_grid + cidx;
                const uint32_t old = atomicCAS(c, CELL_EMPTY, (CELL_EMPTY & 0xffff0000u) | entry);
                if (old != CELL_EMPTY) {
                    // first slot taken (now or in an earlier step): the second one must be free
                    const uint32_t prev = atomicXor(c, (0x7f7fu ^ entry) << 16);
                    grid_overflow = grid_overflow || (prev >> 16) != 0x7f7fu;
                }
                const int pos = count + __popcll(kept & low_mask(lane));
                const uint32_t idx = key & INDEX_MASK;
                if (pos < v.maxK) {
                    slamx_keypoint kp;
                    kp.x = static_cast<float>(x);
                    kp.y = static_cast<float>(y);
                    kp.size = 6.0f;  // constants::DEFAULT_KEYPOINT_SIZE
                    kp.angle = 0.0f;
                    kp.response = static_cast<float>(key >> SCORE_SHIFT);
                    kps[pos] = kp;
                    if (use_bits) {
                        slot_of[pos] = idx;  // slot -> raster index, contiguous
                    } else {
                        slot_of[idx] = static_cast<uint32_t>(pos);
                    }
                }
            }
            if (have && !use_bits) {
                state[key & INDEX_MASK] = mine ? NMS_KEPT : NMS_SUPPRESSED;
            }
            count += __popcll(kept);
            pend = 0;
            wave_mem_fence();
#ifdef SLAMX_NMS_CLOCKS
            nc_flush += clock64() - nc_f0;
#endif
        };
        // wave 0: the survivors of one filtered chunk (mask am, na of them, lane l holds entry e) go to the batch in rank
        // order.  A batch is resolved when it holds 64: a chunk's survivors are split between two batches where they do
        // not fit (any grouping of the rank order into batches gives the sequential result; what a resolve pass costs,
        // ~2,900 cycles of dependent LDS round trips, does not depend on how full the batch is -- round 5: until then a
        // batch was resolved from 44 members on, 38 on average).
        auto append = [&](unsigned long long am, int na, uint2 e) {
            const bool mine = ((am >> lane) & 1ull) != 0;
            const int r = rank_below_mask(am);
            const int room = 64 - pend;
            if (na > room) {
                if (mine && r < room) {
                    stage[pend + r] = e;
                }
                pend = 64;
                flush();
                if (mine && r >= room) {
                    stage[r - room] = e;
                }
                pend = na - room;
            } else {
                if (mine) {
                    stage[pend + r] = e;
                }
                pend += na;
            }
        };
        // filter waves 1 .. L: chunk (wave - 1) + L * j at step j, loads two steps ahead; waves beyond L only join the barriers
        const int L = n >= NMS_DENSE_CANDIDATES ? LMAX : (LMAX < LOOKAHEAD ? LMAX : LOOKAHEAD);  // block-uniform
        const bool filters = wave >= 1 && wave <= L;
        const int hc0 = wave - 1;
        auto chunk_load = [&](const uint32_t* src, int chunk) {
            const int r = chunk * 64 + lane;
            return filters && chunk < nchunks && r < n ? src[r] : 0u;
        };
        uint32_t key_a = chunk_load(sorted, hc0);
        uint32_t xy_a = chunk_load(sorted_xy, hc0);
        uint32_t key_b = chunk_load(sorted, hc0 + L);
        uint32_t xy_b = chunk_load(sorted_xy, hc0 + L);
        const int ngroups = (nchunks + L - 1) / L;
        // One loop per role, the same number of barriers in each.  (In one shared loop the wait for the filter waves'
        // prefetched chunk -- s_waitcnt vmcnt(0) at the loop head -- also made wave 0 wait, every group, until the
        // keypoint records of its last batch had reached memory: stores count in vmcnt too.)
        if (filters) {
            for (int g = -1; g < ngroups; ++g) {
                // group g + 1 into look[(g + 1) & 1]
                const int c = (g + 1) * L + hc0;
                const uint32_t key_c = chunk_load(sorted, c + 2 * L);
                const uint32_t xy_c = chunk_load(sorted_xy, c + 2 * L);
                if (c < nchunks) {
#ifdef SLAMX_NMS_CLOCKS
                    const unsigned long long nc_n0 = clock64();
#endif
                    const bool valid = c * 64 + lane < n;
                    const bool alive = valid && !near_kept(xy_a);
                    const unsigned long long am = __ballot(alive);
                    look[(g + 1) & 1][hc0][lane] = make_uint2(xy_a, key_a);
                    if (lane == 0) {
                        look_mask[(g + 1) & 1][hc0] = am;
                    }
                    if (valid && !alive && !use_bits) {
                        state[key_a & INDEX_MASK] = NMS_SUPPRESSED;
                    }
#ifdef SLAMX_NMS_CLOCKS
                    nc_filter += clock64() - nc_n0;
#endif
                }
                key_a = key_b;
                key_b = key_c;
                xy_a = xy_b;
                xy_b = xy_c;
#ifdef SLAMX_NMS_CLOCKS
                const unsigned long long nc_b0 = clock64();
#endif
                __threadfence_block();
                __syncthreads();
#ifdef SLAMX_NMS_CLOCKS
                nc_barrier += clock64() - nc_b0;
#endif
            }
        } else if (wave == 0) {
            for (int g = -1; g < ngroups; ++g) {
                if (g >= 0) {
                    if constexpr (LMAX > LOOKAHEAD) {
                        // (dense instantiation) the group's masks and entries are all requested before the first is
                        // used: one LDS round trip for the group instead of two per chunk -- c3 0.77 -> 0.70 ms; with
                        // three chunks a group, most of them without a survivor, the same form costs 3 %: the small
                        // instantiation keeps the loop
                        unsigned long long am[LMAX];
                        uint2 ent[LMAX];
#pragma unroll
                        for (int h = 0; h < LMAX; ++h) {
                            const bool there = h < L && g * L + h < nchunks;
                            am[h] = there ? look_mask[g & 1][h] : 0ull;
                            ent[h] = look[g & 1][h][lane];
                        }
#pragma unroll
                        for (int h = 0; h < LMAX; ++h) {
                            const int na = __popcll(am[h]);
                            if (na == 0) {
                                continue;
                            }
                            append(am[h], na, ent[h]);
                        }
                    } else {
                        // the group's masks in one LDS round trip, the entries only of the chunks that have survivors
                        unsigned long long am[LMAX];
#pragma unroll
                        for (int h = 0; h < LMAX; ++h) {
                            am[h] = look_mask[g & 1][h];
                        }
#pragma unroll
                        for (int h = 0; h < LMAX; ++h) {
                            const int na = h < L && g * L + h < nchunks ? __popcll(am[h]) : 0;
                            if (na == 0) {
                                continue;
                            }
                            append(am[h], na, look[g & 1][h][lane]);
                        }
                    }
                
                }
                __threadfence_block();
                __syncthreads();
            }
        } else {
            for (int g = -1; g < ngroups; ++g) {
                __threadfence_block();
                __syncthreads();
            }
        }
        if (wave == 0) {
            flush();
            if (__ballot(grid_overflow) != 0 && lane == 0) {
                atomicOr(&v.status[f], STATUS_KP_OVERFLOW);  // cannot happen for window <= GREEDY_MAX_WINDOW; fail loudly
            }
            if (lane == 0) {
                kept_total = count;
                carry = 0;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
#ifdef SLAMX_NMS_CLOCKS
    if (lane == 0 && wave == 0) {
        nclk[0] = clock64() - nc_t0;
        nclk[1] = nc_flush;
        nclk[2] = static_cast<unsigned long long>(nc_nflush);
        nclk[5] = nc_grid;
        nclk[6] = nc_tab;
        nclk[7] = nc_loop;
    }
    if (lane == 0 && wave == 1) {
        nclk[3] = nc_filter;
        nclk[4] = nc_barrier;
    }
#endif
    const int total_kept = kept_total;
    // raster-order visiting list for the describe kernels (see nms_finish_kernel)
    if (total_kept <= v.maxK && use_bits) {
        // rank of a kept candidate among the kept ones = kept bits before it: the bitmap from the slot -> raster index
        // records, the exclusive prefix of its word popcounts, then one independent lookup per keypoint
        uint32_t* kept_bits = nms_grid;
        uint32_t* word_prefix = nms_grid + n_words;
        for (int i = threadIdx.x; i < n_words; i += GREEDY_THREADS) {
            kept_bits[i] = 0;
        }
        __syncthreads();
        for (int pos = threadIdx.x; pos < total_kept; pos += GREEDY_THREADS) {
            const uint32_t idx = slot_of[pos];
            atomicOr(&kept_bits[idx >> 5], 1u << (idx & 31u));
        }
        __syncthreads();
        constexpr int PER = 8;
        uint32_t run_total = 0;  // identical in all threads
        int buf = 0;
        for (int base = 0; base < n_words; base += GREEDY_THREADS * PER) {
            const int i0 = base + static_cast<int>(threadIdx.x) * PER;
            uint32_t c[PER];
            uint32_t t = 0;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                c[k] = i0 + k < n_words ? static_cast<uint32_t>(__popc(kept_bits[i0 + k])) : 0u;
                t += c[k];
            }
            uint32_t s = t;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t u = __shfl_up(s, d);
                if (lane >= d) {
                    s += u;
                }
            }
            if (lane == 63) {
                wave_sums[buf][wave] = s;
            }
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int q = 0; q < GREEDY_WAVES; ++q) {
                const uint32_t ws = wave_sums[buf][q];
                before += q < wave ? ws : 0u;
                total += ws;
            }
            uint32_t run = run_total + before + s - t;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                if (i0 + k < n_words) {
                    word_prefix[i0 + k] = run;
                }
                run += c[k];
            }
            run_total += total;
            buf ^= 1;
        }
        __syncthreads();
        uint32_t* visit = v.visit + static_cast<size_t>(f) * v.maxK;
        for (int pos = threadIdx.x; pos < total_kept; pos += GREEDY_THREADS) {
            const uint32_t idx = slot_of[pos];
            const uint32_t w = idx >> 5;
            const uint32_t below = kept_bits[w] & ((1u << (idx & 31u)) - 1u);
            visit[word_prefix[w] + static_cast<uint32_t>(__popc(below))] = static_cast<uint32_t>(pos);
        }
    } else if (total_kept <= v.maxK) {
        uint32_t* visit = v.visit + static_cast<size_t>(f) * v.maxK;
        for (int base = 0; base < n; base += GREEDY_THREADS) {
            const int i = base + threadIdx.x;
            const bool keep = i < n && state[i] == NMS_KEPT;
            const unsigned long long bm = __ballot(keep);
            const uint32_t c = static_cast<uint32_t>(__popcll(bm));
            if (lane == 0) {
                wave_sums[0][wave] = c;
            }
            __syncthreads();
            uint32_t wbase = carry;
            for (int q = 0; q < wave; ++q) {
                wbase += wave_sums[0][q];
            }
            if (keep) {
                visit[wbase + static_cast<uint32_t>(__popcll(bm & low_mask(lane)))] = slot_of[i];
            }
            __syncthreads();
            if (threadIdx.x == GREEDY_THREADS - 1) {
                carry = wbase + c;
            }
            __syncthreads();
        }
    }
#ifdef SLAMX_NMS_CLOCKS
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        printf("nms frame 0: %d candidates, %d kept: until the end of the loop %llu cycles (wave 0: %llu flushes, %llu cycles in them = grid test %llu + lonely test %llu + rank-order loop %llu + insert; wave 1: %llu cycles filtering, %llu at the group barriers), whole kernel %llu\n",
               n, total_kept, nclk[0], nclk[2], nclk[1], nclk[5], nclk[6], nclk[7], nclk[3], nclk[4], clock64() - nc_t0);
    }
#endif
    if (threadIdx.x == 0) {
        int total = total_kept;
        if (total > v.maxK) {
            atomicOr(&v.status[f], STATUS_KP_OVERFLOW);
            total = 0;
        }
        if ((v.status[f] & STATUS_KP_OVERFLOW) != 0) {
            total = 0;
        }
        v.n_kp[f] = total;
    }
}

}  // namespace

bool nms_uses_greedy_path(const BatchView& v, const DetectorParams& p) {
    if (p.window < 2 || p.window > GREEDY_MAX_WINDOW) {
        return false;
    }
    const int gx2 = (v.cols + p.window - 1) / p.window + 2;
    const int gy2 = (v.rows + p.window - 1) / p.window + 2;
    return static_cast<size_t>(gx2) * gy2 * sizeof(uint32_t) + 1024 <= static_cast<size_t>(GREEDY_LDS_BUDGET);
}

void launch_nms(const BatchView& v, const DetectorParams& p, int nframes, hipStream_t stream, hipEvent_t ev_mid) {
    static std::atomic<unsigned long long> lds_finish{0};
    allow_dynamic_lds(reinterpret_cast<const void*>(nms_finish_kernel), 131072, lds_finish);
    if (nms_uses_greedy_path(v, p)) {
        const int gx2 = (v.cols + p.window - 1) / p.window + 2;
        const int gy2 = (v.rows + p.window - 1) / p.window + 2;
        const size_t bytes = static_cast<size_t>(gx2) * gy2 * sizeof(uint32_t);
        {
            static std::atomic<unsigned long long> lds_greedy{0}, lds_greedy_max{0};
            allow_dynamic_lds(reinterpret_cast<const void*>(nms_greedy_kernel<LOOKAHEAD>), GREEDY_LDS_BUDGET, lds_greedy);
            allow_dynamic_lds(reinterpret_cast<const void*>(nms_greedy_kernel<LOOKAHEAD_MAX>), GREEDY_LDS_BUDGET, lds_greedy_max);
            const uint32_t magic = static_cast<uint32_t>(((1ull << 32) + p.window - 1) / p.window);
            // the hashed survivor table takes what is left of the LDS budget (a power of two of dwords)
            // (a power of two of dwords), but at most half of a CU's 160 KB when that is within reach, so that
            // two frames share a CU in batches of more than 256 frames
            const size_t budget =
                (bytes + 4096 * 4 <= 80 * 1024 ? 80 * 1024 : static_cast<size_t>(GREEDY_LDS_BUDGET)) - GREEDY_STATIC_LDS;
            int tab_entries = 256;
            while (tab_entries < 16384 && bytes + static_cast<size_t>(tab_entries) * 8 <= budget) {
                tab_entries *= 2;
            }
            const size_t total = bytes + static_cast<size_t>(tab_entries) * 4;
            // (grid + table: after the loop they hold the kept-candidate bitmap and the prefix of its word counts)
            // SLAMX_NMS_STATE_ROUTE=1 (tests): no room for the bitmap -- every frame takes the route through `state`, which
            // since round 5 only candidate sets too large for grid + table take by themselves
            const char* force_state = getenv("SLAMX_NMS_STATE_ROUTE");
            const int scratch_words = force_state != nullptr && force_state[0] == '1' ? 0 : gx2 * gy2 + tab_entries;
            if (v.maxM >= 2 * NMS_DENSE_CANDIDATES) {
                hipLaunchKernelGGL(nms_greedy_kernel<LOOKAHEAD_MAX>, dim3(nframes), dim3(64 * (1 + LOOKAHEAD_MAX)), total, stream, v,
                                   p.window, gx2, gy2, magic, tab_entries - 1, scratch_words);
            } else {
                hipLaunchKernelGGL(nms_greedy_kernel<LOOKAHEAD>, dim3(nframes), dim3(64 * (1 + LOOKAHEAD)), total, stream, v, p.window,
                                   gx2, gy2, magic, tab_entries - 1, scratch_words);
            }
            if (ev_mid != nullptr) {
                (void)hipEventRecord(ev_mid, stream);
            }
            return;
        }
    }
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
