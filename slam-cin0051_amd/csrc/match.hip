// match.hip -- brute-force Hamming 1-NN with second-best tracking, jump-radius penalty and
// Lowe ratio test.  Two all-pairs scans with identical results: match_kernel (xor + popcount, below) and
// match_spatial_kernel (int8 matrix cores, further down; which of the two takes a frame pair is decided per pair on the
// device: MatchView::mfma_flag).
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
    if (mv.mfma_flag != nullptr && mv.mfma_flag[tf] != 0) {
        return;  // this pair belongs to match_spatial_kernel (uniform)
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

// ---------------------------------------------------------------------------------------------------
// Hamming distances on the int8 matrix cores (descriptors with at most 62 live bits, i.e. the shipped
// configuration's 46 surviving BRIEF pairs).
//
// For bit vectors a, b:  hamming(a, b) = |a| + |b| - 2 a.b  -- an exact integer identity.  With every
// descriptor expanded to 64 int8 values
//     train form  t' = [ bit 0 .. bit 61 |  |t| |  1  ]        (spatial_order_kernel writes this form)
//     query form  q' = [ -2 bit 0 .. -2 bit 61 |  1  | |q| ]   (derived from the train form at load)
// the dot product q'.t' IS the Hamming distance, and one v_mfma_i32_16x16x64_i8 (K = 64) gives the
// 256 distances of 16 train x 16 query descriptors in 16 cycles: 16 pairs per cycle and SIMD against 4
// for xor + popcount at its peak.  Integer arithmetic, so the values are the reference's
// (calculateHammingDistance, common.hpp:40-50) bit for bit; both operands use the same byte -> k
// assignment, so the k order inside the instruction does not matter.
//
// What stays on the vector ALU is the per-query bookkeeping, and the matrix cores leave it 4
// instructions per MFMA before it becomes the bound.  C/D layout: lane (c = lane & 15, g = lane >> 4)
// receives rows 4g..4g+3 of column c: four consecutive TRAIN descriptors against ONE query.  So:
//   * the accumulator input of the MFMA is -bound(query), a stale upper bound of what can still change the
//     query's result: the product comes out as distance - bound, a pair passes iff it is negative, and a lane
//     knows whether ANY of its 16 pairs of a train tile (4 query tiles x 4 rows) passes from the sign of
//     their OR: 8 v_or3 + 1 compare per 4 MFMAs;
//   * a passing (lane, tile) is appended to the wave's work list in LDS, and whenever 64 entries wait the
//     whole wave works them off, one entry per lane whoever pushed it: xor + popcount of the 16 pairs from the
//     compact bits; the one or two of them that can still change the query's keys go to a second list of PAIRS,
//     which is settled 64 at a time as well: the reference's penalty arithmetic and the update of the query's
//     state, which lives in LDS as two 64-bit keys (distance << 32 | train index):
//         old = atomicMin(best_key, key);  atomicMin(second_key, max(old, key))
//     Every key but the final minimum is offered to second_key exactly once (when it loses), so the two
//     keys end as the two lexicographically smallest of all pairs, whatever the order of the updates:
//     the smallest distance with the LOWEST train index -- the reference's strict '<' in an ascending
//     scan (feature_matcher.cpp:132-141) -- and the second smallest distance of the multiset.
// The slow path thus runs on full waves, needs no per-lane state and leaves no merge for the end.
constexpr int MM_QT = 4;         // query tiles (of 16) per wave: 64 queries per wave
constexpr int MM_MAX_BITS = 62;
constexpr unsigned long long MM_KEY_NONE = 0x7fffffffffffffffull;  // distance INT_MAX, index -1

typedef int v4i __attribute__((ext_vector_type(4)));

// one wave's LDS accesses are issued in order; this only keeps the compiler from moving or caching them across it
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------------
// The matrix-core scan in SPATIAL order.
//
// What the scan above pays per QUERY is the convergence of its bound: `second` is an order statistic of the
// penalised distances seen so far, and only trains within 500 px of the query enter it unpenalised (the jump
// penalty multiplies the distance of everything farther away by 2..5), so in index order -- spatially random,
// the keypoints are sorted by score -- a query's bound tightens as slowly as its NEAR neighbours turn up: ~48
// filter passes per query at 10k keypoints, ~40 at 2k, every one of them a trip through the slow path (measured:
// 54 % of the kernel's time at 10k keypoints, 73 % at 2k).  Both sets are therefore put into a spatial order first
// (spatial_order_kernel: cells of ~64 keypoints, boustrophedon), a wave's 64 queries are neighbours, and the wave
// visits the train set in STAGES of 64 neighbours, nearest stage first: the true match and the look-alikes that
// set `second` come in the first few stages, and the far stages are scanned with the final bound.
// Exactness does not depend on the order: the two keys (distance << 32 | ORIGINAL train index) end as the two
// lexicographically smallest of all pairs whatever the order of the updates (see above) -- but the tile filter's
// strict '<' relied on ascending indices.  In any order a pair of distance d may be dropped iff d > second, or
// d == second > best (it could only replace the second key by an equal distance); a pair with d == second == best
// may carry the lower index and passes.  The bound fed to the MFMA is therefore second + [best == second], read
// stale (stale values only make it larger: a superset passes).
// The geometry also gives every stage a CLASS for the wave (bounding boxes of the stage and of the wave's queries):
//   near  -- no pair of the two boxes is farther apart than 499 px: the penalty never applies (feature_matcher.cpp:164),
//            the slow path skips its arithmetic;
//   far   -- every pair is at least 501 px apart: the penalised distance is at least TWICE the Hamming distance
//            (factor 1 + d / 500 > 2, truncation after the multiply), so a pair can only matter while
//            2 * hamming < bound: the filter runs with ceil(bound / 2), which hardly anything passes;
//   mixed -- everything else (and every stage when a coordinate is not finite: the reference's comparison with NaN
//            is false, i.e. no penalty -- nothing may be assumed then).
// No LDS stage and no barrier: a wave reads its A operands straight from global memory (a tile = 16 consecutive
// expanded descriptors = 1 KB, one dwordx4 per lane, the next stage's four tiles in flight while the current
// stage's 16 MFMAs run); the expanded train set of a frame (0.7 MB at 10k keypoints) is L2-resident.
#ifdef SLAMX_MATCH_CLOCKS  // development build only (make FLAGS_match=-DSLAMX_MATCH_CLOCKS): where a wave of the spatial scan spends its time
__device__ unsigned long long sp_clocks[8];  // waves, then s_memtime ticks (100 MHz): setup, stage order, startup tiles, steady scan, of it slow path, drain
#define SP_CLK(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define SP_CLK(v) ((void)0)
#endif
#ifdef SLAMX_MATCH_STATS  // development build only (make FLAGS_match=-DSLAMX_MATCH_STATS): slow-path volume of the spatial scan
__device__ unsigned long long sp_stats[8];  // waves, tile entries, pairs queued, pairs that reached the keys, stages
#define SP_STAT(i, v) sp_stat_##i += (v)
#else
#define SP_STAT(i, v) ((void)0)
#endif
constexpr int SP_STAGE = 64;          // trains per stage (4 tiles of 16)
#ifndef SLAMX_SP_CELL
#define SLAMX_SP_CELL 64
#endif
// keypoints per cell of the spatial order, on average (match stage per 512 frames of 2.1k keypoints with 16 / 32 / 48 / 64 /
// 96 / 128: 0.43 / 0.47 / 0.40 / 0.42 / 0.42 / 0.45 ms -- how the cell grid happens to cut the frame, not a trend; c3 2.03-2.09)
constexpr int SP_CELL = SLAMX_SP_CELL;
constexpr int SP_MAX_STAGES = 1025;   // maxK <= 65536
constexpr int SP_BUCKETS = 128;       // stage order: counting sort on the centre distance in 32-px steps
constexpr int SP_SEED_STAGES = 2;      // leading near stages scanned densely, a query per lane, before the filter has a bound
constexpr int SP_STARTUP_TILES = 8;   // the first tiles are filtered one by one (the bound still moves with every tile)
constexpr int SP_FIFO = 384;          // tile entries: < 64 waiting + the 4 x 64 pushes of a stage (+ 64 slack)
constexpr int SP_PAIRS = 192;         // pair entries: < 64 waiting + 64 pushes (+ 64 slack)
constexpr uint32_t SP_NEAR = 1u, SP_FAR = 2u;

__device__ __forceinline__ int clamp_coord(float v) {  // NaN -> -2^20; only ever used for ORDERING, never for results
    return static_cast<int>(fminf(fmaxf(v, -1048576.0f), 1048576.0f));
}
__device__ __forceinline__ bool coord_ok(float v) { return fabsf(v) < 1.0e6f; }  // false for NaN and infinities

// Which scan takes the pairs that have a frame as TRAIN set: large sets the matrix cores, small ones xor + popcount,
// medium ones the matrix cores iff their keypoints spread wide (slamx_internal.h).  One WAVE per frame (the box of a
// few thousand keypoints is a handful of loads per lane and a wave reduction).
__global__ __launch_bounds__(1024) void scan_choice_kernel(MatchView mv, int first_frame, int nframes) {
    const int lane = threadIdx.x & 63;
    const int k = static_cast<int>(blockIdx.x) * 16 + static_cast<int>(threadIdx.x >> 6);
    if (k >= nframes) {
        return;
    }
    const int f = first_frame + k;
    const int m = mv.n_kp[f];
    int flag = m >= mv.mfma_min_train ? 1 : 0;
    if (flag == 0 && m >= mv.mfma_min_wide && mv.kps != nullptr) {
        const slamx_keypoint* kp = mv.kps + static_cast<size_t>(f) * mv.maxK;
        int lx = 0x7fffffff, ly = 0x7fffffff, hx = -0x7fffffff, hy = -0x7fffffff;
        for (int i = lane; i < m; i += 64) {
            const int x = clamp_coord(kp[i].x), y = clamp_coord(kp[i].y);
            lx = min(lx, x);
            ly = min(ly, y);
            hx = max(hx, x);
            hy = max(hy, y);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lx = min(lx, __shfl_xor(lx, d));
            ly = min(ly, __shfl_xor(ly, d));
            hx = max(hx, __shfl_xor(hx, d));
            hy = max(hy, __shfl_xor(hy, d));
        }
        const float w = static_cast<float>(hx - lx), h = static_cast<float>(hy - ly);
        flag = w * w + h * h >= MATCH_MFMA_WIDE_DIAGONAL * MATCH_MFMA_WIDE_DIAGONAL ? 1 : 0;
    }
    if (lane == 0) {
        mv.mfma_flag[f] = flag;
    }
}

// One workgroup per frame: keypoints -> spatial order.  Writes, for sorted position p of frame f:
//   xdesc [f][p][64] train-form expanded descriptor, sbits[f][p] compact bits, sxyi[f][p] {x, y, original index, 0},
//   sbox[f][p / 64] stage bounding boxes; positions [n, roundup(n, 64)) get a descriptor no bound lets through.
// Without keypoints (no penalty, no coordinates) the order is the identity.
// [first_query, last_query]: the frames of this launch that are QUERY frames of a pair of the call (the frame before the
// first query and a far-away first_train_frame are train sets only: their own predecessor's flag was not written by this
// call's scan_choice_kernel and is not looked at).
__global__ __launch_bounds__(1024) void spatial_order_kernel(MatchView mv, int first_frame, int first_train_frame, int first_query,
                                                             int last_query) {
    __shared__ int hist[4096];
    __shared__ int bbox[4];
    __shared__ int wave_tot[16];
    __shared__ int stage_box[SP_MAX_STAGES][4];  // min x, min y (floor), max x, max y (ceil); [0] = INT_MIN marks a bad stage
    const int f = first_frame + blockIdx.x;
    const int n = mv.n_kp[f];
    const int pred = f > 0 ? f - 1 : first_train_frame;
    const int tid = threadIdx.x;
    const bool is_query = f >= first_query && f <= last_query && pred >= 0;
    if (mv.mfma_flag[f] == 0 && (!is_query || mv.mfma_flag[pred] == 0)) {
        return;  // neither the train set nor the query frame of a matrix-core pair (scan_choice_kernel; uniform)
    }
    const int stride = mv.width / 4;
    const bool use_kp = mv.kps != nullptr;
    const slamx_keypoint* kps = use_kp ? mv.kps + static_cast<size_t>(f) * mv.maxK : nullptr;
    uint8_t* xd = mv.xdesc + static_cast<size_t>(f) * mv.maxKp * 64;
    uint4* sb = mv.sbits + static_cast<size_t>(f) * mv.maxKp;
    float4* sx = mv.sxyi + static_cast<size_t>(f) * mv.maxKp;
    float4* sc = mv.sbox + static_cast<size_t>(f) * (mv.maxKp / 64);
    const int nst = (n + SP_STAGE - 1) / SP_STAGE;
    for (int i = tid; i < 4096; i += 1024) {
        hist[i] = 0;
    }
    for (int i = tid; i < nst; i += 1024) {
        stage_box[i][0] = 0x7fffffff;
        stage_box[i][1] = 0x7fffffff;
        stage_box[i][2] = -0x7fffffff;
        stage_box[i][3] = -0x7fffffff;
    }
    if (tid < 4) {
        bbox[tid] = tid < 2 ? 0x7fffffff : -0x7fffffff;
    }
    __syncthreads();
    // A thread owns keypoints tid, tid + 1024, ...: up to 16 of them (every set of at most 16384 keypoints) are read
    // ONCE, all loads in flight together, and stay in registers through the three passes (box, cell counts, scatter);
    // larger sets read their keypoints again in every pass.  With one workgroup per frame the kernel is a chain of
    // memory round trips, not of arithmetic.
    constexpr int CK = 16;
    const bool cached = n <= CK * 1024;
    float kx[CK], ky[CK];
    if (use_kp && cached) {
#pragma unroll
        for (int j = 0; j < CK; ++j) {
            const int i = tid + 1024 * j;
            kx[j] = i < n ? kps[i].x : 0.0f;
            ky[j] = i < n ? kps[i].y : 0.0f;
        }
    }
    int side = 1, ncx = 1, ncy = 1, x0 = 0, y0 = 0;
    float inv_side = 1.0f;
    if (use_kp) {
        int lx = 0x7fffffff, ly = 0x7fffffff, hx = -0x7fffffff, hy = -0x7fffffff;
        auto grow = [&](float xf, float yf) {
            const int x = clamp_coord(xf), y = clamp_coord(yf);
            lx = min(lx, x);
            ly = min(ly, y);
            hx = max(hx, x);
            hy = max(hy, y);
        };
        if (cached) {
#pragma unroll
            for (int j = 0; j < CK; ++j) {
                if (tid + 1024 * j < n) {
                    grow(kx[j], ky[j]);
                }
            }
        } else {
            for (int i = tid; i < n; i += 1024) {
                grow(kps[i].x, kps[i].y);
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {  // one atomic per wave, not per thread
            lx = min(lx, __shfl_xor(lx, d));
            ly = min(ly, __shfl_xor(ly, d));
            hx = max(hx, __shfl_xor(hx, d));
            hy = max(hy, __shfl_xor(hy, d));
        }
        if ((tid & 63) == 0) {
            atomicMin(&bbox[0], lx);
            atomicMin(&bbox[1], ly);
            atomicMax(&bbox[2], hx);
            atomicMax(&bbox[3], hy);
        }
        __syncthreads();
        x0 = bbox[0];
        y0 = bbox[1];
        const float w = static_cast<float>(bbox[2] - x0 + 1), h = static_cast<float>(bbox[3] - y0 + 1);
        // cells of about one stage's worth of keypoints, at most 64 x 64 of them
        side = max(1, static_cast<int>(sqrtf(w * h * static_cast<float>(SP_CELL) / static_cast<float>(max(n, 1)))));
        side = max(side, max(static_cast<int>(w) / 64 + 1, static_cast<int>(h) / 64 + 1));
        ncx = min(64, static_cast<int>(w) / side + 1);
        ncy = min(64, static_cast<int>(h) / side + 1);
        inv_side = 1.0f / static_cast<float>(side);
    }
    // (a float multiply instead of the integer division: a keypoint on a cell border may land in either cell, the
    // same one in both passes -- the order is a heuristic, the result does not depend on it)
    auto cell_of = [&](float xf, float yf) {
        const int cx = min(ncx - 1, static_cast<int>(static_cast<float>(clamp_coord(xf) - x0) * inv_side));
        const int cy = min(ncy - 1, static_cast<int>(static_cast<float>(clamp_coord(yf) - y0) * inv_side));
        return cy * ncx + ((cy & 1) != 0 ? ncx - 1 - cx : cx);  // boustrophedon: consecutive cells are neighbours
    };
    if (use_kp) {
        if (cached) {
#pragma unroll
            for (int j = 0; j < CK; ++j) {
                if (tid + 1024 * j < n) {
                    atomicAdd(&hist[cell_of(kx[j], ky[j])], 1);
                }
            }
        } else {
            for (int i = tid; i < n; i += 1024) {
                atomicAdd(&hist[cell_of(kps[i].x, kps[i].y)], 1);
            }
        }
        __syncthreads();
        // exclusive prefix over the 4096 cells: 4 per thread, wave scan, wave totals
        const int lane = tid & 63, wave = tid >> 6;
        int v[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = hist[4 * tid + k];
            s += v[k];
        }
        int inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d);
            if (lane >= d) {
                inc += o;
            }
        }
        if (lane == 63) {
            wave_tot[wave] = inc;
        }
        __syncthreads();
        int base = inc - s;
        for (int w2 = 0; w2 < wave; ++w2) {
            base += wave_tot[w2];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hist[4 * tid + k] = base;
            base += v[k];
        }
        __syncthreads();
    }
    const uint32_t* rows = reinterpret_cast<const uint32_t*>(mv.desc + static_cast<size_t>(f) * mv.maxK * mv.width);
    // one keypoint -> its sorted position: compact bits, position record, expanded descriptor, stage box
    auto place = [&](int i, float x, float y, uint32_t w0, uint32_t w1) {
        const int p = use_kp ? atomicAdd(&hist[cell_of(x, y)], 1) : i;  // (the order inside a cell is immaterial)
        const unsigned long long bits = static_cast<unsigned long long>(w0) | (static_cast<unsigned long long>(w1) << 32);
        if (use_kp) {
            int* box = stage_box[p >> 6];
            if (coord_ok(x) && coord_ok(y)) {
                atomicMin(&box[0], static_cast<int>(floorf(x)));
                atomicMin(&box[1], static_cast<int>(floorf(y)));
                atomicMax(&box[2], static_cast<int>(ceilf(x)));
                atomicMax(&box[3], static_cast<int>(ceilf(y)));
            } else {
                atomicMin(&box[0], -0x7fffffff - 1);  // poisons the stage: it gets no class
            }
        }
        sb[p] = make_uint4(w0, w1, static_cast<uint32_t>(i), 0u);
        sx[p] = make_float4(x, y, __int_as_float(i), 0.0f);
        uint4* o = reinterpret_cast<uint4*>(xd + static_cast<size_t>(p) * 64);
#pragma unroll
        for (int quarter = 0; quarter < 4; ++quarter) {
            const uint32_t b16 = static_cast<uint32_t>(bits >> (16 * quarter)) & 0xffffu;
            uint4 e;
            // four bits -> four bytes: the shifted copies of the nibble (by 0, 7, 14, 21) do not overlap
            e.x = ((b16 & 0xfu) * 0x00204081u) & 0x01010101u;
            e.y = (((b16 >> 4) & 0xfu) * 0x00204081u) & 0x01010101u;
            e.z = (((b16 >> 8) & 0xfu) * 0x00204081u) & 0x01010101u;
            e.w = (((b16 >> 12) & 0xfu) * 0x00204081u) & 0x01010101u;
            if (quarter == 3) {  // bits 62, 63 are never set (MM_MAX_BITS): bytes 62, 63 carry |t| and 1
                e.w = (e.w & 0x0000ffffu) | (static_cast<uint32_t>(__popcll(bits)) << 16) | 0x01000000u;
            }
            o[quarter] = e;
        }
    };
    if (cached) {
        uint32_t w0[CK], w1[CK];
#pragma unroll
        for (int j = 0; j < CK; ++j) {  // the descriptor words of all the thread's keypoints first
            const int i = tid + 1024 * j;
            w0[j] = i < n ? rows[static_cast<size_t>(i) * stride] : 0u;
            w1[j] = (i < n && stride > 1) ? rows[static_cast<size_t>(i) * stride + 1] : 0u;
        }
#pragma unroll
        for (int j = 0; j < CK; ++j) {
            const int i = tid + 1024 * j;
            if (i < n) {
                place(i, use_kp ? kx[j] : 0.0f, use_kp ? ky[j] : 0.0f, w0[j], w1[j]);
            }
        }
    } else {
        for (int i = tid; i < n; i += 1024) {
            place(i, use_kp ? kps[i].x : 0.0f, use_kp ? kps[i].y : 0.0f, rows[static_cast<size_t>(i) * stride],
                  stride > 1 ? rows[static_cast<size_t>(i) * stride + 1] : 0u);
        }
    }
    // padding up to a whole stage: distance 127 + |q|, which no bound below 127 lets through (and the slow path
    // checks the index anyway)
    for (int p = n + tid; p < nst * SP_STAGE; p += 1024) {
        uint4* o = reinterpret_cast<uint4*>(xd + static_cast<size_t>(p) * 64);
        o[0] = o[1] = o[2] = make_uint4(0u, 0u, 0u, 0u);
        o[3] = make_uint4(0u, 0u, 0u, 0x017f0000u);
        sb[p] = make_uint4(0u, 0u, 0xffffffffu, 0u);
        sx[p] = make_float4(0.0f, 0.0f, __int_as_float(-1), 0.0f);
    }
    __syncthreads();
    for (int s2 = tid; s2 < nst; s2 += 1024) {
        const int* box = stage_box[s2];
        const bool good = use_kp && box[0] != -0x7fffffff - 1 && box[0] <= box[2];
        sc[s2] = good ? make_float4(static_cast<float>(box[0]), static_cast<float>(box[1]), static_cast<float>(box[2]),
                                    static_cast<float>(box[3]))
                      : make_float4(__builtin_inff(), 0.0f, 0.0f, 0.0f);
    }
}

// ONE wave per workgroup: the waves share nothing, and a workgroup of four would hold its slot until the slowest of
// them is done (measured: 25 % fewer wave-cycles than the staged scan, but no less time, at four per workgroup).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void match_spatial_kernel(MatchView mv,
                                                                                                    int first_query_frame,
                                                                                                    int first_train_frame,
                                                                                                    int npairs, int qblocks) {
    constexpr int QW = 16 * MM_QT;                      // queries per wave
    __shared__ uint4 qtab[QW];                          // per query: compact bits (x, y), position (z, w as float bits)
    __shared__ unsigned long long keys[QW][2];          // per query: best_key, second_key
    __shared__ uint32_t fifo[SP_FIFO];                  // lane | tile (sorted position / 16) << 6 | class << 30
    __shared__ uint32_t pairq[SP_PAIRS];                // query | class << 8 | sorted train index << 16
    __shared__ uint16_t order[SP_MAX_STAGES + 3];       // the stages, nearest first: stage | class << 14
    __shared__ int bucket[SP_BUCKETS];
    // (A resident set of workgroups looping over the work items instead of one workgroup per item was tried: the outer
    // loop cost 33 registers and spilled scalars into the scan -- 4.3 vs 2.2 ms at 10k keypoints.)
    const int lane = threadIdx.x;
    SP_CLK(clk_begin);
    int pair, qblock, split;
    {
        const int id = static_cast<int>(blockIdx.x);
        const int per_pair = qblocks * mv.nsplit;
        if (npairs >= 8) {  // an XCD takes its pairs one after the other: a pair's sorted tables stay in its L2
            const int k = id >> 3;
            pair = (id & 7) + 8 * (k / per_pair);
            const int in_pair = k - (k / per_pair) * per_pair;
            qblock = in_pair % qblocks;
            split = in_pair / qblocks;
            if (pair >= npairs) {
                return;
            }
        } else {
            pair = id % npairs;
            const int rest = id / npairs;
            qblock = rest % qblocks;
            split = rest / qblocks;
        }
    }
    const int qf = first_query_frame + pair;
    const int tf = qf == 0 ? first_train_frame : qf - 1;
    const int n1 = mv.n_kp[qf];
    const int n2 = mv.n_kp[tf];
    const int wave_first = qblock * QW;  // sorted position of the wave's first query
    if (wave_first >= n1 || mv.mfma_flag[tf] == 0) {
        return;  // no queries here, or a pair of match_kernel's
    }
    const bool use_kp = mv.kps != nullptr;
    const int c = lane & 15;
    const int g = lane >> 4;
    const uint8_t* xq = mv.xdesc + static_cast<size_t>(qf) * mv.maxKp * 64;
    const uint8_t* xtr = mv.xdesc + static_cast<size_t>(tf) * mv.maxKp * 64;
    const uint4* qbits = mv.sbits + static_cast<size_t>(qf) * mv.maxKp;
    const uint4* tbits = mv.sbits + static_cast<size_t>(tf) * mv.maxKp;
    const float4* qxyi = mv.sxyi + static_cast<size_t>(qf) * mv.maxKp;
    const float4* txyi = mv.sxyi + static_cast<size_t>(tf) * mv.maxKp;
    // ---- the wave's queries
    int qorig = -1;
    float qx = 0.0f, qy = 0.0f;
    {
        const int i = wave_first + lane;
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (i < n1) {
            const uint4 b = qbits[i];
            const float4 r = qxyi[i];
            q = make_uint4(b.x, b.y, __float_as_uint(r.x), __float_as_uint(r.y));
            qorig = __float_as_int(r.z);
            qx = r.x;
            qy = r.y;
        }
        qtab[lane] = q;
        keys[lane][0] = MM_KEY_NONE;
        keys[lane][1] = MM_KEY_NONE;
    }
    constexpr int NEVER = 0x3fffffff;  // accumulator input of an absent query: distance + NEVER is never negative
    v4i B[MM_QT];
    v4i C[MM_QT];       // -bound of the lane's query of each tile (for the class of the current stage), in all four rows
    uint32_t bnd[MM_QT];  // the bound itself: second + [best == second] (0x7fffffff while fewer than two pairs were seen)
#pragma unroll
    for (int u = 0; u < MM_QT; ++u) {
        const int i = wave_first + u * 16 + c;
        const bool ok = i < n1;
        B[u] = v4i{0, 0, 0, 0};
        bnd[u] = ok ? 0x7fffffffu : 0u;  // an absent query has bound 0: nothing is below it
        if (ok) {
            const uint4 t = *reinterpret_cast<const uint4*>(xq + static_cast<size_t>(i) * 64 + 16 * g);
            // train form -> query form: bits * -2 (= * 254 per byte, no carries), and |q| / 1 swapped
            uint32_t w3 = t.w * 254u;
            if (g == 3) {
                w3 = ((t.w & 0x0101u) * 254u) | 0x00010000u | ((t.w >> 16 & 0xffu) << 24);
            }
            B[u] = v4i{static_cast<int>(t.x * 254u), static_cast<int>(t.y * 254u), static_cast<int>(t.z * 254u),
                       static_cast<int>(w3)};
        }
    }
    auto load_C = [&](bool far) {  // the MFMA's accumulator input for stages of the given kind
#pragma unroll
        for (int u = 0; u < MM_QT; ++u) {
            const uint32_t b = far ? (bnd[u] + 1u) >> 1 : bnd[u];  // far: 2 * hamming < bound <=> hamming < ceil(bound / 2)
            const int nb = bnd[u] == 0u ? NEVER : -static_cast<int>(b);
            C[u] = v4i{nb, nb, nb, nb};
        }
    };
    SP_CLK(clk_setup);
    // ---- the order of the stages (nearest box centre first: a counting sort on the distance in 32-px steps) and
    //      their classes
    const int nst = (n2 + SP_STAGE - 1) / SP_STAGE;
    int n_not_far = nst;  // stages [0, n_not_far) of the order are near or mixed, the rest far
    if (use_kp) {
        // box of the wave's queries; a coordinate that is not finite (or beyond +-1e6) switches the classes off
        const bool mine = qorig >= 0;
        float wx0 = mine ? floorf(qx) : 3.0e38f, wy0 = mine ? floorf(qy) : 3.0e38f;
        float wx1 = mine ? ceilf(qx) : -3.0e38f, wy1 = mine ? ceilf(qy) : -3.0e38f;
        int bad = (mine && !(coord_ok(qx) && coord_ok(qy))) ? 1 : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            wx0 = fminf(wx0, __shfl_xor(wx0, d));
            wy0 = fminf(wy0, __shfl_xor(wy0, d));
            wx1 = fmaxf(wx1, __shfl_xor(wx1, d));
            wy1 = fmaxf(wy1, __shfl_xor(wy1, d));
            bad |= __shfl_xor(bad, d);
        }
        const float cxq = 0.5f * (wx0 + wx1), cyq = 0.5f * (wy0 + wy1);
        const float4* boxes = mv.sbox + static_cast<size_t>(tf) * (mv.maxKp / 64);
        for (int b = lane; b < SP_BUCKETS; b += 64) {
            bucket[b] = 0;
        }
        wave_lds_fence();
        // bucket = the distance of the box centres in 64-px steps; the far stages (every pair >= 501 px apart: by
        // construction the most distant ones) in a second group of buckets behind all the others, so that the
        // filter's bound changes kind ONCE per scan
        auto classify = [&](int s, uint32_t* cls) {
            const float4 bx = boxes[s];
            *cls = 0u;
            if (bad != 0 || !(bx.x < 3.0e38f)) {
                return SP_BUCKETS / 2 - 1;  // no usable box: mixed, visited last among the stages that are not far
            }
            const float gx = fmaxf(fmaxf(bx.x - wx1, wx0 - bx.z), 0.0f), gy = fmaxf(fmaxf(bx.y - wy1, wy0 - bx.w), 0.0f);
            const float fx = fmaxf(bx.z - wx0, wx1 - bx.x), fy = fmaxf(bx.w - wy0, wy1 - bx.y);
            if (fx * fx + fy * fy <= 499.0f * 499.0f) {
                *cls = SP_NEAR;
            } else if (gx * gx + gy * gy >= 501.0f * 501.0f) {
                *cls = SP_FAR;
            }
            const float dx = 0.5f * (bx.x + bx.z) - cxq, dy = 0.5f * (bx.y + bx.w) - cyq;
            const int d = min(SP_BUCKETS / 2 - 1, static_cast<int>(__builtin_amdgcn_sqrtf(dx * dx + dy * dy)) >> 6);
            return *cls == SP_FAR ? SP_BUCKETS / 2 + d : d;
        };
        for (int s = lane; s < nst; s += 64) {
            uint32_t cls;
            atomicAdd(&bucket[classify(s, &cls)], 1);
        }
        wave_lds_fence();
        {  // exclusive prefix over 128 buckets: two per lane
            const int a = bucket[2 * lane], b = bucket[2 * lane + 1];
            int inc = a + b;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(inc, d);
                if (lane >= d) {
                    inc += o;
                }
            }
            wave_lds_fence();
            bucket[2 * lane] = inc - a - b;
            bucket[2 * lane + 1] = inc - b;
        }
        wave_lds_fence();
        n_not_far = __builtin_amdgcn_readfirstlane(bucket[SP_BUCKETS / 2]);
        wave_lds_fence();
        for (int s = lane; s < nst; s += 64) {
            uint32_t cls;
            const int bk = classify(s, &cls);
            order[atomicAdd(&bucket[bk], 1)] = static_cast<uint16_t>(static_cast<uint32_t>(s) | (cls << 14));
        }
    } else {
        for (int s = lane; s < nst; s += 64) {
            order[s] = static_cast<uint16_t>(static_cast<uint32_t>(s) | (SP_NEAR << 14));  // no keypoints: no penalty anywhere
        }
    }
    wave_lds_fence();
    SP_CLK(clk_order);
#ifdef SLAMX_MATCH_CLOCKS
    unsigned long long clk_slow = 0;
#endif
    int tail = 0;   // tile entries waiting (wave-uniform, kept in an SGPR)
    int ptail = 0;  // pair entries waiting
#ifdef SLAMX_MATCH_STATS
    unsigned long long sp_stat_1 = 0, sp_stat_2 = 0, sp_stat_3 = 0, sp_stat_5 = 0, sp_stat_6 = 0, sp_stat_7 = 0;
#endif
    // level 2: `count` (<= 64) pairs from the head of the pair list, one per lane: the Hamming distance again, the
    // reference's penalty arithmetic (feature_matcher.cpp:161-170) unless the pair's stage is `near`, and the update
    // of the query's two keys
    auto settle = [&](int count) {
        SP_STAT(7, 1);
        if (lane < count) {
            const uint32_t e = pairq[lane];
            const int q = static_cast<int>(e & 63u);
            const int t = static_cast<int>(e >> 16);
            const uint4 tb = tbits[t];
            const float4 tr = txyi[t];  // x, y, original index of the train keypoint
            const uint4 qv = qtab[q];
            unsigned long long* key = keys[q];
            const int sec = static_cast<int>(key[1] >> 32);  // may have shrunk since the pair was queued
            int dist = __popc(qv.x ^ tb.x) + __popc(qv.y ^ tb.y);
            if (dist <= sec && t < n2) {  // (a padding row of the last stage gets here while a query has no bound yet)
                if ((e & (SP_NEAR << 8)) == 0u) {
                    dist = penalised_distance(dist, __uint_as_float(qv.z), __uint_as_float(qv.w), tr.x, tr.y);
                }
                SP_STAT(3, __popcll(__ballot(dist <= sec)));
                if (dist <= sec) {
                    const unsigned long long k = (static_cast<unsigned long long>(static_cast<uint32_t>(dist)) << 32) |
                                                 static_cast<uint32_t>(__float_as_int(tr.z));
                    const unsigned long long old = atomicMin(&key[0], k);
                    atomicMin(&key[1], old > k ? old : k);
                }
            }
        }
        // move what is left to the front (reads of a step before its writes: one wave, LDS in program order)
        for (int base = count; base < ptail; base += 64) {
            const uint32_t v = pairq[base + lane];
            if (base + lane < ptail) {
                pairq[base - count + lane] = v;
            }
        }
        ptail -= count;
    };
    // level 1: `count` (<= 64) entries from the head of the tile list, one per lane: entry = (lane L, tile) stands
    // for the 16 pairs {query u * 16 + (L & 15)} x {train 16 tile + 4 (L >> 4) + r}.  Their distances are recomputed
    // with xor + popcount, "below the query's bound" collected as one bit per pair (the sign of distance - bound,
    // shifted in), and the lanes then queue their set bits one per round.
    auto work = [&](int count) {
        SP_STAT(1, count);
        const bool valid = lane < count;
        const uint32_t e = fifo[lane];
        for (int base = count; base < tail; base += 64) {  // what remains is moved to the front
            const uint32_t v = fifo[base + lane];
            if (base + lane < tail) {
                fifo[base - count + lane] = v;
            }
        }
        tail -= count;
        const int ec = static_cast<int>(e & 15u);
        const uint32_t cls = e >> 30;
        const int trow = valid ? static_cast<int>(e >> 6 & 0x1fffu) * 16 + 4 * static_cast<int>(e >> 4 & 3u) : 0;
        // everything is requested before anything is used
        uint4 tb[4];    // compact bits, original index
        int rowpen[4];  // rows past the end of the train set (the last stage's padding) start their popcount at 2^20
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            tb[r] = tbits[trow + r];  // (padding positions exist up to a whole stage: no range check on the address)
            rowpen[r] = trow + r < n2 ? 0 : (1 << 20);
        }
        uint32_t bits = 0u;  // bit 15 - (4 u + r): pair (u, r) is below the bound
#pragma unroll
        for (int u = 0; u < MM_QT; ++u) {
            const int q = u * 16 + ec;
            const uint2 qb = *reinterpret_cast<const uint2*>(&qtab[q]);
            const unsigned long long kbest = keys[q][0];
            const uint32_t bestd = static_cast<uint32_t>(kbest >> 32);
            const uint32_t sec = static_cast<uint32_t>(keys[q][1] >> 32);
            // A pair that merely TIES with `second` changes an output only if it also ties with `best` AND has the
            // lower original index (then it is the new best); with best < second it could at most replace the second
            // key by an equal distance.  Sets with the same corner at several pyramid levels sit in that tie state
            // for most queries: without the index test every further tie went to the pair list (87 pairs per query).
            const bool tie = bestd == sec && sec != 0x7fffffffu;
            const uint32_t below = tie ? static_cast<uint32_t>(kbest) : 0u;  // original indices below this one may tie
            uint32_t b0 = sec, b1 = sec + (tie ? 1u : 0u);
            if (cls == SP_FAR) {
                b0 = (b0 + 1u) >> 1;
                b1 = (b1 + 1u) >> 1;
            }
            b0 = valid ? b0 : 0u;  // lanes without an entry queue nothing
            b1 = valid ? b1 : 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int dist = __popc(qb.x ^ tb[r].x) + (__popc(qb.y ^ tb[r].y) + rowpen[r]);
                const uint32_t bound = tb[r].z < below ? b1 : b0;
                bits = __builtin_amdgcn_alignbit(bits, static_cast<uint32_t>(dist - static_cast<int>(bound)), 31);
            }
        }
        bits &= 0xffffu;
        const uint32_t tag = (cls << 8) | static_cast<uint32_t>(ec);
        while (true) {
            const bool has = bits != 0u;
            const unsigned long long pm = __ballot(has);
            if (pm == 0) {
                break;
            }
            SP_STAT(5, 1);
            const int k = 15 - (31 - __clz(static_cast<int>(bits)));  // highest set bit first: pair index 4 u + r
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(pm >> 32),
                                                            __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(pm), 0u));
            if (has) {
                pairq[ptail + static_cast<int>(rank)] = tag | (static_cast<uint32_t>(k >> 2) << 4) |
                                                        (static_cast<uint32_t>(trow + (k & 3)) << 16);
                bits &= ~(0x8000u >> k);
            }
            ptail += __popcll(pm);
            SP_STAT(2, __popcll(pm));
            if (ptail >= 64) {
                settle(64);
            }
        }
    };
    bool far_now = false;  // the kind of stage C[] is loaded for
    auto refresh_bounds = [&]() {
#pragma unroll
        for (int u = 0; u < MM_QT; ++u) {
            if (bnd[u] != 0u) {
                const uint32_t bestd = static_cast<uint32_t>(keys[u * 16 + c][0] >> 32);
                const uint32_t sec = static_cast<uint32_t>(keys[u * 16 + c][1] >> 32);
                bnd[u] = sec + ((bestd == sec && sec != 0x7fffffffu) ? 1u : 0u);
                bnd[u] = bnd[u] == 0u ? 1u : bnd[u];  // (cannot happen: second == 0 implies best == 0)
            }
        }
        load_C(far_now);
    };
    load_C(false);
    // this workgroup's share of the stages: every nsplit-th of the ordered list (merged by emit_kernel)
    const int nmine = (nst - split + mv.nsplit - 1) / mv.nsplit;
    const int nmine_not_far = min(nmine, max(0, (n_not_far - split + mv.nsplit - 1) / mv.nsplit));
    auto stage_at = [&](int k) {  // k-th stage of this workgroup's share: stage | class << 14 (wave-uniform)
        return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(order[min(split + k * mv.nsplit, nst - 1)])));
    };
    auto tile_ptr = [&](uint32_t sc) {  // this lane's 16 bytes of the stage's first tile
        return xtr + (static_cast<size_t>(sc & 0x3fffu) * SP_STAGE + c) * 64 + 16 * g;
    };
    // one tile: distance - bound of the lane's 4 x 4 pairs, and a list entry when any of them is negative
    auto filter_tile = [&](const v4i& A, uint32_t tile, uint32_t cls) {
        v4i D[MM_QT];
#pragma unroll
        for (int u = 0; u < MM_QT; ++u) {
            D[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B[u], C[u], 0, 0, 0);
        }
        int any = 0;
#pragma unroll
        for (int u = 0; u < MM_QT; ++u) {
            any |= D[u][0] | D[u][1] | D[u][2] | D[u][3];
        }
        const unsigned long long pm = __ballot(any < 0);  // lanes with a pair below its (stale) bound
        if (pm != 0) {  // wave-uniform: far stages rarely have one
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(pm >> 32),
                                                            __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(pm), 0u));
            if (any < 0) {
                fifo[tail + static_cast<int>(rank)] = static_cast<uint32_t>(lane) | (tile << 6) | (cls << 30);
            }
            tail += __popcll(pm);
        }
    };
    auto slow_path = [&](int need) {  // works the list off while it holds at least `need` entries
        if (tail >= need) {
            SP_CLK(c0);
            do {
                work(min(64, tail));
            } while (tail >= need);
            refresh_bounds();
#ifdef SLAMX_MATCH_CLOCKS
            clk_slow += __builtin_amdgcn_s_memtime() - c0;
#endif
        }
    };
    // The scan: the first SP_STARTUP_TILES tiles one by one (operand straight from memory: the bound moves with every
    // tile there), then stage by stage -- a tile's operand registers are refilled with the NEXT stage's tile as soon
    // as its MFMAs are issued, so four tile loads are always in flight -- first the stages that are not far, then,
    // with the halved bound, the far ones; then the list drains.
    // ---- seed: the leading NEAR stages of the share (at most SP_SEED_STAGES: the 128 nearest trains), a query per
    //      lane against every train of the stage, both keys in registers.  No penalty in a near stage, so a distance is
    //      at most 64 and (distance << 16 | original index) orders like the 64-bit key: best = min, second = median of
    //      (best, second, new) -- 7 vector instructions per train (the rows come through scalar loads), where the same
    //      trains through the filter cost three times that: nothing is below a bound that does not exist yet, so every
    //      pair went through both lists.  A padding row carries original index 0xffffffff: its key is "none" by itself.
    int k_seed = 0;
    {
        uint32_t kb = 0xffffffffu, ks = 0xffffffffu;
        const uint2 qown = *reinterpret_cast<const uint2*>(&qtab[lane]);
        while (k_seed < min(SP_SEED_STAGES, nmine)) {
            const uint32_t sc = stage_at(k_seed);
            if ((sc >> 14) != SP_NEAR) {
                break;
            }
            const uint4* rows = tbits + static_cast<size_t>(sc & 0x3fffu) * SP_STAGE;
            for (int j0 = 0; j0 < SP_STAGE; j0 += 8) {
                uint4 tb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    tb[u] = rows[j0 + u];  // (wave-uniform address)
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t dist = static_cast<uint32_t>(__popc(qown.x ^ tb[u].x) + __popc(qown.y ^ tb[u].y));
                    const uint32_t k = (dist << 16) | tb[u].z;
                    const uint32_t lo = min(kb, k);
                    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(ks) : "v"(kb), "v"(ks), "v"(k));
                    kb = lo;
                }
            }
            ++k_seed;
        }
        if (k_seed > 0) {
            if (qorig >= 0) {
                auto wide = [](uint32_t k) {
                    return k == 0xffffffffu ? MM_KEY_NONE
                                            : (static_cast<unsigned long long>(k >> 16) << 32) | static_cast<unsigned long long>(k & 0xffffu);
                };
                keys[lane][0] = wide(kb);
                keys[lane][1] = wide(ks);
            }
            wave_lds_fence();
            refresh_bounds();
        }
    }
    // without a seed (the nearest stage is not near: sparse or spread-out sets) the first SP_STARTUP_TILES tiles go
    // through the filter one by one
    const int ntiles = nmine * 4;
    const int startup = k_seed > 0 ? 0 : min(SP_STARTUP_TILES, ntiles);
    const int k_steady = k_seed > 0 ? k_seed : startup / 4;  // first stage of the steady part (SP_STARTUP_TILES is a multiple of 4)
    for (int it = 0; it < startup; ++it) {
        const uint32_t sc = stage_at(it >> 2);
        const bool far = (sc >> 14) == SP_FAR;
        if (far != far_now) {
            far_now = far;
            load_C(far);
        }
        const v4i A1 = *reinterpret_cast<const v4i*>(tile_ptr(sc) + (it & 3) * 16 * 64);
        filter_tile(A1, (sc & 0x3fffu) * 4u + static_cast<uint32_t>(it & 3), sc >> 14);
        slow_path(64);
    }
    SP_CLK(clk_startup);
#ifdef SLAMX_MATCH_CLOCKS
    const unsigned long long clk_slow_startup = clk_slow;
#endif
    v4i A[4];
    if (k_steady < nmine) {
        const uint8_t* p0 = tile_ptr(stage_at(k_steady));
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            A[tl] = *reinterpret_cast<const v4i*>(p0 + tl * 16 * 64);
        }
    }
    auto steady = [&](int k0, int k1) {
        for (int k = k0; k < k1; ++k) {
            const uint32_t s0 = stage_at(k);
            const uint8_t* next = tile_ptr(stage_at(min(k + 1, nmine - 1)));  // (the last stage is simply loaded again)
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) {
                filter_tile(A[tl], (s0 & 0x3fffu) * 4u + tl, s0 >> 14);
                A[tl] = *reinterpret_cast<const v4i*>(next + tl * 16 * 64);
            }
            slow_path(64);
        }
    };
    const int k_far = max(k_steady, nmine_not_far);
    if (k_steady < k_far) {
        if (far_now) {
            far_now = false;
            load_C(false);
        }
        steady(k_steady, k_far);
    }
    if (k_far < nmine) {
        if (!far_now) {
            far_now = true;
            load_C(true);
        }
        steady(k_far, nmine);
    }
    SP_CLK(clk_steady);
    slow_path(1);
    while (ptail > 0) {
        settle(min(64, ptail));
    }
    wave_lds_fence();
#ifdef SLAMX_MATCH_CLOCKS
    if (lane == 0) {
        const unsigned long long clk_end = __builtin_amdgcn_s_memtime();
        atomicAdd(&sp_clocks[0], 1ull);
        atomicAdd(&sp_clocks[1], clk_setup - clk_begin);
        atomicAdd(&sp_clocks[2], clk_order - clk_setup);
        atomicAdd(&sp_clocks[3], clk_startup - clk_order);
        atomicAdd(&sp_clocks[4], clk_steady - clk_startup);
        atomicAdd(&sp_clocks[5], clk_slow - clk_slow_startup);
        atomicAdd(&sp_clocks[6], clk_end - clk_steady);
        atomicAdd(&sp_clocks[7], clk_slow_startup);
    }
#endif
#ifdef SLAMX_MATCH_STATS
    if (lane == 0) {
        atomicAdd(&sp_stats[0], 1ull);
        atomicAdd(&sp_stats[1], sp_stat_1);
        atomicAdd(&sp_stats[2], sp_stat_2);
        atomicAdd(&sp_stats[3], sp_stat_3);
        atomicAdd(&sp_stats[4], static_cast<unsigned long long>(nmine));
        atomicAdd(&sp_stats[5], sp_stat_5);
        atomicAdd(&sp_stats[6], sp_stat_6);
        atomicAdd(&sp_stats[7], sp_stat_7);
    }
#endif
    if (qorig >= 0) {
        const unsigned long long kb = keys[lane][0], ks = keys[lane][1];
        const size_t slot = (static_cast<size_t>(qf) * mv.nsplit + split) * mv.maxK + qorig;
        mv.best[slot] = make_int2(static_cast<int>(kb >> 32), static_cast<int>(static_cast<uint32_t>(kb)));
        mv.second[slot] = static_cast<int>(ks >> 32);
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
                // merge the per-chunk summaries (best, idx, second) of disjoint parts of the train set: the best is the
                // lexicographically smallest (distance, train index) -- the reference's strict '<' in an ascending
                // scan (feature_matcher.cpp:132-141) -- and second = min(the losing best, both seconds).  Chunks of the
                // popcount scan are ascending index ranges; the spatial scan's are interleaved, hence the index.
                int best = 0x7fffffff;
                int second = 0x7fffffff;
                int best_idx = -1;
                for (int c = 0; c < mv.nsplit; ++c) {
                    const size_t slot = (static_cast<size_t>(qf) * mv.nsplit + c) * mv.maxK + i;
                    const int2 pb = mv.best[slot];
                    const int ps = mv.second[slot];
                    if (pb.x < best || (pb.x == best && static_cast<uint32_t>(pb.y) < static_cast<uint32_t>(best_idx))) {
                        second = min(second, best);
                        best = pb.x;
                        best_idx = pb.y;
                    } else {
                        second = min(second, pb.x);
                    }
                    second = min(second, ps);
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

#ifdef SLAMX_MATCH_CLOCKS
extern "C" void slamx_debug_match_clocks(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(sp_clocks), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(sp_clocks), z, sizeof(z));
}
#endif
#ifdef SLAMX_MATCH_STATS
extern "C" void slamx_debug_match_stats(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(sp_stats), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(sp_stats), z, sizeof(z));
}
#endif

void launch_match_consecutive(const MatchView& mv_in, int first_query_frame, int npairs, int first_train_frame,
                              hipStream_t stream, hipEvent_t ev_mid) {
    if (npairs <= 0) {
        return;
    }
    MatchView mv = mv_in;
    const int stride = mv.width / 4;
    const bool mfma = mv.xdesc != nullptr && mv.width % 4 == 0 && mv.live_bits > 0 && mv.live_bits <= MM_MAX_BITS &&
                      mv.maxK <= 65536;  // (pair entries hold the train index in 16 bits)
    if (!mfma) {
        mv.xdesc = nullptr;  // match_kernel then takes every pair
        mv.mfma_flag = nullptr;
    }
    if (mfma) {
        // The matrix-core scan of the pairs with large train sets; the popcount scan below takes the other pairs (the
        // keypoint counts are only known on the device: both kernels are launched and each leaves the other's pairs
        // alone).  First every frame that is a query or a train set of a matrix-core pair of this call is put into
        // spatial order and expanded (spatial_order_kernel, which also decides which pairs those are), then the scan.
        // frames in use: the queries, their predecessors, and first_train_frame when frame 0 is a query
        const int lo_frame = first_query_frame > 0 ? first_query_frame - 1 : 0;
        const int hi_frame = first_query_frame + npairs - 1;
        const bool extra = first_query_frame == 0 && first_train_frame >= 0 && (first_train_frame < lo_frame || first_train_frame > hi_frame);
        const int pair_slots = npairs >= 8 ? 8 * ((npairs + 7) / 8) : npairs;
        const int nfr = hi_frame - lo_frame + 1;
        hipLaunchKernelGGL(scan_choice_kernel, dim3((nfr + 15) / 16), dim3(1024), 0, stream, mv, lo_frame, nfr);
        if (extra) {
            hipLaunchKernelGGL(scan_choice_kernel, dim3(1), dim3(1024), 0, stream, mv, first_train_frame, 1);
        }
        hipLaunchKernelGGL(spatial_order_kernel, dim3(nfr), dim3(1024), 0, stream, mv, lo_frame, first_train_frame,
                           first_query_frame, hi_frame);
        if (extra) {
            hipLaunchKernelGGL(spatial_order_kernel, dim3(1), dim3(1024), 0, stream, mv, first_train_frame, first_train_frame,
                               0x7fffffff, -1);  // a train set only
        }
        const int wblocks = (mv.maxK + 16 * MM_QT - 1) / (16 * MM_QT);  // one wave of 64 queries per workgroup
        dim3 wgrid(static_cast<unsigned>(wblocks) * pair_slots * mv.nsplit);
        hipLaunchKernelGGL(match_spatial_kernel, wgrid, dim3(64), 0, stream, mv, first_query_frame, first_train_frame, npairs,
                           wblocks);
    }
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
