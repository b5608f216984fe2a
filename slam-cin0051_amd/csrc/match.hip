// match.hip -- brute-force Hamming 1-NN with second-best tracking, jump-radius penalty and
// Lowe ratio test.  Two all-pairs scans with identical results: match_kernel (xor + popcount, below) and
// match_mfma_kernel (int8 matrix cores, further down; taken per frame pair when the train set has >= mfma_min_train = 3072 keypoints).
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
    if (mv.xdesc != nullptr && n2 >= mv.mfma_min_train) {
        return;  // this pair belongs to match_mfma_kernel (uniform)
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
//     train form  t' = [ bit 0 .. bit 61 |  |t| |  1  ]        (xdesc_kernel writes this form)
//     query form  q' = [ -2 bit 0 .. -2 bit 61 |  1  | |q| ]   (derived from the train form at load)
// the dot product q'.t' IS the Hamming distance, and one v_mfma_i32_16x16x64_i8 (K = 64) gives the
// 256 distances of 16 train x 16 query descriptors in 16 cycles: 16 pairs per cycle and SIMD against 4
// for xor + popcount at its peak.  Integer arithmetic, so the values are the reference's
// (calculateHammingDistance, common.hpp:40-50) bit for bit; both operands use the same byte -> k
// assignment, so the k order inside the instruction does not matter.
//
// What stays on the vector ALU is the per-query bookkeeping, and the matrix cores leave it 4
// instructions per MFMA before it becomes the bound (counters: the first versions of this kernel
// spent 16 and ran no faster than the popcount kernel).  C/D layout: lane (c = lane & 15, g = lane >> 4)
// receives rows 4g..4g+3 of column c: four consecutive TRAIN descriptors against ONE query.  So:
//   * the accumulator input of the MFMA is -bound(query), the stale upper bound of the query's `second`
//     that match_kernel filters with: the product comes out as distance - bound, a pair passes iff it
//     is negative, and a lane knows whether ANY of its 16 pairs of a train tile (4 query tiles x 4
//     rows) passes from the sign of their OR: 8 v_or3 + 1 compare per 4 MFMAs;
//   * a passing (lane, tile) -- a few per cent of them -- is appended to a per-WAVE work list in LDS, and
//     whenever 64 entries wait the whole wave works them off, one entry per lane whoever pushed it:
//     xor + popcount of the 16 pairs from the compact bits (read from global memory: entries carry train
//     indices, not positions in the LDS stage, so nothing is flushed half-full when a stage is replaced);
//     the one or two of them that can still change the query's keys go to a second list of PAIRS, which
//     is settled 64 at a time as well:
//     the reference's penalty arithmetic (with every pair slot of the first level taking it, at one
//     active lane in 64, it was two thirds of the kernel's instructions) and the update of the
//     query's state, which lives in LDS as two 64-bit keys (distance << 32 | train index):
//         old = atomicMin(best_key, key);  atomicMin(second_key, max(old, key))
//     Every key but the final minimum is offered to second_key exactly once (when it loses), so the two
//     keys end as the two lexicographically smallest of all pairs, whatever the order of the updates:
//     the smallest distance with the LOWEST train index -- the reference's strict '<' in an ascending
//     scan (feature_matcher.cpp:132-141) -- and the second smallest distance of the multiset.
// The slow path thus runs on full waves, needs no per-lane state and leaves no merge for the end.
// Filtering with strict '<' is exact: when the filter of a train tile runs, the keys hold only pairs of
// EARLIER tiles (lower indices), so a pair that merely ties with `second` cannot be one of the two
// smallest keys.  Behind the filter, where entries of different tiles are worked off together and two
// lanes may update one query at the same time, the index part of the keys decides: a tie with
// `second` is let through when it also ties with `best` (it may carry the lower index), and dropped
// otherwise (it could only replace the second key by one of the same distance).
constexpr int MM_STAGE = 256;    // train descriptors staged in LDS at a time
constexpr int MM_XSTRIDE = 80;   // bytes between staged expanded descriptors: 16 lanes x 16 B then cover all 64 banks
constexpr int MM_QT = 4;         // query tiles (of 16) per wave: 64 queries per wave, 256 per workgroup
// The two work lists are plain arrays filled from the front: the OLDEST 64 entries are taken (an entry that
// waits keeps its query's bound loose) and the few that remain are moved down -- on the slow path, so that the
// always-executed push needs no ring arithmetic.
constexpr int MM_FIFO = 128;     // tile entries per wave: < 64 waiting + the 64 pushes of one train tile
constexpr int MM_TRASH = MM_FIFO;  // one more slot: where the lanes that do not pass write
constexpr int MM_PAIRS = 320;    // pair entries per wave: < 64 waiting + the 4 x 64 pushes of one query tile
constexpr int MM_MAX_BITS = 62;
constexpr unsigned long long MM_KEY_NONE = 0x7fffffffffffffffull;  // distance INT_MAX, index -1

typedef int v4i __attribute__((ext_vector_type(4)));

// desc (compact bits) -> train-form expanded descriptors, 64 bytes each; 4 lanes per keypoint
// A frame is only expanded when a pair that takes the matrix-core scan uses it: as a train set of at least
// min_train keypoints, or as the query frame of such a train set (its predecessor; first_train_frame for frame 0).
__global__ __launch_bounds__(256) void xdesc_kernel(const uint8_t* __restrict__ desc, const int* __restrict__ n_kp, int maxK,
                                                    int width, uint8_t* __restrict__ xdesc, int first_frame,
                                                    int first_train_frame, int min_train) {
    const int f = first_frame + blockIdx.y;
    const int n = n_kp[f];
    const int pred = f > 0 ? f - 1 : first_train_frame;
    if (n < min_train && (pred < 0 || n_kp[pred] < min_train)) {
        return;
    }
    const int stride = width / 4;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n * 4; e += gridDim.x * 256) {
        const int i = e >> 2;
        const int quarter = e & 3;
        const uint32_t* row = reinterpret_cast<const uint32_t*>(desc + (static_cast<size_t>(f) * maxK + i) * width);
        const uint32_t w0 = row[0];
        const uint32_t w1 = stride > 1 ? row[1] : 0u;
        const unsigned long long bits = static_cast<unsigned long long>(w0) | (static_cast<unsigned long long>(w1) << 32);
        const uint32_t b16 = static_cast<uint32_t>(bits >> (16 * quarter)) & 0xffffu;
        uint4 o;
        // four bits -> four bytes: the shifted copies of the nibble (by 0, 7, 14, 21) do not overlap
        o.x = ((b16 & 0xfu) * 0x00204081u) & 0x01010101u;
        o.y = (((b16 >> 4) & 0xfu) * 0x00204081u) & 0x01010101u;
        o.z = (((b16 >> 8) & 0xfu) * 0x00204081u) & 0x01010101u;
        o.w = (((b16 >> 12) & 0xfu) * 0x00204081u) & 0x01010101u;
        if (quarter == 3) {  // bits 62, 63 are never set (MM_MAX_BITS): bytes 62, 63 carry |t| and 1
            o.w = (o.w & 0x0000ffffu) | (static_cast<uint32_t>(__popcll(bits)) << 16) | 0x01000000u;
        }
        *reinterpret_cast<uint4*>(xdesc + (static_cast<size_t>(f) * maxK + i) * 64 + 16 * quarter) = o;
    }
}

// (amdgpu_waves_per_eu >= 2 caps the kernel at 256 registers, which lets the compiler give the MFMAs plain
// VGPR destinations; with the default cap of 512 it allocates accumulator registers and copies every result)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void match_mfma_kernel(MatchView mv,
                                                                                                  int first_query_frame,
                                                                                                  int first_train_frame,
                                                                                                  int stride, int npairs,
                                                                                                  int qblocks) {
    __shared__ __attribute__((aligned(16))) uint8_t xt[MM_STAGE * MM_XSTRIDE];
    __shared__ uint4 qtab[4][16 * MM_QT];                  // per wave and query: compact bits (x, y), position (z, w as float bits)
    __shared__ unsigned long long keys[4][16 * MM_QT][2];  // per wave and query: best_key, second_key
    __shared__ uint32_t fifo[4][MM_FIFO + 64];             // per wave: lane | first train index of the tile << 6
    __shared__ uint32_t pairq[4][MM_PAIRS + 64];           // per wave: query | raw distance << 8 | train index << 16
    constexpr int QW = 16 * MM_QT;                         // queries per wave
    // Workgroups are dealt to the 8 XCDs round-robin.  An XCD takes its pairs ONE AFTER THE OTHER (all query
    // blocks and train chunks of a pair on consecutive ids of that XCD): the workgroups resident on an XCD then
    // belong to ~3 pairs and their train descriptors (64 expanded bytes each, 0.7 MB per 10k-keypoint frame,
    // re-read by every query block) stay in its 4 MB L2.  With the pair index fastest -- the popcount kernel's
    // order, fine for its 8-byte descriptors -- 32 pairs were in flight per XCD and the staging loads went to
    // the Infinity Cache / HBM (7 GB per step at 10k keypoints).
    int pair, qblock, split;
    {
        const int id = static_cast<int>(blockIdx.x);
        const int per_pair = qblocks * mv.nsplit;
        if (npairs >= 8) {
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
    const int block_first = qblock * (QW * 4);
    if (block_first >= n1 || n2 < mv.mfma_min_train) {
        return;  // no queries here, or a pair of match_kernel's
    }
    const bool use_kp = mv.kps != nullptr;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int c = lane & 15;
    const int g = lane >> 4;
    const uint32_t* qd = reinterpret_cast<const uint32_t*>(mv.desc + static_cast<size_t>(qf) * mv.maxK * mv.width);
    const uint32_t* td = reinterpret_cast<const uint32_t*>(mv.desc + static_cast<size_t>(tf) * mv.maxK * mv.width);
    const uint8_t* xq = mv.xdesc + static_cast<size_t>(qf) * mv.maxK * 64;
    const uint8_t* xtr = mv.xdesc + static_cast<size_t>(tf) * mv.maxK * 64;
    const int wave_first = block_first + wave * QW;
    const bool wave_has_queries = wave_first < n1;
    for (int k = lane; k < QW; k += 64) {
        // query wave_first + k of the wave
        const int i = wave_first + k;
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (i < n1) {
            q.x = qd[static_cast<size_t>(i) * stride];
            q.y = stride > 1 ? qd[static_cast<size_t>(i) * stride + 1] : 0u;
            if (use_kp) {
                const slamx_keypoint kp = mv.kps[static_cast<size_t>(qf) * mv.maxK + i];
                q.z = __float_as_uint(kp.x);
                q.w = __float_as_uint(kp.y);
            }
        }
        qtab[wave][k] = q;
        keys[wave][k][0] = MM_KEY_NONE;
        keys[wave][k][1] = MM_KEY_NONE;
    }
    constexpr int NEVER = 0x3fffffff;  // accumulator input of an absent query: distance + NEVER is never negative
    v4i B[MM_QT];
    v4i C[MM_QT];  // -bound of the lane's query of each tile, in all four rows
#pragma unroll
    for (int u = 0; u < MM_QT; ++u) {
        const int i = wave_first + u * 16 + c;
        const bool ok = i < n1;
        B[u] = v4i{0, 0, 0, 0};
        const int nb = ok ? -0x7fffffff : NEVER;
        C[u] = v4i{nb, nb, nb, nb};
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
    uint32_t* const myfifo = fifo[wave];
    uint32_t* const mypairs = pairq[wave];
    int tail = 0;   // tile entries waiting (wave-uniform, kept in an SGPR)
    int ptail = 0;  // pair entries waiting
    // level 2: `count` (<= 64) pairs from the head of the pair FIFO, one per lane: the reference's penalty
    // arithmetic (feature_matcher.cpp:161-170) and the update of the query's two keys
    const slamx_keypoint* tkps = use_kp ? mv.kps + static_cast<size_t>(tf) * mv.maxK : nullptr;
    auto settle = [&](int count) {
        if (lane < count) {
            const uint32_t e = mypairs[lane];
            const int q = static_cast<int>(e & 0xffu);
            int dist = static_cast<int>(e >> 8 & 0xffu);
            const int t = static_cast<int>(e >> 16);
            unsigned long long* key = keys[wave][q];
            const int sec = static_cast<int>(key[1] >> 32);  // may have shrunk since the pair was queued
            if (dist <= sec) {
                if (use_kp) {
                    const uint4 qv = qtab[wave][q];
                    const float2 p = *reinterpret_cast<const float2*>(&tkps[t]);  // x, y lead the keypoint record
                    dist = penalised_distance(dist, __uint_as_float(qv.z), __uint_as_float(qv.w), p.x, p.y);
                }
                if (dist <= sec) {
                    const unsigned long long k = (static_cast<unsigned long long>(static_cast<uint32_t>(dist)) << 32) |
                                                 static_cast<uint32_t>(t);
                    const unsigned long long old = atomicMin(&key[0], k);
                    atomicMin(&key[1], old > k ? old : k);
                }
            }
        }
        // move what is left to the front (reads of a step before its writes: one wave, LDS in program order)
        for (int base = count; base < ptail; base += 64) {
            const uint32_t v = mypairs[base + lane];
            if (base + lane < ptail) {
                mypairs[base - count + lane] = v;
            }
        }
        ptail -= count;
    };
    // level 1: `count` (<= 64) entries from the head of the tile FIFO, one per lane: entry = (lane L, tile) stands
    // for the 16 pairs {query u * 16 + (L & 15)} x {train 16 tile + 4 (L >> 4) + r}; their distances are
    // recomputed with xor + popcount and the ones that still reach the query's `second` go to the pair FIFO
    auto work = [&](int count, int t_end) {
        const bool valid = lane < count;
        const uint32_t e = myfifo[lane];
        {
            const uint32_t v = myfifo[count + lane];  // at most 63 entries remain: moved to the front
            if (count + lane < tail) {
                myfifo[lane] = v;
            }
        }
        tail -= count;
        const int ec = static_cast<int>(e & 15u);
        const int trow = static_cast<int>(e >> 6) + 4 * static_cast<int>(e >> 4 & 3u);
        // the compact bits of the four trains come from global memory (L2): entries outlive the LDS stage they
        // were found in, so nothing has to be flushed when the stage is replaced
        uint2 tb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = valid ? min(trow + r, t_end - 1) : 0;
            tb[r] = make_uint2(td[static_cast<size_t>(t) * stride], stride > 1 ? td[static_cast<size_t>(t) * stride + 1] : 0u);
        }
        // a loop over the query tiles, not unrolled: `settle` (with the penalty arithmetic inside) is instantiated once
#pragma unroll 1
        for (int u = 0; u < MM_QT; ++u) {
            const int q = u * 16 + ec;
            const uint2 qb = *reinterpret_cast<const uint2*>(&qtab[wave][q]);
            const int bestd = static_cast<int>(keys[wave][q][0] >> 32);
            int sec = static_cast<int>(keys[wave][q][1] >> 32);
            // A pair that merely TIES with `second` changes an output only if it also ties with `best` and has
            // the lower index; with best < second it could at most replace the second key by an equal distance.
            const int limit = valid ? (bestd == sec ? sec : sec - 1) : -1;  // lanes without an entry queue nothing
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = trow + r;
                const int dist = __popc(qb.x ^ tb[r].x) + __popc(qb.y ^ tb[r].y);
                const bool pass = dist <= limit && t < t_end;
                const unsigned long long pm = __ballot(pass);
                if (pm != 0) {  // wave-uniform
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(pm >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(pm), 0u));
                    const uint32_t slot = pass ? static_cast<uint32_t>(ptail) + rank : static_cast<uint32_t>(MM_PAIRS);
                    mypairs[slot] = static_cast<uint32_t>(q) | (static_cast<uint32_t>(dist) << 8) | (static_cast<uint32_t>(t) << 16);
                    ptail += __popcll(pm);
                }
            }
            while (ptail >= 64) {
                settle(64);
            }
        }
    };
    auto refresh_bounds = [&]() {
#pragma unroll
        for (int u = 0; u < MM_QT; ++u) {
            const int sec = static_cast<int>(keys[wave][u * 16 + c][1] >> 32);
            const int nb = C[u][0] == NEVER ? NEVER : -sec;
            C[u] = v4i{nb, nb, nb, nb};
        }
    };
    const int chunk = (n2 + mv.nsplit - 1) / mv.nsplit;
    const int t_begin = split * chunk;
    const int t_end = min(n2, t_begin + chunk);
    // thread t stages train descriptor t0 + t: its 64 expanded bytes.  The loads of a stage are issued a whole
    // stage ahead (into 16 registers), so their latency passes under the tiles of the current one.
    auto fetch = [&](int t0, uint4* x) {
        const int t = threadIdx.x;
        if (t0 + t < t_end) {
            const uint4* src = reinterpret_cast<const uint4*>(xtr + static_cast<size_t>(t0 + t) * 64);
            x[0] = src[0];
            x[1] = src[1];
            x[2] = src[2];
            x[3] = src[3];
        } else {
            // past the end of the train set: distance 127 + |q|, which no bound below 127 lets through
            // (and the slow path checks the index anyway)
            x[0] = x[1] = x[2] = make_uint4(0u, 0u, 0u, 0u);
            x[3] = make_uint4(0u, 0u, 0u, 0x017f0000u);
        }
    };
    uint4 xnext[4];
    fetch(t_begin, xnext);
    for (int t0 = t_begin; t0 < t_end; t0 += MM_STAGE) {
        const int tn = min(MM_STAGE, t_end - t0);
        __syncthreads();  // every wave is done with the previous stage
        {
            uint4* dst = reinterpret_cast<uint4*>(xt + threadIdx.x * MM_XSTRIDE);
            dst[0] = xnext[0];
            dst[1] = xnext[1];
            dst[2] = xnext[2];
            dst[3] = xnext[3];
        }
        if (t0 + MM_STAGE < t_end) {
            fetch(t0 + MM_STAGE, xnext);
        }
        __syncthreads();
        if (!wave_has_queries) {
            continue;  // wave-uniform: this wave only helps staging
        }
        const int ntiles = (tn + 15) >> 4;
        const uint8_t* arow = xt + c * MM_XSTRIDE + 16 * g;
        v4i A = *reinterpret_cast<const v4i*>(arow);
        for (int tile = 0; tile < ntiles; ++tile) {
            // distance - bound of the 4 x 4 pairs this lane sees of the tile; the next tile's operand behind the products
            v4i D[MM_QT];
#pragma unroll
            for (int u = 0; u < MM_QT; ++u) {
                D[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B[u], C[u], 0, 0, 0);
            }
            A = *reinterpret_cast<const v4i*>(arow + min(tile + 1, ntiles - 1) * 16 * MM_XSTRIDE);
            int any = 0;
#pragma unroll
            for (int u = 0; u < MM_QT; ++u) {
                any |= D[u][0] | D[u][1] | D[u][2] | D[u][3];
            }
            const bool pass = any < 0;  // some pair is below its (stale) bound
            const unsigned long long pm = __ballot(pass);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(pm >> 32),
                                                            __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(pm), 0u));
            // branch-free: every lane stores, the ones that do not pass into the trash slot; the list's write position
            // is a scalar, so the address is one shift-add and one select
            typedef __attribute__((address_space(3))) uint32_t lds_u32;
            lds_u32* const wr = (lds_u32*)myfifo + tail;
            lds_u32* const dst = pass ? wr + rank : (lds_u32*)myfifo + MM_TRASH;
            *dst = static_cast<uint32_t>(lane) | (static_cast<uint32_t>(t0 + tile * 16) << 6);
            tail += __popcll(pm);
            if (tail >= 64) {  // wave-uniform
                do {
                    work(64, t_end);
                } while (tail >= 64);
                refresh_bounds();
            }
        }
    }
    if (wave_has_queries) {  // end of the scan: what is still waiting
        while (tail > 0) {
            work(min(64, tail), t_end);
        }
        while (ptail > 0) {
            settle(min(64, ptail));
        }
    }
    for (int k = lane; k < QW; k += 64) {
        const int i = wave_first + k;
        if (i < n1) {
            const unsigned long long kb = keys[wave][k][0], ks = keys[wave][k][1];
            const size_t slot = (static_cast<size_t>(qf) * mv.nsplit + split) * mv.maxK + i;
            mv.best[slot] = make_int2(static_cast<int>(kb >> 32), static_cast<int>(static_cast<uint32_t>(kb)));
            mv.second[slot] = static_cast<int>(ks >> 32);
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
    }
    if (mfma) {
        // Expanded descriptors of every frame that is a query or a train set of this call, then the matrix-core scan
        // of the pairs with large train sets; the popcount scan below takes the other pairs (the keypoint counts
        // are only known on the device: both kernels are launched and each leaves the other's pairs alone).
        const int qblocks = (mv.maxK + 64 * MM_QT - 1) / (64 * MM_QT);
        // frames in use: the queries, their predecessors, and first_train_frame when frame 0 is a query
        const int lo_frame = first_query_frame > 0 ? first_query_frame - 1 : 0;
        const int hi_frame = first_query_frame + npairs - 1;
        dim3 gx(grid_x(mv.maxK * 4, 256, hi_frame - lo_frame + 1), hi_frame - lo_frame + 1);
        hipLaunchKernelGGL(xdesc_kernel, gx, dim3(256), 0, stream, mv.desc, mv.n_kp, mv.maxK, mv.width, mv.xdesc, lo_frame,
                           first_train_frame, mv.mfma_min_train);
        if (first_query_frame == 0 && first_train_frame >= 0 && (first_train_frame < lo_frame || first_train_frame > hi_frame)) {
            dim3 g1(grid_x(mv.maxK * 4, 256, 1), 1);
            hipLaunchKernelGGL(xdesc_kernel, g1, dim3(256), 0, stream, mv.desc, mv.n_kp, mv.maxK, mv.width, mv.xdesc,
                               first_train_frame, first_train_frame, mv.mfma_min_train);
        }
        const int pair_slots = npairs >= 8 ? 8 * ((npairs + 7) / 8) : npairs;
        dim3 grid(static_cast<unsigned>(qblocks) * pair_slots * mv.nsplit);
        hipLaunchKernelGGL(match_mfma_kernel, grid, dim3(256), 0, stream, mv, first_query_frame, first_train_frame, stride,
                           npairs, qblocks);
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
