// order.hip -- the std::sort of applyNonMaxSuppression, bit-exact on the GPU (the greedy
// suppression that consumes its order is in nms.hip).
//
// Reference: src/frontend/feature_detector.cpp:147-188
//   :160-161  std::sort(keypoints, response descending)   -- UNSTABLE; with ~97 % of the
//             candidates tied on score the output order AND the kept set depend on libstdc++'s
//             exact introsort permutation (hard part 1);
//   :165-185  greedy suppression in that order, radius = SuppressionWindowSize.
//
// What std::sort does (libstdc++ bits/stl_algo.h, GCC 11):
//   __introsort_loop : while (n > 16) { if (depth == 0) heapsort; pivot = median of
//                      (first+1, mid, last-1) swapped into *first; cut = Hoare
//                      __unguarded_partition(first+1, last, *first); recurse right; loop left }
//   __final_insertion_sort over the whole range.
//
// Exact parallel restatement used here (proved in DESIGN.md, "sort"):
//   (1) The Hoare partition's scans only ever test ORIGINAL values: the k-th stop of the left
//       scan is the k-th element (from the left) with !(a > p) and the k-th stop of the right
//       scan is the k-th element (from the right) with !(p > a); pairs are swapped while
//       L_k < R_k and cut = min(L_{K+1}, R_K).  A wave therefore consumes the range from both
//       ends in 64-element chunks, finds the stoppers with ballots, pairs the k-th lowest left
//       stopper with the k-th highest right stopper with a select-kth on the ballot masks, and
//       finishes the crossing inside one chunk.  Same swaps, same cut as the sequential loop.
//   (2) Every partition leaves left >= pivot >= right, so leaf segments (<= 16 elements) are
//       totally ordered among themselves and the final insertion sort -- a stable sort --
//       never moves an element across a leaf boundary: it is a stable sort INSIDE each leaf,
//       done here by rank counting, one lane per element.
//   (3) Segments of one recursion level are independent: one wave per segment, level by level.
//   The depth-limit heapsort fallback is the libstdc++ heap code run by one lane.
//
// The greedy suppression is the lexicographically-first maximal independent set in sorted
// order; it is computed by monotone rounds (a candidate is KEPT once every earlier-ranked
// neighbour is SUPPRESSED, SUPPRESSED once one is KEPT), with neighbours enumerated from the
// FAST corner bitmap + its prefix sums (fast.hip) instead of a cell grid.
#include "slamx_internal.h"

namespace slamx {

namespace {

__device__ __forceinline__ uint32_t score_of(uint32_t key) { return key >> SCORE_SHIFT; }
// comp(a,b) of the reference's lambda: a.response > b.response
__device__ __forceinline__ bool comp(uint32_t a, uint32_t b) { return score_of(a) > score_of(b); }

// position of the k-th (0-based) set bit of m; k < popcount(m) required
__device__ __forceinline__ int select64(unsigned long long m, int k) {
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
// position of the k-th set bit counted from the top
__device__ __forceinline__ int select64_top(unsigned long long m, int k) { return 63 - select64(__brevll(m), k); }

__device__ __forceinline__ unsigned long long low_mask(int nbits) {  // bits [0, nbits)
    return nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
}

__device__ __forceinline__ void wave_mem_fence() {
    // same-wave accesses are issued in order; this only stops compiler reordering / caching
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------- libstdc++ heap code (bits/stl_heap.h), one lane ----------------
__device__ void push_heap_(uint32_t* first, int hole, int top, uint32_t value) {
    int parent = (hole - 1) / 2;
    while (hole > top && comp(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
__device__ void adjust_heap_(uint32_t* first, int hole, int len, uint32_t value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (comp(first[child], first[child - 1])) {
            child--;
        }
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    push_heap_(first, hole, top, value);
}
// std::__partial_sort(first, last, last) == make_heap + sort_heap
__device__ void heap_sort_(uint32_t* first, int len) {
    if (len >= 2) {
        int parent = (len - 2) / 2;
        while (true) {
            uint32_t value = first[parent];
            adjust_heap_(first, parent, len, value);
            if (parent == 0) {
                break;
            }
            parent--;
        }
    }
    int last = len;
    while (last > 1) {
        --last;
        uint32_t value = first[last];
        first[last] = first[0];
        adjust_heap_(first, 0, last, value);
    }
}

// ---------------- one wave: pivot + Hoare partition of [first,last), last-first > 16 ----------------
// returns cut (wave-uniform).  a may point to LDS or global memory.
__device__ int wave_partition(uint32_t* a, int first, int last, int lane) {
    // __move_median_to_first(first, first+1, mid, last-1)
    const int ia = first + 1;
    const int ib = first + (last - first) / 2;
    const int ic = last - 1;
    const uint32_t vf = a[first];
    const uint32_t va = a[ia];
    const uint32_t vb = a[ib];
    const uint32_t vc = a[ic];
    int im;
    uint32_t vm;
    if (comp(va, vb)) {
        if (comp(vb, vc)) {
            im = ib, vm = vb;
        } else if (comp(va, vc)) {
            im = ic, vm = vc;
        } else {
            im = ia, vm = va;
        }
    } else if (comp(va, vc)) {
        im = ia, vm = va;
    } else if (comp(vb, vc)) {
        im = ic, vm = vc;
    } else {
        im = ib, vm = vb;
    }
    wave_mem_fence();
    if (lane == 0) {
        a[first] = vm;
        a[im] = vf;
    }
    wave_mem_fence();
    const uint32_t p = score_of(vm);

    int li = first + 1;  // unread range [li, rj)
    int rj = last;
    unsigned long long Lm = 0;  // pending left stoppers  (score <= p), bit = lane of last left chunk
    unsigned long long Rm = 0;  // pending right stoppers (score >= p), bit = lane of last right chunk
    int lb = li, rb = rj;       // bases of the last chunks
    int nl = 0, nr = 0;         // their lengths
    uint32_t vl = 0, vr = 0;    // their ORIGINAL values, one per lane

    while (true) {
        if (Lm == 0 && li < rj) {
            nl = min(64, rj - li);
            lb = li;
            vl = lane < nl ? a[lb + lane] : 0u;
            Lm = __ballot(lane < nl && score_of(vl) <= p);
            li += nl;
        }
        if (Rm == 0 && li < rj) {
            nr = min(64, rj - li);
            rb = rj - nr;
            vr = lane < nr ? a[rb + lane] : 0u;
            Rm = __ballot(lane < nr && score_of(vr) >= p);
            rj -= nr;
        }
        if (Lm == 0 || Rm == 0) {
            if (li >= rj) {
                break;
            }
            continue;
        }
        const int c = min(__popcll(Lm), __popcll(Rm));
        const int k = lane < c ? lane : c - 1;
        const int pl = select64(Lm, k);
        const int pr = select64_top(Rm, k);
        const uint32_t xl = __shfl(vl, pl);
        const uint32_t xr = __shfl(vr, pr);
        if (lane < c) {
            a[lb + pl] = xr;
            a[rb + pr] = xl;
        }
        // drop the c lowest left stoppers and the c highest right stoppers
        const int plast = select64(Lm, c - 1);
        const int rlast = select64_top(Rm, c - 1);
        Lm &= ~low_mask(plast + 1);
        Rm &= low_mask(rlast);
    }
    wave_mem_fence();

    int cut;
    if (Lm != 0) {
        // case A: unpaired left stoppers remain in the last left chunk [lb, lb+nl); every
        // position >= lb+nl has been consumed by right chunks.  The right scan continues
        // downward inside this chunk over original values.
        const int lstar = __ffsll(static_cast<long long>(Lm)) - 1;
        const unsigned long long Rw = __ballot(lane < nl && lane >= lstar && score_of(vl) >= p);
        const int cl = __popcll(Lm);
        const int cr = __popcll(Rw);
        const int cmax = min(cl, cr);
        int pl = 0, pr = 0;
        bool ok = false;
        if (lane < cmax) {
            pl = select64(Lm, lane);
            pr = select64_top(Rw, lane);
            ok = pl < pr;
        }
        const unsigned long long okm = __ballot(ok);
        const int K = __popcll(okm);  // ok is a prefix property
        const int k = lane < K ? lane : 0;
        const int spl = __shfl(pl, k);
        const int spr = __shfl(pr, k);
        const uint32_t xl = __shfl(vl, spl);
        const uint32_t xr = __shfl(vl, spr);
        if (lane < K) {
            a[lb + spl] = xr;
            a[lb + spr] = xl;
        }
        cut = lb + nl;  // first position of the right-consumed territory (or `last`)
        if (cl > K) {
            cut = min(cut, lb + select64(Lm, K));
        }
        if (K >= 1) {
            cut = min(cut, lb + select64_top(Rw, K - 1));
        }
    } else if (Rm != 0) {
        // case B: unpaired right stoppers remain in the last right chunk [rb, rb+nr); the left
        // scan continues upward inside this chunk over original values.
        const int rstar = 63 - __clzll(static_cast<long long>(Rm));
        const unsigned long long Lw = __ballot(lane < nr && lane <= rstar && score_of(vr) <= p);
        const int cl = __popcll(Lw);
        const int cr = __popcll(Rm);
        const int cmax = min(cl, cr);
        int pl = 0, pr = 0;
        bool ok = false;
        if (lane < cmax) {
            pl = select64(Lw, lane);
            pr = select64_top(Rm, lane);
            ok = pl < pr;
        }
        const unsigned long long okm = __ballot(ok);
        const int K = __popcll(okm);
        const int k = lane < K ? lane : 0;
        const int spl = __shfl(pl, k);
        const int spr = __shfl(pr, k);
        const uint32_t xl = __shfl(vr, spl);
        const uint32_t xr = __shfl(vr, spr);
        if (lane < K) {
            a[rb + spl] = xr;
            a[rb + spr] = xl;
        }
        cut = rb + rstar + 1;  // everything above r* is already a stopper for the left scan
        if (cl > K) {
            cut = min(cut, rb + select64(Lw, K));
        }
        if (K >= 1) {
            cut = min(cut, rb + select64_top(Rm, K - 1));
        }
    } else {
        // case C: both queues drained exactly at the meeting point
        cut = rj;
    }
    wave_mem_fence();
    return cut;
}

// Sort n keys descending by score exactly as std::sort does.  One workgroup (256 threads).
//   a        : n keys (LDS or global), permuted in place by the introsort loop
//   out      : final order (global)
//   segstart : n+1 flags (global or LDS); seglist: 2*maxSeg entries
// Returns 0 or STATUS_SEG_OVERFLOW.
__device__ int block_sort_desc(uint32_t* a, uint32_t* out, uint8_t* segstart, int2* seglist, int maxSeg, int n,
                               int* lds_counters /* >= 4 ints */) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    for (int i = tid; i <= n; i += blockDim.x) {
        segstart[i] = (i == 0 || i == n) ? 1 : 0;
    }
    int depth = 0;
    if (n > 1) {
        depth = 2 * (31 - __clz(n));  // std::__lg(n) * 2
    }
    if (tid == 0) {
        lds_counters[0] = n > SORT_LEAF ? 1 : 0;  // current level count
        lds_counters[1] = 0;                      // next level count
        lds_counters[2] = 0;                      // overflow flag
        if (n > SORT_LEAF) {
            seglist[0] = make_int2(0, n);
        }
    }
    __syncthreads();
    int cur = 0;
    while (true) {
        const int count = lds_counters[0];
        if (count == 0) {
            break;
        }
        int2* src = seglist + cur * maxSeg;
        int2* dst = seglist + (cur ^ 1) * maxSeg;
        for (int s = wave; s < count; s += nwaves) {
            const int2 seg = src[s];
            const int first = seg.x;
            const int last = seg.y;
            if (depth == 0) {
                // __partial_sort(first, last, last): heapsort by one lane; every element becomes
                // its own leaf (the segment is now sorted, the final pass must not touch it)
                if (lane == 0) {
                    heap_sort_(a + first, last - first);
                }
                wave_mem_fence();
                for (int i = first + lane; i < last; i += 64) {
                    segstart[i] = 1;
                }
                continue;
            }
            const int cut = wave_partition(a, first, last, lane);
            if (lane == 0) {
                segstart[cut] = 1;
                // right child [cut,last) first, then left [first,cut): any order is fine, the
                // children are independent
                if (last - cut > SORT_LEAF) {
                    const int slot = atomicAdd(&lds_counters[1], 1);
                    if (slot < maxSeg) {
                        dst[slot] = make_int2(cut, last);
                    } else {
                        lds_counters[2] = 1;
                    }
                }
                if (cut - first > SORT_LEAF) {
                    const int slot = atomicAdd(&lds_counters[1], 1);
                    if (slot < maxSeg) {
                        dst[slot] = make_int2(first, cut);
                    } else {
                        lds_counters[2] = 1;
                    }
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        if (tid == 0) {
            lds_counters[0] = min(lds_counters[1], maxSeg);
            lds_counters[1] = 0;
        }
        depth = depth > 0 ? depth - 1 : 0;
        cur ^= 1;
        __syncthreads();
    }
    const int overflow = lds_counters[2];
    __syncthreads();
    // __final_insertion_sort == stable sort inside each leaf: rank counting, one lane per element
    for (int i = tid; i < n; i += blockDim.x) {
        int s = i;
        while (segstart[s] == 0) {
            --s;
        }
        int e = i + 1;
        while (segstart[e] == 0) {
            ++e;
        }
        const uint32_t me = a[i];
        const uint32_t sc = score_of(me);
        int pos = s;
        for (int j = s; j < e; ++j) {
            const uint32_t o = score_of(a[j]);
            pos += (o > sc || (o == sc && j < i)) ? 1 : 0;
        }
        out[pos] = me;
    }
    __threadfence_block();
    __syncthreads();
    return overflow != 0 ? STATUS_SEG_OVERFLOW : 0;
}

extern __shared__ uint32_t dyn_lds[];

__global__ __launch_bounds__(256) void sort_kernel(BatchView v, int lds_keys) {
    __shared__ int counters[4];
    const int f = blockIdx.x;
    const int n = v.n_cand[f];
    if (n == 0) {
        return;
    }
    uint32_t* gkeys = v.keys + static_cast<size_t>(f) * v.maxM;
    uint32_t* out = v.sorted + static_cast<size_t>(f) * v.maxM;
    uint8_t* segstart = v.segstart + static_cast<size_t>(f) * (v.maxM + 1);
    int2* seglist = v.seglist + static_cast<size_t>(f) * 2 * v.maxSeg;
    uint32_t* a = gkeys;
    if (n <= lds_keys) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            dyn_lds[i] = gkeys[i];
        }
        a = dyn_lds;
        __syncthreads();
    }
    const int st = block_sort_desc(a, out, segstart, seglist, v.maxSeg, n, counters);
    uint32_t* rank_of = v.rank_of + static_cast<size_t>(f) * v.maxM;
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        rank_of[out[r] & INDEX_MASK] = static_cast<uint32_t>(r);
    }
    if (threadIdx.x == 0 && st != 0) {
        atomicOr(&v.status[f], st);
    }
}

__global__ __launch_bounds__(256) void debug_sort_kernel(uint32_t* keys, uint32_t* sorted, uint8_t* segstart,
                                                         int2* seglist, int maxSeg, int n, int* status,
                                                         int lds_keys) {
    __shared__ int counters[4];
    uint32_t* a = keys;
    if (n <= lds_keys) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            dyn_lds[i] = keys[i];
        }
        a = dyn_lds;
        __syncthreads();
    }
    const int st = block_sort_desc(a, sorted, segstart, seglist, maxSeg, n, counters);
    if (threadIdx.x == 0) {
        *status = st;
    }
}

int lds_keys_for(int maxM) {
    // keys kept in LDS while they fit ~96 KiB (leaves room for a second workgroup per CU on
    // small frames); larger frames fall back to global memory inside the same kernel
    int cap = 24576;
    return maxM < cap ? maxM : cap;
}

}  // namespace

void allow_big_lds(const void* fn) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4);
}

void launch_sort(const BatchView& v, int nframes, hipStream_t stream) {
    static const bool once = (allow_big_lds(reinterpret_cast<const void*>(sort_kernel)), true);
    (void)once;
    const int lds_keys = lds_keys_for(v.maxM);
    hipLaunchKernelGGL(sort_kernel, dim3(nframes), dim3(256), static_cast<size_t>(lds_keys) * 4, stream, v, lds_keys);
}

void launch_debug_sort(uint32_t* keys, uint32_t* sorted, uint8_t* segstart, int2* seglist, int maxSeg, int n,
                       int* status, hipStream_t stream) {
    static const bool once = (allow_big_lds(reinterpret_cast<const void*>(debug_sort_kernel)), true);
    (void)once;
    const int lds_keys = lds_keys_for(n);
    hipLaunchKernelGGL(debug_sort_kernel, dim3(1), dim3(256), static_cast<size_t>(lds_keys) * 4, stream, keys, sorted,
                       segstart, seglist, maxSeg, n, status, lds_keys);
}

}  // namespace slamx
