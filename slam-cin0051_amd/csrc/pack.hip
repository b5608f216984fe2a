// pack.hip -- the return path of a streaming caller: the VALID keypoints, descriptors and matches of a batch,
// gathered into one contiguous buffer before they cross PCIe.
//
// Reference: the results the reference hands back per frame are `std::vector<Keypoint>&`, `DescriptorMatrix&`
// (src/frontend/feature_detector.cpp:49-54: exactly n rows) and `std::vector<Match>&` (feature_matcher.cpp:71-95).
// The device keeps them in capacity-padded per-frame slices ([F][maxK]); copying those to the host ships 2-3x the
// bytes that are results (262 KB per 1080p frame for ~116 KB valid).  Here: a prefix over the per-frame counts
// (one workgroup), then a copy kernel that moves every frame's valid prefix to its packed position.  HBM-bound
// byte moving (read + write of the valid bytes, dword loads / stores, coalesced); layout in include/slamx.h
// (slamx_packed_header).  When the caller's buffer is page-locked memory the device can address, the copy kernel
// writes every dword there as well -- the results then cross PCIe as the kernel's own stores, beside the uploads of
// the next batches (a copy-engine download was measured to take its turn with them: 1.3 ms per 512 frames).
#include "slamx_internal.h"

namespace slamx {

namespace {

__device__ __forceinline__ long long align16(long long x) { return (x + 15) & ~15LL; }

// One workgroup: exclusive prefixes of n_kp / n_matches over the frames, the header, the per-frame status.
__global__ __launch_bounds__(1024) void pack_offsets_kernel(const int* __restrict__ n_kp, const int* __restrict__ n_matches,
                                                            const int* __restrict__ status, int nframes, int width,
                                                            uint8_t* __restrict__ pack) {
    __shared__ int wave_k[16], wave_m[16];
    __shared__ int carry_k, carry_m;
    slamx_packed_header* h = reinterpret_cast<slamx_packed_header*>(pack);
    const long long first_off = align16(sizeof(slamx_packed_header));
    const long long mfirst_off = align16(first_off + 4LL * (nframes + 1));
    const long long status_off = align16(mfirst_off + 4LL * (nframes + 1));
    const long long kps_off = align16(status_off + 4LL * nframes);
    int* kp_first = reinterpret_cast<int*>(pack + first_off);
    int* m_first = reinterpret_cast<int*>(pack + mfirst_off);
    int* st = reinterpret_cast<int*>(pack + status_off);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    if (tid == 0) {
        carry_k = 0;
        carry_m = 0;
    }
    __syncthreads();
    for (int base = 0; base < nframes; base += 1024) {
        const int f = base + tid;
        const int k = f < nframes ? n_kp[f] : 0;
        const int m = (f < nframes && n_matches != nullptr) ? n_matches[f] : 0;
        // inclusive wave scans
        int sk = k, sm = m;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int ok = __shfl_up(sk, d);
            const int om = __shfl_up(sm, d);
            if (lane >= d) {
                sk += ok;
                sm += om;
            }
        }
        if (lane == 63) {
            wave_k[wave] = sk;
            wave_m[wave] = sm;
        }
        __syncthreads();
        int bk = carry_k, bm = carry_m;
        for (int w = 0; w < wave; ++w) {
            bk += wave_k[w];
            bm += wave_m[w];
        }
        if (f < nframes) {
            kp_first[f] = bk + sk - k;
            m_first[f] = bm + sm - m;
            st[f] = status[f];
        }
        __syncthreads();
        if (tid == 1023) {
            carry_k = bk + sk;
            carry_m = bm + sm;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int tk = carry_k, tm = carry_m;
        kp_first[nframes] = tk;
        m_first[nframes] = tm;
        const long long desc_off = align16(kps_off + 20LL * tk);
        const long long matches_off = align16(desc_off + static_cast<long long>(width) * tk);
        h->nframes = nframes;
        h->width = width;
        h->total_keypoints = tk;
        h->total_matches = tm;
        h->first_off = first_off;
        h->mfirst_off = mfirst_off;
        h->status_off = status_off;
        h->kps_off = kps_off;
        h->desc_off = desc_off;
        h->matches_off = matches_off;
        h->total_bytes = align16(matches_off + 12LL * tm);
    }
}

// dst (HBM) always; `mirror` (the caller's page-locked buffer, written over PCIe by the kernel itself) where it is given
// and the dword lies inside its first mirror_bytes
__device__ __forceinline__ void copy_dwords(const uint32_t* __restrict__ src, uint8_t* __restrict__ pack, long long off, int n,
                                            int tid, int nthreads, uint8_t* __restrict__ mirror, long long mirror_bytes) {
    uint32_t* dst = reinterpret_cast<uint32_t*>(pack + off);
    if (mirror == nullptr) {
        for (int i = tid; i < n; i += nthreads) {
            dst[i] = src[i];
        }
        return;
    }
    uint32_t* far = reinterpret_cast<uint32_t*>(mirror + off);
    const long long room = (mirror_bytes - off) / 4;  // whole dwords of the mirror from `off` on (may be <= 0)
    for (int i = tid; i < n; i += nthreads) {
        const uint32_t v = src[i];
        dst[i] = v;
        if (i < room) {
            far[i] = v;
        }
    }
}

// grid (blocks per frame, frames): a frame's valid keypoints, descriptors and matches to their packed positions
__global__ __launch_bounds__(256) void pack_copy_kernel(const slamx_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                        const slamx_match* __restrict__ matches, int maxK, int width,
                                                        uint8_t* __restrict__ pack, uint8_t* __restrict__ mirror,
                                                        long long mirror_bytes) {
    const slamx_packed_header* h = reinterpret_cast<const slamx_packed_header*>(pack);
    const int f = blockIdx.y;
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int nthreads = gridDim.x * 256;
    const int* kp_first = reinterpret_cast<const int*>(pack + h->first_off);
    const int* m_first = reinterpret_cast<const int*>(pack + h->mfirst_off);
    const int k0 = kp_first[f], nk = kp_first[f + 1] - k0;
    const int m0 = m_first[f], nm = m_first[f + 1] - m0;
    if (mirror != nullptr && f == 0) {  // the header and the per-frame tables (written by pack_offsets_kernel): [0, kps_off)
        const long long head = min(h->kps_off, mirror_bytes) / 4;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(pack);
        uint32_t* far = reinterpret_cast<uint32_t*>(mirror);
        for (long long i = tid; i < head; i += nthreads) {
            far[i] = src[i];
        }
    }
    if (f == 0 && tid < 48) {  // the (< 16-byte) gaps behind the three sections: zero, so that the buffer is a function of the results
        const int g = tid >> 4;
        const long long tk = kp_first[gridDim.y], tm = m_first[gridDim.y];
        const long long from = g == 0 ? h->kps_off + 20LL * tk : g == 1 ? h->desc_off + width * tk : h->matches_off + 12LL * tm;
        const long long to = g == 0 ? h->desc_off : g == 1 ? h->matches_off : h->total_bytes;
        const long long at = from + (tid & 15);
        if (at < to) {
            pack[at] = 0;
            if (mirror != nullptr && at < mirror_bytes) {
                mirror[at] = 0;
            }
        }
    }
    // keypoint records are 5 dwords, matches 3: whole dwords, aligned on both sides (sections start 16-byte aligned)
    copy_dwords(reinterpret_cast<const uint32_t*>(kps + static_cast<size_t>(f) * maxK), pack, h->kps_off + 20LL * k0, 5 * nk, tid,
                nthreads, mirror, mirror_bytes);
    if ((width & 3) == 0) {
        copy_dwords(reinterpret_cast<const uint32_t*>(desc + static_cast<size_t>(f) * maxK * width), pack,
                    h->desc_off + static_cast<long long>(width) * k0, (width / 4) * nk, tid, nthreads, mirror, mirror_bytes);
    } else {
        const uint8_t* s = desc + static_cast<size_t>(f) * maxK * width;
        const long long off = h->desc_off + static_cast<long long>(width) * k0;
        for (int i = tid; i < nk * width; i += nthreads) {
            pack[off + i] = s[i];
            if (mirror != nullptr && off + i < mirror_bytes) {
                mirror[off + i] = s[i];
            }
        }
    }
    if (matches != nullptr && nm > 0) {
        copy_dwords(reinterpret_cast<const uint32_t*>(matches + static_cast<size_t>(f) * maxK), pack, h->matches_off + 12LL * m0,
                    3 * nm, tid, nthreads, mirror, mirror_bytes);
    }
}

}  // namespace

size_t packed_bound_bytes(int nframes, int maxK, int width, bool with_matches) {
    const size_t F = static_cast<size_t>(nframes);
    size_t b = 16 + sizeof(slamx_packed_header) + 3 * (4 * (F + 1) + 16);
    b += F * maxK * 20 + 16;
    b += F * maxK * static_cast<size_t>(width) + 16;
    if (with_matches) {
        b += F * maxK * 12 + 16;
    }
    return (b + 15) & ~static_cast<size_t>(15);
}

void launch_pack_results(const BatchView& v, const slamx_match* matches, const int* n_matches, int nframes, uint8_t* pack,
                         hipStream_t stream, uint8_t* mirror, size_t mirror_bytes) {
    hipLaunchKernelGGL(pack_offsets_kernel, dim3(1), dim3(1024), 0, stream, v.n_kp, n_matches, v.status, nframes, v.width, pack);
    hipLaunchKernelGGL(pack_copy_kernel, dim3(4, nframes), dim3(256), 0, stream, v.kps, v.desc, matches, v.maxK, v.width, pack,
                       mirror, static_cast<long long>(mirror_bytes));
}

}  // namespace slamx
