// mfma_valu_overlap.hip -- do int8 matrix-core instructions and vector instructions of OTHER waves (or of the same wave)
// share a SIMD's time on gfx950, or do they run side by side?  (match_spatial_kernel: 4 v_mfma_i32_16x16x64_i8 + ~27
// vector instructions per train tile; DESIGN.md 4.12 infers from counters that the times add.)
//   A  every wave: MFMAs only            B  every wave: vector instructions only
//   C  every wave: both, interleaved     D  half the waves of a SIMD MFMAs only, the other half vector only
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>

#include <cstdio>

typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));

template <int MODE, int VPM>  // VPM: vector instructions per MFMA
__global__ __launch_bounds__(512) void k(int* out, int iters) {
    const int wave = threadIdx.x >> 6;
    i4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc[q] = i4{0, 0, 0, 0};
    }
    i4v a = {static_cast<int>(threadIdx.x), 1, 2, 3}, b = {3, 2, 1, static_cast<int>(threadIdx.x)};
    uint32_t v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        v[q] = threadIdx.x * 2654435761u + q;
    }
    // MODE 3: waves 0, 1 of the workgroup (SIMDs 0, 1 ... the hardware deals waves round robin over the SIMDs, so with
    // 8 waves per workgroup waves w and w + 4 share a SIMD) -- here: even "pair index" does MFMAs, odd does vector work
    const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && (wave >> 2) % 2 == 0);
    const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && (wave >> 2) % 2 == 1);
    auto mfma_step = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[q], 0, 0, 0);
        }
    };
    auto valu_step = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int r = 0; r < VPM; ++r) {
                v[(q * VPM + r) & 7] = __builtin_amdgcn_perm(v[(q * VPM + r) & 7], v[(q * VPM + r + 3) & 7], 0x02010003u) | 1u;  // v_perm + v_or
            }
        }
    };
    if (MODE == 3) {
        // two loops, chosen per wave OUTSIDE of them (no exec masking inside): waves w and w + 4 of a workgroup share a SIMD
        if (__builtin_amdgcn_readfirstlane(wave >> 2) % 2 == 0) {
            for (int i = 0; i < iters; ++i) {
                mfma_step();
            }
        } else {
            for (int i = 0; i < iters; ++i) {
                valu_step();
            }
        }
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (do_mfma) {
                    acc[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[q], 0, 0, 0);
                }
                if (do_valu) {
#pragma unroll
                    for (int r = 0; r < VPM; ++r) {
                        v[(q * VPM + r) & 7] = __builtin_amdgcn_perm(v[(q * VPM + r) & 7], v[(q * VPM + r + 3) & 7], 0x02010003u) | 1u;
                    }
                }
            }
        }
    }
    int r = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        r += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        r += static_cast<int>(v[q]);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    int* d = nullptr;
    hipMalloc(&d, sizeof(int) * 512 * 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    auto run = [&](const char* name, auto kernel, int threads, int blocks) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const int waves_per_simd = threads / 64 / 4;
        std::printf("%-66s %8.3f ms  %6.1f cycles per group of 4 MFMA slots per wave (x %d waves per SIMD = %.1f)\n", name, ms,
                    ms * 1e-3 * 2.4e9 / iters, waves_per_simd, ms * 1e-3 * 2.4e9 / iters / 1.0);
    };
    // one workgroup per CU (256 CUs), 4 or 8 waves per workgroup = 1 or 2 per SIMD
    run("A  MFMA only, 1 wave / SIMD", k<0, 6>, 256, 256);
    run("A  MFMA only, 2 waves / SIMD", k<0, 6>, 512, 256);
    run("B  6 x 2 vector instructions per slot only, 1 wave / SIMD", k<1, 6>, 256, 256);
    run("B  the same, 2 waves / SIMD", k<1, 6>, 512, 256);
    run("C  both in every wave, 1 wave / SIMD", k<2, 6>, 256, 256);
    run("C  both in every wave, 2 waves / SIMD", k<2, 6>, 512, 256);
    run("D  per SIMD: one wave MFMA only, one wave vector only (6 x 2)", k<3, 6>, 512, 256);
    run("D3 per SIMD: one wave MFMA only, one wave vector only (3 x 2)", k<3, 3>, 512, 256);
    run("D1 per SIMD: one wave MFMA only, one wave vector only (1 x 2)", k<3, 1>, 512, 256);
    run("B1 1 x 2 vector instructions per slot only, 1 wave / SIMD", k<1, 1>, 256, 256);
    run("B3 3 x 2 vector instructions per slot only, 2 waves / SIMD", k<1, 3>, 512, 256);
    run("C3 both (3 x 2 per MFMA), 2 waves / SIMD", k<2, 3>, 512, 256);
    hipFree(d);
    return 0;
}
