// development microbenchmark: 16 waves of one workgroup, each partitioning its own copy of the same segment at the same time
#include "../../slam-cin0051_amd/csrc/order.hip"
#include <cstdio>
#include <vector>
#include <random>
namespace slamx {
namespace {
__global__ __launch_bounds__(1024) void wp_multi_kernel(const uint32_t* keys, int n, int active_waves, unsigned long long* cycles) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int stride = ((n + 3) & ~3) + ((n + 8) / 2 + 4);  // keys + 16-bit lists, in words
    uint32_t* mine = lds + wave * stride;
    uint16_t* scr = reinterpret_cast<uint16_t*>(mine + ((n + 3) & ~3));
    unsigned long long best = ~0ull;
    for (int r = 0; r < 4; ++r) {
        for (int i = lane; i < n; i += 64) {
            mine[i] = keys[i];
        }
        __syncthreads();
        if (wave < active_waves) {
            const unsigned long long t0 = clock64();
            const int cut = wave_partition_cf(mine, scr, 0, n, lane);
            const unsigned long long t1 = clock64();
            best = t1 - t0 < best ? t1 - t0 : best;
            if (cut < 0) {
                cycles[100] = 1;
            }
        }
        __syncthreads();
    }
    if (lane == 0) {
        cycles[wave] = best;
    }
}
}  // namespace
}  // namespace slamx
int main() {
    using namespace slamx;
    uint32_t* d_keys; unsigned long long* d_cyc;
    hipMalloc(&d_keys, 16384 * 4); hipMalloc(&d_cyc, 128 * 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wp_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    for (int n : {300, 1200}) {
        std::mt19937 rng(99 + n);
        std::vector<uint32_t> h(n);
        for (int i = 0; i < n; ++i) h[i] = ((20 + (rng() % 8) * (rng() % 8)) << SCORE_SHIFT) | uint32_t(i);
        hipMemcpy(d_keys, h.data(), n * 4, hipMemcpyHostToDevice);
        for (int active : {1, 4, 8, 16}) {
            const int stride = ((n + 3) & ~3) + ((n + 8) / 2 + 4);
            hipMemset(d_cyc, 0, 128 * 8);
            hipLaunchKernelGGL(wp_multi_kernel, dim3(1), dim3(1024), size_t(stride) * 16 * 4, 0, d_keys, n, active, d_cyc);
            if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return 1; }
            unsigned long long c[16];
            hipMemcpy(c, d_cyc, sizeof(c), hipMemcpyDeviceToHost);
            printf("n %4d active %2d:", n, active);
            for (int w = 0; w < active; ++w) printf(" %llu", c[w]);
            printf("\n");
        }
    }
    return 0;
}
