// valu_peak.hip -- measures the sustained int32 VALU rate of v_xor_b32 + v_bcnt_u32_b32 (the
// Hamming inner loop) on gfx950, to calibrate the roofline peak used by bench.py / DESIGN.md.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip ; run: ./valu_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int ILP>
__global__ __launch_bounds__(256) void xor_bcnt(uint32_t* out, int iters, uint32_t seed) {
    uint32_t q[ILP], acc[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        q[k] = seed * (threadIdx.x + 1) + k * 977u;
        acc[k] = 0;
    }
    uint32_t t = seed ^ blockIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            acc[k] = __popc(q[k] ^ t) + acc[k];  // v_xor_b32 + v_bcnt_u32_b32 (fused accumulate)
        }
        t = t * 1664525u + 1013904223u;  // uniform -> SALU
    }
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        r += acc[k];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int ILP>
__global__ __launch_bounds__(256) void xor_only(uint32_t* out, int iters, uint32_t seed) {
    uint32_t q[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        q[k] = seed * (threadIdx.x + 1) + k * 977u;
    }
    uint32_t t = seed ^ blockIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            q[k] = (q[k] ^ t) + q[(k + 1) % ILP];  // v_xor + v_add
        }
        t = t * 1664525u + 1013904223u;
    }
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        r += q[k];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    const int blocks = 256 * 8, threads = 256, iters = 20000;
    uint32_t* d = nullptr;
    hipMalloc(&d, sizeof(uint32_t) * blocks * threads);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto run = [&](const char* name, auto kernel, double ops_per_iter_lane) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d, 100, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d, iters, 12345u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double ops = ops_per_iter_lane * iters * double(blocks) * threads;
        std::printf("%-28s %8.3f ms  %8.2f Tera lane-ops/s\n", name, ms, ops / (ms * 1e-3) / 1e12);
    };
    run("xor+bcnt ILP8 (2 ops)", xor_bcnt<8>, 16.0);
    run("xor+bcnt ILP16 (2 ops)", xor_bcnt<16>, 32.0);
    run("xor+add ILP8 (2 ops)", xor_only<8>, 16.0);
    hipFree(d);
    return 0;
}
