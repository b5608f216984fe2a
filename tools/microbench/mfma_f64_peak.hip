// mfma_f64_peak.hip -- sustained rate of v_mfma_f64_16x16x4_f64 on gfx950 (independent accumulators,
// back-to-back issue, 1 / 2 waves per SIMD), to calibrate the MFMA roofline of gn_accumulate_kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak mfma_f64_peak.hip ; run: ./mfma_f64_peak
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(64) void mfma_loop(double* out, int iters) {
    d4 acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        acc[k] = d4{0.0, 0.0, 0.0, 0.0};
    }
    const double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
        }
    }
    double r = 0.0;
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        r += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    }
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

int main() {
    double* d = nullptr;
    hipMalloc(&d, sizeof(double) * 64 * 8192);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    auto run = [&](const char* name, auto kernel, int nacc, int waves) {
        hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), 0, 0, d, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2048.0 * nacc * iters * double(waves);
        std::printf("%-40s %8.3f ms  %8.2f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, ms,
                    flops / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (double(nacc) * iters * (waves / 1024.0)));
    };
    run("16x16x4 f64, 8 acc, 1 wave / SIMD", mfma_loop<8>, 8, 1024);
    run("16x16x4 f64, 8 acc, 2 waves / SIMD", mfma_loop<8>, 8, 2048);
    run("16x16x4 f64, 15 acc, 2 waves / SIMD", mfma_loop<15>, 15, 2048);
    run("16x16x4 f64, 8 acc, 4 waves / SIMD", mfma_loop<8>, 8, 4096);
    run("16x16x4 f64, 4 acc, 8 waves / SIMD", mfma_loop<4>, 4, 8192);
    run("16x16x4 f64, 1 acc (dependent chain)", mfma_loop<1>, 1, 1024);
    hipFree(d);
    return 0;
}
