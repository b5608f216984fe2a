// h2d_beside_kernels -- what slows the host -> HBM copies of the end-to-end leg (bench.py pcie_inclusive) when kernels
// run beside them?  Four lanes (streams) rotate as in the bench: a 1 GiB upload from page-locked memory, then a
// stand-in for the batch's kernels (~3 ms), then a 60 MB download.  The stand-in is one of:
//   none            copies only
//   hbm             a streaming read + write over 1 GiB on every CU (HBM-bound)
//   alu             a register-only spin on every CU (no memory traffic)
//   alu32           the same on 32 workgroups only
// and the uploads go either on the lanes' own streams (as the bench does) or all on ONE copy stream, in order, with an
// event per batch that the lane's stream waits for.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void hbm_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n, int passes) {
    for (int p = 0; p < passes; ++p) {
        for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
            uint4 v = src[i];
            v.x += p;
            dst[i] = v;
        }
    }
}

__global__ __launch_bounds__(256) void alu_kernel(int* out, int iters) {
    int a = threadIdx.x, b = blockIdx.x;
    for (int i = 0; i < iters; ++i) {
        a = a * 1664525 + b;
        b = b * 22695477 + a;
    }
    if (a == 0x12345678 && b == 0x9abcdef) {
        out[0] = a;
    }
}

int main(int argc, char** argv) {
    const size_t N = size_t(1) << 30, D = size_t(60) << 20;
    const int L = 4, steps = 12;
    void *h = nullptr, *h2 = nullptr;
    CK(hipHostMalloc(&h, N));
    CK(hipHostMalloc(&h2, D * L));
    std::memset(h, 1, N);
    std::vector<void*> d(L), dd(L), w(L);
    std::vector<hipStream_t> st(L);
    std::vector<hipEvent_t> up(L);
    hipStream_t copy_stream;
    CK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    for (int k = 0; k < L; ++k) {
        CK(hipMalloc(&d[k], N));
        CK(hipMalloc(&w[k], N));
        CK(hipMalloc(&dd[k], D));
        CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
        CK(hipEventCreateWithFlags(&up[k], hipEventDisableTiming));
    }
    int* sink = nullptr;
    CK(hipMalloc(&sink, 64));
    // calibrate the stand-ins to ~3 ms
    auto time_kernel = [&](int kind, int amount) -> double {
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        if (kind == 1) hipLaunchKernelGGL(hbm_kernel, dim3(2048), dim3(256), 0, st[0], (const uint4*)d[0], (uint4*)w[0], N / 16, amount);
        if (kind == 2) hipLaunchKernelGGL(alu_kernel, dim3(2048), dim3(256), 0, st[0], sink, amount);
        if (kind == 3) hipLaunchKernelGGL(alu_kernel, dim3(32), dim3(256), 0, st[0], sink, amount);
        (void)hipDeviceSynchronize();
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3;
    };
    int amount[4] = {0, 4, 20000, 20000};
    for (int kind = 1; kind <= 3; ++kind) {
        time_kernel(kind, amount[kind]);
        const double ms = time_kernel(kind, amount[kind]);
        amount[kind] = static_cast<int>(amount[kind] * 3.0 / ms + 0.5);
        if (amount[kind] < 1) amount[kind] = 1;
        std::printf("stand-in %d: amount %d -> %.2f ms\n", kind, amount[kind], time_kernel(kind, amount[kind]));
    }
    const char* names[4] = {"none", "hbm", "alu", "alu32"};
    for (int one_copy_stream = 0; one_copy_stream <= 1; ++one_copy_stream) {
        for (int kind = 0; kind <= 3; ++kind) {
            for (int with_d2h = 0; with_d2h <= 1; ++with_d2h) {
                CK(hipDeviceSynchronize());
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < steps; ++i) {
                    const int k = i % L;
                    if (i >= L) CK(hipStreamSynchronize(st[k]));  // the lane's previous batch is consumed first
                    if (one_copy_stream) {
                        CK(hipMemcpyAsync(d[k], h, N, hipMemcpyHostToDevice, copy_stream));
                        CK(hipEventRecord(up[k], copy_stream));
                        CK(hipStreamWaitEvent(st[k], up[k], 0));
                    } else {
                        CK(hipMemcpyAsync(d[k], h, N, hipMemcpyHostToDevice, st[k]));
                    }
                    if (kind == 1) hipLaunchKernelGGL(hbm_kernel, dim3(2048), dim3(256), 0, st[k], (const uint4*)d[k], (uint4*)w[k], N / 16, amount[1]);
                    if (kind == 2) hipLaunchKernelGGL(alu_kernel, dim3(2048), dim3(256), 0, st[k], sink, amount[2]);
                    if (kind == 3) hipLaunchKernelGGL(alu_kernel, dim3(32), dim3(256), 0, st[k], sink, amount[3]);
                    if (with_d2h) CK(hipMemcpyAsync((char*)h2 + k * D, dd[k], D, hipMemcpyDeviceToHost, st[k]));
                }
                CK(hipDeviceSynchronize());
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                std::printf("%-22s stand-in %-6s %-10s %6.1f GB/s host->HBM  (%.2f ms per step)\n",
                            one_copy_stream ? "one copy stream" : "uploads on the lanes", names[kind], with_d2h ? "+ download" : "",
                            steps * (N / 1e9) / dt, dt * 1e3 / steps);
            }
        }
    }
    return 0;
}
