// h2d_rate -- host -> HBM copy rate of one batch of frames (1 GiB of page-locked memory) as a function of how the copy
// is issued: one hipMemcpyAsync, the same bytes in chunks on one stream, and split over 2 / 4 streams (SDMA engines).
// Also the same with a device -> host copy of 60 MB running beside it (the result download of the end-to-end leg).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const size_t N = size_t(1) << 30, D = size_t(60) << 20;
    void *h = nullptr, *d = nullptr, *h2 = nullptr, *d2 = nullptr;
    CK(hipHostMalloc(&h, N)); CK(hipMalloc(&d, N)); CK(hipHostMalloc(&h2, D)); CK(hipMalloc(&d2, D));
    std::vector<hipStream_t> st(5);
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto run = [&](int streams, size_t chunk, bool with_d2h, const char* what) -> int {
        double best = 0;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            if (with_d2h) CK(hipMemcpyAsync(h2, d2, D, hipMemcpyDeviceToHost, st[4]));
            size_t per = N / streams;
            for (int s = 0; s < streams; ++s)
                for (size_t o = 0; o < per; o += chunk)
                    CK(hipMemcpyAsync((char*)d + s * per + o, (char*)h + s * per + o, std::min(chunk, per - o), hipMemcpyHostToDevice, st[s]));
            CK(hipDeviceSynchronize());
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            best = std::max(best, N / dt / 1e9);
        }
        std::printf("%-46s %6.1f GB/s\n", what, best);
        return 0;
    };
    run(1, N, false, "1 stream, one copy");
    run(1, size_t(64) << 20, false, "1 stream, 64 MB chunks");
    run(2, N, false, "2 streams, half each");
    run(4, N, false, "4 streams, quarter each");
    run(2, size_t(32) << 20, false, "2 streams, 32 MB chunks");
    run(1, N, true, "1 stream + 60 MB D2H beside it");
    run(2, N, true, "2 streams + 60 MB D2H beside it");
    run(4, N, true, "4 streams + 60 MB D2H beside it");
    return 0;
}
