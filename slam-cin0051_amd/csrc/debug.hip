// debug.hip -- test-only kernels that expose device sub-steps to the parity tests.
#include "glibc_trig.h"
#include "slamx_internal.h"

#pragma clang fp contract(off)

namespace slamx {

namespace {

// (m01, m10) -> angle in degrees, cos, sin exactly as computeOrientation /
// computeBRIEFDescriptor chain them (feature_detector.cpp:229, :247-249).
__global__ void trig_kernel(const float* m01, const float* m10, int n, float* deg, float* c, float* s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    const float angle = __fmul_rn(glibc_atan2f(m01[i], m10[i]), rad_to_deg());
    const float a = __fmul_rn(angle, deg_to_rad());
    deg[i] = angle;
    // the combined routine of the describe kernels; the generic describe_kernel keeps the two separate
    // functions, and test_compute_other_patterns compares that path with the oracle as well
    glibc_sincosf(a, &s[i], &c[i]);
}

// the penalty arithmetic of match.hip (feature_matcher.cpp:161-170), one tuple per lane
__global__ void penalty_kernel(const int* dist, const float* qx, const float* qy, const float* tx, const float* ty, int n,
                               int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    out[i] = penalised_distance(dist[i], qx[i], qy[i], tx[i], ty[i]);
}

}  // namespace

void launch_debug_penalty(const int* dist, const float* qx, const float* qy, const float* tx, const float* ty, int n,
                          int* out, hipStream_t stream) {
    if (n > 0) {
        hipLaunchKernelGGL(penalty_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, dist, qx, qy, tx, ty, n, out);
    }
}

void launch_debug_trig(const float* m01, const float* m10, int n, float* deg, float* c, float* s, hipStream_t stream) {
    if (n <= 0) {
        return;
    }
    hipLaunchKernelGGL(trig_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, m01, m10, n, deg, c, s);
}

}  // namespace slamx
