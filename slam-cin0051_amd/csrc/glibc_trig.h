// glibc_trig.h -- device-side restatement of the glibc 2.35 x86-64 libm routines that the
// reference reaches from computeOrientation (std::atan2 on floats, feature_detector.cpp:229)
// and computeBRIEFDescriptor (std::cos / std::sin on floats, :248-249).
//
// THIRD-PARTY ARITHMETIC (hard part 2): the reference's results are
// whatever glibc's atan2f / cosf / sinf return, and truncations such as int(px*cos - py*sin)
// flip on a 1-ulp difference, so the device must reproduce these functions bit for bit.
//   * atan2f / atanf: glibc sysdeps/ieee754/flt-32/e_atan2f.c and s_atanf.c (fdlibm): float
//     arithmetic, one operation per rounding, no FMA in the shipped x86-64 build (verified in
//     the disassembly of libm.so.6 2.35-0ubuntu3.11: mulss/addss/subss/divss only).
//   * sinf / cosf: glibc sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h (ARM optimized
//     routines): double arithmetic on a 2/pi reduction and two degree-4/-3 polynomials.  On
//     x86-64 they are IFUNCs; hosts with AVX2+FMA (this container's Xeon and every current
//     EPYC) run the __sinf_fma/__cosf_fma variants, whose contraction pattern was read off the
//     disassembly and is reproduced here with explicit fma() calls.  Constants were dumped
//     from the same binary's .rodata (__sincosf_table).
//   Only |x| < 120 is implemented for sinf/cosf: the argument is atan2f()*57.29578f*0.017453292f,
//   so |x| <= pi(1+2^-22).  Larger arguments return NaN (never reached).
// tests/test_trig_gpu.py compares these against the host libm on hundreds of thousands of
// (m01, m10) pairs and on dense angle sweeps.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slamx {

__device__ __forceinline__ float fm(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fa(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fs(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fd(float a, float b) { return __fdiv_rn(a, b); }

// s_atanf.c
__device__ inline float glibc_atanf(float x) {
    const float atanhi[4] = {__uint_as_float(0x3eed6338u), __uint_as_float(0x3f490fdau), __uint_as_float(0x3f7b985eu),
                             __uint_as_float(0x3fc90fdau)};
    const float atanlo[4] = {__uint_as_float(0x31ac3769u), __uint_as_float(0x33222168u), __uint_as_float(0x33140fb4u),
                             __uint_as_float(0x33a22168u)};
    const float aT0 = __uint_as_float(0x3eaaaaabu);
    const float aT1 = __uint_as_float(0xbe4ccccdu);
    const float aT2 = __uint_as_float(0x3e124925u);
    const float aT3 = __uint_as_float(0xbde38e38u);
    const float aT4 = __uint_as_float(0x3dba2e6eu);
    const float aT5 = __uint_as_float(0xbd9d8795u);
    const float aT6 = __uint_as_float(0x3d886b35u);
    const float aT7 = __uint_as_float(0xbd6ef16bu);
    const float aT8 = __uint_as_float(0x3d4bda59u);
    const float aT9 = __uint_as_float(0xbd15a221u);
    const float aT10 = __uint_as_float(0x3c8569d7u);
    const uint32_t hx = __float_as_uint(x);
    const uint32_t ix = hx & 0x7fffffffu;
    int id;
    if (ix >= 0x4c000000u) {  // |x| >= 2^25
        if (ix > 0x7f800000u) {
            return fa(x, x);
        }
        if (static_cast<int32_t>(hx) > 0) {
            return fa(atanhi[3], atanlo[3]);
        }
        return fs(-atanhi[3], atanlo[3]);
    }
    if (ix < 0x3ee00000u) {      // |x| < 0.4375
        if (ix < 0x31000000u) {  // |x| < 2^-29: huge + x > one always holds
            return x;
        }
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000u) {      // |x| < 1.1875
            if (ix < 0x3f300000u) {  // 7/16 <= |x| < 11/16
                id = 0;
                x = fd(fs(fa(x, x), 1.0f), fa(2.0f, x));
            } else {
                id = 1;
                x = fd(fs(x, 1.0f), fa(x, 1.0f));
            }
        } else {
            if (ix < 0x401c0000u) {  // |x| < 2.4375
                id = 2;
                x = fd(fs(x, 1.5f), fa(1.0f, fm(1.5f, x)));
            } else {
                id = 3;
                x = fd(-1.0f, x);
            }
        }
    }
    const float z = fm(x, x);
    const float w = fm(z, z);
    const float s1 = fm(z, fa(aT0, fm(w, fa(aT2, fm(w, fa(aT4, fm(w, fa(aT6, fm(w, fa(aT8, fm(w, aT10)))))))))));
    const float s2 = fm(w, fa(aT1, fm(w, fa(aT3, fm(w, fa(aT5, fm(w, fa(aT7, fm(w, aT9)))))))));
    if (id < 0) {
        return fs(x, fm(x, fa(s1, s2)));
    }
    const float r = fs(atanhi[id], fs(fs(fm(x, fa(s1, s2)), atanlo[id]), x));
    return (static_cast<int32_t>(hx) < 0) ? -r : r;
}

// e_atan2f.c (__ieee754_atan2f); the w_atan2f wrapper only sets errno.
__device__ inline float glibc_atan2f(float y, float x) {
    const float tiny = __uint_as_float(0x0da24260u);   // 1.0e-30
    const float pi_o_4 = __uint_as_float(0x3f490fdbu);
    const float pi_o_2 = __uint_as_float(0x3fc90fdbu);
    const float pi = __uint_as_float(0x40490fdbu);
    const float pi_lo = __uint_as_float(0xb3bbbd2eu);
    const int32_t hx = static_cast<int32_t>(__float_as_uint(x));
    const int32_t hy = static_cast<int32_t>(__float_as_uint(y));
    const int32_t ix = hx & 0x7fffffff;
    const int32_t iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) {
        return fa(x, y);
    }
    if (hx == 0x3f800000) {
        return glibc_atanf(y);
    }
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
            case 0:
            case 1:
                return y;
            case 2:
                return fa(pi, tiny);
            default:
                return fs(-pi, tiny);
        }
    }
    if (ix == 0) {
        return (hy < 0) ? fs(-pi_o_2, tiny) : fa(pi_o_2, tiny);
    }
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0:
                    return fa(pi_o_4, tiny);
                case 1:
                    return fs(-pi_o_4, tiny);
                case 2:
                    return fa(fm(3.0f, pi_o_4), tiny);
                default:
                    return fs(fm(-3.0f, pi_o_4), tiny);
            }
        } else {
            switch (m) {
                case 0:
                    return 0.0f;
                case 1:
                    return -0.0f;
                case 2:
                    return fa(pi, tiny);
                default:
                    return fs(-pi, tiny);
            }
        }
    }
    if (iy == 0x7f800000) {
        return (hy < 0) ? fs(-pi_o_2, tiny) : fa(pi_o_2, tiny);
    }
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) {
        z = fa(pi_o_2, fm(0.5f, pi_lo));
    } else if (hx < 0 && k < -60) {
        z = 0.0f;
    } else {
        z = glibc_atanf(fabsf(fd(y, x)));
    }
    switch (m) {
        case 0:
            return z;
        case 1:
            return __uint_as_float(__float_as_uint(z) ^ 0x80000000u);
        case 2:
            return fs(pi, fs(z, pi_lo));
        default:
            return fs(fs(z, pi_lo), pi);
    }
}

// __sincosf_table as laid out in libm.so.6 2.35 (.rodata 0xb30c0), entry 1 computes -cos.
struct SinCosTab {
    double c0, c1, s1, c2, s2, c3, s3, c4;
};
__device__ __forceinline__ SinCosTab sincos_tab(int which) {
    SinCosTab t;
    t.s1 = -0x1.555545995a603p-3;
    t.s2 = 0x1.1107605230bc4p-7;
    t.s3 = -0x1.994eb3774cf24p-13;
    if (which == 0) {
        t.c0 = 0x1p0;
        t.c1 = -0x1.ffffffd0c621cp-2;
        t.c2 = 0x1.55553e1068f19p-5;
        t.c3 = -0x1.6c087e89a359dp-10;
        t.c4 = 0x1.99343027bf8c3p-16;
    } else {
        t.c0 = -0x1p0;
        t.c1 = 0x1.ffffffd0c621cp-2;
        t.c2 = -0x1.55553e1068f19p-5;
        t.c3 = 0x1.6c087e89a359dp-10;
        t.c4 = -0x1.99343027bf8c3p-16;
    }
    return t;
}
// sinf_poly, (n & 1) == 0 branch, as contracted in __sinf_fma / __cosf_fma
__device__ __forceinline__ float sincos_sin_poly(double x, double x2, const SinCosTab& t) {
    const double x3 = __dmul_rn(x, x2);
    const double s1 = fma(x2, t.s3, t.s2);
    const double x5 = __dmul_rn(x2, x3);
    const double s = fma(x3, t.s1, x);
    return static_cast<float>(fma(s1, x5, s));
}
// sinf_poly, (n & 1) == 1 branch
__device__ __forceinline__ float sincos_cos_poly(double x2, const SinCosTab& t) {
    const double x4 = __dmul_rn(x2, x2);
    const double c1 = fma(x2, t.c1, t.c0);
    const double c2 = fma(x2, t.c4, t.c3);
    const double x6 = __dmul_rn(x2, x4);
    const double c = fma(x4, t.c2, c1);
    return static_cast<float>(fma(c2, x6, c));
}
// reduce_fast: x - n*pi/2 with n = round(x*2/pi) taken from bits 24..31 of x*(2/pi)*2^24
__device__ __forceinline__ double sincos_reduce(double x, int* np) {
    const double hpi_inv = 0x1.45f306dc9c883p+23;
    const double hpi = 0x1.921fb54442d18p0;
    const double r = __dmul_rn(x, hpi_inv);
    const int n = (__double2int_rz(r) + 0x800000) >> 24;
    *np = n;
    return fma(-static_cast<double>(n), hpi, x);  // vfnmadd: one rounding
}
__device__ __forceinline__ uint32_t abstop12(float y) { return (__float_as_uint(y) >> 20) & 0x7ffu; }

__device__ inline float glibc_sinf(float y) {
    const double x = static_cast<double>(y);
    const uint32_t top = abstop12(y);
    if (top < 0x3f4u) {  // |y| < pi/4
        if (top < 0x398u) {  // |y| < 2^-12
            return y;
        }
        return sincos_sin_poly(x, __dmul_rn(x, x), sincos_tab(0));
    }
    if (top < 0x42fu) {  // |y| < 120
        int n;
        const double xr = sincos_reduce(x, &n);
        const double x2 = __dmul_rn(xr, xr);
        const SinCosTab t = sincos_tab((n & 2) ? 1 : 0);
        if ((n & 1) == 0) {
            const double sgn = ((n & 3) == 0 || (n & 3) == 3) ? 1.0 : -1.0;
            return sincos_sin_poly(__dmul_rn(xr, sgn), x2, t);
        }
        return sincos_cos_poly(x2, t);
    }
    return __uint_as_float(0x7fc00000u);
}

__device__ inline float glibc_cosf(float y) {
    const double x = static_cast<double>(y);
    const uint32_t top = abstop12(y);
    if (top < 0x3f4u) {
        if (top < 0x398u) {
            return 1.0f;
        }
        return sincos_cos_poly(__dmul_rn(x, x), sincos_tab(0));
    }
    if (top < 0x42fu) {
        int n;
        const double xr = sincos_reduce(x, &n);
        const double x2 = __dmul_rn(xr, xr);
        const SinCosTab t = sincos_tab((n & 2) ? 1 : 0);
        if ((n & 1) != 0) {
            const double sgn = ((n & 3) == 0 || (n & 3) == 3) ? 1.0 : -1.0;
            return sincos_sin_poly(__dmul_rn(xr, sgn), x2, t);
        }
        return sincos_cos_poly(x2, t);
    }
    return __uint_as_float(0x7fc00000u);
}

// cosf(y) and sinf(y) together, without divergent branches: both functions reduce the argument the same
// way and evaluate the same two polynomials -- sinf returns the sine polynomial for even n and the cosine
// polynomial for odd n, cosf the other way round -- so one reduction and one evaluation of each serve both.
// Table entry 1 is entry 0 with c0..c4 negated, and every operation involved is odd-symmetric (fma,
// multiply, the final conversion, all round-to-nearest), so "entry 1" is a sign flip of the entry-0
// result.  The small-argument path (|y| < pi/4) is the general one with n = 0 and no reduction.
// Bit-identical to glibc_cosf / glibc_sinf above (tests/test_parity_gpu.py, test_trig_*).
__device__ __forceinline__ void glibc_sincosf(float y, float* sin_out, float* cos_out) {
    const double x = static_cast<double>(y);
    const uint32_t top = abstop12(y);
    const bool small = top < 0x3f4u;  // |y| < pi/4
    int n;
    const double reduced = sincos_reduce(x, &n);
    n = small ? 0 : n;
    const double xr = small ? x : reduced;
    const double x2 = __dmul_rn(xr, xr);
    const SinCosTab t = sincos_tab(0);
    const double sgn = ((n & 3) == 0 || (n & 3) == 3) ? 1.0 : -1.0;
    const float a = sincos_sin_poly(__dmul_rn(xr, sgn), x2, t);
    const float b0 = sincos_cos_poly(x2, t);
    const float b = (n & 2) ? -b0 : b0;
    float sv = (n & 1) == 0 ? a : b;
    float cv = (n & 1) == 0 ? b : a;
    if (top < 0x398u) {  // |y| < 2^-12
        sv = y;
        cv = 1.0f;
    }
    if (top >= 0x42fu) {  // |y| >= 120: not reached from atan2f (see the header)
        sv = __uint_as_float(0x7fc00000u);
        cv = sv;
    }
    *sin_out = sv;
    *cos_out = cv;
}

// constants::RADIANS_TO_DEGREES / DEGREES_TO_RADIANS, feature_detector.hpp:21-22
__device__ __forceinline__ float rad_to_deg() { return __uint_as_float(0x42652ee1u); }
__device__ __forceinline__ float deg_to_rad() { return __uint_as_float(0x3c8efa35u); }

}  // namespace slamx
