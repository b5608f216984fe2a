// op_rates.hip -- sustained issue rate of the VALU / LDS instructions the frontend kernels lean on
// (gfx950), in Tera lane-ops/s over the whole chip.  Used to choose between equivalent encodings
// (v_perm vs and/shift unpacking, packed u16 vs scalar compares, LDS byte vs dword reads).
// build: hipcc --offload-arch=gfx950 -O3 -o op_rates op_rates.hip ; run: ./op_rates
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int ILP = 8;

#define DEFINE_OP_KERNEL(NAME, ASM)                                                                  \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters, uint32_t seed) {           \
        uint32_t q[ILP];                                                                             \
        _Pragma("unroll") for (int k = 0; k < ILP; ++k) { q[k] = seed * (threadIdx.x + 1) + k * 977u; } \
        uint32_t t = seed ^ (threadIdx.x * 2654435761u);                                             \
        for (int i = 0; i < iters; ++i) {                                                            \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                          \
                _Pragma("unroll") for (int k = 0; k < ILP; ++k) { asm volatile(ASM : "+v"(q[k]) : "v"(t)); } \
            }                                                                                        \
        }                                                                                            \
        uint32_t r = 0;                                                                              \
        _Pragma("unroll") for (int k = 0; k < ILP; ++k) { r += q[k]; }                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                              \
    }

DEFINE_OP_KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
DEFINE_OP_KERNEL(k_add, "v_add_u32 %0, %0, %1")
DEFINE_OP_KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %1")
DEFINE_OP_KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
DEFINE_OP_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %1")
DEFINE_OP_KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 3")
DEFINE_OP_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
DEFINE_OP_KERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %1")
DEFINE_OP_KERNEL(k_pk_sub_sat, "v_pk_sub_u16 %0, %0, %1 clamp")
DEFINE_OP_KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
DEFINE_OP_KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %1")
DEFINE_OP_KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
DEFINE_OP_KERNEL(k_sad_u8, "v_sad_u8 %0, %0, %1, %0")
DEFINE_OP_KERNEL(k_msad_u8, "v_msad_u8 %0, %0, %1, %0")
DEFINE_OP_KERNEL(k_max3, "v_max3_u32 %0, %0, %1, %1")
DEFINE_OP_KERNEL(k_med3, "v_med3_u32 %0, %0, %1, %1")
DEFINE_OP_KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %1")
DEFINE_OP_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEFINE_OP_KERNEL(k_cvt_ubyte, "v_cvt_f32_ubyte1 %0, %0")
DEFINE_OP_KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
DEFINE_OP_KERNEL(k_add_f32, "v_add_f32 %0, %0, %1")
DEFINE_OP_KERNEL(k_dot4_u8, "v_dot4_u32_u8 %0, %0, %1, %0")
DEFINE_OP_KERNEL(k_cmp_cndmask, "v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
DEFINE_OP_KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
// round 3: the byte-parallel (SWAR) FAST pretest leans on these
DEFINE_OP_KERNEL(k_sub, "v_sub_u32 %0, %1, %0")
DEFINE_OP_KERNEL(k_and, "v_and_b32 %0, %1, %0")
DEFINE_OP_KERNEL(k_and_sgpr, "v_and_b32 %0, s4, %0")
DEFINE_OP_KERNEL(k_and_literal, "v_and_b32 %0, 0x3f3f3f3f, %0")
DEFINE_OP_KERNEL(k_lshrrev, "v_lshrrev_b32 %0, 2, %0")
DEFINE_OP_KERNEL(k_min_u32, "v_min_u32 %0, %1, %0")
DEFINE_OP_KERNEL(k_mov_dpp_wave_shr, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
DEFINE_OP_KERNEL(k_add_dpp_row_shr, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
DEFINE_OP_KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0xe8")
DEFINE_OP_KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %1")
DEFINE_OP_KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %1")
DEFINE_OP_KERNEL(k_cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc")
DEFINE_OP_KERNEL(k_fmac_f32, "v_fmac_f32 %0, %1, %1")
DEFINE_OP_KERNEL(k_mul_f32, "v_mul_f32 %0, %1, %0")
DEFINE_OP_KERNEL(k_xor_add_mix, "v_xor_b32 %0, %0, %1\n v_bcnt_u32_b32 %0, %1, %0")

// how the fast VOP2 forms (xor / add / sub / and / shifts / add_f32 / mul_f32) mix with full-rate-only
// instructions: independent streams in different orders, and a dependent chain
#define DEFINE_MIX_KERNEL(NAME, A0, A1, A2, A3)                                                      \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters, uint32_t seed) {           \
        uint32_t q[ILP];                                                                             \
        _Pragma("unroll") for (int k = 0; k < ILP; ++k) { q[k] = seed * (threadIdx.x + 1) + k * 977u; } \
        uint32_t t = seed ^ (threadIdx.x * 2654435761u);                                             \
        for (int i = 0; i < iters; ++i) {                                                            \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                          \
                _Pragma("unroll") for (int k = 0; k < ILP; k += 4) {                                 \
                    asm volatile(A0 "\n" A1 "\n" A2 "\n" A3                                          \
                                 : "+v"(q[k]), "+v"(q[k + 1]), "+v"(q[k + 2]), "+v"(q[k + 3])        \
                                 : "v"(t));                                                          \
                }                                                                                    \
            }                                                                                        \
        }                                                                                            \
        uint32_t r = 0;                                                                              \
        _Pragma("unroll") for (int k = 0; k < ILP; ++k) { r += q[k]; }                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                              \
    }
DEFINE_MIX_KERNEL(k_mix_ffss, "v_xor_b32 %0, %0, %4", "v_xor_b32 %1, %1, %4", "v_bcnt_u32_b32 %2, %4, %2", "v_bcnt_u32_b32 %3, %4, %3")
DEFINE_MIX_KERNEL(k_mix_fsfs, "v_xor_b32 %0, %0, %4", "v_bcnt_u32_b32 %2, %4, %2", "v_xor_b32 %1, %1, %4", "v_bcnt_u32_b32 %3, %4, %3")
DEFINE_MIX_KERNEL(k_mix_fffs, "v_xor_b32 %0, %0, %4", "v_add_u32 %1, %1, %4", "v_sub_u32 %2, %4, %2", "v_bcnt_u32_b32 %3, %4, %3")
DEFINE_MIX_KERNEL(k_mix_ffff, "v_xor_b32 %0, %0, %4", "v_add_u32 %1, %1, %4", "v_sub_u32 %2, %4, %2", "v_and_b32 %3, 0x3f3f3f3f, %3")
DEFINE_MIX_KERNEL(k_dep_chain4, "v_xor_b32 %0, %0, %4", "v_add_u32 %0, %0, %4", "v_sub_u32 %0, %4, %0", "v_xor_b32 %0, %0, %4")
DEFINE_MIX_KERNEL(k_dep_pairs, "v_xor_b32 %0, %0, %4", "v_add_u32 %0, %0, %4", "v_sub_u32 %1, %4, %1", "v_xor_b32 %1, %1, %4")
DEFINE_MIX_KERNEL(k_mix_f_pk, "v_add_u32 %0, %0, %4", "v_pk_min_u16 %1, %1, %4", "v_sub_u32 %2, %4, %2", "v_pk_max_u16 %3, %3, %4")
DEFINE_MIX_KERNEL(k_mix_f_bitop3, "v_add_u32 %0, %0, %4", "v_sub_u32 %1, %4, %1", "v_bitop3_b32 %2, %2, %4, %4 bitop3:0xe8", "v_and_b32 %3, %4, %3")

// 64-bit address arithmetic the compiler emits for pointer + lane offset
__global__ __launch_bounds__(256) void k_mad_i64_i32(uint32_t* out, int iters, uint32_t seed) {
    unsigned long long q[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        q[k] = seed * (threadIdx.x + 1) + k * 977u;
    }
    const uint32_t t = seed ^ (threadIdx.x * 2654435761u);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                asm volatile("v_mad_i64_i32 %0, vcc, %1, %1, %0" : "+v"(q[k]) : "v"(t) : "vcc");
            }
        }
    }
    unsigned long long r = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        r += q[k];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = static_cast<uint32_t>(r ^ (r >> 32));
}
__global__ __launch_bounds__(256) void k_lshl_add_u64(uint32_t* out, int iters, uint32_t seed) {
    unsigned long long q[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        q[k] = seed * (threadIdx.x + 1) + k * 977u;
    }
    const unsigned long long t = (static_cast<unsigned long long>(seed) << 20) ^ (threadIdx.x * 2654435761u);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[k]) : "v"(t));
            }
        }
    }
    unsigned long long r = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        r += q[k];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = static_cast<uint32_t>(r ^ (r >> 32));
}
// dword reads from the LDS at addresses that are 1 mod 4 (every read straddles two banks)
__global__ __launch_bounds__(256) void k_lds_unaligned(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t buf[4096 + 4];
    for (int i = threadIdx.x; i < 4096 + 4; i += 256) {
        buf[i] = seed * i;
    }
    __syncthreads();
    uint32_t acc = 0;
    const char* base = reinterpret_cast<const char*>(buf);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int off = ((threadIdx.x * 4 + r * 4096 / 32 * 4 + i * 16) & (16384 - 4)) + 1;
            uint32_t w;
            __builtin_memcpy(&w, base + off, 4);
            acc += w;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// packed f32 needs register pairs
__global__ __launch_bounds__(256) void k_pk_fma_f32(uint32_t* out, int iters, uint32_t seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 q[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        q[k] = f2{static_cast<float>(seed + threadIdx.x + k), 1.0f};
    }
    const f2 t = {1.0001f, 0.9999f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q[k]) : "v"(t));
            }
        }
    }
    float r = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        r += q[k].x + q[k].y;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = static_cast<uint32_t>(r);
}

__global__ __launch_bounds__(256) void k_fma_f64(uint32_t* out, int iters, uint32_t seed) {
    double q[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        q[k] = static_cast<double>(seed + threadIdx.x + k);
    }
    const double t = 1.0000001;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k = 0; k < ILP; ++k) {
                asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(q[k]) : "v"(t));
            }
        }
    }
    double r = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
        r += q[k];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = static_cast<uint32_t>(r);
}

// LDS read rates: every lane reads its own address (conflict-free dword / byte / b64 / b128)
template <int BYTES>
__global__ __launch_bounds__(256) void k_lds(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) {
        buf[i] = seed * i;
    }
    __syncthreads();
    uint32_t acc = 0;
    const int lane_off = BYTES == 1 ? threadIdx.x : threadIdx.x * (BYTES < 4 ? 4 : BYTES);
    const char* base = reinterpret_cast<const char*>(buf);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int off = (lane_off + r * 4096 / 32 * 4 + i * 16) & (16384 - BYTES);
            if (BYTES == 1) {
                acc += *reinterpret_cast<const uint8_t*>(base + off);
            } else if (BYTES == 4) {
                acc += *reinterpret_cast<const uint32_t*>(base + off);
            } else if (BYTES == 8) {
                const uint2 v = *reinterpret_cast<const uint2*>(base + off);
                acc += v.x ^ v.y;
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(base + off);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int blocks = 256 * 8, threads = 256, iters = 4000;
    uint32_t* d = nullptr;
    hipMalloc(&d, sizeof(uint32_t) * blocks * threads);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto run = [&](const char* name, auto kernel, double ops_per_iter_lane, int it) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d, 50, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d, it, 12345u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double ops = ops_per_iter_lane * it * double(blocks) * threads;
        std::printf("%-22s %8.3f ms  %8.2f T lane-ops/s\n", name, ms, ops / (ms * 1e-3) / 1e12);
        std::fflush(stdout);
    };
    const double per = 4.0 * ILP;
#define RUN(K) run(#K, K, per, iters)
    RUN(k_xor);
    RUN(k_add);
    RUN(k_and_or);
    RUN(k_lshl_or);
    RUN(k_perm);
    RUN(k_alignbyte);
    RUN(k_bfe);
    RUN(k_pk_min_u16);
    RUN(k_pk_sub_sat);
    RUN(k_pk_add_u16);
    RUN(k_pk_mad_u16);
    RUN(k_bcnt);
    RUN(k_sad_u8);
    RUN(k_msad_u8);
    RUN(k_max3);
    RUN(k_med3);
    RUN(k_mad_u24);
    RUN(k_mul_lo);
    RUN(k_cvt_ubyte);
    RUN(k_fma_f32);
    RUN(k_add_f32);
    RUN(k_dot4_u8);
    run("k_cmp_cndmask (2)", k_cmp_cndmask, 2 * per, iters);
    RUN(k_mbcnt);
    RUN(k_sub);
    RUN(k_and);
    RUN(k_and_sgpr);
    RUN(k_and_literal);
    RUN(k_lshrrev);
    RUN(k_min_u32);
    RUN(k_mov_dpp_wave_shr);
    RUN(k_add_dpp_row_shr);
    RUN(k_bitop3);
    RUN(k_or3);
    RUN(k_bfi);
    RUN(k_cndmask_vcc);
    RUN(k_fmac_f32);
    RUN(k_mul_f32);
    run("k_xor_bcnt (2)", k_xor_add_mix, 2 * per, iters);
    RUN(k_mix_ffss);
    RUN(k_mix_fsfs);
    RUN(k_mix_fffs);
    RUN(k_mix_ffff);
    RUN(k_dep_chain4);
    RUN(k_dep_pairs);
    RUN(k_mix_f_pk);
    RUN(k_mix_f_bitop3);
    run("k_pk_fma_f32 (instr)", k_pk_fma_f32, per, iters);
    run("k_fma_f64", k_fma_f64, per, iters);
    run("k_mad_i64_i32", k_mad_i64_i32, per, iters);
    run("k_lshl_add_u64", k_lshl_add_u64, per, iters);
    run("lds read u8", k_lds<1>, 32.0, 1000);
    run("lds read b32", k_lds<4>, 32.0, 1000);
    run("lds read b32 at 1 mod 4", k_lds_unaligned, 32.0, 1000);
    run("lds read b64", k_lds<8>, 32.0, 1000);
    run("lds read b128", k_lds<16>, 32.0, 1000);
    hipFree(d);
    return 0;
}
