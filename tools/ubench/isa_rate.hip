// isa_rate.hip -- issue cost of single gfx950 vector instructions (cycles per wave64 instruction per SIMD), measured by
// running 16 independent copies of the instruction in a loop on a grid that keeps every SIMD full (64 waves per SIMD in all).
//   hipcc -O3 --offload-arch=gfx950 -o isa_rate tools/ubench/isa_rate.hip && ./isa_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL_D(NAME, ASM)                                                                    \
    __global__ void NAME(double* out, int iters, double seed) {                                \
        double a[16], b = seed, c = seed * 0.5;                                                \
        for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;                            \
        for (int it = 0; it < iters; ++it) {                                                   \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                      \
        double s = 0; for (int i = 0; i < 16; ++i) s += a[i];                                  \
        if (s == 12345.678) out[0] = s;                                                        \
    }
#define KERNEL_I(NAME, ASM)                                                                    \
    __global__ void NAME(double* out, int iters, double seed) {                                \
        unsigned a[16], b = (unsigned)seed | 1u, c = 77u;                                      \
        for (int i = 0; i < 16; ++i) a[i] = (unsigned)seed + i + threadIdx.x;                  \
        for (int it = 0; it < iters; ++it) {                                                   \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                      \
        unsigned s = 0; for (int i = 0; i < 16; ++i) s += a[i];                                \
        if (s == 0x12345678u && seed == 3.25) out[0] = s;                                      \
    }
// 64-bit destination from 32-bit sources
#define KERNEL_Q(NAME, ASM)                                                                    \
    __global__ void NAME(double* out, int iters, double seed) {                                \
        unsigned long long a[16]; unsigned b = (unsigned)seed | 1u, c = 77u;                   \
        for (int i = 0; i < 16; ++i) a[i] = (unsigned)seed + i + threadIdx.x;                  \
        for (int it = 0; it < iters; ++it) {                                                   \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"); \
        }                                                                                      \
        unsigned long long s = 0; for (int i = 0; i < 16; ++i) s += a[i];                      \
        if (s == 0x12345678u && seed == 3.25) out[0] = (double)s;                              \
    }

KERNEL_D(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL_D(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL_D(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL_D(k_fmac_f64, "v_fmac_f64 %0, %1, %2")
KERNEL_D(k_mov_b64, "v_mov_b64 %0, %1")
KERNEL_D(k_rndne_f64, "v_rndne_f64 %0, %0")
KERNEL_D(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_D(k_rsq_f64, "v_rsq_f64 %0, %0")
KERNEL_D(k_sqrt_f64, "v_sqrt_f64 %0, %0")
KERNEL_D(k_ldexp_f64, "v_ldexp_f64 %0, %0, 3")
KERNEL_D(k_frexp_mant_f64, "v_frexp_mant_f64 %0, %0")
KERNEL_D(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
KERNEL_D(k_div_scale_f64, "v_div_scale_f64 %0, vcc, %0, %1, %2")
KERNEL_D(k_div_fmas_f64, "v_div_fmas_f64 %0, %0, %1, %2")
KERNEL_D(k_div_fixup_f64, "v_div_fixup_f64 %0, %0, %1, %2")
KERNEL_D(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL_I(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL_I(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL_I(k_xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL_I(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL_I(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL_I(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_I(k_cndmask64, "v_cndmask_b32 %0, %0, %1, s[10:11]")
KERNEL_I(k_cndmask_e64vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc")
KERNEL_I(k_cndmask_cmp, "v_cmp_lt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc")
KERNEL_I(k_cmp_u32, "v_cmp_lt_u32 vcc, %1, %0")
KERNEL_I(k_cndmask_sdwa, "v_cndmask_b32 %0, %0, %2, vcc")
KERNEL_I(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL_I(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL_I(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL_I(k_bfe, "v_bfe_u32 %0, %0, 3, 5")
KERNEL_I(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
KERNEL_I(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL_I(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL_I(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL_Q(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL_Q(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %0")

// f64 <-> i32 conversions need mixed register widths
__global__ void k_cvt_i32_f64(double* out, int iters, double seed) {
    double a[16]; int r[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; r[i] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(r[i]) : "v"(a[i]));
    }
    int s = 0; for (int i = 0; i < 16; ++i) s += r[i];
    if (s == 0x12345678 && seed == 3.25) out[0] = s;
}
__global__ void k_cvt_f64_u32(double* out, int iters, double seed) {
    double a[16]; unsigned r[16];
    for (int i = 0; i < 16; ++i) { a[i] = 0; r[i] = (unsigned)seed + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f64_u32 %0, %1" : "+v"(a[i]) : "v"(r[i]));
    }
    double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.678) out[0] = s;
}
__global__ void k_readlane(double* out, int iters, double seed) {
    unsigned a[16]; unsigned acc = 0;
    for (int i = 0; i < 16; ++i) a[i] = (unsigned)seed + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { unsigned s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(a[i])); acc += s; }
    }
    if (acc == 0x12345678u && seed == 3.25) out[0] = acc;
}

// select patterns: what follows a v_cmp that wrote vcc
#define KERNEL_SEL(NAME, ASM)                                                                  \
    __global__ void NAME(double* out, int iters, double seed) {                                \
        unsigned a[16], d[16], b = (unsigned)seed | 1u, c = 77u;                               \
        for (int i = 0; i < 16; ++i) { a[i] = (unsigned)seed + i + threadIdx.x; d[i] = a[i] * 3u; } \
        for (int it = 0; it < iters; ++it) {                                                   \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]), "+v"(d[i]) : "v"(b), "v"(c) : "vcc"); \
        }                                                                                      \
        unsigned s = 0; for (int i = 0; i < 16; ++i) s += a[i] + d[i];                         \
        if (s == 0x12345678u && seed == 3.25) out[0] = s;                                      \
    }
KERNEL_SEL(k_sel_cmp_2cnd, "v_cmp_lt_u32 vcc, %2, %0\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc")
KERNEL_SEL(k_sel_cmp_gap_cnd, "v_cmp_lt_u32 vcc, %2, %0\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL_SEL(k_sel_cmp_gap4_cnd, "v_cmp_lt_u32 vcc, %2, %0\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %2\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %2\n\tv_xor_b32 %1, %1, %3\n\tv_xor_b32 %1, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL_SEL(k_sel_smov_cnd, "s_mov_b64 vcc, s[10:11]\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL_SEL(k_sel_cmpe64_cnd, "v_cmp_lt_u32_e64 s[10:11], %2, %0\n\tv_cndmask_b32_e64 %0, %0, %2, s[10:11]\n\tv_cndmask_b32_e64 %1, %1, %3, s[10:11]")
KERNEL_SEL(k_sel_2xor, "v_xor_b32 %1, %1, %3\n\tv_xor_b32 %0, %0, %2")

typedef void (*kern_t)(double*, int, double);
struct Entry { const char* name; kern_t k; };

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    double* out; hipMalloc(&out, 64);
    Entry tab[] = {
        {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_fmac_f64", k_fmac_f64}, {"v_mov_b64", k_mov_b64},
        {"v_rndne_f64", k_rndne_f64}, {"v_ldexp_f64", k_ldexp_f64}, {"v_frexp_mant_f64", k_frexp_mant_f64}, {"v_cvt_i32_f64", k_cvt_i32_f64},
        {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cmp_lt_f64", k_cmp_f64}, {"v_rcp_f64", k_rcp_f64}, {"v_rsq_f64", k_rsq_f64},
        {"v_sqrt_f64", k_sqrt_f64}, {"v_div_scale_f64", k_div_scale_f64}, {"v_div_fmas_f64", k_div_fmas_f64}, {"v_div_fixup_f64", k_div_fixup_f64},
        {"v_pk_fma_f32", k_pk_fma_f32}, {"v_fma_f32", k_fma_f32},
        {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mul_u32_u24", k_mul_u32_u24},
        {"v_mad_u32_u24", k_mad_u32_u24}, {"v_xor_b32", k_xor_b32}, {"v_add_u32", k_add_u32}, {"v_mov_b32", k_mov_b32},
        {"v_cndmask_b32", k_cndmask}, {"v_cndmask_b32 e64", k_cndmask64}, {"v_cndmask e64 vcc", k_cndmask_e64vcc}, {"v_cmp+v_cndmask", k_cndmask_cmp}, {"v_cmp_lt_u32", k_cmp_u32}, {"v_cndmask other src", k_cndmask_sdwa}, {"cmp;cnd;cnd", k_sel_cmp_2cnd}, {"cmp;2 xor;cnd", k_sel_cmp_gap_cnd}, {"cmp;6 xor;cnd", k_sel_cmp_gap4_cnd}, {"s_mov vcc;cnd", k_sel_smov_cnd},
        {"cmp e64;cnd e64 x2", k_sel_cmpe64_cnd}, {"xor;xor", k_sel_2xor}, {"v_and_or_b32", k_and_or}, {"v_lshl_add_u32", k_lshl_add}, {"v_perm_b32", k_perm}, {"v_bfe_u32", k_bfe}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_readlane_b32", k_readlane},
    };
    const int iters = 2000;
    const int waves_per_simd = 64;                // 64 four-wave blocks per CU in total: the chip stays full whatever the placement
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("device: %s, %d CUs, nominal %.2f GHz; cycles = time * nominal clock / (instructions per wave * waves per SIMD); the first line also pays the clock ramp\n", p.name, cus, ghz);
    for (auto& t : tab) {
        hipLaunchKernelGGL(t.k, grid, block, 0, 0, out, 200, 1.5);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(t.k, grid, block, 0, 0, out, iters, 1.5);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 16 * waves_per_simd;
        printf("%-22s %7.3f ms  %6.2f cycles/instr\n", t.name, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd);
    }
    return 0;
}
