// dx_rng.h -- counter-based random streams for the sampler kernels.
//
// The reference draws from the compiler's RANDOM_NUMBER after an unseeded
// RANDOM_SEED() (src/dang.f90:67; rand_normal src/dang_util_mod.f90:100-110),
// which is irreproducible and thread-order dependent.  The MI355X path keys
// every draw by (seed, stream, GLOBAL pixel, draw slot) through Philox4x32-10,
// so a sample does not depend on launch geometry or on how pixels are sharded.
//
//   counter = { pixel[31:0], draw ^ (pixel[63:32] << 16), stream[31:0], stream[63:32] }
//   key     = { seed[31:0], seed[63:32] }
//   u1 = words(0,1), u2 = words(2,3), each mapped to (0,1) with 53 bits.
// Draw slots:
//   amplitude phase, reference fluctuation term : draw = map number k (1..3)
//   amplitude phase, textbook fluctuation term  : draw = k + 4*(band+1)
//   index phase, step l (1..nsample)            : draw = l, ONE Philox call per step (uniform3):
//        proposal normal from u1 = words(0,1) [53 bits] and u2 = word 2 [32 bits];
//        accept uniform u3 = word 3 [32 bits]  (each 32-bit word maps to (w + 0.5) * 2^-32)
//   index phase, lnl_type=='prior' draw         : draw = 0 (uniform2)
#pragma once
#include "dx_rtc_compat.h"
#include "dx_math.h"
#include "dx_model.h"

namespace dx {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
#ifndef DX_PHILOX_MULHI  // one v_mad_u64_u32 per product (4.3 cycles) instead of v_mul_hi_u32 + v_mul_lo_u32 (4.1 + 4.0)
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#else
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
#endif
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    const unsigned long long x = ((unsigned long long)hi << 32) | lo;
    return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ void uniform2(unsigned long long seed, unsigned long long stream,
                                         unsigned long long pix, uint32_t draw, double& u1, double& u2) {
    uint32_t o[4];
    philox4x32_10((uint32_t)pix, draw ^ ((uint32_t)(pix >> 32) << 16), (uint32_t)stream, (uint32_t)(stream >> 32),
                  (uint32_t)seed, (uint32_t)(seed >> 32), o);
    u1 = u53(o[0], o[1]);
    u2 = u53(o[2], o[3]);
}

__device__ __forceinline__ double u32(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }

__device__ __forceinline__ void uniform3(unsigned long long seed, unsigned long long stream, unsigned long long pix,
                                         uint32_t draw, double& u1, double& u2, double& u3) {
    uint32_t o[4];
    philox4x32_10((uint32_t)pix, draw ^ ((uint32_t)(pix >> 32) << 16), (uint32_t)stream, (uint32_t)(stream >> 32),
                  (uint32_t)seed, (uint32_t)(seed >> 32), o);
    u1 = u53(o[0], o[1]);
    u2 = u32(o[2]);
    u3 = u32(o[3]);
}

// rand_normal, src/dang_util_mod.f90:100-110 (Box-Muller, sine branch only):
//   r = (-2 log u1)**0.5 ; theta = 2 pi u2 ; c = mean + stdev*r*sin(theta)
// sin(2 pi u2) is evaluated as sinpi(2 u2) (no rounding of theta); log by log_pos (u1 is normal).
__device__ __forceinline__ double rand_normal(double mean, double stdev, double u1, double u2) {
    // u1 in (0,1]: u53 rounds to exactly 1.0 for the one word pattern 2^53 - 1 (probability 2^-53 per draw); the argument
    // is then -0.0, where the reciprocal-square-root seed is -inf and the Goldschmidt steps give NaN -- sqrt(-0) = -0 in
    // the reference's arithmetic, i.e. a zero deviate: select it
    const double arg = -2.0 * log_pos(u1);
    const double r = (arg > 0.0) ? fast_sqrt(arg) : 0.0;
    // one explicit fma: "x + (mean + a*b)" at a call site with mean = 0 could otherwise be contracted into fma(a, b, x) in
    // one kernel and left as a rounded product plus an addition in another (see log_pos)
    return fma(stdev * r, sin_2pi(u2), mean);
}

}  // namespace dx
