// dx_math.h -- fp64 math helpers sized for the sampler kernels (gfx950).
//
// The Gibbs path is bound by fp64 transcendentals, and the device library's general-purpose
// log/sin/pow carry argument-range and special-case handling the path never needs
// (measured in ISA instructions on gfx950: exp 19 f64 ops, log 76, sin 108, pow 148).
// The helpers below are valid on the ranges the kernels use and keep <= 1-2 ulp accuracy.
#pragma once
#include "dx_rtc_compat.h"

namespace dx {

// r*p + c with the coefficient in a VECTOR register and the three-address encoding.  For a polynomial that is evaluated
// once per loop iteration the compiler keeps the coefficients in vector registers anyway (the scalar registers are taken
// by the band constants) and then emits  v_mov_b64 tmp, c ; v_fmac_f64 tmp, r, p  -- two instructions per Horner step
// (27 such copies in a Metropolis proposal); v_fma_f64 dst, r, p, c reads the same register without the copy.
// Only translation units that define DX_VCOEF (the register-chain kernels) get it: elsewhere (amplitude kernels, one
// normal deviate per tile) the longer live ranges cost spills and there is no loop to win in.
__device__ __forceinline__ double fma_vc(double r, double p, double c) {
#ifndef DX_VCOEF
    return fma(r, p, c);
#else
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(r), "v"(p), "v"(c));
    return d;
#endif
}

// 1/x for finite normal x, <= 1 ulp (v_rcp_f64 is good to 2^-23; each Newton step squares the error)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
// 1/sqrt(x) for finite normal x > 0: Goldschmidt iteration on g ~ sqrt(x), h ~ 1/(2 sqrt(x))
__device__ __forceinline__ double fast_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-g, h, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-g, h, 0.5);
    h = fma(h, r, h);
    return h + h;
}

// sqrt(x) for finite normal x > 0 by the same Goldschmidt iteration (<= 1 ulp; 8 vector instructions, the IEEE sqrt
// sequence is 15)
__device__ __forceinline__ double fast_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-g, h, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-g, h, 0.5);
    return fma(g, r, g);
}


// Natural logarithm for x > 0, finite and NORMAL (uniform deviates in (0,1), frequencies,
// temperatures).  fdlibm's e_log.c scheme: x = 2^k * (1+f), sqrt(1/2) <= 1+f < sqrt(2),
// s = f/(2+f), log(1+f) = f - (f^2/2 - s*(f^2/2 + R(s^2))), |error| < 1 ulp.
__device__ __forceinline__ double log_pos(double x) {
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                     Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                     Lg7 = 1.479819860511658591e-01;
    unsigned long long ix = (unsigned long long)__double_as_longlong(x);
    int k = (int)(ix >> 52) - 1023;
    ix = (ix & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;  // m in [1,2)
    double m = __longlong_as_double((long long)ix);
    if (m > 1.4142135623730951) { m *= 0.5; k += 1; }
    const double f = m - 1.0;
    const double s = f * fast_rcp(2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    // every sum of two products below is written as ONE explicit fma: left to the compiler, "a*b + c*d" may be contracted
    // either way, and two kernels that inline this function then differ in the last bit for a few arguments in a million
    // (found by comparing the fused and the separate launches over 12.6 M pixels)
    const double t1 = w * fma_vc(w, fma_vc(w, Lg6, Lg4), Lg2);
    const double R = fma(z, fma_vc(w, fma_vc(w, fma_vc(w, Lg7, Lg5), Lg3), Lg1), t1);
    const double hfsq = (0.5 * f) * f;
    const double inner = fma(s, hfsq + R, dk * ln2_lo);
    return fma(dk, ln2_hi, -((hfsq - inner) - f));
}

// exp(x) with the device library's own reduction and polynomial (ocml expD: n = rint(x log2e), r = x - n ln2 in two
// pieces, degree-11 Horner, ldexp) WITHOUT its two range selects (x > 1024 -> inf, x < -1075 -> 0): ldexp saturates to
// inf / 0 by itself, so every finite argument gives the library's result bit for bit; only x = +-inf differ (NaN instead
// of inf / 0), and the kernels never compare such a value with an outcome that depends on it (an accept test
// `exp(diff) > u` is false either way; `diff >= 0` is tested first).  Six of the library routine's 22 vector
// instructions are those selects: the Metropolis chains are vector-issue bound, so this is 1/8 of their proposal loop.
// exp_sat: for arguments of any size (log-normal SED with a narrow width, Planck function at low temperature).
__device__ __forceinline__ double exp_sat(double x) {
    const double dn = rint(x * 0x1.71547652b82fep+0);
    const double r = fma(dn, -0x1.abc9e3b39803fp-56, fma(dn, -0x1.62e42fefa39efp-1, x));
    double p = fma(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
    p = fma(r, p, 0x1.71dee623fde64p-19);
    p = fma(r, p, 0x1.a01997c89e6b0p-16);
    p = fma(r, p, 0x1.a01a014761f6ep-13);
    p = fma(r, p, 0x1.6c16c1852b7b0p-10);
    p = fma(r, p, 0x1.1111111122322p-7);
    p = fma(r, p, 0x1.55555555502a1p-5);
    p = fma(r, p, 0x1.5555555555511p-3);
    p = fma(r, p, 0x1.000000000000bp-1);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    return ldexp(p, (int)dn);
}

// exp_nr: the same reduction and polynomial for |x| < 1e9 -- the SED arguments beta ln(nu/nu_ref) and h nu / (k T) with the
// clamp of mbb_z below.  n = rint(x log2e) comes out of ONE fma with 1.5 * 2^52: the integer then sits in the low word of the
// sum (two's complement), which saves the v_rndne and the v_cvt_i32 of the form above -- ten exponentials per Metropolis
// proposal, 2 % of the proposal loop.  (The single rounding of x log2e + 2^52 can pick the neighbouring n when x log2e is
// within an ulp of a half-integer; r is then just beyond ln2/2 and the result differs in the last bit at most.)
// -DDX_EXP_RINT restores exp_sat everywhere.
__device__ __forceinline__ double exp_nr(double x) {
#ifdef DX_EXP_RINT
    return exp_sat(x);
#else
    const double z = fma(x, 0x1.71547652b82fep+0, 0x1.8p+52);
    const double dn = z - 0x1.8p+52;
    const double r = fma(dn, -0x1.abc9e3b39803fp-56, fma(dn, -0x1.62e42fefa39efp-1, x));
    double p = fma(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
    p = fma(r, p, 0x1.71dee623fde64p-19);
    p = fma(r, p, 0x1.a01997c89e6b0p-16);
    p = fma(r, p, 0x1.a01a014761f6ep-13);
    p = fma(r, p, 0x1.6c16c1852b7b0p-10);
    p = fma(r, p, 0x1.1111111122322p-7);
    p = fma(r, p, 0x1.55555555502a1p-5);
    p = fma(r, p, 0x1.5555555555511p-3);
    p = fma(r, p, 0x1.000000000000bp-1);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    return ldexp(p, (int)(unsigned int)(unsigned long long)__double_as_longlong(z));  // the low word of z
#endif
}

// the same routine for a call site that runs once per loop iteration (the accept test): coefficients by fma_vc
__device__ __forceinline__ double exp_nr_v(double x) {
    const double dn = rint(x * 0x1.71547652b82fep+0);
    const double r = fma(dn, -0x1.abc9e3b39803fp-56, fma(dn, -0x1.62e42fefa39efp-1, x));
    double p = fma_vc(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
    p = fma_vc(r, p, 0x1.71dee623fde64p-19);
    p = fma_vc(r, p, 0x1.a01997c89e6b0p-16);
    p = fma_vc(r, p, 0x1.a01a014761f6ep-13);
    p = fma_vc(r, p, 0x1.6c16c1852b7b0p-10);
    p = fma_vc(r, p, 0x1.1111111122322p-7);
    p = fma_vc(r, p, 0x1.55555555502a1p-5);
    p = fma_vc(r, p, 0x1.5555555555511p-3);
    p = fma_vc(r, p, 0x1.000000000000bp-1);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    return ldexp(p, (int)dn);
}

// sin(2*pi*u) for u in [0,1): exact range reduction on u (t = 2u, k = rint(t) in {0,1,2}, r = t - k in [-1/2, 1/2] are
// all exact; sin(pi t) = (-1)^k sin(pi r)) and ONE odd polynomial for sin(pi r) on [-1/2, 1/2] (Taylor to r^21, truncation
// 1e-18; measured |error| <= 3.4e-16 against extended precision, libm's sin(2*pi*u) has 7e-16 from rounding 2*pi*u).
// 20 vector instructions and 11 coefficients; the library's sinpi evaluates a sine AND a cosine polynomial and selects
// (~36 instructions, twice the coefficients -- scalar registers the chain kernels do not have to spare).
__device__ __forceinline__ double sin_2pi(double u) {
    const double t = u + u;
    const double k = rint(t);
    double r = t - k;
    r = (k == 1.0) ? -r : r;
    const double z = r * r;
    double p = fma_vc(z, 0x1.2877020d52cf0p-31, -0x1.8a404211f9547p-26);
    p = fma_vc(z, p, 0x1.aaec32af93359p-21);
    p = fma_vc(z, p, -0x1.6fadb9f155744p-16);
    p = fma_vc(z, p, 0x1.e8f434d018d63p-12);
    p = fma_vc(z, p, -0x1.e3074fde8871fp-8);
    p = fma_vc(z, p, 0x1.50783487ee782p-4);
    p = fma_vc(z, p, -0x1.32d2cce62bd86p-1);
    p = fma_vc(z, p, 0x1.466bc6775aae2p+1);
    p = fma_vc(z, p, -0x1.4abbce625be53p+2);
    p = fma_vc(z, p, 0x1.921fb54442d18p+1);
    return r * p;
}

}  // namespace dx
