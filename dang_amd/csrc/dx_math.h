// dx_math.h -- fp64 math helpers sized for the sampler kernels (gfx950).
//
// The Gibbs path is bound by fp64 transcendentals, and the device library's general-purpose
// log/sin/pow carry argument-range and special-case handling the path never needs
// (measured in ISA instructions on gfx950: exp 19 f64 ops, log 76, sin 108, pow 148).
// The helpers below are valid on the ranges the kernels use and keep <= 1-2 ulp accuracy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dx {

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
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// sin(2*pi*u) for u in [0,1): exact range reduction on u (no 2*pi*u rounding)
__device__ __forceinline__ double sin_2pi(double u) { return sinpi(2.0 * u); }

}  // namespace dx
