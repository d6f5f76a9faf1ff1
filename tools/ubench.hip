// ubench.hip -- per-operation VALU throughput of the sampler's building blocks on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 -I include -I dang_amd/csrc -o gpurun_out/ubench tools/ubench.hip
// Each kernel runs ITER dependent-chain iterations of one op on 4 independent chains per lane, on a grid that
// fills the chip at 3 waves/SIMD; the figure printed is lane-operations per second for the whole device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "dx_rng.h"
using namespace dx;

// candidates
__device__ __forceinline__ double sin_2pi_fast(double u) {
    // sin(2 pi u), u in [0,1): n = rint(4u), f = 2u - n/2 in [-1/4, 1/4] (exact), then sin or cos of pi*f
    const double n = rint(4.0 * u);
    const double f = fma(-0.5, n, 2.0 * u);
    const int q = (int)n;
    const double x = 3.14159265358979311600e+00 * f, z = x * x;
    const bool c = q & 1;
    // fdlibm __kernel_sin S1..S6 / __kernel_cos C1..C6 on |x| <= pi/4
    const double k5 = c ? -1.13596475577881948265e-11 : 1.58969099521155010221e-10;
    const double k4 = c ? 2.08757232129817482790e-09 : -2.50507602534068634195e-08;
    const double k3 = c ? -2.75573143513906633035e-07 : 2.75573137070700676789e-06;
    const double k2 = c ? 2.48015872894767294178e-05 : -1.98412698298579493134e-04;
    const double k1 = c ? -1.38888888888741095749e-03 : 8.33333333332248946124e-03;
    const double k0 = c ? 4.16666666666666019037e-02 : -1.66666666666666324348e-01;
    const double p = fma(fma(fma(fma(fma(k5, z, k4), z, k3), z, k2), z, k1), z, k0);
    const double s = c ? fma(z * z, p, fma(-0.5, z, 1.0)) : fma(x * z, p, x);
    return (q & 2) ? -s : s;
}
__device__ __forceinline__ double sqrt_pos(double x) {  // x > 0, normal, far from over/underflow
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    return fma(d, h, g);
}
__device__ __forceinline__ double log_fast(double x) {  // log_pos with the division replaced by rcp + one Newton step
    constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                     Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                     Lg7 = 1.479819860511658591e-01;
    unsigned long long ix = (unsigned long long)__double_as_longlong(x);
    int k = (int)(ix >> 52) - 1023;
    ix = (ix & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m = __longlong_as_double((long long)ix);
    if (m > 1.4142135623730951) { m *= 0.5; k += 1; }
    const double f = m - 1.0, den = 2.0 + f;
    double rc = __builtin_amdgcn_rcp(den);
    rc = fma(fma(-den, rc, 1.0), rc, rc);
    double s = f * rc;
    s = fma(fma(-den, s, f), rc, s);
    const double dk = (double)k, z = s * s, w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1, hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

constexpr int ITER = 2048, CH = 4;

template <int OP>
__global__ __launch_bounds__(256, 3) void k(double* out, double seed) {
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    double x[CH];
    for (int c = 0; c < CH; ++c) x[c] = seed + 1e-3 * c + 1e-9 * gid;
    uint32_t w[CH];
    for (int c = 0; c < CH; ++c) w[c] = gid * 4 + c;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (OP == 0) x[c] = fma(x[c], 0.999999, 1e-7);
            else if (OP == 1) x[c] = exp(-x[c]) + 0.5;                 // arguments stay O(1)
            else if (OP == 2) x[c] = log_pos(x[c] + 1.5) + 0.7;
            else if (OP == 3) x[c] = sin_2pi(x[c] * 0.37) + 1.1;
            else if (OP == 4) x[c] = sqrt(x[c] + 2.0);
            else if (OP == 5) x[c] = 1.7 / (x[c] + 0.9);
            else if (OP == 6) { w[c] = __umulhi(0xD2511F53u, w[c]) ^ (0xCD9E8D57u * w[c]); }
            else if (OP == 7) { uint32_t o[4]; philox4x32_10(w[c], it, 7, 9, 11, 13, o); w[c] = o[0] ^ o[1] ^ o[2] ^ o[3]; }
            else if (OP == 8) { double u1, u2, u3; uniform3(11, 13, w[c], it, u1, u2, u3); x[c] += rand_normal(0.0, 1.0, u1, u2) + u3; w[c] += 64; }
            else if (OP == 9) { w[c] = (w[c] ^ 0x5bd1e995u) * (w[c] | 1u); }   // v_mul_lo_u32 + 2 simple
            else if (OP == 11) x[c] = sin_2pi_fast(x[c] * 0.37 - floor(x[c] * 0.37)) + 1.1;
            else if (OP == 12) x[c] = sin_2pi(x[c] * 0.37 - floor(x[c] * 0.37)) + 1.1;
            else if (OP == 13) x[c] = sqrt_pos(x[c] + 2.0);
            else if (OP == 14) x[c] = log_fast(x[c] + 1.5) + 0.7;
            else if (OP == 15) { unsigned long long z = ((unsigned long long)w[c] << 32 | it) + 0x9E3779B97F4A7C15ull * (it + 1);
                                 z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
                                 w[c] = (uint32_t)z ^ (uint32_t)(z >> 32); }
            else if (OP == 10) { w[c] = (w[c] ^ (w[c] >> 7)) + 0x9E3779B9u; }  // plain 32-bit VALU
        }
    }
    double s = 0.0;
    for (int c = 0; c < CH; ++c) s += x[c] + (double)w[c];
    out[gid] = s;
}

template <int OP>
void run(const char* name, double* out, int nblk, double opscale) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(nblk), dim3(256), 0, 0, out, 0.25);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<OP>, dim3(nblk), dim3(256), 0, 0, out, 0.25);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double ops = 3.0 * nblk * 256.0 * ITER * CH * opscale;
    printf("%-28s %10.3f ms  %10.2f G lane-ops/s\n", name, ms / 3, ops / (ms * 1e-3) / 1e9);
}

int main() {
    const int nblk = 256 * 12;  // 256 CUs x 12 blocks of 4 waves = 3 waves/SIMD x 4 rounds
    double* out;
    if (hipMalloc(&out, sizeof(double) * nblk * 256) != hipSuccess) { printf("no device\n"); return 1; }
    run<0>("fma f64", out, nblk, 1);
    run<1>("exp f64 (ocml)", out, nblk, 1);
    run<2>("log_pos", out, nblk, 1);
    run<3>("sin_2pi", out, nblk, 1);
    run<4>("sqrt f64", out, nblk, 1);
    run<5>("div f64", out, nblk, 1);
    run<6>("umulhi+mullo+xor", out, nblk, 1);
    run<7>("philox4x32-10", out, nblk, 1);
    run<8>("uniform3+rand_normal", out, nblk, 1);
    run<9>("mul_lo_u32+add", out, nblk, 1);
    run<10>("xor/shift/add u32", out, nblk, 1);
    run<12>("sin_2pi (ocml sinpi)", out, nblk, 1);
    run<11>("sin_2pi_fast", out, nblk, 1);
    run<13>("sqrt_pos", out, nblk, 1);
    run<14>("log_fast", out, nblk, 1);
    run<15>("splitmix64", out, nblk, 1);
    hipFree(out);
    return 0;
}
