// stream_planes.hip -- what HBM delivers for the amplitude kernel's ACCESS PATTERN with no arithmetic: every thread reads
// one double from each of 2*nb planes that lie 3*npix doubles apart (band-major maps [nb][3][npix], one plane of each
// band), 4 more index/mask planes, and writes 4 planes -- against the same number of bytes read from one contiguous
// range.  hipcc -O3 --offload-arch=gfx950 -o stream_planes tools/ubench/stream_planes.hip && ./stream_planes
#include <hip/hip_runtime.h>
#include <cstdio>

#define NB 10
__global__ void k_planes(const double* __restrict__ sig, const double* __restrict__ rms, const double* __restrict__ aux,
                         double* __restrict__ out, long long npix, int plane) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const long long bs = 3 * npix;
    double d[NB], r[NB], x[4];
#pragma unroll
    for (int j = 0; j < NB; ++j) { d[j] = sig[j * bs + plane * npix + i]; r[j] = rms[j * bs + plane * npix + i]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = aux[q * npix + i];
    double s = 0.0, t = 0.0;
#pragma unroll
    for (int j = 0; j < NB; ++j) { s += d[j] * r[j]; t += d[j] - r[j]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q * npix + i] = s * x[q] + t;
}
__global__ void k_contig(const double* __restrict__ in, double* __restrict__ out, long long npix) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 2 * NB + 4; ++j) s += in[j * npix + i];   // 24 planes back to back: one contiguous range overall
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q * npix + i] = s + q;
}

int main() {
    const long long npix = 12582912;
    double *sig, *rms, *aux, *out;
    hipMalloc(&sig, sizeof(double) * NB * 3 * npix); hipMalloc(&rms, sizeof(double) * NB * 3 * npix);
    hipMalloc(&aux, sizeof(double) * 4 * npix); hipMalloc(&out, sizeof(double) * 4 * npix);
    hipMemset(sig, 0, sizeof(double) * NB * 3 * npix); hipMemset(rms, 0, sizeof(double) * NB * 3 * npix);
    hipMemset(aux, 0, sizeof(double) * 4 * npix);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (2.0 * NB + 4 + 4) * 8 * npix;
    for (int bsz : {256, 512, 1024}) {
        dim3 grid((unsigned)((npix + bsz - 1) / bsz)), block(bsz);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_planes, grid, block, 0, 0, sig, rms, aux, out, npix, 0);
        hipEventRecord(e0);
        for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(k_planes, grid, block, 0, 0, sig, rms, aux, out, npix, rep % 3);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("planes  block %4d: %.3f ms  %.2f TB/s (%.2f GB per launch)\n", bsz, ms, bytes / ms * 1e-9, bytes * 1e-9);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_contig, grid, block, 0, 0, sig, out, npix);
        hipEventRecord(e0);
        for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(k_contig, grid, block, 0, 0, sig + (rep % 3) * npix, out, npix);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("contig  block %4d: %.3f ms  %.2f TB/s\n", bsz, ms, bytes / ms * 1e-9);
    }
    return 0;
}
