// dx_ampdata.h -- the data preparation of compute_rhs shared by the amplitude-phase translation units.
#pragma once
#include "dx_host.h"

namespace {

// data(i,k,j) of compute_rhs (src/dang_cg_mod.f90:367-378, 427-443): the band map with every
// component that is not solved for removed.  a.oc lists only the components whose amplitude
// plane may be non-zero (the host tracks all-zero planes; subtracting 0*sed is skipped, which
// differs from the reference only if that sed is not finite).
__device__ __forceinline__ double remove_others(const Model& M, const GroupArgs& a, int i, int k, int j, double d) {
    for (int o = 0; o < a.no; ++o) {
        const Comp& c = M.comp[a.oc[o]];
        const double amp = c.amp[(long long)(k - 1) * M.npix + i];
        double t0, t1;
        load_theta(M, c, i, k, t0, t1);
        d = d - comp_signal(M, c, i, k, j, amp, sed_prep(c, t0, t1));
    }
    // "Still subtract templates which exist but may not be fit here" (:445-460): EVERY template / monopole of the
    // model, member of this group or not, is removed on its unfitted bands -- for a non-member a second time
    for (int w = 0; w < a.nuc; ++w) {
        const Comp& c = M.comp[a.uc[w]];
        if (!((c.corr_mask >> j) & 1)) d = d - comp_signal(M, c, i, k, j, 0.0, Prep{0, 0, 0});
    }
    return d;
}
__device__ __forceinline__ double rhs_data(const Model& M, const GroupArgs& a, int i, int k, int j) {
    double d = M.sig[((long long)j * M.nmaps + (k - 1)) * M.npix + i];
    if (k == 1) d = d / M.gain[j];
    return remove_others(M, a, i, k, j, d);
}


// rows of the global-amplitude members (shared by the mixed CG operators and the Schur solve)
__device__ __forceinline__ int gl_nplanes(const Comp& c, int flag) { return (c.type == DANGX_TEMPLATE && (flag & DANGX_FLAG_QU)) ? 2 : 1; }

// every thread of the block calls this; thread 0 writes the block's sum
__device__ __forceinline__ void block_row_sum(double v, int row, double* rowpartial, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += sh[w];
        rowpartial[(long long)row * gridDim.x + blockIdx.x] = t;
    }
    __syncthreads();
}

struct UnitId { bool in; int p, i, k; bool msk; };
__device__ __forceinline__ UnitId unit_of(const Model& M, int flag) {
    UnitId q;
    const long long u = (long long)blockIdx.x * BLOCK + threadIdx.x;
    q.in = u < (long long)flag_nplanes(flag) * M.npix;
    q.p = q.in ? (int)(u / M.npix) : 0;
    q.i = q.in ? (int)(u - (long long)q.p * M.npix) : 0;
    q.k = flag_map(flag, q.p);
    q.msk = !q.in || is_masked(M.mask[q.i]);
    return q;
}


}  // namespace
