// dx_kern_chain.h -- the register-resident Metropolis kernels (templates): instantiated at build time for the band counts of
// the BASELINE configurations (dangx_mhreg.hip) and, for any other model shape, at run time (dangx_rtc.hip compiles this
// header with hiprtc for the (mode, planes, bands, lanes) it meets and caches the code object).
#pragma once
#include "dx_chain.h"

namespace dxk {

// Resident waves per SIMD follow the register need: 3 (<= 168 VGPRs) for one plane of <= 10 bands, 2 (<= 256) otherwise.
// Two planes of 20 bands (C5) need ~340 registers in one lane: they run as lane pairs (LP = 2, 10 bands per lane, two
// waves per SIMD); the one-lane form (one wave per SIMD with the overflow in AGPRs) stays selectable with
// DANGX_CHAIN_PAIR=0 for A/B timing.

// waves per SIMD asked of the register allocator: 3 where one plane of up to 10 bands fits 168 registers, 1 for two planes of
// more than 16 bands in one lane, else 2.  (Two planes of 13 / 15 bands need ~260 / ~280 registers: at two waves they spill
// 6 / 22 and are still 12 % / 8 % faster than at one wave, 17 bands are 2 % slower -- bench.py --nbands N with
// DANGX_RTC_DEFS=-DDX_CHAIN_ONE_WAVE_FROM=12 against 18; even counts from 14 run as lane pairs and never get here.)
#ifndef DX_CHAIN_ONE_WAVE_FROM
#define DX_CHAIN_ONE_WAVE_FROM 16
#endif
#ifndef DX_CHAIN_WAVES
#define DX_CHAIN_WAVES(SP, NB, LP) (((SP) == 1 && (NB) <= 10) ? 3 : ((SP) == 2 && (NB) / (LP) > DX_CHAIN_ONE_WAVE_FROM) ? 1 : 2)
#endif
template <int MODE, int SP, int NB, int LP>
__global__ __launch_bounds__(BLOCK, DX_CHAIN_WAVES(SP, NB, LP)) void k_index_mh_reg(const Model* __restrict__ Mp, IndexArgs a,
                                                        unsigned long long* __restrict__ accepted,
                                                        double* __restrict__ chi_partial) {
    const Model& M = *Mp;
    const int tid = threadIdx.x;
    const long long t = (long long)blockIdx.x * BLOCK + tid;
    const int i = (int)(t / LP), half = (int)(t % LP);
    double chi[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned long long nacc = (i < M.npix) ? index_chain_reg<MODE, SP, NB / LP, LP>(M, a, i, half, chi) : 0ull;
    if (accepted) {
        for (int o = 32; o > 0; o >>= 1) nacc += __shfl_down(nacc, o, 64);
        if ((tid & 63) == 0 && nacc) atomicAdd(accepted, nacc);
    }
    if (chi_partial) {
        __shared__ double sh[4][BLOCK / 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double v = chi[q];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if ((tid & 63) == 0) sh[q][tid >> 6] = v;
        }
        __syncthreads();
        if (tid < 4) {
            double s = 0.0;
            for (int w = 0; w < BLOCK / 64; ++w) s += sh[tid][w];
            chi_partial[(long long)tid * gridDim.x + blockIdx.x] = s;
        }
    }
}

// index nind and index nind + 1 of one component on the same planes in one launch (dx_chain.h: index_chain_pair); the
// first chain has mode MODEA, the second MODEA + 1 (mbb: beta then T; log-normal: nu_p then w)
template <int MODEA, int SP, int NB, int LP>
__global__ __launch_bounds__(BLOCK, DX_CHAIN_WAVES(SP, NB, LP)) void k_index_mh_pair(const Model* __restrict__ Mp, IndexArgs a, IndexArgs b,
                                                        unsigned long long* __restrict__ accepted_a, unsigned long long* __restrict__ accepted_b,
                                                        double* __restrict__ chi_partial) {
    const Model& M = *Mp;
    const int tid = threadIdx.x;
    const long long t = (long long)blockIdx.x * BLOCK + tid;
    const int i = (int)(t / LP), half = (int)(t % LP);   // LP = 2: the bands of a pixel over two adjacent lanes (dx_chain.h)
    double chi[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned long long na = 0ull, nb_ = 0ull;
    if (i < M.npix) index_chain_pair<MODEA, MODEA + 1, SP, NB / LP, LP>(M, a, b, i, half, chi, na, nb_);
    if (accepted_a) {
        for (int o = 32; o > 0; o >>= 1) { na += __shfl_down(na, o, 64); nb_ += __shfl_down(nb_, o, 64); }
        if ((tid & 63) == 0) { if (na) atomicAdd(accepted_a, na); if (nb_) atomicAdd(accepted_b, nb_); }
    }
    if (chi_partial) {
        __shared__ double sh[4][BLOCK / 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double v = chi[q];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if ((tid & 63) == 0) sh[q][tid >> 6] = v;
        }
        __syncthreads();
        if (tid < 4) {
            double s = 0.0;
            for (int w = 0; w < BLOCK / 64; ++w) s += sh[tid][w];
            chi_partial[(long long)tid * gridDim.x + blockIdx.x] = s;
        }
    }
}

}  // namespace dxk
