// dangx_mhreg.hip -- Metropolis index kernels, register-resident form.  Compiled once per chain mode
// (-DDX_REG_MODE=1..5) so the instantiations build in parallel, and once without it for the dispatcher.
#ifndef DX_NO_VCOEF   // -DDX_NO_VCOEF: plain fma() in the once-per-proposal polynomials, for A/B timing
#define DX_VCOEF 1   // dx_math.h: fma_vc
#endif
#include "dx_chain.h"

#ifdef DX_REG_MODE
namespace {

// Resident waves per SIMD follow the register need: 3 (<= 168 VGPRs) for one plane of <= 10 bands, 2 (<= 256) otherwise.
// Two planes of 20 bands (C5) need ~340 registers in one lane: they run as lane pairs (LP = 2, 10 bands per lane, two
// waves per SIMD); the one-lane form (one wave per SIMD with the overflow in AGPRs) stays selectable with
// DANGX_CHAIN_PAIR=0 for A/B timing.
template <int MODE, int SP, int NB, int LP>
__global__ __launch_bounds__(BLOCK, (SP == 1 && NB <= 10) ? 3 : (SP == 2 && NB >= 20 && LP == 1) ? 1 : 2) void k_index_mh_reg(const Model* __restrict__ Mp, IndexArgs a,
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

#if DX_REG_MODE == 2 || DX_REG_MODE == 4
// index nind and index nind + 1 of one component on the same planes in one launch (dx_chain.h: index_chain_pair); the
// first chain has mode DX_REG_MODE, the second DX_REG_MODE + 1 (mbb: beta then T; log-normal: nu_p then w)
template <int SP, int NB, int LP>
__global__ __launch_bounds__(BLOCK, (SP == 1 && NB <= 10) ? 3 : 2) void k_index_mh_pair(const Model* __restrict__ Mp, IndexArgs a, IndexArgs b,
                                                        unsigned long long* __restrict__ accepted_a, unsigned long long* __restrict__ accepted_b,
                                                        double* __restrict__ chi_partial) {
    const Model& M = *Mp;
    const int tid = threadIdx.x;
    const long long t = (long long)blockIdx.x * BLOCK + tid;
    const int i = (int)(t / LP), half = (int)(t % LP);   // LP = 2: the bands of a pixel over two adjacent lanes (dx_chain.h)
    double chi[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned long long na = 0ull, nb_ = 0ull;
    if (i < M.npix) index_chain_pair<DX_REG_MODE, DX_REG_MODE + 1, SP, NB / LP, LP>(M, a, b, i, half, chi, na, nb_);
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
#endif

}  // namespace

#if DX_REG_MODE == 2 || DX_REG_MODE == 4
#define DX_CATP2(a, b) a##b
#define DX_CATP(a, b) DX_CATP2(a, b)
// accp: two consecutive counters (first sweep, second sweep) or null
bool DX_CATP(dx_launch_mh_pair_mode, DX_REG_MODE)(dangx_ctx* ctx, const IndexArgs& a, const IndexArgs& b, int Sp, unsigned nblk,
                                                 unsigned long long* accp) {
    const int nb = ctx->hm.nbands;
#define DX_LAUNCH_PAIR(SP_, NB_, LP_)                                                                             \
    hipLaunchKernelGGL((k_index_mh_pair<SP_, NB_, LP_>), dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, b, accp, accp ? accp + 1 : nullptr, ctx->partial)
    // nblk counts blocks of BLOCK lanes: BLOCK / 2 pixels for the lane-pair form (two planes of 20 bands)
    if (Sp == 2) {
        if (nb == 10) DX_LAUNCH_PAIR(2, 10, 1); else if (nb == 5) DX_LAUNCH_PAIR(2, 5, 1);
        else if (nb == 20 && dx_mh_reg_lanes(nb, Sp) == 2) DX_LAUNCH_PAIR(2, 20, 2); else return false;
    } else {
        if (nb == 10) DX_LAUNCH_PAIR(1, 10, 1); else if (nb == 5) DX_LAUNCH_PAIR(1, 5, 1); else if (nb == 20) DX_LAUNCH_PAIR(1, 20, 1);
        else if (nb == 3) DX_LAUNCH_PAIR(1, 3, 1); else return false;
    }
#undef DX_LAUNCH_PAIR
    return true;
}
#endif

#define DX_CAT2(a, b) a##b
#define DX_CAT(a, b) DX_CAT2(a, b)
bool DX_CAT(dx_launch_mh_reg_mode, DX_REG_MODE)(dangx_ctx* ctx, const IndexArgs& a, int Sp, unsigned nblk, unsigned long long* accp) {
    const int nb = ctx->hm.nbands;
    if (dx_mh_reg_lanes(nb, Sp) == 2) {  // nblk counts blocks of BLOCK lanes = BLOCK / 2 pixels
        hipLaunchKernelGGL((k_index_mh_reg<DX_REG_MODE, 2, 20, 2>), dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, accp, ctx->partial);
        return true;
    }
#define DX_LAUNCH_REG(SP_, NB_)                                                                                  \
    hipLaunchKernelGGL((k_index_mh_reg<DX_REG_MODE, SP_, NB_, 1>), dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, accp, ctx->partial)
#define DX_REG_NB(SP_)                                                                                           \
    do { if (nb == 10) DX_LAUNCH_REG(SP_, 10); else if (nb == 5) DX_LAUNCH_REG(SP_, 5);                          \
         else if (nb == 3) DX_LAUNCH_REG(SP_, 3); else if (nb == 6) DX_LAUNCH_REG(SP_, 6);                       \
         else if (nb == 8) DX_LAUNCH_REG(SP_, 8); else if (nb == 20) DX_LAUNCH_REG(SP_, 20);                     \
         else return false; } while (0)
    if (Sp == 2) DX_REG_NB(2); else DX_REG_NB(1);
#undef DX_REG_NB
#undef DX_LAUNCH_REG
    return true;
}
#else
bool dx_launch_mh_reg_mode1(dangx_ctx*, const IndexArgs&, int, unsigned, unsigned long long*);
bool dx_launch_mh_reg_mode2(dangx_ctx*, const IndexArgs&, int, unsigned, unsigned long long*);
bool dx_launch_mh_reg_mode3(dangx_ctx*, const IndexArgs&, int, unsigned, unsigned long long*);
bool dx_launch_mh_reg_mode4(dangx_ctx*, const IndexArgs&, int, unsigned, unsigned long long*);
bool dx_launch_mh_reg_mode5(dangx_ctx*, const IndexArgs&, int, unsigned, unsigned long long*);

// lanes per pixel of the register chain: two planes of 20 bands run as lane pairs (DANGX_CHAIN_PAIR=0: one lane)
int dx_mh_reg_lanes(int nb, int Sp) {
    static const int pair = [] { const char* e = getenv("DANGX_CHAIN_PAIR"); return (e && e[0] == '0') ? 0 : 1; }();
    return (pair && nb == 20 && Sp == 2) ? 2 : 1;
}
bool dx_launch_mh_pair_mode2(dangx_ctx*, const IndexArgs&, const IndexArgs&, int, unsigned, unsigned long long*);
bool dx_launch_mh_pair_mode4(dangx_ctx*, const IndexArgs&, const IndexArgs&, int, unsigned, unsigned long long*);
// index a.nind and a.nind + 1 of one component in one launch; false: not covered (the caller makes the two launches)
bool dx_mh_pair_supported(int mode_a, int mode_b, int nb, int Sp) {
    if (!((mode_a == CH_MBB_BETA && mode_b == CH_MBB_T) || (mode_a == CH_LOGN_NUP && mode_b == CH_LOGN_W))) return false;
    return Sp == 2 ? (nb == 10 || nb == 5 || (nb == 20 && dx_mh_reg_lanes(nb, Sp) == 2)) : (nb == 10 || nb == 5 || nb == 20 || nb == 3);
}
bool dx_launch_mh_pair(dangx_ctx* ctx, const IndexArgs& a, const IndexArgs& b, int Sp, unsigned nblk, unsigned long long* accp) {
    if (!dx_mh_pair_supported(a.mode, b.mode, ctx->hm.nbands, Sp)) return false;
    return a.mode == CH_MBB_BETA ? dx_launch_mh_pair_mode2(ctx, a, b, Sp, nblk, accp) : dx_launch_mh_pair_mode4(ctx, a, b, Sp, nblk, accp);
}
bool dx_mh_reg_supported(int mode, int nb) {
    return mode >= CH_POW && mode <= CH_LOGN_W && (nb == 3 || nb == 5 || nb == 6 || nb == 8 || nb == 10 || nb == 20);
}
bool dx_launch_mh_reg(dangx_ctx* ctx, const IndexArgs& a, int Sp, unsigned nblk, unsigned long long* accp) {
    switch (a.mode) {
    case CH_POW: return dx_launch_mh_reg_mode1(ctx, a, Sp, nblk, accp);
    case CH_MBB_BETA: return dx_launch_mh_reg_mode2(ctx, a, Sp, nblk, accp);
    case CH_MBB_T: return dx_launch_mh_reg_mode3(ctx, a, Sp, nblk, accp);
    case CH_LOGN_NUP: return dx_launch_mh_reg_mode4(ctx, a, Sp, nblk, accp);
    case CH_LOGN_W: return dx_launch_mh_reg_mode5(ctx, a, Sp, nblk, accp);
    default: return false;
    }
}
#endif
