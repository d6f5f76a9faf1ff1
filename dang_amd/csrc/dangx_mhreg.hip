// dangx_mhreg.hip -- Metropolis index kernels, register-resident form.  Compiled once per chain mode
// (-DDX_REG_MODE=1..5) so the instantiations build in parallel, and once without it for the dispatcher.
#ifndef DX_NO_VCOEF   // -DDX_NO_VCOEF: plain fma() in the once-per-proposal polynomials, for A/B timing
#define DX_VCOEF 1   // dx_math.h: fma_vc
#endif
#include "dx_host.h"
#include "dx_kern_chain.h"

#ifdef DX_REG_MODE

#if DX_REG_MODE == 2 || DX_REG_MODE == 4
#define DX_CATP2(a, b) a##b
#define DX_CATP(a, b) DX_CATP2(a, b)
// accp: two consecutive counters (first sweep, second sweep) or null
bool DX_CATP(dx_launch_mh_pair_mode, DX_REG_MODE)(dangx_ctx* ctx, const IndexArgs& a, const IndexArgs& b, int Sp, unsigned nblk,
                                                 unsigned long long* accp) {
    const int nb = ctx->hm.nbands;
#define DX_LAUNCH_PAIR(SP_, NB_, LP_)                                                                             \
    hipLaunchKernelGGL((dxk::k_index_mh_pair<DX_REG_MODE, SP_, NB_, LP_>), dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, b, accp, accp ? accp + 1 : nullptr, ctx->partial)
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
        if (nb != 20) return false;      // other lane-pair shapes are specialised at run time
        hipLaunchKernelGGL((dxk::k_index_mh_reg<DX_REG_MODE, 2, 20, 2>), dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, accp, ctx->partial);
        return true;
    }
#define DX_LAUNCH_REG(SP_, NB_)                                                                                  \
    hipLaunchKernelGGL((dxk::k_index_mh_reg<DX_REG_MODE, SP_, NB_, 1>), dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, accp, ctx->partial)
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

// lanes per pixel of the register chain: two planes of more than 12 bands do not fit one lane's registers at two waves per
// SIMD; an even band count then runs as lane pairs (C5: 20 bands, 10 per lane).  DANGX_CHAIN_PAIR=0: always one lane.
int dx_mh_reg_lanes(int nb, int Sp) {
    static const int pair = [] { const char* e = getenv("DANGX_CHAIN_PAIR"); return (e && e[0] == '0') ? 0 : 1; }();
    return (pair && Sp == 2 && nb > 12 && nb % 2 == 0) ? 2 : 1;
}
bool dx_launch_mh_pair_mode2(dangx_ctx*, const IndexArgs&, const IndexArgs&, int, unsigned, unsigned long long*);
bool dx_launch_mh_pair_mode4(dangx_ctx*, const IndexArgs&, const IndexArgs&, int, unsigned, unsigned long long*);

static std::string chain_name(const char* kernel, int mode, int Sp, int nb, int lanes) {
    return std::string("dxk::") + kernel + "<" + std::to_string(mode) + ", " + std::to_string(Sp) + ", " + std::to_string(nb) + ", " +
           std::to_string(lanes) + ">";
}

// index a.nind and a.nind + 1 of one component in one launch; false: not covered (the caller makes the two launches)
bool dx_mh_pair_supported(int mode_a, int mode_b, int nb, int Sp) {
    if (!((mode_a == CH_MBB_BETA && mode_b == CH_MBB_T) || (mode_a == CH_LOGN_NUP && mode_b == CH_LOGN_W))) return false;
    const bool built = Sp == 2 ? (nb == 10 || nb == 5 || (nb == 20 && dx_mh_reg_lanes(nb, Sp) == 2)) : (nb == 10 || nb == 5 || nb == 20 || nb == 3);
    return built || dx_rtc_enabled();
}
bool dx_launch_mh_pair(dangx_ctx* ctx, const IndexArgs& a, const IndexArgs& b, int Sp, unsigned nblk, unsigned long long* accp) {
    if (!dx_mh_pair_supported(a.mode, b.mode, ctx->hm.nbands, Sp)) return false;
    if (a.mode == CH_MBB_BETA ? dx_launch_mh_pair_mode2(ctx, a, b, Sp, nblk, accp) : dx_launch_mh_pair_mode4(ctx, a, b, Sp, nblk, accp)) return true;
    // no built-in instantiation for this band count: specialise the same template now
    hipFunction_t fn = dx_rtc_get(ctx, "dx_kern_chain.h", chain_name("k_index_mh_pair", a.mode, Sp, ctx->hm.nbands, dx_mh_reg_lanes(ctx->hm.nbands, Sp)));
    if (!fn) return false;
    const Model* dm = ctx->dm;
    IndexArgs aa = a, bb = b;
    unsigned long long *acc_a = accp, *acc_b = accp ? accp + 1 : nullptr;
    double* part = ctx->partial;
    void* args[] = {&dm, &aa, &bb, &acc_a, &acc_b, &part};
    return dx_rtc_launch(ctx, fn, nblk, 0, args) == 0;
}
// Is the register chain available for (mode, bands, planes) on this context?  Built-in instantiations: yes; any other band
// count: specialised NOW (hiprtc, or the disk cache), so that a later launch cannot fail -- the caller sizes its grid and its
// chi^2 buffers for the form it is told about.
bool dx_mh_reg_supported(dangx_ctx* ctx, int mode, int nb, int Sp) {
    if (!(mode >= CH_POW && mode <= CH_LOGN_W)) return false;
    const int lanes = dx_mh_reg_lanes(nb, Sp);
    if (lanes == 2 ? nb == 20 : (nb == 3 || nb == 5 || nb == 6 || nb == 8 || nb == 10 || nb == 20)) return true;
    return dx_rtc_get(ctx, "dx_kern_chain.h", chain_name("k_index_mh_reg", mode, Sp, nb, lanes)) != nullptr;
}
bool dx_launch_mh_reg(dangx_ctx* ctx, const IndexArgs& a, int Sp, unsigned nblk, unsigned long long* accp) {
    bool done = false;
    switch (a.mode) {
    case CH_POW: done = dx_launch_mh_reg_mode1(ctx, a, Sp, nblk, accp); break;
    case CH_MBB_BETA: done = dx_launch_mh_reg_mode2(ctx, a, Sp, nblk, accp); break;
    case CH_MBB_T: done = dx_launch_mh_reg_mode3(ctx, a, Sp, nblk, accp); break;
    case CH_LOGN_NUP: done = dx_launch_mh_reg_mode4(ctx, a, Sp, nblk, accp); break;
    case CH_LOGN_W: done = dx_launch_mh_reg_mode5(ctx, a, Sp, nblk, accp); break;
    default: return false;
    }
    if (done) return true;
    hipFunction_t fn = dx_rtc_get(ctx, "dx_kern_chain.h", chain_name("k_index_mh_reg", a.mode, Sp, ctx->hm.nbands, dx_mh_reg_lanes(ctx->hm.nbands, Sp)));
    if (!fn) return false;
    const Model* dm = ctx->dm;
    IndexArgs aa = a;
    double* part = ctx->partial;
    void* args[] = {&dm, &aa, &accp, &part};
    return dx_rtc_launch(ctx, fn, nblk, 0, args) == 0;
}
#endif
