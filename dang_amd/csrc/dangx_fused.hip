// dangx_fused.hip -- the amplitude solve of a (group, flag) and the FIRST index sweep on the same planes in one launch.
//
// For a model without global-amplitude members every step of a Gibbs iteration is independent per pixel, so the
// reference's order "all pixels: amplitude solve; all pixels: index chain" (src/dang_cg_mod.f90:142-177,
// src/dang_sample_mod.f90:21-86) gives, pixel by pixel, the same numbers as "per pixel: solve, then chain".  The two
// kernels that do this separately (k_amp_reg, k_index_mh_reg) are bound by different things -- the solve waits on the map
// loads with the vector units half idle, the chain is vector-issue bound and barely touches memory -- and read the same
// 2*nb map planes; fused, the solve's arithmetic runs inside an issue-bound kernel (its stalls are covered by other
// waves' chains), the maps are read once, the 1/rms are computed once, and the SEDs the solve evaluated for the group's
// members are the "other components" the chain has to remove from the data, so they are not evaluated again.
//
// Arithmetic: the solve is k_amp_reg's (same SED expressions, same band order of the rank-1 updates, same Cholesky), the
// chain is index_chain_reg's from the staged planes on (chain_finish); the staged planes are formed by the same
// expressions in the same order as index_chain_reg forms them.  dangx_amp_index_sample(...) therefore equals
// dangx_amp_sample(...) followed by dangx_index_sample(...) bit for bit (tests/test_gpu_fused.py), and falls back to
// exactly that pair of calls for every configuration this kernel does not cover.
// Compiled once per chain mode (-DDX_REG_MODE=1..3), and once without it for the dispatcher.
#ifndef DX_NO_VCOEF
#define DX_VCOEF 1   // dx_math.h: fma_vc
#endif
#include "dx_host.h"
#include "dx_kern_fused.h"

// LDS of one block: the constant table and, per lane, the SED columns of the varying members over the lane's bands plus
// their two index values
static size_t fused_lds(int ng, int nb, int nv, int lanes) {
    return ((size_t)(TROWS * ng + 3) * nb + (size_t)nv * (nb / lanes + 2) * BLOCK) * sizeof(double);
}

#ifdef DX_REG_MODE
#define DX_CAT2(a, b) a##b
#define DX_CAT(a, b) DX_CAT2(a, b)
template <int NB, int NG, int LP>
static void launch_fused_case(dangx_ctx* ctx, const GroupArgs& ga, const FusedArgs& fa, const IndexArgs& a, int Sp, unsigned nblk,
                              unsigned long long* accp) {
    const size_t ldsz = fused_lds(NG, NB, fa.nv, LP);
    if (Sp == 2)
        hipLaunchKernelGGL((dxk::k_amp_index<DX_REG_MODE, 2, NB, NG, LP>), dim3(nblk), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, ga, fa, a, ctx->counters, accp, ctx->partial);
    else
        hipLaunchKernelGGL((dxk::k_amp_index<DX_REG_MODE, 1, NB, NG, LP>), dim3(nblk), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, ga, fa, a, ctx->counters, accp, ctx->partial);
}
bool DX_CAT(dx_launch_fused_mode, DX_REG_MODE)(dangx_ctx* ctx, const GroupArgs& ga, const FusedArgs& fa, const IndexArgs& a, int Sp,
                                              int lanes, unsigned nblk, unsigned long long* accp) {
    const int nb = ctx->hm.nbands;
    // the instantiated (band count, group size, lanes) triples: keep fused_builtin below in step
    if (lanes == 1 && nb == 10 && ga.ng == 4) launch_fused_case<10, 4, 1>(ctx, ga, fa, a, Sp, nblk, accp);        // C3
    else if (lanes == 1 && nb == 10 && ga.ng == 3) launch_fused_case<10, 3, 1>(ctx, ga, fa, a, Sp, nblk, accp);
    else if (lanes == 1 && nb == 5 && ga.ng == 3) launch_fused_case<5, 3, 1>(ctx, ga, fa, a, Sp, nblk, accp);     // C2
    else if (lanes == 1 && nb == 3 && ga.ng == 2) launch_fused_case<3, 2, 1>(ctx, ga, fa, a, Sp, nblk, accp);     // C1
#if DX_REG_MODE == 1
    else if (lanes == 2 && nb == 20 && ga.ng == 6) launch_fused_case<20, 6, 2>(ctx, ga, fa, a, Sp, nblk, accp);   // C5 (first sweep: synchrotron beta)
#endif
    else return false;
    return true;
}
#else
bool dx_launch_fused_mode1(dangx_ctx*, const GroupArgs&, const FusedArgs&, const IndexArgs&, int, int, unsigned, unsigned long long*);
bool dx_launch_fused_mode2(dangx_ctx*, const GroupArgs&, const FusedArgs&, const IndexArgs&, int, int, unsigned, unsigned long long*);
bool dx_launch_fused_mode3(dangx_ctx*, const GroupArgs&, const FusedArgs&, const IndexArgs&, int, int, unsigned, unsigned long long*);

static bool fused_builtin(int mode, int nb, int ng, int lanes) {
    if (lanes == 1) return (nb == 10 && (ng == 4 || ng == 3)) || (nb == 5 && ng == 3) || (nb == 3 && ng == 2);
    return mode == CH_POW && nb == 20 && ng == 6;
}

// could a fused launch exist for this (mode, bands, members)?  (dx_fused_lanes decides for the actual model and planes)
bool dx_fused_supported(int mode, int nb, int ng) {
    if (!(mode >= CH_POW && mode <= CH_MBB_T) || ng < 1 || ng > 6) return false;
    return fused_builtin(mode, nb, ng, 1) || fused_builtin(mode, nb, ng, 2) || dx_rtc_enabled();
}

static std::string fused_name(int mode, int Sp, int nb, int ng, int lanes) {
    return "dxk::k_amp_index<" + std::to_string(mode) + ", " + std::to_string(Sp) + ", " + std::to_string(nb) + ", " + std::to_string(ng) + ", " +
           std::to_string(lanes) + ">";
}

// the members' roles in the fused kernel; false: this model does not take it
static bool fused_args(dangx_ctx* ctx, const GroupArgs& ga, const IndexArgs& a, FusedArgs& fa) {
    fa.nv = 0; fa.gself = -1;
    unsigned planes = 0;
    for (int k = a.s1; k <= a.s2; ++k) planes |= 1u << (k - 1);
    for (int g = 0; g < MAXG; ++g) { fa.vslot[g] = -1; fa.vcomp[g] = 0; fa.vtype[g] = 0; }
    for (int g = 0; g < ga.ng; ++g) {
        const Comp& c = ctx->hm.comp[ga.gc[g]];
        if (ga.gc[g] == a.comp) fa.gself = g;
        const unsigned cm = (unsigned)c.const_planes & planes;
        if (cm != 0 && cm != planes) return false;  // constant on one plane only: the two kernels would take different routes
        if (cm != planes) {  // varies on the planes of the launch: evaluated per pixel
            if (c.type != DANGX_POWERLAW && c.type != DANGX_MBB && c.type != DANGX_FREEFREE && c.type != DANGX_LOGNORMAL) return false;
            fa.vcomp[fa.nv] = (signed char)g; fa.vtype[fa.nv] = (signed char)c.type;
            fa.vslot[g] = (signed char)fa.nv++;
        }
    }
    if (fa.gself < 0) return false;
    // band calibration in use (a gain /= 1 or an offset /= 0): the T-plane data are rescaled differently by the solve
    // (d / gain, src/dang_cg_mod.f90:371) and by the chain ((d - offset) / gain, src/dang_sample_mod.f90:174); the T launch reads
    // both from the block's table, the Q / U / Q+U launches are not concerned
    fa.cal = 0;
    if (a.s1 == 1)
        for (int j = 0; j < ctx->hm.nbands; ++j)
            if (ctx->hm.gain[j] != 1.0 || ctx->hm.offset[j] != 0.0) fa.cal = 1;
    return true;
}

// Lanes per pixel of the fused launch for the pending solve `ga` and the sweep `a`, 0 when the pair takes the two launches.
// One lane while the chain's planes and the solve's normal equations fit two waves per SIMD (one plane: up to 16 bands, two
// planes: up to 10) and the members' SED columns two blocks per CU; else lane pairs (even band count, half the bands and
// half the columns per lane: C5's 20 bands and 6 members).  A shape without a built-in instantiation is specialised HERE
// (hiprtc or the disk cache), so that the launch that follows cannot fail: the caller sizes its grid by this answer.
int dx_fused_lanes(dangx_ctx* ctx, const GroupArgs& ga, const IndexArgs& a, int Sp) {
    if (!(a.mode >= CH_POW && a.mode <= CH_MBB_T) || ga.ng < 1 || ga.ng > 6) return 0;
    FusedArgs fa;
    if (!fused_args(ctx, ga, a, fa)) return 0;
    const int nb = ctx->hm.nbands, ng = ga.ng, cap = (Sp == 2) ? 10 : 16;
    int lanes = 0;
    if (nb <= cap && fused_lds(ng, nb, fa.nv, 1) <= 80u * 1024u) lanes = 1;
    // lane pairs only where the chain itself runs as lane pairs (two planes of more than 12 bands): on one plane the chain's
    // one-lane form does not repeat the per-proposal work on a second lane, and the two launches are faster than a paired
    // fused one (C5, T plane: 7.4 + 12.5 ms against 21.1 ms)
    else if (nb % 2 == 0 && nb / 2 <= cap && dx_mh_reg_lanes(nb, Sp) == 2 && fused_lds(ng, nb, fa.nv, 2) <= 80u * 1024u) lanes = 2;
    if (!lanes) return 0;
    if (fused_builtin(a.mode, nb, ng, lanes)) return lanes;
    return dx_rtc_get(ctx, "dx_kern_fused.h", fused_name(a.mode, Sp, nb, ng, lanes)) ? lanes : 0;
}

// ga: the pending amplitude solve; a: the index sweep that follows it on the same planes; lanes: dx_fused_lanes' answer
bool dx_launch_fused(dangx_ctx* ctx, const GroupArgs& ga, const IndexArgs& a, int Sp, int lanes, unsigned nblk, unsigned long long* accp) {
    FusedArgs fa;
    if (lanes < 1 || !fused_args(ctx, ga, a, fa)) return false;
    bool done = false;
    switch (a.mode) {
    case CH_POW: done = dx_launch_fused_mode1(ctx, ga, fa, a, Sp, lanes, nblk, accp); break;
    case CH_MBB_BETA: done = dx_launch_fused_mode2(ctx, ga, fa, a, Sp, lanes, nblk, accp); break;
    case CH_MBB_T: done = dx_launch_fused_mode3(ctx, ga, fa, a, Sp, lanes, nblk, accp); break;
    default: return false;
    }
    if (done) return true;
    const int nb = ctx->hm.nbands, ng = ga.ng;
    hipFunction_t fn = dx_rtc_get(ctx, "dx_kern_fused.h", fused_name(a.mode, Sp, nb, ng, lanes));
    if (!fn) return false;
    const Model* dm = ctx->dm;
    GroupArgs gg = ga;
    IndexArgs aa = a;
    unsigned long long* bad = ctx->counters;
    double* part = ctx->partial;
    void* args[] = {&dm, &gg, &fa, &aa, &bad, &accp, &part};
    return dx_rtc_launch(ctx, fn, nblk, fused_lds(ng, nb, fa.nv, lanes), args) == 0;
}
#endif
