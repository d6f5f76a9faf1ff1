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

#ifdef DX_REG_MODE
#define DX_CAT2(a, b) a##b
#define DX_CAT(a, b) DX_CAT2(a, b)
template <int NB, int NG>
static void launch_fused_case(dangx_ctx* ctx, const GroupArgs& ga, const FusedArgs& fa, const IndexArgs& a, int Sp, unsigned nblk,
                              unsigned long long* accp) {
    const size_t ldsz = ((size_t)(TROWS * NG + 3) * NB + (size_t)fa.nv * (NB + 2) * BLOCK) * sizeof(double);
    if (Sp == 2)
        hipLaunchKernelGGL((dxk::k_amp_index<DX_REG_MODE, 2, NB, NG>), dim3(nblk), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, ga, fa, a, ctx->counters, accp, ctx->partial);
    else
        hipLaunchKernelGGL((dxk::k_amp_index<DX_REG_MODE, 1, NB, NG>), dim3(nblk), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, ga, fa, a, ctx->counters, accp, ctx->partial);
}
bool DX_CAT(dx_launch_fused_mode, DX_REG_MODE)(dangx_ctx* ctx, const GroupArgs& ga, const FusedArgs& fa, const IndexArgs& a, int Sp,
                                              unsigned nblk, unsigned long long* accp) {
    const int nb = ctx->hm.nbands;
    // the instantiated (band count, group size) pairs: keep dx_fused_supported below in step
    if (nb == 10 && ga.ng == 4) launch_fused_case<10, 4>(ctx, ga, fa, a, Sp, nblk, accp);        // C3
    else if (nb == 10 && ga.ng == 3) launch_fused_case<10, 3>(ctx, ga, fa, a, Sp, nblk, accp);
    else if (nb == 5 && ga.ng == 3) launch_fused_case<5, 3>(ctx, ga, fa, a, Sp, nblk, accp);     // C2
    else if (nb == 3 && ga.ng == 2) launch_fused_case<3, 2>(ctx, ga, fa, a, Sp, nblk, accp);     // C1
    else return false;
    return true;
}
#else
bool dx_launch_fused_mode1(dangx_ctx*, const GroupArgs&, const FusedArgs&, const IndexArgs&, int, unsigned, unsigned long long*);
bool dx_launch_fused_mode2(dangx_ctx*, const GroupArgs&, const FusedArgs&, const IndexArgs&, int, unsigned, unsigned long long*);
bool dx_launch_fused_mode3(dangx_ctx*, const GroupArgs&, const FusedArgs&, const IndexArgs&, int, unsigned, unsigned long long*);

// which (band count, group size) the fused kernel is instantiated for
bool dx_fused_supported(int mode, int nb, int ng) {
    if (!(mode >= CH_POW && mode <= CH_MBB_T)) return false;
    if ((nb == 10 && (ng == 4 || ng == 3)) || (nb == 5 && ng == 3) || (nb == 3 && ng == 2)) return true;
    return dx_rtc_enabled() && nb <= 16 && ng >= 1 && ng <= 6;   // any other shape: specialised at run time (dx_launch_fused decides)
}

// ga: the pending amplitude solve; a: the index sweep that follows it on the same planes.  false: not covered.
bool dx_launch_fused(dangx_ctx* ctx, const GroupArgs& ga, const IndexArgs& a, int Sp, unsigned nblk, unsigned long long* accp) {
    FusedArgs fa;
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
            if (c.type != DANGX_POWERLAW && c.type != DANGX_MBB) return false;
            fa.vcomp[fa.nv] = (signed char)g; fa.vtype[fa.nv] = (signed char)c.type;
            fa.vslot[g] = (signed char)fa.nv++;
        }
    }
    if (fa.gself < 0) return false;
    // band calibration in use (a gain /= 1 or an offset /= 0): the T-plane data are rescaled differently by the solve
    // (d / gain) and by the chain ((d - offset) / gain); the kernel carries neither, such models take the two launches
    for (int j = 0; j < ctx->hm.nbands; ++j)
        if (ctx->hm.gain[j] != 1.0 || ctx->hm.offset[j] != 0.0) return false;
    bool done = false;
    switch (a.mode) {
    case CH_POW: done = dx_launch_fused_mode1(ctx, ga, fa, a, Sp, nblk, accp); break;
    case CH_MBB_BETA: done = dx_launch_fused_mode2(ctx, ga, fa, a, Sp, nblk, accp); break;
    case CH_MBB_T: done = dx_launch_fused_mode3(ctx, ga, fa, a, Sp, nblk, accp); break;
    default: return false;
    }
    if (done) return true;
    // No built-in instantiation for (bands, members): specialise the template now -- where the fused form pays: the chain's
    // planes and the solve's normal equations must fit the registers of two waves per SIMD (one plane: up to 16 bands, two
    // planes: up to 10), and the members' SED columns two blocks per CU.  Anything else takes the two launches.
    const int nb = ctx->hm.nbands, ng = ga.ng;
    if (!dx_rtc_enabled() || ng < 1 || ng > 6 || nb > (Sp == 2 ? 10 : 16)) return false;
    const size_t ldsz = ((size_t)(TROWS * ng + 3) * nb + (size_t)fa.nv * (nb + 2) * BLOCK) * sizeof(double);
    if (ldsz > 80u * 1024u) return false;
    hipFunction_t fn = dx_rtc_get(ctx, "dx_kern_fused.h", "dxk::k_amp_index<" + std::to_string(a.mode) + ", " + std::to_string(Sp) + ", " +
                                                            std::to_string(nb) + ", " + std::to_string(ng) + ">");
    if (!fn) return false;
    const Model* dm = ctx->dm;
    GroupArgs gg = ga;
    FusedArgs ff = fa;
    IndexArgs aa = a;
    unsigned long long* bad = ctx->counters;
    double* part = ctx->partial;
    void* args[] = {&dm, &gg, &ff, &aa, &bad, &accp, &part};
    return dx_rtc_launch(ctx, fn, nblk, ldsz, args) == 0;
}
#endif
