// dangx_planeset.hip -- launcher of k_plane_set (dx_kern_planeset.h): a CG group's solve and / or the index sweeps on its planes
// in one launch, the residual of the plane set kept in registers between the sweeps.  Built-in instantiations: the BASELINE
// shapes; other shapes and sweep sequences are specialised at run time (dangx_rtc.hip).
#ifndef DX_NO_VCOEF
#define DX_VCOEF 1   // dx_math.h: fma_vc
#endif
#include "dx_host.h"
#include "dx_kern_planeset.h"

// rows of the lane's column: the members' SEDs for the solve, then 1 / rms of the lane's bands on the Sp planes
static size_t planeset_lds(int ng, int nb, int nv, int lanes, int Sp) {
    const int rows = nv > Sp ? nv : Sp;
    return ((size_t)(TROWS * ng + 3) * nb + (size_t)rows * (nb / lanes) * BLOCK) * sizeof(double);
}

// the sweep items as template arguments: chain mode + 8 for an item that carries the component's next index too, 0 = none
static void item_codes(const SweepList& sl, int code[4]) {
    for (int q = 0; q < 4; ++q) code[q] = (q < sl.n) ? sl.s[q].mode + 8 * sl.s[q].pair : 0;
}
static std::string planeset_name(int Sp, int nb, int ng, int lanes, int solve, const SweepList& sl, int bp) {
    int c[4];
    item_codes(sl, c);
    std::string s = "dxk::k_plane_set<" + std::to_string(Sp) + ", " + std::to_string(nb) + ", " + std::to_string(ng) + ", " + std::to_string(lanes) +
                    ", " + std::to_string(solve);
    for (int q = 0; q < 4; ++q) s += ", " + std::to_string(c[q]);
    return s + ", " + std::to_string(bp) + ">";
}

// bandpass-integrated bands in the model: the BP form of the kernel (one lane per pixel; power-law / mbb chains)
static int planeset_bp(dangx_ctx* ctx) {
    for (int j = 0; j < ctx->hm.nbands; ++j)
        if (ctx->hm.band[j].n != 0) return 1;
    return 0;
}

// the BASELINE shapes: C5 (20 bands, 6 members, lane pairs; sweeps synchrotron beta | dust beta + T | AME nu_p) and C3 / C2 / C1
// (10 bands x 4 members, 5 x 3, 3 x 2; one lane per pixel; sweeps synchrotron beta | dust beta + T).  Built in: the whole
// iteration of a plane set (solve + sweeps); for C3 also the two halves the two-call seam launches (the solve alone with its
// chi^2, the sweeps alone).  Everything else is compiled on first use.
static bool planeset_builtin(int nb, int ng, int lanes, int solve, const SweepList& sl) {
    int c[4];
    item_codes(sl, c);
    if (nb == 20 && ng == 6 && lanes == 2) return solve && c[0] == CH_POW && c[1] == CH_MBB_BETA + 8 && c[2] == CH_LOGN_NUP && c[3] == 0;
    if (lanes == 1 && nb == 10 && ng == 4 && solve && sl.n == 0) return true;
    if (lanes == 1 && ((nb == 10 && ng == 4) || (solve && ((nb == 5 && ng == 3) || (nb == 3 && ng == 2)))))
        return c[0] == CH_POW && c[1] == CH_MBB_BETA + 8 && c[2] == 0 && c[3] == 0;
    return false;
}

template <int NB, int NG, int LP, int SOLVE, int C0, int C1, int C2>
static void launch_builtin(dangx_ctx* ctx, const GroupArgs& ga, const FusedArgs& fa, const SweepList& sl, int Sp, unsigned nblk, size_t ldsz,
                           unsigned long long* accp) {
    if (Sp == 2)
        hipLaunchKernelGGL((dxk::k_plane_set<2, NB, NG, LP, SOLVE, C0, C1, C2, 0>), dim3(nblk), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, ga, fa, sl, ctx->counters, accp, ctx->partial);
    else
        hipLaunchKernelGGL((dxk::k_plane_set<1, NB, NG, LP, SOLVE, C0, C1, C2, 0>), dim3(nblk), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, ga, fa, sl, ctx->counters, accp, ctx->partial);
}

// members' roles (as the fused kernel's), and the conditions the kernel relies on
static bool planeset_args(dangx_ctx* ctx, const GroupArgs& ga, const SweepList& sl, FusedArgs& fa) {
    fa.nv = 0; fa.gself = -1;
    unsigned planes = 0;
    for (int k = sl.s1; k <= sl.s2; ++k) planes |= 1u << (k - 1);
    for (int g = 0; g < MAXG; ++g) { fa.vslot[g] = -1; fa.vcomp[g] = 0; fa.vtype[g] = 0; }
    for (int g = 0; g < ga.ng; ++g) {
        const int l = ga.gc[g];
        const Comp& c = ctx->hm.comp[l];
        const unsigned cm = (unsigned)c.const_planes & planes;
        if (cm != 0 && cm != planes) return false;
        if (cm != planes) {
            if (c.type != DANGX_POWERLAW && c.type != DANGX_MBB && c.type != DANGX_FREEFREE && c.type != DANGX_LOGNORMAL) return false;
            // Q+U: the SED columns are evaluated once for both planes, so every index map of a varying member must be equal on
            // them (as it is once a Q+U sweep has written it; dangx_core.hip keeps track)
            if (sl.s2 > sl.s1 && (ctx->qu_equal[l] & ((1u << c.nind) - 1u)) != ((1u << c.nind) - 1u)) return false;
            fa.vcomp[fa.nv] = (signed char)g; fa.vtype[fa.nv] = (signed char)c.type;
            fa.vslot[g] = (signed char)fa.nv++;
        }
    }
    // band calibration (src/dang_cg_mod.f90:371, src/dang_sample_mod.f90:174) only rescales the temperature plane: the Q / U / Q+U
    // launches never see it, the T launch reads gain and offset from the block's table
    fa.cal = 0;
    if (sl.s1 == 1)
        for (int j = 0; j < ctx->hm.nbands; ++j)
            if (ctx->hm.gain[j] != 1.0 || ctx->hm.offset[j] != 0.0) fa.cal = 1;
    return true;
}

// Lanes per pixel of the plane-set launch, 0 when this (group, sweeps) takes the separate launches.  One lane up to 16 bands on
// one plane and 13 on two (11-13: two waves per SIMD with a few spilled registers, 10 % faster than the separate launches; 14 in
// one lane is 5 % slower than as lane pairs, 15 slower than the separate launches), lane pairs above (even counts).
// solve: the launch starts with the group's amplitude solve (sl.n = 0: nothing else); 0: the sweeps alone.
// Specialises the kernel when there is no built-in instantiation.
int dx_planeset_lanes(dangx_ctx* ctx, const GroupArgs& ga, const SweepList& sl, int solve) {
    static const bool enabled = [] { const char* e = getenv("DANGX_PLANESET"); return !(e && e[0] == '0'); }();
    // DANGX_PLANESET=pairs: only for the shapes that run as lane pairs (A/B switch)
    static const bool small_too = [] { const char* e = getenv("DANGX_PLANESET"); return !(e && e[0] == 'p'); }();
    if (!enabled || sl.n < 0 || sl.n > 4 || (sl.n == 0 && !solve) || ga.ng < 1 || ga.ng > 6) return 0;
    static const int cap2 = [] { const char* e = getenv("DANGX_PS_CAP2"); return e ? atoi(e) : 13; }();  // A/B switch
    const int nb = ctx->hm.nbands, Sp = sl.s2 - sl.s1 + 1, cap = (Sp == 2) ? cap2 : 16;
    for (int q = 0; q < sl.n; ++q) {
        const int m = sl.s[q].mode;
        if (m < CH_POW || m > CH_LOGN_W) return 0;
        if (sl.s[q].pair && !(m == CH_MBB_BETA || m == CH_LOGN_NUP)) return 0;
    }
    FusedArgs fa;
    if (!planeset_args(ctx, ga, sl, fa)) return 0;
    if (solve && fa.cal && ga.nuc > 0) return 0;   // the kernel removes the templates before the solve's T / gain (no such model yet)
    const int bp = planeset_bp(ctx);
    if (bp) {  // sample loops exist for the power-law and mbb chains, in the one-lane form
        for (int q = 0; q < sl.n; ++q)
            if (sl.s[q].mode > CH_MBB_T) return 0;
        if (!(small_too && nb <= cap && planeset_lds(ga.ng, nb, fa.nv, 1, Sp) <= 80u * 1024u)) return 0;
        return dx_rtc_get(ctx, "dx_kern_planeset.h", planeset_name(Sp, nb, ga.ng, 1, solve, sl, 1)) ? 1 : 0;
    }
    // one lane where registers (cap) and the members' SED columns (two blocks per CU: 80 KB each) allow it, else lane pairs
    int lanes = 0;
    if (small_too && nb <= cap && planeset_lds(ga.ng, nb, fa.nv, 1, Sp) <= 80u * 1024u) lanes = 1;
    else if (nb > 12 && nb % 2 == 0 && nb / 2 <= cap && planeset_lds(ga.ng, nb, fa.nv, 2, Sp) <= 80u * 1024u) lanes = 2;
    if (!lanes) return 0;
    if (planeset_builtin(nb, ga.ng, lanes, solve, sl)) return lanes;
    return dx_rtc_get(ctx, "dx_kern_planeset.h", planeset_name(Sp, nb, ga.ng, lanes, solve, sl, 0)) ? lanes : 0;
}

// accp: per-sweep counters (sum over items of 1 + pair entries) or null
bool dx_launch_planeset(dangx_ctx* ctx, const GroupArgs& ga, const SweepList& sl, int lanes, int solve, unsigned nblk, unsigned long long* accp) {
    FusedArgs fa;
    if (lanes < 1 || lanes > 2 || !planeset_args(ctx, ga, sl, fa)) return false;
    const int nb = ctx->hm.nbands, ng = ga.ng, Sp = sl.s2 - sl.s1 + 1;
    const size_t ldsz = planeset_lds(ng, nb, fa.nv, lanes, Sp);
    const int bp = planeset_bp(ctx);
    if (!bp && planeset_builtin(nb, ng, lanes, solve, sl)) {
        if (nb == 20) launch_builtin<20, 6, 2, 1, CH_POW, CH_MBB_BETA + 8, CH_LOGN_NUP>(ctx, ga, fa, sl, Sp, nblk, ldsz, accp);
        else if (nb == 10 && sl.n == 0) launch_builtin<10, 4, 1, 1, 0, 0, 0>(ctx, ga, fa, sl, Sp, nblk, ldsz, accp);
        else if (nb == 10 && !solve) launch_builtin<10, 4, 1, 0, CH_POW, CH_MBB_BETA + 8, 0>(ctx, ga, fa, sl, Sp, nblk, ldsz, accp);
        else if (nb == 10) launch_builtin<10, 4, 1, 1, CH_POW, CH_MBB_BETA + 8, 0>(ctx, ga, fa, sl, Sp, nblk, ldsz, accp);
        else if (nb == 5) launch_builtin<5, 3, 1, 1, CH_POW, CH_MBB_BETA + 8, 0>(ctx, ga, fa, sl, Sp, nblk, ldsz, accp);
        else launch_builtin<3, 2, 1, 1, CH_POW, CH_MBB_BETA + 8, 0>(ctx, ga, fa, sl, Sp, nblk, ldsz, accp);
        return true;
    }
    hipFunction_t fn = dx_rtc_get(ctx, "dx_kern_planeset.h", planeset_name(Sp, nb, ng, lanes, solve, sl, bp));
    if (!fn) return false;
    const Model* dm = ctx->dm;
    GroupArgs gg = ga;
    SweepList ss = sl;
    unsigned long long* bad = ctx->counters;
    double* part = ctx->partial;
    void* args[] = {&dm, &gg, &fa, &ss, &bad, &accp, &part};
    return dx_rtc_launch(ctx, fn, nblk, ldsz, args) == 0;
}
