// dx_ampreg.h -- what the kernels on the amplitude kernel's schedule share (dangx_ampreg.hip: k_amp_reg, k_chisq_reg;
// dangx_schurreg.hip: the three passes of the Schur solve of template groups): global-address-space map pointers, the
// per-launch member roles, the tile SED evaluation, the LDS footprint and the host-side member classification.
#pragma once
#include "dx_ampdata.h"

namespace {


// Map pointers come out of the Model block as generic pointers, which the compiler can only load through FLAT
// instructions -- and a FLAT load counts on lgkmcnt as well as vmcnt, so every wait for an LDS read (the constant table,
// the SED columns) would also wait for all map loads in flight.  Viewed through the global address space they are
// global_load instructions (vmcnt only) and phase A really runs under the map loads.
typedef const double __attribute__((address_space(1)))* gcptr;
typedef double __attribute__((address_space(1)))* gptr;
__device__ __forceinline__ gcptr as_global(const double* p) { return reinterpret_cast<gcptr>(reinterpret_cast<uintptr_t>(p)); }
__device__ __forceinline__ gptr as_global_w(double* p) { return reinterpret_cast<gptr>(reinterpret_cast<uintptr_t>(p)); }

constexpr int MAXU = 4;  // templates / monopoles whose signal the HT form of the kernel removes from the data
struct AmpRegArgs {
    signed char vslot[MAXG];  // LDS column slot of group member g, -1: its SED is a row of the constant table
    signed char vcomp[MAXG];  // group member of slot v
    signed char vtype[MAXG];  // its component type
    int nv;                   // members with a column
    // HT form (pass 2 of the Schur solve of a template group): d_j -= cu[w][j] * template_w(pixel, plane) for w < nu, where
    // cu[w][j] = template_amplitudes(j, plane) on EVERY band for a member of the group (the new global amplitudes) and on
    // the bands it is not fitted at for a non-member (src/dang_cg_mod.f90:445-460)
    int nu;
    int ucomp[MAXU];
    unsigned umember;         // bit w: component ucomp[w] is a global-amplitude member of the group
    unsigned uinuc;           // bit w: a template / monopole (removed on its unfitted bands whichever group it belongs to)
    unsigned uhifit;          // bit w: a hi_fit member: sed = template * B_nu(T(pixel)) / RJ * 1e6 (src/dang_component_mod.f90:850-884)
    // residual pass (k_schur_resid_reg): global row r belongs to template slot rowu[r]; bit r of rowmono: a monopole's row
    signed char rowu[8];
    unsigned rowmono;
};
constexpr int RMAXF = 8;      // global rows the residual pass on this schedule carries per thread

// SEDs of one varying component for the TB bands of a tile -> its LDS column.  Same expressions as sed_eval_tab
// (dx_sed.h); the per-pixel state p comes from sed_prep.
template <int TB>
__device__ __forceinline__ void sed_tile(int type, const double* __restrict__ tab, int nb, int NG, int g, int j0, const Prep& p,
                                         double* __restrict__ colg) {
    const double* lnr = tab + (TROWS * g) * nb + j0;
    const double* cst = lnr + nb;
    const double* lnu9 = cst + nb;
    const double* nuc = tab + (TROWS * NG) * nb + j0;
    switch (type) {
    case DANGX_POWERLAW:  // src/dang_component_mod.f90:908
#pragma unroll
        for (int t = 0; t < TB; ++t) { colg[t * BLOCK] = exp_nr(p.p0 * lnr[t]); }
        break;
    case DANGX_MBB: {  // :947-948, in two passes of TB chains each (bounds the registers the scheduler may spend)
        double f[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) { f[t] = p.p2 * fast_rcp(exp_nr(p.p1 * nuc[t]) - 1.0); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TB; ++t) { colg[t * BLOCK] = f[t] * exp_nr(p.p0 * lnr[t]); }
        break;
    }
    case DANGX_FREEFREE: {  // :1026-1027
        const double rp1 = fast_rcp(p.p1);
#pragma unroll
        for (int t = 0; t < TB; ++t) { colg[t * BLOCK] = (ff_gaunt(lnu9[t], p.p0) * rp1) * cst[t]; }
        break;
    }
    case DANGX_LOGNORMAL: {  // :988
        const double rp1 = fast_rcp(p.p1);
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const double l = (lnu9[t] - p.p2) * rp1;
            colg[t * BLOCK] = exp_sat(-0.5 * (l * l)) * cst[t];
           
        }
        break;
    }
    default:  // cmb: 1/a2t(bp), :799-800
#pragma unroll
        for (int t = 0; t < TB; ++t) colg[t * BLOCK] = cst[t];
        break;
    }
}

// SED of global member w at this unit and band: template(pix, plane), times the Planck-to-RJ factor at the pixel's temperature
// for a hi_fit member (the expression of comp_sed, dx_sed.h)
// HF = false: no hi_fit member in the launch (the Planck factor then costs the template-only kernels no registers)
template <bool HF>
__device__ __forceinline__ double gl_sed(const AmpRegArgs& ra, int w, const double* tv, const double* tT, double nu) {
    if (HF && ((ra.uhifit >> w) & 1u)) return tv[w] * (planck_rj(nu, tT[w]) * 1e6f);
    return tv[w];
}
// template values (and, for hi_fit members, temperatures) of the unit
__device__ __forceinline__ void gl_load(const Model& M, const AmpRegArgs& ra, int i, int k, int npix, double* tv, double* tT) {
#pragma unroll
    for (int w = 0; w < MAXU; ++w) {
        tv[w] = (w < ra.nu) ? as_global(M.comp[ra.ucomp[w]].tmpl)[(long long)(k - 1) * npix + i] : 0.0;
        tT[w] = (w < ra.nu && ((ra.uhifit >> w) & 1u)) ? as_global(M.comp[ra.ucomp[w]].idx)[(long long)(k - 1) * npix + i] : 1.0;
    }
}

template <int TB>
size_t amp_reg_lds(int NG, int nb, int nv, int nu = 0) { return ((size_t)(TROWS * NG + 3 + nu) * nb + (size_t)nv * (TB + 3) * BLOCK) * sizeof(double); }

// the diffuse members' roles: table row or LDS column; non-zero: a member type this kernel does not evaluate
inline int amp_reg_members(dangx_ctx* ctx, const GroupArgs& a, AmpRegArgs& ra) {
    ra.nv = 0; ra.nu = 0; ra.umember = 0u; ra.rowmono = 0u; ra.uinuc = 0u; ra.uhifit = 0u;
    for (int w = 0; w < MAXU; ++w) ra.ucomp[w] = 0;
    for (int r = 0; r < 8; ++r) ra.rowu[r] = -1;
    unsigned planes = 0;
    for (int pl = 0; pl < flag_planes_h(a.flag); ++pl)
        planes |= 1u << (((a.flag & DANGX_FLAG_QU) ? 2 + pl : (a.flag & DANGX_FLAG_T) ? 1 : (a.flag & DANGX_FLAG_Q) ? 2 : 3) - 1);
    for (int g = 0; g < MAXG; ++g) { ra.vslot[g] = -1; ra.vcomp[g] = 0; ra.vtype[g] = 0; }
    for (int g = 0; g < a.ng; ++g) {
        const Comp& c = ctx->hm.comp[a.gc[g]];
        if (c.type < DANGX_POWERLAW || c.type > DANGX_CMB) return 1;
        // constant on EVERY plane of this launch -> a table row; otherwise evaluated per unit
        if (((unsigned)c.const_planes & planes) != planes) {
            ra.vcomp[ra.nv] = (signed char)g; ra.vtype[ra.nv] = (signed char)c.type;
            ra.vslot[g] = (signed char)ra.nv++;
        }
    }
    return 0;
}

// a CG group whose global members are templates / monopoles on delta bands with nothing else on its planes: the member roles
// plus the templates' slots; false: the group needs the run-time-typed passes of dangx_schur.hip
inline bool template_group_args(dangx_ctx* ctx, const GroupArgs& a, AmpRegArgs& ra) {
    static const bool enabled = [] { const char* e = getenv("DANGX_SCHUR_FAST"); return !(e && e[0] == '0'); }();  // A/B switch
    if (!enabled || a.no != 0 || a.ng < 1 || a.nt < 1 || a.nuc > MAXU) return false;
    for (int j = 0; j < ctx->hm.nbands; ++j)
        if (ctx->hm.band[j].n != 0) return false;
    for (int l = 0; l < ctx->hm.ncomp; ++l)
        if (ctx->desc[l].type == DANGX_TCMB) return false;
    if (amp_reg_members(ctx, a, ra)) return false;
    ra.nu = a.nuc; ra.umember = 0u; ra.rowmono = 0u; ra.uinuc = 0u; ra.uhifit = 0u;
    for (int r = 0; r < 8; ++r) ra.rowu[r] = -1;
    for (int w = 0; w < a.nuc; ++w) {
        const int l = a.uc[w];
        if (ctx->desc[l].type != DANGX_TEMPLATE && ctx->desc[l].type != DANGX_MONOPOLE) return false;
        ra.ucomp[w] = l;
        ra.uinuc |= 1u << w;
        for (int t = 0; t < a.nt; ++t) if (a.tc[t] == l) ra.umember |= 1u << w;
    }
    for (int t = 0; t < a.nt; ++t) {   // hi_fit members take slots of their own (they are not in the uc list)
        const int l = a.tc[t];
        if (ctx->desc[l].type != DANGX_HIFIT) continue;
        if (ra.nu >= MAXU || ctx->desc[l].nindices < 1) return false;
        ra.ucomp[ra.nu] = l;
        ra.umember |= 1u << ra.nu; ra.uhifit |= 1u << ra.nu;
        ++ra.nu;
    }
    for (int t = 0; t < a.nt; ++t) {   // every global member must have a slot
        bool found = false;
        for (int w = 0; w < ra.nu; ++w) found = found || ra.ucomp[w] == a.tc[t];
        if (!found) return false;
    }
    return true;
}

}  // namespace
