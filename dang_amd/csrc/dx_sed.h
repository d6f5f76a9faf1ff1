// dx_sed.h -- mixing-matrix elements (SEDs) on the device.
//
// Restates eval_sed / eval_signal and the SED kernels of the reference
// (src/dang_component_mod.f90:754-813, 886-1040; cmb: src/dang_bp_mod.f90:211-243).
// The reference re-dispatches on the type string and calls pow/exp for every
// (pixel, band, component) and every CG iteration.  Here the evaluation is split:
//   sed_prep()  -- everything that depends only on the pixel's spectral indices
//                  (once per pixel, or once per Metropolis proposal);
//   sed_eval()  -- the per-band part, using host-precomputed band scalars.
// (nu/nu_ref)**beta is evaluated as exp_nr(beta*log(nu/nu_ref)) with the logarithm
// precomputed on the host; this differs from pow() by a few ulp (|beta*ln r| * eps).
#pragma once
#include "dx_math.h"
#include "dx_model.h"

namespace dx {

struct Prep {
    double p0, p1, p2;
};

// src/dang_component_mod.f90:1024-1027 -- Gaunt-factor form, literal constants kept:
//   log(exp_nr(5.960 - sqrt(3)/pi * log(nu/1e9 * (T_e/1e4)**(-1.5))) + 2.71828)
// with log(nu9*t15) = log(nu9) + log(t15): lnu9 is host-precomputed per band, lt15 = -1.5*log(T_e/1e4)
// once per pixel (two transcendentals per band instead of three plus a pow per pixel).
__device__ __forceinline__ double ff_gaunt(double lnu9, double lt15) {
    constexpr double S3PI = 1.7320508075688772 / PI;  // sqrt(3.d0)/pi
    return log_pos(exp_nr(5.960 - S3PI * (lnu9 + lt15)) + 2.71828);
}

// B_nu(nu,T)/compute_bnu_prime_RJ(nu)*1e6: evaluate_T_cmb / evaluate_hi_fit (src/dang_component_mod.f90:815-884,
// B_nu :745-752, compute_bnu_prime_RJ src/dang_bp_mod.f90:160-168)
__device__ __forceinline__ double planck_rj(double nu, double T) {
    const double bnu = ((2.0 * H_PLANCK * (nu * nu * nu)) / (C_LIGHT * C_LIGHT)) * (1.0 / (exp_sat((H_PLANCK * nu) / (K_B * T)) - 1));
    const double rj = 2.0 * K_B * (nu * nu) / (C_LIGHT * C_LIGHT);
    return bnu / rj;
}
// eval_signal (src/dang_component_mod.f90:754-776): amplitude*sed, except 'T_cmb' whose signal is the bare sed
__device__ __forceinline__ double signal_of(const Comp& c, double amp, double sed) { return (c.type == DANGX_TCMB) ? sed : amp * sed; }

__device__ __forceinline__ Prep sed_prep(const Comp& c, double th0, double th1) {
    Prep p = {0.0, 0.0, 0.0};
    switch (c.type) {
    case DANGX_POWERLAW:  // :901-905
        p.p0 = th0;
        break;
    case DANGX_MBB: {  // :936-943
        const double z = mbb_z(th1);
        p.p0 = th0 + 1.0;
        p.p1 = z;
        p.p2 = exp_nr(z * c.nu_ref) - 1.0;
        break;
    }
    case DANGX_FREEFREE: {  // :1017-1024
        const double lt15 = -1.5 * log_pos(th0 / 1.0e4);  // log((T_e/1e4)**(-1.5))
        p.p0 = lt15;
        p.p1 = ff_gaunt(c.lnuref9, lt15);  // S_ref
        break;
    }
    case DANGX_LOGNORMAL:  // :978-984
        p.p0 = th0 * 1e9;
        p.p1 = th1;
        p.p2 = log_pos(th0);  // log(nu/(nu_p*1e9)) = log(nu/1e9) - log(nu_p): one log per pixel, the band part tabulated (lnu9)
        break;
    case DANGX_TCMB:  // :830-834
    case DANGX_HIFIT:  // :865-869
        p.p0 = th0;
        break;
    default:
        break;
    }
    return p;
}

// bandpass-integrated forms (bp%id /= 'delta'): tau0-weighted sums in sample order, e.g. :909-913.  The reference
// skips samples with nu0 == 0; the device copies carry tau = 0 and a harmless nu for those (s + 0*finite == s), so
// the loops are branch-free, one loop per component type, four independent transcendental chains in flight.
// (nu/nu_ref)**beta is exp_nr(beta*log(nu/nu_ref)) with the log tabulated per (component, sample) on the host: the
// same identity the delta-bandpass path uses (20 fp64 ops instead of pow's ~150).
// The sample tables are read-only for the whole launch and indexed wave-uniformly: viewed through the constant
// address space they are fetched with scalar loads (s_load, scalar cache) instead of per-lane flat loads.
typedef const double __attribute__((address_space(4))) * kptr;
__device__ __forceinline__ kptr as_const(const double* p) { return reinterpret_cast<kptr>(reinterpret_cast<uintptr_t>(p)); }

__device__ inline double sed_bandpass(const Model& M, const Comp& c, int j, const Prep& p) {
    const Band& b = M.band[j];
    const kptr nu = as_const(M.bp_nu0 + b.off);
    const kptr tau = as_const(M.bp_tau0 + b.off);
    const kptr lnr = as_const(c.bp_lnr + b.off);
    const int n = b.n;
    double s = 0.0;
    switch (c.type) {
    case DANGX_POWERLAW:
#pragma unroll 4
        for (int i = 0; i < n; ++i) s = s + tau[i] * exp_nr(p.p0 * lnr[i]);
        break;
    case DANGX_MBB:
#pragma unroll 4
        for (int i = 0; i < n; ++i) s = s + tau[i] * p.p2 / (exp_nr(p.p1 * nu[i]) - 1.0) * exp_nr(p.p0 * lnr[i]);
        break;
    case DANGX_FREEFREE:
#pragma unroll 2
        for (int i = 0; i < n; ++i) {
            const double r = nu[i] / c.nu_ref;
            s = s + tau[i] * ff_gaunt(log_pos(1.0 * nu[i] / 1.0e9), p.p0) / p.p1 * (1.0 / (r * r));
        }
        break;
    case DANGX_LOGNORMAL:
#pragma unroll 2
        for (int i = 0; i < n; ++i) {
            const double l = log_pos(nu[i] / p.p0) / p.p1;
            const double q = c.nu_ref / nu[i];
            s = s + tau[i] * exp_sat(-0.5 * (l * l)) * (q * q);
        }
        break;
    case DANGX_TCMB:
#pragma unroll 2
        for (int i = 0; i < n; ++i) s = s + tau[i] * planck_rj(nu[i], p.p0);
        break;
    default:
        break;
    }
    return (c.type == DANGX_TCMB) ? s * 1e6f : s;
}

// eval_sed for band j given the prepared pixel state (src/dang_component_mod.f90:778-813)
__device__ __forceinline__ double sed_eval(const Model& M, const Comp& c, int j, const Prep& p) {
    if (c.type == DANGX_CMB) return c.cst[j];  // 1.0/a2t(bp(band)), :799-800
    if (M.band[j].n != 0) return sed_bandpass(M, c, j, p);
    switch (c.type) {
    case DANGX_POWERLAW:  // :908
        return exp_nr(p.p0 * c.lnr[j]);
    case DANGX_MBB:  // :947-948
        return p.p2 / (exp_nr(p.p1 * M.band[j].nu_c) - 1.0) * exp_nr(p.p0 * c.lnr[j]);
    case DANGX_FREEFREE:  // :1026-1027
        return ff_gaunt(c.lnu9[j], p.p0) / p.p1 * c.cst[j];
    case DANGX_LOGNORMAL: {  // :988
        const double l = (c.lnu9[j] - p.p2) / p.p1;
        return exp_sat(-0.5 * (l * l)) * c.cst[j];
    }
    case DANGX_TCMB:  // :836-846
        return planck_rj(M.band[j].nu_c, p.p0) * 1e6f;
    default:
        return 0.0;
    }
}

// Block-shared LDS table of the per-(component, band) constants, so that the band loops read them with
// broadcast ds_read instead of scattered scalar loads:
//   tab[(6*l + q)*nb + j], q = 0: lnr, 1: cst, 2: lnu9, 3..5: csed of planes 1..3   (component l, band j)
//   tab[(6*ncomp + q)*nb + j], q = 0: nu_c, 1: gain, 2: offset
constexpr int TROWS = 6;
__device__ __forceinline__ int sed_table_size(const Model& M) { return (TROWS * M.ncomp + 3) * M.nbands; }
// list == nullptr: one row block per component of the model (nc = M.ncomp); otherwise row block g holds
// component list[g] (nc entries) -- the amplitude kernel only needs its group's components
__device__ __forceinline__ void sed_table_build(const Model& M, double* tab, int tid, int nthreads,
                                                const int* list = nullptr, int nc = -1) {
    if (nc < 0) nc = M.ncomp;
    const int nb = M.nbands, n = (TROWS * nc + 3) * nb;
    for (int t = tid; t < n; t += nthreads) {
        const int row = t / nb, j = t - row * nb;
        double v;
        if (row < TROWS * nc) {
            const Comp& c = M.comp[list ? list[row / TROWS] : row / TROWS];
            const int q = row - TROWS * (row / TROWS);
            v = (q == 0) ? c.lnr[j] : (q == 1) ? c.cst[j] : (q == 2) ? c.lnu9[j] : c.csed[q - 3][j];
        } else {
            const int q = row - TROWS * nc;
            v = (q == 0) ? M.band[j].nu_c : (q == 1) ? M.gain[j] : M.offset[j];
        }
        tab[t] = v;
    }
}
// eval_sed for a delta bandpass from the table (same expressions as sed_eval)
__device__ __forceinline__ double sed_eval_tab(int type, const double* tab, int nb, int ncomp, int l, int j, const Prep& p) {
    const double* tc = tab + (TROWS * l) * nb + j;
    switch (type) {
    case DANGX_POWERLAW: return exp_nr(p.p0 * tc[0]);
    case DANGX_MBB: return p.p2 / (exp_nr(p.p1 * tab[(TROWS * ncomp) * nb + j]) - 1.0) * exp_nr(p.p0 * tc[0]);
    case DANGX_FREEFREE: return ff_gaunt(tc[2 * nb], p.p0) / p.p1 * tc[nb];
    case DANGX_LOGNORMAL: {
        const double l2 = (tab[(TROWS * l + 2) * nb + j] - p.p2) / p.p1;
        return exp_sat(-0.5 * (l2 * l2)) * tc[nb];
    }
    case DANGX_CMB: return tc[nb];
    default: return 0.0;
    }
}

// SED of component l on plane k when its indices are spatially constant there (host-evaluated)
__device__ __forceinline__ double sed_const_tab(const double* tab, int nb, int l, int k, int j) {
    return tab[(TROWS * l + 2 + k) * nb + j];
}

// eval_sed / eval_signal for EVERY component type at (pixel i, map k, band j) -- the generic paths use these
// (src/dang_component_mod.f90:754-813): template / monopole: sed = template(pix,map); hi_fit: template * evaluate_hi_fit;
// signal = template_amplitudes(band,map) * sed for the three global types, the bare sed for T_cmb, amplitude * sed else.
__device__ __forceinline__ double comp_sed(const Model& M, const Comp& c, int i, int k, int j, const Prep& p) {
    if ((c.const_planes >> (k - 1)) & 1) return c.csed[k - 1][j];  // spatially constant indices: evaluated once on the host
    if (c.type == DANGX_TEMPLATE || c.type == DANGX_MONOPOLE) return c.tmpl[(long long)(k - 1) * M.npix + i];
    if (c.type == DANGX_HIFIT) {  // :850-884 (same expression as evaluate_T_cmb)
        double s;
        if (M.band[j].n == 0) s = planck_rj(M.band[j].nu_c, p.p0);
        else {
            s = 0.0;
            for (int q = 0; q < M.band[j].n; ++q)
                s = s + M.bp_tau0[M.band[j].off + q] * planck_rj(M.bp_nu0[M.band[j].off + q], p.p0);
        }
        return c.tmpl[(long long)(k - 1) * M.npix + i] * (s * 1e6f);
    }
    return sed_eval(M, c, j, p);
}
__device__ __forceinline__ double comp_signal(const Model& M, const Comp& c, int i, int k, int j, double amp, const Prep& p) {
    const double s = comp_sed(M, c, i, k, j, p);
    if (is_global_type(c.type)) return c.tamp[k - 1][j] * s;
    return (c.type == DANGX_TCMB) ? s : amp * s;
}

// spectral indices of component c at (pixel i, map k): c%indices(i,k,:)
__device__ __forceinline__ void load_theta(const Model& M, const Comp& c, int i, int k, double& th0, double& th1) {
    const long long plane = (long long)M.npix;
    th0 = (c.nind > 0) ? c.idx[((long long)0 * M.nmaps + (k - 1)) * plane + i] : 0.0;
    th1 = (c.nind > 1) ? c.idx[((long long)1 * M.nmaps + (k - 1)) * plane + i] : 0.0;
}

}  // namespace dx
