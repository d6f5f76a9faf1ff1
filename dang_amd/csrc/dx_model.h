// dx_model.h -- device-resident description of one pixel shard (gfx950 only).
//
// The reference keeps this state in module globals and derived types
// (src/dang_util_mod.f90:16-37, src/dang_component_mod.f90:12-48,
// src/dang_bp_mod.f90:7-15, src/dang_data_mod.f90:23-45).  Here it is one POD
// block in HBM that every kernel receives by pointer; all fields are
// wave-uniform, so the compiler reads them through the scalar cache (s_load).
#pragma once
#include "dx_rtc_compat.h"
#ifdef __HIPCC_RTC__
#include "dangx.h"   // the embedded copy (dangx_rtc.hip)
#else
#include "../../include/dangx.h"
#endif

namespace dx {

constexpr int MAXB = DANGX_MAX_BANDS;
constexpr int MAXC = DANGX_MAX_COMPS;
constexpr int MAXI = DANGX_MAX_IND;
constexpr int MAXG = DANGX_MAX_GROUP;
constexpr int MAXT = 4;  // global-amplitude components per CG group

// constants, src/dang_util_mod.f90:12-15,19 (exact literals)
constexpr double PI = 3.141592653589793238462643383279502884197;
constexpr double K_B = 1.3806503e-23;
constexpr double C_LIGHT = 2.99792458e8;
constexpr double MISSVAL = -1.6375e30;
constexpr double H_PLANCK = 1.0545726691251021e-34 * 2.0 * PI;

// h / (k T) of the modified blackbody (src/dang_component_mod.f90:936), kept within +-1e-4 s: below 5e-7 K the Planck factor
// is exp(> 1e9) = inf either way, and dx::exp_nr needs |z nu| < 1.4e9 (dx_math.h) for frequencies up to 14 THz
__host__ __device__ inline double mbb_z(double T) {
    const double z = H_PLANCK / (K_B * T);
    return z > 1e-4 ? 1e-4 : (z < -1e-4 ? -1e-4 : z);
}

struct Band {
    double nu_c;  // Hz
    int n;        // 0 = delta
    int off;      // offset into bp_nu0 / bp_tau0
};

struct Comp {
    int type, nind, group, sample_amp, is_synch, pad;
    double nu_ref;
    double* amp;  // [nmaps][npix]
    double* idx;  // [nind][nmaps][npix]
    int lnl_type[MAXI], prior_type[MAXI];
    double gauss[MAXI][2], uni[MAXI][2], step[MAXI];
    double lgden[MAXI];  // log(std*sqrt(2*pi)) of the gaussian prior (eval_normal_prior's denominator)
    // per-band host-precomputed scalars (delta bandpass fast path)
    double lnr[MAXB];   // log(nu_c/nu_ref)
    double cst[MAXB];   // cmb: 1/a2t(bp) ; freefree: 1/(r*r) ; lognormal: (nu_ref/nu_c)^2
    double lnu9[MAXB];  // freefree: log(1.0*nu_c/1e9)
    double lnuref9;     // freefree: log(1.0*nu_ref/1e9)
    // planes (bit k-1) on which every spectral index of the component is spatially constant: the SED
    // is then pixel independent and csed[k-1][j] holds it (evaluated once on the host)
    int const_planes, pad1;
    double csed[3][MAXB];
    // global-amplitude types (template / monopole / hi_fit)
    const double* tmpl;      // c%template [nmaps][npix]
    int corr_mask, nfit;     // bit j: band j is fitted (c%corr), number of fitted bands
    double tamp[3][MAXB];    // c%template_amplitudes(band, map) as [map][band]
    // bandpass-integrated bands: log(nu0/nu_ref) of every bandpass sample (same indexing as Model::bp_nu0)
    const double* bp_lnr;
};

struct Model {
    int npix, nmaps, nbands, ncomp;
    int all_delta, pad0;  // every band is a 'delta' bandpass (fast SED path)
    long long pix0;
    const double* sig;   // [nbands][nmaps][npix]
    const double* rms;   // [nbands][nmaps][npix]
    const double* mask;  // [nmaps][npix]
    const double* bp_nu0;
    const double* bp_tau0;
    double tcmb;
    double mbb_batch_z;  // 5 max(nu_c) / 700: h/(kT) * this < 1 <=> five Planck denominators can be multiplied without overflow (dx_chain.h)
    double gain[MAXB], offset[MAXB];
    Band band[MAXB];
    Comp comp[MAXC];
};

__host__ __device__ inline bool is_masked(double m) { return m == 0.0 || m == MISSVAL; }
__host__ __device__ inline bool is_global_type(int t) { return t >= DANGX_TEMPLATE && t <= DANGX_HIFIT; }

}  // namespace dx
