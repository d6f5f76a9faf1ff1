// dangx_mh.hip -- Metropolis index kernels, LDS-column form (any band count, every likelihood / prior type).
#include "dx_host.h"

namespace {

// ---------------------------------------------------------------------------
// Index phase: sample_index_mh, per-pixel branch (src/dang_sample_mod.f90:332-481) with
// update_sample_model (:548-553), evaluate_lnL / evaluate_marginal_lnL
// (src/dang_lnl_mod.f90:126-182, 47-124) and the priors (:394-400) fused.
// One thread per pixel.  The pixel's cleaned data d(k,j), 1/rms(k,j) and the chain-invariant
// SED factor F(j) are staged once into LDS columns [slot][thread] (conflict-free: lane l
// touches bank pair 2l), the chain state lives in registers, the index map is written once.
//
// Chain modes: the SED of the sampled component factorises into a part that is constant
// along the chain (F_j, evaluated once) and a part that depends on the proposal, with the
// reference's multiplication order kept, e.g. mbb (:947-948) = (A/B_j) * P_j:
//   CH_POW       power-law beta     : exp_nr(beta*lnr_j)
//   CH_MBB_BETA  mbb beta (T fixed) : F_j = A/B_j ; sed = F_j * exp_nr((beta+1)*lnr_j)
//   CH_MBB_T     mbb T (beta fixed) : F_j = P_j   ; sed = (A(T)/B_j(T)) * F_j
//   CH_LOGN_NUP  lognormal nu_p     : sed = exp_nr(-0.5*(log(nu_j/(nu_p*1e9))/w)^2) * cst_j
//   CH_LOGN_W    lognormal w        : F_j = log(nu_j/(nu_p*1e9)) ; sed = exp_nr(-0.5*(F_j/w)^2) * cst_j
//   CH_GENERIC   anything else (free-free T_e, bandpass-integrated bands): sed_prep + sed_eval
struct ChainCtx {
    const Model& M;
    const Comp& c;
    const IndexArgs& a;
    double* lds;        // per-thread columns
    const double* tab;  // block-shared constant table (sed_table_build)
    int BS, tid, nb, Sp;
    double amp0, amp1, other;  // amplitudes on the planes; the index that is NOT sampled
    int i, k0;                 // pixel and first map (hi_fit: template(pix,map) * template_amplitudes(band,map))
    __device__ __forceinline__ double& D(int kk, int j) const { return lds[(kk * nb + j) * BS + tid]; }
    __device__ __forceinline__ double& IS(int kk, int j) const { return lds[((Sp + kk) * nb + j) * BS + tid]; }
    __device__ __forceinline__ double& F(int j) const { return lds[(2 * Sp * nb + j) * BS + tid]; }
};

// -1/2 sum ((d-m)/rms)^2 per plane (evaluate_lnL) or the marginal form; acc0/acc1 = per-plane parts
__device__ __forceinline__ double chain_lnl(const ChainCtx& C, double th, int lnl_type, double& acc0, double& acc1) {
    const Model& M = C.M;
    const Comp& c = C.c;
    const bool first = (C.a.nind == 0);
    acc0 = 0.0; acc1 = 0.0;
    if (lnl_type == DANGX_LNL_PRIOR) return 0.0;
    double s0 = 0.0, s1 = 0.0;
    Prep pr = {0.0, 0.0, 0.0};
    switch (C.a.mode) {
    case CH_POW: s0 = th; break;
    case CH_MBB_BETA: s0 = th + 1.0; break;
    case CH_MBB_T: s0 = mbb_z(th); s1 = exp_nr(s0 * c.nu_ref) - 1.0; break;
    case CH_LOGN_NUP: s0 = log_pos(th); s1 = C.other; break;
    case CH_LOGN_W: s1 = th; break;
    default: pr = sed_prep(c, first ? th : C.other, first ? C.other : th); break;
    }
    double lnL = 0.0;
    for (int j = 0; j < C.nb; ++j) {
        double s;
        switch (C.a.mode) {
        case CH_POW: s = exp_nr(s0 * c.lnr[j]); break;
        case CH_MBB_BETA: s = C.F(j) * exp_nr(s0 * c.lnr[j]); break;
        case CH_MBB_T: s = s1 / (exp_nr(s0 * M.band[j].nu_c) - 1.0) * C.F(j); break;
        case CH_LOGN_NUP: { const double l = (c.lnu9[j] - s0) / s1; s = exp_sat(-0.5 * (l * l)) * c.cst[j]; break; }
        case CH_LOGN_W: { const double l = C.F(j) / s1; s = exp_sat(-0.5 * (l * l)) * c.cst[j]; break; }
        default: s = (c.type == DANGX_HIFIT) ? 0.0 : sed_eval(M, c, j, pr); break;
        }
        // eval_signal (src/dang_component_mod.f90:754-776): amplitude(pix,map) * sed; T_cmb: the sed itself;
        // hi_fit: template_amplitudes(band,map) * (template(pix,map) * evaluate_hi_fit)
        double m0 = C.amp0 * s, m1 = C.amp1 * s;
        if (c.type == DANGX_TCMB) {  // the bare sed (:770-771)
            m0 = s; m1 = s;
        } else if (c.type == DANGX_HIFIT) {
            m0 = c.tamp[C.k0 - 1][j] * comp_sed(M, c, C.i, C.k0, j, pr);
            if (C.Sp == 2) m1 = c.tamp[C.k0][j] * comp_sed(M, c, C.i, C.k0 + 1, j, pr);
        }
        if (lnl_type == DANGX_LNL_CHISQ) {
            const double t = (C.D(0, j) - m0) * C.IS(0, j);
            acc0 = acc0 - 0.5 * (t * t);
            if (C.Sp == 2) {
                const double t2 = (C.D(1, j) - m1) * C.IS(1, j);
                acc1 = acc1 - 0.5 * (t2 * t2);
            }
        } else {  // marginal: -0.5*TNd*invTNT*TNd per (band, plane), src/dang_lnl_mod.f90:113-122
            for (int kk = 0; kk < C.Sp; ++kk) {
                const double m = kk ? m1 : m0;
                const double is = C.IS(kk, j);
                const double TN = m * (is * is);
                const double TNd = TN * C.D(kk, j);
                const double TNT = TN * m;
                lnL = lnL - 0.5 * TNd * (1.0 / TNT) * TNd;
            }
        }
    }
    return (lnl_type == DANGX_LNL_CHISQ) ? acc0 + acc1 : lnL;
}

// Fast path of evaluate_lnL for the chain: chisq likelihood, delta bandpasses, compile-time chain mode
// (CH_POW / CH_MBB_BETA / CH_MBB_T), plane count SP and band tile TB (nb % TB == 0).  A tile first issues
// every LDS / scalar load of its TB bands, then runs the TB independent exp chains interleaved, then
// accumulates in band order (same summation order as the plain loop).
// BP: band j may be bandpass-integrated -- its SED is the tau-weighted sum over the samples, in the operation order of
// sed_bandpass (the chain-invariant factor of the mbb modes is per SAMPLE there and is recomputed).
template <int MODE, int SP, int TB, bool BP>
__device__ __forceinline__ double chain_lnl_tiled(const ChainCtx& C, double th, double& acc0, double& acc1) {
    const Model& M = C.M;
    const Comp& c = C.c;
    double s0 = 0.0, s1 = 0.0;
    if (MODE == CH_POW) s0 = th;
    else if (MODE == CH_MBB_BETA) s0 = th + 1.0;
    else { s0 = mbb_z(th); s1 = exp_nr(s0 * c.nu_ref) - 1.0; }
    acc0 = 0.0; acc1 = 0.0;
    if (BP) {
        // chain-invariant scalars of the per-sample factor: mbb beta chain: z, A from the fixed temperature;
        // mbb T chain: the fixed beta + 1
        double bz = 0.0, bA = 0.0, bb1 = 0.0;
        if (MODE == CH_MBB_BETA) { bz = mbb_z(C.other); bA = exp_nr(bz * c.nu_ref) - 1.0; }
        if (MODE == CH_MBB_T) bb1 = C.other + 1.0;
        for (int j = 0; j < C.nb; ++j) {
            const Band& b = M.band[j];
            double s;
            if (b.n == 0) {
                const double e = exp_nr((MODE == CH_MBB_T) ? s0 * b.nu_c : s0 * c.lnr[j]);
                s = (MODE == CH_POW) ? e : (MODE == CH_MBB_BETA) ? C.F(j) * e : s1 / (e - 1.0) * C.F(j);
            } else {
                const kptr nu = as_const(M.bp_nu0 + b.off);
                const kptr tau = as_const(M.bp_tau0 + b.off);
                const kptr lnr = as_const(c.bp_lnr + b.off);
                s = 0.0;
#pragma unroll 4
                for (int q = 0; q < b.n; ++q) {
                    if (MODE == CH_POW) s = s + tau[q] * exp_nr(s0 * lnr[q]);
                    else if (MODE == CH_MBB_BETA) s = s + tau[q] * bA / (exp_nr(bz * nu[q]) - 1.0) * exp_nr(s0 * lnr[q]);
                    else s = s + tau[q] * s1 / (exp_nr(s0 * nu[q]) - 1.0) * exp_nr(bb1 * lnr[q]);
                }
            }
            const double r0 = (C.D(0, j) - C.amp0 * s) * C.IS(0, j);
            acc0 = acc0 - 0.5 * (r0 * r0);
            if (SP == 2) {
                const double r1 = (C.D(1, j) - C.amp1 * s) * C.IS(1, j);
                acc1 = acc1 - 0.5 * (r1 * r1);
            }
        }
        return acc0 + acc1;
    }
    for (int j0 = 0; j0 < C.nb; j0 += TB) {
        double f[TB], d0[TB], i0[TB], d1[TB], i1[TB], x[TB], s[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const int j = j0 + t;
            x[t] = (MODE == CH_MBB_T) ? s0 * C.tab[(TROWS * M.ncomp) * C.nb + j] : s0 * C.tab[(TROWS * C.a.comp) * C.nb + j];
            f[t] = (MODE == CH_POW) ? 1.0 : C.F(j);
            d0[t] = C.D(0, j); i0[t] = C.IS(0, j);
            if (SP == 2) { d1[t] = C.D(1, j); i1[t] = C.IS(1, j); }
        }
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const double e = exp_nr(x[t]);
            if (MODE == CH_POW) s[t] = e;
            else if (MODE == CH_MBB_BETA) s[t] = f[t] * e;
            else s[t] = s1 / (e - 1.0) * f[t];
        }
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const double r0 = (d0[t] - C.amp0 * s[t]) * i0[t];
            acc0 = acc0 - 0.5 * (r0 * r0);
            if (SP == 2) {
                const double r1 = (d1[t] - C.amp1 * s[t]) * i1[t];
                acc1 = acc1 - 0.5 * (r1 * r1);
            }
        }
    }
    return acc0 + acc1;
}

template <bool FAST>
__device__ __forceinline__ double index_prior(const ChainCtx& C, double val) {
    const Comp& c = C.c;
    const int q = C.a.nind;
    const int t = c.prior_type[q];
    if (t == DANGX_PRIOR_GAUSSIAN) {
        // log(eval_normal_prior) (src/dang_util_mod.f90:112-121, src/dang_sample_mod.f90:395):
        // log(exp_nr(-(x-m)^2/(2 var))/(std*sqrt(2 pi))) = -(x-m)^2/(2 var) - log(std*sqrt(2 pi));
        // the reference's exp_nr() underflows to 0 (log -> -inf) beyond ~745
        const double mean = c.gauss[q][0], std = c.gauss[q][1];
        const double arg = ((val - mean) * (val - mean)) / (2 * (std * std));
        return (arg > 745.0) ? -INFINITY : -arg - c.lgden[q];
    }
    if (t == DANGX_PRIOR_JEFFREYS) {  // eval_jeffreys_prior, src/dang_lnl_mod.f90:242-304
        double sum = 0.0;
        if (c.is_synch) {
            const Prep pr = sed_prep(c, val, 0.0);
            for (int kk = 0; kk < C.Sp; ++kk)
                for (int j = 0; j < C.nb; ++j) {
                    const double amp = kk ? C.amp1 : C.amp0;
                    const double ss = amp * (FAST ? sed_eval_tab(c.type, C.tab, C.nb, C.M.ncomp, C.a.comp, j, pr)
                                                   : sed_eval(C.M, c, j, pr));
                    const double rr = C.IS(kk, j);  // 1/rms
                    const double tt = (rr * rr) * (ss / amp) * c.lnr[j];
                    sum = sum + tt * tt;
                }
        }
        return log(sqrt(sum));
    }
    return 0.0;
}

// the chain of one pixel; returns the number of accepted proposals; chi[0..3] = chi^2 of the touched
// planes before (plane0, plane1) and after (plane0, plane1) the sweep
// MODE == CH_GENERIC: everything decided at run time (a.mode, lnl type, plane count, any nb);
// otherwise the chisq fast path above with compile-time MODE / SP / TB.
template <int MODE, int SP, int TB, bool BP>
__device__ __forceinline__ unsigned long long index_chain(const Model& M, const IndexArgs& a, double* lds, const double* tab,
                                                          int BS, int tid, int i, double chi[4]) {
    const int npix = M.npix, nb = M.nbands;
    const Comp& c = M.comp[a.comp];
    const int Sp = a.s2 - a.s1 + 1;
    double* out = c.idx + ((long long)a.nind * M.nmaps) * npix + i;
    if (is_masked(M.mask[i])) {  // :362 cycle; index_map stays 0 (:223) and is copied back (:480-483)
        for (int k = a.s1; k <= a.s2; ++k) out[(long long)(k - 1) * npix] = 0.0;
        return 0ull;
    }
    // chain state: sample(l) = c%indices(i, map_inds(1), l)  (:372-377)
    double sample0, sample1;
    load_theta(M, c, i, a.s1, sample0, sample1);
    const bool first = (a.nind == 0);
    ChainCtx C{M, c, a, lds, tab, BS, tid, nb, Sp, 0.0, 0.0, first ? sample1 : sample0, i, a.s1};
    // --- stage data_raw (:173-177) and 1/rms: loads of ST bands are issued together
    constexpr int ST = (MODE == CH_GENERIC) ? 4 : TB;
    for (int kk = 0; kk < Sp; ++kk) {
        const int k = a.s1 + kk;
        const double ak = c.amp[(long long)(k - 1) * npix + i];
        if (kk) C.amp1 = ak; else C.amp0 = ak;
        const long long bstride = (long long)M.nmaps * npix;
        const double* sigp = M.sig + (long long)(k - 1) * npix + i;
        const double* rmsp = M.rms + (long long)(k - 1) * npix + i;
#pragma unroll 2
        for (int j0 = 0; j0 < nb; j0 += ST) {
            double dv[ST], rv[ST];
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                const int j = (j0 + t < nb) ? j0 + t : nb - 1;
                dv[t] = sigp[j * bstride];
                rv[t] = rmsp[j * bstride];
            }
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                const int j = j0 + t;
                if (j < nb) {
                    C.D(kk, j) = (k == 1) ? (dv[t] - tab[(TROWS * M.ncomp + 2) * nb + j]) / tab[(TROWS * M.ncomp + 1) * nb + j] : dv[t];
                    C.IS(kk, j) = 1.0 / rv[t];
                }
            }
        }
    }
    // --- remove every OTHER component (:180-196), in component_list order; a.others holds the
    // components whose amplitude may be non-zero on these planes (an all-zero plane contributes 0*sed).
    // The next component's amplitude / indices are fetched while the current one is processed.
    {
        unsigned om = a.others;
        double na[2] = {0.0, 0.0}, nt0[2] = {0.0, 0.0}, nt1[2] = {0.0, 0.0};
        auto fetch = [&](int l) {
            const Comp& c2 = M.comp[l];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                if (kk < Sp) {
                    na[kk] = c2.amp[(long long)(a.s1 + kk - 1) * npix + i];
                    if (MODE == CH_GENERIC || BP || !((c2.const_planes >> (a.s1 + kk - 1)) & 1))
                        load_theta(M, c2, i, a.s1 + kk, nt0[kk], nt1[kk]);
                }
        };
        int l = om ? __builtin_ctz(om) : -1;
        if (l >= 0) fetch(l);
        while (l >= 0) {
            const Comp& c2 = M.comp[l];
            const double ca[2] = {na[0], na[1]}, ct0[2] = {nt0[0], nt0[1]}, ct1[2] = {nt1[0], nt1[1]};
            om &= om - 1;
            const int ln = om ? __builtin_ctz(om) : -1;
            if (ln >= 0) fetch(ln);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                if (kk < Sp) {
                    if (MODE != CH_GENERIC && !BP && ((c2.const_planes >> (a.s1 + kk - 1)) & 1)) {
                        for (int j = 0; j < nb; ++j) C.D(kk, j) -= ca[kk] * sed_const_tab(tab, nb, l, a.s1 + kk, j);
                    } else {
                        const Prep pr = sed_prep(c2, ct0[kk], ct1[kk]);
                        const int ty2 = c2.type;
#pragma unroll 1
                        for (int j = 0; j < nb; ++j)
                            C.D(kk, j) -= (MODE != CH_GENERIC && !BP) ? ca[kk] * sed_eval_tab(ty2, tab, nb, M.ncomp, l, j, pr)
                                                                      : comp_signal(M, c2, i, a.s1 + kk, j, ca[kk], pr);
                    }
                }
            l = ln;
        }
    }
    // --- chain-invariant SED factor
    if (a.mode == CH_MBB_BETA) {
        const double z = mbb_z(sample1);
        const double A = exp_nr(z * c.nu_ref) - 1.0;
        for (int j = 0; j < nb; ++j) C.F(j) = A / (exp_nr(z * tab[(TROWS * M.ncomp) * nb + j]) - 1.0);
    } else if (a.mode == CH_MBB_T) {
        for (int j = 0; j < nb; ++j) C.F(j) = exp_nr((sample0 + 1.0) * tab[(TROWS * a.comp) * nb + j]);
    } else if (a.mode == CH_LOGN_W) {
        {
            const double lp = log_pos(sample0);
            for (int j = 0; j < nb; ++j) C.F(j) = c.lnu9[j] - lp;
        }
    }
    const int lnl_type = (MODE == CH_GENERIC) ? c.lnl_type[a.nind] : DANGX_LNL_CHISQ;
    const unsigned long long gpix = (unsigned long long)(M.pix0 + i);
    unsigned long long nacc = 0;
    double cur = first ? sample0 : sample1;  // sample(nind)
    double a0, a1, c0, c1;                   // per-plane likelihood parts: current / proposal
    auto lnl_of = [&](double th, double& p0, double& p1) -> double {
        if (MODE == CH_GENERIC) return chain_lnl(C, th, lnl_type, p0, p1);
        return chain_lnl_tiled<MODE == CH_GENERIC ? CH_POW : MODE, SP, TB, BP>(C, th, p0, p1);
    };
    double lnl = lnl_of(cur, a0, a1);
    if (lnl_type != DANGX_LNL_CHISQ) {       // chi^2 bookkeeping needs the chisq form
        double t0, t1;
        chain_lnl(C, cur, DANGX_LNL_CHISQ, t0, t1);
        chi[0] = -2.0 * t0; chi[1] = -2.0 * t1;
    } else {
        chi[0] = -2.0 * a0; chi[1] = -2.0 * a1;
    }
    bool sample_it = true;
    if (lnl_type == DANGX_LNL_PRIOR) {  // :389-392
        double u1, u2;
        sample_it = false;
        uniform2(a.seed, a.stream, gpix, 0u, u1, u2);
        cur = rand_normal(c.gauss[a.nind][0], c.gauss[a.nind][1], u1, u2);
    }
    double lnl_old = lnl + index_prior<MODE != CH_GENERIC>(C, cur);
    if (sample_it) {
        const double step = c.step[a.nind];
        const double lo = c.uni[a.nind][0], hi = c.uni[a.nind][1];
        for (int l = 1; l <= a.nsample; ++l) {
            double u1, u2, u3;
            uniform3(a.seed, a.stream, gpix, (uint32_t)l, u1, u2, u3);  // one Philox call per step
            const double prop = cur + rand_normal(0.0, step, u1, u2);  // :414
            if (prop < lo || prop > hi) continue;  // :415 (the accept draw is not used)
            lnl = lnl_of(prop, c0, c1);
            const double lnl_new = lnl + index_prior<MODE != CH_GENERIC>(C, prop);
            const double diff = lnl_new - lnl_old;
            bool acc;
            if (a.ml_mode == DANGX_ML_OPTIMIZE) {
                acc = diff > 0.0;  // :443-447
            } else {
                // :448-454  diff > log(u)  <=>  diff >= 0 or exp(diff) > u   (u in (0,1)); diff has no bound: exp_sat
                acc = (diff >= 0.0) || (exp_sat(diff) > u3);
            }
            if (acc) {
                cur = prop;
                lnl_old = lnl_new;
                a0 = c0; a1 = c1;
                ++nacc;
            }
        }
    }
    for (int k = a.s1; k <= a.s2; ++k) out[(long long)(k - 1) * npix] = cur;  // :465, :483
    if (lnl_type != DANGX_LNL_CHISQ) chain_lnl(C, cur, DANGX_LNL_CHISQ, a0, a1);
    chi[2] = -2.0 * a0; chi[3] = -2.0 * a1;
    return nacc;
}

// chi_partial (nullable): [4][gridDim.x] block sums of chi[0..3]
template <int MODE, int SP, int TB, bool BP>
__global__ __launch_bounds__(BLOCK) void k_index_mh(const Model* __restrict__ Mp, IndexArgs a,
                                                    unsigned long long* __restrict__ accepted,
                                                    double* __restrict__ chi_partial) {
    extern __shared__ double lds[];  // [constant table | per-thread columns]
    const Model& M = *Mp;
    const int BS = blockDim.x, tid = threadIdx.x;
    const int i = blockIdx.x * BS + tid;
    sed_table_build(M, lds, tid, BS);
    __syncthreads();
    double chi[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned long long nacc = (i < M.npix) ? index_chain<MODE, SP, TB, BP>(M, a, lds + sed_table_size(M), lds, BS, tid, i, chi) : 0ull;
    if (accepted) {  // every lane takes part in the wave reduction
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
            for (int w = 0; w < BS / 64; ++w) s += sh[tid][w];
            chi_partial[(long long)tid * gridDim.x + blockIdx.x] = s;
        }
    }
}

}  // namespace

void dx_launch_mh_lds(dangx_ctx* ctx, const IndexArgs& a, bool fast, int Sp, unsigned nblk, int bs, size_t lds, unsigned long long* accp) {
    const int nb = ctx->hm.nbands;
    const int tb = (nb % 5 == 0) ? 5 : (nb % 4 == 0) ? 4 : (nb % 3 == 0) ? 3 : 1;
#define DX_LAUNCH_MH(MODE_, SP_, TB_)                                                                            \
    hipLaunchKernelGGL((k_index_mh<MODE_, SP_, TB_, false>), dim3(nblk), dim3(bs), lds, ctx->stream, ctx->dm, a, accp, ctx->partial)
#define DX_LAUNCH_MH_BP(MODE_)                                                                                   \
    do { if (Sp == 2) hipLaunchKernelGGL((k_index_mh<MODE_, 2, 1, true>), dim3(nblk), dim3(bs), lds, ctx->stream, ctx->dm, a, accp, ctx->partial); \
         else hipLaunchKernelGGL((k_index_mh<MODE_, 1, 1, true>), dim3(nblk), dim3(bs), lds, ctx->stream, ctx->dm, a, accp, ctx->partial); } while (0)
#define DX_MH_TB(MODE_, SP_)                                                                                     \
    do { if (tb == 5) DX_LAUNCH_MH(MODE_, SP_, 5); else if (tb == 4) DX_LAUNCH_MH(MODE_, SP_, 4);                \
         else if (tb == 3) DX_LAUNCH_MH(MODE_, SP_, 3); else DX_LAUNCH_MH(MODE_, SP_, 1); } while (0)
#define DX_MH_SP(MODE_) do { if (Sp == 2) DX_MH_TB(MODE_, 2); else DX_MH_TB(MODE_, 1); } while (0)
    if (!fast) DX_LAUNCH_MH(CH_GENERIC, 1, 1);
    else if (a.bp && a.mode == CH_POW) DX_LAUNCH_MH_BP(CH_POW);
    else if (a.bp && a.mode == CH_MBB_BETA) DX_LAUNCH_MH_BP(CH_MBB_BETA);
    else if (a.bp) DX_LAUNCH_MH_BP(CH_MBB_T);
    else if (a.mode == CH_POW) DX_MH_SP(CH_POW);
    else if (a.mode == CH_MBB_BETA) DX_MH_SP(CH_MBB_BETA);
    else DX_MH_SP(CH_MBB_T);
#undef DX_MH_SP
#undef DX_MH_TB
#undef DX_LAUNCH_MH
#undef DX_LAUNCH_MH_BP
}
