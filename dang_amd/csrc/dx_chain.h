// dx_chain.h -- the register-resident Metropolis chain (device code shared by dangx_mhreg.hip and dangx_fused.hip).
#pragma once
#include "dx_args.h"


template <bool V> struct BoolTag { static constexpr bool value = V; };

// the chain's exp: dx::exp_nr = the library routine minus its range selects (dx_math.h); -DDX_CHAIN_LIBEXP restores
// the library call for A/B timing
#ifdef DX_CHAIN_LIBEXP
#define CEXP(x) exp(x)
#define CEXPS(x) exp(x)
#define CEXP1(x) exp(x)
#else
#define CEXP(x) exp_nr(x)      // |x| < 1e9: power law, Planck factor (mbb_z)
#define CEXPS(x) exp_sat(x)    // any x: the log-normal SED, -(ln(nu/nu_p)/w)^2/2 has no bound for a narrow width
#define CEXP1(x) exp_nr_v(x)   // a call site that runs once per proposal (dx_math.h: fma_vc); any x
#endif
// reciprocals of the rms and inside the modified-blackbody SED: v_rcp_f64 + two Newton steps (<= 1 ulp, 6 vector
// instructions) instead of the IEEE division sequence (11); -DDX_CHAIN_IEEEDIV restores a / b
#ifdef DX_CHAIN_IEEEDIV
#define CDIV(a, b) ((a) / (b))
#else
#define CDIV(a, b) ((a) * fast_rcp(b))
#endif

// the chain's residual: with DX_CHAIN_SCALED (default) the staged planes hold d/sigma and a/sigma (the amplitude is fixed
// during an index chain), so a band costs r = d' - a'*s; acc += r*r  (2 instructions per plane instead of 4) and the
// factor -1/2 is applied to the band sum; -DDX_CHAIN_UNSCALED restores ((d - a*s)/sigma), acc -= r*r/2 for A/B timing.
// Both forms carry the rounding of s scaled by the pixel's signal to noise; they differ in the last bits of lnL only.
#ifndef DX_CHAIN_UNSCALED
#define DX_CHAIN_SCALED 1
#endif
// lane-pair kernels: each lane draws the random numbers of every second step for both (1), or both draw all of them (0: A/B)
#ifndef DX_CHAIN_PAIR_RNG
#define DX_CHAIN_PAIR_RNG 1
#endif

namespace {


// ---------------------------------------------------------------------------
// Register-resident form of the same chain (chisq likelihood, delta bandpasses, CH_POW / CH_MBB_BETA /
// CH_MBB_T) for compile-time band count NB and plane count SP: the cleaned data, 1/rms and the chain-
// invariant SED factor live in VGPRs (statically indexed, fully unrolled), per-band constants in SGPRs,
// and the kernel uses no LDS and no barrier.  The CU's vector register file (512 KB) is three times its
// LDS, so this form runs at 2-3 waves/SIMD where the LDS-column form is capped at 1-2.
// Arithmetic and operation order are identical to index_chain<MODE, SP, TB>.
//
// LP = 2 splits the bands of ONE pixel over two adjacent lanes (lane h of the pair owns bands [h*NB, (h+1)*NB) of the
// 2*NB bands): each lane stages and evaluates its half, the two partial band sums of lnL are exchanged with one
// cross-lane add per plane, and everything else (random numbers, prior, accept test) is computed by both lanes on
// identical inputs, so the pair never diverges.  It halves the registers a lane needs -- two planes of 20 bands drop
// from ~340 registers (one wave per SIMD) to the footprint of the 10-band kernel (two waves) -- at the price of the
// duplicated per-proposal work; the per-band constants then differ between the lanes of a pair and live in vector
// registers (K1, K2) instead of being scalar operands.
template <int LP>
struct BandPick {
    int half;  // which half of the bands this lane owns (always 0 for LP == 1)
    // per-band model constant for the lane's band j (j static): a scalar operand for LP == 1, else a select of two scalars
    __device__ __forceinline__ double operator()(const double* arr, int j, int nbh) const {
        if (LP == 1) return arr[j];
        const double a = arr[j], b = arr[nbh + j];
        return half ? b : a;
    }
    __device__ __forceinline__ double nu_c(const Model& M, int j, int nbh) const {
        if (LP == 1) return M.band[j].nu_c;
        const double a = M.band[j].nu_c, b = M.band[nbh + j].nu_c;
        return half ? b : a;
    }
};

// KT: a lane pair's per-band constants are read from a table in LDS (kt1 / kt2 point at the lane's first band) instead of being
// held in 2*NB registers -- for kernels that have the block's constant table anyway (k_plane_set)
// BP (LP == 1 only): some bands are bandpass-integrated (bp%id /= 'delta', src/dang_component_mod.f90:909-913, 949-954): such a
// band's SED is the tau-weighted sum over its samples in sample order, evaluated by a run-time loop around the same per-sample
// expressions -- sample tables (nu0, tau0, log(nu0/nu_ref)) through scalar loads, the reciprocal of the Planck denominator as in
// the delta form.  What is chain invariant per SAMPLE (the other index's exponential of a modified blackbody) has nowhere to
// live (85 values per pixel in bench.py --bandpass 16) and is evaluated again in every proposal: two exponentials per sample
// instead of one for the mbb sweeps.
template <int MODE, int SP, int NB, int LP, bool KT = false, bool BP = false>
struct RegChain {
    static constexpr bool kBP = BP;
    double D[SP][NB], F[NB], ISr[SP][NB];  // cleaned data, chain-invariant SED factor, 1/rms  (scaled form: d/rms, amp/rms)
    double bpa, bpb;                       // BP: CH_MBB_BETA: z = h/(k T), exp(z nu_ref) - 1; CH_MBB_T: beta + 1
    double K1[(LP > 1 && !KT) ? NB : 1], K2[(LP > 1 && !KT) ? NB : 1];  // LP > 1: the lane's per-band constants (see k1 / k2)
    const double *kt1, *kt2;
    double amp[SP];

    __device__ __forceinline__ double is(int kk, int j) const { return ISr[kk][j]; }
    __device__ __forceinline__ void set_is(int kk, int j, double v) { ISr[kk][j] = v; }
    // the constant that multiplies / offsets the sampled parameter at band j, and the band's constant factor
    __device__ __forceinline__ double k1(const Model& M, const Comp& c, int j) const {
        if (KT) return kt1[j];
        if (LP > 1) return K1[j];
        return (MODE == CH_MBB_T) ? M.band[j].nu_c : (MODE == CH_LOGN_NUP) ? c.lnu9[j] : c.lnr[j];
    }
    __device__ __forceinline__ double k2(const Comp& c, int j) const { return KT ? kt2[j] : (LP > 1) ? K2[j] : c.cst[j]; }
    // KT: rows of the block's table (dx_sed.h: sed_table_build with `ng` row blocks) for group member g, from the lane's band jb
    __device__ __forceinline__ void set_kt(const double* tab, int nbands, int ng, int g, int jb) {
        kt1 = tab + ((MODE == CH_MBB_T) ? TROWS * ng : (MODE == CH_LOGN_NUP) ? TROWS * g + 2 : TROWS * g) * nbands + jb;
        kt2 = tab + (TROWS * g + 1) * nbands + jb;
    }
    __device__ __forceinline__ void set_k(const Model& M, const Comp& c, const BandPick<LP>& pick) {
        if (LP > 1 && !KT) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                K1[j] = (MODE == CH_MBB_T) ? pick.nu_c(M, j, NB) : (MODE == CH_LOGN_NUP) ? pick(c.lnu9, j, NB) : pick(c.lnr, j, NB);
                K2[j] = (MODE == CH_LOGN_NUP || MODE == CH_LOGN_W) ? pick(c.cst, j, NB) : 0.0;
            }
        }
    }

    // BATCH (CH_MBB_T only): the tile's Planck denominators share one reciprocal -- chain_finish decides once per chain
    // OP: what the band loop does with the component's signal a/rms * s (the resident-residual form of k_plane_set, LNL_ADD /
    // LNL_SUB: D holds the FULL residual (d - sum of all members) / rms between the sweeps of a launch):
    //   LNL_EVAL  r = D - a' s, acc += r^2                      -- a likelihood evaluation (every proposal)
    //   LNL_ADD   r = D, D += a' s, acc += r^2                  -- the chain's FIRST evaluation, at the current index values: the
    //             member's own signal goes back into the cleaned data, and lnL of the current state is the residual's
    //   LNL_SUB   D -= a' s                                      -- after the chain, at the values it ended on: the residual again
    template <bool BATCH, int OP = 0>
    __device__ __forceinline__ double lnl(const Model& M, const Comp& c, double th, double other, double& acc0, double& acc1) {
        double s0 = 0.0, s1 = 0.0;
        if (MODE == CH_POW) s0 = th;
        else if (MODE == CH_MBB_BETA) s0 = th + 1.0;
        else if (MODE == CH_MBB_T) { s0 = mbb_z(th); s1 = CEXP(s0 * c.nu_ref) - 1.0; }
        else if (MODE == CH_LOGN_NUP) { s0 = log_pos(th); s1 = other; }  // log(nu/(nu_p*1e9)) = lnu9 - log(nu_p)
        else s1 = th;  // CH_LOGN_W
        acc0 = 0.0; acc1 = 0.0;
        // log-normal: ln(nu/nu_p)/w as a product with 1/w (v_rcp_f64 + two Newton steps, once per evaluation) instead of one
        // IEEE division per band -- the expression the amplitude kernels use for the same SED (sed_tile), <= 1 ulp from it
#ifdef DX_CHAIN_IEEEDIV
        const double rs1 = 1.0 / s1;
#else
        const double rs1 = (MODE == CH_LOGN_NUP || MODE == CH_LOGN_W) ? fast_rcp(s1) : 0.0;
#endif
        // bands in tiles of TT: TT independent exp chains interleave, then accumulate in band order
        // (BP: one band at a time -- a bandpass-integrated band interleaves its own samples)
        constexpr int TT = BP ? 1 : (NB % 5 == 0) ? 5 : (NB % 4 == 0) ? 4 : (NB % 3 == 0) ? 3 : 1;
        static_assert(!BP || (LP == 1 && (MODE == CH_POW || MODE == CH_MBB_BETA || MODE == CH_MBB_T)), "bandpass chains: one lane, power law / mbb");
#pragma unroll
        for (int j0 = 0; j0 < NB; j0 += TT) {
            double s[TT];
            if (BP && M.band[j0].n != 0) {   // wave-uniform: LP == 1, j0 is the band number
                const Band& b = M.band[j0];
                const kptr nu = as_const(M.bp_nu0 + b.off);
                const kptr tau = as_const(M.bp_tau0 + b.off);
                const kptr lnr = as_const(c.bp_lnr + b.off);
                const int n = b.n;
                double sj = 0.0;
                if (MODE == CH_POW) {               // :909-913
#pragma unroll 4
                    for (int q = 0; q < n; ++q) sj = sj + tau[q] * CEXP(s0 * lnr[q]);
                } else if (MODE == CH_MBB_BETA) {   // :949-954, T fixed: tau * A / (e^{z nu} - 1) * (nu/nu_ref)^(beta+1)
#pragma unroll 4
                    for (int q = 0; q < n; ++q) sj = sj + (tau[q] * bpb) * fast_rcp(CEXP(bpa * nu[q]) - 1.0) * CEXP(s0 * lnr[q]);
                } else {                            // CH_MBB_T, beta fixed
#pragma unroll 4
                    for (int q = 0; q < n; ++q) sj = sj + (tau[q] * s1) * fast_rcp(CEXP(s0 * nu[q]) - 1.0) * CEXP(bpa * lnr[q]);
                }
                s[0] = sj;
            } else
            if (MODE == CH_MBB_T && TT > 1 && BATCH) {
                // one reciprocal for the tile's TT Planck denominators (prefix products, invert the last, peel backwards):
                // 3(TT-1) multiplications and one v_rcp_f64 + Newton instead of TT of them (v_rcp_f64 issues at a quarter of
                // the fma rate: 6 % of a temperature proposal).  Each 1/den_t carries <= 2(TT-1) more roundings.  Only for
                // chains that cannot reach a temperature where the product of TT denominators overflows (chain_finish).
                double den[TT], pre[TT];
#pragma unroll
                for (int t = 0; t < TT; ++t) den[t] = CEXP(s0 * k1(M, c, j0 + t)) - 1.0;
                pre[0] = den[0];
#pragma unroll
                for (int t = 1; t < TT; ++t) pre[t] = pre[t - 1] * den[t];
                double inv = s1 * fast_rcp(pre[TT - 1]);
#pragma unroll
                for (int t = TT - 1; t > 0; --t) {
                    s[t] = (inv * pre[t - 1]) * F[j0 + t];
                    inv *= den[t];
                }
                s[0] = inv * F[j0];
            } else
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const int j = j0 + t;
                if (MODE == CH_LOGN_NUP) {
                    const double l = (k1(M, c, j) - s0) * rs1;
                    s[t] = CEXPS(-0.5 * (l * l)) * k2(c, j);
                } else if (MODE == CH_LOGN_W) {
                    const double l = F[j] * rs1;
                    s[t] = CEXPS(-0.5 * (l * l)) * k2(c, j);
                } else {
                    const double e = CEXP(s0 * k1(M, c, j));
                    if (MODE == CH_POW) s[t] = e;
                    else if (MODE == CH_MBB_BETA) s[t] = F[j] * e;
                    else s[t] = CDIV(s1, e - 1.0) * F[j];
                }
            }
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const int j = j0 + t;
#ifdef DX_CHAIN_SCALED
                if (OP == 1) {
                    const double r0 = D[0][j];
                    D[0][j] = fma(is(0, j), s[t], r0);
                    acc0 = fma(r0, r0, acc0);
                    if (SP == 2) {
                        const double r1 = D[SP - 1][j];
                        D[SP - 1][j] = fma(is(SP - 1, j), s[t], r1);
                        acc1 = fma(r1, r1, acc1);
                    }
                } else if (OP == 2) {
                    D[0][j] = fma(-is(0, j), s[t], D[0][j]);
                    if (SP == 2) D[SP - 1][j] = fma(-is(SP - 1, j), s[t], D[SP - 1][j]);
                } else {
                const double r0 = fma(-is(0, j), s[t], D[0][j]);
                acc0 = fma(r0, r0, acc0);
                if (SP == 2) {
                    const double r1 = fma(-is(SP - 1, j), s[t], D[SP - 1][j]);
                    acc1 = fma(r1, r1, acc1);
                }
                }
#else
                const double r0 = (D[0][j] - amp[0] * s[t]) * is(0, j);
                acc0 = acc0 - 0.5 * (r0 * r0);
                if (SP == 2) {
                    const double r1 = (D[SP - 1][j] - amp[SP - 1] * s[t]) * is(SP - 1, j);
                    acc1 = acc1 - 0.5 * (r1 * r1);
                }
#endif
            }
        }
        if (OP == 2) return 0.0;
        if (LP > 1) {  // the other half's band sum: a + b on one lane, b + a on the other -- the same value
            acc0 += __shfl_xor(acc0, 1, 64);
            if (SP == 2) acc1 += __shfl_xor(acc1, 1, 64);
        }
#ifdef DX_CHAIN_SCALED
        acc0 *= -0.5; acc1 *= -0.5;
#endif
        return acc0 + acc1;
    }
    // after the other components are removed: d -> d/rms, 1/rms -> amp/rms
    __device__ __forceinline__ void scale() {
#ifdef DX_CHAIN_SCALED
#pragma unroll
        for (int kk = 0; kk < SP; ++kk)
#pragma unroll
            for (int j = 0; j < NB; ++j) { D[kk][j] *= ISr[kk][j]; ISr[kk][j] *= amp[kk]; }
#endif
    }
};

// eval_sed of an "other" component for all NB bands of one plane, subtracted from D (static band index)
template <int NB, int LP>
__device__ __forceinline__ void subtract_other(const Model& M, const Comp& c2, int k, double amp2, double t0, double t1,
                                               double (&Dk)[NB], const BandPick<LP>& pick) {
    if ((c2.const_planes >> (k - 1)) & 1) {  // spatially constant indices: host-evaluated SED
#pragma unroll
        for (int j = 0; j < NB; ++j) Dk[j] -= amp2 * pick(c2.csed[k - 1], j, NB);
        return;
    }
    const Prep pr = sed_prep(c2, t0, t1);
    switch (c2.type) {
    case DANGX_POWERLAW:
#pragma unroll
        for (int j = 0; j < NB; ++j) Dk[j] -= amp2 * CEXP(pr.p0 * pick(c2.lnr, j, NB));
        break;
    case DANGX_MBB:
#pragma unroll
        for (int j = 0; j < NB; ++j) Dk[j] -= amp2 * (CDIV(pr.p2, CEXP(pr.p1 * pick.nu_c(M, j, NB)) - 1.0) * CEXP(pr.p0 * pick(c2.lnr, j, NB)));
        break;
    case DANGX_FREEFREE:
#pragma unroll
        for (int j = 0; j < NB; ++j) Dk[j] -= amp2 * (ff_gaunt(pick(c2.lnu9, j, NB), pr.p0) / pr.p1 * pick(c2.cst, j, NB));
        break;
    case DANGX_LOGNORMAL: {
        const double rp1 = fast_rcp(pr.p1);  // as sed_tile (dangx_ampreg.hip) forms it
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const double l2 = (pick(c2.lnu9, j, NB) - pr.p2) * rp1;
            Dk[j] -= amp2 * (CEXPS(-0.5 * (l2 * l2)) * pick(c2.cst, j, NB));
        }
        break;
    }
    default:  // cmb
#pragma unroll
        for (int j = 0; j < NB; ++j) Dk[j] -= amp2 * pick(c2.cst, j, NB);
        break;
    }
}

// Q and U planes of one pixel whose "other" component has the same indices on both (always the case once a Q+U
// sweep has written them, :465): one SED evaluation per band serves both planes -- same values, same operations.
template <int NB, int LP>
__device__ __forceinline__ void subtract_other_pair(const Model& M, const Comp& c2, double ampa, double ampb, double t0, double t1,
                                                    double (&Da)[NB], double (&Db)[NB], const BandPick<LP>& pick) {
    const Prep pr = sed_prep(c2, t0, t1);
    switch (c2.type) {
    case DANGX_POWERLAW:
#pragma unroll
        for (int j = 0; j < NB; ++j) { const double s = CEXP(pr.p0 * pick(c2.lnr, j, NB)); Da[j] -= ampa * s; Db[j] -= ampb * s; }
        break;
    case DANGX_MBB:
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const double s = CDIV(pr.p2, CEXP(pr.p1 * pick.nu_c(M, j, NB)) - 1.0) * CEXP(pr.p0 * pick(c2.lnr, j, NB));
            Da[j] -= ampa * s; Db[j] -= ampb * s;
        }
        break;
    case DANGX_FREEFREE:
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const double s = ff_gaunt(pick(c2.lnu9, j, NB), pr.p0) / pr.p1 * pick(c2.cst, j, NB);
            Da[j] -= ampa * s; Db[j] -= ampb * s;
        }
        break;
    case DANGX_LOGNORMAL: {
        const double rp1 = fast_rcp(pr.p1);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const double l2 = (pick(c2.lnu9, j, NB) - pr.p2) * rp1;
            const double s = CEXPS(-0.5 * (l2 * l2)) * pick(c2.cst, j, NB);
            Da[j] -= ampa * s; Db[j] -= ampb * s;
        }
        break;
    }
    default:  // cmb
#pragma unroll
        for (int j = 0; j < NB; ++j) { const double s = pick(c2.cst, j, NB); Da[j] -= ampa * s; Db[j] -= ampb * s; }
        break;
    }
}

// Second half of a register chain, from staged planes to the written index map: R holds the cleaned data and 1/rms of
// the lane's bands and the component's amplitudes; sample0/1 are its two current index values on the first plane.
// SCALE = false: the planes are already d/rms and amp/rms (a second chain of the same component on the same planes,
// k_index_mh_pair); final_value (nullable) receives the value the chain ends at.
// ADD / SUB (k_plane_set's resident-residual form, see RegChain::lnl): R.D arrives as the full residual and the first evaluation
// puts the component's own signal back (ADD); after the chain the signal at the values it ended on is taken out again (SUB).
template <int MODE, int SP, int NB, int LP, bool SCALE = true, bool ADD = false, bool SUB = false, class RC>
__device__ __forceinline__ unsigned long long chain_finish(const Model& M, const IndexArgs& a, const Comp& c, RC& R,
                                                           const BandPick<LP>& pick, double sample0, double sample1, int i, int half,
                                                           double chi[4], double* final_value = nullptr,
                                                           const double* acc_in = nullptr, double* acc_out = nullptr) {
    // acc_in (k_plane_set, second chain of a component's pair): the likelihood sums of the state this chain starts from, as the
    // chain before it left them -- the same state, the same SED (another factorisation of it): no first evaluation.
    // acc_out: the sums of the state the chain ends on.
    const int npix = M.npix;
    double* out = c.idx + ((long long)a.nind * M.nmaps) * npix + i;
    const bool first = (a.nind == 0);
    if (SCALE) R.scale();
    // --- chain-invariant SED factor
    if (MODE == CH_MBB_BETA) {
        const double z = mbb_z(sample1);
        const double A = CEXP(z * c.nu_ref) - 1.0;
#pragma unroll
        for (int j = 0; j < NB; ++j) R.F[j] = CDIV(A, CEXP(z * pick.nu_c(M, j, NB)) - 1.0);
        if (RC::kBP) { R.bpa = z; R.bpb = A; }   // bandpass-integrated bands: the same two numbers per SAMPLE (RegChain::lnl)
    } else if (MODE == CH_MBB_T) {
#pragma unroll
        for (int j = 0; j < NB; ++j) R.F[j] = CEXP((sample0 + 1.0) * pick(c.lnr, j, NB));
        if (RC::kBP) R.bpa = sample0 + 1.0;
    } else if (MODE == CH_LOGN_W) {
        {
            const double lp = log_pos(sample0);
#pragma unroll
            for (int j = 0; j < NB; ++j) R.F[j] = pick(c.lnu9, j, NB) - lp;
        }
    }
    const double other = first ? sample1 : sample0;  // the index that is not sampled
    // --- chain (gaussian / uniform prior inline; jeffreys falls back to the LDS form on the host side)
    const unsigned long long gpix = (unsigned long long)(M.pix0 + i);
    const int q = a.nind;
    const bool gauss = c.prior_type[q] == DANGX_PRIOR_GAUSSIAN;
    const double pmean = c.gauss[q][0], pstd = c.gauss[q][1], lgden = c.lgden[q];
    // 1/(2 sigma^2) once per sweep; the proposal then multiplies ((x/d and x*(1/d) differ by <= 1 ulp of the prior term;
    // -DDX_CHAIN_IEEEDIV restores the division: 15 instructions of a ~400-instruction proposal)
    const double inv2v = 1.0 / (2 * (pstd * pstd));
    auto prior = [&](double v) -> double {
        if (!gauss) return 0.0;
#ifdef DX_CHAIN_IEEEDIV
        const double arg = ((v - pmean) * (v - pmean)) / (2 * (pstd * pstd));
#else
        const double arg = ((v - pmean) * (v - pmean)) * inv2v;
#endif
        return (arg > 745.0) ? -INFINITY : -arg - lgden;
    };
    unsigned long long nacc = 0;
    double cur = first ? sample0 : sample1;
    double a0, a1, c0, c1;
    const double step = c.step[q], lo = c.uni[q][0], hi = c.uni[q][1];
    auto chain = [&](auto batch_tag) {
        constexpr bool B = decltype(batch_tag)::value;
        double lnl;
        if (acc_in) { a0 = acc_in[0]; a1 = acc_in[1]; lnl = a0 + a1; }
        else lnl = R.template lnl<B, ADD ? 1 : 0>(M, c, cur, other, a0, a1);
        chi[0] = -2.0 * a0; chi[1] = -2.0 * a1;
        double lnl_old = lnl + prior(cur);
        if (LP == 1 || DX_CHAIN_PAIR_RNG == 0) {
            for (int l = 1; l <= a.nsample; ++l) {
                double u1, u2, u3;
                uniform3(a.seed, a.stream, gpix, (uint32_t)l, u1, u2, u3);
                const double prop = cur + rand_normal(0.0, step, u1, u2);  // :414
                if (prop < lo || prop > hi) continue;                      // :415
                lnl = R.template lnl<B>(M, c, prop, other, c0, c1);
                const double lnl_new = lnl + prior(prop);
                const double diff = lnl_new - lnl_old;
                const bool acc = (a.ml_mode == DANGX_ML_OPTIMIZE) ? (diff > 0.0) : ((diff >= 0.0) || (CEXP1(diff) > u3));  // :443-454
                if (acc) { cur = prop; lnl_old = lnl_new; a0 = c0; a1 = c1; ++nacc; }
            }
        } else {
            // Lane pairs: both lanes of a pixel would draw the SAME numbers for every step (a third of a proposal's instructions).
            // Instead lane h draws for step l + h, and the two steps take their numbers from the lane that made them: the random
            // numbers of a pixel are computed once per TWO steps -- the same draws, the same arithmetic, half the instructions.
            auto mh_step = [&](double g, double u3) {   // one step from the proposal deviate g = rand_normal(0, step) and the accept uniform
                const double prop = cur + g;                               // :414
                if (prop < lo || prop > hi) return;                        // :415
                lnl = R.template lnl<B>(M, c, prop, other, c0, c1);
                const double lnl_new = lnl + prior(prop);
                const double diff = lnl_new - lnl_old;
                const bool acc = (a.ml_mode == DANGX_ML_OPTIMIZE) ? (diff > 0.0) : ((diff >= 0.0) || (CEXP1(diff) > u3));  // :443-454
                if (acc) { cur = prop; lnl_old = lnl_new; a0 = c0; a1 = c1; ++nacc; }
            };
            for (int l = 1; l <= a.nsample; l += 2) {
                double u1, u2, u3;
                uniform3(a.seed, a.stream, gpix, (uint32_t)(l + half), u1, u2, u3);
                const double g = rand_normal(0.0, step, u1, u2);
                const double go = __shfl_xor(g, 1, 64), uo = __shfl_xor(u3, 1, 64);
                mh_step(half == 0 ? g : go, half == 0 ? u3 : uo);                          // step l: the even lane's numbers
                if (l + 1 <= a.nsample) mh_step(half == 0 ? go : g, half == 0 ? uo : u3);  // step l + 1: the odd lane's
            }
        }
        if (SUB) { double u0_, u1_; (void)R.template lnl<B, 2>(M, c, cur, other, u0_, u1_); }
    };
#ifdef DX_CHAIN_NO_BATCH_RCP
    chain(BoolTag<false>{});
#else
    if (MODE == CH_MBB_T && !RC::kBP) {
        // The chain evaluates the SED at its starting temperature and at proposals inside the hard bounds (:415) only.  If the
        // lowest of those keeps the sum of h nu / (k T) over a tile of five bands below 700, no product of five Planck
        // denominators can overflow and the whole chain takes the batched form; otherwise (T < 0.3 K at 857 GHz, T <= 0) the
        // one-by-one form -- decided once per chain, not per proposal (a branch inside the proposal costs what it saves).
        const double tmin = fmin(lo, cur);
#ifdef DX_BATCH_ALWAYS
        chain(BoolTag<true>{});
#elif defined(DX_BATCH_LANEWISE)
        if (tmin > 0.0 && mbb_z(tmin) * M.mbb_batch_z < 1.0) chain(BoolTag<true>{});
        else chain(BoolTag<false>{});
#else
        // (one decision per WAVEFRONT: the one-by-one form is right for every lane, and a branch the scalar unit takes
        // keeps the loop's control flow free of execution-mask bookkeeping)
        const bool unsafe = !(tmin > 0.0 && mbb_z(tmin) * M.mbb_batch_z < 1.0);
        if (__builtin_amdgcn_ballot_w64(unsafe) == 0ull) chain(BoolTag<true>{});
        else chain(BoolTag<false>{});
#endif
    } else {
        chain(BoolTag<false>{});
    }
#endif
    if (final_value) *final_value = cur;
    if (acc_out) { acc_out[0] = a0; acc_out[1] = a1; }
    if (half != 0) {  // the pair's second lane carries the same chain: its sums and counts are the first lane's
        chi[0] = chi[1] = 0.0;
        return 0ull;
    }
#pragma unroll
    for (int kk = 0; kk < SP; ++kk) out[(long long)(a.s1 + kk - 1) * npix] = cur;  // :465, :483
    chi[2] = -2.0 * a0; chi[3] = -2.0 * a1;
    return nacc;
}

// Staging of a register chain: the component's index values, data_raw (:173-177) and 1/rms of the lane's bands, every
// other component removed (:180-196).  The caller has dealt with masked pixels.
template <int MODE, int SP, int NB, int LP>
__device__ __forceinline__ void index_chain_stage(const Model& M, const IndexArgs& a, const Comp& c, RegChain<MODE, SP, NB, LP>& R,
                                                  const BandPick<LP>& pick, int i, double& sample0, double& sample1) {
    const int npix = M.npix;
    const int jb = pick.half * NB;  // first band of this lane
    R.set_k(M, c, pick);
    load_theta(M, c, i, a.s1, sample0, sample1);  // sample(l) = c%indices(i, map_inds(1), l), :372-377
    // --- stage data_raw (:173-177) and rms: every load issued before the first use
    const long long bstride = (long long)M.nmaps * npix;
#pragma unroll
    for (int kk = 0; kk < SP; ++kk) {
        const int k = a.s1 + kk;
        R.amp[kk] = c.amp[(long long)(k - 1) * npix + i];
        const double* sigp = M.sig + (long long)(k - 1) * npix + i;
        const double* rmsp = M.rms + (long long)(k - 1) * npix + i;
        double rv[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            R.D[kk][j] = sigp[(jb + j) * bstride];
            rv[j] = rmsp[(jb + j) * bstride];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (k == 1) R.D[kk][j] = (R.D[kk][j] - pick(M.offset, j, NB)) / pick(M.gain, j, NB);
            R.set_is(kk, j, CDIV(1.0, rv[j]));
        }
    }
    // --- remove every OTHER component (:180-196) in component_list order, next one prefetched
    {
        unsigned om = a.others;
        double na[SP], nt0[SP], nt1[SP];
        auto fetch = [&](int l) {
            const Comp& c2 = M.comp[l];
#pragma unroll
            for (int kk = 0; kk < SP; ++kk) {
                na[kk] = c2.amp[(long long)(a.s1 + kk - 1) * npix + i];
                nt0[kk] = nt1[kk] = 0.0;
                if (!((c2.const_planes >> (a.s1 + kk - 1)) & 1)) load_theta(M, c2, i, a.s1 + kk, nt0[kk], nt1[kk]);
            }
        };
        int l = om ? __builtin_ctz(om) : -1;
        if (l >= 0) fetch(l);
        while (l >= 0) {
            const Comp& c2 = M.comp[l];
            double ca[SP], ct0[SP], ct1[SP];
#pragma unroll
            for (int kk = 0; kk < SP; ++kk) { ca[kk] = na[kk]; ct0[kk] = nt0[kk]; ct1[kk] = nt1[kk]; }
            om &= om - 1;
            const int ln = om ? __builtin_ctz(om) : -1;
            if (ln >= 0) fetch(ln);
            const unsigned cp = (c2.const_planes >> (a.s1 - 1)) & 3u;
            if (SP == 2 && cp == 0 && ct0[0] == ct0[SP - 1] && ct1[0] == ct1[SP - 1]) {
                subtract_other_pair<NB, LP>(M, c2, ca[0], ca[SP - 1], ct0[0], ct1[0], R.D[0], R.D[SP - 1], pick);
            } else {
#pragma unroll
                for (int kk = 0; kk < SP; ++kk) subtract_other<NB, LP>(M, c2, a.s1 + kk, ca[kk], ct0[kk], ct1[kk], R.D[kk], pick);
            }
            l = ln;
        }
    }
}

// NB = bands per lane (all of them for LP == 1, half for LP == 2); half = which half this lane owns
template <int MODE, int SP, int NB, int LP>
__device__ __forceinline__ unsigned long long index_chain_reg(const Model& M, const IndexArgs& a, int i, int half, double chi[4]) {
    const int npix = M.npix;
    const Comp& c = M.comp[a.comp];
    double* out = c.idx + ((long long)a.nind * M.nmaps) * npix + i;
    if (is_masked(M.mask[i])) {  // :362 cycle; index_map stays 0 (:223) and is copied back (:480-483)
        if (half == 0) {
#pragma unroll
            for (int kk = 0; kk < SP; ++kk) out[(long long)(a.s1 + kk - 1) * npix] = 0.0;
        }
        return 0ull;
    }
    const BandPick<LP> pick = {half};
    RegChain<MODE, SP, NB, LP> R;
    double sample0, sample1;
    index_chain_stage<MODE, SP, NB, LP>(M, a, c, R, pick, i, sample0, sample1);
    return chain_finish<MODE, SP, NB, LP>(M, a, c, R, pick, sample0, sample1, i, half, chi);
}

// Two consecutive sweeps of ONE component on the same planes -- index nind, then index nind + 1 (the dust beta and dust T
// sweeps of every configuration) -- in one pass: nothing the second sweep removes from the data has changed (only the
// component's own index did), so its staged planes ARE the first sweep's; it needs a new chain-invariant factor and its
// own chain.  Same numbers as the two sweeps, bit for bit (the second sweep would have staged exactly these planes).
template <int MODEA, int MODEB, int SP, int NB, int LP>
__device__ __forceinline__ void index_chain_pair(const Model& M, const IndexArgs& a, const IndexArgs& b, int i, int half, double chi[4],
                                                 unsigned long long& nacc_a, unsigned long long& nacc_b) {
    const int npix = M.npix;
    const Comp& c = M.comp[a.comp];
    nacc_a = nacc_b = 0ull;
    if (is_masked(M.mask[i])) {
        if (half == 0) {
#pragma unroll
            for (int kk = 0; kk < SP; ++kk) {
                c.idx[((long long)a.nind * M.nmaps + (a.s1 + kk - 1)) * npix + i] = 0.0;
                c.idx[((long long)b.nind * M.nmaps + (a.s1 + kk - 1)) * npix + i] = 0.0;
            }
        }
        return;
    }
    const BandPick<LP> pick = {half};
    RegChain<MODEA, SP, NB, LP> RA;
    double sample0, sample1;
    index_chain_stage<MODEA, SP, NB, LP>(M, a, c, RA, pick, i, sample0, sample1);
    double chia[4] = {0.0, 0.0, 0.0, 0.0}, chib[4] = {0.0, 0.0, 0.0, 0.0}, va;
    nacc_a = chain_finish<MODEA, SP, NB, LP>(M, a, c, RA, pick, sample0, sample1, i, half, chia, &va);
    RegChain<MODEB, SP, NB, LP> RB;
    RB.set_k(M, c, pick);
#pragma unroll
    for (int kk = 0; kk < SP; ++kk) {
        RB.amp[kk] = RA.amp[kk];
#pragma unroll
        for (int j = 0; j < NB; ++j) { RB.D[kk][j] = RA.D[kk][j]; RB.ISr[kk][j] = RA.ISr[kk][j]; }
    }
    if (a.nind == 0) sample0 = va; else sample1 = va;
    nacc_b = chain_finish<MODEB, SP, NB, LP, false>(M, b, c, RB, pick, sample0, sample1, i, half, chib);
    chi[0] = chia[0]; chi[1] = chia[1]; chi[2] = chib[2]; chi[3] = chib[3];
}

}  // namespace
