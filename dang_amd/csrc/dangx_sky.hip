// dangx_sky.hip -- the SKY-WIDE steps of the Gibbs loop behind the C ABI: the full-sky Metropolis chain of
// sample_index_mh (index_mode == 1, src/dang_sample_mod.f90:229-329), tune_spectral_parameter_length (:623-717), the
// 'Tuning!' block of the per-pixel branch (:337-346), fit_band_gain (:570-621) and the T_CMB update of
// sample_spectral_parameters (:75-78).
//
// In these steps one number describes the whole sky, so a Metropolis step is a pass over the maps that leaves a few sums
// (the kernels of dangx_coarse.hip / dangx_core.hip: k_fullsky_rows, k_gain_rows, k_index_plain_sum) and a few scalar operations between two
// such passes.  The scalar part lives HERE, once, for every host language: the Python mirror (dang_amd/api.py), the Fortran
// layers (fortran/dangx_multi_mod.f90, fortran/reference_side/dang_gpu_mod.f90) and a C driver all make the same call.
//
// Every entry point takes the contexts of THIS process in shard order (ctxs[0..nctx); nctx = 1 for a whole-sky context
// or for one context per process).  A sky-wide sum is the contexts' sums added in shard order, then summed over the
// ranks through ctxs[0]'s dangx_set_allreduce callback -- every rank runs the same chain on the same sums and draws the
// same keyed random numbers (Philox4x32-10 keyed by (seed, stream, pixel label 2^40 - 1, running draw counter), the
// host twin of csrc/dx_rng.h; the reference draws from an unseeded RANDOM_NUMBER, src/dang.f90:67).
#include "dx_host.h"

namespace {

// ---- keyed random streams on the host (same words as dx_rng.h: philox4x32_10, u53, u32; libm for log / sin / sqrt)
constexpr unsigned long long GLOBAL_PIX = 0xFFFFFFFFFFull;  // pixel label of sky-wide draws (no real pixel uses it)

void h_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void h_words(unsigned long long seed, unsigned long long stream, unsigned long long pix, uint32_t draw, uint32_t o[4]) {
    h_philox((uint32_t)pix, draw ^ ((uint32_t)(pix >> 32) << 16), (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed,
             (uint32_t)(seed >> 32), o);
}
double h_u53(uint32_t hi, uint32_t lo) {
    const unsigned long long x = ((unsigned long long)hi << 32) | lo;
    return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
double h_u32(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }
void h_uniform2(unsigned long long seed, unsigned long long stream, uint32_t draw, double& u1, double& u2) {
    uint32_t o[4];
    h_words(seed, stream, GLOBAL_PIX, draw, o);
    u1 = h_u53(o[0], o[1]); u2 = h_u53(o[2], o[3]);
}
void h_uniform3(unsigned long long seed, unsigned long long stream, uint32_t draw, double& u1, double& u2, double& u3) {
    uint32_t o[4];
    h_words(seed, stream, GLOBAL_PIX, draw, o);
    u1 = h_u53(o[0], o[1]); u2 = h_u32(o[2]); u3 = h_u32(o[3]);
}
// rand_normal, src/dang_util_mod.f90:100-110
double h_rand_normal(double mean, double stdev, double u1, double u2) {
    const double r = std::sqrt(-2.0 * std::log(u1));
    const double theta = 2.0 * PI * u2;
    return mean + stdev * r * std::sin(theta);
}
// log(eval_normal_prior(prop, mean, std)), src/dang_util_mod.f90:112-121 and its callers (:260, :309, :657, :691)
double h_log_normal_prior(double prop, double mean, double sd) {
    const double var = sd * sd;
    const double p = std::exp(-((prop - mean) * (prop - mean)) / (2 * var)) / (sd * std::sqrt(2.0 * PI));
    return (p > 0.0) ? std::log(p) : -INFINITY;
}

struct Sky {
    dangx_ctx* const* c;
    int n;
    dangx_ctx* root() const { return c[0]; }
};

// an error of context r becomes the error of the call: the caller reads dangx_last_error(ctxs[0])
int sky_err(const Sky& s, dangx_ctx* who) {
    if (who != s.root()) s.root()->err = who->err;
    return 1;
}
int sky_check(dangx_ctx* const* ctxs, int nctx) {
    if (!ctxs || nctx < 1) return 1;
    for (int r = 0; r < nctx; ++r)
        if (!ctxs[r]) return 1;
    for (int r = 1; r < nctx; ++r)
        if (ctxs[r]->dims.nbands != ctxs[0]->dims.nbands || ctxs[r]->dims.ncomp != ctxs[0]->dims.ncomp ||
            ctxs[r]->dims.nmaps != ctxs[0]->dims.nmaps)
            return fail(ctxs[0], "the contexts of one sky must share nbands / nmaps / ncomp");
    return 0;
}
// sum over the ranks (the contexts of this process are already added)
int sky_ranks(const Sky& s, double* buf, int n) {
    dangx_ctx* c0 = s.root();
    if (!c0->allreduce || n <= 0) return 0;
    if (c0->allreduce(c0->allreduce_user, buf, n)) return fail(c0, "the all-reduce callback reported an error");
    return 0;
}

// sky-wide row sums at theta (dangx_fullsky_sums on every context, added in shard order, then over the ranks)
int sky_rows(const Sky& s, int what, const double* theta, double* rows, int nrows) {
    std::vector<double> part((size_t)nrows);
    for (int q = 0; q < nrows; ++q) rows[q] = 0.0;
    for (int r = 0; r < s.n; ++r) {
        if (dangx_fullsky_sums(s.c[r], what, theta, part.data(), nrows)) return sky_err(s, s.c[r]);
        for (int q = 0; q < nrows; ++q) rows[q] += part[q];
    }
    return sky_ranks(s, rows, nrows);
}

// The chisq likelihood of a full-sky chain from ONE pass over the maps.  The index is one value for the whole sky, so the model of
// band j on plane k is a(p) s_j(theta) with a pixel-independent s_j: about the chain's starting point theta0, with r0 = (d - a
// s_j(theta0)) / sigma,
//   -2 lnL(theta) = sum_jk [ W0_jk - 2 ds_j U_jk + ds_j^2 V_jk ],  ds_j = s_j(theta) - s_j(theta0),
//   W0 = sum_p r0^2, U = sum_p r0 a / sigma, V = sum_p a^2 / sigma^2  (unmasked pixels; k_fullsky_rows, selector 3).
// Exact algebra, no cancellation near theta0 (W0 is chi^2 itself, the other terms are of the size of the change); every proposal of
// the chain and of the tuner then costs nb SED evaluations on the host instead of a pass over the maps and a sky-wide reduction:
// NUMSAMPLE + 1 passes per sweep become one.  DANGX_FULLSKY_STATS=0: every evaluation a pass (A/B, and the form the reference has).
struct SkyStats {
    bool on = false;
    int comp = 0, nb = 0, Sp = 0;
    std::vector<double> s0, rows;   // s_j(theta0); W0, U, V per (band, plane) as the kernel writes them
    std::vector<std::vector<double>> part;   // the same rows of every context of this process (before the sums over contexts and ranks)
};
int sky_stats(const Sky& s, int comp, int Sp, const double* theta0, SkyStats& st) {
    static const bool enabled = [] { const char* e = getenv("DANGX_FULLSKY_STATS"); return !(e && e[0] == '0'); }();
    dangx_ctx* c0 = s.root();
    const int ty = c0->desc[comp].type;
    st.on = false;
    if (!enabled || !(ty == DANGX_POWERLAW || ty == DANGX_MBB || ty == DANGX_FREEFREE || ty == DANGX_LOGNORMAL)) return 0;
    st.comp = comp; st.nb = c0->dims.nbands; st.Sp = Sp;
    const int nrows = 3 * st.nb * Sp;
    st.rows.assign((size_t)nrows, 0.0);
    st.part.assign((size_t)s.n, std::vector<double>((size_t)nrows, 0.0));
    for (int r = 0; r < s.n; ++r) {   // (sky_rows, keeping every context's own rows: its planes' chi^2 comes from them at the end)
        if (dangx_fullsky_sums(s.c[r], 3, theta0, st.part[(size_t)r].data(), nrows)) return sky_err(s, s.c[r]);
        for (int q = 0; q < nrows; ++q) st.rows[(size_t)q] += st.part[(size_t)r][(size_t)q];
    }
    if (sky_ranks(s, st.rows.data(), nrows)) return 1;
    st.s0.resize((size_t)st.nb);
    for (int j = 0; j < st.nb; ++j) st.s0[(size_t)j] = dx_host_band_sed(c0, comp, j, theta0[0], theta0[1]);
    st.on = true;
    return 0;
}
double stats_lnl(const Sky& s, const SkyStats& st, const double* theta) {
    double chi = 0.0;
    for (int j = 0; j < st.nb; ++j) {
        const double ds = dx_host_band_sed(s.root(), st.comp, j, theta[0], theta[1]) - st.s0[(size_t)j];
        for (int kk = 0; kk < st.Sp; ++kk) {
            const double* r = &st.rows[(size_t)3 * (j * st.Sp + kk)];
            chi = chi + (r[0] - 2.0 * ds * r[1] + ds * ds * r[2]);
        }
    }
    return -0.5 * chi;
}

// evaluate_lnL / evaluate_marginal_lnL over the whole sky (src/dang_lnl_mod.f90:126-182, 47-124)
int sky_lnl(const Sky& s, int lnl_type, const double* theta, int Sp, double* out, const SkyStats* st = nullptr) {
    const int nb = s.root()->dims.nbands;
    *out = 0.0;
    if (lnl_type == DANGX_LNL_CHISQ && st && st->on) { *out = stats_lnl(s, *st, theta); return 0; }
    if (lnl_type == DANGX_LNL_CHISQ) return sky_rows(s, 0, theta, out, 1);
    if (lnl_type == DANGX_LNL_MARGINAL) {
        std::vector<double> rows((size_t)2 * nb * Sp);
        if (sky_rows(s, 1, theta, rows.data(), 2 * nb * Sp)) return 1;
        double lnl = 0.0;
        for (int q = 0; q < nb * Sp; ++q) {  // j outer, k inner (:113-122)
            const double TNd = rows[2 * q], TNT = rows[2 * q + 1];
            lnl = lnl - 0.5 * TNd * (1.0 / TNT) * TNd;
        }
        *out = lnl;
    }
    return 0;
}

// the prior term of the full-sky chain (:260-268, :304-313): gaussian / jeffreys (eval_jeffreys_prior over the sky,
// src/dang_lnl_mod.f90:242-304) / uniform
int sky_prior(const Sky& s, const dangx_comp_desc& d, int nind, double val, double* out) {
    *out = 0.0;
    if (d.prior_type[nind] == DANGX_PRIOR_GAUSSIAN) {
        *out = h_log_normal_prior(val, d.gauss_prior[nind][0], d.gauss_prior[nind][1]);
    } else if (d.prior_type[nind] == DANGX_PRIOR_JEFFREYS) {
        const double th[2] = {val, 0.0};
        double sum = 0.0;
        if (sky_rows(s, 2, th, &sum, 1)) return 1;
        *out = (sum > 0.0) ? std::log(std::sqrt(sum)) : -INFINITY;
    }
    return 0;
}

// c%step_size(nind) of every context (the per-pixel chains read it from the device copy of the model)
void set_step(const Sky& s, int comp, int nind, double step) {
    for (int r = 0; r < s.n; ++r) {
        s.c[r]->desc[comp].step_size[nind] = step;
        s.c[r]->dirty = true;
    }
}

// tune_spectral_parameter_length, src/dang_sample_mod.f90:623-717, on data prepared by dangx_fullsky_prepare[_coarse].
// theta_init has two entries whatever nindices is (:628, 632).  The reference's `do while (.not. c%tuned(nind))` does not
// return when nothing is accepted any more (optimize mode at the optimum: the step only ever halves); stopped after 64
// rounds here, as in the oracle.
int tune(const Sky& s, int comp, int nind, int Sp, int nsample, int ml_mode, unsigned long long seed, unsigned long long stream,
         const double* theta_init, uint32_t* draw, int32_t* tuned, const SkyStats* st = nullptr) {
    dangx_ctx* c0 = s.root();
    const dangx_comp_desc& d = c0->desc[comp];
    double sample[2] = {theta_init[0], theta_init[1]}, theta[2] = {theta_init[0], theta_init[1]};
    double lnl = 0.0, lnl_new = 0.0, lnl_old = 0.0;
    const int lt = d.lnl_type[nind], pt = d.prior_type[nind];
    if (lt == DANGX_LNL_CHISQ || lt == DANGX_LNL_MARGINAL) {
        if (sky_lnl(s, lt, sample, Sp, &lnl, st)) return 1;
    } else if (lt == DANGX_LNL_PRIOR) {
        double u1, u2;
        h_uniform2(seed, stream, (*draw)++, u1, u2);
        sample[nind] = h_rand_normal(d.gauss_prior[nind][0], d.gauss_prior[nind][1], u1, u2);
    }
    if (pt == DANGX_PRIOR_GAUSSIAN) lnl_old = lnl + h_log_normal_prior(sample[nind], d.gauss_prior[nind][0], d.gauss_prior[nind][1]);
    else if (pt == DANGX_PRIOR_UNIFORM) lnl_old = lnl;
    double step = d.step_size[nind];
    for (int round = 0; !tuned[nind] && round < 64; ++round) {
        double accept = 0.0;
        for (int l = 1; l <= nsample; ++l) {
            double u1, u2, u3;
            h_uniform3(seed, stream, (*draw)++, u1, u2, u3);
            theta[nind] = sample[nind] + h_rand_normal(0.0, step, u1, u2);
            if (theta[nind] < d.uni_prior[nind][0] || theta[nind] > d.uni_prior[nind][1]) continue;
            if (lt == DANGX_LNL_CHISQ || lt == DANGX_LNL_MARGINAL) {
                if (sky_lnl(s, lt, theta, Sp, &lnl, st)) return 1;
            }
            if (pt == DANGX_PRIOR_GAUSSIAN) lnl_new = lnl + h_log_normal_prior(theta[nind], d.gauss_prior[nind][0], d.gauss_prior[nind][1]);
            else if (pt == DANGX_PRIOR_UNIFORM) lnl_new = lnl;
            const double diff = lnl_new - lnl_old, ratio = std::exp(diff);
            if ((ml_mode == DANGX_ML_OPTIMIZE && ratio > 1.0) || (ml_mode == DANGX_ML_SAMPLE && ratio > u3)) {
                sample[nind] = theta[nind];
                lnl_old = lnl_new;
                accept = accept + 1;
            }
            lnl = 0.0;  // :705
        }
        const int l_after = nsample + 1;  // the loop variable after the loop (:707)
        if (accept / l_after < (double)0.4f) step = step - 0.5 * step;
        else if (accept / l_after > (double)0.6f) step = step + 0.5 * step;
        else
            for (int q = 0; q < std::max(d.nindices, 1); ++q) tuned[q] = 1;  // c%tuned = .true. for ALL indices (:712)
        set_step(s, comp, nind, step);
    }
    return 0;
}

int planes_of(dangx_ctx* c0, int map_n, int& s1, int& Sp) {
    if (map_n == -1) { s1 = 2; Sp = 2; }
    else if (map_n >= 1 && map_n <= 3) { s1 = map_n; Sp = 1; }
    else return fail(c0, "There is something wrong with the poltype flag (map_n must be 1,2,3 or -1)");
    if (s1 + Sp - 1 > c0->dims.nmaps) return fail(c0, "map_n exceeds nmaps");
    return 0;
}

int check_index(dangx_ctx* c0, int comp, int nind) {
    if (comp < 0 || comp >= c0->dims.ncomp || !c0->comp_set[comp]) return fail(c0, "component index out of range / component not set");
    if (nind < 0 || nind >= c0->desc[comp].nindices) return fail(c0, "index number out of range");
    const int lt = c0->desc[comp].lnl_type[nind];
    if (lt < DANGX_LNL_CHISQ || lt > DANGX_LNL_PRIOR) return fail(c0, "bad lnl_type");
    return 0;
}

// data_raw minus every other component on every context (:173-196); with sample_nside /= nside the degraded data / rms /
// mask (:199-217) -- a coarse pixel's children are scattered over the shards, so the shards' child sums are added first
int prepare(const Sky& s, int comp, int map_n, int nside, int sample_nside) {
    dangx_ctx* c0 = s.root();
    if (sample_nside <= 0 || sample_nside == nside) {
        for (int r = 0; r < s.n; ++r)
            if (dx_fullsky_prepare_lazy(s.c[r], comp, map_n)) return sky_err(s, s.c[r]);
        return 0;
    }
    const bool whole = s.n == 1 && c0->dims.pix0 == 0 && c0->dims.npix == c0->dims.npix_global && !c0->allreduce;
    if (whole) {
        if (dangx_fullsky_prepare_coarse(c0, comp, map_n, nside, sample_nside)) return 1;
        return 0;
    }
    int64_t np = 0, ni = 0;
    if (dangx_coarse_sizes(c0, map_n, sample_nside, &np, &ni)) return 1;
    std::vector<double> sum((size_t)np, 0.0), part((size_t)np);
    for (int r = 0; r < s.n; ++r) {
        if (dangx_coarse_partials(s.c[r], comp, map_n, nside, sample_nside, part.data())) return sky_err(s, s.c[r]);
        for (int64_t q = 0; q < np; ++q) sum[(size_t)q] += part[(size_t)q];
    }
    if (sky_ranks(s, sum.data(), (int)np)) return 1;
    for (int r = 0; r < s.n; ++r)
        if (dangx_fullsky_finish_coarse(s.c[r], comp, map_n, nside, sample_nside, sum.data())) return sky_err(s, s.c[r]);
    return 0;
}

// c%indices(0, map_inds(1), :) (:240-242): pixel 0 lives on the first shard of the first rank
int first_pixel(const Sky& s, int comp, int s1, double out[2]) {
    out[0] = out[1] = 0.0;
    dangx_ctx* c0 = s.root();
    if (c0->dims.pix0 == 0 && dangx_peek_indices(c0, comp, s1, 0, out)) return 1;
    return sky_ranks(s, out, 2);
}

}  // namespace

extern "C" {

int dangx_fullsky_sample(dangx_ctx* const* ctxs, int nctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                         uint64_t stream, int nside, int sample_nside, int32_t* tuned, double* step_size, double* value,
                         int64_t* accepted) {
    if (sky_check(ctxs, nctx)) return 1;
    const Sky s{ctxs, nctx};
    dangx_ctx* c0 = s.root();
    int s1, Sp;
    if (check_index(c0, comp, nind) || planes_of(c0, map_n, s1, Sp)) return 1;
    if (ml_mode != DANGX_ML_SAMPLE && ml_mode != DANGX_ML_OPTIMIZE) return fail(c0, "bad ml_mode");
    const dangx_comp_desc& d = c0->desc[comp];
    int32_t all_tuned[DANGX_MAX_IND] = {1, 1};
    if (!tuned) tuned = all_tuned;
    if (prepare(s, comp, map_n, nside, sample_nside)) return 1;
    double first[2], sample[2], theta[2];
    if (first_pixel(s, comp, s1, first)) return 1;
    sample[0] = theta[0] = first[0]; sample[1] = theta[1] = first[1];
    const int lt = d.lnl_type[nind];
    double lnl = 0.0, pr = 0.0;
    bool sample_it = true;
    SkyStats st;
    if (lt == DANGX_LNL_CHISQ && sky_stats(s, comp, Sp, first, st)) return 1;
    if (lt == DANGX_LNL_CHISQ || lt == DANGX_LNL_MARGINAL) {
        if (sky_lnl(s, lt, sample, Sp, &lnl, &st)) return 1;
    } else {  // 'prior': one draw from the gaussian prior, no chain (:255-257)
        double u1, u2;
        sample_it = false;
        h_uniform2(seed, stream, 0u, u1, u2);
        sample[nind] = h_rand_normal(d.gauss_prior[nind][0], d.gauss_prior[nind][1], u1, u2);
    }
    if (sky_prior(s, d, nind, sample[nind], &pr)) return 1;
    double lnl_old = lnl + pr;
    int64_t nacc = 0;
    if (sample_it) {
        if (!tuned[nind]) {  // :272-275
            uint32_t draw = 1;
            if (tune(s, comp, nind, Sp, nsample, ml_mode, seed, stream ^ 0x5555555555555555ull, sample, &draw, tuned, &st)) return 1;
        }
        sample[0] = theta[0] = first[0]; sample[1] = theta[1] = first[1];
        const double step = c0->desc[comp].step_size[nind];
        for (int l = 1; l <= nsample; ++l) {  // :282-324
            double u1, u2, u3;
            h_uniform3(seed, stream, (uint32_t)l, u1, u2, u3);
            theta[nind] = sample[nind] + h_rand_normal(0.0, step, u1, u2);
            if (theta[nind] < d.uni_prior[nind][0] || theta[nind] > d.uni_prior[nind][1]) continue;
            if (sky_lnl(s, lt, theta, Sp, &lnl, &st) || sky_prior(s, d, nind, theta[nind], &pr)) return 1;
            const double lnl_new = lnl + pr;
            const double diff = lnl_new - lnl_old, ratio = std::exp(diff);
            if ((ml_mode == DANGX_ML_OPTIMIZE && ratio > 1.0) || (ml_mode == DANGX_ML_SAMPLE && ratio > u3)) {
                sample[nind] = theta[nind];
                lnl_old = lnl_new;
                ++nacc;
            }
        }
    }
    for (int r = 0; r < nctx; ++r)  // :329, :483: every pixel, masked ones too
        if (dangx_fill_index(ctxs[r], comp, nind, map_n, sample[nind])) return sky_err(s, ctxs[r]);
    // chi^2 of the swept planes at the value the chain ended on, from the statistics: -2 lnL without the prior, per context its own
    // pixels' share -- what a per-pixel sweep's launch leaves behind, so compute_chisq needs no pass of its own (map resolution only:
    // the coarse chain's likelihood runs over degraded maps)
    if (st.on && sample_it && (sample_nside <= 0 || sample_nside == nside)) {
        std::vector<double> ds((size_t)st.nb);
        for (int j = 0; j < st.nb; ++j) ds[(size_t)j] = dx_host_band_sed(c0, comp, j, sample[0], sample[1]) - st.s0[(size_t)j];
        for (int r = 0; r < nctx; ++r)
            for (int kk = 0; kk < Sp; ++kk) {
                double chi = 0.0;
                for (int j = 0; j < st.nb; ++j) {
                    const double* q = &st.part[(size_t)r][(size_t)3 * (j * Sp + kk)];
                    chi = chi + (q[0] - 2.0 * ds[(size_t)j] * q[1] + ds[(size_t)j] * ds[(size_t)j] * q[2]);
                }
                if (dx_set_chi_after(ctxs[r], s1 + kk, chi)) return sky_err(s, ctxs[r]);
            }
    }
    if (step_size) *step_size = c0->desc[comp].step_size[nind];
    if (value) *value = sample[nind];
    if (accepted) *accepted = nacc;
    return 0;
}

int dangx_tune_step_size(dangx_ctx* const* ctxs, int nctx, int comp, int nind, int nsample, int ml_mode, uint64_t seed, uint64_t stream,
                         const double* theta_init, uint32_t* draw, int32_t* tuned, double* step_size) {
    if (sky_check(ctxs, nctx) || !theta_init || !draw || !tuned) return 1;
    const Sky s{ctxs, nctx};
    dangx_ctx* c0 = s.root();
    if (check_index(c0, comp, nind)) return 1;
    for (int r = 0; r < nctx; ++r)
        if (ctxs[r]->fs_comp != comp) return fail(c0, "dangx_fullsky_prepare has not been called for this component");
    if (tune(s, comp, nind, c0->fs_s2 - c0->fs_s1 + 1, nsample, ml_mode, seed, stream, theta_init, draw, tuned)) return 1;
    if (step_size) *step_size = c0->desc[comp].step_size[nind];
    return 0;
}

// The 'Tuning!' block of the per-pixel branch, src/dang_sample_mod.f90:337-346: one pass of the tuner per index of the
// component, pass l starting at sample(l) = sum(c%indices(:,map_inds(1),l)) / sum(mask(:,1)) with the entries not yet reached
// still 0 -- every pixel enters both sums and the mask's VALUES are summed.  The tuner marks all indices tuned, so later
// passes only evaluate their starting likelihood.
int dangx_tune_perpixel(dangx_ctx* const* ctxs, int nctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                        uint64_t stream, int32_t* tuned, double* step_size) {
    if (sky_check(ctxs, nctx) || !tuned) return 1;
    const Sky s{ctxs, nctx};
    dangx_ctx* c0 = s.root();
    int s1, Sp;
    if (check_index(c0, comp, nind) || planes_of(c0, map_n, s1, Sp)) return 1;
    if (prepare(s, comp, map_n, 0, 0)) return 1;  // data_raw minus every other component, :173-196
    double sample[2] = {0.0, 0.0};
    uint32_t draw = 1;
    for (int q = 0; q < c0->desc[comp].nindices; ++q) {
        double sums[2] = {0.0, 0.0};
        for (int r = 0; r < nctx; ++r) {
            double si, sm;
            if (dangx_index_plain_sum(ctxs[r], comp, q, s1, &si, &sm)) return sky_err(s, ctxs[r]);
            sums[0] += si; sums[1] += sm;
        }
        if (sky_ranks(s, sums, 2)) return 1;
        sample[q] = sums[0] / sums[1];
        if (tune(s, comp, nind, Sp, nsample, ml_mode, seed, stream ^ 0x5555555555555555ull, sample, &draw, tuned)) return 1;
    }
    if (step_size) *step_size = c0->desc[comp].step_size[nind];
    return 0;
}

// fit_band_gain(ddata, 1, band), src/dang_sample_mod.f90:570-621 (band 0-based): the two sky-wide sums, the draw (slot =
// band of the caller's stream), and ddata%gain(band) = gain on every context
int dangx_fit_band_gain(dangx_ctx* const* ctxs, int nctx, int band, int ml_mode, uint64_t seed, uint64_t stream, double* gain) {
    if (sky_check(ctxs, nctx)) return 1;
    const Sky s{ctxs, nctx};
    dangx_ctx* c0 = s.root();
    if (band < 0 || band >= c0->dims.nbands) return fail(c0, "band index out of range");
    if (ml_mode != DANGX_ML_SAMPLE && ml_mode != DANGX_ML_OPTIMIZE) return fail(c0, "bad ml_mode");
    double sums[2] = {0.0, 0.0};
    for (int r = 0; r < nctx; ++r) {
        double o[2];
        if (dangx_gain_sums(ctxs[r], band, o)) return sky_err(s, ctxs[r]);
        sums[0] += o[0]; sums[1] += o[1];
    }
    if (sky_ranks(s, sums, 2)) return 1;
    const double mu = sums[0] / sums[1];
    const double sigma = std::sqrt(1.0 / sums[1]);
    double g = mu;
    if (ml_mode == DANGX_ML_SAMPLE) {
        double u1, u2;
        h_uniform2(seed, stream, (uint32_t)band, u1, u2);
        g = mu + sigma * h_rand_normal(0.0, 1.0, u1, u2);
    }
    for (int r = 0; r < nctx; ++r) {  // ddata%gain(band) = gain (:619)
        ctxs[r]->hm.gain[band] = g;
        ctxs[r]->dirty = true;
        invalidate_chi(ctxs[r]);
    }
    if (gain) *gain = g;
    return 0;
}

// Which solves of sample_cg_groups may be issued TOGETHER with the first index sweep on their planes (dangx_amp_index_sample)
// without changing what the main loop computes.  The reference runs every solve (src/dang_cg_mod.f90:142-177) before any sweep
// of sample_spectral_parameters (src/dang_sample_mod.f90:21-86, component-major order).  Pulling a sweep forward to its group's
// solve is the same computation only when
//   * no OTHER (group, flag) solve works on any of these planes (it would otherwise see the swept index map),
//   * that sweep is the first one, in the reference's order, on these planes, it has exactly the solve's flag, and its
//     component is an amplitude-sampled member of the group,
//   * every sweep of the iteration is a plain per-pixel one (a full-sky chain, a coarse-Nside sweep or a pending step-size
//     tuning carry sky-wide state) and no 'T_cmb' component is sampled (its sweep changes the global T_CMB of every plane).
// Sweeps on disjoint planes are independent of each other (amplitudes, indices and data are all per plane), so their relative
// order is free.  first_sweep[p] = position in the sweep list, or -1 (make the two calls in the reference's order).
int dangx_plan_fusion(dangx_ctx* ctx, int npairs, const int32_t* pair_group, const int32_t* pair_flag, int nsweeps,
                      const int32_t* sweep_comp, const int32_t* sweep_nind, const int32_t* sweep_flag, const int32_t* sweep_plain,
                      int solver, int32_t* first_sweep) {
    if (!ctx || npairs < 0 || nsweeps < 0 || (npairs && (!pair_group || !pair_flag || !first_sweep)) ||
        (nsweeps && (!sweep_comp || !sweep_nind || !sweep_flag || !sweep_plain)))
        return 1;
    auto planes = [](int flag) -> unsigned {
        return flag == DANGX_FLAG_T ? 1u : flag == DANGX_FLAG_Q ? 2u : flag == DANGX_FLAG_U ? 4u : flag == DANGX_FLAG_QU ? 6u : 7u;
    };
    for (int p = 0; p < npairs; ++p) first_sweep[p] = -1;
    if (solver != DANGX_SOLVER_DIRECT) return 0;
    for (int e = 0; e < nsweeps; ++e) {
        const int l = sweep_comp[e];
        if (l < 0 || l >= ctx->dims.ncomp || !ctx->comp_set[l]) return fail(ctx, "sweep list: component index out of range / component not set");
        if (!sweep_plain[e] || ctx->desc[l].type == DANGX_TCMB) return 0;
    }
    for (int p = 0; p < npairs; ++p) {
        const unsigned pl = planes(pair_flag[p]);
        if (pl == 7u) continue;
        bool alone = true;
        for (int q = 0; q < npairs; ++q)
            if (q != p && (planes(pair_flag[q]) & pl)) alone = false;
        if (!alone) continue;
        for (int e = 0; e < nsweeps; ++e)
            if (planes(sweep_flag[e]) & pl) {  // the first sweep on these planes
                const dangx_comp_desc& d = ctx->desc[sweep_comp[e]];
                if (sweep_flag[e] == pair_flag[p] && d.cg_group == pair_group[p] && d.sample_amplitude) first_sweep[p] = e;
                break;
            }
    }
    return 0;
}

// "Update the global variable T_CMB" (src/dang_sample_mod.f90:75-78): T_CMB = c%indices(0, 1, 1) of a 'T_cmb' component
int dangx_update_tcmb(dangx_ctx* const* ctxs, int nctx, int comp, double* tcmb) {
    if (sky_check(ctxs, nctx)) return 1;
    const Sky s{ctxs, nctx};
    dangx_ctx* c0 = s.root();
    if (comp < 0 || comp >= c0->dims.ncomp || c0->desc[comp].type != DANGX_TCMB) return fail(c0, "not a T_cmb component");
    double first[2];
    if (first_pixel(s, comp, 1, first)) return 1;
    for (int r = 0; r < nctx; ++r)
        if (dangx_set_tcmb(ctxs[r], first[0])) return sky_err(s, ctxs[r]);
    if (tcmb) *tcmb = first[0];
    return 0;
}

}  // extern "C"
