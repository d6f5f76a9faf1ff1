/*
 * dang_oracle.c -- CPU restatement of the dang Gibbs inner loop.
 * TEST INFRASTRUCTURE ONLY -- see dang_oracle.h.  PARITY UNPINNED (see header).
 *
 * Every routine restates the cited reference lines (paths relative to the
 * reference checkout, e.g. src/dang_cg_mod.f90:598-911) with the same loop
 * structure and the same order of floating-point operations, so that it can
 * stand in for the reference's CPU path both as the checker and as the
 * `cpu_baseline` ("port") timing.  Global-amplitude component types
 * (template / monopole / hi_fit) are not restated yet.
 */
#include "dang_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants: src/dang_util_mod.f90:12-15,19; pi: HEALPix healpix_types ---- */
static const double PI_ = 3.141592653589793238462643383279502884197;
static const double K_B = 1.3806503e-23;
static const double C_LIGHT = 2.99792458e8;
static const double MISSVAL = -1.6375e30;
static double planck_h(void) { return 1.0545726691251021e-34 * 2.0 * PI_; }

double dgo_const_h(void) { return planck_h(); }
double dgo_const_kB(void) { return K_B; }
double dgo_const_c(void) { return C_LIGHT; }
double dgo_missval(void) { return MISSVAL; }

static int masked(double m) { return m == 0.0 || m == MISSVAL; }

static void set_threads(const dgo_ctx *ctx) {
#ifdef _OPENMP
    if (ctx->nthreads > 0) omp_set_num_threads(ctx->nthreads);
#else
    (void)ctx;
#endif
}

/* ------------------------------------------------------------------ RNG */

/* Philox4x32-10 (Salmon et al. 2011, Random123).  Builder-defined stream:
 * the reference calls the compiler's RANDOM_NUMBER after an unseeded
 * RANDOM_SEED() (src/dang.f90:67), which is irreproducible by construction. */
void dgo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u53(uint32_t hi, uint32_t lo) {
    uint64_t x = ((uint64_t)hi << 32) | lo;
    return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0); /* (0,1) */
}

void dgo_uniform2(uint64_t seed, uint64_t stream, uint64_t pix, uint32_t draw, double u[2]) {
    uint32_t ctr[4] = {(uint32_t)pix, draw, (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    /* pixel indices above 2^32 fold their high word into the draw slot */
    ctr[1] ^= (uint32_t)(pix >> 32) << 16;
    dgo_philox4x32_10(ctr, key, o);
    u[0] = u53(o[0], o[1]);
    u[1] = u53(o[2], o[3]);
}

/* one Philox call per Metropolis step: u[0] 53 bits (words 0,1), u[1] and u[2] 32 bits (words 2, 3) */
void dgo_uniform3(uint64_t seed, uint64_t stream, uint64_t pix, uint32_t draw, double u[3]) {
    uint32_t ctr[4] = {(uint32_t)pix, draw, (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t o[4];
    ctr[1] ^= (uint32_t)(pix >> 32) << 16;
    dgo_philox4x32_10(ctr, key, o);
    u[0] = u53(o[0], o[1]);
    u[1] = ((double)o[2] + 0.5) * (1.0 / 4294967296.0);
    u[2] = ((double)o[3] + 0.5) * (1.0 / 4294967296.0);
}

/* src/dang_util_mod.f90:100-110: Box-Muller, sine branch only */
double dgo_rand_normal(double mean, double stdev, double u1, double u2) {
    double r = sqrt(-2.0 * log(u1));
    double theta = 2.0 * PI_ * u2;
    return mean + stdev * r * sin(theta);
}

/* src/dang_util_mod.f90:112-121 */
double dgo_eval_normal_prior(double prop, double mean, double std) {
    double var = std * std;
    double num = exp(-((prop - mean) * (prop - mean)) / (2 * var));
    double denom = std * sqrt(2.0 * PI_);
    return num / denom;
}

/* ------------------------------------------------------------------ sky model */

#define IDX3(ctx, j, k, i) (((int64_t)(j) * (ctx)->nmaps + ((k)-1)) * (int64_t)(ctx)->npix + (i))
#define IDX2(ctx, k, i) (((int64_t)((k)-1)) * (int64_t)(ctx)->npix + (i))

/* src/dang_bp_mod.f90:211-243 */
double dgo_a2t(const dgo_ctx *ctx, int band) {
    const dgo_band *b = &ctx->bands[band];
    const double h = planck_h();
    double sum = 0.0, y;
    if (b->n == 0) {
        if (b->nu_c > 1e7f) y = (h * b->nu_c) / (K_B * ctx->T_CMB);
        else y = (h * b->nu_c * 1e9) / (K_B * ctx->T_CMB);
        sum = ((exp(y) - 1.0) * (exp(y) - 1.0)) / ((y * y) * exp(y));
    } else {
        for (int i = 0; i < b->n; ++i) {
            if (b->nu0[i] == 0.0) continue;
            if (b->nu0[i] > 1e7f) y = (h * b->nu0[i]) / (K_B * ctx->T_CMB);
            else y = (h * b->nu0[i] * 1e9) / (K_B * ctx->T_CMB);
            sum = sum + b->tau0[i] * ((exp(y) - 1.0) * (exp(y) - 1.0)) / ((y * y) * exp(y));
        }
    }
    return sum;
}

/* src/dang_bp_mod.f90:160-179 (nu in Hz); double defined below with B_nu */
double dgo_bnu_prime_RJ(double nu);
double dgo_bnu_prime(double nu, double T_CMB) {
    const double h = planck_h();
    double y = h * nu / (K_B * T_CMB);
    return (2.0 * h * (nu * nu * nu)) / (pow(C_LIGHT, 2.0) * (exp(y) - 1)) * (exp(y) / (exp(y) - 1)) * h * nu / (K_B * (T_CMB * T_CMB));
}

/* a2f, src/dang_bp_mod.f90:181-209.  `a2f = sum*1e14`: the literal is SINGLE precision (1e14 -> 100000000376832) */
double dgo_a2f(const dgo_ctx *ctx, int band) {
    const dgo_band *b = &ctx->bands[band];
    double sum = 0.0;
    if (b->n == 0) {
        if (b->nu_c > 1e7f) sum = dgo_bnu_prime_RJ(b->nu_c);
        else sum = dgo_bnu_prime_RJ(b->nu_c * 1e9);
    } else {
        for (int i = 0; i < b->n; ++i) {
            if (b->nu0[i] == 0.0) continue;
            if (b->nu0[i] > 1e7f) sum = sum + b->tau0[i] * dgo_bnu_prime_RJ(b->nu0[i]);
            else sum = sum + b->tau0[i] * dgo_bnu_prime_RJ(b->nu0[i] * 1e9);
        }
    }
    return sum * 1e14f;
}

/* f2t, src/dang_bp_mod.f90:245-274 */
double dgo_f2t(const dgo_ctx *ctx, int band) {
    const dgo_band *b = &ctx->bands[band];
    double sum = 0.0;
    if (b->n == 0) {
        if (b->nu_c > 1e7f) sum = 1.0 / (dgo_bnu_prime(b->nu_c, ctx->T_CMB)) * 1.0e-14;
        else sum = 1.0 / (dgo_bnu_prime(b->nu_c * 1e9, ctx->T_CMB)) * 1.0e-14;
    } else {
        for (int i = 0; i < b->n; ++i) {
            if (b->nu0[i] == 0.0) continue;
            if (b->nu0[i] > 1e7f) sum = sum + b->tau0[i] / (dgo_bnu_prime(b->nu0[i], ctx->T_CMB)) * 1.0e-14;
            else sum = sum + b->tau0[i] / (dgo_bnu_prime(b->nu0[i] * 1e9, ctx->T_CMB)) * 1.0e-14;
        }
    }
    return sum;
}

/* normalize_bandpass, src/dang_bp_mod.f90:62-81 */
void dgo_normalize_bandpass(const double *tau_in, int n, double *tau_out) {
    double total = 0.0;
    for (int i = 0; i < n; ++i) total += tau_in[i];
    for (int i = 0; i < n; ++i) tau_out[i] = tau_in[i] / total;
}

/* convert_maps, src/dang_data_mod.f90:429-463: unit 0 uK_RJ, 1 uK_cmb, 2 MJy/sr; cg_map may be NULL.  Scales ctx->sig,
 * ctx->rms, ctx->offset in place and copies the offsets into template_amplitudes(:,1) of every monopole after each band. */
int dgo_convert_maps(dgo_ctx *ctx, const int *unit, const int *cg_map, double *conversion) {
    for (int j = 0; j < ctx->nbands; ++j) {
        if (cg_map && cg_map[j]) continue;
        if (unit[j] == 0) conversion[j] = 1.0;
        else if (unit[j] == 1) conversion[j] = 1.0 / dgo_a2t(ctx, j);
        else if (unit[j] == 2) conversion[j] = 1.0 / dgo_a2f(ctx, j);
        else return 1; /* "Not a unit" -> stop */
        double *sig = (double *)ctx->sig, *rms = (double *)ctx->rms; /* the maps are the caller's writable arrays */
        for (int k = 1; k <= ctx->nmaps; ++k)
            for (int i = 0; i < ctx->npix; ++i) {
                sig[IDX3(ctx, j, k, i)] = sig[IDX3(ctx, j, k, i)] * conversion[j];
                rms[IDX3(ctx, j, k, i)] = rms[IDX3(ctx, j, k, i)] * conversion[j];
            }
        ctx->offset[j] = ctx->offset[j] * conversion[j];
        for (int l = 0; l < ctx->ncomp; ++l)
            if (ctx->comps[l].type == DGO_MONOPOLE)
                for (int jj = 0; jj < ctx->nbands; ++jj) ctx->comps[l].template_amplitudes[jj] = ctx->offset[jj];
    }
    return 0;
}

static void get_theta(const dgo_ctx *ctx, const dgo_comp *c, int pix, int map_n, const double *theta, double th[DGO_MAX_IND]) {
    for (int l = 0; l < c->nindices && l < DGO_MAX_IND; ++l)
        th[l] = theta ? theta[l] : c->indices[((int64_t)l * ctx->nmaps + (map_n - 1)) * (int64_t)ctx->npix + pix];
}

/* src/dang_component_mod.f90:886-918 */
static double sed_powerlaw(const dgo_band *b, double nu_ref, double beta) {
    if (b->n == 0) return pow(b->nu_c / nu_ref, beta);
    double s = 0.0;
    for (int i = 0; i < b->n; ++i) {
        if (b->nu0[i] == 0.0) continue;
        s = s + b->tau0[i] * pow(b->nu0[i] / nu_ref, beta);
    }
    return s;
}

/* src/dang_component_mod.f90:920-958 */
static double sed_mbb(const dgo_band *b, double nu_ref, double beta, double td) {
    const double z = planck_h() / (K_B * td);
    if (b->n == 0)
        return (exp(z * nu_ref) - 1.0) / (exp(z * b->nu_c) - 1.0) * pow(b->nu_c / nu_ref, beta + 1.0);
    double s = 0.0;
    for (int i = 0; i < b->n; ++i) {
        if (b->nu0[i] == 0.0) continue;
        s = s + b->tau0[i] * (exp(z * nu_ref) - 1.0) / (exp(z * b->nu0[i]) - 1.0) * pow(b->nu0[i] / nu_ref, beta + 1.0);
    }
    return s;
}

/* src/dang_component_mod.f90:960-999 */
static double sed_lognormal(const dgo_band *b, double nu_ref, double nu_p, double w) {
    if (b->n == 0) {
        double l = log(b->nu_c / (nu_p * 1e9)) / w;
        double q = nu_ref / b->nu_c;
        return exp(-0.5 * (l * l)) * (q * q);
    }
    double s = 0.0;
    for (int i = 0; i < b->n; ++i) {
        if (b->nu0[i] == 0.0) continue;
        double l = log(b->nu0[i] / (nu_p * 1e9)) / w;
        double q = nu_ref / b->nu0[i];
        s = s + b->tau0[i] * exp(-0.5 * (l * l)) * (q * q);
    }
    return s;
}

/* src/dang_component_mod.f90:1001-1040 */
static double ff_gaunt(double nu, double T_e) {
    return log(exp(5.960 - sqrt(3.0) / PI_ * log(1.0 * nu / 1.0e9 * pow(T_e / 1.0e4, -1.5))) + 2.71828);
}
static double sed_freefree(const dgo_band *b, double nu_ref, double T_e) {
    const double S_ref = ff_gaunt(nu_ref, T_e);
    if (b->n == 0) {
        double r = b->nu_c / nu_ref;
        return ff_gaunt(b->nu_c, T_e) / S_ref * (1.0 / (r * r));
    }
    double s = 0.0;
    for (int i = 0; i < b->n; ++i) {
        if (b->nu0[i] == 0.0) continue;
        double r = b->nu0[i] / nu_ref;
        s = s + b->tau0[i] * ff_gaunt(b->nu0[i], T_e) / S_ref * (1.0 / (r * r));
    }
    return s;
}

/* src/dang_component_mod.f90:745-752 */
double dgo_B_nu(double nu, double T) {
    const double h = planck_h();
    return ((2.0 * h * pow(nu, 3.0)) / pow(C_LIGHT, 2.0)) * (1.0 / (exp((h * nu) / (K_B * T)) - 1));
}
/* src/dang_bp_mod.f90:160-168 */
double dgo_bnu_prime_RJ(double nu) { return 2.0 * K_B * pow(nu, 2.0) / pow(C_LIGHT, 2.0); }
/* evaluate_T_cmb / evaluate_hi_fit, src/dang_component_mod.f90:815-884 (same expression; hi_fit multiplies by its template) */
static double sed_planck_rj(const dgo_band *b, double T) {
    double s = 0.0;
    if (b->n == 0) s = dgo_B_nu(b->nu_c, T) / dgo_bnu_prime_RJ(b->nu_c);
    else
        for (int i = 0; i < b->n; ++i) {
            if (b->nu0[i] == 0.0) continue;
            s = s + b->tau0[i] * dgo_B_nu(b->nu0[i], T) / dgo_bnu_prime_RJ(b->nu0[i]);
        }
    return s * 1e6f;
}

/* src/dang_component_mod.f90:778-813 */
double dgo_eval_sed(const dgo_ctx *ctx, int comp, int band, int pix, int map_n, const double *theta) {
    const dgo_comp *c = &ctx->comps[comp];
    const dgo_band *b = &ctx->bands[band];
    double th[DGO_MAX_IND] = {0.0, 0.0};
    get_theta(ctx, c, pix, map_n, theta, th);
    switch (c->type) {
    case DGO_POWERLAW: return sed_powerlaw(b, c->nu_ref, th[0]);
    case DGO_MBB: return sed_mbb(b, c->nu_ref, th[0], th[1]);
    case DGO_FREEFREE: return sed_freefree(b, c->nu_ref, th[0]);
    case DGO_LOGNORMAL: return sed_lognormal(b, c->nu_ref, th[0], th[1]);
    case DGO_CMB: return 1.0 / dgo_a2t(ctx, band);
    case DGO_TCMB: return sed_planck_rj(b, th[0]);
    case DGO_TEMPLATE: /* :803-806 */
    case DGO_MONOPOLE: return c->tmpl[IDX2(ctx, map_n, pix)];
    case DGO_HIFIT: return c->tmpl[IDX2(ctx, map_n, pix)] * sed_planck_rj(b, th[0]); /* :807-808 */
    default: return NAN;
    }
}

/* src/dang_component_mod.f90:754-776 (diffuse branch, :773) */
double dgo_eval_signal(const dgo_ctx *ctx, int comp, int band, int pix, int map_n, const double *theta) {
    const dgo_comp *c = &ctx->comps[comp];
    if (c->type == DGO_TCMB) return dgo_eval_sed(ctx, comp, band, pix, map_n, theta); /* :770-771 */
    if (c->type == DGO_HIFIT) /* :764-765 */
        return c->template_amplitudes[(map_n - 1) * ctx->nbands + band] * dgo_eval_sed(ctx, comp, band, pix, map_n, theta);
    if (c->type == DGO_TEMPLATE || c->type == DGO_MONOPOLE) /* :766-769 */
        return c->template_amplitudes[(map_n - 1) * ctx->nbands + band] * c->tmpl[IDX2(ctx, map_n, pix)];
    return c->amplitude[IDX2(ctx, map_n, pix)] * dgo_eval_sed(ctx, comp, band, pix, map_n, theta);
}

/* ------------------------------------------------------------------ amplitude phase */

static int flag_nplanes(int flag) { return (flag & DGO_FLAG_QU) ? 2 : 1; }
/* map number of plane p (0/1) for a flag: src/dang_cg_mod.f90:357-363, 488-508 */
static int flag_map(int flag, int p) {
    if (flag & DGO_FLAG_QU) return 2 + p;
    if (flag & DGO_FLAG_T) return 1;
    if (flag & DGO_FLAG_Q) return 2;
    return 3;
}
static int in_group(const dgo_comp *c, int group) { return c->cg_group == group && c->sample_amplitude; }
static int is_global(const dgo_comp *c) { return c->type == DGO_TEMPLATE || c->type == DGO_MONOPOLE || c->type == DGO_HIFIT; }
/* plane(s) a global component's row sums run over: hi_fit and monopole use plane 1 only (:531, :550, :840, :857);
 * a template uses both planes under Q+U (:571-572) or map_n */
static int glob_nplanes(const dgo_comp *c, int flag) { return (c->type == DGO_TEMPLATE && (flag & DGO_FLAG_QU)) ? 2 : 1; }
static int glob_map(const dgo_comp *c, int flag, int p) { return (c->type == DGO_TEMPLATE) ? flag_map(flag, p) : 1; }
/* slot of the noise vector temp1 that plane p of a global component reads/writes (:723, :737, :752-759) */
static int64_t glob_slot(const dgo_ctx *ctx, const dgo_comp *c, int flag, int p, int i) {
    (void)c; (void)flag;
    return (int64_t)p * ctx->npix + i;
}

int64_t dgo_group_size(const dgo_ctx *ctx, int group, int flag, int *ncg) {
    int n = 0;
    int64_t nglob = 0;
    for (int l = 0; l < ctx->ncomp; ++l)
        if (in_group(&ctx->comps[l], group)) {
            if (is_global(&ctx->comps[l])) nglob += ctx->comps[l].nfit; /* :401-410 (the reference omits a template's
                                                                           nfit for single-plane flags: out of bounds there) */
            else ++n;
        }
    if (ncg) *ncg = n;
    return (int64_t)n * flag_nplanes(flag) * ctx->npix + nglob;
}

/* src/dang_cg_mod.f90:326-596 */
void dgo_compute_rhs(const dgo_ctx *ctx, int group, int flag, double *b) {
    const int npix = ctx->npix, nb = ctx->nbands, nmaps = ctx->nmaps;
    const int S = flag_nplanes(flag);
    set_threads(ctx);
    double *data = (double *)malloc(sizeof(double) * (size_t)nb * nmaps * npix);
    /* :367-378  T is divided by the gain, the offset is NOT removed here */
    for (int k = 1; k <= nmaps; ++k)
        for (int j = 0; j < nb; ++j)
            for (int i = 0; i < npix; ++i)
                data[IDX3(ctx, j, k, i)] = (k == 1) ? ctx->sig[IDX3(ctx, j, k, i)] / ctx->gain[j] : ctx->sig[IDX3(ctx, j, k, i)];
    for (int l = 0; l < ctx->ncomp; ++l) {
        const dgo_comp *c = &ctx->comps[l];
        /* :427-443 remove components that are not solved for in this group */
        if (!in_group(c, group)) {
#pragma omp parallel for schedule(static)
            for (int i = 0; i < npix; ++i) {
                if (masked(ctx->mask[i])) continue;
                for (int k = 1; k <= nmaps; ++k)
                    for (int j = 0; j < nb; ++j)
                        data[IDX3(ctx, j, k, i)] = data[IDX3(ctx, j, k, i)] - dgo_eval_signal(ctx, l, j, i, k, NULL);
            }
        }
        /* :445-460 "still subtract templates which exist but may not be fit here": every template / monopole
         * (whatever its group -- a non-member is thereby removed TWICE on these bands), bands with corr == false */
        if (c->type == DGO_TEMPLATE || c->type == DGO_MONOPOLE)
            for (int j = 0; j < nb; ++j) {
                if (c->corr[j]) continue;
                for (int i = 0; i < npix; ++i) {
                    if (masked(ctx->mask[i])) continue;
                    for (int k = 1; k <= nmaps; ++k)
                        data[IDX3(ctx, j, k, i)] = data[IDX3(ctx, j, k, i)] - dgo_eval_signal(ctx, l, j, i, k, NULL);
                }
            }
    }
    int64_t n = dgo_group_size(ctx, group, flag, NULL);
    for (int64_t q = 0; q < n; ++q) b[q] = 0.0;
    /* :464-587 */
    int64_t offset = 0;
    for (int l = 0; l < ctx->ncomp; ++l) {
        const dgo_comp *c = &ctx->comps[l];
        if (!in_group(c, group)) continue;
        if (!is_global(c)) {
#pragma omp parallel for schedule(static)
            for (int i = 0; i < npix; ++i) {
                for (int j = 0; j < nb; ++j) {
                    if (ctx->mask[i] == 0.0) { /* :474 -- only the ==0 test here */
                        for (int p = 0; p < S; ++p) b[(int64_t)p * npix + i] = 0.0;
                        continue;
                    }
                    for (int p = 0; p < S; ++p) {
                        int k = flag_map(flag, p);
                        double rms = ctx->rms[IDX3(ctx, j, k, i)];
                        b[offset + (int64_t)p * npix + i] = b[offset + (int64_t)p * npix + i] +
                            (data[IDX3(ctx, j, k, i)] * dgo_eval_sed(ctx, l, j, i, k, NULL)) / (rms * rms);
                    }
                }
            }
            offset += (int64_t)S * npix;
        } else { /* :522-587: one entry per fitted band, b(offset+l) += sum(val_array) */
            int lfit = 0;
            for (int j = 0; j < nb; ++j) {
                if (!c->corr[j]) continue;
                double sum = 0.0;
                for (int p = 0; p < glob_nplanes(c, flag); ++p) {
                    const int k = glob_map(c, flag, p);
                    for (int i = 0; i < npix; ++i) {
                        if (masked(ctx->mask[i])) continue;
                        double rms = ctx->rms[IDX3(ctx, j, k, i)];
                        sum += data[IDX3(ctx, j, k, i)] / (rms * rms) * dgo_eval_sed(ctx, l, j, i, k, NULL);
                    }
                }
                b[offset + lfit] = b[offset + lfit] + sum;
                lfit++;
            }
            offset += c->nfit;
        }
    }
    free(data);
}

/* src/dang_cg_mod.f90:598-911 */
void dgo_compute_Ax(const dgo_ctx *ctx, int group, int flag, const double *x, double *res) {
    const int npix = ctx->npix, nb = ctx->nbands;
    const int S = flag_nplanes(flag);
    const int64_t m = (int64_t)S * npix;
    const int64_t n = dgo_group_size(ctx, group, flag, NULL);
    set_threads(ctx);
    double *temp1 = (double *)malloc(sizeof(double) * (size_t)m);
    double *temp3 = (double *)malloc(sizeof(double) * (size_t)n);
    int lcount[64]; /* l(:) = 1, one counter per template-like component (:654) */
    for (int t = 0; t < 64; ++t) lcount[t] = 0;
    for (int64_t q = 0; q < n; ++q) res[q] = 0.0;
    for (int j = 0; j < nb; ++j) {
        for (int64_t q = 0; q < m; ++q) temp1[q] = 0.0;
        for (int64_t q = 0; q < n; ++q) temp3[q] = 0.0;
        int64_t offset = 0;
        int l_ind = 0;
        /* :685-769 temp1 = T_nu x */
        for (int l = 0; l < ctx->ncomp; ++l) {
            const dgo_comp *c = &ctx->comps[l];
            if (!in_group(c, group)) continue;
            if (!is_global(c)) {
#pragma omp parallel for schedule(static)
                for (int i = 0; i < npix; ++i) {
                    if (masked(ctx->mask[i])) continue;
                    for (int p = 0; p < S; ++p)
                        temp1[(int64_t)p * npix + i] = temp1[(int64_t)p * npix + i] +
                            x[offset + (int64_t)p * npix + i] * dgo_eval_sed(ctx, l, j, i, flag_map(flag, p), NULL);
                }
                offset += m;
            } else {
                if (c->corr[j])
                    for (int p = 0; p < glob_nplanes(c, flag); ++p)
                        for (int i = 0; i < npix; ++i) {
                            if (masked(ctx->mask[i])) continue;
                            temp1[glob_slot(ctx, c, flag, p, i)] = temp1[glob_slot(ctx, c, flag, p, i)] +
                                x[offset + lcount[l_ind]] * dgo_eval_sed(ctx, l, j, i, glob_map(c, flag, p), NULL);
                        }
                l_ind++;
                offset += c->nfit;
            }
        }
        /* :775-791 temp1 = N^-1 temp1 */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < npix; ++i) {
            if (masked(ctx->mask[i])) continue;
            for (int p = 0; p < S; ++p) {
                double rms = ctx->rms[IDX3(ctx, j, flag_map(flag, p), i)];
                temp1[(int64_t)p * npix + i] = temp1[(int64_t)p * npix + i] / (rms * rms);
            }
        }
        /* :801-894 temp3 = T_nu^t temp1 */
        offset = 0;
        l_ind = 0;
        for (int l = 0; l < ctx->ncomp; ++l) {
            const dgo_comp *c = &ctx->comps[l];
            if (!in_group(c, group)) continue;
            if (!is_global(c)) {
#pragma omp parallel for schedule(static)
                for (int i = 0; i < npix; ++i) {
                    if (masked(ctx->mask[i])) continue;
                    for (int p = 0; p < S; ++p)
                        temp3[offset + (int64_t)p * npix + i] =
                            temp1[(int64_t)p * npix + i] * dgo_eval_sed(ctx, l, j, i, flag_map(flag, p), NULL);
                }
                offset += m;
            } else {
                if (c->corr[j]) {
                    double sum = 0.0;
                    for (int p = 0; p < glob_nplanes(c, flag); ++p)
                        for (int i = 0; i < npix; ++i) {
                            if (masked(ctx->mask[i])) continue;
                            /* :857 the monopole row sums temp1(i) WITHOUT its template factor */
                            const double w = (c->type == DGO_MONOPOLE) ? 1.0 : dgo_eval_sed(ctx, l, j, i, glob_map(c, flag, p), NULL);
                            sum += temp1[glob_slot(ctx, c, flag, p, i)] * w;
                        }
                    temp3[offset + lcount[l_ind]] = temp3[offset + lcount[l_ind]] + sum;
                    lcount[l_ind] = lcount[l_ind] + 1;
                }
                l_ind++;
                offset += c->nfit;
            }
        }
        for (int64_t q = 0; q < n; ++q) res[q] = res[q] + temp3[q]; /* :904 */
    }
    free(temp1);
    free(temp3);
}

/* src/dang_cg_mod.f90:913-1100.  NOTE (quirks 2,3): the diffuse branch writes
 * temp2(i)/temp2(npix+i) WITHOUT the component offset and with '=' (:1033-1040),
 * and the same eta is used for every band (:1008-1015).  The global rows use ONE offset (the total diffuse
 * length, :975-990) and ONE running counter l over bands and components (:970, :1057, :1071, :1094). */
void dgo_compute_sample_vector(const dgo_ctx *ctx, int group, int flag, const double *eta, double *res) {
    const int npix = ctx->npix, nb = ctx->nbands;
    const int S = flag_nplanes(flag);
    const int64_t m = (int64_t)S * npix;
    const int64_t n = dgo_group_size(ctx, group, flag, NULL);
    set_threads(ctx);
    double *temp1 = (double *)malloc(sizeof(double) * (size_t)m);
    double *temp2 = (double *)malloc(sizeof(double) * (size_t)n);
    int64_t offset = 0;
    for (int l = 0; l < ctx->ncomp; ++l)
        if (in_group(&ctx->comps[l], group) && !is_global(&ctx->comps[l])) offset += m;
    int lrun = 0;
    for (int64_t q = 0; q < n; ++q) res[q] = 0.0;
    for (int j = 0; j < nb; ++j) {
        for (int64_t q = 0; q < m; ++q) temp1[q] = 0.0;
        for (int64_t q = 0; q < n; ++q) temp2[q] = 0.0;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < npix; ++i) {
            if (masked(ctx->mask[i])) continue;
            for (int p = 0; p < S; ++p)
                temp1[(int64_t)p * npix + i] = eta[(int64_t)p * npix + i] / ctx->rms[IDX3(ctx, j, flag_map(flag, p), i)];
        }
        for (int l = 0; l < ctx->ncomp; ++l) {
            const dgo_comp *c = &ctx->comps[l];
            if (!in_group(c, group)) continue;
            if (!is_global(c)) {
#pragma omp parallel for schedule(static)
                for (int i = 0; i < npix; ++i) {
                    if (masked(ctx->mask[i])) continue;
                    for (int p = 0; p < S; ++p)
                        temp2[(int64_t)p * npix + i] =
                            temp1[(int64_t)p * npix + i] * dgo_eval_sed(ctx, l, j, i, flag_map(flag, p), NULL);
                }
            } else if (c->corr[j]) {
                double sum = 0.0;
                for (int p = 0; p < glob_nplanes(c, flag); ++p)
                    for (int i = 0; i < npix; ++i) {
                        if (masked(ctx->mask[i])) continue;
                        const double w = (c->type == DGO_MONOPOLE) ? 1.0 : dgo_eval_sed(ctx, l, j, i, glob_map(c, flag, p), NULL);
                        sum += temp1[glob_slot(ctx, c, flag, p, i)] * w;
                    }
                if (offset + lrun < n) temp2[offset + lrun] = temp2[offset + lrun] + sum; /* (the reference would run out of bounds) */
                lrun++;
            }
        }
        for (int64_t q = 0; q < n; ++q) res[q] = res[q] + temp2[q];
    }
    free(temp1);
    free(temp2);
}

/* src/dang_cg_mod.f90:1173-1282 */
void dgo_initialize_x(const dgo_ctx *ctx, int group, int flag, double *x) {
    const int S = flag_nplanes(flag);
    int64_t offset = 0;
    for (int l = 0; l < ctx->ncomp; ++l) {
        const dgo_comp *c = &ctx->comps[l];
        if (!in_group(c, group)) continue;
        if (!is_global(c)) {
            for (int p = 0; p < S; ++p) {
                for (int i = 0; i < ctx->npix; ++i) x[offset + i] = c->amplitude[IDX2(ctx, flag_map(flag, p), i)];
                offset += ctx->npix;
            }
        } else { /* :1244-1279: template_amplitudes(j, map_n); a template under Q+U reads plane 2 (:1270) */
            const int k = (c->type == DGO_TEMPLATE) ? flag_map(flag, 0) : flag_map(flag, 0);
            int lfit = 0;
            for (int j = 0; j < ctx->nbands && lfit < c->nfit; ++j)
                if (c->corr[j]) x[offset + lfit++] = c->template_amplitudes[(k - 1) * ctx->nbands + j];
            offset += lfit;
        }
    }
}

/* src/dang_cg_mod.f90:1284-1396 */
void dgo_unpack_amplitudes(dgo_ctx *ctx, int group, int flag, const double *x) {
    const int S = flag_nplanes(flag);
    int64_t offset = 0;
    for (int l = 0; l < ctx->ncomp; ++l) {
        dgo_comp *c = &ctx->comps[l];
        if (!in_group(c, group)) continue;
        if (!is_global(c)) {
            for (int p = 0; p < S; ++p) {
                for (int i = 0; i < ctx->npix; ++i) c->amplitude[IDX2(ctx, flag_map(flag, p), i)] = x[offset + i];
                offset += ctx->npix;
            }
        } else {
            int lfit = 0;
            for (int j = 0; j < ctx->nbands && lfit < c->nfit; ++j)
                if (c->corr[j]) {
                    /* :1380-1382 a template under Q+U shares ONE amplitude between Q and U */
                    if (c->type == DGO_TEMPLATE && (flag & DGO_FLAG_QU)) {
                        c->template_amplitudes[1 * ctx->nbands + j] = x[offset + lfit];
                        c->template_amplitudes[2 * ctx->nbands + j] = x[offset + lfit];
                    } else {
                        c->template_amplitudes[(flag_map(flag, 0) - 1) * ctx->nbands + j] = x[offset + lfit];
                    }
                    lfit++;
                }
            offset += lfit;
        }
    }
}

/* src/dang_cg_mod.f90:254-262: eta(i)=rand_normal(0,1), i=1..m.  The keyed
 * stream uses (pixel, draw = map number) so that eta does not depend on sharding. */
void dgo_draw_eta(const dgo_ctx *ctx, int flag, uint64_t seed, uint64_t stream, double *eta) {
    const int S = flag_nplanes(flag);
    for (int p = 0; p < S; ++p)
        for (int i = 0; i < ctx->npix; ++i) {
            double u[2];
            dgo_uniform2(seed, stream, (uint64_t)(ctx->pix0 + i), (uint32_t)flag_map(flag, p), u);
            eta[(int64_t)p * ctx->npix + i] = dgo_rand_normal(0.0, 1.0, u[0], u[1]);
        }
}

static double dot(const double *a, const double *b, int64_t n) {
    double s = 0.0;
    for (int64_t q = 0; q < n; ++q) s += a[q] * b[q];
    return s;
}

/* src/dang_cg_mod.f90:179-324 (Shewchuk B2, no preconditioner) */
int dgo_cg_search(const dgo_ctx *ctx, int group, int flag, const double *b, int ml_mode, const double *eta,
                  double *x, int i_max, double converge, double *delta_trace) {
    const int64_t n = dgo_group_size(ctx, group, flag, NULL);
    double *b2 = (double *)malloc(sizeof(double) * (size_t)n);
    double *r = (double *)malloc(sizeof(double) * (size_t)n);
    double *d = (double *)malloc(sizeof(double) * (size_t)n);
    double *q = (double *)malloc(sizeof(double) * (size_t)n);
    if (ml_mode == DGO_ML_SAMPLE) { /* :254-264 */
        dgo_compute_sample_vector(ctx, group, flag, eta, q);
        for (int64_t t = 0; t < n; ++t) b2[t] = b[t] + q[t];
    } else {
        for (int64_t t = 0; t < n; ++t) b2[t] = b[t];
    }
    dgo_compute_Ax(ctx, group, flag, x, q); /* :283 */
    for (int64_t t = 0; t < n; ++t) { r[t] = b2[t] - q[t]; d[t] = r[t]; }
    double delta_new = dot(r, r, n), delta_old;
    int i = 1;
    if (delta_trace) delta_trace[0] = delta_new;
    while (i < i_max && delta_new > converge) { /* :293 */
        dgo_compute_Ax(ctx, group, flag, d, q);
        double alpha = delta_new / dot(d, q, n);
        for (int64_t t = 0; t < n; ++t) x[t] = x[t] + alpha * d[t];
        for (int64_t t = 0; t < n; ++t) r[t] = r[t] - alpha * q[t];
        delta_old = delta_new;
        delta_new = dot(r, r, n);
        double beta = delta_new / delta_old;
        for (int64_t t = 0; t < n; ++t) d[t] = r[t] + beta * d[t];
        if (delta_trace) delta_trace[i] = delta_new;
        i = i + 1;
    }
    free(b2); free(r); free(d); free(q);
    return i;
}

/* one (group, flag) pass of sample_cg_groups, src/dang_cg_mod.f90:166-171 */
int dgo_amp_sample_cg(dgo_ctx *ctx, int group, int flag, int ml_mode, uint64_t seed, uint64_t stream,
                      int i_max, double converge, double *x_state) {
    const int64_t n = dgo_group_size(ctx, group, flag, NULL);
    const int64_t m = (int64_t)flag_nplanes(flag) * ctx->npix;
    double *b = (double *)malloc(sizeof(double) * (size_t)n);
    double *eta = (double *)calloc((size_t)m, sizeof(double));
    double *x = x_state ? x_state : (double *)malloc(sizeof(double) * (size_t)n);
    dgo_compute_rhs(ctx, group, flag, b);
    if (!x_state) dgo_initialize_x(ctx, group, flag, x); /* iter==1, :227-239 */
    if (ml_mode == DGO_ML_SAMPLE) dgo_draw_eta(ctx, flag, seed, stream, eta);
    int it = dgo_cg_search(ctx, group, flag, b, ml_mode, eta, x, i_max, converge, NULL);
    dgo_unpack_amplitudes(ctx, group, flag, x);
    free(b); free(eta);
    if (!x_state) free(x);
    return it;
}

/* Direct block solve: the system of compute_rhs/compute_Ax/compute_sample_vector is
 * block diagonal with one ncg x ncg SPD block per (pixel, plane) when all group
 * components are diffuse (every term of :697-704, :813-820 couples only index i).
 * This restates the arithmetic order of the HIP kernel, so GPU-vs-oracle agreement
 * is at rounding level; agreement with dgo_amp_sample_cg is at CG-residual level. */
#define DGO_MAX_NC 8
int dgo_amp_sample_direct(dgo_ctx *ctx, int group, int flag, int ml_mode, int fluct_mode, uint64_t seed,
                          uint64_t stream, int64_t *n_not_spd) {
    const int npix = ctx->npix, nb = ctx->nbands, S = flag_nplanes(flag);
    int gc[DGO_MAX_NC], oc[64], ng = 0, no = 0;
    for (int l = 0; l < ctx->ncomp; ++l) {
        if (in_group(&ctx->comps[l], group) && is_global(&ctx->comps[l])) return -2; /* coupled system: CG only */
        if (in_group(&ctx->comps[l], group)) { if (ng >= DGO_MAX_NC) return -1; gc[ng++] = l; }
        else { if (no >= 64) return -1; oc[no++] = l; }
    }
    if (ng == 0) return -1;
    int64_t bad = 0;
    set_threads(ctx);
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (int i = 0; i < npix; ++i) {
        if (masked(ctx->mask[i])) continue;
        for (int p = 0; p < S; ++p) {
            const int k = flag_map(flag, p);
            double A[DGO_MAX_NC][DGO_MAX_NC], bv[DGO_MAX_NC], mrow[DGO_MAX_NC];
            double f0 = 0.0, eta = 0.0;
            for (int a = 0; a < ng; ++a) { bv[a] = 0.0; for (int c = 0; c <= a; ++c) A[a][c] = 0.0; }
            if (ml_mode == DGO_ML_SAMPLE && fluct_mode == DGO_FLUCT_REFERENCE) {
                double u[2];
                dgo_uniform2(seed, stream, (uint64_t)(ctx->pix0 + i), (uint32_t)k, u);
                eta = dgo_rand_normal(0.0, 1.0, u[0], u[1]);
            }
            for (int j = 0; j < nb; ++j) {
                double d = (k == 1) ? ctx->sig[IDX3(ctx, j, k, i)] / ctx->gain[j] : ctx->sig[IDX3(ctx, j, k, i)];
                for (int o = 0; o < no; ++o) d = d - dgo_eval_signal(ctx, oc[o], j, i, k, NULL);
                for (int l = 0; l < ctx->ncomp; ++l) { /* :445-460: templates / monopoles on their unfitted bands, again */
                    const dgo_comp *ct = &ctx->comps[l];
                    if ((ct->type == DGO_TEMPLATE || ct->type == DGO_MONOPOLE) && !ct->corr[j])
                        d = d - dgo_eval_signal(ctx, l, j, i, k, NULL);
                }
                for (int a = 0; a < ng; ++a) mrow[a] = dgo_eval_sed(ctx, gc[a], j, i, k, NULL);
                const double is = 1.0 / ctx->rms[IDX3(ctx, j, k, i)];
                const double inv = is * is;
                for (int a = 0; a < ng; ++a) {
                    const double t = mrow[a] * inv;
                    bv[a] += d * t;
                    for (int c = 0; c <= a; ++c) A[a][c] += t * mrow[c];
                }
                if (ml_mode == DGO_ML_SAMPLE) {
                    if (fluct_mode == DGO_FLUCT_REFERENCE) {
                        f0 += (eta * is) * mrow[ng - 1];
                    } else {
                        double u[2];
                        dgo_uniform2(seed, stream, (uint64_t)(ctx->pix0 + i), (uint32_t)(k + 4 * (j + 1)), u);
                        const double ej = dgo_rand_normal(0.0, 1.0, u[0], u[1]) * is;
                        for (int a = 0; a < ng; ++a) bv[a] += ej * mrow[a];
                    }
                }
            }
            bv[0] += f0;
            /* in-place Cholesky A = L L^t (lower), then two triangular solves */
            int ok = 1;
            for (int a = 0; a < ng && ok; ++a) {
                for (int c = 0; c <= a; ++c) {
                    double s = A[a][c];
                    for (int t = 0; t < c; ++t) s -= A[a][t] * A[c][t];
                    if (a == c) {
                        if (!(s > 0.0)) { ok = 0; break; }
                        A[a][a] = sqrt(s);
                    } else {
                        A[a][c] = s / A[c][c];
                    }
                }
            }
            if (!ok) { bad += 1; continue; }
            for (int a = 0; a < ng; ++a) {
                double s = bv[a];
                for (int t = 0; t < a; ++t) s -= A[a][t] * bv[t];
                bv[a] = s / A[a][a];
            }
            for (int a = ng - 1; a >= 0; --a) {
                double s = bv[a];
                for (int t = a + 1; t < ng; ++t) s -= A[t][a] * bv[t];
                bv[a] = s / A[a][a];
            }
            for (int a = 0; a < ng; ++a) ctx->comps[gc[a]].amplitude[IDX2(ctx, k, i)] = bv[a];
        }
    }
    if (n_not_spd) *n_not_spd = bad;
    return 0;
}

/* ------------------------------------------------------------------ sky model + chisq */

/* src/dang_data_mod.f90:339-396 */
void dgo_update_sky_model(const dgo_ctx *ctx, double *sky, double *res) {
    const int npix = ctx->npix, nb = ctx->nbands, nmaps = ctx->nmaps;
    set_threads(ctx);
    for (int64_t q = 0; q < (int64_t)nb * nmaps * npix; ++q) sky[q] = 0.0;
    for (int l = 0; l < ctx->ncomp; ++l) {
        if (ctx->comps[l].type == DGO_MONOPOLE) { /* :357-361: sets the band offsets, not part of the sky model */
            for (int j = 0; j < nb; ++j) ctx->offset[j] = ctx->comps[l].template_amplitudes[j];
            continue;
        }
#pragma omp parallel for schedule(static)
        for (int i = 0; i < npix; ++i)
            for (int k = 1; k <= nmaps; ++k)
                for (int j = 0; j < nb; ++j)
                    sky[IDX3(ctx, j, k, i)] = sky[IDX3(ctx, j, k, i)] + dgo_eval_signal(ctx, l, j, i, k, NULL);
    }
    if (!res) return;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < npix; ++i)
        for (int k = 1; k <= nmaps; ++k)
            for (int j = 0; j < nb; ++j) {
                if (k == 1) res[IDX3(ctx, j, 1, i)] = (ctx->sig[IDX3(ctx, j, 1, i)] - ctx->offset[j]) / ctx->gain[j] - sky[IDX3(ctx, j, 1, i)];
                else res[IDX3(ctx, j, k, i)] = ctx->sig[IDX3(ctx, j, k, i)] - sky[IDX3(ctx, j, k, i)];
            }
}

/* src/dang_data_mod.f90:494-526; nump is an input (quirk 9) */
double dgo_compute_chisq(const dgo_ctx *ctx, const double *sky, int pol_lo, int pol_hi, double nump, double *chi_map) {
    const int npix = ctx->npix, nb = ctx->nbands, nmaps = ctx->nmaps;
    double *cm = chi_map ? chi_map : (double *)malloc(sizeof(double) * (size_t)nmaps * npix);
    for (int64_t q = 0; q < (int64_t)nmaps * npix; ++q) cm[q] = 0.0;
    for (int i = 0; i < npix; ++i) {
        if (masked(ctx->mask[i])) continue;
        for (int k = pol_lo; k <= pol_hi; ++k)
            for (int j = 0; j < nb; ++j) {
                double rms = ctx->rms[IDX3(ctx, j, k, i)];
                double r = (k == 1) ? (ctx->sig[IDX3(ctx, j, k, i)] - ctx->offset[j]) / ctx->gain[j] - sky[IDX3(ctx, j, k, i)]
                                    : ctx->sig[IDX3(ctx, j, k, i)] - sky[IDX3(ctx, j, k, i)];
                cm[IDX2(ctx, k, i)] = cm[IDX2(ctx, k, i)] + (r * r) / (rms * rms);
            }
    }
    double s = 0.0;
    for (int64_t q = 0; q < (int64_t)nmaps * npix; ++q) { cm[q] = cm[q] / nb; s += cm[q]; }
    if (!chi_map) free(cm);
    return s / nump;
}

/* ------------------------------------------------------------------ index phase */

/* src/dang_lnl_mod.f90:126-182, single-pixel call.  data/rms/model point at
 * element (band 0, map 1, pixel 0) of arrays with the given strides. */
double dgo_evaluate_lnL(int nbands, int s1, int s2, const double *data, const double *rms, const double *model,
                        int64_t band_stride, int64_t map_stride, int pix, double maskval) {
    double lnL_local = 0.0;
    if (masked(maskval)) return 0.0;
    for (int k = s1; k <= s2; ++k)
        for (int j = 0; j < nbands; ++j) {
            int64_t q = (int64_t)j * band_stride + (int64_t)(k - 1) * map_stride + pix;
            double t = (data[q] - model[q]) / rms[q];
            lnL_local = lnL_local - 0.5 * (t * t);
        }
    return 0.0 + lnL_local;
}

/* src/dang_lnl_mod.f90:47-124, single-pixel call (no mask test, no log-det term) */
double dgo_evaluate_marginal_lnL(int nbands, int s1, int s2, const double *data, const double *rms, const double *model,
                                 int64_t band_stride, int64_t map_stride, int pix) {
    double lnL = 0.0;
    for (int j = 0; j < nbands; ++j)
        for (int k = s1; k <= s2; ++k) {
            int64_t q = (int64_t)j * band_stride + (int64_t)(k - 1) * map_stride + pix;
            double TN = model[q] / (rms[q] * rms[q]);
            double TNd = TN * data[q];
            double TNT = TN * model[q];
            double invTNT = 1.0 / TNT;
            lnL = lnL - 0.5 * TNd * invTNT * TNd;
        }
    return lnL;
}

/* src/dang_lnl_mod.f90:242-304, single-pixel call */
static double jeffreys_prior(const dgo_ctx *ctx, int comp, int s1, int s2, int pix, double val) {
    const dgo_comp *c = &ctx->comps[comp];
    double sum = 0.0, theta[DGO_MAX_IND] = {val, 0.0};
    if (c->is_synch) {
        if (!masked(ctx->mask[pix])) {
            for (int k = s1; k <= s2; ++k)
                for (int j = 0; j < ctx->nbands; ++j) {
                    double ss = dgo_eval_signal(ctx, comp, j, pix, k, theta);
                    double rr = 1.0 / ctx->rms[IDX3(ctx, j, k, pix)];
                    double t = ((rr * rr) * (ss / c->amplitude[IDX2(ctx, k, pix)]) * log(ctx->bands[j].nu_c / c->nu_ref));
                    sum = sum + t * t;
                }
        }
    }
    return sqrt(sum);
}

static double index_prior(const dgo_ctx *ctx, int comp, int nind, int s1, int s2, int pix, double val) {
    const dgo_comp *c = &ctx->comps[comp];
    switch (c->prior_type[nind]) {
    case DGO_PRIOR_GAUSSIAN: return log(dgo_eval_normal_prior(val, c->gauss_prior[nind][0], c->gauss_prior[nind][1]));
    case DGO_PRIOR_JEFFREYS: return log(jeffreys_prior(ctx, comp, s1, s2, pix, val));
    default: return 0.0;
    }
}

/* src/dang_sample_mod.f90:88-485, index_mode==2 with sample_nside==nside */
int64_t dgo_sample_index_mh(dgo_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                            uint64_t stream) {
    dgo_comp *c = &ctx->comps[comp];
    const int npix = ctx->npix, nb = ctx->nbands;
    const int s1 = (map_n == -1) ? 2 : (map_n == -2 ? 1 : map_n); /* :157-163 */
    const int s2 = (map_n == -1) ? 3 : (map_n == -2 ? 3 : map_n);
    const int64_t ms = npix, bs = (int64_t)ctx->nmaps * npix;
    int64_t accepted = 0;
    set_threads(ctx);
    double *index_map = (double *)calloc((size_t)3 * npix, sizeof(double)); /* :221-223 zero-initialised */
#pragma omp parallel for schedule(static) reduction(+ : accepted)
    for (int i = 0; i < npix; ++i) {
        if (masked(ctx->mask[i])) continue; /* :362 */
        double data[3 * 64], model[3 * 64], rmsl[3 * 64];
        double sample[DGO_MAX_IND] = {0, 0}, theta[DGO_MAX_IND] = {0, 0};
        /* :173-196 data_raw minus every OTHER component, evaluated for this pixel */
        for (int k = s1; k <= s2; ++k)
            for (int j = 0; j < nb; ++j) {
                double d = (k == 1) ? (ctx->sig[IDX3(ctx, j, 1, i)] - ctx->offset[j]) / ctx->gain[j] : ctx->sig[IDX3(ctx, j, k, i)];
                for (int l = 0; l < ctx->ncomp; ++l)
                    if (l != comp) d = d - dgo_eval_signal(ctx, l, j, i, k, NULL);
                data[(k - 1) * 64 + j] = d;
                rmsl[(k - 1) * 64 + j] = ctx->rms[IDX3(ctx, j, k, i)];
            }
        for (int l = 0; l < c->nindices; ++l) sample[l] = c->indices[((int64_t)l * ctx->nmaps + (s1 - 1)) * (int64_t)npix + i]; /* :372-374 */
        for (int l = 0; l < DGO_MAX_IND; ++l) theta[l] = sample[l];
#define FILL_MODEL(th)                                                                     \
        for (int k = s1; k <= s2; ++k)                                                     \
            for (int j = 0; j < nb; ++j) model[(k - 1) * 64 + j] = dgo_eval_signal(ctx, comp, j, i, k, (th));
#define LNL()                                                                              \
        (c->lnl_type[nind] == DGO_LNL_CHISQ    ? dgo_evaluate_lnL(nb, s1, s2, data, rmsl, model, 1, 64, 0, ctx->mask[i]) \
         : c->lnl_type[nind] == DGO_LNL_MARGINAL ? dgo_evaluate_marginal_lnL(nb, s1, s2, data, rmsl, model, 1, 64, 0)    \
                                                 : 0.0)
        FILL_MODEL(sample) /* :380 */
        int sample_it = 1;
        double lnl = LNL();
        if (c->lnl_type[nind] == DGO_LNL_PRIOR) { /* :389-392 */
            double u[2];
            sample_it = 0;
            dgo_uniform2(seed, stream, (uint64_t)(ctx->pix0 + i), 0u, u);
            sample[nind] = dgo_rand_normal(c->gauss_prior[nind][0], c->gauss_prior[nind][1], u[0], u[1]);
        }
        double lnl_old = lnl + index_prior(ctx, comp, nind, s1, s2, i, sample[nind]); /* :394-402 */
        if (sample_it) {
            for (int l = 1; l <= nsample; ++l) {
                double u[3];
                dgo_uniform3(seed, stream, (uint64_t)(ctx->pix0 + i), (uint32_t)l, u);
                theta[nind] = sample[nind] + dgo_rand_normal(0.0, c->step_size[nind], u[0], u[1]); /* :414 */
                if (theta[nind] < c->uni_prior[nind][0] || theta[nind] > c->uni_prior[nind][1]) continue; /* :415 */
                FILL_MODEL(theta)
                lnl = LNL();
                double lnl_new = lnl + index_prior(ctx, comp, nind, s1, s2, i, theta[nind]);
                double diff = lnl_new - lnl_old;
                if (ml_mode == DGO_ML_OPTIMIZE) { /* :443-447 */
                    if (diff > 0.0) { sample[nind] = theta[nind]; lnl_old = lnl_new; accepted += 1; }
                } else { /* :448-454 */
                    /* :449-450 call RANDOM_NUMBER(num); if (diff > log(num)) */
                    if (diff > log(u[2])) { sample[nind] = theta[nind]; lnl_old = lnl_new; accepted += 1; }
                }
            }
        }
        for (int k = s1; k <= s2; ++k) index_map[(int64_t)(k - 1) * npix + i] = sample[nind]; /* :465 */
#undef FILL_MODEL
#undef LNL
    }
    /* :480-483 udgrade_ring at equal Nside is a copy; masked pixels receive 0 */
    for (int k = s1; k <= s2; ++k)
        for (int i = 0; i < npix; ++i)
            c->indices[((int64_t)nind * ctx->nmaps + (k - 1)) * (int64_t)npix + i] = index_map[(int64_t)(k - 1) * npix + i];
    free(index_map);
    (void)ms; (void)bs;
    return accepted;
}

/* ------------------------------------------------------------------ coarse-Nside index sampling
 * HEALPix is an external library of the reference (not in its tree; linked as -lhealpix, any 3.x release): nest2ring
 * and udgrade_ring are restated from the published algorithm (Gorski et al. 2005, ApJ 622, 759; pix_tools::nest2ring,
 * udgrade_nr::udgrade_ring -> sub_udgrade_nest with pessimistic = .false.). */

/* face f = ipnest / nside^2; (ix, iy) = even / odd bits of the in-face index; ring jr from the north pole; jp in ring */
int64_t dgo_nest2ring(int nside, int64_t ipnest) {
    static const int jrll[12] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4}, jpll[12] = {1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7};
    const int64_t ns2 = (int64_t)nside * nside, npix = 12 * ns2, ncap = 2 * (int64_t)nside * (nside - 1);
    const int face = (int)(ipnest / ns2);
    const int64_t ipf = ipnest % ns2;
    int ix = 0, iy = 0;
    for (int b = 0; b < 16; ++b) {
        ix |= (int)((ipf >> (2 * b)) & 1) << b;
        iy |= (int)((ipf >> (2 * b + 1)) & 1) << b;
    }
    const int64_t jr = (int64_t)jrll[face] * nside - ix - iy - 1;
    int64_t nr, n_before;
    int kshift;
    if (jr < nside) { nr = jr; n_before = 2 * nr * (nr - 1); kshift = 0; }
    else if (jr > 3 * (int64_t)nside) { nr = 4 * (int64_t)nside - jr; n_before = npix - 2 * (nr + 1) * nr; kshift = 0; }
    else { nr = nside; n_before = ncap + (jr - nside) * 4 * (int64_t)nside; kshift = (int)((jr - nside) & 1); }
    int64_t jp = ((int64_t)jpll[face] * nr + ix - iy + 1 + kshift) / 2;
    if (jp > 4 * nr) jp -= 4 * nr;
    if (jp < 1) jp += 4 * nr;
    return n_before + jp - 1;
}

/* udgrade_ring for one map: RING -> NEST, mean of the good children (degrade) or the parent's value (upgrade),
 * NEST -> RING.  mode 0: as is; 1: udgrade_rms (src/dang_util_mod.f90:341-356); 2: udgrade_mask with threshold 0.5
 * (:358-376) */
void dgo_udgrade(int mode, const double *in, int nside_in, double *out, int nside_out) {
    const int64_t npi = 12 * (int64_t)nside_in * nside_in, npo = 12 * (int64_t)nside_out * nside_out;
    const int degrade = nside_in > nside_out;
    const int r1 = degrade ? nside_in / nside_out : nside_out / nside_in;
    const int64_t ratio = (int64_t)r1 * r1;
    double *nin = (double *)malloc(sizeof(double) * (size_t)npi);
    for (int64_t p = 0; p < npi; ++p) {
        double v = in[dgo_nest2ring(nside_in, p)];
        nin[p] = (mode == 1) ? v * v : v;
    }
    for (int64_t q = 0; q < npo; ++q) {
        double v;
        if (degrade) {
            double total = 0.0;
            int nobs = 0;
            for (int64_t ip = 0; ip < ratio; ++ip) {
                double x = nin[q * ratio + ip];
                if (fabs(x - MISSVAL) > fabs(1e-5 * MISSVAL)) { total = total + x; ++nobs; }
            }
            v = nobs ? total / nobs : MISSVAL;
        } else {
            v = nin[q / ratio];
        }
        if (mode == 1) v = sqrt(v) * ((double)nside_out * 1.0 / nside_in);
        if (mode == 2 && degrade) v = (v < 0.5) ? 0.0 : 1.0;
        out[dgo_nest2ring(nside_out, q)] = v;
    }
    free(nin);
}

/* src/dang_sample_mod.f90:88-485, index_mode == 2 with sample_nside < nside -- literally: the chain of COARSE pixel i
 * reads ddata%masks(i,1) (:362), c%indices(i, map_inds(1), :) (:372-377) and, inside eval_signal, c%amplitude(i,k)
 * (:548-553 -> src/dang_component_mod.f90:773) from the FULL-resolution arrays at the same index i. */
int64_t dgo_sample_index_mh_coarse(dgo_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                                   uint64_t stream, int nside, int sample_nside) {
    dgo_comp *c = &ctx->comps[comp];
    const int npix = ctx->npix, nb = ctx->nbands;
    const int s1 = (map_n == -1) ? 2 : map_n, s2 = (map_n == -1) ? 3 : map_n, Sp = s2 - s1 + 1;
    const int64_t npc = 12 * (int64_t)sample_nside * sample_nside;
    if (ctx->pix0 != 0 || npix != 12 * nside * nside || sample_nside >= nside) return -1;
    double *full = (double *)malloc(sizeof(double) * (size_t)npix);
    double *cdata = (double *)malloc(sizeof(double) * (size_t)(Sp * nb) * npc);
    double *crms = (double *)malloc(sizeof(double) * (size_t)(Sp * nb) * npc);
    double *cmask = (double *)malloc(sizeof(double) * (size_t)npc);
    double *index_map = (double *)calloc((size_t)npc, sizeof(double));
    double *index_full = (double *)malloc(sizeof(double) * (size_t)npix);
    int64_t accepted = 0;
    set_threads(ctx);
    for (int kk = 0; kk < Sp; ++kk)
        for (int j = 0; j < nb; ++j) {
            const int k = s1 + kk;
#pragma omp parallel for schedule(static)
            for (int i = 0; i < npix; ++i) { /* :173-196 */
                double d = (k == 1) ? (ctx->sig[IDX3(ctx, j, 1, i)] - ctx->offset[j]) / ctx->gain[j] : ctx->sig[IDX3(ctx, j, k, i)];
                for (int l = 0; l < ctx->ncomp; ++l)
                    if (l != comp) d = d - dgo_eval_signal(ctx, l, j, i, k, NULL);
                full[i] = d;
            }
            dgo_udgrade(0, full, nside, cdata + ((int64_t)kk * nb + j) * npc, sample_nside);                 /* :213 */
            dgo_udgrade(1, ctx->rms + IDX3(ctx, j, k, 0), nside, crms + ((int64_t)kk * nb + j) * npc, sample_nside); /* :214 */
        }
    dgo_udgrade(2, ctx->mask, nside, cmask, sample_nside); /* :209 */
#pragma omp parallel for schedule(static) reduction(+ : accepted)
    for (int64_t i = 0; i < npc; ++i) {
        if (masked(ctx->mask[i])) continue; /* :362: the full-resolution mask at the coarse index */
        double sample[DGO_MAX_IND] = {0, 0}, theta[DGO_MAX_IND] = {0, 0};
        double data[3 * 64], model[3 * 64], rmsl[3 * 64];
        for (int kk = 0; kk < Sp; ++kk)
            for (int j = 0; j < nb; ++j) {
                data[(s1 + kk - 1) * 64 + j] = cdata[((int64_t)kk * nb + j) * npc + i];
                rmsl[(s1 + kk - 1) * 64 + j] = crms[((int64_t)kk * nb + j) * npc + i];
            }
        for (int l = 0; l < c->nindices; ++l) sample[l] = c->indices[((int64_t)l * ctx->nmaps + (s1 - 1)) * (int64_t)npix + i];
        for (int l = 0; l < DGO_MAX_IND; ++l) theta[l] = sample[l];
#define FILL_MODEL(th)                                                                     \
        for (int k = s1; k <= s2; ++k)                                                     \
            for (int j = 0; j < nb; ++j) model[(k - 1) * 64 + j] = dgo_eval_signal(ctx, comp, j, (int)i, k, (th));
#define LNL()                                                                              \
        (c->lnl_type[nind] == DGO_LNL_CHISQ    ? dgo_evaluate_lnL(nb, s1, s2, data, rmsl, model, 1, 64, 0, cmask[i]) \
         : c->lnl_type[nind] == DGO_LNL_MARGINAL ? dgo_evaluate_marginal_lnL(nb, s1, s2, data, rmsl, model, 1, 64, 0) \
                                                 : 0.0)
/* eval_jeffreys_prior(c,data,rms,model,map_inds,i,mask(:,1),val), src/dang_lnl_mod.f90:242-304: rms and mask are the
 * DEGRADED ones, c%eval_signal(j,i,k,theta) and c%amplitude(i,k) the full-resolution arrays at the coarse pixel number */
#define JEFFREYS(v, out)                                                                                         \
        do {                                                                                                     \
            double sum_ = 0.0, th_[DGO_MAX_IND] = {(v), 0.0};                                                    \
            if (c->is_synch && !masked(cmask[i]))                                                                \
                for (int k = s1; k <= s2; ++k)                                                                   \
                    for (int j = 0; j < nb; ++j) {                                                               \
                        double ss_ = dgo_eval_signal(ctx, comp, j, (int)i, k, th_);                              \
                        double rr_ = 1.0 / rmsl[(k - 1) * 64 + j];                                               \
                        double t_ = ((rr_ * rr_) * (ss_ / c->amplitude[IDX2(ctx, k, i)]) * log(ctx->bands[j].nu_c / c->nu_ref)); \
                        sum_ = sum_ + t_ * t_;                                                                   \
                    }                                                                                            \
            (out) = log(sqrt(sum_));                                                                             \
        } while (0)
#define PRIOR(v, out)                                                                                            \
        do {                                                                                                     \
            if (c->prior_type[nind] == DGO_PRIOR_GAUSSIAN) (out) = log(dgo_eval_normal_prior((v), c->gauss_prior[nind][0], c->gauss_prior[nind][1])); \
            else if (c->prior_type[nind] == DGO_PRIOR_JEFFREYS) JEFFREYS((v), (out));                            \
            else (out) = 0.0;                                                                                    \
        } while (0)
        FILL_MODEL(sample)
        int sample_it = 1;
        double lnl = LNL();
        if (c->lnl_type[nind] == DGO_LNL_PRIOR) {
            double u[2];
            sample_it = 0;
            dgo_uniform2(seed, stream, (uint64_t)i, 0u, u);
            sample[nind] = dgo_rand_normal(c->gauss_prior[nind][0], c->gauss_prior[nind][1], u[0], u[1]);
        }
        double lnl_prior;
        PRIOR(sample[nind], lnl_prior);
        double lnl_old = lnl + lnl_prior;
        if (sample_it) {
            for (int l = 1; l <= nsample; ++l) {
                double u[3];
                dgo_uniform3(seed, stream, (uint64_t)i, (uint32_t)l, u);
                theta[nind] = sample[nind] + dgo_rand_normal(0.0, c->step_size[nind], u[0], u[1]);
                if (theta[nind] < c->uni_prior[nind][0] || theta[nind] > c->uni_prior[nind][1]) continue;
                FILL_MODEL(theta)
                lnl = LNL();
                PRIOR(theta[nind], lnl_prior);
                double lnl_new = lnl + lnl_prior;
                double diff = lnl_new - lnl_old;
                if (ml_mode == DGO_ML_OPTIMIZE) {
                    if (diff > 0.0) { sample[nind] = theta[nind]; lnl_old = lnl_new; accepted += 1; }
                } else {
                    if (diff > log(u[2])) { sample[nind] = theta[nind]; lnl_old = lnl_new; accepted += 1; }
                }
            }
        }
        index_map[i] = sample[nind]; /* :465 */
#undef FILL_MODEL
#undef LNL
#undef PRIOR
#undef JEFFREYS
    }
    dgo_udgrade(0, index_map, sample_nside, index_full, nside); /* :480 */
    for (int k = s1; k <= s2; ++k)
        for (int i = 0; i < npix; ++i) c->indices[((int64_t)nind * ctx->nmaps + (k - 1)) * (int64_t)npix + i] = index_full[i]; /* :483 */
    free(full); free(cdata); free(crms); free(cmask); free(index_map); free(index_full);
    return accepted;
}

/* ------------------------------------------------------------------ full-sky index mode, tuner, gain fit */

#define DGO_GLOBAL_PIX ((uint64_t)0xFFFFFFFFFFull)

typedef struct {
    const dgo_ctx *ctx;
    int comp, nind, s1, s2;
    double *data; /* [nb][nmaps][npix] data_raw minus every other component (:173-196) */
    /* sample_nside /= nside (:199-217): npc > 0 coarse pixels, degraded data / rms [kk][j][npc] and mask [npc]; the model
     * (eval_signal) still reads c%amplitude of the full-resolution array at the coarse pixel number */
    int64_t npc;
    double *cdata, *crms, *cmask;
} fs_state;

static int fs_npix(const fs_state *S) { return S->npc > 0 ? (int)S->npc : S->ctx->npix; }
static double fs_d(const fs_state *S, int j, int k, int i) {
    return S->npc > 0 ? S->cdata[((int64_t)(k - S->s1) * S->ctx->nbands + j) * S->npc + i] : S->data[IDX3(S->ctx, j, k, i)];
}
static double fs_r(const fs_state *S, int j, int k, int i) {
    return S->npc > 0 ? S->crms[((int64_t)(k - S->s1) * S->ctx->nbands + j) * S->npc + i] : S->ctx->rms[IDX3(S->ctx, j, k, i)];
}
static double fs_m(const fs_state *S, int i) { return S->npc > 0 ? S->cmask[i] : S->ctx->mask[i]; }

/* update_sample_model without pixel (:555-563) + evaluate_lnL / evaluate_marginal_lnL over the sky */
static double fs_lnl(const fs_state *S, const double *theta) {
    const dgo_ctx *ctx = S->ctx;
    const dgo_comp *c = &ctx->comps[S->comp];
    const int npix = fs_npix(S), nb = ctx->nbands;
    if (c->lnl_type[S->nind] == DGO_LNL_CHISQ) { /* src/dang_lnl_mod.f90:168-180: i outer, k, j inner */
        double lnL = 0.0;
        for (int i = 0; i < npix; ++i) {
            if (masked(fs_m(S, i))) continue;
            for (int k = S->s1; k <= S->s2; ++k)
                for (int j = 0; j < nb; ++j) {
                    double m = dgo_eval_signal(ctx, S->comp, j, i, k, theta);
                    double t = (fs_d(S, j, k, i) - m) / fs_r(S, j, k, i);
                    lnL = lnL - 0.5 * (t * t);
                }
        }
        return lnL;
    }
    if (c->lnl_type[S->nind] == DGO_LNL_MARGINAL) { /* src/dang_lnl_mod.f90:113-122: no mask test */
        double lnL = 0.0;
        for (int j = 0; j < nb; ++j)
            for (int k = S->s1; k <= S->s2; ++k) {
                double TNd = 0.0, TNT = 0.0;
                for (int i = 0; i < npix; ++i) {
                    double m = dgo_eval_signal(ctx, S->comp, j, i, k, theta);
                    double rms = fs_r(S, j, k, i);
                    double TN = m / (rms * rms);
                    TNd += TN * fs_d(S, j, k, i);
                    TNT += TN * m;
                }
                lnL = lnL - 0.5 * TNd * (1.0 / TNT) * TNd;
            }
        return lnL;
    }
    return 0.0;
}

/* eval_jeffreys_prior over the sky (src/dang_lnl_mod.f90:289-302) */
static double fs_jeffreys(const fs_state *S, double val) {
    const dgo_ctx *ctx = S->ctx;
    const dgo_comp *c = &ctx->comps[S->comp];
    double sum = 0.0, theta[DGO_MAX_IND] = {val, 0.0};
    if (c->is_synch)
        for (int i = 0; i < fs_npix(S); ++i) {
            if (masked(fs_m(S, i))) continue;
            for (int k = S->s1; k <= S->s2; ++k)
                for (int j = 0; j < ctx->nbands; ++j) {
                    double ss = dgo_eval_signal(ctx, S->comp, j, i, k, theta);
                    double rr = 1.0 / fs_r(S, j, k, i);
                    double t = ((rr * rr) * (ss / c->amplitude[IDX2(ctx, k, i)]) * log(ctx->bands[j].nu_c / c->nu_ref));
                    sum = sum + t * t;
                }
        }
    return sqrt(sum);
}

static double fs_prior(const fs_state *S, double val) {
    const dgo_comp *c = &S->ctx->comps[S->comp];
    switch (c->prior_type[S->nind]) {
    case DGO_PRIOR_GAUSSIAN: return log(dgo_eval_normal_prior(val, c->gauss_prior[S->nind][0], c->gauss_prior[S->nind][1]));
    case DGO_PRIOR_JEFFREYS: return log(fs_jeffreys(S, val));
    default: return 0.0;
    }
}

/* tune_spectral_parameter_length, src/dang_sample_mod.f90:623-717.  theta_init has 2 entries (quirk 10). */
static void fs_tune(fs_state *S, dgo_comp *c, const double *theta_init, int nsample, int ml_mode, uint64_t seed,
                    uint64_t stream, uint32_t *draw, int *tuned) {
    double sample[2] = {theta_init[0], theta_init[1]}, theta[2] = {theta_init[0], theta_init[1]};
    const int nind = S->nind;
    double lnl = 0.0, lnl_new = 0.0, lnl_old = 0.0;
    if (c->lnl_type[nind] == DGO_LNL_CHISQ || c->lnl_type[nind] == DGO_LNL_MARGINAL) lnl = fs_lnl(S, sample);
    else if (c->lnl_type[nind] == DGO_LNL_PRIOR) {
        double u[2];
        dgo_uniform2(seed, stream, DGO_GLOBAL_PIX, (*draw)++, u);
        sample[nind] = dgo_rand_normal(c->gauss_prior[nind][0], c->gauss_prior[nind][1], u[0], u[1]);
    }
    if (c->prior_type[nind] == DGO_PRIOR_GAUSSIAN) lnl_old = lnl + log(dgo_eval_normal_prior(sample[nind], c->gauss_prior[nind][0], c->gauss_prior[nind][1]));
    else if (c->prior_type[nind] == DGO_PRIOR_UNIFORM) lnl_old = lnl;
    int guard = 0;
    while (!*tuned && guard++ < 64) { /* :663 do while (.not. c%tuned(nind)); guard: the reference can loop forever */
        double accept = 0.0;
        int l;
        for (l = 1; l <= nsample; ++l) {
            double u[3];
            dgo_uniform3(seed, stream, DGO_GLOBAL_PIX, (*draw)++, u);
            theta[nind] = sample[nind] + dgo_rand_normal(0.0, c->step_size[nind], u[0], u[1]);
            if (theta[nind] < c->uni_prior[nind][0] || theta[nind] > c->uni_prior[nind][1]) continue;
            if (c->lnl_type[nind] == DGO_LNL_CHISQ || c->lnl_type[nind] == DGO_LNL_MARGINAL) lnl = fs_lnl(S, theta);
            if (c->prior_type[nind] == DGO_PRIOR_GAUSSIAN) lnl_new = lnl + log(dgo_eval_normal_prior(theta[nind], c->gauss_prior[nind][0], c->gauss_prior[nind][1]));
            else if (c->prior_type[nind] == DGO_PRIOR_UNIFORM) lnl_new = lnl;
            double diff = lnl_new - lnl_old, ratio = exp(diff);
            if (ml_mode == DGO_ML_OPTIMIZE) {
                if (ratio > 1.0) { sample[nind] = theta[nind]; lnl_old = lnl_new; accept = accept + 1; }
            } else {
                if (ratio > u[2]) { sample[nind] = theta[nind]; lnl_old = lnl_new; accept = accept + 1; }
            }
            lnl = 0.0; /* :705 */
        }
        /* :707-713  after the loop l == nsample+1 */
        if (accept / l < 0.4f) c->step_size[nind] = c->step_size[nind] - 0.5f * c->step_size[nind];
        else if (accept / l > 0.6f) c->step_size[nind] = c->step_size[nind] + 0.5f * c->step_size[nind];
        else *tuned = 1;
    }
}

static int64_t sample_index_fullsky_impl(dgo_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                                         uint64_t stream, int *tuned, int nside, int sample_nside);

int64_t dgo_sample_index_fullsky(dgo_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                                 uint64_t stream, int *tuned) {
    return sample_index_fullsky_impl(ctx, comp, nind, map_n, nsample, ml_mode, seed, stream, tuned, 0, 0);
}

/* index_mode == 1 with c%sample_nside(nind) /= nside (src/dang_sample_mod.f90:199-217, 229-329) */
int64_t dgo_sample_index_fullsky_coarse(dgo_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                                        uint64_t stream, int *tuned, int nside, int sample_nside) {
    if (ctx->pix0 != 0 || ctx->npix != 12 * nside * nside || sample_nside >= nside) return -1;
    return sample_index_fullsky_impl(ctx, comp, nind, map_n, nsample, ml_mode, seed, stream, tuned, nside, sample_nside);
}

static int64_t sample_index_fullsky_impl(dgo_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                                         uint64_t stream, int *tuned, int nside, int sample_nside) {
    dgo_comp *c = &ctx->comps[comp];
    const int npix = ctx->npix, nb = ctx->nbands, nmaps = ctx->nmaps;
    fs_state S = {ctx, comp, nind, (map_n == -1) ? 2 : map_n, (map_n == -1) ? 3 : map_n, NULL, 0, NULL, NULL, NULL};
    uint32_t draw = 1;
    int64_t accepted = 0;
    S.data = (double *)malloc(sizeof(double) * (size_t)nb * nmaps * npix);
    /* :173-196 (all pixels, no mask test) */
    for (int i = 0; i < npix; ++i)
        for (int k = 1; k <= nmaps; ++k)
            for (int j = 0; j < nb; ++j) {
                double d = (k == 1) ? (ctx->sig[IDX3(ctx, j, 1, i)] - ctx->offset[j]) / ctx->gain[j] : ctx->sig[IDX3(ctx, j, k, i)];
                for (int l = 0; l < ctx->ncomp; ++l)
                    if (l != comp) d = d - dgo_eval_signal(ctx, l, j, i, k, NULL);
                S.data[IDX3(ctx, j, k, i)] = d;
            }
    if (sample_nside > 0) { /* :204-217: udgrade_ring(data), udgrade_rms(rms), udgrade_mask(mask) */
        const int Sp = S.s2 - S.s1 + 1;
        S.npc = 12 * (int64_t)sample_nside * sample_nside;
        S.cdata = (double *)malloc(sizeof(double) * (size_t)(Sp * nb) * S.npc);
        S.crms = (double *)malloc(sizeof(double) * (size_t)(Sp * nb) * S.npc);
        S.cmask = (double *)malloc(sizeof(double) * (size_t)S.npc);
        for (int k = S.s1; k <= S.s2; ++k)
            for (int j = 0; j < nb; ++j) {
                dgo_udgrade(0, S.data + IDX3(ctx, j, k, 0), nside, S.cdata + ((int64_t)(k - S.s1) * nb + j) * S.npc, sample_nside);
                dgo_udgrade(1, ctx->rms + IDX3(ctx, j, k, 0), nside, S.crms + ((int64_t)(k - S.s1) * nb + j) * S.npc, sample_nside);
            }
        dgo_udgrade(2, ctx->mask, nside, S.cmask, sample_nside);
    }
    double sample[DGO_MAX_IND] = {0, 0}, theta[DGO_MAX_IND] = {0, 0};
    for (int l = 0; l < c->nindices; ++l) sample[l] = c->indices[((int64_t)l * nmaps + (S.s1 - 1)) * (int64_t)npix + 0]; /* :240-242 */
    for (int l = 0; l < DGO_MAX_IND; ++l) theta[l] = sample[l];
    double lnl = 0.0;
    int sample_it = 1;
    if (c->lnl_type[nind] == DGO_LNL_CHISQ || c->lnl_type[nind] == DGO_LNL_MARGINAL) lnl = fs_lnl(&S, sample);
    else if (c->lnl_type[nind] == DGO_LNL_PRIOR) { /* :255-257 */
        double u[2];
        sample_it = 0;
        dgo_uniform2(seed, stream, DGO_GLOBAL_PIX, 0u, u);
        sample[nind] = dgo_rand_normal(c->gauss_prior[nind][0], c->gauss_prior[nind][1], u[0], u[1]);
    }
    double lnl_old = lnl + fs_prior(&S, sample[nind]); /* :260-268 */
    if (sample_it) {
        if (!*tuned) fs_tune(&S, c, sample, nsample, ml_mode, seed, stream ^ 0x5555555555555555ull, &draw, tuned); /* :272-275 */
        for (int l = 0; l < c->nindices; ++l) sample[l] = c->indices[((int64_t)l * nmaps + (S.s1 - 1)) * (int64_t)npix + 0];
        for (int l = 0; l < DGO_MAX_IND; ++l) theta[l] = sample[l];
        for (int l = 1; l <= nsample; ++l) { /* :282-324 */
            double u[3];
            dgo_uniform3(seed, stream, DGO_GLOBAL_PIX, (uint32_t)l, u);
            theta[nind] = sample[nind] + dgo_rand_normal(0.0, c->step_size[nind], u[0], u[1]);
            if (theta[nind] < c->uni_prior[nind][0] || theta[nind] > c->uni_prior[nind][1]) continue;
            lnl = fs_lnl(&S, theta);
            double lnl_new = lnl + fs_prior(&S, theta[nind]);
            double diff = lnl_new - lnl_old, ratio = exp(diff);
            if (ml_mode == DGO_ML_OPTIMIZE) {
                if (ratio > 1.0) { sample[nind] = theta[nind]; lnl_old = lnl_new; accepted += 1; }
            } else {
                if (ratio > u[2]) { sample[nind] = theta[nind]; lnl_old = lnl_new; accepted += 1; }
            }
        }
    }
    /* :329, :483  every pixel (masked ones too) receives the sampled value */
    for (int k = S.s1; k <= S.s2; ++k)
        for (int i = 0; i < npix; ++i) c->indices[((int64_t)nind * nmaps + (k - 1)) * (int64_t)npix + i] = sample[nind];
    free(S.data);
    free(S.cdata); free(S.crms); free(S.cmask);
    return accepted;
}

/* The step-size tuning of the PER-PIXEL branch, src/dang_sample_mod.f90:341-346:
 *     if (.not. c%tuned(nind)) then
 *        do l = 1, c%nindices
 *           sample(l) = sum(c%indices(:,map_inds(1),l))/sum(mask(:,1))
 *           call tune_spectral_parameter_length(c,nind,sample,data,rms,model,map_inds,mask(:,1))
 *        end do
 *     end if
 * `sample` starts as zeros (:337) and is filled one entry per pass, so the first pass tunes index nind of a two-index
 * component with the OTHER index at 0 when nind is the first one; the tuner marks all indices tuned (:712), so later
 * passes only evaluate the starting likelihood.  The sums run over every pixel, the mask VALUES are summed.  The data
 * are data_raw minus every other component (:173-196), as in the full-sky mode (sample_nside == nside here).
 * Draw counter and stream follow dgo_sample_index_fullsky's tuner call. */
void dgo_tune_perpixel(dgo_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                       uint64_t stream, int *tuned) {
    dgo_comp *c = &ctx->comps[comp];
    const int npix = ctx->npix, nb = ctx->nbands, nmaps = ctx->nmaps;
    fs_state S = {ctx, comp, nind, (map_n == -1) ? 2 : map_n, (map_n == -1) ? 3 : map_n, NULL, 0, NULL, NULL, NULL};
    if (*tuned) return;
    S.data = (double *)malloc(sizeof(double) * (size_t)nb * nmaps * npix);
    for (int i = 0; i < npix; ++i)
        for (int k = 1; k <= nmaps; ++k)
            for (int j = 0; j < nb; ++j) {
                double d = (k == 1) ? (ctx->sig[IDX3(ctx, j, 1, i)] - ctx->offset[j]) / ctx->gain[j] : ctx->sig[IDX3(ctx, j, k, i)];
                for (int l = 0; l < ctx->ncomp; ++l)
                    if (l != comp) d = d - dgo_eval_signal(ctx, l, j, i, k, NULL);
                S.data[IDX3(ctx, j, k, i)] = d;
            }
    double msum = 0.0;
    for (int i = 0; i < npix; ++i) msum += ctx->mask[i];
    double sample[DGO_MAX_IND] = {0.0, 0.0};
    uint32_t draw = 1;
    for (int l = 0; l < c->nindices; ++l) {
        double isum = 0.0;
        for (int i = 0; i < npix; ++i) isum += c->indices[((int64_t)l * nmaps + (S.s1 - 1)) * (int64_t)npix + i];
        sample[l] = isum / msum;
        fs_tune(&S, c, sample, nsample, ml_mode, seed, stream ^ 0x5555555555555555ull, &draw, tuned);
    }
    free(S.data);
}

/* fit_band_gain, src/dang_sample_mod.f90:570-621 (map_n = 1) */
double dgo_fit_band_gain(const dgo_ctx *ctx, const double *sky, const double *res, int band, int ml_mode, uint64_t seed,
                         uint64_t stream) {
    double mu = 0.0, sigma = 0.0;
    for (int i = 0; i < ctx->npix; ++i) {
        if (masked(ctx->mask[i])) continue; /* :597-604 */
        double noise = ctx->rms[IDX3(ctx, band, 1, i)];
        double N_inv = 1.0 / (noise * noise);
        double map1 = sky[IDX3(ctx, band, 1, i)];
        double map2 = res[IDX3(ctx, band, 1, i)] + sky[IDX3(ctx, band, 1, i)];
        mu += map2 * N_inv * map1;
        sigma += map1 * N_inv * map1;
    }
    mu = mu / sigma;
    sigma = sqrt(1.0 / sigma);
    if (ml_mode == DGO_ML_OPTIMIZE) return mu;
    double u[2];
    dgo_uniform2(seed, stream, DGO_GLOBAL_PIX, (uint32_t)band, u);
    return mu + sigma * dgo_rand_normal(0.0, 1.0, u[0], u[1]);
}
