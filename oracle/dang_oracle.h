/*
 * dang_oracle.h -- CPU restatement of the dang Gibbs inner loop (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X path in dang_amd/.  It is a plain-C
 * restatement of the reference algorithm (hermda02/dang, Fortran 90); every
 * function cites the reference file:line it follows.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it.  The product
 * (libdangx.so) never includes, links or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures,
 * and it cannot be built in this image (every module `use`s the HEALPix-F90 /
 * CFITSIO-F90 / MPI Fortran modules, which are absent; building it would need
 * stand-ins for those libraries, which is not allowed).  The oracle is therefore
 * pinned only by (i) published known-answer vectors for Philox4x32-10, (ii)
 * closed-form values of the SED formulas, (iii) algebraic identities of the
 * solve (noise-free recovery, CG == direct block solve, posterior moments), and
 * (iv) for the HEALPix pieces of the coarse-Nside path (an external library of
 * the reference, restated from the published algorithm) the documented nside = 2
 * nest2ring table and the agreement of the RING / NESTED pixel-centre formulas.
 *
 * Array layout (= the Fortran arrays as they sit in memory, passed unchanged):
 *   sig/rms  : Fortran (0:npix-1, nmaps, nbands)  ->  C [band][map][pix]
 *   masks    : Fortran (0:npix-1, nmaps)          ->  C [map][pix]   (plane 0 is tested)
 *   amplitude: Fortran (0:npix-1, nmaps)          ->  C [map][pix]
 *   indices  : Fortran (0:npix-1, nmaps, nind)    ->  C [ind][map][pix]
 * map numbers (map_n) are 1-based as in the reference: 1=T, 2=Q, 3=U.
 */
#ifndef DANG_ORACLE_H
#define DANG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* component types: src/dang_component_mod.f90:791-809 (diffuse ones) */
enum { DGO_POWERLAW = 1, DGO_MBB = 2, DGO_FREEFREE = 3, DGO_LOGNORMAL = 4, DGO_CMB = 5, DGO_TCMB = 6,
       DGO_TEMPLATE = 7, DGO_MONOPOLE = 8, DGO_HIFIT = 9 }; /* 7-9: global-amplitude types */
/* lnl_type / prior_type: src/dang_sample_mod.f90:383-400 */
enum { DGO_LNL_CHISQ = 1, DGO_LNL_MARGINAL = 2, DGO_LNL_PRIOR = 3 };
enum { DGO_PRIOR_GAUSSIAN = 1, DGO_PRIOR_UNIFORM = 2, DGO_PRIOR_JEFFREYS = 3 };
/* ml_mode: src/dang_cg_mod.f90:254-267 */
enum { DGO_ML_SAMPLE = 1, DGO_ML_OPTIMIZE = 2 };
/* poltype flags: src/dang_util_mod.f90:228-292 */
enum { DGO_FLAG_T = 1, DGO_FLAG_Q = 2, DGO_FLAG_U = 4, DGO_FLAG_QU = 8 };
/* fluctuation-term mode: 1 reproduces quirks 2+3 of compute_sample_vector
 * (src/dang_cg_mod.f90:1008-1040), 0 is the textbook sampler. */
enum { DGO_FLUCT_CORRECT = 0, DGO_FLUCT_REFERENCE = 1 };

#define DGO_MAX_IND 2

typedef struct {
    double nu_c;          /* Hz, src/dang_bp_mod.f90:34-37 */
    int n;                /* 0 => 'delta' bandpass */
    const double *nu0;    /* [n] Hz */
    const double *tau0;   /* [n] normalised weights */
} dgo_band;

typedef struct {
    int type;
    int is_synch;         /* label=='synch' (jeffreys prior, src/dang_lnl_mod.f90:289) */
    double nu_ref;        /* Hz */
    int nindices;
    int cg_group;
    int sample_amplitude;
    double *amplitude;    /* [nmaps][npix] */
    double *indices;      /* [nindices][nmaps][npix] */
    int lnl_type[DGO_MAX_IND];
    int prior_type[DGO_MAX_IND];
    double gauss_prior[DGO_MAX_IND][2];
    double uni_prior[DGO_MAX_IND][2];
    double step_size[DGO_MAX_IND];
    /* global-amplitude types only (src/dang_component_mod.f90:17,22,31,33) */
    int nfit;                     /* number of fitted (corr) bands */
    const int *corr;              /* [nbands] c%corr(j) */
    const double *tmpl;           /* c%template, [nmaps][npix] */
    double *template_amplitudes;  /* c%template_amplitudes(band,map) stored [map][band] */
} dgo_comp;

typedef struct {
    int npix, nmaps, nbands, ncomp;
    int64_t pix0;         /* global index of local pixel 0 (RNG keying when sharded) */
    const double *sig;    /* [nbands][nmaps][npix] */
    const double *rms;    /* [nbands][nmaps][npix] */
    const double *mask;   /* [nmaps][npix] */
    const double *gain;   /* [nbands] */
    double *offset;       /* [nbands]; update_sky_model overwrites it from a monopole component (:357-361) */
    const dgo_band *bands;
    dgo_comp *comps;
    double T_CMB;         /* src/dang_util_mod.f90:15 (global, mutable in the reference) */
    int nthreads;         /* OpenMP threads for the pixel loops (0 = runtime default) */
} dgo_ctx;

/* ---- constants (src/dang_util_mod.f90:12-19) ---- */
double dgo_const_h(void);
double dgo_const_kB(void);
double dgo_const_c(void);
double dgo_missval(void);

/* ---- RNG: Philox4x32-10 keyed counter stream (builder-defined; replaces the
 * reference's unseeded RANDOM_NUMBER, src/dang.f90:67) ---- */
void dgo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* two uniforms in (0,1) for (seed, stream, pixel, draw) */
void dgo_uniform2(uint64_t seed, uint64_t stream, uint64_t pix, uint32_t draw, double u[2]);
/* three uniforms from ONE Philox call (Metropolis step): u[0] 53 bits, u[1], u[2] 32 bits */
void dgo_uniform3(uint64_t seed, uint64_t stream, uint64_t pix, uint32_t draw, double u[3]);
/* rand_normal, src/dang_util_mod.f90:100-110 */
double dgo_rand_normal(double mean, double stdev, double u1, double u2);
/* eval_normal_prior, src/dang_util_mod.f90:112-121 */
double dgo_eval_normal_prior(double prop, double mean, double std);

/* ---- sky model (src/dang_component_mod.f90:754-813, src/dang_bp_mod.f90:211-243) ---- */
double dgo_a2t(const dgo_ctx *ctx, int band);
/* B_nu (src/dang_component_mod.f90:745-752) and compute_bnu_prime_RJ (src/dang_bp_mod.f90:160-168) */
double dgo_B_nu(double nu, double T);
double dgo_bnu_prime_RJ(double nu);
double dgo_eval_sed(const dgo_ctx *ctx, int comp, int band /*0-based*/, int pix, int map_n /*1-based*/,
                    const double *theta /* NULL or [nindices] */);
double dgo_eval_signal(const dgo_ctx *ctx, int comp, int band, int pix, int map_n, const double *theta);

/* ---- amplitude phase (src/dang_cg_mod.f90) ---- */
/* length of x/b for (group, flag): Sf*npix per sampled diffuse comp; also returns #comps */
int64_t dgo_group_size(const dgo_ctx *ctx, int group, int flag, int *ncg);
void dgo_compute_rhs(const dgo_ctx *ctx, int group, int flag, double *b);
void dgo_compute_Ax(const dgo_ctx *ctx, int group, int flag, const double *x, double *res);
void dgo_compute_sample_vector(const dgo_ctx *ctx, int group, int flag, const double *eta, double *res);
void dgo_initialize_x(const dgo_ctx *ctx, int group, int flag, double *x);
void dgo_unpack_amplitudes(dgo_ctx *ctx, int group, int flag, const double *x);
/* eta(m), m = Sf*npix, from the keyed stream (plane-major like the reference's eta) */
void dgo_draw_eta(const dgo_ctx *ctx, int flag, uint64_t seed, uint64_t stream, double *eta);
/* cg_search: x in/out (warm start); returns iteration counter i as printed by the reference */
int dgo_cg_search(const dgo_ctx *ctx, int group, int flag, const double *b, int ml_mode,
                  const double *eta, double *x, int i_max, double converge,
                  double *delta_trace /* NULL or [i_max+1] */);
/* one (group,flag) of sample_cg_groups: rhs -> cg -> unpack.  Returns CG iterations. */
int dgo_amp_sample_cg(dgo_ctx *ctx, int group, int flag, int ml_mode, uint64_t seed, uint64_t stream,
                      int i_max, double converge, double *x_state /* persistent x or NULL */);
/* Direct per-(pixel,plane) block solve of the same system (restates the GPU algorithm). */
int dgo_amp_sample_direct(dgo_ctx *ctx, int group, int flag, int ml_mode, int fluct_mode,
                          uint64_t seed, uint64_t stream, int64_t *n_not_spd);

/* ---- sky model + chisq (src/dang_data_mod.f90:339-396, 494-526) ---- */
void dgo_update_sky_model(const dgo_ctx *ctx, double *sky /*[nb][nmaps][npix]*/, double *res /*same*/);
double dgo_compute_chisq(const dgo_ctx *ctx, const double *sky, int pol_lo, int pol_hi, double nump,
                         double *chi_map /* NULL or [nmaps][npix] */);

/* ---- index phase (src/dang_lnl_mod.f90, src/dang_sample_mod.f90:88-485 per-pixel branch) ---- */
double dgo_evaluate_lnL(int nbands, int s1, int s2, const double *data, const double *rms,
                        const double *model, int64_t band_stride, int64_t map_stride, int pix, double maskval);
double dgo_evaluate_marginal_lnL(int nbands, int s1, int s2, const double *data, const double *rms,
                                 const double *model, int64_t band_stride, int64_t map_stride, int pix);
/* per-pixel Metropolis sweep of index `nind` (0-based) of component `comp` for map_n (1,2,3,-1).
 * Writes c%indices(:, s1:s2, nind).  Returns number of accepted proposals (diagnostic). */
int64_t dgo_sample_index_mh(dgo_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode,
                            uint64_t seed, uint64_t stream);
/* coarse-Nside sampling (sample_nside < nside), the reference's behaviour reproduced literally; HEALPix pieces restated
 * from the published algorithm (the library itself is not part of the reference tree) */
int64_t dgo_nest2ring(int nside, int64_t ipnest);
void dgo_udgrade(int mode, const double *in, int nside_in, double *out, int nside_out);
int64_t dgo_sample_index_mh_coarse(dgo_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode,
                                   uint64_t seed, uint64_t stream, int nside, int sample_nside);

/* ---- "next" rows: full-sky index mode, step-size tuner, band-gain fit ------------- */
/* Full-sky Metropolis for index `nind` of component `comp` (index_mode==1, src/dang_sample_mod.f90:229-329),
 * including the tuner call when *tuned == 0 (:272-275).  step_size / *tuned are updated like c%step_size /
 * c%tuned.  Draw slots of the keyed stream: pixel = 2^40-1 (a label no real pixel uses), draw = running
 * counter.  Returns accepted proposals of the sampling block. */
int64_t dgo_sample_index_fullsky(dgo_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode,
                                 uint64_t seed, uint64_t stream, int *tuned);
/* tune_spectral_parameter_length (src/dang_sample_mod.f90:623-717) on prepared full-sky data;
 * exposed through dgo_sample_index_fullsky. */
/* fit_band_gain(ddata, map_n=1, band) (src/dang_sample_mod.f90:570-621): returns the new gain; sky/res are
 * the arrays update_sky_model left (dgo_update_sky_model). */
int64_t dgo_sample_index_fullsky_coarse(dgo_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                                        uint64_t stream, int *tuned, int nside, int sample_nside);
/* unit conversions and bandpass normalisation, src/dang_bp_mod.f90:62-81, 160-274; convert_maps src/dang_data_mod.f90:429-463 */
double dgo_bnu_prime(double nu, double T_CMB);
double dgo_a2f(const dgo_ctx *ctx, int band);
double dgo_f2t(const dgo_ctx *ctx, int band);
void dgo_normalize_bandpass(const double *tau_in, int n, double *tau_out);
int dgo_convert_maps(dgo_ctx *ctx, const int *unit, const int *cg_map, double *conversion);
/* step-size tuning of the per-pixel branch (src/dang_sample_mod.f90:341-346); *tuned in/out, c->step_size updated */
void dgo_tune_perpixel(dgo_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                       uint64_t stream, int *tuned);
double dgo_fit_band_gain(const dgo_ctx *ctx, const double *sky, const double *res, int band, int ml_mode,
                         uint64_t seed, uint64_t stream);

#ifdef __cplusplus
}
#endif
#endif
