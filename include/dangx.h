/*
 * dangx.h -- C ABI of libdangx.so, the MI355X (gfx950) implementation of dang's
 * Gibbs inner loop.  Plain pointers and sizes only; no C++/torch types.
 *
 * The reference (hermda02/dang, Fortran 90) has no FFI layer; the seam is its
 * module API.  Each entry point below names the reference procedure whose work it
 * replaces.  A Fortran driver binds these through ISO_C_BINDING
 * (fortran/dangx_mod.f90); INTEGRATION.md shows the reference-side wrapper.
 *
 * Array layouts are the Fortran arrays as they sit in memory (no transposition):
 *   sig_map/rms_map (0:npix-1, nmaps, nbands)  == C [band][map][pix]
 *   masks           (0:npix-1, nmaps)          == C [map][pix]   (plane 1 is tested,
 *                                                 as everywhere in the reference)
 *   c%amplitude     (0:npix-1, nmaps)          == C [map][pix]
 *   c%indices       (0:npix-1, nmaps, nind)    == C [ind][map][pix]
 * All reals are IEEE fp64 (real(dp)); map numbers are 1-based (1=T,2=Q,3=U).
 *
 * A context holds ONE pixel shard [pix0, pix0+npix) of the sky on ONE device.
 * Every per-pixel operation is shard-local; the random streams are keyed by the
 * GLOBAL pixel index, so results do not depend on how the sky is sharded.
 * Global reductions (chi^2) are returned as un-normalised local sums for the
 * caller to all-reduce.
 *
 * Return value: 0 on success, non-zero on error (message: dangx_last_error).
 * The reference's convention is print + stop (e.g. src/dang_cg_mod.f90:97-101);
 * the Fortran wrapper reproduces that from the status code.
 */
#ifndef DANGX_H
#define DANGX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DANGX_MAX_BANDS 32
#define DANGX_MAX_COMPS 16
#define DANGX_MAX_IND 2
#define DANGX_MAX_GROUP 8

/* component type == c%type string, src/dang_component_mod.f90:791-809 */
/* DANGX_TCMB: 'T_cmb' -- evaluate_T_cmb (:815-848); its eval_signal is the bare sed (:770-771).  Supported by
 * dangx_eval_sed, wherever a component is REMOVED from the data / summed into the sky model, as an index-sampled
 * component (per pixel or full sky), and as an amplitude-sampled member of a CG group: there the reference treats it as
 * a diffuse member whose mixing element is eval_sed (src/dang_cg_mod.f90:469, 691, 807 test only for template / hi_fit /
 * monopole), i.e. it solves for a c%amplitude that eval_signal then ignores -- reproduced as it is. */
enum { DANGX_POWERLAW = 1, DANGX_MBB = 2, DANGX_FREEFREE = 3, DANGX_LOGNORMAL = 4, DANGX_CMB = 5, DANGX_TCMB = 6,
       /* global-amplitude types: one amplitude per fitted band instead of one per pixel (c%template,
        * c%template_amplitudes, c%corr, c%nfit; src/dang_component_mod.f90:536-710).  A CG group that contains
        * them is a coupled system: DANGX_SOLVER_CG runs the reference's iteration on the device, DANGX_SOLVER_DIRECT
        * eliminates the per-pixel blocks and solves the Schur system of the (<= 32) global rows exactly. */
       DANGX_TEMPLATE = 7, DANGX_MONOPOLE = 8, DANGX_HIFIT = 9 };
/* c%lnl_type / c%prior_type strings, src/dang_sample_mod.f90:383-400 */
enum { DANGX_LNL_CHISQ = 1, DANGX_LNL_MARGINAL = 2, DANGX_LNL_PRIOR = 3 };
enum { DANGX_PRIOR_GAUSSIAN = 1, DANGX_PRIOR_UNIFORM = 2, DANGX_PRIOR_JEFFREYS = 3 };
/* ml_mode string, src/dang_cg_mod.f90:254-267 */
enum { DANGX_ML_SAMPLE = 1, DANGX_ML_OPTIMIZE = 2 };
/* poltype bit flags, src/dang_util_mod.f90:228-292 */
enum { DANGX_FLAG_T = 1, DANGX_FLAG_Q = 2, DANGX_FLAG_U = 4, DANGX_FLAG_QU = 8 };
/* amplitude solver: DIRECT = per-(pixel,plane) Cholesky block solve (MI355X path);
 * CG = the reference's global conjugate-gradient iteration run on the device. */
enum { DANGX_SOLVER_DIRECT = 0, DANGX_SOLVER_CG = 1 };
/* fluctuation term: REFERENCE reproduces compute_sample_vector exactly, including
 * its two quirks (src/dang_cg_mod.f90:1008-1015 one eta for all bands; :1033-1040 no
 * component offset); CORRECT draws the textbook sum_nu T^t N^-1/2 eta_nu. */
enum { DANGX_FLUCT_CORRECT = 0, DANGX_FLUCT_REFERENCE = 1 };

/* kernel ids for dangx_profile_get / dangx_profile_get_planes */
enum {
    DANGX_K_AMP_DIRECT = 0, DANGX_K_INDEX_MH = 1, DANGX_K_SKY_CHISQ = 2, DANGX_K_REDUCE = 3,
    DANGX_K_CG_AX = 4, DANGX_K_CG_VEC = 5, DANGX_K_AMP_INDEX = 6, DANGX_K_COUNT = 8
};

typedef struct dangx_ctx dangx_ctx;

typedef struct {
    int32_t npix;         /* pixels in this shard */
    int32_t nmaps;        /* 1 or 3 (global nmaps, src/dang.f90:49-63) */
    int32_t nbands;       /* global nbands */
    int32_t ncomp;        /* global ncomp */
    int64_t pix0;         /* global index of this shard's first pixel */
    int64_t npix_global;  /* 12*nside^2 */
    int32_t device;       /* HIP device ordinal, -1 = current device */
    int32_t reserved;
} dangx_dims;

/* mirrors the fields of type(dang_comps) the hot path reads, src/dang_component_mod.f90:12-48 */
typedef struct {
    int32_t type;              /* DANGX_POWERLAW ... */
    int32_t is_synch;          /* trim(c%label)=='synch' (jeffreys prior, src/dang_lnl_mod.f90:289) */
    int32_t nindices;
    int32_t cg_group;          /* c%cg_group (1-based as in the parameter file) */
    int32_t sample_amplitude;  /* c%sample_amplitude */
    int32_t reserved;
    double nu_ref;             /* Hz */
    int32_t lnl_type[DANGX_MAX_IND];
    int32_t prior_type[DANGX_MAX_IND];
    double gauss_prior[DANGX_MAX_IND][2]; /* mean, std */
    double uni_prior[DANGX_MAX_IND][2];   /* low, high */
    double step_size[DANGX_MAX_IND];
} dangx_comp_desc;

/* ---- lifetime ------------------------------------------------------------ */
int dangx_create(dangx_ctx **ctx, const dangx_dims *dims);
int dangx_destroy(dangx_ctx *ctx);
const char *dangx_last_error(const dangx_ctx *ctx);
const char *dangx_version(void);
/* run all work of this context on the given hipStream_t (NULL = default stream) */
int dangx_set_stream(dangx_ctx *ctx, void *hip_stream);
/* Pixel-sharded runs (one context per rank, dangx_dims.pix0/npix = the rank's RING range): the few places where the
 * reference sums over the WHOLE sky inside a call -- the dot products of cg_search (src/dang_cg_mod.f90:279-312) and
 * the global-amplitude rows of compute_rhs / compute_Ax / compute_sample_vector (:522-587, :833-893, :1045-1096) --
 * hand their local sums (host doubles, n <= 32*32+96; the coarse-Nside sweeps of a pixel shard: their degrade buffers) to this callback, which must replace buf[0..n) by its sum over
 * all ranks and return 0 (an MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_DOUBLE, MPI_SUM) or a torch.distributed
 * all_reduce).  is_root != 0 on exactly one rank: the replicated global rows are counted there in dot products.
 * fn == NULL (the default) = single rank.  Every rank must make the same sequence of calls. */
typedef int (*dangx_allreduce_fn)(void *user, double *buf, int64_t n);
int dangx_set_allreduce(dangx_ctx *ctx, dangx_allreduce_fn fn, void *user, int is_root);
int dangx_synchronize(dangx_ctx *ctx);
/* number of HIP devices visible to the process (a single-process driver creates one context per device) */
int dangx_device_count(int *n);
/* Host map arrays of the calls below (dangx_upload_data, dangx_put / get_amplitude / indices, dangx_set_template, the
 * sky / res / chi_map outputs of dangx_sky_model_chisq) are by default packed per shard, [planes][npix].  A driver
 * that keeps FULL-SKY arrays -- the reference's sig_map(0:npix-1,nmaps,nbands) etc. -- and runs several pixel-shard
 * contexts over them (one per GPU) passes, for each context, the address of the shard's FIRST pixel in plane 1 and sets
 * plane_stride = the full-sky pixel count here: plane q of the shard then starts plane_stride doubles after plane
 * q-1.  plane_stride = 0 returns to the packed layout. */
int dangx_set_host_stride(dangx_ctx *ctx, int64_t plane_stride);

/* ---- static description (once, after src/dang.f90:73) ------------------------ */
/* bp(j): src/dang_bp_mod.f90:7-15,31-57.  n = 0 for a 'delta' bandpass, else n
 * samples nu0[] (Hz) with normalised weights tau0[].  band is 0-based. */
int dangx_set_band(dangx_ctx *ctx, int band, double nu_c_hz, int n, const double *nu0, const double *tau0);
int dangx_set_component(dangx_ctx *ctx, int comp /*0-based*/, const dangx_comp_desc *desc);
/* global T_CMB used by the 'cmb' SED through a2t (src/dang_util_mod.f90:15) */
int dangx_set_tcmb(dangx_ctx *ctx, double T_cmb);
/* ddata%gain(:), ddata%offset(:) (src/dang_data_mod.f90:31-32) */
int dangx_set_calibration(dangx_ctx *ctx, const double *gain, const double *offset);

/* Unit conversions of a band (src/dang_bp_mod.f90): a2t [uK_cmb/uK_RJ] :211-243, a2f [MJy sr-1/uK_RJ] :181-209 (its
 * single-precision `1e14` literal included), f2t [uK_cmb/(MJy sr-1)] :245-274, evaluated on the host from the band set
 * with dangx_set_band (delta: at nu_c; otherwise the tau0-weighted sum over the samples) and the current T_CMB. */
enum { DANGX_A2T = 0, DANGX_A2F = 1, DANGX_F2T = 2 };
int dangx_unit_conversion(dangx_ctx *ctx, int band, int which, double *out);
/* normalize_bandpass, src/dang_bp_mod.f90:62-81: tau_out = tau_in / sum(tau_in) (what init_bp_mod applies to tau0
 * before the hot path sees it); no context needed */
int dangx_normalize_bandpass(const double *tau_in, int n, double *tau_out);
/* convert_maps, src/dang_data_mod.f90:429-463, on the RESIDENT maps: for every band with cg_map[j] == 0 (cg_map may be
 * NULL = none swapped) conversion[j] = 1 (uK_RJ), 1/a2t (uK_cmb) or 1/a2f (MJy/sr); sig_map(:,:,j), rms_map(:,:,j) and
 * offset(j) are multiplied by it and the offsets are copied into template_amplitudes(:,1) of every monopole.
 * conversion[nbands] is an output (ddata%conversion).  Call after dangx_upload_data / dangx_adopt_device_data (adopted
 * buffers are scaled in place). */
enum { DANGX_UNIT_UK_RJ = 0, DANGX_UNIT_UK_CMB = 1, DANGX_UNIT_MJY_SR = 2 };
int dangx_convert_maps(dangx_ctx *ctx, const int32_t *unit, const int32_t *cg_map, double *conversion);

/* ---- map data ------------------------------------------------------------------ */
/* ddata%sig_map, rms_map, masks: host pointers, copied to HBM */
int dangx_upload_data(dangx_ctx *ctx, const double *sig, const double *rms, const double *mask);
/* same arrays already resident in HBM (borrowed, not copied, not freed) */
int dangx_adopt_device_data(dangx_ctx *ctx, const double *sig_dev, const double *rms_dev, const double *mask_dev);
/* c%amplitude(:, :) and c%indices(:, :, :) of component comp; host pointers */
int dangx_put_amplitude(dangx_ctx *ctx, int comp, const double *amp);
int dangx_get_amplitude(dangx_ctx *ctx, int comp, double *amp);
int dangx_put_indices(dangx_ctx *ctx, int comp, const double *ind);
int dangx_get_indices(dangx_ctx *ctx, int comp, double *ind);
/* global-amplitude components: c%template ([map][pix], host pointer, copied), c%corr(j) (1 = band j is fitted),
 * c%nfit; and c%template_amplitudes(band, map) exchanged as [map][band] */
int dangx_set_template(dangx_ctx *ctx, int comp, const double *tmpl, const int32_t *corr, int nfit);
int dangx_put_template_amplitudes(dangx_ctx *ctx, int comp, const double *ta);
int dangx_get_template_amplitudes(dangx_ctx *ctx, int comp, double *ta);
/* let caller-owned HBM buffers BE the resident amplitude / index maps of a component
 * (borrowed, not freed; idx_dev may be NULL when the component has no indices) */
int dangx_adopt_device_state(dangx_ctx *ctx, int comp, double *amp_dev, double *idx_dev);
/* device addresses of the resident copies ([map][pix] and [ind][map][pix]) */
void *dangx_amplitude_devptr(dangx_ctx *ctx, int comp);
void *dangx_indices_devptr(dangx_ctx *ctx, int comp);

/* ---- amplitude phase: one (group, flag) pass of sample_cg_groups ---------------
 * replaces compute_rhs + cg_search + unpack_amplitudes (src/dang_cg_mod.f90:166-171).
 * flag is one poltype bit (T, Q, U or Q+U).  seed/stream select the keyed random
 * stream (the reference uses an unseeded RANDOM_NUMBER, src/dang.f90:67).
 * cg_iters (nullable): CG iteration counter as printed by the reference (CG solver); 0 for DIRECT, or -k when the
 * DIRECT solve of a group with template / monopole / hi_fit members found k directions of the global amplitudes
 * that the diffuse members absorb completely (those keep their current value, as they do under the reference's CG).
 * n_not_spd (nullable): units whose block was not SPD (left unchanged). */
int dangx_amp_sample(dangx_ctx *ctx, int group, int flag, int ml_mode, int solver, int fluct_mode,
                     uint64_t seed, uint64_t stream, int i_max, double converge,
                     int *cg_iters, int64_t *n_not_spd);

/* Accuracy of the last DANGX_SOLVER_DIRECT solve of a group with template / monopole / hi_fit members, measured
 * directly at the new state after the last step of iterative refinement (b, A: the system of compute_rhs /
 * compute_sample_vector / compute_Ax, src/dang_cg_mod.f90:326-1096): rel_residual[0] = the largest |b - A x| over the
 * global rows relative to that row of b; rel_residual[1] = the same relative to the size of the row's terms
 * (sum |b terms| + |A x terms|: rounding alone leaves ~1e-16 sqrt(npix) there, and weakly constrained amplitudes are
 * large, so [0] can sit well above [1]); refinements = steps taken (0: the first solution already met 1e-12).
 * A well-conditioned system of `template` members (every pivot of the equilibrated Schur matrix >= 1e-3) is not measured: both numbers are then the
 * a-priori bound 16 R eps / (smallest pivot) and refinements = 0; the environment variable DANGX_SCHUR_CHECK=1 measures always. */
int dangx_schur_info(dangx_ctx *ctx, double *rel_residual /*[2]*/, int *refinements);
/* Residual of the reference's linear system at the CURRENT amplitudes, through the reference's own operators
 * (dangx_compute_rhs + dangx_compute_sample_vector(eta(seed, stream)) - dangx_compute_Ax(x), vectors kept on the device):
 * out[0] = |b - A x|_2 / |b|_2 over the rows of unmasked units and the global rows, out[1] = the largest global-row
 * residual relative to that row of b (0 without global members).  What cg_search's delta_new measures
 * (src/dang_cg_mod.f90:285, 303); call it with the seed / stream / ml_mode of the solve it checks. */
int dangx_amp_residual(dangx_ctx *ctx, int group, int flag, int ml_mode, uint64_t seed, uint64_t stream, double *out);

/* ---- index phase: sample_index_mh, per-pixel branch (src/dang_sample_mod.f90:88-485,
 * index_mode==2, sample_nside==nside).  nind 0-based; map_n = 1,2,3 or -1 (Q+U).
 * accepted (nullable): number of accepted proposals over the shard. */
int dangx_index_sample(dangx_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode,
                       uint64_t seed, uint64_t stream, int64_t *accepted);

/* ---- two consecutive indices of ONE component on the same planes, in one call: exactly
 * dangx_index_sample(comp, nind, map_n, nsample, ml_mode, seed, stream_first, accepted_first) followed by
 * dangx_index_sample(comp, nind + 1, map_n, nsample, ml_mode, seed, stream_second, accepted_second) -- consecutive passes of
 * the loop over a component's indices in sample_spectral_parameters (src/dang_sample_mod.f90:40-75; dust beta, then dust T)
 * -- and bit for bit their result.  Nothing the second sweep removes from the data has changed in between, so where the
 * register chain covers both indices they run in ONE launch on one staging of the maps. */
int dangx_index_sample_pair(dangx_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                            uint64_t stream_first, uint64_t stream_second, int64_t *accepted_first, int64_t *accepted_second);

/* ---- the amplitude solve of a CG group and the FIRST index sweep on the same planes, in one call: exactly
 * dangx_amp_sample(group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, i_max, converge, cg_iters, n_not_spd) followed by
 * dangx_index_sample(comp, nind, map_n, nsample, ml_mode, seed_index, stream_index, accepted) -- the way sample_cg_groups
 * (src/dang_cg_mod.f90:142-177) and sample_spectral_parameters (src/dang_sample_mod.f90:21-86) follow each other plane set by
 * plane set in the main loop (src/dang.f90) -- and bit for bit their result.  When every step is independent per pixel
 * (delta bands, diffuse members only, direct solver, reference fluctuation term, chisq likelihood with a gaussian / uniform
 * prior, the sampled component a member of the group whose other members are the only other components on these planes)
 * both run in ONE kernel launch: the maps are read once and the solve's arithmetic hides under the chain's. */
int dangx_amp_index_sample(dangx_ctx *ctx, int group, int flag, int ml_mode, int solver, int fluct_mode, uint64_t seed_amp,
                           uint64_t stream_amp, int i_max, double converge, int comp, int nind, int map_n, int nsample,
                           uint64_t seed_index, uint64_t stream_index, int *cg_iters, int64_t *n_not_spd, int64_t *accepted);

/* ---- everything one iteration does on ONE plane set of a CG group, in one call: exactly
 * dangx_amp_sample(group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, i_max, converge, cg_iters, n_not_spd) followed by
 * dangx_index_sample(comp[s], nind[s], map_n(flag), nsample, ml_mode, seed_index, stream[s], &accepted[s]) for s = 0 .. nsweeps-1 --
 * the solve of sample_cg_groups (src/dang_cg_mod.f90:166-171) and the passes of sample_spectral_parameters on these planes
 * (src/dang_sample_mod.f90:40-75) in the reference's order (the caller passes them in that order; dangx_plan_fusion says when the
 * grouping leaves the main loop's result unchanged).  For models with many bands and members whose swept components are all
 * members of the group (C5) this is ONE kernel launch that keeps the members' SED columns in LDS across the sweeps instead of
 * evaluating every other member's SED again in every sweep; every other case IS the calls above (through dangx_amp_index_sample
 * and dangx_index_sample_pair where they apply).  accepted[nsweeps] nullable. */
int dangx_plane_set_sample(dangx_ctx *ctx, int group, int flag, int ml_mode, int solver, int fluct_mode, uint64_t seed_amp,
                           uint64_t stream_amp, int i_max, double converge, int nsweeps, const int32_t *comp, const int32_t *nind,
                           const uint64_t *stream, int nsample, uint64_t seed_index, int *cg_iters, int64_t *n_not_spd,
                           int64_t *accepted);

/* ---- the index phase of ONE plane set in one call: exactly
 * dangx_index_sample(comp[s], nind[s], map_n(flag), nsample, ml_mode, seed, stream[s], &accepted[s]) for s = 0 .. nsweeps-1 -- the
 * passes of sample_spectral_parameters (src/dang_sample_mod.f90:40-75) that touch the planes of `flag`, in the reference's order:
 * what the two-call seam (src/dang.f90:101, 106) issues after sample_cg_groups has returned.  Sweeps on disjoint planes are
 * independent (c%indices(:, k, :) is per plane), so a caller may collect the sweeps of a plane set from the reference's
 * component-major loop.  Where every swept component is an amplitude-sampled member of ONE CG group whose members are the only
 * components on these planes (delta bands, chisq likelihood, gaussian / uniform priors) the sweeps are ONE kernel launch on the
 * amplitudes in memory: one staging of the maps, the residual kept in registers between the sweeps; every other case IS the
 * calls above (consecutive indices of a component through dangx_index_sample_pair).  accepted[nsweeps] nullable. */
int dangx_plane_sweeps_sample(dangx_ctx *ctx, int flag, int nsweeps, const int32_t *comp, const int32_t *nind, const uint64_t *stream,
                              int nsample, int ml_mode, uint64_t seed, int64_t *accepted);

/* ---- sky model + chi^2: update_sky_model + compute_chisq
 * (src/dang_data_mod.f90:339-396, 494-526).  pol_lo..pol_hi = ddata%pol_type range.
 * chisq_sum receives the LOCAL sum over unmasked pixels and planes of
 * sum_j res^2/rms^2 (not divided by nbands or nump); the reference's chisq is
 * allreduce(chisq_sum)/nbands/nump.  sky/res ([band][map][pix]) and chi_map
 * ([map][pix], already divided by nbands as in the reference) are optional host outputs. */
int dangx_sky_model_chisq(dangx_ctx *ctx, int pol_lo, int pol_hi, double *chisq_sum,
                          double *sky, double *res, double *chi_map);
/* asynchronous form: writes the local sum to a device double (for an RCCL all-reduce) */
int dangx_sky_model_chisq_dev(dangx_ctx *ctx, int pol_lo, int pol_hi, double *chisq_sum_dev);

/* chi^2 sums that the index sweeps compute as a by-product (no extra pass over the maps):
 * which = 0: the state the amplitude phase left (captured by the first sweep on each plane since its last
 * amplitude update), which = 1: the current state.  Same un-normalised local sum as above.  Returns 2 when
 * a plane in pol_lo..pol_hi was not covered by a sweep -- fall back to dangx_sky_model_chisq. */
int dangx_chisq_cached(dangx_ctx *ctx, int which, int pol_lo, int pol_hi, double *chisq_sum);
int dangx_chisq_cached_dev(dangx_ctx *ctx, int which, int pol_lo, int pol_hi, double *chisq_sum_dev);

/* the same number for the CURRENT state at the least cost: cached sums where the last sweeps left them, one explicit pass over
 * each remaining plane (whose sum is then cached too).  What gpu_chisq of the reference-side wrapper calls after every CG group
 * and every phase (write_stats_to_term, src/dang_data_mod.f90:528-570). */
int dangx_chisq_current(dangx_ctx *ctx, int pol_lo, int pol_hi, double *chisq_sum);

/* ---- index phase with sample_nside /= nside (src/dang_sample_mod.f90:199-217, 332-483) --------------------------
 * One whole-sky context (npix = 12*nside^2, pix0 = 0), or a pixel shard with an all-reduce (see below).  As in the reference: the data minus every other component is
 * formed at full resolution, then degraded with HEALPix's udgrade_ring (RING -> NEST, mean of the good children,
 * NEST -> RING), the rms with dang's udgrade_rms (sqrt(mean(rms^2)) * nside_out/nside_in), the mask with udgrade_mask
 * (mean >= 0.5); one chain per COARSE pixel i; the coarse index map is upgraded (children take the parent's value) and
 * written to c%indices(:, s1:s2, nind) for EVERY pixel.  Also as in the reference, the chain of coarse pixel i reads
 * the FULL-resolution arrays at the same index i (ddata%masks(i,1), c%indices(i,...), and eval_signal's
 * c%amplitude(i,k): src/dang_sample_mod.f90:362, 372-377, 548-553): reproduced literally, see DESIGN.md section 7.
 * HEALPix itself is absent from the reference tree (an external library): udgrade_ring / nest2ring are restated from
 * the published algorithm (Gorski et al. 2005, ApJ 622, 759; HEALPix 3.x pix_tools / udgrade_nr).
 * Likelihoods: chisq / marginal / prior; priors: gaussian / uniform / jeffreys; diffuse component types and T_cmb. */
int dangx_index_sample_coarse(dangx_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode,
                              uint64_t seed, uint64_t stream, int nside, int sample_nside, int64_t *accepted);
/* On a PIXEL SHARD (pix0 /= 0 or npix /= 12*nside^2) dangx_index_sample_coarse needs sums over all shards -- the
 * children of a coarse pixel are scattered over the RING ranges -- and takes them through the dangx_set_allreduce callback
 * (one process per GPU; the buffers below, i.e. up to 2*(2*Sp*nbands+1)*12*sample_nside^2 doubles, not the few of the
 * amplitude solves).  A single process that drives several contexts calls the three phases itself and ADDS the buffers
 * of all contexts between them, in shard order:
 *   dangx_coarse_sizes     : lengths of the two buffers (n_partials; n_index = 12*sample_nside^2 + 1)
 *   dangx_coarse_partials  : A -- per coarse pixel and plane the sum of this shard's good children and their number, for the
 *                            cleaned data, rms^2 and the mask -> buf[n_partials]
 *   dangx_coarse_chains    : B -- finish data / rms / mask from the SUMMED partials; run the chains of the coarse pixels i
 *                            whose full-resolution pixel i this shard holds -> index_out[0..npc) (0 elsewhere),
 *                            index_out[npc] = accepted proposals
 *   dangx_coarse_writeback : C -- c%indices(:, s1:s2, nind) of this shard's pixels from the SUMMED coarse index map
 * With one shard the result equals the whole-sky call bit for bit; with several the child sums are associated
 * differently (each shard's children first), i.e. the degraded maps agree to rounding. */
int dangx_coarse_sizes(dangx_ctx *ctx, int map_n, int sample_nside, int64_t *n_partials, int64_t *n_index);
int dangx_coarse_partials(dangx_ctx *ctx, int comp, int map_n, int nside, int sample_nside, double *buf);
int dangx_coarse_chains(dangx_ctx *ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed, uint64_t stream,
                        int nside, int sample_nside, const double *partials_sum, double *index_out);
int dangx_coarse_writeback(dangx_ctx *ctx, int comp, int nind, int map_n, int nside, int sample_nside, const double *index_sum);
/* the degrade / upgrade primitives on their own (whole-sky context): mode 0 = udgrade_ring, 1 = udgrade_rms,
 * 2 = udgrade_mask(threshold 0.5); host pointers, one map each ([12*nside_in^2] -> [12*nside_out^2]) */
int dangx_udgrade(dangx_ctx *ctx, int mode, const double *map_in, int nside_in, double *map_out, int nside_out);

/* ---- per-iteration trace output without pulling maps: mask_avg(c%indices(:,map_n,nind), masks) as printed by
 * write_data every Gibbs iteration (src/dang_data_mod.f90:716-731, src/dang_util_mod.f90:186-206).  Returns this
 * shard's sum over unmasked pixels and their number; mask_avg = (all-reduced sum) / (all-reduced count). */
int dangx_index_masked_sum(dangx_ctx *ctx, int comp, int nind, int map_n, double *sum, int64_t *count);

/* n <= 16 such sums in one launch and one wait: (comp[e], nind[e], map_n[e]) -> sums[e], counts[e] */
int dangx_index_masked_sums(dangx_ctx *ctx, int n, const int32_t *comp, const int32_t *nind, const int32_t *map_n, double *sums,
                            int64_t *counts);

/* Step-size tuning in the per-pixel branch (src/dang_sample_mod.f90:341-346) starts the tuner's sky-wide chain at
 * sample(l) = sum(c%indices(:,map_inds(1),l)) / sum(mask(:,1)): both sums run over EVERY pixel and the mask VALUES are
 * summed.  Returns this shard's two sums; the chain itself is dangx_fullsky_prepare / dangx_fullsky_sums (below). */
int dangx_index_plain_sum(dangx_ctx *ctx, int comp, int nind, int map_n, double *sum_index, double *sum_mask);

/* ---- full-sky index mode (index_mode == 1, src/dang_sample_mod.f90:229-329), tune_spectral_parameter_length
 * (:623-717) and fit_band_gain (:570-621).  With one index for the whole sky every Metropolis step is one pass
 * that yields a few global sums; the chain (proposal, prior, accept, step-size tuning) stays with the caller,
 * between its all-reduces (dang_amd/api.py: sample_index_mh_fullsky, tune_spectral_parameter_length,
 * fit_band_gain).
 *   dangx_fullsky_prepare: data_raw minus every other component for the planes of map_n (:173-196), kept in HBM.
 *   dangx_fullsky_sums   : LOCAL sums at theta[2]: what = 0 evaluate_lnL (1 value); 1 evaluate_marginal_lnL
 *                          (2*nbands*Sp values: TNd(j,k), TNT(j,k) interleaved, j outer / k inner; the caller forms
 *                          sum -1/2 TNd^2/TNT after the all-reduce); 2 the jeffreys-prior sum (1 value).
 *                          3 the chisq likelihood's sufficient statistics about theta (3*nbands*Sp values: W0, U, V of
 *                          (band j, plane k) at 3*(j*Sp + k): sum r0^2, sum r0 a/sigma, sum a^2/sigma^2 with r0 = (d - a s_j(theta))/sigma).
 *   dangx_fill_index     : c%indices(:, s1:s2, nind) = value for every pixel (:329, :483).
 *   dangx_gain_sums      : out[0] = sum map2*N_inv*map1, out[1] = sum map1*N_inv*map1 of fit_band_gain (:606-607). */
int dangx_fullsky_prepare(dangx_ctx *ctx, int comp, int map_n);
/* the same mode with c%sample_nside(nind) /= nside (:199-217): the cleaned data, the rms and the mask are degraded first
 * (udgrade_ring / udgrade_rms / udgrade_mask), dangx_fullsky_sums then runs over the 12*sample_nside^2 coarse pixels --
 * with eval_signal's c%amplitude read from the full-resolution array at the coarse pixel number, as in the reference
 * (see dangx_index_sample_coarse).  One whole-sky context. */
int dangx_fullsky_prepare_coarse(dangx_ctx *ctx, int comp, int map_n, int nside, int sample_nside);
int dangx_fullsky_sums(dangx_ctx *ctx, int what, const double *theta, double *out, int nout);
int dangx_fill_index(dangx_ctx *ctx, int comp, int nind, int map_n, double value);
int dangx_gain_sums(dangx_ctx *ctx, int band, double *out);
/* c%indices(pix, map_n, :) of one local pixel (the full-sky chain starts from pixel 0, :240-242) */
int dangx_peek_indices(dangx_ctx *ctx, int comp, int map_n, long long pix, double *out);

/* dangx_fullsky_prepare_coarse for a PIXEL SHARD: dangx_coarse_partials of every shard (map_n's planes), the buffers
 * added over the shards, then this call on every shard with the sum.  The degraded data / rms / mask are then whole-sky
 * on each shard, and dangx_fullsky_sums adds, per shard, the coarse pixels i whose full-resolution pixel i it holds
 * (that is where the reference reads c%amplitude(i), src/dang_sample_mod.f90:548-563).  dangx_fullsky_sample does this. */
int dangx_fullsky_finish_coarse(dangx_ctx *ctx, int comp, int map_n, int nside, int sample_nside, const double *partials_sum);

/* ---- the SKY-WIDE steps of the Gibbs loop, chain included (dang_amd/csrc/dangx_sky.hip) --------------------------------
 * One number describes the whole sky in these steps: a Metropolis step is a pass over the maps that leaves a few sums, plus
 * a few scalar operations -- or, for the chisq likelihood of a diffuse component (whose model of a band is amplitude(pixel) x
 * one SED value), ONE pass per sweep that leaves the sufficient statistics W0, U, V per (band, plane) about the starting point:
 * -2 lnL(theta) = sum [W0 - 2 ds U + ds^2 V], every proposal then costs nbands SED evaluations on the host (dangx_fullsky_sums
 * selector 3; the same chain to rounding, DANGX_FULLSKY_STATS=0 for a pass per proposal).  Each entry point runs the whole chain of the reference procedure it names, so the chain exists
 * once for every host language.  ctxs[0..nctx) are the contexts of THIS process in shard order (nctx = 1: a whole-sky
 * context, or one context per process); a sky-wide sum is the contexts' sums added in shard order, then summed over the
 * ranks through ctxs[0]'s dangx_set_allreduce callback.  Random numbers: Philox4x32-10 keyed by (seed, stream, pixel label
 * 2^40 - 1, running draw counter) -- the same on every rank.  Errors: dangx_last_error(ctxs[0]).
 *
 * dangx_fullsky_sample: sample_index_mh with index_mode == 1 (src/dang_sample_mod.f90:229-329) for index nind (0-based) of
 *   component comp on map_n (1, 2, 3 or -1 = Q+U): data_raw minus the other components (:173-196; degraded when
 *   sample_nside /= nside, :199-217; 0 or nside = full resolution), start at c%indices(0, map_inds(1), :) (:240-242), the
 *   tuner when tuned[nind] == 0 (:272-275), nsample steps (:282-324), c%indices(:, s1:s2, nind) = result (:329, :483).
 *   tuned[nindices] in/out = c%tuned (NULL: all tuned); step_size out (nullable) = c%step_size(nind) after the call (the
 *   contexts' descriptors are updated); value out (nullable) = the sampled index; accepted out (nullable).
 * dangx_tune_step_size: tune_spectral_parameter_length (:623-717) on data prepared by dangx_fullsky_prepare[_coarse] on every
 *   context; theta_init[2]; *draw = running draw counter of the stream (in/out).  +-50 % until the acceptance over nsample
 *   steps is within [0.4, 0.6]; then c%tuned = .true. for ALL indices (:712).  Bounded at 64 rounds (the reference's
 *   `do while (.not. c%tuned(nind))` does not return when nothing is accepted any more).
 * dangx_tune_perpixel: the 'Tuning!' block of the per-pixel branch (:337-346): one tuner pass per index l of the component,
 *   starting at sample(l) = sum(c%indices(:,map_inds(1),l)) / sum(mask(:,1)) over EVERY pixel (mask VALUES summed).
 * dangx_fit_band_gain: fit_band_gain(ddata, 1, band) (:570-621), band 0-based: gain = mu (optimize) or mu + sigma*N(0,1)
 *   (draw slot = band); stored as ddata%gain(band) on every context (:619) and returned.
 * dangx_update_tcmb: "Update the global variable T_CMB" (:75-78): T_CMB = c%indices(0,1,1) of a 'T_cmb' component, set on
 *   every context (it enters a2t of the 'cmb' SED) and returned. */
int dangx_fullsky_sample(dangx_ctx *const *ctxs, int nctx, int comp, int nind, int map_n, int nsample, int ml_mode,
                         uint64_t seed, uint64_t stream, int nside, int sample_nside, int32_t *tuned, double *step_size,
                         double *value, int64_t *accepted);
int dangx_tune_step_size(dangx_ctx *const *ctxs, int nctx, int comp, int nind, int nsample, int ml_mode, uint64_t seed,
                         uint64_t stream, const double *theta_init, uint32_t *draw, int32_t *tuned, double *step_size);
int dangx_tune_perpixel(dangx_ctx *const *ctxs, int nctx, int comp, int nind, int map_n, int nsample, int ml_mode,
                        uint64_t seed, uint64_t stream, int32_t *tuned, double *step_size);
int dangx_fit_band_gain(dangx_ctx *const *ctxs, int nctx, int band, int ml_mode, uint64_t seed, uint64_t stream, double *gain);
int dangx_update_tcmb(dangx_ctx *const *ctxs, int nctx, int comp, double *tcmb);

/* ---- which solves may be issued together with the first index sweep on their planes (dangx_amp_index_sample) without changing
 * the result of the main loop (src/dang.f90:101-106 runs every solve of sample_cg_groups before any sweep of
 * sample_spectral_parameters).  pairs: the (group, flag) passes of the sampled CG groups in sample_cg_groups' order; sweeps: the
 * (component, index, flag) sweeps in sample_spectral_parameters' order (component-major), sweep_plain = 1 for an ordinary
 * per-pixel sweep at the map resolution with a tuned step.  first_sweep[npairs] out: position in the sweep list of the sweep
 * to issue with that solve, or -1.  Pure host logic on the context's descriptors; the rule is stated at its definition
 * (dang_amd/csrc/dangx_sky.hip). */
int dangx_plan_fusion(dangx_ctx *ctx, int npairs, const int32_t *pair_group, const int32_t *pair_flag, int nsweeps,
                      const int32_t *sweep_comp, const int32_t *sweep_nind, const int32_t *sweep_flag, const int32_t *sweep_plain,
                      int solver, int32_t *first_sweep);

/* ---- one (group, flag) pass of sample_cg_groups over SEVERAL contexts of one process (src/dang_cg_mod.f90:166-171).
 * Diffuse groups: dangx_amp_sample on every context (enqueued on all before the first result is awaited).  Groups with
 * template / monopole / hi_fit members couple all pixels through their global rows (:522-587, :833-893): with
 * DANGX_SOLVER_DIRECT every context eliminates its per-pixel blocks (pass 1), the Schur rows are added over the contexts
 * (shard order) and the ranks, the small system is solved once, every context back-substitutes (pass 2); the residual
 * check and the refinement steps are shared the same way.  nctx = 1 IS dangx_amp_sample.  DANGX_SOLVER_CG needs nctx = 1. */
int dangx_sky_amp_sample(dangx_ctx *const *ctxs, int nctx, int group, int flag, int ml_mode, int solver, int fluct_mode,
                         uint64_t seed, uint64_t stream, int i_max, double converge, int *cg_iters, int64_t *n_not_spd);

/* ---- dangx_plane_set_sample over SEVERAL contexts of one process: dangx_sky_amp_sample(ctxs, nctx, group, flag, ...) followed by
 * dangx_plane_sweeps_sample(ctxs[r], flag, nsweeps, comp, nind, stream, ...) on every context -- what sample_cg_groups
 * (src/dang_cg_mod.f90:166-171) and the passes of sample_spectral_parameters on the group's planes (src/dang_sample_mod.f90:40-75)
 * do for one (group, flag) pair.  Diffuse groups: dangx_plane_set_sample on every context.  Groups with global-amplitude members
 * (:522-587, :833-893) share their Schur rows over the contexts and ranks as in dangx_sky_amp_sample; when the global members are
 * `template` components, the plane-set kernel covers the model and the small system is well conditioned (every pivot of the
 * equilibrated Schur matrix >= 1e-3, which bounds the global rows' residual by ~1e-12 |b| without measuring it), the
 * back-substitution runs inside the launch that does the sweeps: per context pass 1 and ONE launch.  nctx = 1 IS
 * dangx_plane_set_sample.  accepted[nsweeps], n_not_spd, cg_iters (= -nullity for a coupled group) nullable. */
int dangx_sky_plane_set_sample(dangx_ctx *const *ctxs, int nctx, int group, int flag, int ml_mode, int solver, int fluct_mode,
                               uint64_t seed_amp, uint64_t stream_amp, int i_max, double converge, int nsweeps, const int32_t *comp,
                               const int32_t *nind, const uint64_t *stream, int nsample, uint64_t seed_index, int *cg_iters,
                               int64_t *n_not_spd, int64_t *accepted);

/* ---- secondary seams (type-bound procedures of dang_cg_group), host vectors in the
 * reference's packing [c1: plane1(npix), plane2(npix) | c2: ... ] -------------------- */
int64_t dangx_group_size(dangx_ctx *ctx, int group, int flag);
int dangx_compute_rhs(dangx_ctx *ctx, int group, int flag, double *b);                       /* :326 */
int dangx_compute_Ax(dangx_ctx *ctx, int group, int flag, const double *x, double *res);     /* :598 */
int dangx_compute_sample_vector(dangx_ctx *ctx, int group, int flag, const double *eta, double *res); /* :913 */
/* eval_sed(band, pix, map_n) for all local pixels of one component/band/map -> out[npix] (:778) */
int dangx_eval_sed(dangx_ctx *ctx, int comp, int band, int map_n, double *out);

/* ---- kernels specialised at run time (dang_amd/csrc/dangx_rtc.hip) -------------------------------------------------------
 * The register-resident Metropolis kernels and the fused solve + first sweep are templates on the band count, the plane
 * count and the group size; the library carries instantiations for 3 / 5 / 6 / 8 / 10 / 20 bands.  Any other model shape is
 * compiled by hiprtc from the library's own (embedded) headers on the first sweep that needs it and cached on disk
 * ($DANGX_CACHE_DIR, default ~/.cache/dangx); DANGX_RTC=0 disables this (such shapes then take the LDS-column kernels).
 *   dangx_rtc_kernels: how many kernels this context obtained that way, and (names nullable) their template-ids, one per line.
 *   dangx_rtc_compile: compile-only check (no device needed), e.g. ("dx_kern_chain.h", "dxk::k_index_mh_reg<1, 1, 9, 1>");
 *                      0 on success; log receives the compiler's messages (or the kernel's symbol). */
int dangx_rtc_kernels(dangx_ctx *ctx, int *n, char *names, int names_len);
int dangx_rtc_compile(const char *header, const char *name_expr, char *log, int log_len);

/* ---- per-kernel timing with HIP events on the context's stream ----------------- */
int dangx_profile_enable(dangx_ctx *ctx, int on);
int dangx_profile_reset(dangx_ctx *ctx);
int dangx_profile_get(dangx_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches);
/* the same restricted to the launches of that family that worked on `nplanes` (1: T, Q or U alone; 2: Q+U) planes: the T and
 * Q+U instances of a kernel are different code objects with different costs (bench.py's per-kernel roofline entries).
 * DANGX_ROCTX=1 in the environment additionally wraps every timed launch group in a roctx range named after its family
 * (the ROCm marker library is loaded at run time; `rocprofv3 --marker-trace` shows the ranges). */
int dangx_profile_get_planes(dangx_ctx *ctx, int kernel_id, int nplanes, double *total_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif
