// dangx_entry.hip -- the sampling entry points of the C ABI (include/dangx.h): one (group, flag) pass of sample_cg_groups
// (src/dang_cg_mod.f90:142-177), the per-pixel passes of sample_spectral_parameters (src/dang_sample_mod.f90:21-86), their fusions
// (dangx_amp_index_sample, dangx_plane_set_sample, dangx_plane_sweeps_sample) and update_sky_model + compute_chisq
// (src/dang_data_mod.f90:339-396, 494-526) with the cache of the chi^2 sums the launches leave behind.  Context, model
// synchronisation, the chi^2 ring and the host solvers live in dangx_core.hip; the kernels next to their launchers
// (dangx_amp*.hip, dangx_mh*.hip, dangx_fused.hip, dangx_planeset.hip).
#include "dx_host.h"

namespace {

// out[0] = sum over planes pol_lo..pol_hi of cache[which*3 + plane-1]
__global__ void k_chi_from_cache(const double* __restrict__ cache, int which, int pol_lo, int pol_hi, double* __restrict__ out) {
    double s = 0.0;
    for (int k = pol_lo; k <= pol_hi; ++k) s += cache[which * 3 + k - 1];
    out[0] = s;
}


// ---------------------------------------------------------------------------
// update_sky_model + compute_chisq (src/dang_data_mod.f90:339-396, 494-526), one thread per
// pixel.  sky(i,k,j) is accumulated over components in component_list order in an LDS column;
// the residual and chi^2 follow the reference's expressions.  Block partials of
// sum_k sum_j res^2/rms^2 go to `partial` (second stage: k_reduce).
// (4 waves/SIMD asked for: the kernel streams 2 nb maps per plane against a few SED evaluations, and left to itself the register
// allocator drifts to 130 registers = 3 waves with any small change of the SED helpers: 1.36 -> 1.70 ms per plane at C3)
__global__ __launch_bounds__(BLOCK, 4) void k_sky_chisq(const Model* __restrict__ Mp, int pol_lo, int pol_hi, double* __restrict__ sky,
                            double* __restrict__ res, double* __restrict__ chi_map, double* __restrict__ partial) {
    extern __shared__ double lds[];  // [nb][BS]
    const Model& M = *Mp;
    const int BS = blockDim.x, tid = threadIdx.x;
    const int npix = M.npix, nb = M.nbands;
    const int i = blockIdx.x * BS + tid;
    double chi_sum = 0.0;
    if (i < npix) {
        const bool msk = is_masked(M.mask[i]);
        const bool want_maps = (sky != nullptr) || (res != nullptr);
        if (!msk || want_maps) {
            for (int k = 1; k <= M.nmaps; ++k) {
                const bool in_pol = (k >= pol_lo && k <= pol_hi);
                if (!want_maps && !in_pol) continue;
                for (int j = 0; j < nb; ++j) lds[j * BS + tid] = 0.0;
                for (int l = 0; l < M.ncomp; ++l) {
                    const Comp& c = M.comp[l];
                    const double amp = c.amp[(long long)(k - 1) * npix + i];
                    if (c.type == DANGX_MONOPOLE) continue;  // sets the band offsets instead (src/dang_data_mod.f90:357-361)
                    if (amp == 0.0 && !want_maps && c.type != DANGX_TCMB && !is_global_type(c.type)) continue;
                    double t0, t1;
                    load_theta(M, c, i, k, t0, t1);
                    const Prep pr = sed_prep(c, t0, t1);
                    for (int j = 0; j < nb; ++j) lds[j * BS + tid] = lds[j * BS + tid] + comp_signal(M, c, i, k, j, amp, pr);
                }
                double chi = 0.0;
                for (int j = 0; j < nb; ++j) {
                    const long long q = ((long long)j * M.nmaps + (k - 1)) * npix + i;
                    const double s = lds[j * BS + tid];
                    const double r = (k == 1) ? (M.sig[q] - M.offset[j]) / M.gain[j] - s : M.sig[q] - s;
                    if (sky) sky[q] = s;
                    if (res) res[q] = r;
                    if (!msk && in_pol) {
                        const double rms = M.rms[q];
                        chi = chi + (r * r) / (rms * rms);
                    }
                }
                if (!msk && in_pol) {
                    chi_sum += chi;
                    if (chi_map) chi_map[(long long)(k - 1) * npix + i] = chi / nb;
                }
            }
        }
    }
    __shared__ double sh[16];
    for (int o = 32; o > 0; o >>= 1) chi_sum += __shfl_down(chi_sum, o, 64);
    if ((tid & 63) == 0) sh[tid >> 6] = chi_sum;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < BS / 64; ++w) s += sh[w];
        partial[blockIdx.x] = s;
    }
}


}  // namespace

// chi^2 of plane k as some sweep has just computed it on the host (the full-sky chain's sufficient statistics at the value it ended
// on): the "after" slot of the cache, as a sweep's launch would leave it.  Pending launches are reduced first (their slots are
// older).  Not on planes where a monopole has a signal (chi_byproduct_ok).
int dx_set_chi_after(dangx_ctx* ctx, int k, double chi);

extern "C" {

static int planeset_launch(dangx_ctx* ctx, const GroupArgs& g, const SweepList& sl, int lanes, int solve, int64_t* n_not_spd, int64_t* accepted);
static bool planeset_group(dangx_ctx* ctx, GroupArgs& g, int s1, int s2, int solve);
static bool planeset_items(dangx_ctx* ctx, const GroupArgs& g, int map_n, int nsweeps, const int32_t* comp, const int32_t* nind, const uint64_t* stream, SweepList& sl);

// The sums of squares a sweep leaves behind are chi^2 of update_sky_model's residual -- unless a monopole has a signal on the planes:
// the chain removes it as one more component (eval_signal, src/dang_sample_mod.f90:180-196) ON TOP of the band offset it has
// become (src/dang_data_mod.f90:357-361), while the sky model leaves it out.  Such planes take the explicit pass.
static bool chi_byproduct_ok(const dangx_ctx* ctx, int s1, int s2) {
    unsigned planes = 0;
    for (int k = s1; k <= s2; ++k) planes |= 1u << (k - 1);
    for (int l = 0; l < ctx->hm.ncomp; ++l)
        if (ctx->desc[l].type == DANGX_MONOPOLE && (ctx->tmpl_nz[l] & planes)) return false;
    return true;
}

}  // extern "C"
int dx_set_chi_after(dangx_ctx* ctx, int k, double chi) {
    if (k < 1 || k > ctx->dims.nmaps || !chi_byproduct_ok(ctx, k, k)) return 0;
    (void)hipSetDevice(ctx->device);
    if (chi_flush(ctx)) return 1;
    ctx->chi_host[k - 1] = chi;
    HIPCHK(ctx, hipMemcpyAsync(ctx->chi_cache + 3 + (k - 1), &ctx->chi_host[k - 1], sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ctx->chi_after_valid[k - 1] = true;
    return 0;
}
extern "C" {

int dangx_amp_sample(dangx_ctx* ctx, int group, int flag, int ml_mode, int solver, int fluct_mode, uint64_t seed,
                     uint64_t stream, int i_max, double converge, int* cg_iters, int64_t* n_not_spd) {
    DxRange rg_("dangx_amp_sample");
    if (!ctx) return 1;
    (void)hipSetDevice(ctx->device);
    if (ml_mode != DANGX_ML_SAMPLE && ml_mode != DANGX_ML_OPTIMIZE) return fail(ctx, "bad ml_mode");
    GroupArgs a;
    if (make_group(ctx, group, flag, a) || sync_model(ctx)) return 1;
    a.ml_mode = ml_mode; a.fluct = fluct_mode; a.seed = seed; a.stream = stream;
    const long long SN = (long long)flag_planes_h(flag) * ctx->hm.npix;
    for (int pl = 0; pl < flag_planes_h(flag); ++pl) {  // the planes' cached chi^2 is stale now
        const int k = (flag & DANGX_FLAG_QU) ? 2 + pl : (flag & DANGX_FLAG_T) ? 1 : (flag & DANGX_FLAG_Q) ? 2 : 3;
        ctx->chi_before_valid[k - 1] = ctx->chi_after_valid[k - 1] = ctx->touched_since_amp[k - 1] = false;
        for (int g = 0; g < a.ng; ++g) ctx->plane_nz[a.gc[g]] |= 1u << (k - 1);  // about to be written
    }
    if (cg_iters) *cg_iters = 0;
    if (n_not_spd) *n_not_spd = 0;
    if (a.nt > 0 && solver != DANGX_SOLVER_CG) {
        if (fluct_mode != DANGX_FLUCT_REFERENCE && ml_mode == DANGX_ML_SAMPLE)
            return fail(ctx, "groups with template / monopole / hi_fit members reproduce the reference's fluctuation term only");
        int nullity = 0;
        dangx_ctx* one[1] = {ctx};
        if (device_schur(one, 1, &a, &SN, n_not_spd, &nullity)) return 1;
        if (cg_iters) *cg_iters = -nullity;  // 0: regular system; -k: k directions of the global amplitudes left at their current value
        return 0;
    }
    if (solver == DANGX_SOLVER_CG) {
        if (fluct_mode != DANGX_FLUCT_REFERENCE && ml_mode == DANGX_ML_SAMPLE)
            return fail(ctx, "the CG solver reproduces the reference's fluctuation term only");
        return device_cg(ctx, a, i_max, converge, cg_iters);
    }
    if (ctx->defer_amp) {  // dangx_amp_index_sample: the launch waits for the index sweep it is fused with
        ctx->pending = a; ctx->pending_SN = SN; ctx->have_pending = true;
        return 0;
    }
    // The amplitude phase with chi^2 of the state it leaves as a by-product (src/dang_cg_mod.f90:172-173 asks for it after every
    // group): the plane-set kernel without sweep items forms the residual of the new amplitudes anyway -- where it covers the
    // model (delta bands, the group's members the only components on the planes, reference fluctuation term) the statistics need
    // no pass of their own over the maps.  DANGX_AMP_CHI=0: the stand-alone amplitude kernel (A/B timing).
    static const bool with_chi = [] { const char* e = getenv("DANGX_AMP_CHI"); return !(e && e[0] == '0'); }();
    if (with_chi && (ml_mode == DANGX_ML_OPTIMIZE || fluct_mode == DANGX_FLUCT_REFERENCE)) {
        SweepList sl;
        std::memset(&sl, 0, sizeof(sl));
        sl.s1 = (flag & DANGX_FLAG_QU) ? 2 : (flag & DANGX_FLAG_T) ? 1 : (flag & DANGX_FLAG_Q) ? 2 : 3;
        sl.s2 = (flag & DANGX_FLAG_QU) ? 3 : sl.s1;
        sl.ml_mode = ml_mode;
        GroupArgs ps = a;
        if (planeset_group(ctx, ps, sl.s1, sl.s2, 1)) {
            const int lanes = dx_planeset_lanes(ctx, ps, sl, 1);
            if (lanes) return planeset_launch(ctx, ps, sl, lanes, 1, n_not_spd, nullptr);
        }
    }
    if (n_not_spd) HIPCHK(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long), ctx->stream));
    if (dx_launch_amp(ctx, a, SN)) return 1;
    HIPCHK(ctx, hipGetLastError());
    if (n_not_spd) {
        unsigned long long v = 0;
        HIPCHK(ctx, hipMemcpyAsync(&v, ctx->counters, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *n_not_spd = (int64_t)v;
    }
    return 0;
}

// One (group, flag) pass of sample_cg_groups over several contexts of ONE process (include/dangx.h).  Independent per-pixel
// work is enqueued on every device before the first result is awaited; a coupled group shares its Schur rows.
int dangx_sky_amp_sample(dangx_ctx* const* ctxs, int nctx, int group, int flag, int ml_mode, int solver, int fluct_mode, uint64_t seed,
                         uint64_t stream, int i_max, double converge, int* cg_iters, int64_t* n_not_spd) {
    DxRange rg_("dangx_sky_amp_sample");
    if (!ctxs || nctx < 1) return 1;
    for (int r = 0; r < nctx; ++r) if (!ctxs[r]) return 1;
    dangx_ctx* c0 = ctxs[0];
    if (nctx == 1) return dangx_amp_sample(c0, group, flag, ml_mode, solver, fluct_mode, seed, stream, i_max, converge, cg_iters, n_not_spd);
    if (nctx > 64) return fail(c0, "too many contexts");
    auto bubble = [&](dangx_ctx* who) { if (who != c0) c0->err = who->err; return 1; };
    if (ml_mode != DANGX_ML_SAMPLE && ml_mode != DANGX_ML_OPTIMIZE) return fail(c0, "bad ml_mode");
    GroupArgs probe;
    if (make_group(c0, group, flag, probe)) return 1;
    if (cg_iters) *cg_iters = 0;
    if (n_not_spd) *n_not_spd = 0;
    if (probe.nt == 0 && solver != DANGX_SOLVER_CG) {  // block diagonal: every shard on its own
        for (int r = 0; r < nctx; ++r) {
            if (n_not_spd) {  // the amplitude kernels add to counters[0]: each context starts its launch from zero, on its own stream
                (void)hipSetDevice(ctxs[r]->device);
                HIPCHK(c0, hipMemsetAsync(ctxs[r]->counters, 0, sizeof(unsigned long long), ctxs[r]->stream));
            }
            if (dangx_amp_sample(ctxs[r], group, flag, ml_mode, solver, fluct_mode, seed, stream, i_max, converge, nullptr, nullptr))
                return bubble(ctxs[r]);
        }
        if (n_not_spd)   // the counters are read after every device has its launch
            for (int r = 0; r < nctx; ++r) {
                unsigned long long v = 0;
                (void)hipSetDevice(ctxs[r]->device);
                HIPCHK(c0, hipMemcpyAsync(&v, ctxs[r]->counters, sizeof(v), hipMemcpyDeviceToHost, ctxs[r]->stream));
                HIPCHK(c0, hipStreamSynchronize(ctxs[r]->stream));
                *n_not_spd += (int64_t)v;
            }
        return 0;
    }
    if (solver == DANGX_SOLVER_CG)
        return fail(c0, "the device CG (DANGX_SOLVER_CG) iterates on ONE context per process: use DANGX_SOLVER_DIRECT, or one process per GPU with dangx_set_allreduce");
    if (fluct_mode != DANGX_FLUCT_REFERENCE && ml_mode == DANGX_ML_SAMPLE)
        return fail(c0, "groups with template / monopole / hi_fit members reproduce the reference's fluctuation term only");
    std::vector<GroupArgs> as((size_t)nctx);
    std::vector<long long> SNs((size_t)nctx);
    for (int r = 0; r < nctx; ++r) {
        dangx_ctx* c = ctxs[r];
        (void)hipSetDevice(c->device);
        if (make_group(c, group, flag, as[r]) || sync_model(c)) return bubble(c);
        as[r].ml_mode = ml_mode; as[r].fluct = fluct_mode; as[r].seed = seed; as[r].stream = stream;
        SNs[r] = (long long)flag_planes_h(flag) * c->hm.npix;
        for (int pl = 0; pl < flag_planes_h(flag); ++pl) {
            const int k = (flag & DANGX_FLAG_QU) ? 2 + pl : (flag & DANGX_FLAG_T) ? 1 : (flag & DANGX_FLAG_Q) ? 2 : 3;
            c->chi_before_valid[k - 1] = c->chi_after_valid[k - 1] = c->touched_since_amp[k - 1] = false;
            for (int g = 0; g < as[r].ng; ++g) c->plane_nz[as[r].gc[g]] |= 1u << (k - 1);
        }
        if (as[r].nglob != as[0].nglob || as[r].nt != as[0].nt) return fail(c0, "the contexts disagree on the group's global-amplitude members");
    }
    int nullity = 0;
    if (device_schur(ctxs, nctx, as.data(), SNs.data(), n_not_spd, &nullity)) return 1;
    if (cg_iters) *cg_iters = -nullity;
    return 0;
}

int dangx_index_sample(dangx_ctx* ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                       uint64_t stream, int64_t* accepted) {
    DxRange rg_("dangx_index_sample");
    if (!ctx || check_comp(ctx, comp)) return 1;
    (void)hipSetDevice(ctx->device);
    {   // this sweep makes the component's index map pixel dependent on the touched planes
        unsigned touched = 0;
        if (map_n == -1) touched = 6u; else if (map_n >= 1 && map_n <= 3) touched = 1u << (map_n - 1);
        if (ctx->idx_const[comp] & touched) { ctx->idx_const[comp] &= ~touched; ctx->dirty = true; }
        idx_written(ctx, comp);
        if (nind >= 0 && nind < DANGX_MAX_IND) {  // a Q+U sweep writes one value to both planes (:465); a Q or U sweep to one
            if (map_n == -1) ctx->qu_equal[comp] |= 1u << nind;
            else if (map_n == 2 || map_n == 3) ctx->qu_equal[comp] &= ~(1u << nind);
        }
    }
    if (sync_model(ctx)) return 1;
    const dangx_comp_desc& d = ctx->desc[comp];
    if (nind < 0 || nind >= d.nindices) return fail(ctx, "index number out of range");
    IndexArgs a;
    a.comp = comp; a.nind = nind; a.nsample = nsample; a.ml_mode = ml_mode; a.seed = seed; a.stream = stream;
    if (map_n == -1) { a.s1 = 2; a.s2 = 3; }                       // src/dang_sample_mod.f90:157-163
    else if (map_n >= 1 && map_n <= 3) { a.s1 = a.s2 = map_n; }
    else return fail(ctx, "There is something wrong with the poltype flag (map_n must be 1,2,3 or -1)");
    if (a.s2 > ctx->dims.nmaps) return fail(ctx, "map_n exceeds nmaps");
    if (d.lnl_type[nind] < DANGX_LNL_CHISQ || d.lnl_type[nind] > DANGX_LNL_PRIOR) return fail(ctx, "bad lnl_type");
    if (ml_mode != DANGX_ML_SAMPLE && ml_mode != DANGX_ML_OPTIMIZE) return fail(ctx, "bad ml_mode");
    const int Sp = a.s2 - a.s1 + 1;
    a.others = 0;
    for (int l = 0; l < ctx->hm.ncomp; ++l)
        if (l != comp && ((ctx->plane_nz[l] & ((1u << (a.s1 - 1)) | (1u << (a.s2 - 1)))) || ctx->desc[l].type == DANGX_TCMB ||
                          is_global_type(ctx->desc[l].type)))
            a.others |= 1u << l;
    // chain mode: factorised SED when every band is a delta bandpass
    const bool all_delta = ctx->hm.all_delta != 0;
    a.mode = CH_GENERIC;
    a.bp = all_delta ? 0 : 1;
    if (d.type == DANGX_POWERLAW) a.mode = CH_POW;
    else if (d.type == DANGX_MBB) a.mode = nind == 0 ? CH_MBB_BETA : CH_MBB_T;
    else if (d.type == DANGX_LOGNORMAL && all_delta) a.mode = nind == 0 ? CH_LOGN_NUP : CH_LOGN_W;
    // with bandpass-integrated bands (or T_cmb / template-type components present) the compile-time modes exist for the
    // chisq likelihood with a gaussian / uniform prior only; everything else takes the run-time generic chain
    if (a.bp && (d.lnl_type[nind] != DANGX_LNL_CHISQ || d.prior_type[nind] == DANGX_PRIOR_JEFFREYS)) a.mode = CH_GENERIC;
    // LDS columns: (2*Sp+1)*nb doubles per thread; pick the block so that >= 2 blocks fit in 160 KiB
    const size_t per_thread = (size_t)(2 * Sp + 1) * ctx->hm.nbands * sizeof(double);
    const size_t tabsz = (size_t)(TROWS * ctx->hm.ncomp + 3) * ctx->hm.nbands * sizeof(double);
    int bs = 256;
    while (bs > 64 && tabsz + per_thread * bs > 76 * 1024) bs >>= 1;
    const size_t lds = tabsz + per_thread * bs;
    const bool reg_ok = !a.bp && d.lnl_type[nind] == DANGX_LNL_CHISQ && d.prior_type[nind] != DANGX_PRIOR_JEFFREYS &&
                        a.mode != CH_GENERIC && dx_mh_reg_supported(ctx, a.mode, ctx->hm.nbands, Sp);
    if (reg_ok) bs = BLOCK;  // register-resident form: no LDS columns
    // lanes per pixel: of the chain's register form, or of the fused launch when a solve on these planes is waiting for this
    // sweep and the model takes the one-launch form (decided now: the grid and the chi^2 buffers are sized by it)
    int lanes = reg_ok ? dx_mh_reg_lanes(ctx->hm.nbands, Sp) : 1;
    const int fused_lanes = (ctx->have_pending && reg_ok) ? dx_fused_lanes(ctx, ctx->pending, a, Sp) : 0;
    if (fused_lanes) lanes = fused_lanes;
    const unsigned nblk = nblocks((long long)ctx->hm.npix * lanes, bs);
    constexpr int RSTAGE = 128;  // blocks of the first reduction stage
    double* chi_buf = nullptr;
    if (chi_next(ctx, nblk, &chi_buf)) return 1;
    // the sweep kernels take their [4][nblk] chi^2 partial buffer from ctx->partial: lend them the ring's, give the
    // context's own back on every way out of the launch section
    struct Lend {
        dangx_ctx* c; double* saved;
        Lend(dangx_ctx* c_, double* b) : c(c_), saved(c_->partial) { c->partial = b; }
        void back() { if (c) { c->partial = saved; c = nullptr; } }
        ~Lend() { back(); }
    } lend(ctx, chi_buf);
    if (accepted) HIPCHK(ctx, hipMemsetAsync(ctx->counters + 1, 0, 2 * sizeof(unsigned long long), ctx->stream));
    bool fused = false;
    if (ctx->have_pending) {  // an amplitude solve on these planes is waiting: one launch for both, or the solve first
        ctx->have_pending = false;
        unsigned long long* accp = accepted ? ctx->counters + 1 : nullptr;
        if (fused_lanes) {
            Timed t(ctx, DANGX_K_AMP_INDEX, Sp);
            fused = dx_launch_fused(ctx, ctx->pending, a, Sp, fused_lanes, nblk, accp);
        }
        if (!fused) {
            if (fused_lanes) return fail(ctx, "the fused solve + sweep launch failed after its kernel was prepared");
            if (dx_launch_amp(ctx, ctx->pending, ctx->pending_SN)) return 1;
        }
    }
    if (!fused && ctx->pair_on) {  // dangx_index_sample_pair: this sweep and the sweep of index nind + 1 in one launch
        ctx->pair_on = false;
        if (reg_ok && nind + 1 < d.nindices) {
            IndexArgs b = a;
            b.nind = nind + 1; b.stream = ctx->pair_stream;
            b.mode = (d.type == DANGX_MBB) ? CH_MBB_T : (d.type == DANGX_LOGNORMAL && all_delta) ? CH_LOGN_W : CH_GENERIC;
            const bool ok_b = d.lnl_type[nind + 1] == DANGX_LNL_CHISQ && d.prior_type[nind + 1] != DANGX_PRIOR_JEFFREYS;
            unsigned long long* accp = accepted ? ctx->counters + 1 : nullptr;  // counters[1], counters[2]
            if (ok_b) {
                Timed t(ctx, DANGX_K_INDEX_MH, Sp);
                fused = ctx->pair_done = dx_launch_mh_pair(ctx, a, b, Sp, nblk, accp);
            }
        }
    }
    if (!fused) {
        Timed t(ctx, DANGX_K_INDEX_MH, Sp);
        unsigned long long* accp = accepted ? ctx->counters + 1 : nullptr;
        const bool fast = d.lnl_type[nind] == DANGX_LNL_CHISQ &&
                          (a.mode == CH_POW || a.mode == CH_MBB_BETA || a.mode == CH_MBB_T);
        if (!(reg_ok && dx_launch_mh_reg(ctx, a, Sp, nblk, accp))) dx_launch_mh_lds(ctx, a, fast, Sp, nblk, bs, lds, accp);
    }
    lend.back();
    HIPCHK(ctx, hipGetLastError());  // a failed launch must not leave a pending entry over partials nobody wrote
    {   // fused chi^2 of the touched planes (before = state left by the amplitude phase, after = new state): the block
        // partials wait in the ring (chi_flush) until a value is asked for
        const bool wb = !ctx->touched_since_amp[a.s1 - 1];
        auto& pend = ctx->chi_pend[ctx->chi_npend++];
        pend.nblk = nblk; pend.s1 = a.s1; pend.s2 = a.s2; pend.wb = wb ? 1 : 0;
        const bool chi_ok = chi_byproduct_ok(ctx, a.s1, a.s2);
        for (int k = a.s1; k <= a.s2; ++k) {
            if (wb) ctx->chi_before_valid[k - 1] = chi_ok;
            ctx->chi_after_valid[k - 1] = chi_ok;
            ctx->touched_since_amp[k - 1] = true;
        }
    }
    if (accepted) {
        unsigned long long v = 0;
        HIPCHK(ctx, hipMemcpyAsync(&v, ctx->counters + 1, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *accepted = (int64_t)v;
    }
    return 0;
}

// dangx_index_sample(comp, nind, ...) followed by dangx_index_sample(comp, nind + 1, ...) on the same planes: two
// consecutive indices of ONE component (the dust beta and dust T sweeps).  Nothing the second sweep removes from the data
// has changed in between, so where the register chain covers both (chisq likelihood, gaussian / uniform priors, delta
// bands; mbb beta -> T, log-normal nu_p -> w) they run in one launch on one staging of the maps -- bit for bit the two
// calls, which everything else takes.
int dangx_index_sample_pair(dangx_ctx* ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                            uint64_t stream_first, uint64_t stream_second, int64_t* accepted_first, int64_t* accepted_second) {
    DxRange rg_("dangx_index_sample_pair");
    if (!ctx || check_comp(ctx, comp)) return 1;
    static const bool enabled = [] { const char* e = getenv("DANGX_FUSE"); return !(e && e[0] == '0'); }();
    const bool want_counts = accepted_first || accepted_second;
    int64_t acc1 = 0;
    ctx->pair_on = enabled; ctx->pair_done = false; ctx->pair_stream = stream_second;
    int rc = dangx_index_sample(ctx, comp, nind, map_n, nsample, ml_mode, seed, stream_first, want_counts ? &acc1 : nullptr);
    ctx->pair_on = false;
    if (rc) return rc;
    if (accepted_first) *accepted_first = acc1;
    if (!ctx->pair_done) return dangx_index_sample(ctx, comp, nind + 1, map_n, nsample, ml_mode, seed, stream_second, accepted_second);
    ctx->pair_done = false;
    if (nind + 1 < DANGX_MAX_IND) {
        if (map_n == -1) ctx->qu_equal[comp] |= 1u << (nind + 1);
        else if (map_n == 2 || map_n == 3) ctx->qu_equal[comp] &= ~(1u << (nind + 1));
    }
    if (accepted_second) {
        unsigned long long v = 0;
        HIPCHK(ctx, hipMemcpyAsync(&v, ctx->counters + 2, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *accepted_second = (int64_t)v;
    }
    return 0;
}

// dangx_amp_sample(group, flag, ...) followed by dangx_index_sample(comp, nind, map_n, ...) -- the amplitude solve of a CG
// group and the first index sweep on the same planes, which is how sample_cg_groups / sample_spectral_parameters follow
// each other plane set by plane set (src/dang.f90 main loop) -- with ONE kernel launch when the model allows it
// (dangx_fused.hip: delta bands, diffuse members only, direct solver, reference fluctuation term, chisq likelihood,
// gaussian / uniform prior, the sampled component a member of the group whose other members are the only other
// components on these planes).  Results are those of the two calls, bit for bit; every other configuration IS the two calls.
int dangx_amp_index_sample(dangx_ctx* ctx, int group, int flag, int ml_mode, int solver, int fluct_mode, uint64_t seed_amp,
                           uint64_t stream_amp, int i_max, double converge, int comp, int nind, int map_n, int nsample,
                           uint64_t seed_index, uint64_t stream_index, int* cg_iters, int64_t* n_not_spd, int64_t* accepted) {
    DxRange rg_("dangx_amp_index_sample");
    if (!ctx || check_comp(ctx, comp)) return 1;
    if (cg_iters) *cg_iters = 0;
    static const bool enabled = [] { const char* e = getenv("DANGX_FUSE"); return !(e && e[0] == '0'); }();  // A/B switch
    (void)hipSetDevice(ctx->device);
    if (sync_model(ctx)) return 1;  // all_delta and the constant-plane flags the decision below reads are set there
    bool can = enabled && solver == DANGX_SOLVER_DIRECT && ctx->hm.all_delta != 0 &&
               (ml_mode == DANGX_ML_OPTIMIZE || fluct_mode == DANGX_FLUCT_REFERENCE);
    // the planes of the sweep are the planes of the solve
    const int want = (flag == DANGX_FLAG_T) ? 1 : (flag == DANGX_FLAG_Q) ? 2 : (flag == DANGX_FLAG_U) ? 3 : (flag == DANGX_FLAG_QU) ? -1 : 0;
    can = can && want != 0 && want == map_n;
    if (can) {
        const dangx_comp_desc& d = ctx->desc[comp];
        const unsigned touched = (map_n == -1) ? 6u : 1u << (map_n - 1);
        // a sweep that turns a spatially constant index map into a varying one changes which SED route the SOLVE takes
        // (host-evaluated row against per-pixel evaluation) if it is launched after the descriptor update: first sweeps
        // on constant maps go the two-call way
        can = nind >= 0 && nind < d.nindices && !(ctx->idx_const[comp] & touched) && d.cg_group == group && d.sample_amplitude &&
              d.lnl_type[nind] == DANGX_LNL_CHISQ && d.prior_type[nind] != DANGX_PRIOR_JEFFREYS &&
              (d.type == DANGX_POWERLAW || d.type == DANGX_MBB);
    }
    if (can) {
        GroupArgs g;
        if (make_group(ctx, group, flag, g)) return 1;
        can = g.nt == 0 && g.no == 0 && g.nuc == 0 &&
              dx_fused_supported(ctx->desc[comp].type == DANGX_POWERLAW ? CH_POW : (nind == 0 ? CH_MBB_BETA : CH_MBB_T), ctx->hm.nbands, g.ng);
        for (int l = 0; can && l < ctx->hm.ncomp; ++l)  // a T_cmb component is an "other" of every sweep and never a diffuse member
            if (ctx->desc[l].type == DANGX_TCMB) can = false;
    }
    if (!can) {
        const int rc = dangx_amp_sample(ctx, group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, i_max, converge, cg_iters, n_not_spd);
        return rc ? rc : dangx_index_sample(ctx, comp, nind, map_n, nsample, ml_mode, seed_index, stream_index, accepted);
    }
    if (n_not_spd) HIPCHK(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long), ctx->stream));
    ctx->defer_amp = true;
    int rc = dangx_amp_sample(ctx, group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, i_max, converge, nullptr, nullptr);
    ctx->defer_amp = false;
    if (rc) { ctx->have_pending = false; return rc; }
    rc = dangx_index_sample(ctx, comp, nind, map_n, nsample, ml_mode, seed_index, stream_index, accepted);
    if (ctx->have_pending) {  // the sweep failed before its launch site: the solve still happens, as with the two calls
        ctx->have_pending = false;
        if (dx_launch_amp(ctx, ctx->pending, ctx->pending_SN)) return 1;
    }
    if (rc) return rc;
    if (n_not_spd) {
        unsigned long long v = 0;
        HIPCHK(ctx, hipMemcpyAsync(&v, ctx->counters, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *n_not_spd = (int64_t)v;
    }
    return 0;
}

// One k_plane_set launch (dx_kern_planeset.h) with its bookkeeping: solve = 1 starts with the group's amplitude solve (what
// dangx_amp_sample records: the planes' cached chi^2 is stale, the members' amplitudes are about to be written), sl.n sweep items
// follow (what dangx_index_sample records per sweep).  The chi^2 by-products go to the ring: with a solve "before" = the state the
// solve leaves and "after" = the last sweep's (both the same value without sweeps); without, as for any sweep.
static int planeset_launch(dangx_ctx* ctx, const GroupArgs& g, const SweepList& sl, int lanes, int solve, int64_t* n_not_spd, int64_t* accepted) {
    const bool qu = sl.s2 > sl.s1;
    bool wb = true;
    if (solve) {
        for (int k = sl.s1; k <= sl.s2; ++k) {
            ctx->chi_before_valid[k - 1] = ctx->chi_after_valid[k - 1] = ctx->touched_since_amp[k - 1] = false;
            for (int q = 0; q < g.ng; ++q) ctx->plane_nz[g.gc[q]] |= 1u << (k - 1);
        }
    } else {
        wb = !ctx->touched_since_amp[sl.s1 - 1];
    }
    for (int q = 0; q < sl.n; ++q)
        for (int e = 0; e <= sl.s[q].pair; ++e) {
            idx_written(ctx, sl.s[q].comp);
            if (qu) ctx->qu_equal[sl.s[q].comp] |= 1u << (sl.s[q].nind + e);
            else if (sl.s1 == 2 || sl.s1 == 3) ctx->qu_equal[sl.s[q].comp] &= ~(1u << (sl.s[q].nind + e));
        }
    const unsigned nblk = nblocks((long long)ctx->hm.npix * lanes, BLOCK);
    double* chi_buf = nullptr;
    if (chi_next(ctx, nblk, &chi_buf)) return 1;
    HIPCHK(ctx, hipMemsetAsync(ctx->counters, 0, 16 * sizeof(unsigned long long), ctx->stream));
    {
        double* saved = ctx->partial;
        ctx->partial = chi_buf;
        bool ok;
        {
            Timed t(ctx, !solve ? DANGX_K_INDEX_MH : sl.n ? DANGX_K_AMP_INDEX : DANGX_K_AMP_DIRECT, sl.s2 - sl.s1 + 1);
            ok = dx_launch_planeset(ctx, g, sl, lanes, solve, nblk, accepted ? ctx->counters + 4 : nullptr);
        }
        ctx->partial = saved;
        if (!ok) return fail(ctx, "the plane-set launch failed after its kernel was prepared");
    }
    HIPCHK(ctx, hipGetLastError());
    {
        auto& pend = ctx->chi_pend[ctx->chi_npend++];
        pend.nblk = nblk; pend.s1 = sl.s1; pend.s2 = sl.s2; pend.wb = wb ? 1 : 0;
        pend.ns = 0;
        for (int q = 0; q < sl.n; ++q)   // the masked sums of the swept index maps ride along (rows 4 ..), in the items' order
            for (int e = 0; e <= sl.s[q].pair; ++e) {
                pend.slot[pend.ns++] = idx_slot(sl.s[q].comp, sl.s[q].nind + e, sl.s1);
                for (int k = sl.s1; k <= sl.s2; ++k) ctx->idxsum_dev[sl.s[q].comp][sl.s[q].nind + e][k - 1] = !ctx->idx_ext[sl.s[q].comp];
            }
        const bool chi_ok = chi_byproduct_ok(ctx, sl.s1, sl.s2);
        for (int k = sl.s1; k <= sl.s2; ++k) {
            if (wb) ctx->chi_before_valid[k - 1] = chi_ok;
            ctx->chi_after_valid[k - 1] = chi_ok;
            ctx->touched_since_amp[k - 1] = sl.n > 0 || (!solve && ctx->touched_since_amp[k - 1]);
        }
    }
    if (n_not_spd || accepted) {
        unsigned long long v[16];
        HIPCHK(ctx, hipMemcpyAsync(v, ctx->counters, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (n_not_spd) *n_not_spd = (int64_t)v[0];
        if (accepted) {   // the kernel counts per item (1 + pair entries each); the items follow the list's order
            int slot = 4, s = 0;
            for (int q = 0; q < sl.n; ++q)
                for (int e = 0; e <= sl.s[q].pair; ++e) accepted[s++] = (int64_t)v[slot++];
        }
    }
    return 0;
}

// the sweeps (comp[s], nind[s]), s = 0 .. nsweeps-1, as the items of a plane-set launch over group g's members: consecutive
// indices of a component travel in one item.  false: some sweep has no register-chain form, or the list is too long
static bool planeset_items(dangx_ctx* ctx, const GroupArgs& g, int map_n, int nsweeps, const int32_t* comp, const int32_t* nind,
                           const uint64_t* stream, SweepList& sl) {
    const unsigned touched = (map_n == -1) ? 6u : 1u << (map_n - 1);
    std::memset(&sl, 0, sizeof(sl));
    for (int s = 0; s < nsweeps; ++s) {
        const dangx_comp_desc& d = ctx->desc[comp[s]];
        int gm = -1;
        for (int q = 0; q < g.ng; ++q) if (g.gc[q] == comp[s]) gm = q;
        if (!(gm >= 0 && !(ctx->idx_const[comp[s]] & touched) && d.lnl_type[nind[s]] == DANGX_LNL_CHISQ &&
              d.prior_type[nind[s]] != DANGX_PRIOR_JEFFREYS && (d.type == DANGX_POWERLAW || d.type == DANGX_MBB || d.type == DANGX_LOGNORMAL)))
            return false;
        const int mode = (d.type == DANGX_POWERLAW) ? CH_POW : (d.type == DANGX_MBB) ? (nind[s] == 0 ? CH_MBB_BETA : CH_MBB_T) : (nind[s] == 0 ? CH_LOGN_NUP : CH_LOGN_W);
        if (sl.n > 0 && sl.s[sl.n - 1].comp == comp[s] && !sl.s[sl.n - 1].pair && sl.s[sl.n - 1].nind + 1 == nind[s] &&
            (sl.s[sl.n - 1].mode == CH_MBB_BETA || sl.s[sl.n - 1].mode == CH_LOGN_NUP)) {
            sl.s[sl.n - 1].pair = 1; sl.s[sl.n - 1].stream2 = stream[s];   // index nind + 1 of the same component: one item
            continue;
        }
        for (int q = 0; q < sl.n; ++q) if (sl.s[q].comp == comp[s]) return false;  // a component's sweeps must be consecutive
        if (sl.n == DX_MAX_SWEEPS) return false;
        SweepItem& it = sl.s[sl.n++];
        it.comp = comp[s]; it.nind = nind[s]; it.mode = mode; it.pair = 0; it.gmember = gm; it.stream = stream[s]; it.stream2 = 0;
    }
    sl.s1 = (map_n == -1) ? 2 : map_n; sl.s2 = (map_n == -1) ? 3 : map_n;
    return true;
}

// What every plane-set launch needs: the group's diffuse members are the only components with a signal on planes s1..s2 --
// except, for the sweeps alone (solve = 0), `template` and `monopole` components: their signal template_amplitudes(band, map) * template(pix, map)
// (eval_signal, src/dang_component_mod.f90:754-776) is one more term of "every other component" and is removed when the residual
// is formed; g.uc / g.nuc become the list of those (at most 4).  A global-amplitude component whose template map is identically
// zero on these planes (a Q/U dust template seen from the T plane set) has no signal there and is ignored -- also by the solve,
// whose compute_rhs would remove tamp * 0 on its unfitted bands (:445-460).  hi_fit (a per-pixel Planck factor) and T_cmb
// components with a signal on the planes keep the run-time-typed kernels.
// solve = 2: the back-substitution of a template group's Schur solve (dangx_sky_plane_set_sample) followed by the sweeps: every
// template with a signal on the planes must then be a global member of THIS group (its new amplitudes, on every band, are what pass 2
// removes from the data: dx_ampreg.h, HT form), and the group has no monopole / hi_fit member.
static bool planeset_group(dangx_ctx* ctx, GroupArgs& g, int s1, int s2, int solve) {
    unsigned planes = 0;
    for (int k = s1; k <= s2; ++k) planes |= 1u << (k - 1);
    int ntg = 0, tg[MAXC];
    for (int l = 0; l < ctx->hm.ncomp; ++l) {
        const int t = ctx->desc[l].type;
        if (t == DANGX_TCMB) return false;
        if (!is_global_type(t) || !(ctx->tmpl_nz[l] & planes)) continue;
        // (a monopole's signal is template_amplitudes(band, map) * template(pix, map) too -- its amplitudes are also the band offsets,
        // which the T launch reads from the block's table; as the member of a solve it keeps the separate passes)
        if (solve == 1 || ntg == 4 || !(t == DANGX_TEMPLATE || (t == DANGX_MONOPOLE && solve == 0))) return false;
        if (solve == 2) {
            bool member = false;
            for (int q = 0; q < g.nt; ++q) member = member || g.tc[q] == l;
            if (!member) return false;
        }
        tg[ntg++] = l;
    }
    if (solve == 1 && g.nt != 0) return false;          // a coupled solve is the Schur path
    if (solve == 2 && g.nt != ntg) return false;        // a global member without a signal here, or of another type
    for (int o = 0; o < g.no; ++o)                      // any other diffuse component with a signal on the planes
        if (!is_global_type(ctx->desc[g.oc[o]].type)) return false;
    g.nuc = ntg;
    for (int t = 0; t < ntg; ++t) g.uc[t] = tg[t];
    return true;
}

// dangx_amp_sample(group, flag, ...) followed by dangx_index_sample(comp[s], nind[s], map_n of the flag, ...) for s = 0 ..
// nsweeps-1 -- everything one iteration of the main loop does on ONE plane set of a CG group: the solve of sample_cg_groups
// (src/dang_cg_mod.f90:166-171) and the passes of sample_spectral_parameters that touch these planes (src/dang_sample_mod.f90:
// 40-75), in the reference's order.  Where k_plane_set covers the model (dx_kern_planeset.h: many bands and members, every swept
// component a member of the group) all of it is ONE launch with the members' SED columns kept in LDS; everything else IS those
// calls (through dangx_amp_index_sample / dangx_index_sample_pair where they apply).
int dangx_plane_set_sample(dangx_ctx* ctx, int group, int flag, int ml_mode, int solver, int fluct_mode, uint64_t seed_amp, uint64_t stream_amp,
                           int i_max, double converge, int nsweeps, const int32_t* comp, const int32_t* nind, const uint64_t* stream,
                           int nsample, uint64_t seed_index, int* cg_iters, int64_t* n_not_spd, int64_t* accepted) {
    DxRange rg_("dangx_plane_set_sample");
    if (!ctx || nsweeps < 1 || !comp || !nind || !stream) return 1;
    (void)hipSetDevice(ctx->device);
    if (sync_model(ctx)) return 1;
    const int map_n = (flag == DANGX_FLAG_T) ? 1 : (flag == DANGX_FLAG_Q) ? 2 : (flag == DANGX_FLAG_U) ? 3 : (flag == DANGX_FLAG_QU) ? -1 : 0;
    if (map_n == 0) return fail(ctx, "flag must be exactly one of T(1), Q(2), U(4), Q+U(8)");
    for (int s = 0; s < nsweeps; ++s)
        if (check_comp(ctx, comp[s]) || nind[s] < 0 || nind[s] >= ctx->desc[comp[s]].nindices) return fail(ctx, "sweep list: component / index out of range");
    if (cg_iters) *cg_iters = 0;
    if (n_not_spd) *n_not_spd = 0;
    static const bool enabled = [] { const char* e = getenv("DANGX_FUSE"); return !(e && e[0] == '0'); }();
    // ---- does the one-launch form cover this?  (the conditions of dangx_amp_index_sample, for every sweep of the list)
    bool can = enabled && nsweeps <= 2 * DX_MAX_SWEEPS && solver == DANGX_SOLVER_DIRECT &&
               (ml_mode == DANGX_ML_OPTIMIZE || fluct_mode == DANGX_FLUCT_REFERENCE) &&
               (ml_mode == DANGX_ML_SAMPLE || ml_mode == DANGX_ML_OPTIMIZE);
    GroupArgs g;
    SweepList sl;
    std::memset(&sl, 0, sizeof(sl));
    if (can) {
        if (make_group(ctx, group, flag, g)) return 1;
        can = planeset_group(ctx, g, (map_n == -1) ? 2 : map_n, (map_n == -1) ? 3 : map_n, 1) && planeset_items(ctx, g, map_n, nsweeps, comp, nind, stream, sl);
    }
    int lanes = 0;
    if (can) {
        sl.nsample = nsample; sl.ml_mode = ml_mode; sl.seed = seed_index;
        g.ml_mode = ml_mode; g.fluct = fluct_mode; g.seed = seed_amp; g.stream = stream_amp;
        lanes = dx_planeset_lanes(ctx, g, sl, 1);
    }
    if (!lanes) {  // the calls this entry point stands for, through the two-step fusions where they apply
        {   // template / monopole / hi_fit components in the model: the solve is the Schur path (or sees them as other components),
            // and the sweeps beside them go together where the sweeps-only launch covers them (dangx_plane_sweeps_sample)
            GroupArgs gq;
            if (make_group(ctx, group, flag, gq)) return 1;
            if (gq.nt != 0 && solver == DANGX_SOLVER_DIRECT) {   // a coupled group: the Schur solve, its back-substitution with the sweeps
                dangx_ctx* one[1] = {ctx};
                return dangx_sky_plane_set_sample(one, 1, group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, i_max, converge, nsweeps,
                                                  comp, nind, stream, nsample, seed_index, cg_iters, n_not_spd, accepted);
            }
            if (gq.nt != 0 || gq.nuc != 0) {
                const int rc = dangx_amp_sample(ctx, group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, i_max, converge, cg_iters, n_not_spd);
                return rc ? rc : dangx_plane_sweeps_sample(ctx, flag, nsweeps, comp, nind, stream, nsample, ml_mode, seed_index, accepted);
            }
        }
        int s = 0;
        int64_t acc = 0, acc2 = 0;
        int rc = dangx_amp_index_sample(ctx, group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, i_max, converge, comp[0], nind[0],
                                        map_n, nsample, seed_index, stream[0], cg_iters, n_not_spd, accepted ? &acc : nullptr);
        if (rc) return rc;
        if (accepted) accepted[0] = acc;
        for (s = 1; s < nsweeps; ++s) {
            if (s + 1 < nsweeps && comp[s + 1] == comp[s] && nind[s + 1] == nind[s] + 1) {
                rc = dangx_index_sample_pair(ctx, comp[s], nind[s], map_n, nsample, ml_mode, seed_index, stream[s], stream[s + 1],
                                             accepted ? &acc : nullptr, accepted ? &acc2 : nullptr);
                if (rc) return rc;
                if (accepted) { accepted[s] = acc; accepted[s + 1] = acc2; }
                ++s;
            } else {
                rc = dangx_index_sample(ctx, comp[s], nind[s], map_n, nsample, ml_mode, seed_index, stream[s], accepted ? &acc : nullptr);
                if (rc) return rc;
                if (accepted) accepted[s] = acc;
            }
        }
        return 0;
    }
    return planeset_launch(ctx, g, sl, lanes, 1, n_not_spd, accepted);
}

// dangx_plane_set_sample over several contexts of ONE process (include/dangx.h).  Diffuse groups: that call on every context.  A
// group with global-amplitude members couples the contexts through its Schur rows (dangx_sky_amp_sample); where its global members
// are `template` components and the plane-set kernel covers the model, a well-conditioned solve (device_schur) leaves its
// back-substitution -- the per-pixel solve on the data minus the templates' NEW signal -- to the launch that runs the sweeps:
// pass 1, the small host solve, then ONE launch per context instead of pass 2, the residual pass and the sweeps.
int dangx_sky_plane_set_sample(dangx_ctx* const* ctxs, int nctx, int group, int flag, int ml_mode, int solver, int fluct_mode,
                               uint64_t seed_amp, uint64_t stream_amp, int i_max, double converge, int nsweeps, const int32_t* comp,
                               const int32_t* nind, const uint64_t* stream, int nsample, uint64_t seed_index, int* cg_iters,
                               int64_t* n_not_spd, int64_t* accepted) {
    DxRange rg_("dangx_sky_plane_set_sample");
    if (!ctxs || nctx < 1 || nctx > 64 || nsweeps < 1 || !comp || !nind || !stream) return 1;
    for (int r = 0; r < nctx; ++r) if (!ctxs[r]) return 1;
    dangx_ctx* c0 = ctxs[0];
    auto bubble = [&](dangx_ctx* who) { if (who != c0) c0->err = who->err; return 1; };
    GroupArgs probe;
    if (make_group(c0, group, flag, probe)) return 1;
    if (cg_iters) *cg_iters = 0;
    if (n_not_spd) *n_not_spd = 0;
    std::vector<int64_t> acc((size_t)nsweeps, 0);
    if (accepted) for (int s = 0; s < nsweeps; ++s) accepted[s] = 0;
    auto add_acc = [&]() { if (accepted) for (int s = 0; s < nsweeps; ++s) accepted[s] += acc[(size_t)s]; };
    if (probe.nt == 0 || solver != DANGX_SOLVER_DIRECT) {
        if (probe.nt != 0 && nctx > 1)
            return fail(c0, "the device CG (DANGX_SOLVER_CG) iterates on ONE context per process: use DANGX_SOLVER_DIRECT, or one process per GPU with dangx_set_allreduce");
        for (int r = 0; r < nctx; ++r) {
            int64_t bad = 0;
            if (dangx_plane_set_sample(ctxs[r], group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, i_max, converge, nsweeps, comp, nind,
                                       stream, nsample, seed_index, r == 0 ? cg_iters : nullptr, n_not_spd ? &bad : nullptr, accepted ? acc.data() : nullptr))
                return bubble(ctxs[r]);
            if (n_not_spd) *n_not_spd += bad;
            add_acc();
        }
        return 0;
    }
    const int map_n = (flag == DANGX_FLAG_T) ? 1 : (flag == DANGX_FLAG_Q) ? 2 : (flag == DANGX_FLAG_U) ? 3 : (flag == DANGX_FLAG_QU) ? -1 : 0;
    if (map_n == 0) return fail(c0, "flag must be exactly one of T(1), Q(2), U(4), Q+U(8)");
    if (ml_mode != DANGX_ML_SAMPLE && ml_mode != DANGX_ML_OPTIMIZE) return fail(c0, "bad ml_mode");
    if (fluct_mode != DANGX_FLUCT_REFERENCE && ml_mode == DANGX_ML_SAMPLE)
        return fail(c0, "groups with template / monopole / hi_fit members reproduce the reference's fluctuation term only");
    static const bool enabled = [] { const char* e = getenv("DANGX_FUSE"); return !(e && e[0] == '0'); }();
    const int s1 = (map_n == -1) ? 2 : map_n, s2 = (map_n == -1) ? 3 : map_n;
    std::vector<GroupArgs> as((size_t)nctx), gs((size_t)nctx);
    std::vector<SweepList> sls((size_t)nctx);
    std::vector<long long> SNs((size_t)nctx);
    std::vector<int> lanes((size_t)nctx, 0);
    bool fuse = enabled && nsweeps <= 2 * DX_MAX_SWEEPS;
    for (int r = 0; r < nctx; ++r) {
        dangx_ctx* c = ctxs[r];
        (void)hipSetDevice(c->device);
        for (int s = 0; s < nsweeps; ++s)
            if (check_comp(c, comp[s]) || nind[s] < 0 || nind[s] >= c->desc[comp[s]].nindices) { fail(c, "sweep list: component / index out of range"); return bubble(c); }
        if (make_group(c, group, flag, as[r]) || sync_model(c)) return bubble(c);
        as[r].ml_mode = ml_mode; as[r].fluct = fluct_mode; as[r].seed = seed_amp; as[r].stream = stream_amp;
        SNs[r] = (long long)flag_planes_h(flag) * c->hm.npix;
        for (int k = s1; k <= s2; ++k) {
            c->chi_before_valid[k - 1] = c->chi_after_valid[k - 1] = c->touched_since_amp[k - 1] = false;
            for (int g = 0; g < as[r].ng; ++g) c->plane_nz[as[r].gc[g]] |= 1u << (k - 1);
        }
        if (as[r].nglob != as[0].nglob || as[r].nt != as[0].nt) return fail(c0, "the contexts disagree on the group's global-amplitude members");
        if (fuse) {
            gs[r] = as[r];
            std::memset(&sls[r], 0, sizeof(SweepList));
            fuse = planeset_group(c, gs[r], s1, s2, 2) && planeset_items(c, gs[r], map_n, nsweeps, comp, nind, stream, sls[r]);
            if (fuse) {
                sls[r].nsample = nsample; sls[r].ml_mode = ml_mode; sls[r].seed = seed_index;
                lanes[r] = dx_planeset_lanes(c, gs[r], sls[r], 1);
                fuse = lanes[r] != 0;
            }
        }
    }
    int nullity = 0, defer = fuse ? 1 : 0;
    if (device_schur(ctxs, nctx, as.data(), SNs.data(), n_not_spd, &nullity, &defer)) return 1;
    if (cg_iters) *cg_iters = -nullity;
    for (int r = 0; r < nctx; ++r) {
        dangx_ctx* c = ctxs[r];
        (void)hipSetDevice(c->device);
        // (pass 1 has counted the units whose block is not positive definite: the launch's own count of the same units is not read)
        const int rc = defer ? planeset_launch(c, gs[r], sls[r], lanes[r], 1, nullptr, accepted ? acc.data() : nullptr)
                             : dangx_plane_sweeps_sample(c, flag, nsweeps, comp, nind, stream, nsample, ml_mode, seed_index, accepted ? acc.data() : nullptr);
        if (rc) return bubble(c);
        add_acc();
    }
    return 0;
}

// dangx_index_sample(comp[s], nind[s], map_n of the flag, ...) for s = 0 .. nsweeps-1: the passes of sample_spectral_parameters
// (src/dang_sample_mod.f90:40-75) that touch ONE plane set, in the reference's order -- what the two-call seam issues after
// sample_cg_groups has returned.  Where k_plane_set covers the model (every swept component an amplitude-sampled member of ONE CG
// group whose members are the only components on these planes, register-chain modes) the sweeps are one launch on the amplitudes
// in memory (SOLVE = 0: one staging of the maps, the residual kept between the sweeps); everything else IS those calls, with
// consecutive indices of a component through dangx_index_sample_pair.
int dangx_plane_sweeps_sample(dangx_ctx* ctx, int flag, int nsweeps, const int32_t* comp, const int32_t* nind, const uint64_t* stream,
                              int nsample, int ml_mode, uint64_t seed, int64_t* accepted) {
    DxRange rg_("dangx_plane_sweeps_sample");
    if (!ctx || nsweeps < 1 || !comp || !nind || !stream) return 1;
    (void)hipSetDevice(ctx->device);
    if (sync_model(ctx)) return 1;
    const int map_n = (flag == DANGX_FLAG_T) ? 1 : (flag == DANGX_FLAG_Q) ? 2 : (flag == DANGX_FLAG_U) ? 3 : (flag == DANGX_FLAG_QU) ? -1 : 0;
    if (map_n == 0) return fail(ctx, "flag must be exactly one of T(1), Q(2), U(4), Q+U(8)");
    for (int s = 0; s < nsweeps; ++s)
        if (check_comp(ctx, comp[s]) || nind[s] < 0 || nind[s] >= ctx->desc[comp[s]].nindices) return fail(ctx, "sweep list: component / index out of range");
    static const bool enabled = [] { const char* e = getenv("DANGX_FUSE"); return !(e && e[0] == '0'); }();
    bool can = enabled && nsweeps <= 2 * DX_MAX_SWEEPS && (ml_mode == DANGX_ML_SAMPLE || ml_mode == DANGX_ML_OPTIMIZE);
    const int group = ctx->desc[comp[0]].cg_group;
    for (int s = 0; can && s < nsweeps; ++s) can = ctx->desc[comp[s]].cg_group == group && ctx->desc[comp[s]].sample_amplitude;
    GroupArgs g;
    SweepList sl;
    int lanes = 0;
    if (can) {
        if (make_group(ctx, group, flag, g)) return 1;
        if (planeset_group(ctx, g, (map_n == -1) ? 2 : map_n, (map_n == -1) ? 3 : map_n, 0) && planeset_items(ctx, g, map_n, nsweeps, comp, nind, stream, sl)) {
            sl.nsample = nsample; sl.ml_mode = ml_mode; sl.seed = seed;
            g.ml_mode = ml_mode; g.fluct = DANGX_FLUCT_REFERENCE; g.seed = 0; g.stream = 0;
            lanes = dx_planeset_lanes(ctx, g, sl, 0);
        }
    }
    if (lanes) return planeset_launch(ctx, g, sl, lanes, 0, nullptr, accepted);
    int64_t acc = 0, acc2 = 0;
    for (int s = 0; s < nsweeps; ++s) {
        if (s + 1 < nsweeps && comp[s + 1] == comp[s] && nind[s + 1] == nind[s] + 1) {
            const int rc = dangx_index_sample_pair(ctx, comp[s], nind[s], map_n, nsample, ml_mode, seed, stream[s], stream[s + 1],
                                                   accepted ? &acc : nullptr, accepted ? &acc2 : nullptr);
            if (rc) return rc;
            if (accepted) { accepted[s] = acc; accepted[s + 1] = acc2; }
            ++s;
        } else {
            const int rc = dangx_index_sample(ctx, comp[s], nind[s], map_n, nsample, ml_mode, seed, stream[s], accepted ? &acc : nullptr);
            if (rc) return rc;
            if (accepted) accepted[s] = acc;
        }
    }
    return 0;
}

// chi^2 of planes pol_lo..pol_hi from the values fused into the index sweeps: which = 0 -> the state the
// amplitude phase left (captured by the first sweep on each plane), 1 -> the current state.  Fails (status 2)
// if some plane has not been covered by a sweep since its last amplitude update: use dangx_sky_model_chisq.
int dangx_chisq_cached_dev(dangx_ctx* ctx, int which, int pol_lo, int pol_hi, double* out_dev) {
    if (!ctx || !out_dev || (which != 0 && which != 1)) return 1;
    (void)hipSetDevice(ctx->device);
    if (pol_lo < 1 || pol_hi > ctx->dims.nmaps || pol_lo > pol_hi) return fail(ctx, "bad pol_type range");
    for (int k = pol_lo; k <= pol_hi; ++k)
        if (!(which ? ctx->chi_after_valid[k - 1] : ctx->chi_before_valid[k - 1])) {
            ctx->err = "cached chi^2 not available for plane " + std::to_string(k);
            return 2;
        }
    if (chi_flush(ctx)) return 1;
    hipLaunchKernelGGL(k_chi_from_cache, dim3(1), dim3(1), 0, ctx->stream, ctx->chi_cache, which, pol_lo, pol_hi, out_dev);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int dangx_chisq_cached(dangx_ctx* ctx, int which, int pol_lo, int pol_hi, double* chisq_sum) {
    if (!ctx || !chisq_sum) return 1;
    const int rc = dangx_chisq_cached_dev(ctx, which, pol_lo, pol_hi, ctx->scalars + 1);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(chisq_sum, ctx->scalars + 1, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

static int sky_chisq_launch(dangx_ctx* ctx, int pol_lo, int pol_hi, double* sky_d, double* res_d, double* chi_d, double* out_dev);

// ddata%chisq's sum for the CURRENT state at the least cost: planes whose sum the last sweeps left behind come from the cache,
// every other plane gets one explicit update_sky_model + compute_chisq pass over THAT plane, whose result is cached too (the
// two-call form of the main loop asks after every CG group: only the group's own planes have changed since the last answer)
int dangx_chisq_current(dangx_ctx* ctx, int pol_lo, int pol_hi, double* chisq_sum) {
    DxRange rg_("dangx_chisq_current");
    if (!ctx || !chisq_sum) return 1;
    (void)hipSetDevice(ctx->device);
    if (pol_lo < 1 || pol_hi > ctx->dims.nmaps || pol_lo > pol_hi) return fail(ctx, "bad pol_type range");
    if (sync_model(ctx) || chi_flush(ctx)) return 1;
    for (int k = pol_lo; k <= pol_hi; ++k)
        if (!ctx->chi_after_valid[k - 1]) {
            if (sky_chisq_launch(ctx, k, k, nullptr, nullptr, nullptr, ctx->chi_cache + 3 + (k - 1))) return 1;
            ctx->chi_after_valid[k - 1] = true;
        }
    double v[3] = {0.0, 0.0, 0.0};
    HIPCHK(ctx, hipMemcpyAsync(v, ctx->chi_cache + 3, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double s = 0.0;
    for (int k = pol_lo; k <= pol_hi; ++k) s += v[k - 1];
    *chisq_sum = s;
    return 0;
}

static int sky_chisq_launch(dangx_ctx* ctx, int pol_lo, int pol_hi, double* sky_d, double* res_d, double* chi_d, double* out_dev) {
    if (pol_lo < 1 || pol_hi > ctx->dims.nmaps || pol_lo > pol_hi) return fail(ctx, "bad pol_type range");
    int bs = 256;
    while (bs > 64 && (size_t)ctx->hm.nbands * bs * sizeof(double) > 32 * 1024) bs >>= 1;
    const unsigned nblk = nblocks(ctx->hm.npix, bs);
    constexpr int RSTAGE = 128;
    // delta bandpasses, diffuse components, no maps asked for: one launch per plane on the amplitude kernel's schedule
    // (k_chisq_reg, dangx_ampreg.hip), the planes' block partials side by side and summed together
    if (!sky_d && !res_d && !chi_d) {   // (dx_launch_chisq_reg says -1 for what it does not cover)
        const unsigned nb256 = nblocks(ctx->hm.npix);
        const int npl = pol_hi - pol_lo + 1;
        if (ensure_partial(ctx, (long long)npl * nb256 + RSTAGE)) return 1;
        bool all = true;
        {
            Timed t(ctx, DANGX_K_SKY_CHISQ);
            for (int k = pol_lo; k <= pol_hi && all; ++k) all = dx_launch_chisq_reg(ctx, k, ctx->partial + (long long)(k - pol_lo) * nb256) == 0;
        }
        if (all) {
            Timed t(ctx, DANGX_K_REDUCE);
            double* stage = ctx->partial + (long long)npl * nb256;
            dx_reduce_two_stage(ctx, ctx->partial, (long long)npl * nb256, stage, out_dev);
            HIPCHK(ctx, hipGetLastError());
            return 0;
        }
    }
    if (ensure_partial(ctx, (long long)nblk + RSTAGE)) return 1;
    {
        Timed t(ctx, DANGX_K_SKY_CHISQ);
        hipLaunchKernelGGL(k_sky_chisq, dim3(nblk), dim3(bs), (size_t)ctx->hm.nbands * bs * sizeof(double), ctx->stream,
                           ctx->dm, pol_lo, pol_hi, sky_d, res_d, chi_d, ctx->partial);
    }
    {
        Timed t(ctx, DANGX_K_REDUCE);
        double* stage = ctx->partial + nblk;
        dx_reduce_two_stage(ctx, ctx->partial, (long long)nblk, stage, out_dev);
    }
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int dangx_sky_model_chisq_dev(dangx_ctx* ctx, int pol_lo, int pol_hi, double* chisq_sum_dev) {
    if (!ctx || !chisq_sum_dev) return 1;
    (void)hipSetDevice(ctx->device);
    if (sync_model(ctx)) return 1;
    return sky_chisq_launch(ctx, pol_lo, pol_hi, nullptr, nullptr, nullptr, chisq_sum_dev);
}

int dangx_sky_model_chisq(dangx_ctx* ctx, int pol_lo, int pol_hi, double* chisq_sum, double* sky, double* res, double* chi_map) {
    DxRange rg_("dangx_sky_model_chisq");
    if (!ctx) return 1;
    (void)hipSetDevice(ctx->device);
    if (sync_model(ctx)) return 1;
    const size_t nmap = (size_t)ctx->dims.npix * ctx->dims.nmaps * sizeof(double);
    const size_t nall = nmap * ctx->dims.nbands;
    double *sky_d = nullptr, *res_d = nullptr, *chi_d = nullptr;
    int rc = 0;
    if (sky) HIPCHK(ctx, hipMalloc(&sky_d, nall));
    if (res) HIPCHK(ctx, hipMalloc(&res_d, nall));
    if (chi_map) { HIPCHK(ctx, hipMalloc(&chi_d, nmap)); HIPCHK(ctx, hipMemsetAsync(chi_d, 0, nmap, ctx->stream)); }
    rc = sky_chisq_launch(ctx, pol_lo, pol_hi, sky_d, res_d, chi_d, ctx->scalars);
    if (!rc) {
        double v = 0.0;
        const size_t nplanes = (size_t)ctx->dims.nmaps * ctx->dims.nbands;
        if (hipMemcpyAsync(&v, ctx->scalars, sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = 1;
        if (sky && copy_planes(ctx, sky, sky_d, nplanes, false)) rc = 1;
        if (res && copy_planes(ctx, res, res_d, nplanes, false)) rc = 1;
        if (chi_map && copy_planes(ctx, chi_map, chi_d, (size_t)ctx->dims.nmaps, false)) rc = 1;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = 1;
        if (rc) ctx->err = "copy-back failed in dangx_sky_model_chisq";
        if (chisq_sum) *chisq_sum = v;
    }
    if (sky_d) (void)hipFree(sky_d);
    if (res_d) (void)hipFree(res_d);
    if (chi_d) (void)hipFree(chi_d);
    return rc;
}



}  // extern "C"
