// dx_host.h -- host-side context and launch-argument structs shared by the translation units of libdangx.so
// (dangx_core.hip: context, model, chi^2 ring, host solvers, small kernels; dangx_entry.hip: the sampling entry points;
// dangx_coarse.hip: full-sky / coarse-Nside device side; dangx_amp.hip: amplitude kernels of diffuse groups; dangx_mixed.hip /
// dangx_schur.hip: groups with template-type members (mixed CG operators / direct Schur solve); dangx_mh.hip: LDS-form
// Metropolis kernels; dangx_mhreg.hip: register-resident Metropolis kernels, compiled once per chain mode).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dangx.h"
#include "dx_args.h"

struct dangx_ctx {
    dangx_dims dims{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // host mirror of the model + device copy
    Model hm{};
    Model* dm = nullptr;
    bool dirty = true;
    bool comp_set[MAXC] = {};
    bool band_set[MAXB] = {};
    dangx_comp_desc desc[MAXC] = {};
    // owned device buffers
    double *sig = nullptr, *rms = nullptr, *mask = nullptr;
    bool own_data = false;
    double* amp[MAXC] = {};
    double* idx[MAXC] = {};
    bool own_amp[MAXC] = {};
    bool own_idx[MAXC] = {};
    unsigned plane_nz[MAXC] = {};  // bit k-1: amplitude plane k of the component may be non-zero
    double* tmpl[MAXC] = {};       // device c%template of global-amplitude components
    int corr_mask[MAXC] = {}, nfit[MAXC] = {};
    double tamp[MAXC][3][MAXB] = {};  // c%template_amplitudes, host mirror [map][band]
    unsigned tmpl_nz[MAXC] = {};   // bit k-1: plane k of c%template has a non-zero pixel (a template without one has no signal there)
    unsigned tmpl_one[MAXC] = {};  // bit k-1: plane k of c%template is identically 1 (a monopole's, src/dang_component_mod.f90:593-595)
    unsigned idx_const[MAXC] = {}; // bit k-1: every index of the component is spatially constant on plane k
    unsigned qu_equal[MAXC] = {};  // bit q: index map q of the component is equal on the Q and U planes for every pixel
    double idx_val[MAXC][3][MAXI] = {};
    // masked sums of the index maps (dangx_index_masked_sums: what write_stats_to_term prints after EVERY phase,
    // src/dang_data_mod.f90:540-567) are kept until something writes the map or the mask: an amplitude phase does not move them
    bool idxsum_ok[MAXC][MAXI][3] = {};
    double idxsum[MAXC][MAXI][3] = {};
    long long idxcnt[MAXC][MAXI][3] = {};
    bool idx_ext[MAXC] = {};       // the maps live in a buffer of the caller's (dangx_adopt_device_state): never cached
    // ... and the plane-set launches leave the sums of the maps they swept on the DEVICE (chi_cache, after the six chi^2 slots),
    // as by-products like chi^2: valid until the map is written again; the count of unmasked pixels is a property of the mask
    bool idxsum_dev[MAXC][MAXI][3] = {};
    long long mask_count = -1;
    std::vector<double> bp_nu0, bp_tau0;
    double *d_bp_nu0 = nullptr, *d_bp_tau0 = nullptr, *d_bp_lnr = nullptr;  // lnr: [ncomp][samples]
    bool bp_dirty = true;  // bandpass samples or component reference frequencies changed since the last upload
    // scratch
    double* partial = nullptr;
    long long partial_cap = 0;
    double* scalars = nullptr;              // device scalars [8]
    double* chi_cache = nullptr;            // device [6]: chi^2 before/after of planes 1..3 (fused in k_index_mh)
    double chi_host[3] = {0.0, 0.0, 0.0};   // host staging of dx_set_chi_after
    // block partials of the sweeps' chi^2 sums wait here until somebody asks for a value (dangx_chisq_cached) or the ring
    // is full: ONE pair of reduction launches then serves every sweep since the last one (in launch order), instead of
    // two small launches behind every sweep -- 6 % of a rank's iteration at the 8-rank shard size
    static constexpr int CHI_RING = 8;
    // ns > 0: rows 4 .. 4+ns-1 of the buffer are block partials of the masked sums of the index maps the launch swept;
    // slot[q] = their place in chi_cache (idx_slot)
    struct ChiPend { double* buf = nullptr; long long cap = 0, nblk = 0; int s1 = 0, s2 = 0, wb = 0, ns = 0; int slot[DX_MAX_IDXSUM] = {}; } chi_pend[CHI_RING];
    int chi_npend = 0;
    double* chi_stage = nullptr;
    bool chi_before_valid[3] = {}, chi_after_valid[3] = {}, touched_since_amp[3] = {};
    unsigned long long* counters = nullptr; // device counters [4]
    long long host_stride = 0;              // doubles between consecutive planes of host map arrays (0 = npix: packed)
    dangx_allreduce_fn allreduce = nullptr; // sum over ranks of host doubles (pixel-sharded runs); null = single rank
    void* allreduce_user = nullptr;
    bool is_root = true;
    double* work[6] = {};                   // CG vectors
    double* fs_data = nullptr;              // full-sky mode: cleaned data [Sp][nb][npix]
    long long fs_cap = 0;
    int fs_comp = -1, fs_s1 = 0, fs_s2 = 0;
    unsigned fs_others = 0;                 // components dangx_fullsky_prepare removes from the data
    bool fs_lazy = false;                   // fs_data not written yet (dx_fullsky_prepare_lazy): dangx_fullsky_sums fills it when a sum needs it
    long long fs_npc = 0;                   // > 0: the full-sky sums run over the degraded maps (cs_*) of that many pixels
    double* rows_out = nullptr;             // device [2*MAXB*2 + 8] row sums
    // coarse-Nside index sampling: HEALPix RING<->NEST maps of both resolutions + degraded data / rms / mask
    int hp_nside = 0, hp_cnside = 0;
    int *hp_n2r_f = nullptr, *hp_r2n_f = nullptr, *hp_n2r_c = nullptr, *hp_r2n_c = nullptr;
    double *cs_data = nullptr, *cs_rms = nullptr, *cs_mask = nullptr, *cs_index = nullptr;
    long long cs_cap = 0;
    // the degraded rms / mask of a plane set do not change between sweeps (the maps are the run's input): two kept copies, keyed by
    // (planes, nside, sample_nside, generation of the map data); a hit replaces two degrade passes by two device copies
    struct CsKept { int s1 = 0, s2 = 0, nside = 0, sample_nside = 0; long long gen = -1, cap = 0, capm = 0, stamp = 0; double *rms = nullptr, *mask = nullptr; };
    CsKept cs_kept[2];
    long long data_gen = 0, cs_stamp = 0;   // data_gen: bumped whenever sig / rms / mask are replaced or rescaled
    double* cs_part = nullptr;              // per-shard sums / counts of the degrade step (pixel-sharded coarse sampling)
    long long cs_part_cap = 0;
    long long work_cap = 0;
    // last DANGX_SOLVER_DIRECT solve of a group with global-amplitude members: largest |b - A x| of a global row relative
    // to that row of b after the last refinement, and the number of refinement steps taken
    double schur_resid = 0.0, schur_backward = 0.0;  // relative to the row of b / to the size of the row's terms
    // dangx_amp_index_sample: an amplitude solve whose launch waits for the index sweep it is fused with
    bool defer_amp = false, have_pending = false;
    // dangx_index_sample_pair: the sweep of index nind is launched together with the sweep of index nind + 1
    bool pair_on = false, pair_done = false;
    unsigned long long pair_stream = 0;
    GroupArgs pending{};
    long long pending_SN = 0;
    int schur_refine = 0;
    // kernels specialised at run time for this context's model (dangx_rtc.hip)
    std::vector<std::pair<std::string, hipFunction_t>> rtc_fns;
    std::vector<hipModule_t> rtc_mods;
    std::vector<std::string> rtc_failed;
    // profiling
    bool prof = false;
    struct Ev { hipEvent_t a, b; int kid, planes; };
    std::vector<Ev> events;
    double prof_ms[DANGX_K_COUNT] = {};
    long long prof_n[DANGX_K_COUNT] = {};
    double prof_ms_pl[DANGX_K_COUNT][3] = {};      // the same by the number of planes the launch worked on (0: not recorded)
    long long prof_n_pl[DANGX_K_COUNT][3] = {};
};

// roctx ranges around every timed launch group and the entry points that issue them ("rocprof/roctx ranges
// inside the library"): DANGX_ROCTX=1 loads the ROCm marker library at run time (no link-time dependency; `rocprofv3
// --marker-trace` then shows the ranges), otherwise the two calls are no-ops.
struct DxRoctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    DxRoctx();
};
const DxRoctx& dx_roctx();
struct DxRange {
    bool on;
    explicit DxRange(const char* name) : on(dx_roctx().push != nullptr) { if (on) (void)dx_roctx().push(name); }
    ~DxRange() { if (on) (void)dx_roctx().pop(); }
    DxRange(const DxRange&) = delete;
    DxRange& operator=(const DxRange&) = delete;
};
const char* dx_kernel_family(int kid);

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

// the chi^2 sums cached by the index sweeps describe a model that no longer exists: every setter that changes the
// model (T_CMB, calibration, host pushes of state, new data, new descriptors) calls this
inline void invalidate_chi(dangx_ctx* ctx) {
    for (int k = 0; k < 3; ++k) ctx->chi_before_valid[k] = ctx->chi_after_valid[k] = false;
    if (ctx->chi_npend) {   // the dropped entries carry the index sums of their launches too: those fall back to the explicit pass
        for (int l = 0; l < MAXC; ++l)
            for (int q = 0; q < MAXI; ++q) for (int k = 0; k < 3; ++k) ctx->idxsum_dev[l][q][k] = false;
    }
    ctx->chi_npend = 0;  // block partials still waiting in the ring belong to that model too: nobody may read them
}

// an index map of component `comp` (comp < 0: any map, or the mask) is about to change: its cached masked sums are stale
inline void idx_written(dangx_ctx* ctx, int comp) {
    for (int l = 0; l < MAXC; ++l)
        if (comp < 0 || l == comp)
            for (int q = 0; q < MAXI; ++q) for (int k = 0; k < 3; ++k) ctx->idxsum_ok[l][q][k] = ctx->idxsum_dev[l][q][k] = false;
    if (comp < 0) ctx->mask_count = -1;
}
// place of the masked sum of c%indices(:, plane, nind) of component comp in chi_cache
inline int idx_slot(int comp, int nind, int plane) { return 6 + (comp * MAXI + nind) * 3 + (plane - 1); }
constexpr int CHI_CACHE_DOUBLES = 6 + MAXC * MAXI * 3;

inline int fail(dangx_ctx* ctx, const std::string& msg) {
    ctx->err = msg;
    return 1;
}

struct Timed {
    dangx_ctx* ctx;
    dangx_ctx::Ev ev{};
    bool on;
    DxRange range;
    Timed(dangx_ctx* c, int kid, int planes = 0) : ctx(c), on(c->prof), range(dx_kernel_family(kid)) {
        if (!on) return;
        ev.kid = kid; ev.planes = (planes >= 0 && planes <= 2) ? planes : 0;
        (void)hipEventCreate(&ev.a);
        (void)hipEventCreate(&ev.b);
        (void)hipEventRecord(ev.a, ctx->stream);
    }
    ~Timed() {
        if (!on) return;
        (void)hipEventRecord(ev.b, ctx->stream);
        ctx->events.push_back(ev);
    }
};


inline int ensure_partial(dangx_ctx* ctx, long long n) {
    if (n <= ctx->partial_cap) return 0;
    if (ctx->partial) (void)hipFree(ctx->partial);
    HIPCHK(ctx, hipMalloc(&ctx->partial, sizeof(double) * (size_t)n));
    ctx->partial_cap = n;
    return 0;
}

inline int flag_planes_h(int flag) { return (flag & DANGX_FLAG_QU) ? 2 : 1; }

inline unsigned nblocks(long long n, int bs = BLOCK) { return (unsigned)((n + bs - 1) / bs); }

// out[row] = sum(partial[row][0..nblk)) for rows 0..rows-1 (deterministic; defined in dangx_core.hip)
void dx_reduce_rows_to(dangx_ctx* ctx, const double* partial, unsigned nblk, int rows, double* out_dev);
// out_dev[0] = sum(partial[0..n)) in two deterministic stages; stage: DX_RSTAGE doubles of scratch behind the partials
constexpr int DX_RSTAGE = 128;
void dx_reduce_two_stage(dangx_ctx* ctx, const double* partial, long long n, double* stage, double* out_dev);

// ---- host helpers of dangx_core.hip used by the other translation units of the ABI
int sync_model(dangx_ctx* ctx);                 // host model -> device copy when something changed (constant-index rows, bandpass tables)
int prof_collect(dangx_ctx* ctx);
int ensure_work(dangx_ctx* ctx, long long n);
int ensure_state(dangx_ctx* ctx, int comp);     // allocate a component's maps on first use
int check_comp(dangx_ctx* ctx, int comp);
int copy_planes(dangx_ctx* ctx, void* dst, const void* src, size_t planes, bool to_device);
int make_group(dangx_ctx* ctx, int group, int flag, GroupArgs& a);
int dx_set_chi_after(dangx_ctx* ctx, int k, double chi);   // dangx_entry.hip
int chi_flush(dangx_ctx* ctx);                  // reduce every pending launch's chi^2 / index-sum partials into chi_cache
int chi_next(dangx_ctx* ctx, long long nblk, double** buf);
int reduce_to_host(dangx_ctx* ctx, long long nblk, double* out);
int rank_sum(dangx_ctx* ctx, double* buf, int64_t n);
int device_cg(dangx_ctx* ctx, const GroupArgs& a, int i_max, double converge, int* iters);
extern "C" int dx_fullsky_prepare_lazy(dangx_ctx* ctx, int comp, int map_n);   // dangx_fullsky_prepare without the staging pass (dangx_coarse.hip)
double dx_host_band_sed(dangx_ctx* ctx, int comp, int j, double t0, double t1);   // eval_sed of a diffuse component, host side
int device_schur(dangx_ctx* const* cs, int nc, const GroupArgs* as, const long long* SNs, int64_t* n_not_spd, int* nullity, int* defer = nullptr);
// map_n of sample_index_mh (src/dang_sample_mod.f90:53-64) -> first and last map plane
inline int map_planes(dangx_ctx* ctx, int map_n, int& s1, int& s2) {
    if (map_n == -1) { s1 = 2; s2 = 3; }
    else if (map_n >= 1 && map_n <= 3) { s1 = s2 = map_n; }
    else return fail(ctx, "There is something wrong with the poltype flag (map_n must be 1,2,3 or -1)");
    if (s2 > ctx->dims.nmaps) return fail(ctx, "map_n exceeds nmaps");
    return 0;
}

// run-time specialisation (dangx_rtc.hip): the kernel named by a template-id of `header`, or nullptr (+ ctx->err)
bool dx_rtc_enabled();
hipFunction_t dx_rtc_get(dangx_ctx* ctx, const char* header, const std::string& name_expr);
int dx_rtc_launch(dangx_ctx* ctx, hipFunction_t fn, unsigned nblk, size_t lds, void** args);
void dx_rtc_release(dangx_ctx* ctx);

// launchers defined next to their kernels
int dx_launch_amp(dangx_ctx* ctx, const GroupArgs& a, long long SN);
// latency-hiding form of the direct solve (dangx_ampreg.hip): 0 = launched, -1 = case not covered
int dx_launch_amp_reg(dangx_ctx* ctx, const GroupArgs& a, long long SN);
int dx_launch_chisq_reg(dangx_ctx* ctx, int k, double* partial);  // dangx_ampreg.hip
int dx_launch_amp_reg_templates(dangx_ctx* ctx, const GroupArgs& a, long long SN);  // pass 2 of a template group's Schur solve
struct SchurArgs;
int dx_launch_schur_resid_reg(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev);  // its residual pass
int dx_launch_schur_pass1_reg(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev);  // its pass 1
int dx_launch_rhs(dangx_ctx* ctx, const GroupArgs& a, long long SN, double* b);
int dx_launch_Ax(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* x, double* res, double* part);
int dx_launch_sample_vector(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* eta, double* res);
// direct (Schur complement) solve of a group with global-amplitude members; rows <= DX_MAX_ROWS
constexpr int DX_MAX_ROWS = 32;
struct SchurArgs {
    int nrows;
    unsigned char rt[DX_MAX_ROWS], rj[DX_MAX_ROWS];  // row r <-> (global member rt[r], band rj[r])
    signed char ftarget[DX_MAX_ROWS];                // row that receives row r's fluctuation sum (running counter of
                                                     // compute_sample_vector, src/dang_cg_mod.f90:970-1094); -1 = dropped
    signed char bslot[MAXB];                         // LDS slot of band j (-1: no global row at that band)
    int nslots;
};
int dx_launch_schur_pass1(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev);
int dx_launch_schur_pass2(dangx_ctx* ctx, const GroupArgs& a, long long SN);
// rows_dev[0..R): global rows of b - A x at the current state (without the fluctuation term); [R..2R): those rows of b
int dx_launch_schur_resid(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev);
// groups with global-amplitude members (a.nt > 0): vectors are [diffuse | global rows]
int dx_launch_rhs_mixed(dangx_ctx* ctx, const GroupArgs& a, long long SN, double* b);
int dx_launch_Ax_mixed(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* x, double* res);
int dx_launch_sv_mixed(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* eta, double* res);
// LDS-form Metropolis kernel (fast = chisq likelihood with CH_POW / CH_MBB_*; otherwise the generic chain)
void dx_launch_mh_lds(dangx_ctx* ctx, const IndexArgs& a, bool fast, int Sp, unsigned nblk, int bs, size_t lds, unsigned long long* accp);
// register-resident Metropolis kernels; return false when (mode, nb) is not instantiated
bool dx_mh_reg_supported(dangx_ctx* ctx, int mode, int nb, int Sp);
bool dx_mh_pair_supported(int mode_a, int mode_b, int nb, int Sp);
bool dx_launch_mh_pair(dangx_ctx* ctx, const IndexArgs& a, const IndexArgs& b, int Sp, unsigned nblk, unsigned long long* accp);
bool dx_fused_supported(int mode, int nb, int ng);
int dx_fused_lanes(dangx_ctx* ctx, const GroupArgs& ga, const IndexArgs& a, int Sp);  // 0: the two launches
bool dx_launch_fused(dangx_ctx* ctx, const GroupArgs& ga, const IndexArgs& a, int Sp, int lanes, unsigned nblk, unsigned long long* accp);
int dx_mh_reg_lanes(int nb, int Sp);  // lanes per pixel of the register chain (dangx_mhreg.hip)
// a group's solve and every sweep on its planes in one launch (dangx_planeset.hip): lanes per pixel, 0 = the separate launches
int dx_planeset_lanes(dangx_ctx* ctx, const GroupArgs& ga, const SweepList& sl, int solve);
bool dx_launch_planeset(dangx_ctx* ctx, const GroupArgs& ga, const SweepList& sl, int lanes, int solve, unsigned nblk, unsigned long long* accp);
bool dx_launch_mh_reg(dangx_ctx* ctx, const IndexArgs& a, int Sp, unsigned nblk, unsigned long long* accp);
