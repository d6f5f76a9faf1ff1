// dangx_mixed.hip -- amplitude phase of CG groups with global-amplitude members (template / monopole / hi_fit):
// the mixed operators of the reference's CG and the direct Schur-complement solve.
#include "dx_ampdata.h"

namespace {

// ---------------------------------------------------------------------------
// CG building blocks for groups that contain global-amplitude components (template / monopole / hi_fit;
// src/dang_cg_mod.f90:522-587, 717-768, 833-893, 1045-1096).  A global component contributes one row per fitted
// band: its entries of T^t(...) are sums over pixels, formed here as block partials `rowpartial[row][block]`
// (second stage: k_reduce_rows_final).  Restrictions checked on the host: global members follow the diffuse
// ones in the group; hi_fit / monopole only under flag T (they read plane 1 whatever the flag in the reference).
template <int NG>
__global__ __launch_bounds__(BLOCK) void k_Ax_mixed(const Model* __restrict__ Mp, GroupArgs a, const double* __restrict__ x,
                                                    double* __restrict__ res, double* __restrict__ rowpartial) {
    __shared__ double sh[BLOCK / 64];
    constexpr int NA = NG > 0 ? NG : 1;
    const Model& M = *Mp;
    const int npix = M.npix;
    const long long SN = (long long)flag_nplanes(a.flag) * npix;
    const long long u = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const UnitId q = unit_of(M, a.flag);
    const double* xg = x + (long long)NG * SN;
    double acc[NA], xv[NA], mrow[NA];
    Prep pr[NA], prt[MAXT];
#pragma unroll
    for (int g = 0; g < NA; ++g) { acc[g] = 0.0; xv[g] = 0.0; pr[g] = Prep{0, 0, 0}; }
    if (!q.msk) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            xv[g] = x[(long long)g * SN + u];
            double t0, t1;
            load_theta(M, M.comp[a.gc[g]], q.i, q.k, t0, t1);
            pr[g] = sed_prep(M.comp[a.gc[g]], t0, t1);
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < a.nt) {
                double t0, t1;
                load_theta(M, M.comp[a.tc[t]], q.i, q.k, t0, t1);
                prt[t] = sed_prep(M.comp[a.tc[t]], t0, t1);
            }
    }
    for (int j = 0; j < M.nbands; ++j) {
        double temp1 = 0.0, st[MAXT];
        if (!q.msk) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                mrow[g] = sed_eval(M, M.comp[a.gc[g]], j, pr[g]);
                temp1 = temp1 + xv[g] * mrow[g];  // :697-704
            }
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            st[t] = 0.0;
            if (t < a.nt) {
                const Comp& c = M.comp[a.tc[t]];
                if (((c.corr_mask >> j) & 1) && !q.msk && q.p < gl_nplanes(c, a.flag)) {
                    const int lt = __popc(c.corr_mask & ((1 << j) - 1));
                    st[t] = comp_sed(M, c, q.i, q.k, j, prt[t]);
                    temp1 = temp1 + xg[a.trow[t] + lt] * st[t];  // :723, :737, :752-759
                }
            }
        }
        if (!q.msk) {
            const double rms = M.rms[((long long)j * M.nmaps + (q.k - 1)) * npix + q.i];
            temp1 = temp1 / (rms * rms);  // :775-791
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = acc[g] + temp1 * mrow[g];  // :813-820
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < a.nt) {
                const Comp& c = M.comp[a.tc[t]];
                if ((c.corr_mask >> j) & 1) {  // uniform: every thread joins the row sum
                    const int lt = __popc(c.corr_mask & ((1 << j) - 1));
                    const bool on = !q.msk && q.p < gl_nplanes(c, a.flag);
                    // :857 the monopole row sums temp1(i) WITHOUT its template factor
                    const double v = on ? temp1 * ((c.type == DANGX_MONOPOLE) ? 1.0 : st[t]) : 0.0;
                    block_row_sum(v, a.trow[t] + lt, rowpartial, sh);
                }
            }
    }
    if (q.in) {
#pragma unroll
        for (int g = 0; g < NG; ++g) res[(long long)g * SN + u] = acc[g];
    }
}

template <int NG>
__global__ __launch_bounds__(BLOCK) void k_rhs_mixed(const Model* __restrict__ Mp, GroupArgs a, double* __restrict__ b,
                                                     double* __restrict__ rowpartial) {
    __shared__ double sh[BLOCK / 64];
    constexpr int NA = NG > 0 ? NG : 1;
    const Model& M = *Mp;
    const int npix = M.npix;
    const long long SN = (long long)flag_nplanes(a.flag) * npix;
    const long long u = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const UnitId q = unit_of(M, a.flag);
    const bool zero_mask = !q.in || M.mask[q.i] == 0.0;  // :474 tests ==0 only for the diffuse rows
    double acc[NA];
    Prep pr[NA], prt[MAXT];
#pragma unroll
    for (int g = 0; g < NA; ++g) { acc[g] = 0.0; pr[g] = Prep{0, 0, 0}; }
    if (q.in) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double t0, t1;
            load_theta(M, M.comp[a.gc[g]], q.i, q.k, t0, t1);
            pr[g] = sed_prep(M.comp[a.gc[g]], t0, t1);
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < a.nt) {
                double t0, t1;
                load_theta(M, M.comp[a.tc[t]], q.i, q.k, t0, t1);
                prt[t] = sed_prep(M.comp[a.tc[t]], t0, t1);
            }
    }
    for (int j = 0; j < M.nbands; ++j) {
        double d = 0.0, rms = 1.0;
        if (q.in) {
            d = M.sig[((long long)j * M.nmaps + (q.k - 1)) * npix + q.i];
            if (q.k == 1) d = d / M.gain[j];  // :371
            rms = M.rms[((long long)j * M.nmaps + (q.k - 1)) * npix + q.i];
            if (!q.msk) {
                d = remove_others(M, a, q.i, q.k, j, d);  // :427-460
            }
            if (!zero_mask) {
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = acc[g] + (d * sed_eval(M, M.comp[a.gc[g]], j, pr[g])) / (rms * rms);  // :489-508
            }
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < a.nt) {
                const Comp& c = M.comp[a.tc[t]];
                if ((c.corr_mask >> j) & 1) {
                    const int lt = __popc(c.corr_mask & ((1 << j) - 1));
                    const bool on = !q.msk && q.p < gl_nplanes(c, a.flag);
                    const double v = on ? d / (rms * rms) * comp_sed(M, c, q.i, q.k, j, prt[t]) : 0.0;  // :531, :550, :571-578
                    block_row_sum(v, a.trow[t] + lt, rowpartial, sh);
                }
            }
    }
    if (q.in) {
#pragma unroll
        for (int g = 0; g < NG; ++g) b[(long long)g * SN + u] = acc[g];
    }
}

template <int NG>
__global__ __launch_bounds__(BLOCK) void k_sv_mixed(const Model* __restrict__ Mp, GroupArgs a, const double* __restrict__ eta,
                                                    double* __restrict__ res, double* __restrict__ rowpartial) {
    __shared__ double sh[BLOCK / 64];
    const Model& M = *Mp;
    const int npix = M.npix;
    const long long SN = (long long)flag_nplanes(a.flag) * npix;
    const long long u = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const UnitId q = unit_of(M, a.flag);
    double acc = 0.0;
    Prep prl = {0, 0, 0}, prt[MAXT];
    double e = 0.0;
    if (!q.msk) {
        e = eta[u];
        if (NG > 0) {
            double t0, t1;
            load_theta(M, M.comp[a.gc[NG > 0 ? NG - 1 : 0]], q.i, q.k, t0, t1);
            prl = sed_prep(M.comp[a.gc[NG > 0 ? NG - 1 : 0]], t0, t1);
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < a.nt) {
                double t0, t1;
                load_theta(M, M.comp[a.tc[t]], q.i, q.k, t0, t1);
                prt[t] = sed_prep(M.comp[a.tc[t]], t0, t1);
            }
    }
    int lrun = 0;  // ONE running row counter over bands and components (:970, :1057, :1071, :1094)
    for (int j = 0; j < M.nbands; ++j) {
        double temp1 = 0.0;
        if (!q.msk) {
            temp1 = e / M.rms[((long long)j * M.nmaps + (q.k - 1)) * npix + q.i];  // :1008-1015
            if (NG > 0) acc = acc + temp1 * sed_eval(M, M.comp[a.gc[NG > 0 ? NG - 1 : 0]], j, prl);  // :1033-1040 (quirk 2)
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
            if (t < a.nt) {
                const Comp& c = M.comp[a.tc[t]];
                if ((c.corr_mask >> j) & 1) {
                    const bool on = !q.msk && q.p < gl_nplanes(c, a.flag);
                    const double v = on ? temp1 * ((c.type == DANGX_MONOPOLE) ? 1.0 : comp_sed(M, c, q.i, q.k, j, prt[t])) : 0.0;
                    if (lrun < a.nglob) block_row_sum(v, lrun, rowpartial, sh);  // (the reference would run out of bounds)
                    ++lrun;
                }
            }
    }
    if (q.in) {
        if (NG > 0) res[u] = acc;
#pragma unroll
        for (int g = 1; g < NG; ++g) res[(long long)g * SN + u] = 0.0;
    }
}

template <int NG>
struct LaunchMixed {
    static int run(dangx_ctx* ctx, const GroupArgs& a, long long SN, int what, const double* in, double* out) {
        const unsigned nblk = nblocks(SN);
        if (ensure_partial(ctx, (long long)std::max(a.nglob, 1) * nblk)) return 1;
        double* og = out + (long long)NG * SN;
        // rows that no kernel writes (sample vector: rows beyond the running counter) must read as zero
        HIPCHK(ctx, hipMemsetAsync(ctx->partial, 0, sizeof(double) * (size_t)std::max(a.nglob, 1) * nblk, ctx->stream));
        {
            Timed t(ctx, what == 1 ? DANGX_K_CG_AX : DANGX_K_CG_VEC);
            if (what == 0) hipLaunchKernelGGL(k_rhs_mixed<NG>, dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, out, ctx->partial);
            else if (what == 1) hipLaunchKernelGGL(k_Ax_mixed<NG>, dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, in, out, ctx->partial);
            else hipLaunchKernelGGL(k_sv_mixed<NG>, dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, in, out, ctx->partial);
        }
        dx_reduce_rows_to(ctx, ctx->partial, nblk, a.nglob, og);
        HIPCHK(ctx, hipGetLastError());
        return 0;
    }
};

int dispatch_mixed(dangx_ctx* ctx, const GroupArgs& a, long long SN, int what, const double* in, double* out) {
    switch (a.ng) {
    case 0: return LaunchMixed<0>::run(ctx, a, SN, what, in, out);
    case 1: return LaunchMixed<1>::run(ctx, a, SN, what, in, out);
    case 2: return LaunchMixed<2>::run(ctx, a, SN, what, in, out);
    case 3: return LaunchMixed<3>::run(ctx, a, SN, what, in, out);
    case 4: return LaunchMixed<4>::run(ctx, a, SN, what, in, out);
    case 5: return LaunchMixed<5>::run(ctx, a, SN, what, in, out);
    case 6: return LaunchMixed<6>::run(ctx, a, SN, what, in, out);
    case 7: return LaunchMixed<7>::run(ctx, a, SN, what, in, out);
    case 8: return LaunchMixed<8>::run(ctx, a, SN, what, in, out);
    default: return fail(ctx, "unsupported group size");
    }
}

}  // namespace
int dx_launch_rhs_mixed(dangx_ctx* ctx, const GroupArgs& a, long long SN, double* b) { return dispatch_mixed(ctx, a, SN, 0, nullptr, b); }
int dx_launch_Ax_mixed(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* x, double* res) { return dispatch_mixed(ctx, a, SN, 1, x, res); }
int dx_launch_sv_mixed(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* eta, double* res) { return dispatch_mixed(ctx, a, SN, 2, eta, res); }
