// dangx_schur.hip -- direct (Schur complement) solve of CG groups with global-amplitude members.
#include "dx_ampdata.h"

namespace {

// ---------------------------------------------------------------------------
// Direct solve of a group with global-amplitude members (the MI355X counterpart of the reference's CG for the
// coupled system).  With x = [x_u per unit | g global rows] the matrix of compute_Ax is
//     [ D_u   B_u ] [x_u]   [b_u]        D_u = sum_j M_j M_j^t / s_j^2           (NG x NG, per unit)
//     [ C_u^t G   ] [ g ] = [b_g]        B_u[:,r] = M_jr s_r / s_jr^2,  C_u[:,r] = M_jr w_r / s_jr^2
// with s_r the global member's SED at the row's band and w_r its row weight (1 for a monopole, :857).
// Pass 1 eliminates every unit: with D_u = L L^t, Q_j = L^-1 (M_j/s_j^2) and yh = L^-1 (b_u + f_u) it accumulates
//     S[r,r'] = sum_u ( [j_r = j_r'] w_r s_r'/s_j^2 - w_r s_r' Q_jr . Q_jr' ),   t[r] = sum_u ( d_jr s_r/s_jr^2 - w_r Q_jr . yh )
// (and the fluctuation sums of the global rows) as deterministic block partials; the host solves S g = t; pass 2 is
// the per-unit block solve with the global members' new signal removed from the data.
template <int NG>
struct MixedUnit {
    double A[NG > 0 ? NG * (NG + 1) / 2 : 1], bv[NG > 0 ? NG : 1];  // NG == 0: a group of global members only
    bool ok;
};

// data prep of compute_rhs for groups with global members + per-unit normal equations.  When lds != nullptr the
// per-band vectors W_j = M_j/s_j^2 (NG), d_j/s_j^2 and eta/s_j of the bands that carry a global row are parked
// in LDS columns [(bslot[j]*(NG+2)+q)*BLOCK].
template <int NG, bool SUBTRACT_MEMBERS>
__device__ __forceinline__ void mixed_normal_eq(const Model& M, const GroupArgs& a, const UnitId& q, double* lds,
                                                const signed char* bslot, MixedUnit<NG>& U) {
    constexpr int NA = NG > 0 ? NG : 1;
    Prep pr[NA];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        double t0, t1;
        load_theta(M, M.comp[a.gc[g]], q.i, q.k, t0, t1);
        pr[g] = sed_prep(M.comp[a.gc[g]], t0, t1);
    }
#pragma unroll
    for (int e = 0; e < NG * (NG + 1) / 2; ++e) U.A[e] = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) U.bv[g] = 0.0;
    const bool sample = (a.ml_mode == DANGX_ML_SAMPLE);
    double eta = 0.0, f0 = 0.0;
    if (sample) {
        double u1, u2;
        uniform2(a.seed, a.stream, (unsigned long long)(M.pix0 + q.i), (uint32_t)q.k, u1, u2);
        eta = rand_normal(0.0, 1.0, u1, u2);
    }
    for (int j = 0; j < M.nbands; ++j) {
        double d = M.sig[((long long)j * M.nmaps + (q.k - 1)) * M.npix + q.i];
        if (q.k == 1) d = d / M.gain[j];
        d = remove_others(M, a, q.i, q.k, j, d);
        if (SUBTRACT_MEMBERS)
            for (int t = 0; t < a.nt; ++t) {
                const Comp& c = M.comp[a.tc[t]];
                if (((c.corr_mask >> j) & 1) && q.p < gl_nplanes(c, a.flag)) {
                    double t0, t1;
                    load_theta(M, c, q.i, q.k, t0, t1);
                    d = d - comp_signal(M, c, q.i, q.k, j, 0.0, sed_prep(c, t0, t1));
                }
            }
        const double is = 1.0 / M.rms[((long long)j * M.nmaps + (q.k - 1)) * M.npix + q.i];
        const double inv = is * is;
        double mrow[NA];
#pragma unroll
        for (int g = 0; g < NG; ++g) mrow[g] = sed_eval(M, M.comp[a.gc[g]], j, pr[g]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const double t2 = mrow[g] * inv;
            U.bv[g] += d * t2;
#pragma unroll
            for (int h = 0; h <= g; ++h) U.A[g * (g + 1) / 2 + h] += t2 * mrow[h];
            if (lds && bslot[j] >= 0) lds[(bslot[j] * (NG + 2) + g) * BLOCK] = t2;
        }
        if (NG > 0 && sample) f0 += (eta * is) * mrow[NG > 0 ? NG - 1 : 0];  // :1033-1040 (reference fluctuation term)
        if (lds && bslot[j] >= 0) {
            lds[(bslot[j] * (NG + 2) + NG) * BLOCK] = d * inv;
            lds[(bslot[j] * (NG + 2) + NG + 1) * BLOCK] = eta * is;
        }
    }
    if (NG > 0) U.bv[0] += f0;
    U.ok = true;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int h = 0; h <= g; ++h) {
            double s = U.A[g * (g + 1) / 2 + h];
#pragma unroll
            for (int t = 0; t < h; ++t) s -= U.A[g * (g + 1) / 2 + t] * U.A[h * (h + 1) / 2 + t];
            if (h == g) {
                if (!(s > 0.0)) U.ok = false;
                U.A[g * (g + 1) / 2 + g] = sqrt(s);
            } else {
                U.A[g * (g + 1) / 2 + h] = s / U.A[h * (h + 1) / 2 + h];
            }
        }
    }
}

// forward substitution v <- L^-1 v (packed lower L)
template <int NG>
__device__ __forceinline__ void fwd_subst(const double* A, double* v) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        double s = v[g];
#pragma unroll
        for (int t = 0; t < g; ++t) s -= A[g * (g + 1) / 2 + t] * v[t];
        v[g] = s / A[g * (g + 1) / 2 + g];
    }
}

// rowpartial rows: [0, R*R): S[r][r'] ; [R*R, R*R+R): t[r] ; [R*R+R, R*R+2R): fluctuation sum of natural row r ;
// [R*R+2R, R*R+3R): G[r][r], the diagonal before elimination (scale for the degeneracy test on the host)
template <int NG>
__global__ __launch_bounds__(BLOCK, NG <= 4 ? 3 : 1) void k_schur_pass1(const Model* __restrict__ Mp, GroupArgs a, SchurArgs sa,
                                                       double* __restrict__ rowpartial, unsigned long long* __restrict__ not_spd) {
    extern __shared__ double ldsall[];  // [nb*(NG+2)][BLOCK] columns + reduction scratch
    __shared__ double sh[BLOCK / 64];
    const Model& M = *Mp;
    const UnitId q = unit_of(M, a.flag);
    double* lds = ldsall + threadIdx.x;
    const int R = sa.nrows;
    double* srow = lds + (long long)sa.nslots * (NG + 2) * BLOCK;  // s_r of this unit, R columns
    MixedUnit<NG> U;
    double yh[NG > 0 ? NG : 1];
    bool live = !q.msk;
    if (live) {
        mixed_normal_eq<NG, false>(M, a, q, lds, sa.bslot, U);
        if (!U.ok) { live = false; atomicAdd(not_spd, 1ull); }
    }
    if (live) {
#pragma unroll
        for (int g = 0; g < NG; ++g) yh[g] = U.bv[g];
        fwd_subst<NG>(U.A, yh);
        for (int sl = 0; sl < sa.nslots; ++sl) {  // Q_j = L^-1 W_j, in place
            double v[NG > 0 ? NG : 1];
#pragma unroll
            for (int g = 0; g < NG; ++g) v[g] = lds[(sl * (NG + 2) + g) * BLOCK];
            fwd_subst<NG>(U.A, v);
#pragma unroll
            for (int g = 0; g < NG; ++g) lds[(sl * (NG + 2) + g) * BLOCK] = v[g];
        }
        for (int r = 0; r < R; ++r) {  // SED of every global row on this unit
            const Comp& cr = M.comp[a.tc[sa.rt[r]]];
            double t0, t1;
            load_theta(M, cr, q.i, q.k, t0, t1);
            srow[r * BLOCK] = comp_sed(M, cr, q.i, q.k, sa.rj[r], sed_prep(cr, t0, t1));
        }
    }
    for (int r = 0; r < R; ++r) {
        const Comp& cr = M.comp[a.tc[sa.rt[r]]];
        const int jr = sa.rj[r], sl = sa.bslot[jr];
        const bool on_r = live && q.p < gl_nplanes(cr, a.flag);
        const double s_r = on_r ? srow[r * BLOCK] : 0.0;
        const double w_r = on_r ? ((cr.type == DANGX_MONOPOLE) ? 1.0 : s_r) : 0.0;  // :857
        double tv = 0.0, fv = 0.0, inv_r = 0.0;
        if (on_r) {
            double dot = 0.0;
#pragma unroll
            for (int g = 0; g < NG; ++g) dot += lds[(sl * (NG + 2) + g) * BLOCK] * yh[g];
            tv = lds[(sl * (NG + 2) + NG) * BLOCK] * s_r - w_r * dot;
            fv = lds[(sl * (NG + 2) + NG + 1) * BLOCK] * w_r;
            const double rms = M.rms[((long long)jr * M.nmaps + (q.k - 1)) * M.npix + q.i];
            inv_r = 1.0 / (rms * rms);
        }
        block_row_sum(tv, R * R + r, rowpartial, sh);
        block_row_sum(fv, R * R + R + r, rowpartial, sh);
        block_row_sum(w_r * s_r * inv_r, R * R + 2 * R + r, rowpartial, sh);
        for (int r2 = 0; r2 < R; ++r2) {
            const Comp& c2 = M.comp[a.tc[sa.rt[r2]]];
            const int sl2 = sa.bslot[sa.rj[r2]];
            double v = 0.0;
            if (on_r && q.p < gl_nplanes(c2, a.flag)) {
                const double s2 = srow[r2 * BLOCK];
                double dot = 0.0;
#pragma unroll
                for (int g = 0; g < NG; ++g) dot += lds[(sl * (NG + 2) + g) * BLOCK] * lds[(sl2 * (NG + 2) + g) * BLOCK];
                v = ((sl == sl2) ? w_r * s2 * inv_r : 0.0) - w_r * s2 * dot;
            }
            block_row_sum(v, r * R + r2, rowpartial, sh);
        }
    }
}

// pass 2: block solve with the global members' (new) signal removed from the data; writes the amplitudes
template <int NG>
__global__ __launch_bounds__(BLOCK) void k_schur_pass2(const Model* __restrict__ Mp, GroupArgs a) {
    const Model& M = *Mp;
    const UnitId q = unit_of(M, a.flag);
    if (q.msk) return;
    MixedUnit<NG> U;
    mixed_normal_eq<NG, true>(M, a, q, nullptr, nullptr, U);
    if (!U.ok) return;
    double v[NG > 0 ? NG : 1];
#pragma unroll
    for (int g = 0; g < NG; ++g) v[g] = U.bv[g];
    fwd_subst<NG>(U.A, v);
#pragma unroll
    for (int g = NG - 1; g >= 0; --g) {
        double s = v[g];
#pragma unroll
        for (int t = g + 1; t < NG; ++t) s -= U.A[t * (t + 1) / 2 + g] * v[t];
        v[g] = s / U.A[g * (g + 1) / 2 + g];
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) M.comp[a.gc[g]].amp[(long long)(q.k - 1) * M.npix + q.i] = v[g];
}

// Residual of the GLOBAL rows of the reference's system at the current state (amplitude maps + template amplitudes),
// evaluated directly -- no elimination, hence none of pass 1's cancellation:
//   rows [0, R):  sum_u ( d_j s_r / sigma_j^2  -  w_r * (sum_g a_g M_gj + sum_t' g_t' s_t') / sigma_j^2 )   (b - A x, without
//                 the fluctuation term, which the host adds from pass 1's sums)
//   rows [R, 2R): sum_u d_j s_r / sigma_j^2                                                                  (the row of b)
//   rows [2R,3R): sum_u ( |d_j s_r| + |w_r model| ) / sigma_j^2       (size of the terms that cancel: the rounding floor)
// for row r = (global member t_r, band j = rj[r]); d_j is compute_rhs's data (src/dang_cg_mod.f90:367-460), the row weight
// w_r is 1 for a monopole (:857) and s_r otherwise, as in k_rhs_mixed / k_Ax_mixed.
template <int NG>
__global__ __launch_bounds__(BLOCK) void k_schur_resid(const Model* __restrict__ Mp, GroupArgs a, SchurArgs sa,
                                                       double* __restrict__ rowpartial) {
    __shared__ double sh[BLOCK / 64];
    constexpr int NA = NG > 0 ? NG : 1;
    const Model& M = *Mp;
    const UnitId q = unit_of(M, a.flag);
    const int R = sa.nrows;
    Prep pr[NA], prt[MAXT];
    double av[NA];
#pragma unroll
    for (int g = 0; g < NA; ++g) { pr[g] = Prep{0, 0, 0}; av[g] = 0.0; }
    if (!q.msk) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const Comp& c = M.comp[a.gc[g]];
            double t0, t1;
            load_theta(M, c, q.i, q.k, t0, t1);
            pr[g] = sed_prep(c, t0, t1);
            av[g] = c.amp[(long long)(q.k - 1) * M.npix + q.i];
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
        if (sa.bslot[j] < 0) continue;  // uniform: no global row at this band
        double d = 0.0, inv = 0.0, model = 0.0, st[MAXT];
        if (!q.msk) {
            d = M.sig[((long long)j * M.nmaps + (q.k - 1)) * M.npix + q.i];
            if (q.k == 1) d = d / M.gain[j];
            d = remove_others(M, a, q.i, q.k, j, d);
            const double rms = M.rms[((long long)j * M.nmaps + (q.k - 1)) * M.npix + q.i];
            inv = 1.0 / (rms * rms);
#pragma unroll
            for (int g = 0; g < NG; ++g) model = model + av[g] * sed_eval(M, M.comp[a.gc[g]], j, pr[g]);
        }
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            st[t] = 0.0;
            if (t < a.nt) {
                const Comp& c = M.comp[a.tc[t]];
                if (((c.corr_mask >> j) & 1) && !q.msk && q.p < gl_nplanes(c, a.flag)) {
                    st[t] = comp_sed(M, c, q.i, q.k, j, prt[t]);
                    model = model + c.tamp[q.k - 1][j] * st[t];
                }
            }
        }
        for (int r = 0; r < R; ++r) {
            if (sa.rj[r] != j) continue;
            const int t = sa.rt[r];
            const Comp& c = M.comp[a.tc[t]];
            const bool on = !q.msk && q.p < gl_nplanes(c, a.flag);
            const double w = (c.type == DANGX_MONOPOLE) ? 1.0 : st[t];
            const double bterm = on ? d * inv * st[t] : 0.0;
            block_row_sum(on ? bterm - w * (model * inv) : 0.0, r, rowpartial, sh);
            block_row_sum(bterm, R + r, rowpartial, sh);
            block_row_sum(on ? fabs(bterm) + fabs(w * (model * inv)) : 0.0, 2 * R + r, rowpartial, sh);
        }
    }
}

template <int NG>
struct LaunchSchur {
    static int run(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs* sa, long long SN, double* rows_dev, bool resid = false) {
        const unsigned nblk = nblocks(SN);
        if (sa && resid) {
            const int nrows = 3 * sa->nrows;
            if (ensure_partial(ctx, (long long)nrows * nblk)) return 1;
            hipLaunchKernelGGL(k_schur_resid<NG>, dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, *sa, ctx->partial);
            dx_reduce_rows_to(ctx, ctx->partial, nblk, nrows, rows_dev);
        } else if (sa) {
            const int R = sa->nrows, nrows = R * R + 3 * R;
            if (ensure_partial(ctx, (long long)nrows * nblk)) return 1;
            const size_t lds = ((size_t)sa->nslots * (NG + 2) + R) * BLOCK * sizeof(double);
            if (lds > 150 * 1024) return fail(ctx, "too many bands x components for the direct solve of a template group: use DANGX_SOLVER_CG");
            HIPCHK(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long), ctx->stream));
            hipLaunchKernelGGL(k_schur_pass1<NG>, dim3(nblk), dim3(BLOCK), lds, ctx->stream, ctx->dm, a, *sa, ctx->partial, ctx->counters);
            dx_reduce_rows_to(ctx, ctx->partial, nblk, nrows, rows_dev);
        } else {
            Timed t(ctx, DANGX_K_AMP_DIRECT);
            hipLaunchKernelGGL(k_schur_pass2<NG>, dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, a);
        }
        HIPCHK(ctx, hipGetLastError());
        return 0;
    }
};
int dispatch_schur(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs* sa, long long SN, double* rows_dev, bool resid = false) {
    switch (a.ng) {
    case 0: return sa ? LaunchSchur<0>::run(ctx, a, sa, SN, rows_dev, resid) : 0;  // no diffuse member: nothing to back-substitute
    case 1: return LaunchSchur<1>::run(ctx, a, sa, SN, rows_dev, resid);
    case 2: return LaunchSchur<2>::run(ctx, a, sa, SN, rows_dev, resid);
    case 3: return LaunchSchur<3>::run(ctx, a, sa, SN, rows_dev, resid);
    case 4: return LaunchSchur<4>::run(ctx, a, sa, SN, rows_dev, resid);
    case 5: return LaunchSchur<5>::run(ctx, a, sa, SN, rows_dev, resid);
    case 6: return LaunchSchur<6>::run(ctx, a, sa, SN, rows_dev, resid);
    default: return fail(ctx, "direct solve of a template group supports up to 6 diffuse members: use DANGX_SOLVER_CG");
    }
}

}  // namespace

int dx_launch_schur_pass1(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev) {
    const int rc = dx_launch_schur_pass1_reg(ctx, a, sa, SN, rows_dev);  // template / monopole members on the amplitude kernel's schedule
    if (rc >= 0) return rc;
    return dispatch_schur(ctx, a, &sa, SN, rows_dev);
}
int dx_launch_schur_pass2(dangx_ctx* ctx, const GroupArgs& a, long long SN) {
    // template / monopole members, delta bands, nothing else on the planes: the amplitude kernel's schedule (dangx_ampreg.hip)
    if (dx_launch_amp_reg_templates(ctx, a, SN) == 0) return 0;
    return dispatch_schur(ctx, a, nullptr, SN, nullptr);
}
int dx_launch_schur_resid(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev) {
    const int rc = dx_launch_schur_resid_reg(ctx, a, sa, SN, rows_dev);  // template / monopole members on the amplitude kernel's schedule
    if (rc >= 0) return rc;
    return dispatch_schur(ctx, a, &sa, SN, rows_dev, true);
}
