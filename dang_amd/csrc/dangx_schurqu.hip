// dangx_schurqu.hip -- pass 1 of the Schur solve of a Q+U template group with one thread per pixel (see the kernel's comment;
// dangx_schurreg.hip holds the per-plane passes and calls in here first).  Compiled twice (dang_amd/_build.py):
// -DDX_QU_TB=5 / 2, the band tile (five or more bands / fewer; the nb % TB bands left over are taken one by one) -- 48 kernels
// per unit (members 1-6 x global rows 1-8) build side by side.
#include "dx_ampreg.h"

#ifndef DX_QU_TB
#define DX_QU_TB 5
#endif
#ifndef DX_QU_NP      // planes a thread works through: 2 (Q+U with equal index maps) or 1 (any single plane: a T group with monopoles,
#define DX_QU_NP 2    // Q or U alone, Q+U with index maps that differ -- one block row per plane)
#endif
#ifndef DX_QU_WAVES   // waves per SIMD the kernel is compiled for, by its number of global rows.  Three (<= 168 registers, possible
#define DX_QU_WAVES(SS) 2   // with two varying members: 41 KB of SED columns) costs 28-32 spilled registers and gains nothing
#endif

namespace {

// Pass 1 for a Q+U group whose varying members have the same indices on both planes (as a Q+U sweep leaves them): ONE thread per
// pixel works through both planes at once.  The members' SED columns are evaluated once, for every band, into the thread's LDS
// column (the plane-set kernel's layout) instead of once per plane; the band loop accumulates the normal equations of BOTH planes
// from one tile of maps (half the memory round trips: each map plane is a 100-300 MB stride from the next, and two waves per SIMD
// hide little of one); the vectors of the global rows' bands are read again after the loop (from the cache the maps just passed
// through) instead of being picked out of it band by band; and (up to five rows) the row values of the two planes are added in
// registers before the one block reduction.  SS = R, the group's global rows (up to 8).  A Q/U template's row weight is its
// template value (a monopole exists on T only), so the Schur matrix of a unit is symmetric: R (R + 1) / 2 + 3 R row values instead
// of R^2 + 3 R.
template <int NG, int TB, int SS, int NP>
__global__ __launch_bounds__(BLOCK, DX_QU_WAVES(SS)) void k_schur_pass1_qu(const Model* __restrict__ Mp, GroupArgs a, AmpRegArgs ra, SchurArgs sa,
                                                            double* __restrict__ rowpartial, unsigned long long* __restrict__ not_spd) {
    constexpr int NP_ = SS * (SS + 1) / 2, NV = NP_ + 3 * SS, NA = NG * (NG + 1) / 2;   // NP_: entries of the symmetric S
    extern __shared__ double lds[];
    __shared__ double wsum[NV][BLOCK / 64];
    const Model& M = *Mp;
    const int npix = M.npix, nb = M.nbands, tid = threadIdx.x, nfull = (nb / TB) * TB;   // (the launcher: nb >= TB)
    double* tab = lds;
    double* cu = lds + (TROWS * NG + 3) * nb;          // [plane][MAXU templates][band], zero where nothing is removed
    double* col = lds + (TROWS * NG + 3 + NP * MAXU) * nb + tid;   // [member slot][band] x BLOCK
    const long long u = (long long)blockIdx.x * BLOCK + tid;
    const bool in_range = u < npix;
    const int i = in_range ? (int)u : 0;
    const double mk = as_global(M.mask)[i];
    // NP = 2: planes 2 and 3 (plane 3's maps are npix further on); NP = 1: the plane of this block row.  Tile 0 of every plane is
    // requested before the table and the SED columns
    const int k0 = (NP == 2) ? 2 : flag_map(a.flag, (int)blockIdx.y);
    const long long bstride = (long long)M.nmaps * npix;
    const gcptr sig0 = as_global(M.sig) + (long long)(k0 - 1) * npix + i, rms0 = as_global(M.rms) + (long long)(k0 - 1) * npix + i;
    double dcur[NP][TB], rcur[NP][TB], tv[NP][MAXU], tT[MAXU];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int t = 0; t < TB; ++t) { dcur[p][t] = sig0[(long long)p * npix + t * bstride]; rcur[p][t] = rms0[(long long)p * npix + t * bstride]; }
        gl_load(M, ra, i, k0 + p, npix, tv[p], tT);
    }
    // (every template slot is read for every band, unconditionally: a test per template and band -- uniform, but compiled as an
    // exec-masked branch with a full wait on the maps in flight -- costs more than the multiplications by zero)
    for (int t = tid; t < NP * MAXU * nb; t += BLOCK) {
        const int p = t / (MAXU * nb), q = t - p * MAXU * nb, w = q / nb, j = q - w * nb;
        double v = 0.0;
        if (w < ra.nu && ((ra.uinuc >> w) & 1u)) {
            const Comp& c = M.comp[ra.ucomp[w]];
            if (!((c.corr_mask >> j) & 1)) v = c.tamp[k0 - 1 + p][j];
        }
        cu[t] = v;
    }
    sed_table_build(M, tab, tid, BLOCK, a.gc, NG);
    const bool sample = (a.ml_mode == DANGX_ML_SAMPLE);
    const bool live = in_range && !is_masked(mk);
    __syncthreads();
    // compute_rhs divides the temperature plane by the band gains (src/dang_cg_mod.f90:371): only where some gain is not 1
    const double* gain = tab + (TROWS * NG + 1) * nb;
    bool cal = false;
    if (NP == 1 && k0 == 1)
        for (int j = 0; j < nb; ++j) cal = cal || gain[j] != 1.0;
    double eta[NP], f0[NP];
    double A[NP][NA], bv[NP][NG];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        eta[p] = 0.0; f0[p] = 0.0;
#pragma unroll
        for (int q = 0; q < NA; ++q) A[p][q] = 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) bv[p][g] = 0.0;
    }
    if (live) {
        // ---- the SED columns, once (NP = 2: plane 2's index values, equal on plane 3 -- the launcher has checked)
#pragma unroll 1
        for (int v = 0; v < ra.nv; ++v) {
            const Comp& c = M.comp[a.gc[ra.vcomp[v]]];
            const gcptr ix = as_global(c.idx) + (long long)(k0 - 1) * npix + i;
            const double t0 = (c.nind > 0) ? ix[0] : 0.0, t1 = (c.nind > 1) ? ix[(long long)M.nmaps * npix] : 0.0;
            const Prep pr = sed_prep(c, t0, t1);
#pragma unroll 1
            for (int j0 = 0; j0 < nfull; j0 += TB) sed_tile<TB>(ra.vtype[v], tab, nb, NG, ra.vcomp[v], j0, pr, col + (v * nb + j0) * BLOCK);
#pragma unroll 1
            for (int j = nfull; j < nb; ++j) sed_tile<1>(ra.vtype[v], tab, nb, NG, ra.vcomp[v], j, pr, col + (v * nb + j) * BLOCK);
        }
        if (sample) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                double u1, u2;
                uniform2(a.seed, a.stream, (unsigned long long)(M.pix0 + i), (uint32_t)(k0 + p), u1, u2);
                eta[p] = rand_normal(0.0, 1.0, u1, u2);
            }
        }
        // ---- the normal equations of both planes, band tile by band tile (then the nb % TB bands left over, one by one)
        auto accumulate = [&](int p, int j, double d, double rms) {
            if (cal) d = d / gain[j];
#pragma unroll
            for (int w = 0; w < MAXU; ++w) d = d - cu[(p * MAXU + w) * nb + j] * tv[p][w];
            const double is = fast_rcp(rms);
            const double inv = is * is;
            double mrow[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g)
                mrow[g] = (ra.vslot[g] >= 0) ? col[(ra.vslot[g] * nb + j) * BLOCK] : tab[(TROWS * g + 2 + k0 + p) * nb + j];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const double t2 = mrow[g] * inv;
                bv[p][g] += d * t2;
#pragma unroll
                for (int h = 0; h <= g; ++h) A[p][g * (g + 1) / 2 + h] += t2 * mrow[h];
            }
            f0[p] += (eta[p] * is) * mrow[NG - 1];
        };
#pragma unroll 1
        for (int j0 = 0; j0 < nfull; j0 += TB) {
            if (j0 > 0) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int t = 0; t < TB; ++t) {
                        dcur[p][t] = sig0[(long long)p * npix + (j0 + t) * bstride];
                        rcur[p][t] = rms0[(long long)p * npix + (j0 + t) * bstride];
                    }
            }
#pragma unroll
            for (int t = 0; t < TB; ++t)
#pragma unroll
                for (int p = 0; p < NP; ++p) accumulate(p, j0 + t, dcur[p][t], rcur[p][t]);
        }
#pragma unroll 1
        for (int j = nfull; j < nb; ++j) {
            double dl[NP], rl[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) { dl[p] = sig0[(long long)p * npix + j * bstride]; rl[p] = rms0[(long long)p * npix + j * bstride]; }
#pragma unroll
            for (int p = 0; p < NP; ++p) accumulate(p, j, dl[p], rl[p]);
        }
    }
    // ---- per plane: W = M / sigma^2, d / sigma^2, eta / sigma and 1 / sigma^2 of the band of every global row (two rows on one band
    // carry the same vector twice), the Cholesky factor, yh = L^-1 b and Q_r = L^-1 W_r; `good`: the unit contributes row values
    auto plane_rows = [&](int p, double (&Wv)[SS][NG], double (&dn)[SS], double (&en)[SS], double (&iv)[SS], double (&sr)[SS]) -> bool {
        if (!live) return false;
        const int k = k0 + p;
        double ri[NG], dr[SS], rr[SS];
#pragma unroll
        for (int r = 0; r < SS; ++r) {
            const int j = __builtin_amdgcn_readfirstlane((int)sa.rj[r]);
            dr[r] = sig0[(long long)p * npix + j * bstride];
            rr[r] = rms0[(long long)p * npix + j * bstride];
        }
#pragma unroll
        for (int r = 0; r < SS; ++r) {
            const int j = __builtin_amdgcn_readfirstlane((int)sa.rj[r]);
            double d = dr[r];
            if (cal) d = d / gain[j];
#pragma unroll
            for (int w = 0; w < MAXU; ++w) d = d - cu[(p * MAXU + w) * nb + j] * tv[p][w];
            const double is = fast_rcp(rr[r]);
            const double inv = is * is;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const double m = (ra.vslot[g] >= 0) ? col[(ra.vslot[g] * nb + j) * BLOCK] : tab[(TROWS * g + 2 + k) * nb + j];
                Wv[r][g] = m * inv;
            }
            dn[r] = d * inv; en[r] = eta[p] * is; iv[r] = inv;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < MAXU; ++w) s = (ra.rowu[r] == w) ? tv[p][w] : s;
            sr[r] = s;
        }
        bv[p][0] += f0[p];
        bool ok = true;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int h = 0; h <= g; ++h) {
                double sacc = A[p][g * (g + 1) / 2 + h];
#pragma unroll
                for (int t = 0; t < h; ++t) sacc -= A[p][g * (g + 1) / 2 + t] * A[p][h * (h + 1) / 2 + t];
                if (h == g) {
                    if (!(sacc > 0.0) || !(sacc < 1.0e300)) ok = false;
                    ri[g] = fast_rsqrt(sacc);
                } else {
                    A[p][g * (g + 1) / 2 + h] = sacc * ri[h];
                }
            }
        }
        if (!ok) { atomicAdd(not_spd, 1ull); return false; }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double sacc = bv[p][g];
#pragma unroll
            for (int t = 0; t < g; ++t) sacc -= A[p][g * (g + 1) / 2 + t] * bv[p][t];
            bv[p][g] = sacc * ri[g];   // yh
        }
#pragma unroll
        for (int r = 0; r < SS; ++r)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                double sacc = Wv[r][g];
#pragma unroll
                for (int t = 0; t < g; ++t) sacc -= A[p][g * (g + 1) / 2 + t] * Wv[r][t];
                Wv[r][g] = sacc * ri[g];   // Q_r
            }
        return true;
    };
    // the values of row r: t[r], its fluctuation sum, G[r][r] and S[r][r2], r2 >= r (S[r][r2] = S[r2][r])
    auto row_values = [&](int p, int r, const double (&Wv)[SS][NG], const double (&dn)[SS], const double (&en)[SS], const double (&iv)[SS],
                          const double (&sr)[SS], double* out3, double* outS) {
        double dot = 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) dot += Wv[r][g] * bv[p][g];
        out3[0] = dn[r] * sr[r] - sr[r] * dot;
        out3[1] = en[r] * sr[r];
        out3[2] = sr[r] * sr[r] * iv[r];
#pragma unroll
        for (int r2 = r; r2 < SS; ++r2) {
            double dot2 = 0.0;
#pragma unroll
            for (int g = 0; g < NG; ++g) dot2 += Wv[r][g] * Wv[r2][g];
            outS[r2 - r] = ((sa.rj[r] == sa.rj[r2]) ? sr[r] * sr[r2] * iv[r] : 0.0) - sr[r] * sr[r2] * dot2;
        }
    };
    auto wave_sum_to = [&](double v, int e, bool first) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((tid & 63) == 0) wsum[e][tid >> 6] = first ? v : wsum[e][tid >> 6] + v;
    };
    if constexpr (SS <= 5) {
        // few rows: both planes' row values added in registers, ONE reduction
        double rv[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) rv[e] = 0.0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            double Wv[SS][NG], dn[SS], en[SS], iv[SS], sr[SS];
            if (plane_rows(p, Wv, dn, en, iv, sr)) {
#pragma unroll
                for (int r = 0; r < SS; ++r) {
                    double o3[3], oS[SS];
                    row_values(p, r, Wv, dn, en, iv, sr, o3, oS);
                    rv[NP_ + r] += o3[0]; rv[NP_ + SS + r] += o3[1]; rv[NP_ + 2 * SS + r] += o3[2];
#pragma unroll
                    for (int q = 0; q < SS - r; ++q) rv[r * SS - r * (r - 1) / 2 + q] += oS[q];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) wave_sum_to(rv[e], e, true);
    } else {
        // many rows: R (R + 1) / 2 + 3 R values do not fit beside the Q vectors -- row by row, each row's values reduced at once and
        // the second plane's added to the first's in the wave's LDS slots
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            double Wv[SS][NG], dn[SS], en[SS], iv[SS], sr[SS];
            const bool good = plane_rows(p, Wv, dn, en, iv, sr);
#pragma unroll
            for (int r = 0; r < SS; ++r) {
                double o3[3] = {0.0, 0.0, 0.0}, oS[SS];
#pragma unroll
                for (int q = 0; q < SS; ++q) oS[q] = 0.0;
                if (good) row_values(p, r, Wv, dn, en, iv, sr, o3, oS);
                wave_sum_to(o3[0], NP_ + r, p == 0);
                wave_sum_to(o3[1], NP_ + SS + r, p == 0);
                wave_sum_to(o3[2], NP_ + 2 * SS + r, p == 0);
#pragma unroll
                for (int q = 0; q < SS - r; ++q) wave_sum_to(oS[q], r * SS - r * (r - 1) / 2 + q, p == 0);
            }
        }
    }
    __syncthreads();
    if (tid < NV) {   // rows of the layout the host reads: [0, R^2) S, then t, the fluctuation sums, G's diagonal (R = SS)
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += wsum[tid][w];
        const long long nblk = (long long)gridDim.x * gridDim.y, blk = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        if (tid >= NP_) {
            rowpartial[(long long)(SS * SS + (tid - NP_)) * nblk + blk] = t;
        } else {
            int r = 0, e = tid;
            while (e >= SS - r) { e -= SS - r; ++r; }
            const int r2 = r + e;
            rowpartial[(long long)(r * SS + r2) * nblk + blk] = t;
            if (r2 != r) rowpartial[(long long)(r2 * SS + r) * nblk + blk] = t;
        }
    }
}

template <int NG, int SS>
int launch_qu(dangx_ctx* ctx, const GroupArgs& a, const AmpRegArgs& ra, const SchurArgs& sa, long long SN, double* rows_dev, size_t ldsz) {
    const int R = sa.nrows, nrows = R * R + 3 * R, planes = flag_planes_h(a.flag);
    const unsigned gx = nblocks(SN / planes), gy = (DX_QU_NP == 2) ? 1u : (unsigned)planes, nblk = gx * gy;
    if (ensure_partial(ctx, (long long)nrows * nblk)) return 1;
    HIPCHK(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL((k_schur_pass1_qu<NG, DX_QU_TB, SS, DX_QU_NP>), dim3(gx, gy), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, a, ra, sa, ctx->partial, ctx->counters);
    dx_reduce_rows_to(ctx, ctx->partial, nblk, nrows, rows_dev);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}
template <int NG>
int launch_qu_ng(dangx_ctx* ctx, const GroupArgs& a, const AmpRegArgs& ra, const SchurArgs& sa, long long SN, double* rows_dev, size_t ldsz) {
    switch (sa.nrows) {
    case 1: return launch_qu<NG, 1>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    case 2: return launch_qu<NG, 2>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    case 3: return launch_qu<NG, 3>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    case 4: return launch_qu<NG, 4>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    case 5: return launch_qu<NG, 5>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    case 6: return launch_qu<NG, 6>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    case 7: return launch_qu<NG, 7>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    case 8: return launch_qu<NG, 8>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    default: return -1;
    }
}

}  // namespace

#define DX_QU_CAT3(a, b, c, d) a##b##c##d
#define DX_QU_CAT(a, b, c, d) DX_QU_CAT3(a, b, c, d)
// 0 launched, 1 error, -1 not covered: fewer bands than this unit's tile, more than eight global rows, a hi_fit member (its row
// weight carries a per-pixel Planck factor: dangx_schurreg.hip), a monopole whose template is not identically 1 on the plane (its
// row weight IS 1, src/dang_cg_mod.f90:857: the unit's Schur matrix would not be symmetric), SED columns beyond 80 KB per block;
// for the two-plane form also: the flag is not Q+U, or index maps that differ between Q and U
int DX_QU_CAT(dx_schurqu, DX_QU_NP, _pass1_tb, DX_QU_TB)(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev) {
    static const bool qu_on = [] { const char* e = getenv("DANGX_SCHUR_QU"); return !(e && e[0] == '0'); }();  // A/B switch
    AmpRegArgs ra;
    const int nb = ctx->hm.nbands;
    if (!qu_on || nb < DX_QU_TB || sa.nrows < 1 || sa.nrows > 8 || !template_group_args(ctx, a, ra)) return -1;
    if (DX_QU_NP == 2 && a.flag != DANGX_FLAG_QU) return -1;
    if (ra.uhifit != 0u) return -1;
    if (a.ml_mode == DANGX_ML_SAMPLE && a.fluct != DANGX_FLUCT_REFERENCE) return -1;
    unsigned planes = 0;
    for (int pl = 0; pl < flag_planes_h(a.flag); ++pl)
        planes |= 1u << (((a.flag & DANGX_FLAG_QU) ? 2 + pl : (a.flag & DANGX_FLAG_T) ? 1 : (a.flag & DANGX_FLAG_Q) ? 2 : 3) - 1);
    for (int t = 0; t < a.nt; ++t) {
        const int l = a.tc[t], ty = ctx->desc[l].type;
        if (ty == DANGX_MONOPOLE) { if ((ctx->tmpl_one[l] & planes) != planes) return -1; }
        else if (ty != DANGX_TEMPLATE) return -1;
    }
    for (int r = 0; r < sa.nrows; ++r) {
        const int l = a.tc[sa.rt[r]];
        for (int w = 0; w < ra.nu; ++w) if (ra.ucomp[w] == l) ra.rowu[r] = (signed char)w;
        if (ra.rowu[r] < 0) return -1;
    }
    if (DX_QU_NP == 2)
        for (int v = 0; v < ra.nv; ++v) {   // the SED columns are evaluated once for both planes
            const int l = a.gc[ra.vcomp[v]];
            const unsigned all = (1u << ctx->hm.comp[l].nind) - 1u;
            if ((ctx->qu_equal[l] & all) != all) return -1;
        }
    const size_t ldsz = ((size_t)(TROWS * a.ng + 3 + DX_QU_NP * MAXU) * nb + (size_t)ra.nv * nb * BLOCK) * sizeof(double);
    if (ldsz > 80u * 1024u) return -1;
    switch (a.ng) {
    case 1: return launch_qu_ng<1>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    case 2: return launch_qu_ng<2>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    case 3: return launch_qu_ng<3>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    case 4: return launch_qu_ng<4>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    case 5: return launch_qu_ng<5>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    case 6: return launch_qu_ng<6>(ctx, a, ra, sa, SN, rows_dev, ldsz);
    default: return -1;
    }
}
