// dangx_schurreg.hip -- the three passes of the Schur-complement solve of a CG group with template / monopole members
// (dangx_schur.hip holds the derivation and the run-time-typed kernels for every other case) on the amplitude kernel's
// schedule (dangx_ampreg.hip): band tiles requested before their SEDs are evaluated, constants from the block's LDS table,
// v_rcp / v_rsq based reciprocals.  At the C3 size (Q+U group of four diffuse members + a template fitted at three bands) one
// solve takes 7.3 ms instead of 13.9: pass 1 3.8 (6.6), pass 2 = k_amp_reg<.., true> 1.6 (3.7), residual check 1.6 (3.3).
#include "dx_ampreg.h"

// Compiled four times (dang_amd/_build.py): -DDX_SCHUR_PART=1 (pass 1) / 2 (the residual pass) x -DDX_SCHUR_HF=0 (groups without
// a hi_fit member; these units also hold the public launchers) / 1 (kernels that carry the Planck factor of hi_fit members):
// four units of ~45 kernels build side by side instead of one of 180.
#ifndef DX_SCHUR_HF
#define DX_SCHUR_HF 0
#endif
#ifndef DX_P1_WAVES        // waves per SIMD pass 1 is compiled for (3 costs it 275 spilled registers)
#define DX_P1_WAVES 2
#endif
#ifndef DX_SCHUR_PART      // 1: pass 1, 2: the residual pass (four units in all)
#define DX_SCHUR_PART 1
#endif

namespace {
constexpr bool HFV = DX_SCHUR_HF != 0;

#if DX_SCHUR_PART == 1
// Pass 1 of the Schur solve of a template group (dangx_schur.hip: k_schur_pass1, whose header derives the sums) on this
// kernel's schedule, for groups whose global members are templates / monopoles fitted at up to SS bands: the normal equations of
// the diffuse members as k_amp_reg forms them, with the vectors W_j = M_j / sigma_j^2, d_j / sigma_j^2, eta / sigma_j and
// 1 / sigma_j^2 of the bands that carry a global row kept in REGISTERS (slot = sa.bslot[j]) instead of LDS columns; after the
// Cholesky factor the R^2 + 3R row values of the unit and one block reduction for all of them.
//   rows [0, R^2): S[r][r'] ; [R^2, R^2+R): t[r] ; [R^2+R, R^2+2R): fluctuation sum of natural row r ; [R^2+2R, R^2+3R): G[r][r]
template <int NG, int TB, int SS, bool HF>
__global__ __launch_bounds__(BLOCK, DX_P1_WAVES) void k_schur_pass1_reg(const Model* __restrict__ Mp, GroupArgs a, AmpRegArgs ra, SchurArgs sa,
                                                             double* __restrict__ rowpartial, unsigned long long* __restrict__ not_spd) {
    constexpr int NV = SS * SS + 3 * SS;   // row values of a unit (R <= SS)
    extern __shared__ double lds[];
    __shared__ double wsum[NV][BLOCK / 64];
    const Model& M = *Mp;
    const int npix = M.npix, nb = M.nbands, tid = threadIdx.x, R = sa.nrows;
    double* tab = lds;
    double* cu = lds + (TROWS * NG + 3) * nb;   // templates' amplitudes on the bands they are NOT fitted at (:445-460)
    double* prl = lds + (TROWS * NG + 3 + ra.nu) * nb + tid;
    double* col = prl + 3 * ra.nv * BLOCK;
    const long long u = (long long)blockIdx.x * BLOCK + tid;
    const bool in_range = u < npix;
    const int i = in_range ? (int)u : 0;
    const int k = flag_map(a.flag, (int)blockIdx.y);
    const double mk = as_global(M.mask)[i];
    double th[NG][2], tv[MAXU], tT[MAXU];
    gl_load(M, ra, i, k, npix, tv, tT);
#pragma unroll
    for (int v = 0; v < NG; ++v) {
        th[v][0] = th[v][1] = 0.0;
        if (v < ra.nv) {
            const Comp& c = M.comp[a.gc[ra.vcomp[v]]];
            const gcptr ix = as_global(c.idx) + (long long)(k - 1) * npix + i;
            if (c.nind > 0) th[v][0] = ix[0];
            if (c.nind > 1) th[v][1] = ix[(long long)M.nmaps * npix];
        }
    }
    for (int t = tid; t < ra.nu * nb; t += BLOCK) {
        const int w = t / nb, j = t - w * nb;
        const Comp& c = M.comp[ra.ucomp[w]];
        cu[t] = (((ra.uinuc >> w) & 1u) && !((c.corr_mask >> j) & 1)) ? c.tamp[k - 1][j] : 0.0;
    }
    sed_table_build(M, tab, tid, BLOCK, a.gc, NG);
    const bool sample = (a.ml_mode == DANGX_ML_SAMPLE);
    double eta = 0.0, f0 = 0.0;
    if (sample) {
        double u1, u2;
        uniform2(a.seed, a.stream, (unsigned long long)(M.pix0 + i), (uint32_t)k, u1, u2);
        eta = rand_normal(0.0, 1.0, u1, u2);
    }
    bool live = in_range && !is_masked(mk);
    if (live) {
#pragma unroll
        for (int v = 0; v < NG; ++v)
            if (v < ra.nv) { prl[(3 * v + 0) * BLOCK] = th[v][0]; prl[(3 * v + 1) * BLOCK] = th[v][1]; }
#pragma unroll 1
        for (int v = 0; v < ra.nv; ++v) {
            const Comp& c = M.comp[a.gc[ra.vcomp[v]]];
            const Prep pr = sed_prep(c, prl[(3 * v + 0) * BLOCK], prl[(3 * v + 1) * BLOCK]);
            prl[(3 * v + 0) * BLOCK] = pr.p0;
            prl[(3 * v + 1) * BLOCK] = pr.p1;
            prl[(3 * v + 2) * BLOCK] = pr.p2;
        }
    }
    __syncthreads();
    double A[NG * (NG + 1) / 2], bv[NG], ri[NG];
    double Wv[SS][NG], dn[SS], en[SS], iv[SS];
    bool ok = false;
    if (live) {
#pragma unroll
        for (int q = 0; q < NG * (NG + 1) / 2; ++q) A[q] = 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) bv[g] = 0.0;
#pragma unroll
        for (int sl = 0; sl < SS; ++sl) {
            dn[sl] = en[sl] = iv[sl] = 0.0;
#pragma unroll
            for (int g = 0; g < NG; ++g) Wv[sl][g] = 0.0;
        }
        const double* gain = tab + (TROWS * NG + 1) * nb;
        const long long bstride = (long long)M.nmaps * npix;
        const gcptr sigp = as_global(M.sig) + (long long)(k - 1) * npix + i;
        const gcptr rmsp = as_global(M.rms) + (long long)(k - 1) * npix + i;
#pragma unroll 1
        for (int j0 = 0; j0 < nb; j0 += TB) {
            double dcur[TB], rcur[TB];
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                dcur[t] = sigp[(j0 + t) * bstride];
                rcur[t] = rmsp[(j0 + t) * bstride];
            }
#pragma unroll 1
            for (int v = 0; v < ra.nv; ++v) {
                const Prep pr = {prl[(3 * v + 0) * BLOCK], prl[(3 * v + 1) * BLOCK], prl[(3 * v + 2) * BLOCK]};
                sed_tile<TB>(ra.vtype[v], tab, nb, NG, ra.vcomp[v], j0, pr, col + (v * TB) * BLOCK);
            }
            const double* mp[NG];
            int ms[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const bool var = ra.vslot[g] >= 0;
                mp[g] = var ? col + (ra.vslot[g] * TB) * BLOCK : tab + (TROWS * g + 2 + k) * nb + j0;
                ms[g] = var ? BLOCK : 1;
            }
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const int j = j0 + t;
                double d = dcur[t];
                if (k == 1) { const double gj = gain[j]; if (gj != 1.0) d = d / gj; }  // :371
#pragma unroll
                for (int w = 0; w < MAXU; ++w)
                    if (w < ra.nu && ((ra.uinuc >> w) & 1u)) d = d - cu[w * nb + j] * tv[w];
                const double is = fast_rcp(rcur[t]);
                const double inv = is * is;
                double mrow[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) mrow[g] = mp[g][t * ms[g]];
                const int slot = __builtin_amdgcn_readfirstlane(sa.bslot[j]);   // the same for every unit: scalar branches below
                double t2v[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const double t2 = mrow[g] * inv;
                    t2v[g] = t2;
                    bv[g] += d * t2;
#pragma unroll
                    for (int h = 0; h <= g; ++h) A[g * (g + 1) / 2 + h] += t2 * mrow[h];
                }
                f0 += (eta * is) * mrow[NG - 1];
#pragma unroll
                for (int sl = 0; sl < SS; ++sl)
                    if (sl == slot) {
                        dn[sl] = d * inv; en[sl] = eta * is; iv[sl] = inv;
#pragma unroll
                        for (int g = 0; g < NG; ++g) Wv[sl][g] = t2v[g];
                    }
            }
        }
        bv[0] += f0;
        // ---- Cholesky A = L L^t (ri[g] = 1 / L_gg), yh = L^-1 b, Q_sl = L^-1 W_sl
        ok = true;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int h = 0; h <= g; ++h) {
                double sacc = A[g * (g + 1) / 2 + h];
#pragma unroll
                for (int t = 0; t < h; ++t) sacc -= A[g * (g + 1) / 2 + t] * A[h * (h + 1) / 2 + t];
                if (h == g) {
                    if (!(sacc > 0.0) || !(sacc < 1.0e300)) ok = false;
                    ri[g] = fast_rsqrt(sacc);
                } else {
                    A[g * (g + 1) / 2 + h] = sacc * ri[h];
                }
            }
        }
        if (!ok) {
            atomicAdd(not_spd, 1ull);
        } else {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                double sacc = bv[g];
#pragma unroll
                for (int t = 0; t < g; ++t) sacc -= A[g * (g + 1) / 2 + t] * bv[t];
                bv[g] = sacc * ri[g];   // yh
            }
#pragma unroll
            for (int sl = 0; sl < SS; ++sl)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    double sacc = Wv[sl][g];
#pragma unroll
                    for (int t = 0; t < g; ++t) sacc -= A[g * (g + 1) / 2 + t] * Wv[sl][t];
                    Wv[sl][g] = sacc * ri[g];   // Q_sl
                }
        }
    }
    // ---- the unit's row values (zeros for masked units and for units whose block was not positive definite)
    double rv[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) rv[e] = 0.0;
    if (live && ok) {
        {
#pragma unroll
            for (int r = 0; r < SS; ++r) {
                if (r < R) {
                    const int sl = __builtin_amdgcn_readfirstlane(sa.bslot[sa.rj[r]]);
                    double s_r = 0.0, dns = 0.0, ens = 0.0, ivs = 0.0, dot = 0.0, Qr[NG];
#pragma unroll
                    for (int w = 0; w < MAXU; ++w) s_r = (ra.rowu[r] == w) ? gl_sed<HF>(ra, w, tv, tT, tab[(TROWS * NG) * nb + sa.rj[r]]) : s_r;
#pragma unroll
                    for (int q = 0; q < SS; ++q)
                        if (q == sl) {
                            dns = dn[q]; ens = en[q]; ivs = iv[q];
#pragma unroll
                            for (int g = 0; g < NG; ++g) Qr[g] = Wv[q][g];
                        }
                    const double w_r = ((ra.rowmono >> r) & 1u) ? 1.0 : s_r;   // :857
#pragma unroll
                    for (int g = 0; g < NG; ++g) dot += Qr[g] * bv[g];
                    rv[SS * SS + r] = dns * s_r - w_r * dot;            // t[r]
                    rv[SS * SS + SS + r] = ens * w_r;                    // fluctuation sum of natural row r
                    rv[SS * SS + 2 * SS + r] = w_r * s_r * ivs;          // G[r][r]
#pragma unroll
                    for (int r2 = 0; r2 < SS; ++r2) {
                        if (r2 < R) {
                            const int sl2 = __builtin_amdgcn_readfirstlane(sa.bslot[sa.rj[r2]]);
                            double s2 = 0.0, dot2 = 0.0;
#pragma unroll
                            for (int w = 0; w < MAXU; ++w) s2 = (ra.rowu[r2] == w) ? gl_sed<HF>(ra, w, tv, tT, tab[(TROWS * NG) * nb + sa.rj[r2]]) : s2;
#pragma unroll
                            for (int q = 0; q < SS; ++q)
                                if (q == sl2) {
#pragma unroll
                                    for (int g = 0; g < NG; ++g) dot2 += Qr[g] * Wv[q][g];
                                }
                            rv[r * SS + r2] = ((sl == sl2) ? w_r * s2 * ivs : 0.0) - w_r * s2 * dot2;
                        }
                    }
                }
            }
        }
    }
    // ---- one block reduction for all row values (wave tree, then the four wave sums in order)
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const bool used = (e < SS * SS) ? (e / SS < R && e % SS < R) : ((e - SS * SS) % SS < R);   // uniform
        if (used) {
            double v = rv[e];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if ((tid & 63) == 0) wsum[e][tid >> 6] = v;
        }
    }
    __syncthreads();
    if (tid < NV) {
        // slot e of the SS-strided layout -> row of the R-strided layout the host reads
        int row = -1;
        if (tid < SS * SS) { const int r = tid / SS, r2 = tid - r * SS; if (r < R && r2 < R) row = r * R + r2; }
        else { const int q3 = (tid - SS * SS) / SS, r = (tid - SS * SS) - q3 * SS; if (r < R) row = R * R + q3 * R + r; }
        if (row >= 0) {
            double t = 0.0;
            for (int w = 0; w < BLOCK / 64; ++w) t += wsum[tid][w];
            const long long nblk = (long long)gridDim.x * gridDim.y, blk = (long long)blockIdx.y * gridDim.x + blockIdx.x;
            rowpartial[(long long)row * nblk + blk] = t;
        }
    }
}



#endif
#if DX_SCHUR_PART == 2
// Residual of the GLOBAL rows of a template group's system at the current state (dangx_schur.hip: k_schur_resid, whose
// comment defines the three row blocks) on the same schedule, for groups whose global members are templates / monopoles:
//   cu[w][j]  = template_amplitudes of template w on the bands it is NOT fitted at (removed from the data, :445-460)
//   gm[w][j]  = template_amplitudes of a MEMBER on its fitted bands (part of the model A x)
// rows [0,R): b - A x without the fluctuation term, [R,2R): the row of b, [2R,3R): the size of the terms.
// RR: rows the thread carries (4 or 8): the 3 RR row values live in registers for the whole band loop, and with 8 of them the
// amplitude kernel's four waves per SIMD cost 73 spilled registers.
template <int NG, int TB, int RR, bool HF>
__global__ __launch_bounds__(BLOCK, RR <= 4 ? (NG <= 4 ? 4 : 3) : 2) void k_schur_resid_reg(const Model* __restrict__ Mp, GroupArgs a, AmpRegArgs ra, SchurArgs sa,
                                                                           double* __restrict__ rowpartial) {
    extern __shared__ double lds[];
    __shared__ double wsum[3 * RR][BLOCK / 64];
    const Model& M = *Mp;
    const int npix = M.npix, nb = M.nbands, tid = threadIdx.x, R = sa.nrows;
    double* tab = lds;
    double* cu = lds + (TROWS * NG + 3) * nb;
    double* gm = cu + ra.nu * nb;
    double* prl = lds + (TROWS * NG + 3 + 2 * ra.nu) * nb + tid;
    double* col = prl + 3 * ra.nv * BLOCK;
    const long long u = (long long)blockIdx.x * BLOCK + tid;
    const bool in_range = u < npix;
    const int i = in_range ? (int)u : 0;
    const int k = flag_map(a.flag, (int)blockIdx.y);
    const double mk = as_global(M.mask)[i];
    double th[NG][2], av[NG], tv[MAXU], tT[MAXU];
#pragma unroll
    for (int g = 0; g < NG; ++g) av[g] = as_global(M.comp[a.gc[g]].amp)[(long long)(k - 1) * npix + i];
    gl_load(M, ra, i, k, npix, tv, tT);
#pragma unroll
    for (int v = 0; v < NG; ++v) {
        th[v][0] = th[v][1] = 0.0;
        if (v < ra.nv) {
            const Comp& c = M.comp[a.gc[ra.vcomp[v]]];
            const gcptr ix = as_global(c.idx) + (long long)(k - 1) * npix + i;
            if (c.nind > 0) th[v][0] = ix[0];
            if (c.nind > 1) th[v][1] = ix[(long long)M.nmaps * npix];
        }
    }
    for (int t = tid; t < ra.nu * nb; t += BLOCK) {
        const int w = t / nb, j = t - w * nb;
        const Comp& c = M.comp[ra.ucomp[w]];
        const bool member = (ra.umember >> w) & 1u, fitted = (c.corr_mask >> j) & 1;
        cu[t] = (((ra.uinuc >> w) & 1u) && !fitted) ? c.tamp[k - 1][j] : 0.0;
        gm[t] = (member && fitted) ? c.tamp[k - 1][j] : 0.0;
    }
    sed_table_build(M, tab, tid, BLOCK, a.gc, NG);
    const bool live = in_range && !is_masked(mk);
    if (live) {
#pragma unroll
        for (int v = 0; v < NG; ++v)
            if (v < ra.nv) { prl[(3 * v + 0) * BLOCK] = th[v][0]; prl[(3 * v + 1) * BLOCK] = th[v][1]; }
#pragma unroll 1
        for (int v = 0; v < ra.nv; ++v) {
            const Comp& c = M.comp[a.gc[ra.vcomp[v]]];
            const Prep pr = sed_prep(c, prl[(3 * v + 0) * BLOCK], prl[(3 * v + 1) * BLOCK]);
            prl[(3 * v + 0) * BLOCK] = pr.p0;
            prl[(3 * v + 1) * BLOCK] = pr.p1;
            prl[(3 * v + 2) * BLOCK] = pr.p2;
        }
    }
    __syncthreads();
    double rv[3 * RR];
#pragma unroll
    for (int e = 0; e < 3 * RR; ++e) rv[e] = 0.0;
    if (live) {
        const double* gain = tab + (TROWS * NG + 1) * nb;
        const long long bstride = (long long)M.nmaps * npix;
        const gcptr sigp = as_global(M.sig) + (long long)(k - 1) * npix + i;
        const gcptr rmsp = as_global(M.rms) + (long long)(k - 1) * npix + i;
#pragma unroll 1
        for (int j0 = 0; j0 < nb; j0 += TB) {
            bool any = false;   // uniform: a tile without a global row needs neither its maps nor its SEDs
#pragma unroll
            for (int t = 0; t < TB; ++t) any = any || sa.bslot[j0 + t] >= 0;
            if (!any) continue;
            double dcur[TB], rcur[TB];
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                dcur[t] = sigp[(j0 + t) * bstride];
                rcur[t] = rmsp[(j0 + t) * bstride];
            }
#pragma unroll 1
            for (int v = 0; v < ra.nv; ++v) {
                const Prep pr = {prl[(3 * v + 0) * BLOCK], prl[(3 * v + 1) * BLOCK], prl[(3 * v + 2) * BLOCK]};
                sed_tile<TB>(ra.vtype[v], tab, nb, NG, ra.vcomp[v], j0, pr, col + (v * TB) * BLOCK);
            }
            const double* mp[NG];
            int ms[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const bool var = ra.vslot[g] >= 0;
                mp[g] = var ? col + (ra.vslot[g] * TB) * BLOCK : tab + (TROWS * g + 2 + k) * nb + j0;
                ms[g] = var ? BLOCK : 1;
            }
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const int j = j0 + t;
                if (sa.bslot[j] < 0) continue;
                double d = dcur[t];
                if (k == 1) { const double gj = gain[j]; if (gj != 1.0) d = d / gj; }  // compute_rhs' data, :367-378
                double model = 0.0;
#pragma unroll
                for (int g = 0; g < NG; ++g) model = model + av[g] * mp[g][t * ms[g]];
#pragma unroll
                for (int w = 0; w < MAXU; ++w)
                    if (w < ra.nu) {
                        const double sw = gl_sed<HF>(ra, w, tv, tT, tab[(TROWS * NG) * nb + j]);
                        d = d - cu[w * nb + j] * sw;
                        model = model + gm[w * nb + j] * sw;
                    }
                const double is = fast_rcp(rcur[t]);
                const double inv = is * is;
#pragma unroll
                for (int r = 0; r < RR; ++r) {
                    if (r < R && (int)sa.rj[r] == j) {
                        double st = 0.0;
#pragma unroll
                        for (int w = 0; w < MAXU; ++w) st = (ra.rowu[r] == w) ? gl_sed<HF>(ra, w, tv, tT, tab[(TROWS * NG) * nb + j]) : st;
                        const double wgt = ((ra.rowmono >> r) & 1u) ? 1.0 : st;   // :857
                        const double bterm = d * inv * st, mterm = wgt * (model * inv);
                        rv[3 * r] = bterm - mterm;
                        rv[3 * r + 1] = bterm;
                        rv[3 * r + 2] = fabs(bterm) + fabs(mterm);
                    }
                }
            }
        }
    }
    // ---- block sums of the 3R values: wave tree, then the four wave sums in order (as block_row_sum)
#pragma unroll
    for (int e = 0; e < 3 * RR; ++e) {
        if (e < 3 * R) {
            double v = rv[e];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if ((tid & 63) == 0) wsum[e][tid >> 6] = v;
        }
    }
    __syncthreads();
    if (tid < 3 * R) {
        const int r = tid / 3, q3 = tid - 3 * r;
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += wsum[tid][w];
        const long long nblk = (long long)gridDim.x * gridDim.y, blk = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        rowpartial[(long long)(q3 * R + r) * nblk + blk] = t;
    }
}

template <int NG, int TB>
int launch_resid_tb(dangx_ctx* ctx, const GroupArgs& a, const AmpRegArgs& ra, const SchurArgs& sa, long long SN, double* rows_dev) {
    const int planes = flag_planes_h(a.flag), nrows = 3 * sa.nrows;
    const unsigned gx = nblocks(SN / planes), nblk = gx * planes;
    if (ensure_partial(ctx, (long long)nrows * nblk)) return 1;
    const size_t ldsz = amp_reg_lds<TB>(NG, ctx->hm.nbands, ra.nv, 2 * ra.nu);
    if (sa.nrows <= 4)
        hipLaunchKernelGGL((k_schur_resid_reg<NG, TB, 4, HFV>), dim3(gx, planes), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, a, ra, sa, ctx->partial);
    else
        hipLaunchKernelGGL((k_schur_resid_reg<NG, TB, RMAXF, HFV>), dim3(gx, planes), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, a, ra, sa, ctx->partial);
    dx_reduce_rows_to(ctx, ctx->partial, nblk, nrows, rows_dev);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}
template <int NG>
int launch_resid_ng(dangx_ctx* ctx, const GroupArgs& a, const AmpRegArgs& ra, const SchurArgs& sa, long long SN, double* rows_dev) {
    const int nb = ctx->hm.nbands, nu = 2 * ra.nu;
    const size_t most = 80u * 1024u;
    if (nb % 5 == 0 && amp_reg_lds<5>(NG, nb, ra.nv, nu) <= most) return launch_resid_tb<NG, 5>(ctx, a, ra, sa, SN, rows_dev);
    if (nb % 4 == 0 && amp_reg_lds<4>(NG, nb, ra.nv, nu) <= most) return launch_resid_tb<NG, 4>(ctx, a, ra, sa, SN, rows_dev);
    if (nb % 3 == 0 && amp_reg_lds<3>(NG, nb, ra.nv, nu) <= most) return launch_resid_tb<NG, 3>(ctx, a, ra, sa, SN, rows_dev);
    if (nb % 2 == 0 && amp_reg_lds<2>(NG, nb, ra.nv, nu) <= most) return launch_resid_tb<NG, 2>(ctx, a, ra, sa, SN, rows_dev);
    if (amp_reg_lds<1>(NG, nb, ra.nv, nu) <= most) return launch_resid_tb<NG, 1>(ctx, a, ra, sa, SN, rows_dev);
    return -1;
}

#endif
}  // namespace

// the residual pass of the Schur solve (dangx_schur.hip: k_schur_resid) on this schedule: 0 launched, 1 error, -1 not covered
#if DX_SCHUR_HF
#define DX_RESID_WORKER dx_schurreg_resid_hf1
#define DX_PASS1_WORKER dx_schurreg_pass1_hf1
#else
#define DX_RESID_WORKER dx_schurreg_resid_hf0
#define DX_PASS1_WORKER dx_schurreg_pass1_hf0
int dx_schurreg_resid_hf1(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev);
int dx_schurreg_pass1_hf1(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev);
int dx_schurreg_resid_hf0(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev);
int dx_schurreg_pass1_hf0(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev);
int dx_schurqu2_pass1_tb5(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev);
int dx_schurqu2_pass1_tb2(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev);
int dx_schurqu1_pass1_tb5(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev);
int dx_schurqu1_pass1_tb2(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev);
static bool group_has_hifit(dangx_ctx* ctx, const GroupArgs& a) {
    for (int t = 0; t < a.nt; ++t) if (ctx->desc[a.tc[t]].type == DANGX_HIFIT) return true;
    return false;
}
#if DX_SCHUR_PART == 2
int dx_launch_schur_resid_reg(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev) {
    return group_has_hifit(ctx, a) ? dx_schurreg_resid_hf1(ctx, a, sa, SN, rows_dev) : dx_schurreg_resid_hf0(ctx, a, sa, SN, rows_dev);
}
#else
int dx_launch_schur_pass1_reg(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev) {
    return group_has_hifit(ctx, a) ? dx_schurreg_pass1_hf1(ctx, a, sa, SN, rows_dev) : dx_schurreg_pass1_hf0(ctx, a, sa, SN, rows_dev);
}
#endif
#endif

#if DX_SCHUR_PART == 2
int DX_RESID_WORKER(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev) {
    AmpRegArgs ra;
    if (sa.nrows < 1 || sa.nrows > RMAXF || !template_group_args(ctx, a, ra)) return -1;
    if ((ra.uhifit != 0u) != HFV) return -1;
    for (int r = 0; r < sa.nrows; ++r) {
        const int l = a.tc[sa.rt[r]];
        for (int w = 0; w < ra.nu; ++w) if (ra.ucomp[w] == l) ra.rowu[r] = (signed char)w;
        if (ra.rowu[r] < 0) return -1;
        if (ctx->desc[l].type == DANGX_MONOPOLE) ra.rowmono |= 1u << r;
    }
    switch (a.ng) {
    case 1: return launch_resid_ng<1>(ctx, a, ra, sa, SN, rows_dev);
    case 2: return launch_resid_ng<2>(ctx, a, ra, sa, SN, rows_dev);
    case 3: return launch_resid_ng<3>(ctx, a, ra, sa, SN, rows_dev);
    case 4: return launch_resid_ng<4>(ctx, a, ra, sa, SN, rows_dev);
    case 5: return launch_resid_ng<5>(ctx, a, ra, sa, SN, rows_dev);
    case 6: return launch_resid_ng<6>(ctx, a, ra, sa, SN, rows_dev);
    default: return -1;
    }
}
#endif

#if DX_SCHUR_PART == 1
namespace {

template <int NG, int TB>
int launch_pass1_tb(dangx_ctx* ctx, const GroupArgs& a, const AmpRegArgs& ra, const SchurArgs& sa, long long SN, double* rows_dev) {
    const int planes = flag_planes_h(a.flag), R = sa.nrows, nrows = R * R + 3 * R;
    const unsigned gx = nblocks(SN / planes), nblk = gx * planes;
    if (ensure_partial(ctx, (long long)nrows * nblk)) return 1;
    const size_t ldsz = amp_reg_lds<TB>(NG, ctx->hm.nbands, ra.nv, ra.nu);
    HIPCHK(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL((k_schur_pass1_reg<NG, TB, 4, HFV>), dim3(gx, planes), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, a, ra, sa, ctx->partial, ctx->counters);
    dx_reduce_rows_to(ctx, ctx->partial, nblk, nrows, rows_dev);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}
template <int NG>
int launch_pass1_ng(dangx_ctx* ctx, const GroupArgs& a, const AmpRegArgs& ra, const SchurArgs& sa, long long SN, double* rows_dev) {
    const int nb = ctx->hm.nbands;
    const size_t most = 80u * 1024u;
    if (nb % 5 == 0 && amp_reg_lds<5>(NG, nb, ra.nv, ra.nu) <= most) return launch_pass1_tb<NG, 5>(ctx, a, ra, sa, SN, rows_dev);
    if (nb % 4 == 0 && amp_reg_lds<4>(NG, nb, ra.nv, ra.nu) <= most) return launch_pass1_tb<NG, 4>(ctx, a, ra, sa, SN, rows_dev);
    if (nb % 3 == 0 && amp_reg_lds<3>(NG, nb, ra.nv, ra.nu) <= most) return launch_pass1_tb<NG, 3>(ctx, a, ra, sa, SN, rows_dev);
    if (nb % 2 == 0 && amp_reg_lds<2>(NG, nb, ra.nv, ra.nu) <= most) return launch_pass1_tb<NG, 2>(ctx, a, ra, sa, SN, rows_dev);
    if (amp_reg_lds<1>(NG, nb, ra.nv, ra.nu) <= most) return launch_pass1_tb<NG, 1>(ctx, a, ra, sa, SN, rows_dev);
    return -1;
}

}  // namespace

// pass 1 of the Schur solve (dangx_schur.hip: k_schur_pass1) on this schedule: 0 launched, 1 error, -1 not covered (more than
// four global rows or fitted bands, a hi_fit member, bandpass-integrated bands, the textbook fluctuation term ...)
int DX_PASS1_WORKER(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev) {
#if !DX_SCHUR_HF
    {   // dangx_schurqu.hip: up to eight global rows (templates, monopoles); Q+U with equal index maps as one thread per pixel,
        // everything else as one thread per (pixel, plane)
        const int nb = ctx->hm.nbands;
        int rc = -1;
        if (a.flag == DANGX_FLAG_QU)
            rc = (nb >= 5) ? dx_schurqu2_pass1_tb5(ctx, a, sa, SN, rows_dev) : (nb >= 2) ? dx_schurqu2_pass1_tb2(ctx, a, sa, SN, rows_dev) : -1;
        if (rc < 0) rc = (nb >= 5) ? dx_schurqu1_pass1_tb5(ctx, a, sa, SN, rows_dev) : (nb >= 2) ? dx_schurqu1_pass1_tb2(ctx, a, sa, SN, rows_dev) : -1;
        if (rc >= 0) return rc;
    }
#endif
    AmpRegArgs ra;
    if (sa.nrows < 1 || sa.nrows > 4 || sa.nslots > 4 || !template_group_args(ctx, a, ra)) return -1;
    if ((ra.uhifit != 0u) != HFV) return -1;
    if (a.ml_mode == DANGX_ML_SAMPLE && a.fluct != DANGX_FLUCT_REFERENCE) return -1;
    for (int r = 0; r < sa.nrows; ++r) {
        const int l = a.tc[sa.rt[r]];
        for (int w = 0; w < ra.nu; ++w) if (ra.ucomp[w] == l) ra.rowu[r] = (signed char)w;
        if (ra.rowu[r] < 0) return -1;
        if (ctx->desc[l].type == DANGX_MONOPOLE) ra.rowmono |= 1u << r;
    }
    switch (a.ng) {
    case 1: return launch_pass1_ng<1>(ctx, a, ra, sa, SN, rows_dev);
    case 2: return launch_pass1_ng<2>(ctx, a, ra, sa, SN, rows_dev);
    case 3: return launch_pass1_ng<3>(ctx, a, ra, sa, SN, rows_dev);
    case 4: return launch_pass1_ng<4>(ctx, a, ra, sa, SN, rows_dev);
    case 5: return launch_pass1_ng<5>(ctx, a, ra, sa, SN, rows_dev);
    case 6: return launch_pass1_ng<6>(ctx, a, ra, sa, SN, rows_dev);
    default: return -1;
    }
}

#endif
