// dangx_ampreg.hip -- amplitude phase, direct block solve, latency-hiding form (delta bandpasses, nothing to remove
// from the data: the case every BASELINE configuration runs).
//
// Same mathematics as k_amp_direct (dangx_amp.hip): per (pixel, plane) unit the NG x NG normal equations of
// compute_rhs / compute_Ax / compute_sample_vector (src/dang_cg_mod.f90:326-1096) are accumulated band by band and
// solved by Cholesky.  What differs is the schedule inside one wave:
//   * the data and rms of a tile of TB bands are loaded into registers at the top of the tile, before anything needs them;
//   * phase A of a tile evaluates the SEDs of the components whose spectral indices vary over the sky -- the
//     transcendental half of the kernel, which depends on the (few) index loads only -- component by component (one
//     type switch per component, TB independent exp chains in flight) into the thread's own LDS column;
//     components whose indices are spatially constant on the plane are rows of the block's constant table;
//   * phase B is a fully unrolled rank-1 update per band that reads the mixing row through one (pointer, stride) pair
//     per component -- stride 1 into the constant table (a broadcast read) or stride BLOCK into the column;
// so the HBM latency of the 2*nb map loads is covered by phase A of the same wave instead of by other waves, and
// the kernel needs no barrier after the table is built (a thread only ever reads the column it wrote).
// Measured and not kept: Q and U lanes of a pixel sharing one SED evaluation (lane pairs, -19 % vector instructions in the
// Q+U launch: 1.47 -> 1.46 ms -- the kernel waits on memory, not on issue slots), requesting the first tile in the prologue,
// requesting the next tile band by band from phase B into the registers a band has just vacated (the map registers
// then live across phase A: 22 spills, 0.78 -> 1.09 ms), tiles of two bands at five waves per SIMD (93 registers: 0.79 ms,
// no change).  tools/ubench/stream_planes.hip: the same loads and stores with no arithmetic take 0.48 ms (5.8 TB/s).
// Divisions by the rms, inside the mbb SED and in the Cholesky use v_rcp_f64 / v_rsq_f64 plus two Newton steps
// (<= 1 ulp) instead of the IEEE division / sqrt sequences (12 / 18 fp64 instructions each).
#include "dx_ampreg.h"

namespace {

template <int NG, int TB, bool HT = false, bool HF = false>
__global__ __launch_bounds__(BLOCK, NG <= 4 ? 4 : 3) void k_amp_reg(const Model* __restrict__ Mp, GroupArgs a, AmpRegArgs ra,
                                                                   unsigned long long* __restrict__ not_spd) {
    extern __shared__ double lds[];  // [constant table (| HT: nu rows of template coefficients) | per-thread columns: 3*nv rows of sed_prep state, nv*TB rows of SEDs]
    const Model& M = *Mp;
    const int npix = M.npix, nb = M.nbands, tid = threadIdx.x;
    double* tab = lds;
    double* cu = lds + (TROWS * NG + 3) * nb;          // HT: row w = coefficients of template w on this block's plane
    double* prl = lds + (TROWS * NG + 3 + (HT ? ra.nu : 0)) * nb + tid;  // row (3*v + q): sed_prep value q of the v-th varying member
    double* col = prl + 3 * ra.nv * BLOCK;            // row (v*TB + t): its SED at band j0 + t
    // grid = (pixel chunks, planes): no 64-bit division of a unit number by npix
    const long long u = (long long)blockIdx.x * BLOCK + tid;
    const bool in_range = u < npix;
    const int i = in_range ? (int)u : 0;
    const int k = flag_map(a.flag, (int)blockIdx.y);
    // ---- every load the prologue needs is issued NOW and unconditionally (a masked pixel's indices are read and dropped):
    // the mask, then the indices behind an `if (live)`, then the first tile would be three HBM latencies in series for a
    // wave that lives ~4 us; the table build and the random deviate below run under them
    const double mk = as_global(M.mask)[i];
    double th[NG][2];
#pragma unroll
    for (int v = 0; v < NG; ++v) {
        th[v][0] = th[v][1] = 0.0;
        if (v < ra.nv) {
            const Comp& c = M.comp[a.gc[ra.vcomp[v]]];
            const gcptr ix = as_global(c.idx) + (long long)(k - 1) * npix + i;   // c%indices(i,k,:), as load_theta
            if (c.nind > 0) th[v][0] = ix[0];
            if (c.nind > 1) th[v][1] = ix[(long long)M.nmaps * npix];
        }
    }
    double tv[MAXU], tT[MAXU];
    if (HT) {
        gl_load(M, ra, i, k, npix, tv, tT);
        for (int t = tid; t < ra.nu * nb; t += BLOCK) {
            const int w = t / nb, j = t - w * nb;
            const Comp& c = M.comp[ra.ucomp[w]];
            const bool member = (ra.umember >> w) & 1u, inuc = (ra.uinuc >> w) & 1u, fitted = (c.corr_mask >> j) & 1;
            // a member: its new amplitude on the bands it is fitted at; a template / monopole, member or not: its amplitude on the others
            cu[t] = ((member && fitted) || (inuc && !fitted)) ? c.tamp[k - 1][j] : 0.0;
        }
    }
    sed_table_build(M, tab, tid, BLOCK, a.gc, NG);
#ifdef DX_AMP_TABLE_TWICE   // timing experiment: what one table build costs
    __syncthreads();
    sed_table_build(M, tab, tid, BLOCK, a.gc, NG);
#endif
    // ---- fluctuation term of the reference: ONE eta per unit (:258-260), no memory dependency
    const bool sample = (a.ml_mode == DANGX_ML_SAMPLE);
    const unsigned long long gpix = (unsigned long long)(M.pix0 + i);
    double eta = 0.0, f0 = 0.0;
    if (sample) {
        double u1, u2;
        uniform2(a.seed, a.stream, gpix, (uint32_t)k, u1, u2);
        eta = rand_normal(0.0, 1.0, u1, u2);
    }
    const bool live = in_range && !is_masked(mk);  // masked rows/cols are zero: x keeps its value (:695)

    // ---- per-pixel SED state of the varying members -> the thread's LDS rows (the indices pass through the rows so that
    // ONE rolled copy of sed_prep serves every member: unrolled, its type switches cost the scalar registers the
    // polynomial coefficients of exp live in)
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
    __syncthreads();  // constant table complete (the only barrier; a thread only ever reads the columns it wrote)
    if (!live) return;

    double A[NG * (NG + 1) / 2], bv[NG];
#pragma unroll
    for (int q = 0; q < NG * (NG + 1) / 2; ++q) A[q] = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) bv[g] = 0.0;
    const double* gain = tab + (TROWS * NG + 1) * nb;
    const long long bstride = (long long)M.nmaps * npix;
    const gcptr sigp = as_global(M.sig) + (long long)(k - 1) * npix + i;
    const gcptr rmsp = as_global(M.rms) + (long long)(k - 1) * npix + i;

#pragma unroll 1
    for (int j0 = 0; j0 < nb; j0 += TB) {
        // this tile's maps: issued now, consumed by phase B -- phase A below covers their latency (requesting the first
        // tile in the prologue as well was measured: no gain, 8 registers more)
        double dcur[TB], rcur[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            dcur[t] = sigp[(j0 + t) * bstride];
            rcur[t] = rmsp[(j0 + t) * bstride];
        }
        // phase A: SEDs of the varying members for this tile -> own LDS columns
#pragma unroll 1
        for (int v = 0; v < ra.nv; ++v) {
            const Prep pr = {prl[(3 * v + 0) * BLOCK], prl[(3 * v + 1) * BLOCK], prl[(3 * v + 2) * BLOCK]};
            sed_tile<TB>(ra.vtype[v], tab, nb, NG, ra.vcomp[v], j0, pr, col + (v * TB) * BLOCK);
        }
        // phase B: rank-1 updates, band by band; the mixing row through one (pointer, stride) pair per member
        const double* mp[NG];
        int ms[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const bool var = ra.vslot[g] >= 0;
            mp[g] = var ? col + (ra.vslot[g] * TB) * BLOCK : tab + (TROWS * g + 2 + k) * nb + j0;  // else csed of plane k
            ms[g] = var ? BLOCK : 1;
        }
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            double d = dcur[t];
            if (k == 1) { const double gj = gain[j0 + t]; if (gj != 1.0) d = d / gj; }  // :371
            if (HT) {
#pragma unroll
                for (int w = 0; w < MAXU; ++w)
                    if (w < ra.nu) d = d - cu[w * nb + j0 + t] * gl_sed<HF>(ra, w, tv, tT, tab[(TROWS * NG) * nb + j0 + t]);
            }
            const double is = fast_rcp(rcur[t]);
            const double inv = is * is;
            double mrow[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) mrow[g] = mp[g][t * ms[g]];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const double t2 = mrow[g] * inv;
                bv[g] += d * t2;  // b = T^t N^-1 d, :489-508
#pragma unroll
                for (int h = 0; h <= g; ++h) A[g * (g + 1) / 2 + h] += t2 * mrow[h];  // T^t N^-1 T
            }
            f0 += (eta * is) * mrow[NG - 1];  // :1033-1040: slot 0 only, the LAST component's SED product
        }
    }
    bv[0] += f0;

    // ---- Cholesky A = L L^t (packed lower triangle; ri[g] = 1/L_gg, the diagonal itself is never needed)
    bool ok = true;
    double ri[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int h = 0; h <= g; ++h) {
            double s = A[g * (g + 1) / 2 + h];
#pragma unroll
            for (int t = 0; t < h; ++t) s -= A[g * (g + 1) / 2 + t] * A[h * (h + 1) / 2 + t];
            if (h == g) {
                if (!(s > 0.0) || !(s < 1.0e300)) ok = false;
                ri[g] = fast_rsqrt(s);
            } else {
                A[g * (g + 1) / 2 + h] = s * ri[h];
            }
        }
    }
    if (!ok) {
        atomicAdd(not_spd, 1ull);
        return;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        double s = bv[g];
#pragma unroll
        for (int t = 0; t < g; ++t) s -= A[g * (g + 1) / 2 + t] * bv[t];
        bv[g] = s * ri[g];
    }
#pragma unroll
    for (int g = NG - 1; g >= 0; --g) {
        double s = bv[g];
#pragma unroll
        for (int t = g + 1; t < NG; ++t) s -= A[t * (t + 1) / 2 + g] * bv[t];
        bv[g] = s * ri[g];
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) as_global_w(M.comp[a.gc[g]].amp)[(long long)(k - 1) * npix + i] = bv[g];  // unpack, :1327-1354
}

// update_sky_model + compute_chisq (src/dang_data_mod.f90:339-396, 494-526) of ONE plane on the same schedule: the plane's
// components with a non-zero amplitude play the group (sky = sum of amplitude * SED in component order), the residual replaces
// the normal equations.  The run-time-typed kernel of dangx_entry.hip (k_sky_chisq) spends 1.36 ms per C3 plane on the same
// numbers; this one is bound by the 2 nb map loads like k_amp_reg.  Block partials -> partial[blockIdx.x].
// GT: `template` components with a signal on the plane are part of the sky model too -- template_amplitudes(band, map) * template(pix,
// map) (eval_signal, src/dang_component_mod.f90:754-776); their coefficients sit behind the table in MAXU zero-padded rows and are
// read unconditionally.  (A monopole is NOT: update_sky_model turns it into the band offsets, src/dang_data_mod.f90:357-361.)
template <int NG, int TB, bool GT = false>
__global__ __launch_bounds__(BLOCK, NG <= 4 ? 4 : 3) void k_chisq_reg(const Model* __restrict__ Mp, GroupArgs a, AmpRegArgs ra,
                                                                     double* __restrict__ partial) {
    extern __shared__ double lds[];  // as k_amp_reg
    __shared__ double wsum[BLOCK / 64];
    const Model& M = *Mp;
    const int npix = M.npix, nb = M.nbands, tid = threadIdx.x;
    double* tab = lds;
    double* cu = lds + (TROWS * NG + 3) * nb;
    double* prl = lds + (TROWS * NG + 3 + (GT ? MAXU : 0)) * nb + tid;
    double* col = prl + 3 * ra.nv * BLOCK;
    const long long u = (long long)blockIdx.x * BLOCK + tid;
    const bool in_range = u < npix;
    const int i = in_range ? (int)u : 0;
    const int k = flag_map(a.flag, 0);
    const double mk = as_global(M.mask)[i];
    double th[NG][2], av[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) av[g] = as_global(M.comp[a.gc[g]].amp)[(long long)(k - 1) * npix + i];
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
    double tv[MAXU];
    if (GT) {
#pragma unroll
        for (int w = 0; w < MAXU; ++w) tv[w] = (w < ra.nu) ? as_global(M.comp[ra.ucomp[w]].tmpl)[(long long)(k - 1) * npix + i] : 0.0;
        for (int t = tid; t < MAXU * nb; t += BLOCK) {
            const int w = t / nb, j = t - w * nb;
            cu[t] = (w < ra.nu) ? M.comp[ra.ucomp[w]].tamp[k - 1][j] : 0.0;
        }
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
    double chi = 0.0;
    if (live) {
        const double* gain = tab + (TROWS * NG + 1) * nb;
        const double* offs = gain + nb;
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
                double d = dcur[t];
                if (k == 1) {  // res = (sig - offset)/gain - sky on the temperature plane, :379-391
                    const double gj = gain[j0 + t], oj = offs[j0 + t];
                    if (gj != 1.0 || oj != 0.0) d = (d - oj) / gj;
                }
                double sky = 0.0;
#pragma unroll
                for (int g = 0; g < NG; ++g) sky = sky + av[g] * mp[g][t * ms[g]];  // component order, :349-356
                if (GT) {
#pragma unroll
                    for (int w = 0; w < MAXU; ++w) sky = sky + cu[w * nb + j0 + t] * tv[w];
                }
                const double r = (d - sky) * fast_rcp(rcur[t]);
                chi = fma(r, r, chi);  // :505-523
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) chi += __shfl_down(chi, o, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = chi;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += wsum[w];
        partial[blockIdx.x] = t;
    }
}


template <int NG, int TB, bool HT = false, bool HF = false>
int launch_tb(dangx_ctx* ctx, const GroupArgs& a, const AmpRegArgs& ra, long long SN) {
    const size_t ldsz = amp_reg_lds<TB>(NG, ctx->hm.nbands, ra.nv, HT ? ra.nu : 0);
    Timed t(ctx, DANGX_K_AMP_DIRECT);
    const int planes = flag_planes_h(a.flag);
    hipLaunchKernelGGL((k_amp_reg<NG, TB, HT, HF>), dim3(nblocks(SN / planes), planes), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, a, ra, ctx->counters);
    return 0;
}

// band tile: the largest of 5 / 4 / 3 dividing nb whose LDS footprint lets as many blocks stay resident as the
// registers allow (4 for NG <= 4, else 3); failing that the largest that leaves two blocks per CU
template <int NG, bool HT = false, bool HF = false>
int launch_ng(dangx_ctx* ctx, const GroupArgs& a, const AmpRegArgs& ra, long long SN) {
    const int nb = ctx->hm.nbands, nu = HT ? ra.nu : 0;
    const size_t want = (160u * 1024u) / (NG <= 4 ? 4 : 3), most = 80u * 1024u;
    for (const size_t cap : {want, most}) {
        if (nb % 5 == 0 && amp_reg_lds<5>(NG, nb, ra.nv, nu) <= cap) return launch_tb<NG, 5, HT, HF>(ctx, a, ra, SN);
        if (nb % 4 == 0 && amp_reg_lds<4>(NG, nb, ra.nv, nu) <= cap) return launch_tb<NG, 4, HT, HF>(ctx, a, ra, SN);
        if (nb % 3 == 0 && amp_reg_lds<3>(NG, nb, ra.nv, nu) <= cap) return launch_tb<NG, 3, HT, HF>(ctx, a, ra, SN);
        // band counts that 3, 4 and 5 do not divide (7, 11, 13, 14 ...): tiles of two bands, or of one
        if (nb % 5 && nb % 4 && nb % 3) {
            if (nb % 2 == 0 && amp_reg_lds<2>(NG, nb, ra.nv, nu) <= cap) return launch_tb<NG, 2, HT, HF>(ctx, a, ra, SN);
            if (nb % 2 && amp_reg_lds<1>(NG, nb, ra.nv, nu) <= cap) return launch_tb<NG, 1, HT, HF>(ctx, a, ra, SN);
        }
    }
    return -1;  // the LDS-column kernel is the better fit
}

template <int NG, int TB>
int launch_chi_tb(dangx_ctx* ctx, const GroupArgs& a, const AmpRegArgs& ra, double* partial) {
    if (ra.nu > 0) {
        const size_t ldsz = amp_reg_lds<TB>(NG, ctx->hm.nbands, ra.nv, MAXU);
        hipLaunchKernelGGL((k_chisq_reg<NG, TB, true>), dim3(nblocks(ctx->hm.npix)), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, a, ra, partial);
    } else {
        const size_t ldsz = amp_reg_lds<TB>(NG, ctx->hm.nbands, ra.nv);
        hipLaunchKernelGGL((k_chisq_reg<NG, TB>), dim3(nblocks(ctx->hm.npix)), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, a, ra, partial);
    }
    return 0;
}
template <int NG>
int launch_chi_ng(dangx_ctx* ctx, const GroupArgs& a, const AmpRegArgs& ra, double* partial) {
    const int nb = ctx->hm.nbands, nu = ra.nu > 0 ? MAXU : 0;
    const size_t want = (160u * 1024u) / (NG <= 4 ? 4 : 3), most = 80u * 1024u;
    for (const size_t cap : {want, most}) {
        if (nb % 5 == 0 && amp_reg_lds<5>(NG, nb, ra.nv, nu) <= cap) return launch_chi_tb<NG, 5>(ctx, a, ra, partial);
        if (nb % 4 == 0 && amp_reg_lds<4>(NG, nb, ra.nv, nu) <= cap) return launch_chi_tb<NG, 4>(ctx, a, ra, partial);
        if (nb % 3 == 0 && amp_reg_lds<3>(NG, nb, ra.nv, nu) <= cap) return launch_chi_tb<NG, 3>(ctx, a, ra, partial);
        if (nb % 5 && nb % 4 && nb % 3) {
            if (nb % 2 == 0 && amp_reg_lds<2>(NG, nb, ra.nv, nu) <= cap) return launch_chi_tb<NG, 2>(ctx, a, ra, partial);
            if (nb % 2 && amp_reg_lds<1>(NG, nb, ra.nv, nu) <= cap) return launch_chi_tb<NG, 1>(ctx, a, ra, partial);
        }
    }
    return -1;
}

}  // namespace

// Pass 2 of the Schur solve of a group with template / monopole members (dangx_schur.hip: k_schur_pass2) on this kernel's
// schedule: the block solve of the diffuse members with the global members' signal -- at the amplitudes the host has just
// solved for -- and every other template's unfitted bands removed from the data.  0 when launched, -1 when the group needs
// the run-time-typed pass (a hi_fit member, bandpass-integrated bands, other components on the planes, more than MAXU
// templates / monopoles in the model; dx_ampreg.h: template_group_args).  Passes 1 and 3: dangx_schurreg.hip.
int dx_launch_amp_reg_templates(dangx_ctx* ctx, const GroupArgs& a, long long SN) {
    if (a.ml_mode == DANGX_ML_SAMPLE && a.fluct != DANGX_FLUCT_REFERENCE) return -1;
    AmpRegArgs ra;
    if (!template_group_args(ctx, a, ra)) return -1;
    switch (a.ng) {
    case 1: return ra.uhifit ? launch_ng<1, true, true>(ctx, a, ra, SN) : launch_ng<1, true>(ctx, a, ra, SN);
    case 2: return ra.uhifit ? launch_ng<2, true, true>(ctx, a, ra, SN) : launch_ng<2, true>(ctx, a, ra, SN);
    case 3: return ra.uhifit ? launch_ng<3, true, true>(ctx, a, ra, SN) : launch_ng<3, true>(ctx, a, ra, SN);
    case 4: return ra.uhifit ? launch_ng<4, true, true>(ctx, a, ra, SN) : launch_ng<4, true>(ctx, a, ra, SN);
    case 5: return ra.uhifit ? launch_ng<5, true, true>(ctx, a, ra, SN) : launch_ng<5, true>(ctx, a, ra, SN);
    case 6: return ra.uhifit ? launch_ng<6, true, true>(ctx, a, ra, SN) : launch_ng<6, true>(ctx, a, ra, SN);
    default: return -1;
    }
}

// returns 0 when launched, -1 when this form does not cover the case (the caller falls back to k_amp_direct)
int dx_launch_amp_reg(dangx_ctx* ctx, const GroupArgs& a, long long SN) {
    if (!ctx->hm.all_delta || a.no != 0 || a.nuc != 0 || a.nt != 0) return -1;
    // the textbook fluctuation term (one normal per band) stays with k_amp_direct: a Philox call per band does not fit
    // the unrolled phase B without spilling
    if (a.ml_mode == DANGX_ML_SAMPLE && a.fluct != DANGX_FLUCT_REFERENCE) return -1;
    AmpRegArgs ra;
    if (amp_reg_members(ctx, a, ra)) return -1;
    switch (a.ng) {
    case 1: return launch_ng<1>(ctx, a, ra, SN);
    case 2: return launch_ng<2>(ctx, a, ra, SN);
    case 3: return launch_ng<3>(ctx, a, ra, SN);
    case 4: return launch_ng<4>(ctx, a, ra, SN);
    case 5: return launch_ng<5>(ctx, a, ra, SN);
    case 6: return launch_ng<6>(ctx, a, ra, SN);
    default: return -1;
    }
}

// chi^2 block partials of plane k (1..nmaps) into partial[0 .. nblocks(npix)): 0 when launched, -1 when the plane needs the
// run-time-typed kernel (bandpass-integrated bands, hi_fit or T_cmb components, more than four templates or six components with an
// amplitude on the plane, none at all)
int dx_launch_chisq_reg(dangx_ctx* ctx, int k, double* partial) {
    static const bool enabled = [] { const char* e = getenv("DANGX_CHISQ_FAST"); return !(e && e[0] == '0'); }();  // A/B switch
    if (!enabled || k < 1 || k > ctx->dims.nmaps) return -1;
    for (int j = 0; j < ctx->hm.nbands; ++j)
        if (ctx->hm.band[j].n != 0) return -1;                   // (hm.all_delta is also 0 when the model holds a global-type component)
    GroupArgs a = {};
    a.ng = 0;
    AmpRegArgs ra;
    ra.nv = 0; ra.nu = 0; ra.umember = 0u; ra.rowmono = 0u; ra.uinuc = 0u; ra.uhifit = 0u;
    for (int w = 0; w < MAXU; ++w) ra.ucomp[w] = 0;
    for (int l = 0; l < ctx->hm.ncomp; ++l) {
        const Comp& c = ctx->hm.comp[l];
        if (c.type == DANGX_TCMB) return -1;                     // a signal without an amplitude: the run-time-typed kernel
        if (c.type == DANGX_MONOPOLE) continue;                  // the band offsets, not a term of the sky model (:357-361)
        if (c.type == DANGX_TEMPLATE) {                          // template_amplitudes * template: a term wherever the map is not zero
            if (!((ctx->tmpl_nz[l] >> (k - 1)) & 1u)) continue;
            if (ra.nu >= MAXU) return -1;
            ra.ucomp[ra.nu++] = l;
            continue;
        }
        if (c.type == DANGX_HIFIT) {                             // (a per-pixel Planck factor: the run-time-typed kernel where it has a signal)
            if (!((ctx->tmpl_nz[l] >> (k - 1)) & 1u)) continue;
            return -1;
        }
        if (!((ctx->plane_nz[l] >> (k - 1)) & 1u)) continue;     // amplitude 0 everywhere on the plane: 0 * sed
        if (c.type < DANGX_POWERLAW || c.type > DANGX_CMB || a.ng >= 6) return -1;
        a.gc[a.ng++] = l;
    }
    if (a.ng < 1) return -1;
    a.flag = (k == 1) ? DANGX_FLAG_T : (k == 2) ? DANGX_FLAG_Q : DANGX_FLAG_U;
    for (int r = 0; r < 8; ++r) ra.rowu[r] = -1;
    for (int g = 0; g < MAXG; ++g) { ra.vslot[g] = -1; ra.vcomp[g] = 0; ra.vtype[g] = 0; }
    for (int g = 0; g < a.ng; ++g) {
        const Comp& c = ctx->hm.comp[a.gc[g]];
        if (!(((unsigned)c.const_planes >> (k - 1)) & 1u)) {
            ra.vcomp[ra.nv] = (signed char)g; ra.vtype[ra.nv] = (signed char)c.type;
            ra.vslot[g] = (signed char)ra.nv++;
        }
    }
    switch (a.ng) {
    case 1: return launch_chi_ng<1>(ctx, a, ra, partial);
    case 2: return launch_chi_ng<2>(ctx, a, ra, partial);
    case 3: return launch_chi_ng<3>(ctx, a, ra, partial);
    case 4: return launch_chi_ng<4>(ctx, a, ra, partial);
    case 5: return launch_chi_ng<5>(ctx, a, ra, partial);
    case 6: return launch_chi_ng<6>(ctx, a, ra, partial);
    default: return -1;
    }
}
