// dx_kern_fused.h -- k_amp_index (template): a CG group's solve and the first index sweep on its planes in one launch.
// Instantiated at build time for the BASELINE shapes (dangx_fused.hip), at run time for any other (bands, members) pair
// (dangx_rtc.hip).  What the kernel does and why it equals the two launches: dangx_fused.hip.
#pragma once
#include "dx_chain.h"

struct FusedArgs {
    signed char vslot[MAXG];  // LDS column slot of group member g, -1: its SED on these planes is a row of the constant table
    signed char vcomp[MAXG];  // group member of slot v
    signed char vtype[MAXG];  // its component type (power law, modified blackbody, free-free or log-normal)
    int nv;                   // members with a column
    int gself;                // group member whose index is sampled
    int cal;                  // the launch is on the temperature plane and some band has gain /= 1 or offset /= 0: the solve takes
                              // d / gain (src/dang_cg_mod.f90:371), the chains (d - offset) / gain (src/dang_sample_mod.f90:174)
};

#ifndef DX_FUSED_WAVES
#define DX_FUSED_WAVES(SP, NB) 2
#endif
#ifndef DX_FUSED_GRP
#define DX_FUSED_GRP 5
#endif
namespace dxk {

// SEDs of one varying member for the lane's NBL bands (bands jb .. jb + NBL - 1 of the model's NB) -> its LDS column; the
// expressions of k_amp_reg's sed_tile (dangx_ampreg.hip), band by band
template <int NBL>
__device__ __forceinline__ void sed_column(int type, const double* __restrict__ tab, int NB, int NG, int g, int jb, const Prep& p,
                                           double* __restrict__ colg) {
    const double* lnr = tab + (TROWS * g) * NB + jb;
    const double* cst = lnr + NB;
    const double* lnu9 = cst + NB;
    const double* nuc = tab + (TROWS * NG) * NB + jb;
    constexpr int TT = (NBL % 5 == 0) ? 5 : (NBL % 4 == 0) ? 4 : (NBL % 3 == 0) ? 3 : 1;
    if (type == DANGX_POWERLAW) {
#pragma unroll 1
        for (int j0 = 0; j0 < NBL; j0 += TT) {
#pragma unroll
            for (int t = 0; t < TT; ++t) colg[(j0 + t) * BLOCK] = exp_nr(p.p0 * lnr[j0 + t]);
        }
    } else if (type == DANGX_MBB) {  // in tiles of TT chains as the amplitude kernel does
#pragma unroll 1
        for (int j0 = 0; j0 < NBL; j0 += TT) {
            double f[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) f[t] = p.p2 * fast_rcp(exp_nr(p.p1 * nuc[j0 + t]) - 1.0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TT; ++t) colg[(j0 + t) * BLOCK] = f[t] * exp_nr(p.p0 * lnr[j0 + t]);
        }
    } else if (type == DANGX_FREEFREE) {
        const double rp1 = fast_rcp(p.p1);
#pragma unroll 1
        for (int j0 = 0; j0 < NBL; j0 += TT) {
#pragma unroll
            for (int t = 0; t < TT; ++t) colg[(j0 + t) * BLOCK] = (ff_gaunt(lnu9[j0 + t], p.p0) * rp1) * cst[j0 + t];
        }
    } else {  // DANGX_LOGNORMAL
        const double rp1 = fast_rcp(p.p1);
#pragma unroll 1
        for (int j0 = 0; j0 < NBL; j0 += TT) {
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const double l = (lnu9[j0 + t] - p.p2) * rp1;
                colg[(j0 + t) * BLOCK] = exp_sat(-0.5 * (l * l)) * cst[j0 + t];
            }
        }
    }
}

// the same with bandpass-integrated bands (one lane per pixel: band j of the lane IS band j of the model): such a band's value is
// eval_sed's tau-weighted sum over its samples (dx_sed.h: sed_bandpass), the delta bands as above one at a time
template <int NBL>
__device__ __forceinline__ void sed_column_bp(const Model& M, const Comp& c2, const double* __restrict__ tab, int NB, int NG, int g,
                                              const Prep& p, double* __restrict__ colg) {
#pragma unroll 1
    for (int j = 0; j < NBL; ++j) colg[j * BLOCK] = (M.band[j].n != 0) ? sed_bandpass(M, c2, j, p) : sed_eval_tab(c2.type, tab, NB, NG, g, j, p);
}

// LP = 2: the bands of a pixel over two adjacent lanes (lane h owns bands [h*NB/2, (h+1)*NB/2)), as the chain's lane-pair form
// (dx_chain.h).  Each lane evaluates the SED columns of its bands and accumulates its part of the normal equations; one
// cross-lane add per entry joins them (a + b on one lane, b + a on the other: the same value), both lanes factorise the same
// small system, and the chain continues as index_chain_reg's lane pairs do.  It halves the LDS columns and the registers a
// lane needs, which is what lets a 20-band, 6-member group (C5) keep two waves per SIMD.  The band sums are then associated
// as (first half) + (second half) -- the stand-alone amplitude kernel adds band by band: the last bits of the amplitudes
// differ from the two-launch form (LP = 1 instantiations are bit for bit the two launches).
template <int MODE, int SP, int NB, int NG, int LP>
__global__ __launch_bounds__(BLOCK, DX_FUSED_WAVES(SP, NB)) void k_amp_index(const Model* __restrict__ Mp, GroupArgs ga, FusedArgs fa, IndexArgs a,
                                                                                  unsigned long long* __restrict__ not_spd,
                                                                                  unsigned long long* __restrict__ accepted,
                                                                                  double* __restrict__ chi_partial) {
    constexpr int NBL = NB / LP;       // bands of this lane
    extern __shared__ double lds[];  // [constant table | per-lane columns: nv*NBL rows of SEDs + 2*nv rows of their indices]
    const Model& M = *Mp;
    const int npix = M.npix, tid = threadIdx.x;
    double* tab = lds;
    double* col = lds + (TROWS * NG + 3) * NB + tid;  // row (v*NBL + j): SED of varying member v at the lane's band j
    const long long t0 = (long long)blockIdx.x * BLOCK + tid;
    const long long u = t0 / LP;
    const int half = (int)(t0 % LP), jb = half * NBL;
    const bool in_range = u < npix;
    const int i = in_range ? (int)u : 0;
    const double mk = M.mask[i];
    sed_table_build(M, tab, tid, BLOCK, ga.gc, NG);
    __syncthreads();  // the only barrier: a thread only ever reads the columns it wrote
    double chi[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned long long nacc = 0ull;
    const Comp& c = M.comp[a.comp];
    const bool live = in_range && !is_masked(mk);
    if (in_range && !live && half == 0) {  // masked: the solve leaves x as it is (:695), the chain writes a zero index (:223, :480-483)
        double* out = c.idx + ((long long)a.nind * M.nmaps) * npix + i;
#pragma unroll
        for (int kk = 0; kk < SP; ++kk) out[(long long)(a.s1 + kk - 1) * npix] = 0.0;
    }
    if (live) {
        const BandPick<LP> pick = {half};
        RegChain<MODE, SP, NBL, LP> R;
        R.set_k(M, c, pick);
        const long long bstride = (long long)M.nmaps * npix;
        const unsigned long long gpix = (unsigned long long)(M.pix0 + i);
        const bool sample = (ga.ml_mode == DANGX_ML_SAMPLE);
#pragma unroll
        for (int kk = 0; kk < SP; ++kk) {
            const int k = a.s1 + kk;
            // ---- this plane's maps: requested first, used after the SED columns (rms parked where 1/rms will live)
            {
                const double* sigp = M.sig + (long long)(k - 1) * npix + i;
                const double* rmsp = M.rms + (long long)(k - 1) * npix + i;
#pragma unroll
                for (int j = 0; j < NBL; ++j) { R.D[kk][j] = sigp[(jb + j) * bstride]; R.ISr[kk][j] = rmsp[(jb + j) * bstride]; }
            }
            // ---- SED columns of the varying members at this plane's indices (kept from the plane before when equal)
            bool fresh = (kk == 0);
            double th0[NG], th1[NG];
#pragma unroll
            for (int v = 0; v < NG; ++v) {
                th0[v] = th1[v] = 0.0;
                if (v < fa.nv) load_theta(M, M.comp[ga.gc[fa.vcomp[v]]], i, k, th0[v], th1[v]);
            }
            if (kk > 0) {
#pragma unroll
                for (int v = 0; v < NG; ++v)
                    if (v < fa.nv) fresh = fresh || !(th0[v] == col[(fa.nv * NBL + 2 * v) * BLOCK] && th1[v] == col[(fa.nv * NBL + 2 * v + 1) * BLOCK]);
            }
            if (fresh) {
#pragma unroll
                for (int v = 0; v < NG; ++v)
                    if (v < fa.nv) { col[(fa.nv * NBL + 2 * v) * BLOCK] = th0[v]; col[(fa.nv * NBL + 2 * v + 1) * BLOCK] = th1[v]; }
#pragma unroll 1
                for (int v = 0; v < fa.nv; ++v) {
                    const Comp& c2 = M.comp[ga.gc[fa.vcomp[v]]];
                    const Prep pr = sed_prep(c2, col[(fa.nv * NBL + 2 * v) * BLOCK], col[(fa.nv * NBL + 2 * v + 1) * BLOCK]);
                    sed_column<NBL>(fa.vtype[v], tab, NB, NG, fa.vcomp[v], jb, pr, col + (v * NBL) * BLOCK);
                }
            }
            // ---- the block solve of unit (i, k): k_amp_reg's phase B and Cholesky
            double eta = 0.0, f0 = 0.0;
            if (sample) {
                double u1, u2;
                uniform2(ga.seed, ga.stream, gpix, (uint32_t)k, u1, u2);
                eta = rand_normal(0.0, 1.0, u1, u2);
            }
            double A[NG * (NG + 1) / 2], bv[NG];
#pragma unroll
            for (int q = 0; q < NG * (NG + 1) / 2; ++q) A[q] = 0.0;
#pragma unroll
            for (int g = 0; g < NG; ++g) bv[g] = 0.0;
            const double* mp[NG];
            int ms[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const bool var = fa.vslot[g] >= 0;
                mp[g] = var ? col + (fa.vslot[g] * NBL) * BLOCK : tab + (TROWS * g + 2 + k) * NB + jb;  // else csed of plane k
                ms[g] = var ? BLOCK : 1;
            }
#pragma unroll
            for (int j = 0; j < NBL; ++j) {
                double d = R.D[kk][j];
                if (SP == 1 && fa.cal) d = d / tab[(TROWS * NG + 1) * NB + jb + j];  // T / gain, no offset (:371)
                const double is = fast_rcp(R.ISr[kk][j]);
                R.set_is(kk, j, is);  // = CDIV(1.0, rms) of the chain's staging
                const double inv = is * is;
                double mrow[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) mrow[g] = mp[g][j * ms[g]];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const double t2 = mrow[g] * inv;
                    bv[g] += d * t2;
#pragma unroll
                    for (int h = 0; h <= g; ++h) A[g * (g + 1) / 2 + h] += t2 * mrow[h];
                }
                f0 += (eta * is) * mrow[NG - 1];
                // keep the scheduler from hoisting every band's LDS reads to the top (two registers each): groups of DX_FUSED_GRP
                if (j % DX_FUSED_GRP == DX_FUSED_GRP - 1) __builtin_amdgcn_sched_barrier(0);
            }
            if (LP > 1) {  // the other lane's bands
#pragma unroll
                for (int q = 0; q < NG * (NG + 1) / 2; ++q) A[q] += __shfl_xor(A[q], 1, 64);
#pragma unroll
                for (int g = 0; g < NG; ++g) bv[g] += __shfl_xor(bv[g], 1, 64);
                f0 += __shfl_xor(f0, 1, 64);
            }
            bv[0] += f0;
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
            if (ok) {
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
                if (half == 0) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) M.comp[ga.gc[g]].amp[(long long)(k - 1) * npix + i] = bv[g];
                }
            } else {  // not positive definite: counted, x keeps its value -- the chain then runs on the old amplitudes
                if (half == 0) atomicAdd(not_spd, 1ull);
#pragma unroll
                for (int g = 0; g < NG; ++g) bv[g] = M.comp[ga.gc[g]].amp[(long long)(k - 1) * npix + i];
            }
            // ---- the chain's staged plane: data_raw (:173-177) minus every other component (:180-196), in
            // component_list order = member order; the members' SEDs are the ones the solve just used
            if (SP == 1 && fa.cal) {  // data_raw = (sig - offset) / gain on the temperature plane (:174)
                const double* gn = tab + (TROWS * NG + 1) * NB + jb;
#pragma unroll
                for (int j = 0; j < NBL; ++j) R.D[kk][j] = (R.D[kk][j] - gn[NB + j]) / gn[j];
            }
            R.amp[kk] = 0.0;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g == fa.gself) { R.amp[kk] = bv[g]; continue; }
                if (!((a.others >> ga.gc[g]) & 1u)) continue;
                const double amp2 = bv[g];
                // the stride goes through an opaque copy: otherwise the 4 x NB LDS addresses of the solve's band loop are
                // kept in registers (one each) across the Cholesky just to be used again here
                // (two planes only: 256 + 6 spilled -> 231 registers, 2.95 -> 2.86 ms; one plane has the room, and its schedule
                // is better left alone: 2.17 against 2.25 ms)
                int ms2 = ms[g];
                if (SP == 2) asm volatile("" : "+s"(ms2));
#pragma unroll
                for (int j = 0; j < NBL; ++j) R.D[kk][j] -= amp2 * mp[g][j * ms2];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        double sample0, sample1;
        load_theta(M, c, i, a.s1, sample0, sample1);
        nacc = chain_finish<MODE, SP, NBL, LP>(M, a, c, R, pick, sample0, sample1, i, half, chi);
    }
    if (accepted) {
        for (int o = 32; o > 0; o >>= 1) nacc += __shfl_down(nacc, o, 64);
        if ((tid & 63) == 0 && nacc) atomicAdd(accepted, nacc);
    }
    if (chi_partial) {
        __shared__ double sh[4][BLOCK / 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double v = chi[q];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if ((tid & 63) == 0) sh[q][tid >> 6] = v;
        }
        __syncthreads();
        if (tid < 4) {
            double s = 0.0;
            for (int w = 0; w < BLOCK / 64; ++w) s += sh[tid][w];
            chi_partial[(long long)tid * gridDim.x + blockIdx.x] = s;
        }
    }
}

}  // namespace dxk
