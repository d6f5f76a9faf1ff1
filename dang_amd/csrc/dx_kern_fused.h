// dx_kern_fused.h -- k_amp_index (template): a CG group's solve and the first index sweep on its planes in one launch.
// Instantiated at build time for the BASELINE shapes (dangx_fused.hip), at run time for any other (bands, members) pair
// (dangx_rtc.hip).  What the kernel does and why it equals the two launches: dangx_fused.hip.
#pragma once
#include "dx_chain.h"

struct FusedArgs {
    signed char vslot[MAXG];  // LDS column slot of group member g, -1: its SED on these planes is a row of the constant table
    signed char vcomp[MAXG];  // group member of slot v
    signed char vtype[MAXG];  // its component type (power law or modified blackbody)
    int nv;                   // members with a column
    int gself;                // group member whose index is sampled
};

#ifndef DX_FUSED_WAVES
#define DX_FUSED_WAVES(SP, NB) 2
#endif
#ifndef DX_FUSED_GRP
#define DX_FUSED_GRP 5
#endif
namespace dxk {

// SEDs of one varying member for all NB bands -> its LDS column (k_amp_reg's sed_tile with one tile of NB bands)
template <int NB>
__device__ __forceinline__ void sed_column(int type, const double* __restrict__ tab, int NG, int g, const Prep& p,
                                           double* __restrict__ colg) {
    const double* lnr = tab + (TROWS * g) * NB;
    const double* nuc = tab + (TROWS * NG) * NB;
    constexpr int TT = (NB % 5 == 0) ? 5 : (NB % 4 == 0) ? 4 : (NB % 3 == 0) ? 3 : 1;
    if (type == DANGX_POWERLAW) {
#pragma unroll 1
        for (int j0 = 0; j0 < NB; j0 += TT) {
#pragma unroll
            for (int t = 0; t < TT; ++t) colg[(j0 + t) * BLOCK] = exp_nr(p.p0 * lnr[j0 + t]);
        }
    } else {  // DANGX_MBB, in tiles of TT chains as the amplitude kernel does
#pragma unroll 1
        for (int j0 = 0; j0 < NB; j0 += TT) {
            double f[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) f[t] = p.p2 * fast_rcp(exp_nr(p.p1 * nuc[j0 + t]) - 1.0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TT; ++t) colg[(j0 + t) * BLOCK] = f[t] * exp_nr(p.p0 * lnr[j0 + t]);
        }
    }
}

template <int MODE, int SP, int NB, int NG>
__global__ __launch_bounds__(BLOCK, DX_FUSED_WAVES(SP, NB)) void k_amp_index(const Model* __restrict__ Mp, GroupArgs ga, FusedArgs fa, IndexArgs a,
                                                                                  unsigned long long* __restrict__ not_spd,
                                                                                  unsigned long long* __restrict__ accepted,
                                                                                  double* __restrict__ chi_partial) {
    extern __shared__ double lds[];  // [constant table | per-thread columns: nv*NB rows of SEDs]
    const Model& M = *Mp;
    const int npix = M.npix, tid = threadIdx.x;
    double* tab = lds;
    double* col = lds + (TROWS * NG + 3) * NB + tid;  // row (v*NB + j): SED of varying member v at band j
    const long long u = (long long)blockIdx.x * BLOCK + tid;
    const bool in_range = u < npix;
    const int i = in_range ? (int)u : 0;
    const double mk = M.mask[i];
    sed_table_build(M, tab, tid, BLOCK, ga.gc, NG);
    __syncthreads();  // the only barrier: a thread only ever reads the columns it wrote
    double chi[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned long long nacc = 0ull;
    const Comp& c = M.comp[a.comp];
    const bool live = in_range && !is_masked(mk);
    if (in_range && !live) {  // masked: the solve leaves x as it is (:695), the chain writes a zero index (:223, :480-483)
        double* out = c.idx + ((long long)a.nind * M.nmaps) * npix + i;
#pragma unroll
        for (int kk = 0; kk < SP; ++kk) out[(long long)(a.s1 + kk - 1) * npix] = 0.0;
    }
    if (live) {
        const BandPick<1> pick = {0};
        RegChain<MODE, SP, NB, 1> R;
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
                for (int j = 0; j < NB; ++j) { R.D[kk][j] = sigp[j * bstride]; R.ISr[kk][j] = rmsp[j * bstride]; }
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
                    if (v < fa.nv) fresh = fresh || !(th0[v] == col[(fa.nv * NB + 2 * v) * BLOCK] && th1[v] == col[(fa.nv * NB + 2 * v + 1) * BLOCK]);
            }
            if (fresh) {
#pragma unroll
                for (int v = 0; v < NG; ++v)
                    if (v < fa.nv) { col[(fa.nv * NB + 2 * v) * BLOCK] = th0[v]; col[(fa.nv * NB + 2 * v + 1) * BLOCK] = th1[v]; }
#pragma unroll 1
                for (int v = 0; v < fa.nv; ++v) {
                    const Comp& c2 = M.comp[ga.gc[fa.vcomp[v]]];
                    const Prep pr = sed_prep(c2, col[(fa.nv * NB + 2 * v) * BLOCK], col[(fa.nv * NB + 2 * v + 1) * BLOCK]);
                    sed_column<NB>(fa.vtype[v], tab, NG, fa.vcomp[v], pr, col + (v * NB) * BLOCK);
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
                mp[g] = var ? col + (fa.vslot[g] * NB) * BLOCK : tab + (TROWS * g + 2 + k) * NB;  // else csed of plane k
                ms[g] = var ? BLOCK : 1;
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const double d = R.D[kk][j];  // the launcher takes this kernel only with unit gains and zero offsets (:371)
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
#pragma unroll
                for (int g = 0; g < NG; ++g) M.comp[ga.gc[g]].amp[(long long)(k - 1) * npix + i] = bv[g];
            } else {  // not positive definite: counted, x keeps its value -- the chain then runs on the old amplitudes
                atomicAdd(not_spd, 1ull);
#pragma unroll
                for (int g = 0; g < NG; ++g) bv[g] = M.comp[ga.gc[g]].amp[(long long)(k - 1) * npix + i];
            }
            // ---- the chain's staged plane: data_raw (:173-177) minus every other component (:180-196), in
            // component_list order = member order; the members' SEDs are the ones the solve just used
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
                for (int j = 0; j < NB; ++j) R.D[kk][j] -= amp2 * mp[g][j * ms2];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        double sample0, sample1;
        load_theta(M, c, i, a.s1, sample0, sample1);
        nacc = chain_finish<MODE, SP, NB, 1>(M, a, c, R, pick, sample0, sample1, i, 0, chi);
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
