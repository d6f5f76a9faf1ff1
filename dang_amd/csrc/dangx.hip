// dangx.hip -- libdangx.so: hand-written HIP (gfx950 / CDNA4) kernels for dang's
// Gibbs inner loop and the C ABI declared in include/dangx.h.
//
// Design: one thread owns one (pixel, Stokes plane) unit -- the
// reference's global CG system is block diagonal for diffuse components, so the
// amplitude phase is a single streaming pass (mixing rows -> normal equations ->
// Cholesky) and the index phase a single pass with the Metropolis chain held in
// LDS/registers.  All map arrays are pixel-major, so a wavefront's 64 lanes read 64
// consecutive doubles (512 B) per load.  Everything is fp64.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dangx.h"
#include "dx_model.h"
#include "dx_rng.h"
#include "dx_sed.h"

using namespace dx;

// ======================================================================= kernels

namespace {

constexpr int BLOCK = 256;

struct GroupArgs {
    int ng;          // sampled diffuse components of the group
    int gc[MAXG];    // their component indices, in component_list order
    int no;          // components NOT solved for (removed from the data)
    int oc[MAXC];
    int flag;        // one poltype bit
    int ml_mode, fluct;
    unsigned long long seed, stream;
};

__device__ __forceinline__ int flag_nplanes(int flag) { return (flag & DANGX_FLAG_QU) ? 2 : 1; }
// src/dang_cg_mod.f90:357-363 and the flag-8 branches (:488-494): plane p -> map number
__device__ __forceinline__ int flag_map(int flag, int p) {
    if (flag & DANGX_FLAG_QU) return 2 + p;
    if (flag & DANGX_FLAG_T) return 1;
    if (flag & DANGX_FLAG_Q) return 2;
    return 3;
}

// data(i,k,j) of compute_rhs (src/dang_cg_mod.f90:367-378, 427-443): the band map with every
// component that is not solved for removed.  a.oc lists only the components whose amplitude
// plane may be non-zero (the host tracks all-zero planes; subtracting 0*sed is skipped, which
// differs from the reference only if that sed is not finite).
__device__ __forceinline__ double remove_others(const Model& M, const GroupArgs& a, int i, int k, int j, double d) {
    for (int o = 0; o < a.no; ++o) {
        const Comp& c = M.comp[a.oc[o]];
        const double amp = c.amp[(long long)(k - 1) * M.npix + i];
        double t0, t1;
        load_theta(M, c, i, k, t0, t1);
        d = d - signal_of(c, amp, sed_eval(M, c, j, sed_prep(c, t0, t1)));
    }
    return d;
}
__device__ __forceinline__ double rhs_data(const Model& M, const GroupArgs& a, int i, int k, int j) {
    double d = M.sig[((long long)j * M.nmaps + (k - 1)) * M.npix + i];
    if (k == 1) d = d / M.gain[j];
    return remove_others(M, a, i, k, j, d);
}

// ---------------------------------------------------------------------------
// Amplitude phase, direct solve.  Replaces compute_rhs + cg_search (compute_Ax,
// compute_sample_vector) + unpack_amplitudes (src/dang_cg_mod.f90:166-171) for
// groups of diffuse components: every term of compute_Ax couples only index i
// (:697-704, :813-820), so A^t N^-1 A is one NG x NG SPD block per (pixel, plane).
// Per unit: stream the nb bands once, accumulate the lower triangle of the block
// and the right-hand side in registers, Cholesky, two triangular solves, store.
// FAST: every band is a delta bandpass and nothing has to be removed from the data (the common case);
// the generic instantiation carries the bandpass-integrated SEDs and the other-component removal.
template <int NG, bool FAST>
__global__ __launch_bounds__(BLOCK, (FAST && NG <= 4) ? 3 : 1) void k_amp_direct(const Model* __restrict__ Mp, GroupArgs a,
                                                      unsigned long long* __restrict__ not_spd) {
    extern __shared__ double lds[];  // [table | D(j) and IS(j) columns: (2*nb) x blockDim]
    const Model& M = *Mp;
    const int npix = M.npix, nb = M.nbands, BS = blockDim.x, tid = threadIdx.x;
    double* tab = lds;  // rows of the group's NG components only
    double* col = lds + (TROWS * NG + 3) * nb;
    sed_table_build(M, tab, tid, BS, a.gc, NG);
    const long long u = (long long)blockIdx.x * BS + tid;
    const bool in_range = u < (long long)flag_nplanes(a.flag) * npix;
    const int p = in_range ? (int)(u / npix) : 0;
    const int i = in_range ? (int)(u - (long long)p * npix) : 0;
    const int k = flag_map(a.flag, p);
    const bool live = in_range && !is_masked(M.mask[i]);  // masked rows/cols are zero: x keeps its value (:695)

    // ---- phase 1: every HBM load of this unit is issued up front (d, rms for all bands in tiles of 5;
    // the group's spectral indices), results parked in LDS columns
    Prep pr[NG];
    int ty[NG], gl[NG];
    bool cs[NG];  // SED is a per-band constant on this plane (spatially constant indices)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        gl[g] = a.gc[g]; ty[g] = M.comp[gl[g]].type;
        cs[g] = FAST && ((M.comp[gl[g]].const_planes >> (k - 1)) & 1);
    }
    if (live) {
        double th0[NG], th1[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            th0[g] = th1[g] = 0.0;
            if (!cs[g]) load_theta(M, M.comp[gl[g]], i, k, th0[g], th1[g]);
        }
        const long long bstride = (long long)M.nmaps * npix;
        const double* sigp = M.sig + (long long)(k - 1) * npix + i;
        const double* rmsp = M.rms + (long long)(k - 1) * npix + i;
#pragma unroll 2
        for (int j0 = 0; j0 < nb; j0 += 5) {
            double dv[5], rv[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const int j = (j0 + t < nb) ? j0 + t : nb - 1;
                dv[t] = sigp[j * bstride];
                rv[t] = rmsp[j * bstride];
            }
#pragma unroll
            for (int t = 0; t < 5; ++t)
                if (j0 + t < nb) {
                    col[(j0 + t) * BS + tid] = dv[t];
                    col[(nb + j0 + t) * BS + tid] = 1.0 / rv[t];
                }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            pr[g] = Prep{0.0, 0.0, 0.0};
            if (!cs[g]) pr[g] = sed_prep(M.comp[gl[g]], th0[g], th1[g]);
        }
    }
    __syncthreads();  // constant table complete
    if (!live) return;

    double A[NG * (NG + 1) / 2], bv[NG], mrow[NG];
#pragma unroll
    for (int q = 0; q < NG * (NG + 1) / 2; ++q) A[q] = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) bv[g] = 0.0;
    const bool sample = (a.ml_mode == DANGX_ML_SAMPLE);
    const unsigned long long gpix = (unsigned long long)(M.pix0 + i);
    double eta = 0.0, f0 = 0.0;
    if (sample && a.fluct == DANGX_FLUCT_REFERENCE) {
        double u1, u2;
        uniform2(a.seed, a.stream, gpix, (uint32_t)k, u1, u2);
        eta = rand_normal(0.0, 1.0, u1, u2);  // eta(i), :258-260: ONE draw per unit, reused per band
    }
    // ---- phase 2: rolled band loop (one copy of the SED code per group component)
    const double* gain = tab + (TROWS * NG + 1) * nb;
#pragma unroll 1
    for (int j = 0; j < nb; ++j) {
        double d = col[j * BS + tid];
        const double is = col[(nb + j) * BS + tid];
        if (k == 1) d = d / gain[j];
        if (!FAST) d = remove_others(M, a, i, k, j, d);
        const double inv = is * is;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            mrow[g] = !FAST ? sed_eval(M, M.comp[gl[g]], j, pr[g])
                      : cs[g] ? sed_const_tab(tab, nb, g, k, j) : sed_eval_tab(ty[g], tab, nb, NG, g, j, pr[g]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const double t2 = mrow[g] * inv;
            bv[g] += d * t2;  // b = T^t N^-1 d, :489-508
#pragma unroll
            for (int h = 0; h <= g; ++h) A[g * (g + 1) / 2 + h] += t2 * mrow[h];  // T^t N^-1 T
        }
        if (sample) {
            if (a.fluct == DANGX_FLUCT_REFERENCE) {
                // :1033-1040 '=' without component offset: only slot 0 receives the term,
                // holding the LAST component's SED product
                f0 += (eta * is) * mrow[NG - 1];
            } else {
                double u1, u2;
                uniform2(a.seed, a.stream, gpix, (uint32_t)(k + 4 * (j + 1)), u1, u2);
                const double ej = rand_normal(0.0, 1.0, u1, u2) * is;
#pragma unroll
                for (int g = 0; g < NG; ++g) bv[g] += ej * mrow[g];
            }
        }
    }
    bv[0] += f0;

    // Cholesky A = L L^t in place (packed lower triangle)
    bool ok = true;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int h = 0; h <= g; ++h) {
            double s = A[g * (g + 1) / 2 + h];
#pragma unroll
            for (int t = 0; t < h; ++t) s -= A[g * (g + 1) / 2 + t] * A[h * (h + 1) / 2 + t];
            if (h == g) {
                if (!(s > 0.0)) ok = false;
                A[g * (g + 1) / 2 + g] = sqrt(s);
            } else {
                A[g * (g + 1) / 2 + h] = s / A[h * (h + 1) / 2 + h];
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
        bv[g] = s / A[g * (g + 1) / 2 + g];
    }
#pragma unroll
    for (int g = NG - 1; g >= 0; --g) {
        double s = bv[g];
#pragma unroll
        for (int t = g + 1; t < NG; ++t) s -= A[t * (t + 1) / 2 + g] * bv[t];
        bv[g] = s / A[g * (g + 1) / 2 + g];
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) M.comp[a.gc[g]].amp[(long long)(k - 1) * npix + i] = bv[g];  // unpack, :1327-1354
}

// ---------------------------------------------------------------------------
// Secondary seams on the reference's packed vectors (device CG = parity mode).
// Packing (src/dang_cg_mod.f90:1216-1243): x = [c1: plane0(npix), plane1(npix) | c2: ...],
// so element (comp g, unit u) is x[g*S*npix + u] with u = p*npix + i.

// compute_rhs, src/dang_cg_mod.f90:326-596 (diffuse branch)
template <int NG>
__global__ __launch_bounds__(BLOCK) void k_rhs(const Model* __restrict__ Mp, GroupArgs a, double* __restrict__ b) {
    const Model& M = *Mp;
    const int npix = M.npix;
    const long long SN = (long long)flag_nplanes(a.flag) * npix;
    const long long u = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (u >= SN) return;
    const int p = (int)(u / npix), i = (int)(u - (long long)p * npix), k = flag_map(a.flag, p);
    double acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = 0.0;
    if (M.mask[i] != 0.0) {  // :474 tests ==0 only
        const bool removed = !is_masked(M.mask[i]);  // :434 other components are removed only off-mask
        Prep pr[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double t0, t1;
            load_theta(M, M.comp[a.gc[g]], i, k, t0, t1);
            pr[g] = sed_prep(M.comp[a.gc[g]], t0, t1);
        }
        for (int j = 0; j < M.nbands; ++j) {
            double d;
            if (removed) d = rhs_data(M, a, i, k, j);
            else {
                d = M.sig[((long long)j * M.nmaps + (k - 1)) * npix + i];
                if (k == 1) d = d / M.gain[j];
            }
            const double rms = M.rms[((long long)j * M.nmaps + (k - 1)) * npix + i];
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = acc[g] + (d * sed_eval(M, M.comp[a.gc[g]], j, pr[g])) / (rms * rms);
        }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) b[(long long)g * SN + u] = acc[g];
}

// compute_Ax, src/dang_cg_mod.f90:598-911 (diffuse branch), same operation order per unit:
// temp1 = sum_c x_c*sed_c ; temp1 /= rms**2 ; res_c += temp1*sed_c, band by band.
// Also returns the block-local partial of dot(x, res) for cg_search's sum(d*q) (:297).
template <int NG>
__global__ __launch_bounds__(BLOCK) void k_Ax(const Model* __restrict__ Mp, GroupArgs a, const double* __restrict__ x,
                                              double* __restrict__ res, double* __restrict__ dot_partial) {
    const Model& M = *Mp;
    const int npix = M.npix;
    const long long SN = (long long)flag_nplanes(a.flag) * npix;
    const long long u = (long long)blockIdx.x * BLOCK + threadIdx.x;
    double dotv = 0.0;
    if (u < SN) {
        const int p = (int)(u / npix), i = (int)(u - (long long)p * npix), k = flag_map(a.flag, p);
        double acc[NG], xv[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) { acc[g] = 0.0; xv[g] = x[(long long)g * SN + u]; }
        if (!is_masked(M.mask[i])) {
            Prep pr[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                double t0, t1;
                load_theta(M, M.comp[a.gc[g]], i, k, t0, t1);
                pr[g] = sed_prep(M.comp[a.gc[g]], t0, t1);
            }
            for (int j = 0; j < M.nbands; ++j) {
                double mrow[NG], temp1 = 0.0;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    mrow[g] = sed_eval(M, M.comp[a.gc[g]], j, pr[g]);
                    temp1 = temp1 + xv[g] * mrow[g];
                }
                const double rms = M.rms[((long long)j * M.nmaps + (k - 1)) * npix + i];
                temp1 = temp1 / (rms * rms);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = acc[g] + temp1 * mrow[g];
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            res[(long long)g * SN + u] = acc[g];
            dotv += xv[g] * acc[g];
        }
    }
    if (dot_partial) {
        __shared__ double sh[BLOCK / 64];
        for (int o = 32; o > 0; o >>= 1) dotv += __shfl_down(dotv, o, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dotv;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int w = 0; w < BLOCK / 64; ++w) s += sh[w];
            dot_partial[blockIdx.x] = s;
        }
    }
}

// compute_sample_vector, src/dang_cg_mod.f90:913-1100 (diffuse branch, with its quirks)
template <int NG>
__global__ __launch_bounds__(BLOCK) void k_sample_vector(const Model* __restrict__ Mp, GroupArgs a,
                                                         const double* __restrict__ eta, double* __restrict__ res) {
    const Model& M = *Mp;
    const int npix = M.npix;
    const long long SN = (long long)flag_nplanes(a.flag) * npix;
    const long long u = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (u >= SN) return;
    const int p = (int)(u / npix), i = (int)(u - (long long)p * npix), k = flag_map(a.flag, p);
    double acc = 0.0;
    if (!is_masked(M.mask[i])) {
        const Comp& c = M.comp[a.gc[NG - 1]];
        double t0, t1;
        load_theta(M, c, i, k, t0, t1);
        const Prep pr = sed_prep(c, t0, t1);
        const double e = eta[u];
        for (int j = 0; j < M.nbands; ++j) {
            const double temp1 = e / M.rms[((long long)j * M.nmaps + (k - 1)) * npix + i];
            acc = acc + temp1 * sed_eval(M, c, j, pr);
        }
    }
    res[u] = acc;
#pragma unroll
    for (int g = 1; g < NG; ++g) res[(long long)g * SN + u] = 0.0;
}

// eta(i) = rand_normal(0,1), src/dang_cg_mod.f90:256-262, from the keyed stream
__global__ __launch_bounds__(BLOCK) void k_draw_eta(const Model* __restrict__ Mp, GroupArgs a, double* __restrict__ eta) {
    const Model& M = *Mp;
    const long long SN = (long long)flag_nplanes(a.flag) * M.npix;
    const long long u = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (u >= SN) return;
    const int p = (int)(u / M.npix), i = (int)(u - (long long)p * M.npix), k = flag_map(a.flag, p);
    double u1, u2;
    uniform2(a.seed, a.stream, (unsigned long long)(M.pix0 + i), (uint32_t)k, u1, u2);
    eta[u] = rand_normal(0.0, 1.0, u1, u2);
}

// pack / unpack between c%amplitude and x (initialize_x :1173-1282, unpack_amplitudes :1284-1396)
__global__ __launch_bounds__(BLOCK) void k_pack(const Model* __restrict__ Mp, GroupArgs a, double* __restrict__ x, int unpack) {
    const Model& M = *Mp;
    const long long SN = (long long)flag_nplanes(a.flag) * M.npix;
    const long long u = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (u >= SN) return;
    const int p = (int)(u / M.npix), i = (int)(u - (long long)p * M.npix), k = flag_map(a.flag, p);
    for (int g = 0; g < a.ng; ++g) {
        double* amp = M.comp[a.gc[g]].amp + (long long)(k - 1) * M.npix + i;
        if (unpack) *amp = x[(long long)g * SN + u];
        else x[(long long)g * SN + u] = *amp;
    }
}

// CG vector updates (src/dang_cg_mod.f90:283-305) with block partials of sum(r*r)
//  mode 0: r = b2 - q ; d = r                        -> partial sum(r*r)
//  mode 1: x += alpha*d ; r -= alpha*q               -> partial sum(r*r)
//  mode 2: d = r + beta*d
//  mode 3: b2 = b + f
__global__ __launch_bounds__(BLOCK) void k_cg_vec(int mode, long long n, double alpha, double* __restrict__ x,
                                                  double* __restrict__ r, double* __restrict__ d,
                                                  const double* __restrict__ q, const double* __restrict__ b2,
                                                  double* __restrict__ partial) {
    const long long t = (long long)blockIdx.x * BLOCK + threadIdx.x;
    double rr = 0.0;
    if (t < n) {
        if (mode == 0) {
            const double rv = b2[t] - q[t];
            r[t] = rv; d[t] = rv; rr = rv * rv;
        } else if (mode == 1) {
            x[t] = x[t] + alpha * d[t];
            const double rv = r[t] - alpha * q[t];
            r[t] = rv; rr = rv * rv;
        } else if (mode == 2) {
            d[t] = r[t] + alpha * d[t];
        } else {
            x[t] = b2[t] + q[t];
        }
    }
    if (partial) {
        __shared__ double sh[BLOCK / 64];
        for (int o = 32; o > 0; o >>= 1) rr += __shfl_down(rr, o, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = rr;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int w = 0; w < BLOCK / 64; ++w) s += sh[w];
            partial[blockIdx.x] = s;
        }
    }
}

// deterministic second stage: out[0] = sum(partial[0..n)) in a fixed order
__global__ __launch_bounds__(BLOCK) void k_reduce(const double* __restrict__ partial, long long n, double* __restrict__ out) {
    __shared__ double sh[BLOCK];
    double s = 0.0;
    for (long long t = threadIdx.x; t < n; t += BLOCK) s += partial[t];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = BLOCK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// ---------------------------------------------------------------------------
// Index phase: sample_index_mh, per-pixel branch (src/dang_sample_mod.f90:332-481) with
// update_sample_model (:548-553), evaluate_lnL / evaluate_marginal_lnL
// (src/dang_lnl_mod.f90:126-182, 47-124) and the priors (:394-400) fused.
// One thread per pixel.  The pixel's cleaned data d(k,j), 1/rms(k,j) and the chain-invariant
// SED factor F(j) are staged once into LDS columns [slot][thread] (conflict-free: lane l
// touches bank pair 2l), the chain state lives in registers, the index map is written once.
//
// Chain modes: the SED of the sampled component factorises into a part that is constant
// along the chain (F_j, evaluated once) and a part that depends on the proposal, with the
// reference's multiplication order kept, e.g. mbb (:947-948) = (A/B_j) * P_j:
//   CH_POW       power-law beta     : exp(beta*lnr_j)
//   CH_MBB_BETA  mbb beta (T fixed) : F_j = A/B_j ; sed = F_j * exp((beta+1)*lnr_j)
//   CH_MBB_T     mbb T (beta fixed) : F_j = P_j   ; sed = (A(T)/B_j(T)) * F_j
//   CH_LOGN_NUP  lognormal nu_p     : sed = exp(-0.5*(log(nu_j/(nu_p*1e9))/w)^2) * cst_j
//   CH_LOGN_W    lognormal w        : F_j = log(nu_j/(nu_p*1e9)) ; sed = exp(-0.5*(F_j/w)^2) * cst_j
//   CH_GENERIC   anything else (free-free T_e, bandpass-integrated bands): sed_prep + sed_eval
enum { CH_GENERIC = 0, CH_POW = 1, CH_MBB_BETA = 2, CH_MBB_T = 3, CH_LOGN_NUP = 4, CH_LOGN_W = 5 };

struct IndexArgs {
    int comp, nind, s1, s2, nsample, ml_mode, mode;
    unsigned others;  // bit l: component l (/= comp) may have a non-zero amplitude on planes s1..s2
    unsigned long long seed, stream;
};

struct ChainCtx {
    const Model& M;
    const Comp& c;
    const IndexArgs& a;
    double* lds;        // per-thread columns
    const double* tab;  // block-shared constant table (sed_table_build)
    int BS, tid, nb, Sp;
    double amp0, amp1, other;  // amplitudes on the planes; the index that is NOT sampled
    __device__ __forceinline__ double& D(int kk, int j) const { return lds[(kk * nb + j) * BS + tid]; }
    __device__ __forceinline__ double& IS(int kk, int j) const { return lds[((Sp + kk) * nb + j) * BS + tid]; }
    __device__ __forceinline__ double& F(int j) const { return lds[(2 * Sp * nb + j) * BS + tid]; }
};

// -1/2 sum ((d-m)/rms)^2 per plane (evaluate_lnL) or the marginal form; acc0/acc1 = per-plane parts
__device__ __forceinline__ double chain_lnl(const ChainCtx& C, double th, int lnl_type, double& acc0, double& acc1) {
    const Model& M = C.M;
    const Comp& c = C.c;
    const bool first = (C.a.nind == 0);
    acc0 = 0.0; acc1 = 0.0;
    if (lnl_type == DANGX_LNL_PRIOR) return 0.0;
    double s0 = 0.0, s1 = 0.0;
    Prep pr = {0.0, 0.0, 0.0};
    switch (C.a.mode) {
    case CH_POW: s0 = th; break;
    case CH_MBB_BETA: s0 = th + 1.0; break;
    case CH_MBB_T: s0 = H_PLANCK / (K_B * th); s1 = exp(s0 * c.nu_ref) - 1.0; break;
    case CH_LOGN_NUP: s0 = th * 1e9; s1 = C.other; break;
    case CH_LOGN_W: s1 = th; break;
    default: pr = sed_prep(c, first ? th : C.other, first ? C.other : th); break;
    }
    double lnL = 0.0;
    for (int j = 0; j < C.nb; ++j) {
        double s;
        switch (C.a.mode) {
        case CH_POW: s = exp(s0 * c.lnr[j]); break;
        case CH_MBB_BETA: s = C.F(j) * exp(s0 * c.lnr[j]); break;
        case CH_MBB_T: s = s1 / (exp(s0 * M.band[j].nu_c) - 1.0) * C.F(j); break;
        case CH_LOGN_NUP: { const double l = log_pos(M.band[j].nu_c / s0) / s1; s = exp(-0.5 * (l * l)) * c.cst[j]; break; }
        case CH_LOGN_W: { const double l = C.F(j) / s1; s = exp(-0.5 * (l * l)) * c.cst[j]; break; }
        default: s = sed_eval(M, c, j, pr); break;
        }
        if (lnl_type == DANGX_LNL_CHISQ) {
            const double t = (C.D(0, j) - C.amp0 * s) * C.IS(0, j);
            acc0 = acc0 - 0.5 * (t * t);
            if (C.Sp == 2) {
                const double t2 = (C.D(1, j) - C.amp1 * s) * C.IS(1, j);
                acc1 = acc1 - 0.5 * (t2 * t2);
            }
        } else {  // marginal: -0.5*TNd*invTNT*TNd per (band, plane), src/dang_lnl_mod.f90:113-122
            for (int kk = 0; kk < C.Sp; ++kk) {
                const double m = (kk ? C.amp1 : C.amp0) * s;
                const double is = C.IS(kk, j);
                const double TN = m * (is * is);
                const double TNd = TN * C.D(kk, j);
                const double TNT = TN * m;
                lnL = lnL - 0.5 * TNd * (1.0 / TNT) * TNd;
            }
        }
    }
    return (lnl_type == DANGX_LNL_CHISQ) ? acc0 + acc1 : lnL;
}

// Fast path of evaluate_lnL for the chain: chisq likelihood, delta bandpasses, compile-time chain mode
// (CH_POW / CH_MBB_BETA / CH_MBB_T), plane count SP and band tile TB (nb % TB == 0).  A tile first issues
// every LDS / scalar load of its TB bands, then runs the TB independent exp chains interleaved, then
// accumulates in band order (same summation order as the plain loop).
template <int MODE, int SP, int TB>
__device__ __forceinline__ double chain_lnl_tiled(const ChainCtx& C, double th, double& acc0, double& acc1) {
    const Model& M = C.M;
    const Comp& c = C.c;
    double s0 = 0.0, s1 = 0.0;
    if (MODE == CH_POW) s0 = th;
    else if (MODE == CH_MBB_BETA) s0 = th + 1.0;
    else { s0 = H_PLANCK / (K_B * th); s1 = exp(s0 * c.nu_ref) - 1.0; }
    acc0 = 0.0; acc1 = 0.0;
    for (int j0 = 0; j0 < C.nb; j0 += TB) {
        double f[TB], d0[TB], i0[TB], d1[TB], i1[TB], x[TB], s[TB];
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const int j = j0 + t;
            x[t] = (MODE == CH_MBB_T) ? s0 * C.tab[(TROWS * M.ncomp) * C.nb + j] : s0 * C.tab[(TROWS * C.a.comp) * C.nb + j];
            f[t] = (MODE == CH_POW) ? 1.0 : C.F(j);
            d0[t] = C.D(0, j); i0[t] = C.IS(0, j);
            if (SP == 2) { d1[t] = C.D(1, j); i1[t] = C.IS(1, j); }
        }
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const double e = exp(x[t]);
            if (MODE == CH_POW) s[t] = e;
            else if (MODE == CH_MBB_BETA) s[t] = f[t] * e;
            else s[t] = s1 / (e - 1.0) * f[t];
        }
#pragma unroll
        for (int t = 0; t < TB; ++t) {
            const double r0 = (d0[t] - C.amp0 * s[t]) * i0[t];
            acc0 = acc0 - 0.5 * (r0 * r0);
            if (SP == 2) {
                const double r1 = (d1[t] - C.amp1 * s[t]) * i1[t];
                acc1 = acc1 - 0.5 * (r1 * r1);
            }
        }
    }
    return acc0 + acc1;
}

template <bool FAST>
__device__ __forceinline__ double index_prior(const ChainCtx& C, double val) {
    const Comp& c = C.c;
    const int q = C.a.nind;
    const int t = c.prior_type[q];
    if (t == DANGX_PRIOR_GAUSSIAN) {
        // log(eval_normal_prior) (src/dang_util_mod.f90:112-121, src/dang_sample_mod.f90:395):
        // log(exp(-(x-m)^2/(2 var))/(std*sqrt(2 pi))) = -(x-m)^2/(2 var) - log(std*sqrt(2 pi));
        // the reference's exp() underflows to 0 (log -> -inf) beyond ~745
        const double mean = c.gauss[q][0], std = c.gauss[q][1];
        const double arg = ((val - mean) * (val - mean)) / (2 * (std * std));
        return (arg > 745.0) ? -INFINITY : -arg - c.lgden[q];
    }
    if (t == DANGX_PRIOR_JEFFREYS) {  // eval_jeffreys_prior, src/dang_lnl_mod.f90:242-304
        double sum = 0.0;
        if (c.is_synch) {
            const Prep pr = sed_prep(c, val, 0.0);
            for (int kk = 0; kk < C.Sp; ++kk)
                for (int j = 0; j < C.nb; ++j) {
                    const double amp = kk ? C.amp1 : C.amp0;
                    const double ss = amp * (FAST ? sed_eval_tab(c.type, C.tab, C.nb, C.M.ncomp, C.a.comp, j, pr)
                                                   : sed_eval(C.M, c, j, pr));
                    const double rr = C.IS(kk, j);  // 1/rms
                    const double tt = (rr * rr) * (ss / amp) * c.lnr[j];
                    sum = sum + tt * tt;
                }
        }
        return log(sqrt(sum));
    }
    return 0.0;
}

// the chain of one pixel; returns the number of accepted proposals; chi[0..3] = chi^2 of the touched
// planes before (plane0, plane1) and after (plane0, plane1) the sweep
// MODE == CH_GENERIC: everything decided at run time (a.mode, lnl type, plane count, any nb);
// otherwise the chisq fast path above with compile-time MODE / SP / TB.
template <int MODE, int SP, int TB>
__device__ __forceinline__ unsigned long long index_chain(const Model& M, const IndexArgs& a, double* lds, const double* tab,
                                                          int BS, int tid, int i, double chi[4]) {
    const int npix = M.npix, nb = M.nbands;
    const Comp& c = M.comp[a.comp];
    const int Sp = a.s2 - a.s1 + 1;
    double* out = c.idx + ((long long)a.nind * M.nmaps) * npix + i;
    if (is_masked(M.mask[i])) {  // :362 cycle; index_map stays 0 (:223) and is copied back (:480-483)
        for (int k = a.s1; k <= a.s2; ++k) out[(long long)(k - 1) * npix] = 0.0;
        return 0ull;
    }
    // chain state: sample(l) = c%indices(i, map_inds(1), l)  (:372-377)
    double sample0, sample1;
    load_theta(M, c, i, a.s1, sample0, sample1);
    const bool first = (a.nind == 0);
    ChainCtx C{M, c, a, lds, tab, BS, tid, nb, Sp, 0.0, 0.0, first ? sample1 : sample0};
    // --- stage data_raw (:173-177) and 1/rms: loads of ST bands are issued together
    constexpr int ST = (MODE == CH_GENERIC) ? 4 : TB;
    for (int kk = 0; kk < Sp; ++kk) {
        const int k = a.s1 + kk;
        const double ak = c.amp[(long long)(k - 1) * npix + i];
        if (kk) C.amp1 = ak; else C.amp0 = ak;
        const long long bstride = (long long)M.nmaps * npix;
        const double* sigp = M.sig + (long long)(k - 1) * npix + i;
        const double* rmsp = M.rms + (long long)(k - 1) * npix + i;
#pragma unroll 2
        for (int j0 = 0; j0 < nb; j0 += ST) {
            double dv[ST], rv[ST];
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                const int j = (j0 + t < nb) ? j0 + t : nb - 1;
                dv[t] = sigp[j * bstride];
                rv[t] = rmsp[j * bstride];
            }
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                const int j = j0 + t;
                if (j < nb) {
                    C.D(kk, j) = (k == 1) ? (dv[t] - tab[(TROWS * M.ncomp + 2) * nb + j]) / tab[(TROWS * M.ncomp + 1) * nb + j] : dv[t];
                    C.IS(kk, j) = 1.0 / rv[t];
                }
            }
        }
    }
    // --- remove every OTHER component (:180-196), in component_list order; a.others holds the
    // components whose amplitude may be non-zero on these planes (an all-zero plane contributes 0*sed).
    // The next component's amplitude / indices are fetched while the current one is processed.
    {
        unsigned om = a.others;
        double na[2] = {0.0, 0.0}, nt0[2] = {0.0, 0.0}, nt1[2] = {0.0, 0.0};
        auto fetch = [&](int l) {
            const Comp& c2 = M.comp[l];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                if (kk < Sp) {
                    na[kk] = c2.amp[(long long)(a.s1 + kk - 1) * npix + i];
                    if (MODE == CH_GENERIC || !((c2.const_planes >> (a.s1 + kk - 1)) & 1))
                        load_theta(M, c2, i, a.s1 + kk, nt0[kk], nt1[kk]);
                }
        };
        int l = om ? __builtin_ctz(om) : -1;
        if (l >= 0) fetch(l);
        while (l >= 0) {
            const Comp& c2 = M.comp[l];
            const double ca[2] = {na[0], na[1]}, ct0[2] = {nt0[0], nt0[1]}, ct1[2] = {nt1[0], nt1[1]};
            om &= om - 1;
            const int ln = om ? __builtin_ctz(om) : -1;
            if (ln >= 0) fetch(ln);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                if (kk < Sp) {
                    if (MODE != CH_GENERIC && ((c2.const_planes >> (a.s1 + kk - 1)) & 1)) {
                        for (int j = 0; j < nb; ++j) C.D(kk, j) -= ca[kk] * sed_const_tab(tab, nb, l, a.s1 + kk, j);
                    } else {
                        const Prep pr = sed_prep(c2, ct0[kk], ct1[kk]);
                        const int ty2 = c2.type;
#pragma unroll 1
                        for (int j = 0; j < nb; ++j)
                            C.D(kk, j) -= (MODE != CH_GENERIC) ? ca[kk] * sed_eval_tab(ty2, tab, nb, M.ncomp, l, j, pr)
                                                               : signal_of(c2, ca[kk], sed_eval(M, c2, j, pr));
                    }
                }
            l = ln;
        }
    }
    // --- chain-invariant SED factor
    if (a.mode == CH_MBB_BETA) {
        const double z = H_PLANCK / (K_B * sample1);
        const double A = exp(z * c.nu_ref) - 1.0;
        for (int j = 0; j < nb; ++j) C.F(j) = A / (exp(z * tab[(TROWS * M.ncomp) * nb + j]) - 1.0);
    } else if (a.mode == CH_MBB_T) {
        for (int j = 0; j < nb; ++j) C.F(j) = exp((sample0 + 1.0) * tab[(TROWS * a.comp) * nb + j]);
    } else if (a.mode == CH_LOGN_W) {
        for (int j = 0; j < nb; ++j) C.F(j) = log_pos(M.band[j].nu_c / (sample0 * 1e9));
    }
    const int lnl_type = (MODE == CH_GENERIC) ? c.lnl_type[a.nind] : DANGX_LNL_CHISQ;
    const unsigned long long gpix = (unsigned long long)(M.pix0 + i);
    unsigned long long nacc = 0;
    double cur = first ? sample0 : sample1;  // sample(nind)
    double a0, a1, c0, c1;                   // per-plane likelihood parts: current / proposal
    auto lnl_of = [&](double th, double& p0, double& p1) -> double {
        if (MODE == CH_GENERIC) return chain_lnl(C, th, lnl_type, p0, p1);
        return chain_lnl_tiled<MODE == CH_GENERIC ? CH_POW : MODE, SP, TB>(C, th, p0, p1);
    };
    double lnl = lnl_of(cur, a0, a1);
    if (lnl_type != DANGX_LNL_CHISQ) {       // chi^2 bookkeeping needs the chisq form
        double t0, t1;
        chain_lnl(C, cur, DANGX_LNL_CHISQ, t0, t1);
        chi[0] = -2.0 * t0; chi[1] = -2.0 * t1;
    } else {
        chi[0] = -2.0 * a0; chi[1] = -2.0 * a1;
    }
    bool sample_it = true;
    if (lnl_type == DANGX_LNL_PRIOR) {  // :389-392
        double u1, u2;
        sample_it = false;
        uniform2(a.seed, a.stream, gpix, 0u, u1, u2);
        cur = rand_normal(c.gauss[a.nind][0], c.gauss[a.nind][1], u1, u2);
    }
    double lnl_old = lnl + index_prior<MODE != CH_GENERIC>(C, cur);
    if (sample_it) {
        const double step = c.step[a.nind];
        const double lo = c.uni[a.nind][0], hi = c.uni[a.nind][1];
        for (int l = 1; l <= a.nsample; ++l) {
            double u1, u2, u3;
            uniform3(a.seed, a.stream, gpix, (uint32_t)l, u1, u2, u3);  // one Philox call per step
            const double prop = cur + rand_normal(0.0, step, u1, u2);  // :414
            if (prop < lo || prop > hi) continue;  // :415 (the accept draw is not used)
            lnl = lnl_of(prop, c0, c1);
            const double lnl_new = lnl + index_prior<MODE != CH_GENERIC>(C, prop);
            const double diff = lnl_new - lnl_old;
            bool acc;
            if (a.ml_mode == DANGX_ML_OPTIMIZE) {
                acc = diff > 0.0;  // :443-447
            } else {
                // :448-454  diff > log(u)  <=>  diff >= 0 or exp(diff) > u   (u in (0,1))
                acc = (diff >= 0.0) || (exp(diff) > u3);
            }
            if (acc) {
                cur = prop;
                lnl_old = lnl_new;
                a0 = c0; a1 = c1;
                ++nacc;
            }
        }
    }
    for (int k = a.s1; k <= a.s2; ++k) out[(long long)(k - 1) * npix] = cur;  // :465, :483
    if (lnl_type != DANGX_LNL_CHISQ) chain_lnl(C, cur, DANGX_LNL_CHISQ, a0, a1);
    chi[2] = -2.0 * a0; chi[3] = -2.0 * a1;
    return nacc;
}

// ---------------------------------------------------------------------------
// Register-resident form of the same chain (chisq likelihood, delta bandpasses, CH_POW / CH_MBB_BETA /
// CH_MBB_T) for compile-time band count NB and plane count SP: the cleaned data, 1/rms and the chain-
// invariant SED factor live in VGPRs (statically indexed, fully unrolled), per-band constants in SGPRs,
// and the kernel uses no LDS and no barrier.  The CU's vector register file (512 KB) is three times its
// LDS, so this form runs at 2-3 waves/SIMD where the LDS-column form is capped at 1-2.
// Arithmetic and operation order are identical to index_chain<MODE, SP, TB>.
template <int MODE, int SP, int NB, bool ISLDS>
struct RegChain {
    double D[SP][NB], F[NB];
    double ISr[ISLDS ? 1 : SP][ISLDS ? 1 : NB];  // 1/rms in registers ...
    double* isl;                                  // ... or in LDS columns [slot][thread] when registers run out
    double amp[SP];

    __device__ __forceinline__ double is(int kk, int j) const { return ISLDS ? isl[(kk * NB + j) * BLOCK] : ISr[ISLDS ? 0 : kk][ISLDS ? 0 : j]; }
    __device__ __forceinline__ void set_is(int kk, int j, double v) {
        if (ISLDS) isl[(kk * NB + j) * BLOCK] = v; else ISr[ISLDS ? 0 : kk][ISLDS ? 0 : j] = v;
    }

    __device__ __forceinline__ double lnl(const Model& M, const Comp& c, double th, double other, double& acc0, double& acc1) const {
        double s0 = 0.0, s1 = 0.0;
        if (MODE == CH_POW) s0 = th;
        else if (MODE == CH_MBB_BETA) s0 = th + 1.0;
        else if (MODE == CH_MBB_T) { s0 = H_PLANCK / (K_B * th); s1 = exp(s0 * c.nu_ref) - 1.0; }
        else if (MODE == CH_LOGN_NUP) { s0 = th * 1e9; s1 = other; }
        else s1 = th;  // CH_LOGN_W
        acc0 = 0.0; acc1 = 0.0;
        // bands in tiles of TT: TT independent exp chains interleave, then accumulate in band order
        constexpr int TT = (NB % 5 == 0) ? 5 : (NB % 4 == 0) ? 4 : (NB % 3 == 0) ? 3 : 1;
#pragma unroll
        for (int j0 = 0; j0 < NB; j0 += TT) {
            double s[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const int j = j0 + t;
                if (MODE == CH_LOGN_NUP) {
                    const double l = log_pos(M.band[j].nu_c / s0) / s1;
                    s[t] = exp(-0.5 * (l * l)) * c.cst[j];
                } else if (MODE == CH_LOGN_W) {
                    const double l = F[j] / s1;
                    s[t] = exp(-0.5 * (l * l)) * c.cst[j];
                } else {
                    const double e = exp((MODE == CH_MBB_T) ? s0 * M.band[j].nu_c : s0 * c.lnr[j]);
                    if (MODE == CH_POW) s[t] = e;
                    else if (MODE == CH_MBB_BETA) s[t] = F[j] * e;
                    else s[t] = s1 / (e - 1.0) * F[j];
                }
            }
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const int j = j0 + t;
                const double r0 = (D[0][j] - amp[0] * s[t]) * is(0, j);
                acc0 = acc0 - 0.5 * (r0 * r0);
                if (SP == 2) {
                    const double r1 = (D[SP - 1][j] - amp[SP - 1] * s[t]) * is(SP - 1, j);
                    acc1 = acc1 - 0.5 * (r1 * r1);
                }
            }
        }
        return acc0 + acc1;
    }
};

// eval_sed of an "other" component for all NB bands of one plane, subtracted from D (static band index)
template <int NB>
__device__ __forceinline__ void subtract_other(const Model& M, const Comp& c2, int k, double amp2, double t0, double t1,
                                               double (&Dk)[NB]) {
    if ((c2.const_planes >> (k - 1)) & 1) {  // spatially constant indices: host-evaluated SED
#pragma unroll
        for (int j = 0; j < NB; ++j) Dk[j] -= amp2 * c2.csed[k - 1][j];
        return;
    }
    const Prep pr = sed_prep(c2, t0, t1);
    switch (c2.type) {
    case DANGX_POWERLAW:
#pragma unroll
        for (int j = 0; j < NB; ++j) Dk[j] -= amp2 * exp(pr.p0 * c2.lnr[j]);
        break;
    case DANGX_MBB:
#pragma unroll
        for (int j = 0; j < NB; ++j) Dk[j] -= amp2 * (pr.p2 / (exp(pr.p1 * M.band[j].nu_c) - 1.0) * exp(pr.p0 * c2.lnr[j]));
        break;
    case DANGX_FREEFREE:
#pragma unroll
        for (int j = 0; j < NB; ++j) Dk[j] -= amp2 * (ff_gaunt(c2.lnu9[j], pr.p0) / pr.p1 * c2.cst[j]);
        break;
    case DANGX_LOGNORMAL:
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const double l2 = log_pos(M.band[j].nu_c / pr.p0) / pr.p1;
            Dk[j] -= amp2 * (exp(-0.5 * (l2 * l2)) * c2.cst[j]);
        }
        break;
    default:  // cmb
#pragma unroll
        for (int j = 0; j < NB; ++j) Dk[j] -= amp2 * c2.cst[j];
        break;
    }
}

template <int MODE, int SP, int NB, bool ISLDS>
__device__ __forceinline__ unsigned long long index_chain_reg(const Model& M, const IndexArgs& a, int i, double chi[4], double* isl) {
    const int npix = M.npix;
    const Comp& c = M.comp[a.comp];
    double* out = c.idx + ((long long)a.nind * M.nmaps) * npix + i;
    if (is_masked(M.mask[i])) {  // :362 cycle; index_map stays 0 (:223) and is copied back (:480-483)
#pragma unroll
        for (int kk = 0; kk < SP; ++kk) out[(long long)(a.s1 + kk - 1) * npix] = 0.0;
        return 0ull;
    }
    RegChain<MODE, SP, NB, ISLDS> R;
    R.isl = isl;
    double sample0, sample1;
    load_theta(M, c, i, a.s1, sample0, sample1);  // sample(l) = c%indices(i, map_inds(1), l), :372-377
    const bool first = (a.nind == 0);
    // --- stage data_raw (:173-177) and rms: every load issued before the first use
    const long long bstride = (long long)M.nmaps * npix;
#pragma unroll
    for (int kk = 0; kk < SP; ++kk) {
        const int k = a.s1 + kk;
        R.amp[kk] = c.amp[(long long)(k - 1) * npix + i];
        const double* sigp = M.sig + (long long)(k - 1) * npix + i;
        const double* rmsp = M.rms + (long long)(k - 1) * npix + i;
        double rv[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            R.D[kk][j] = sigp[j * bstride];
            rv[j] = rmsp[j * bstride];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (k == 1) R.D[kk][j] = (R.D[kk][j] - M.offset[j]) / M.gain[j];
            R.set_is(kk, j, 1.0 / rv[j]);
        }
    }
    // --- remove every OTHER component (:180-196) in component_list order, next one prefetched
    {
        unsigned om = a.others;
        double na[SP], nt0[SP], nt1[SP];
        auto fetch = [&](int l) {
            const Comp& c2 = M.comp[l];
#pragma unroll
            for (int kk = 0; kk < SP; ++kk) {
                na[kk] = c2.amp[(long long)(a.s1 + kk - 1) * npix + i];
                nt0[kk] = nt1[kk] = 0.0;
                if (!((c2.const_planes >> (a.s1 + kk - 1)) & 1)) load_theta(M, c2, i, a.s1 + kk, nt0[kk], nt1[kk]);
            }
        };
        int l = om ? __builtin_ctz(om) : -1;
        if (l >= 0) fetch(l);
        while (l >= 0) {
            const Comp& c2 = M.comp[l];
            double ca[SP], ct0[SP], ct1[SP];
#pragma unroll
            for (int kk = 0; kk < SP; ++kk) { ca[kk] = na[kk]; ct0[kk] = nt0[kk]; ct1[kk] = nt1[kk]; }
            om &= om - 1;
            const int ln = om ? __builtin_ctz(om) : -1;
            if (ln >= 0) fetch(ln);
#pragma unroll
            for (int kk = 0; kk < SP; ++kk) subtract_other<NB>(M, c2, a.s1 + kk, ca[kk], ct0[kk], ct1[kk], R.D[kk]);
            l = ln;
        }
    }
    // --- chain-invariant SED factor
    if (MODE == CH_MBB_BETA) {
        const double z = H_PLANCK / (K_B * sample1);
        const double A = exp(z * c.nu_ref) - 1.0;
#pragma unroll
        for (int j = 0; j < NB; ++j) R.F[j] = A / (exp(z * M.band[j].nu_c) - 1.0);
    } else if (MODE == CH_MBB_T) {
#pragma unroll
        for (int j = 0; j < NB; ++j) R.F[j] = exp((sample0 + 1.0) * c.lnr[j]);
    } else if (MODE == CH_LOGN_W) {
#pragma unroll
        for (int j = 0; j < NB; ++j) R.F[j] = log_pos(M.band[j].nu_c / (sample0 * 1e9));
    }
    const double other = first ? sample1 : sample0;  // the index that is not sampled
    // --- chain (gaussian / uniform prior inline; jeffreys falls back to the LDS form on the host side)
    const unsigned long long gpix = (unsigned long long)(M.pix0 + i);
    const int q = a.nind;
    const bool gauss = c.prior_type[q] == DANGX_PRIOR_GAUSSIAN;
    const double pmean = c.gauss[q][0], pstd = c.gauss[q][1], lgden = c.lgden[q];
    auto prior = [&](double v) -> double {
        if (!gauss) return 0.0;
        const double arg = ((v - pmean) * (v - pmean)) / (2 * (pstd * pstd));
        return (arg > 745.0) ? -INFINITY : -arg - lgden;
    };
    unsigned long long nacc = 0;
    double cur = first ? sample0 : sample1;
    double a0, a1, c0, c1;
    double lnl = R.lnl(M, c, cur, other, a0, a1);
    chi[0] = -2.0 * a0; chi[1] = -2.0 * a1;
    double lnl_old = lnl + prior(cur);
    const double step = c.step[q], lo = c.uni[q][0], hi = c.uni[q][1];
    for (int l = 1; l <= a.nsample; ++l) {
        double u1, u2, u3;
        uniform3(a.seed, a.stream, gpix, (uint32_t)l, u1, u2, u3);
        const double prop = cur + rand_normal(0.0, step, u1, u2);  // :414
        if (prop < lo || prop > hi) continue;                      // :415
        lnl = R.lnl(M, c, prop, other, c0, c1);
        const double lnl_new = lnl + prior(prop);
        const double diff = lnl_new - lnl_old;
        const bool acc = (a.ml_mode == DANGX_ML_OPTIMIZE) ? (diff > 0.0) : ((diff >= 0.0) || (exp(diff) > u3));  // :443-454
        if (acc) { cur = prop; lnl_old = lnl_new; a0 = c0; a1 = c1; ++nacc; }
    }
#pragma unroll
    for (int kk = 0; kk < SP; ++kk) out[(long long)(a.s1 + kk - 1) * npix] = cur;  // :465, :483
    chi[2] = -2.0 * a0; chi[3] = -2.0 * a1;
    return nacc;
}

template <int MODE, int SP, int NB, bool ISLDS>
__global__ __launch_bounds__(BLOCK, (SP == 1 && NB <= 10) ? 3 : 2) void k_index_mh_reg(const Model* __restrict__ Mp, IndexArgs a,
                                                        unsigned long long* __restrict__ accepted,
                                                        double* __restrict__ chi_partial) {
    extern __shared__ double lds[];  // ISLDS: 1/rms columns [SP*NB][BLOCK]
    const Model& M = *Mp;
    const int tid = threadIdx.x;
    const int i = blockIdx.x * BLOCK + tid;
    double chi[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned long long nacc = (i < M.npix) ? index_chain_reg<MODE, SP, NB, ISLDS>(M, a, i, chi, lds + tid) : 0ull;
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

// chi_partial (nullable): [4][gridDim.x] block sums of chi[0..3]
template <int MODE, int SP, int TB>
__global__ __launch_bounds__(BLOCK) void k_index_mh(const Model* __restrict__ Mp, IndexArgs a,
                                                    unsigned long long* __restrict__ accepted,
                                                    double* __restrict__ chi_partial) {
    extern __shared__ double lds[];  // [constant table | per-thread columns]
    const Model& M = *Mp;
    const int BS = blockDim.x, tid = threadIdx.x;
    const int i = blockIdx.x * BS + tid;
    sed_table_build(M, lds, tid, BS);
    __syncthreads();
    double chi[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned long long nacc = (i < M.npix) ? index_chain<MODE, SP, TB>(M, a, lds + sed_table_size(M), lds, BS, tid, i, chi) : 0ull;
    if (accepted) {  // every lane takes part in the wave reduction
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
            for (int w = 0; w < BS / 64; ++w) s += sh[tid][w];
            chi_partial[(long long)tid * gridDim.x + blockIdx.x] = s;
        }
    }
}

// first stage of a deterministic row-wise reduction: in[q][0..n) -> out[q][0..gridDim.x), fixed chunking
__global__ __launch_bounds__(BLOCK) void k_reduce_rows(const double* __restrict__ in, long long n, int rows,
                                                       double* __restrict__ out) {
    __shared__ double sh[BLOCK];
    const long long chunk = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    for (int q = 0; q < rows; ++q) {
        double s = 0.0;
        for (long long t = lo + threadIdx.x; t < hi; t += BLOCK) s += in[(long long)q * n + t];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int o = BLOCK / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[(long long)q * gridDim.x + blockIdx.x] = sh[0];
        __syncthreads();
    }
}

// second stage for the fused chi^2: cache[0..2] = chi^2 "before" of planes 1..3, cache[3..5] = "after".
// rows of `partial`: before(plane s1), before(plane s2), after(s1), after(s2); write_before = first sweep
// on these planes since the last amplitude update.
__global__ __launch_bounds__(BLOCK) void k_reduce_chi(const double* __restrict__ partial, long long n, int s1, int s2,
                                                      int write_before, double* __restrict__ cache) {
    __shared__ double sh[BLOCK];
    for (int q = 0; q < 4; ++q) {
        double s = 0.0;
        for (long long t = threadIdx.x; t < n; t += BLOCK) s += partial[(long long)q * n + t];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int o = BLOCK / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const int plane = (q & 1) ? s2 : s1;
            const bool after = q >= 2;
            if (!((q & 1) && s1 == s2) && (after || write_before)) cache[(after ? 3 : 0) + plane - 1] = sh[0];
        }
        __syncthreads();
    }
}

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
__global__ __launch_bounds__(BLOCK) void k_sky_chisq(const Model* __restrict__ Mp, int pol_lo, int pol_hi, double* __restrict__ sky,
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
                    if (amp == 0.0 && !want_maps && c.type != DANGX_TCMB) continue;
                    double t0, t1;
                    load_theta(M, c, i, k, t0, t1);
                    const Prep pr = sed_prep(c, t0, t1);
                    for (int j = 0; j < nb; ++j) lds[j * BS + tid] = lds[j * BS + tid] + signal_of(c, amp, sed_eval(M, c, j, pr));
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

// ---------------------------------------------------------------------------
// Full-sky index mode (index_mode == 1, src/dang_sample_mod.f90:229-329), the tuner (:623-717) and the
// band-gain fit (:570-621).  With one spectral index for the whole sky the model's SED is pixel
// independent, so each Metropolis step is ONE memory-bound pass that produces a few global sums; the
// chain itself (proposal, prior, accept) runs on the host between the all-reduces (dang_amd/api.py).

// data_raw minus every other component for planes s1..s2 (:173-196, all pixels) -> out[(kk*nb + j)*npix + i]
__global__ __launch_bounds__(BLOCK) void k_fullsky_prepare(const Model* __restrict__ Mp, int comp, int s1, int s2,
                                                           unsigned others, double* __restrict__ out) {
    const Model& M = *Mp;
    const int npix = M.npix, nb = M.nbands;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= npix) return;
    for (int k = s1; k <= s2; ++k)
        for (int j = 0; j < nb; ++j) {
            double d = M.sig[((long long)j * M.nmaps + (k - 1)) * npix + i];
            if (k == 1) d = (d - M.offset[j]) / M.gain[j];
            for (unsigned om = others; om; om &= om - 1) {
                const Comp& c2 = M.comp[__builtin_ctz(om)];
                double t0, t1;
                load_theta(M, c2, i, k, t0, t1);
                d = d - signal_of(c2, c2.amp[(long long)(k - 1) * npix + i], sed_eval(M, c2, j, sed_prep(c2, t0, t1)));
            }
            out[((long long)(k - s1) * nb + j) * npix + i] = d;
        }
}

// row sums for one evaluation at theta: what = 0: evaluate_lnL (1 row: -1/2 sum ((d-m)/rms)^2, unmasked);
// what = 1: evaluate_marginal_lnL (2*nb*Sp rows: TNd(j,k), TNT(j,k), all pixels); what = 2: jeffreys (1 row).
// partial[row][gridDim.x]
__global__ __launch_bounds__(BLOCK) void k_fullsky_rows(const Model* __restrict__ Mp, int comp, int s1, int s2, int what,
                                                        double th0, double th1, const double* __restrict__ data,
                                                        double* __restrict__ partial) {
    __shared__ double sh[BLOCK / 64];
    const Model& M = *Mp;
    const Comp& c = M.comp[comp];
    const int npix = M.npix, nb = M.nbands, Sp = s2 - s1 + 1;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const bool in = i < npix;
    const bool msk = in ? is_masked(M.mask[i]) : true;
    const Prep pr = sed_prep(c, th0, th1);
    const int nrows = (what == 1) ? 2 * nb * Sp : 1;
    double amp[2] = {0.0, 0.0};
    if (in) for (int kk = 0; kk < Sp; ++kk) amp[kk] = c.amp[(long long)(s1 + kk - 1) * npix + i];
    for (int row = 0; row < nrows; ++row) {
        double v = 0.0;
        if (in) {
            if (what == 0 && !msk) {
                for (int kk = 0; kk < Sp; ++kk)
                    for (int j = 0; j < nb; ++j) {
                        const double m = signal_of(c, amp[kk], sed_eval(M, c, j, pr));
                        const double t = (data[((long long)kk * nb + j) * npix + i] - m) / M.rms[((long long)j * M.nmaps + (s1 + kk - 1)) * npix + i];
                        v = v - 0.5 * (t * t);
                    }
            } else if (what == 1) {
                const int q = row >> 1, j = q / Sp, kk = q - j * Sp;  // (j outer, k inner) as the reference sums
                const double m = signal_of(c, amp[kk], sed_eval(M, c, j, pr));
                const double rms = M.rms[((long long)j * M.nmaps + (s1 + kk - 1)) * npix + i];
                const double TN = m / (rms * rms);
                v = (row & 1) ? TN * m : TN * data[((long long)kk * nb + j) * npix + i];
            } else if (what == 2 && !msk && c.is_synch) {
                for (int kk = 0; kk < Sp; ++kk)
                    for (int j = 0; j < nb; ++j) {
                        const double ss = signal_of(c, amp[kk], sed_eval(M, c, j, pr));
                        const double rr = 1.0 / M.rms[((long long)j * M.nmaps + (s1 + kk - 1)) * npix + i];
                        const double t = (rr * rr) * (ss / amp[kk]) * c.lnr[j];
                        v = v + t * t;
                    }
            }
        }
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < BLOCK / 64; ++w) t += sh[w];
            partial[(long long)row * gridDim.x + blockIdx.x] = t;
        }
        __syncthreads();
    }
}

// fit_band_gain sums (src/dang_sample_mod.f90:590-607): rows 0: sum map2*N_inv*map1, 1: sum map1*N_inv*map1
__global__ __launch_bounds__(BLOCK) void k_gain_rows(const Model* __restrict__ Mp, int band, double* __restrict__ partial) {
    __shared__ double sh[2][BLOCK / 64];
    const Model& M = *Mp;
    const int npix = M.npix;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    double v0 = 0.0, v1 = 0.0;
    if (i < npix && !is_masked(M.mask[i])) {
        double sky = 0.0;  // sky_model(i,1,band), update_sky_model order (:355-373)
        for (int l = 0; l < M.ncomp; ++l) {
            const Comp& c = M.comp[l];
            double t0, t1;
            load_theta(M, c, i, 1, t0, t1);
            sky = sky + signal_of(c, c.amp[i], sed_eval(M, c, band, sed_prep(c, t0, t1)));
        }
        const long long q = ((long long)band * M.nmaps) * npix + i;
        const double res = (M.sig[q] - M.offset[band]) / M.gain[band] - sky;  // res_map(i,1,band), :384
        const double noise = M.rms[q];
        const double N_inv = 1.0 / (noise * noise);
        const double map2 = res + sky;
        v0 = map2 * N_inv * sky;
        v1 = sky * N_inv * sky;
    }
    for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_down(v0, o, 64); v1 += __shfl_down(v1, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = v0; sh[1][threadIdx.x >> 6] = v1; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += sh[threadIdx.x][w];
        partial[(long long)threadIdx.x * gridDim.x + blockIdx.x] = t;
    }
}

// second stage: out[row] = sum(partial[row][0..n))
__global__ __launch_bounds__(BLOCK) void k_reduce_rows_final(const double* __restrict__ partial, long long n, int rows,
                                                             double* __restrict__ out) {
    __shared__ double sh[BLOCK];
    for (int q = 0; q < rows; ++q) {
        double s = 0.0;
        for (long long t = threadIdx.x; t < n; t += BLOCK) s += partial[(long long)q * n + t];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int o = BLOCK / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[q] = sh[0];
        __syncthreads();
    }
}

// c%indices(:, s1:s2, nind) = value (src/dang_sample_mod.f90:329, 483: every pixel, masked ones too)
__global__ __launch_bounds__(BLOCK) void k_fill_index(const Model* __restrict__ Mp, int comp, int nind, int s1, int s2, double value) {
    const Model& M = *Mp;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= M.npix) return;
    for (int k = s1; k <= s2; ++k) M.comp[comp].idx[((long long)nind * M.nmaps + (k - 1)) * M.npix + i] = value;
}

// bit k of flags[0] is set when plane k+1 of a [nmaps][npix] amplitude map holds a non-zero value
__global__ __launch_bounds__(BLOCK) void k_any_nonzero(const double* __restrict__ amp, long long npix, int nmaps,
                                                       unsigned* __restrict__ flags) {
    for (int k = 0; k < nmaps; ++k) {
        bool nz = false;
        for (long long t = (long long)blockIdx.x * BLOCK + threadIdx.x; t < npix; t += (long long)gridDim.x * BLOCK)
            nz = nz || (amp[(long long)k * npix + t] != 0.0);
        if (__ballot(nz) && (threadIdx.x & 63) == 0) atomicOr(flags, 1u << k);
    }
}

// flags bit (q*3 + k) is set when index map q, plane k+1 of an [nind][nmaps][npix] array is NOT spatially
// constant; first[q*3 + k] receives its first element
__global__ __launch_bounds__(BLOCK) void k_not_constant(const double* __restrict__ idx, long long npix, int nmaps, int nind,
                                                        unsigned* __restrict__ flags, double* __restrict__ first) {
    for (int q = 0; q < nind; ++q)
        for (int k = 0; k < nmaps; ++k) {
            const double* m = idx + ((long long)q * nmaps + k) * npix;
            const double m0 = m[0];
            bool diff = false;
            for (long long t = (long long)blockIdx.x * BLOCK + threadIdx.x; t < npix; t += (long long)gridDim.x * BLOCK)
                diff = diff || (m[t] != m0);
            if (__ballot(diff) && (threadIdx.x & 63) == 0) atomicOr(flags, 1u << (q * 3 + k));
            if (blockIdx.x == 0 && threadIdx.x == 0) first[q * 3 + k] = m0;
        }
}

// eval_sed(band, pix, map_n) over the shard (src/dang_component_mod.f90:778-813)
__global__ __launch_bounds__(BLOCK) void k_eval_sed(const Model* __restrict__ Mp, int comp, int band, int map_n,
                                                    double* __restrict__ out) {
    const Model& M = *Mp;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= M.npix) return;
    const Comp& c = M.comp[comp];
    double t0, t1;
    load_theta(M, c, i, map_n, t0, t1);
    out[i] = sed_eval(M, c, band, sed_prep(c, t0, t1));
}

}  // namespace

// ======================================================================= host side

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
    unsigned idx_const[MAXC] = {}; // bit k-1: every index of the component is spatially constant on plane k
    double idx_val[MAXC][3][MAXI] = {};
    std::vector<double> bp_nu0, bp_tau0;
    double *d_bp_nu0 = nullptr, *d_bp_tau0 = nullptr;
    // scratch
    double* partial = nullptr;
    long long partial_cap = 0;
    double* scalars = nullptr;              // device scalars [8]
    double* chi_cache = nullptr;            // device [6]: chi^2 before/after of planes 1..3 (fused in k_index_mh)
    bool chi_before_valid[3] = {}, chi_after_valid[3] = {}, touched_since_amp[3] = {};
    unsigned long long* counters = nullptr; // device counters [4]
    double* work[6] = {};                   // CG vectors
    double* fs_data = nullptr;              // full-sky mode: cleaned data [Sp][nb][npix]
    long long fs_cap = 0;
    int fs_comp = -1, fs_s1 = 0, fs_s2 = 0;
    double* rows_out = nullptr;             // device [2*MAXB*2 + 8] row sums
    long long work_cap = 0;
    // profiling
    bool prof = false;
    struct Ev { hipEvent_t a, b; int kid; };
    std::vector<Ev> events;
    double prof_ms[DANGX_K_COUNT] = {};
    long long prof_n[DANGX_K_COUNT] = {};
};

namespace {

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

int fail(dangx_ctx* ctx, const std::string& msg) {
    ctx->err = msg;
    return 1;
}

// a2t(bp), src/dang_bp_mod.f90:211-243
double host_a2t(const dangx_ctx* ctx, int j) {
    const Band& b = ctx->hm.band[j];
    double sum = 0.0, y;
    if (b.n == 0) {
        if (b.nu_c > 1e7f) y = (H_PLANCK * b.nu_c) / (K_B * ctx->hm.tcmb);
        else y = (H_PLANCK * b.nu_c * 1e9) / (K_B * ctx->hm.tcmb);
        sum = ((std::exp(y) - 1.0) * (std::exp(y) - 1.0)) / ((y * y) * std::exp(y));
    } else {
        for (int i = 0; i < b.n; ++i) {
            const double nu = ctx->bp_nu0[b.off + i], tau = ctx->bp_tau0[b.off + i];
            if (nu == 0.0) continue;
            if (nu > 1e7f) y = (H_PLANCK * nu) / (K_B * ctx->hm.tcmb);
            else y = (H_PLANCK * nu * 1e9) / (K_B * ctx->hm.tcmb);
            sum = sum + tau * ((std::exp(y) - 1.0) * (std::exp(y) - 1.0)) / ((y * y) * std::exp(y));
        }
    }
    return sum;
}

// eval_sed for a delta bandpass on the host (src/dang_component_mod.f90:886-1040), used for components whose
// indices are spatially constant on a plane
double host_sed(const Comp& c, double nu, double cmb_cst, double th0, double th1) {
    switch (c.type) {
    case DANGX_POWERLAW: return std::pow(nu / c.nu_ref, th0);
    case DANGX_MBB: {
        const double z = H_PLANCK / (K_B * th1);
        return (std::exp(z * c.nu_ref) - 1.0) / (std::exp(z * nu) - 1.0) * std::pow(nu / c.nu_ref, th0 + 1.0);
    }
    case DANGX_FREEFREE: {
        auto g = [&](double v) {
            return std::log(std::exp(5.960 - std::sqrt(3.0) / PI * std::log(1.0 * v / 1.0e9 * std::pow(th0 / 1.0e4, -1.5))) + 2.71828);
        };
        const double r = nu / c.nu_ref;
        return g(nu) / g(c.nu_ref) * (1.0 / (r * r));
    }
    case DANGX_LOGNORMAL: {
        const double l = std::log(nu / (th0 * 1e9)) / th1;
        const double q = c.nu_ref / nu;
        return std::exp(-0.5 * (l * l)) * (q * q);
    }
    case DANGX_CMB: return cmb_cst;
    default: return 0.0;
    }
}

int sync_model(dangx_ctx* ctx) {
    if (!ctx->dirty) return 0;
    Model& M = ctx->hm;
    for (int j = 0; j < M.nbands; ++j)
        if (!ctx->band_set[j]) return fail(ctx, "band " + std::to_string(j) + " not set");
    for (int l = 0; l < M.ncomp; ++l)
        if (!ctx->comp_set[l]) return fail(ctx, "component " + std::to_string(l) + " not set");
    if (!ctx->sig || !ctx->rms || !ctx->mask) return fail(ctx, "map data not uploaded");
    M.sig = ctx->sig; M.rms = ctx->rms; M.mask = ctx->mask;
    M.all_delta = 1;
    for (int j = 0; j < M.nbands; ++j) if (M.band[j].n != 0) M.all_delta = 0;
    for (int l = 0; l < M.ncomp; ++l) if (ctx->desc[l].type == DANGX_TCMB) M.all_delta = 0;  // bare-sed signal: generic paths only
    if (!ctx->bp_nu0.empty()) {
        if (ctx->d_bp_nu0) { (void)hipFree(ctx->d_bp_nu0); (void)hipFree(ctx->d_bp_tau0); }
        const size_t nbytes = ctx->bp_nu0.size() * sizeof(double);
        HIPCHK(ctx, hipMalloc(&ctx->d_bp_nu0, nbytes));
        HIPCHK(ctx, hipMalloc(&ctx->d_bp_tau0, nbytes));
        HIPCHK(ctx, hipMemcpy(ctx->d_bp_nu0, ctx->bp_nu0.data(), nbytes, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(ctx->d_bp_tau0, ctx->bp_tau0.data(), nbytes, hipMemcpyHostToDevice));
    }
    M.bp_nu0 = ctx->d_bp_nu0; M.bp_tau0 = ctx->d_bp_tau0;
    for (int l = 0; l < M.ncomp; ++l) {
        Comp& c = M.comp[l];
        const dangx_comp_desc& d = ctx->desc[l];
        c.type = d.type; c.nind = d.nindices; c.group = d.cg_group; c.sample_amp = d.sample_amplitude;
        c.is_synch = d.is_synch; c.nu_ref = d.nu_ref;
        c.amp = ctx->amp[l]; c.idx = ctx->idx[l];
        for (int q = 0; q < MAXI; ++q) {
            c.lnl_type[q] = d.lnl_type[q]; c.prior_type[q] = d.prior_type[q];
            c.gauss[q][0] = d.gauss_prior[q][0]; c.gauss[q][1] = d.gauss_prior[q][1];
            c.uni[q][0] = d.uni_prior[q][0]; c.uni[q][1] = d.uni_prior[q][1];
            c.step[q] = d.step_size[q];
            c.lgden[q] = std::log(d.gauss_prior[q][1] * std::sqrt(2.0 * PI));
        }
        c.lnuref9 = std::log(1.0 * c.nu_ref / 1.0e9);
        for (int j = 0; j < M.nbands; ++j) {
            const double nu = M.band[j].nu_c;
            const double r = nu / c.nu_ref;
            c.lnr[j] = std::log(r);
            c.lnu9[j] = std::log(1.0 * nu / 1.0e9);
            c.cst[j] = 0.0;
            if (c.type == DANGX_CMB) c.cst[j] = 1.0 / host_a2t(ctx, j);
            else if (c.type == DANGX_FREEFREE) c.cst[j] = 1.0 / (r * r);
            else if (c.type == DANGX_LOGNORMAL) { const double q = c.nu_ref / nu; c.cst[j] = q * q; }
        }
        c.const_planes = 0;
        if (M.all_delta) {
            const unsigned cp = (c.nind == 0) ? 7u : ctx->idx_const[l];
            for (int k = 0; k < M.nmaps; ++k)
                if ((cp >> k) & 1) {
                    c.const_planes |= 1 << k;
                    for (int j = 0; j < M.nbands; ++j)
                        c.csed[k][j] = host_sed(c, M.band[j].nu_c, c.cst[j], ctx->idx_val[l][k][0], ctx->idx_val[l][k][1]);
                }
        }
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->dm, &ctx->hm, sizeof(Model), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->dirty = false;
    return 0;
}

struct Timed {
    dangx_ctx* ctx;
    dangx_ctx::Ev ev{};
    bool on;
    Timed(dangx_ctx* c, int kid) : ctx(c), on(c->prof) {
        if (!on) return;
        ev.kid = kid;
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

int prof_collect(dangx_ctx* ctx) {
    for (auto& e : ctx->events) {
        float ms = 0.f;
        HIPCHK(ctx, hipEventSynchronize(e.b));
        HIPCHK(ctx, hipEventElapsedTime(&ms, e.a, e.b));
        ctx->prof_ms[e.kid] += ms;
        ctx->prof_n[e.kid] += 1;
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    ctx->events.clear();
    return 0;
}

int ensure_partial(dangx_ctx* ctx, long long n) {
    if (n <= ctx->partial_cap) return 0;
    if (ctx->partial) (void)hipFree(ctx->partial);
    HIPCHK(ctx, hipMalloc(&ctx->partial, sizeof(double) * (size_t)n));
    ctx->partial_cap = n;
    return 0;
}

int ensure_work(dangx_ctx* ctx, long long n) {
    if (n <= ctx->work_cap) return 0;
    for (auto& w : ctx->work) {
        if (w) (void)hipFree(w);
        w = nullptr;
        HIPCHK(ctx, hipMalloc(&w, sizeof(double) * (size_t)n));
    }
    ctx->work_cap = n;
    return 0;
}

int flag_planes_h(int flag) { return (flag & DANGX_FLAG_QU) ? 2 : 1; }

int make_group(dangx_ctx* ctx, int group, int flag, GroupArgs& a) {
    if (flag != DANGX_FLAG_T && flag != DANGX_FLAG_Q && flag != DANGX_FLAG_U && flag != DANGX_FLAG_QU)
        return fail(ctx, "flag must be exactly one of T(1), Q(2), U(4), Q+U(8)");
    if (ctx->dims.nmaps < 3 && flag != DANGX_FLAG_T) return fail(ctx, "polarisation flag needs nmaps == 3");
    std::memset(&a, 0, sizeof(a));
    a.flag = flag;
    for (int l = 0; l < ctx->hm.ncomp; ++l) {
        const dangx_comp_desc& d = ctx->desc[l];
        if (d.cg_group == group && d.sample_amplitude) {
            if (a.ng >= MAXG) return fail(ctx, "too many components in CG group");
            a.gc[a.ng++] = l;
        } else {
            unsigned planes = 0;  // planes this (group, flag) works on
            for (int pl = 0; pl < flag_planes_h(flag); ++pl)
                planes |= 1u << (((flag & DANGX_FLAG_QU) ? 2 + pl : (flag & DANGX_FLAG_T) ? 1 : (flag & DANGX_FLAG_Q) ? 2 : 3) - 1);
            if ((ctx->plane_nz[l] & planes) || ctx->desc[l].type == DANGX_TCMB) a.oc[a.no++] = l;  // all-zero plane: 0*sed, skipped
        }
    }
    if (a.ng == 0) return fail(ctx, "Woah there, number of CG components = 0 for CG group " + std::to_string(group));
    return 0;
}

template <template <int> class L, typename... Args>
int dispatch_ng(dangx_ctx* ctx, int ng, Args&&... args) {
    switch (ng) {
    case 1: return L<1>::run(ctx, args...);
    case 2: return L<2>::run(ctx, args...);
    case 3: return L<3>::run(ctx, args...);
    case 4: return L<4>::run(ctx, args...);
    case 5: return L<5>::run(ctx, args...);
    case 6: return L<6>::run(ctx, args...);
    case 7: return L<7>::run(ctx, args...);
    case 8: return L<8>::run(ctx, args...);
    default: return fail(ctx, "unsupported group size");
    }
}

inline unsigned nblocks(long long n, int bs = BLOCK) { return (unsigned)((n + bs - 1) / bs); }

template <int NG>
struct LaunchAmp {
    static int run(dangx_ctx* ctx, const GroupArgs& a, long long SN) {
        Timed t(ctx, DANGX_K_AMP_DIRECT);
        const int nb = ctx->hm.nbands;
        const size_t tabsz = (size_t)(TROWS * NG + 3) * nb * sizeof(double);
        int bs = 64, best = 0;  // block size that keeps the most waves resident in 160 KiB of LDS (ties: larger block)
        for (int cand : {256, 128, 64}) {
            const size_t need = tabsz + (size_t)2 * nb * cand * sizeof(double);
            if (need > 160 * 1024) continue;
            const int waves = std::min<int>((int)((160 * 1024) / need) * (cand / 64), 32);
            if (waves > best) { best = waves; bs = cand; }
        }
        const size_t ldsz = tabsz + (size_t)2 * nb * bs * sizeof(double);
        if (ctx->hm.all_delta && a.no == 0)
            hipLaunchKernelGGL((k_amp_direct<NG, true>), dim3(nblocks(SN, bs)), dim3(bs), ldsz, ctx->stream, ctx->dm, a, ctx->counters);
        else
            hipLaunchKernelGGL((k_amp_direct<NG, false>), dim3(nblocks(SN, bs)), dim3(bs), ldsz, ctx->stream, ctx->dm, a, ctx->counters);
        return 0;
    }
};
template <int NG>
struct LaunchRhs {
    static int run(dangx_ctx* ctx, const GroupArgs& a, long long SN, double* b) {
        Timed t(ctx, DANGX_K_CG_VEC);
        hipLaunchKernelGGL(k_rhs<NG>, dim3(nblocks(SN)), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, b);
        return 0;
    }
};
template <int NG>
struct LaunchAx {
    static int run(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* x, double* res, double* part) {
        Timed t(ctx, DANGX_K_CG_AX);
        hipLaunchKernelGGL(k_Ax<NG>, dim3(nblocks(SN)), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, x, res, part);
        return 0;
    }
};
template <int NG>
struct LaunchSv {
    static int run(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* eta, double* res) {
        Timed t(ctx, DANGX_K_CG_VEC);
        hipLaunchKernelGGL(k_sample_vector<NG>, dim3(nblocks(SN)), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, eta, res);
        return 0;
    }
};

// sum of block partials -> host double (deterministic order)
int reduce_to_host(dangx_ctx* ctx, long long nblk, double* out) {
    {
        Timed t(ctx, DANGX_K_REDUCE);
        hipLaunchKernelGGL(k_reduce, dim3(1), dim3(BLOCK), 0, ctx->stream, ctx->partial, nblk, ctx->scalars);
    }
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->scalars, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// cg_search on the device, src/dang_cg_mod.f90:179-324.  work[0]=x, [1]=r, [2]=d, [3]=q, [4]=b2, [5]=eta/b
int device_cg(dangx_ctx* ctx, const GroupArgs& a, int i_max, double converge, int* iters) {
    const long long SN = (long long)flag_planes_h(a.flag) * ctx->hm.npix;
    const long long n = SN * a.ng;
    if (ensure_work(ctx, n)) return 1;
    if (ensure_partial(ctx, nblocks(n))) return 1;
    double *x = ctx->work[0], *r = ctx->work[1], *d = ctx->work[2], *q = ctx->work[3], *b2 = ctx->work[4], *tmp = ctx->work[5];
    hipStream_t st = ctx->stream;
    // b = compute_rhs
    if (dispatch_ng<LaunchRhs>(ctx, a.ng, a, SN, tmp)) return 1;
    if (a.ml_mode == DANGX_ML_SAMPLE) {  // b2 = b + compute_sample_vector(eta)
        hipLaunchKernelGGL(k_draw_eta, dim3(nblocks(SN)), dim3(BLOCK), 0, st, ctx->dm, a, r);
        if (dispatch_ng<LaunchSv>(ctx, a.ng, a, SN, r, q)) return 1;
        hipLaunchKernelGGL(k_cg_vec, dim3(nblocks(n)), dim3(BLOCK), 0, st, 3, n, 0.0, b2, nullptr, nullptr, q, tmp, nullptr);
    } else {
        HIPCHK(ctx, hipMemcpyAsync(b2, tmp, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
    }
    // x0 = current amplitudes (the reference keeps self%x; identical as amplitudes only change via unpack)
    hipLaunchKernelGGL(k_pack, dim3(nblocks(SN)), dim3(BLOCK), 0, st, ctx->dm, a, x, 0);
    if (dispatch_ng<LaunchAx>(ctx, a.ng, a, SN, x, q, nullptr)) return 1;
    hipLaunchKernelGGL(k_cg_vec, dim3(nblocks(n)), dim3(BLOCK), 0, st, 0, n, 0.0, x, r, d, q, b2, ctx->partial);
    double delta_new = 0.0, delta_old, dq = 0.0;
    if (reduce_to_host(ctx, nblocks(n), &delta_new)) return 1;
    int i = 1;
    while (i < i_max && delta_new > converge) {
        if (ensure_partial(ctx, nblocks(SN))) return 1;
        if (dispatch_ng<LaunchAx>(ctx, a.ng, a, SN, d, q, ctx->partial)) return 1;
        if (reduce_to_host(ctx, nblocks(SN), &dq)) return 1;
        const double alpha = delta_new / dq;
        {
            Timed t(ctx, DANGX_K_CG_VEC);
            hipLaunchKernelGGL(k_cg_vec, dim3(nblocks(n)), dim3(BLOCK), 0, st, 1, n, alpha, x, r, d, q, b2, ctx->partial);
        }
        delta_old = delta_new;
        if (reduce_to_host(ctx, nblocks(n), &delta_new)) return 1;
        const double beta = delta_new / delta_old;
        {
            Timed t(ctx, DANGX_K_CG_VEC);
            hipLaunchKernelGGL(k_cg_vec, dim3(nblocks(n)), dim3(BLOCK), 0, st, 2, n, beta, x, r, d, q, b2, nullptr);
        }
        i = i + 1;
    }
    hipLaunchKernelGGL(k_pack, dim3(nblocks(SN)), dim3(BLOCK), 0, st, ctx->dm, a, x, 1);
    if (iters) *iters = i;
    return 0;
}

int check_comp(dangx_ctx* ctx, int comp) {
    if (comp < 0 || comp >= ctx->dims.ncomp) return fail(ctx, "component index out of range");
    return 0;
}

}  // namespace

// ======================================================================= C ABI

extern "C" {

const char* dangx_version(void) { return "dangx 0.1 (gfx950)"; }

int dangx_create(dangx_ctx** out, const dangx_dims* dims) {
    if (!out || !dims) return 1;
    *out = nullptr;
    if (dims->npix <= 0 || (dims->nmaps != 1 && dims->nmaps != 3) || dims->nbands <= 0 || dims->nbands > MAXB ||
        dims->ncomp <= 0 || dims->ncomp > MAXC)
        return 2;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return 3;  // no HIP device: fail loudly, no CPU fallback
    dangx_ctx* ctx = new dangx_ctx();
    ctx->dims = *dims;
    if (dims->device >= 0) {
        if (hipSetDevice(dims->device) != hipSuccess) { delete ctx; return 4; }
        ctx->device = dims->device;
    } else {
        (void)hipGetDevice(&ctx->device);
    }
    Model& M = ctx->hm;
    std::memset(&M, 0, sizeof(M));
    M.npix = dims->npix; M.nmaps = dims->nmaps; M.nbands = dims->nbands; M.ncomp = dims->ncomp;
    M.pix0 = dims->pix0; M.tcmb = 2.7255;  // src/dang_util_mod.f90:15
    for (int j = 0; j < MAXB; ++j) { M.gain[j] = 1.0; M.offset[j] = 0.0; }  // src/dang_data_mod.f90:127-128
    if (hipMalloc(&ctx->dm, sizeof(Model)) != hipSuccess || hipMalloc(&ctx->scalars, 8 * sizeof(double)) != hipSuccess ||
        hipMalloc(&ctx->chi_cache, 6 * sizeof(double)) != hipSuccess ||
        hipMalloc(&ctx->rows_out, (4 * MAXB + 8) * sizeof(double)) != hipSuccess ||
        hipMalloc(&ctx->counters, 4 * sizeof(unsigned long long)) != hipSuccess) {
        delete ctx;
        return 5;
    }
    *out = ctx;
    return 0;
}

int dangx_destroy(dangx_ctx* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->own_data) { (void)hipFree(ctx->sig); (void)hipFree(ctx->rms); (void)hipFree(ctx->mask); }
    for (int l = 0; l < MAXC; ++l) {
        if (ctx->amp[l] && ctx->own_amp[l]) (void)hipFree(ctx->amp[l]);
        if (ctx->idx[l] && ctx->own_idx[l]) (void)hipFree(ctx->idx[l]);
    }
    for (auto& w : ctx->work) if (w) (void)hipFree(w);
    if (ctx->partial) (void)hipFree(ctx->partial);
    if (ctx->d_bp_nu0) { (void)hipFree(ctx->d_bp_nu0); (void)hipFree(ctx->d_bp_tau0); }
    if (ctx->fs_data) (void)hipFree(ctx->fs_data);
    (void)hipFree(ctx->rows_out);
    (void)hipFree(ctx->dm); (void)hipFree(ctx->scalars); (void)hipFree(ctx->counters); (void)hipFree(ctx->chi_cache);
    for (auto& e : ctx->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    delete ctx;
    return 0;
}

const char* dangx_last_error(const dangx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int dangx_set_stream(dangx_ctx* ctx, void* s) {
    if (!ctx) return 1;
    ctx->stream = (hipStream_t)s;
    return 0;
}

int dangx_synchronize(dangx_ctx* ctx) {
    if (!ctx) return 1;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dangx_set_band(dangx_ctx* ctx, int band, double nu_c, int n, const double* nu0, const double* tau0) {
    if (!ctx) return 1;
    if (band < 0 || band >= ctx->dims.nbands) return fail(ctx, "band index out of range");
    if (n < 0 || (n > 0 && (!nu0 || !tau0))) return fail(ctx, "bad bandpass arrays");
    Band& b = ctx->hm.band[band];
    b.nu_c = (nu_c < 1e9) ? nu_c * 1e9 : nu_c;  // src/dang_bp_mod.f90:35-37
    b.n = n;
    b.off = (int)ctx->bp_nu0.size();
    for (int i = 0; i < n; ++i) { ctx->bp_nu0.push_back(nu0[i]); ctx->bp_tau0.push_back(tau0[i]); }
    ctx->band_set[band] = true;
    ctx->dirty = true;
    return 0;
}

int dangx_set_component(dangx_ctx* ctx, int comp, const dangx_comp_desc* d) {
    if (!ctx || !d) return 1;
    if (check_comp(ctx, comp)) return 1;
    if (d->type < DANGX_POWERLAW || d->type > DANGX_TCMB)
        return fail(ctx, "Error - unrecognized component type (only diffuse types are built)");
    const int want = (d->type == DANGX_MBB || d->type == DANGX_LOGNORMAL) ? 2 : (d->type == DANGX_CMB ? 0 : 1);
    if (d->type == DANGX_TCMB && d->sample_amplitude)
        return fail(ctx, "T_cmb cannot be amplitude-sampled on the device yet (SURVEY 8f rank 1)");
    if (d->nindices != want) return fail(ctx, "nindices does not match the component type");
    (void)hipSetDevice(ctx->device);
    const size_t plane = (size_t)ctx->dims.npix * ctx->dims.nmaps * sizeof(double);
    if (!ctx->amp[comp]) {
        HIPCHK(ctx, hipMalloc(&ctx->amp[comp], plane));
        HIPCHK(ctx, hipMemset(ctx->amp[comp], 0, plane));
        ctx->own_amp[comp] = true;
    }
    if (d->nindices > 0 && !ctx->idx[comp]) {
        HIPCHK(ctx, hipMalloc(&ctx->idx[comp], plane * d->nindices));
        HIPCHK(ctx, hipMemset(ctx->idx[comp], 0, plane * d->nindices));
        ctx->own_idx[comp] = true;
    }
    ctx->desc[comp] = *d;
    if (ctx->desc[comp].nu_ref < 1e7) ctx->desc[comp].nu_ref *= 1e9;  // src/dang_param_mod.f90:571-573
    ctx->comp_set[comp] = true;
    ctx->dirty = true;
    return 0;
}

int dangx_set_tcmb(dangx_ctx* ctx, double T) {
    if (!ctx) return 1;
    ctx->hm.tcmb = T;
    ctx->dirty = true;
    return 0;
}

int dangx_set_calibration(dangx_ctx* ctx, const double* gain, const double* offset) {
    if (!ctx) return 1;
    for (int j = 0; j < ctx->dims.nbands; ++j) {
        if (gain) ctx->hm.gain[j] = gain[j];
        if (offset) ctx->hm.offset[j] = offset[j];
    }
    ctx->dirty = true;
    return 0;
}

int dangx_upload_data(dangx_ctx* ctx, const double* sig, const double* rms, const double* mask) {
    if (!ctx || !sig || !rms || !mask) return 1;
    (void)hipSetDevice(ctx->device);
    const size_t nmap = (size_t)ctx->dims.npix * ctx->dims.nmaps * sizeof(double);
    const size_t nall = nmap * ctx->dims.nbands;
    if (!ctx->own_data) {
        ctx->sig = ctx->rms = ctx->mask = nullptr;
        HIPCHK(ctx, hipMalloc(&ctx->sig, nall));
        HIPCHK(ctx, hipMalloc(&ctx->rms, nall));
        HIPCHK(ctx, hipMalloc(&ctx->mask, nmap));
        ctx->own_data = true;
    }
    HIPCHK(ctx, hipMemcpy(ctx->sig, sig, nall, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->rms, rms, nall, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->mask, mask, nmap, hipMemcpyHostToDevice));
    ctx->dirty = true;
    return 0;
}

int dangx_adopt_device_data(dangx_ctx* ctx, const double* sig, const double* rms, const double* mask) {
    if (!ctx || !sig || !rms || !mask) return 1;
    if (ctx->own_data) { (void)hipFree(ctx->sig); (void)hipFree(ctx->rms); (void)hipFree(ctx->mask); ctx->own_data = false; }
    ctx->sig = const_cast<double*>(sig);
    ctx->rms = const_cast<double*>(rms);
    ctx->mask = const_cast<double*>(mask);
    ctx->dirty = true;
    return 0;
}

int dangx_put_amplitude(dangx_ctx* ctx, int comp, const double* amp) {
    if (!ctx || !amp || check_comp(ctx, comp)) return 1;
    if (!ctx->amp[comp]) return fail(ctx, "component not set");
    ctx->plane_nz[comp] = 0;
    for (int k = 0; k < ctx->dims.nmaps; ++k)
        for (long long t = 0; t < ctx->dims.npix; ++t)
            if (amp[(long long)k * ctx->dims.npix + t] != 0.0) { ctx->plane_nz[comp] |= 1u << k; break; }
    HIPCHK(ctx, hipMemcpyAsync(ctx->amp[comp], amp, (size_t)ctx->dims.npix * ctx->dims.nmaps * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int dangx_get_amplitude(dangx_ctx* ctx, int comp, double* amp) {
    if (!ctx || !amp || check_comp(ctx, comp)) return 1;
    if (!ctx->amp[comp]) return fail(ctx, "component not set");
    HIPCHK(ctx, hipMemcpyAsync(amp, ctx->amp[comp], (size_t)ctx->dims.npix * ctx->dims.nmaps * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int dangx_put_indices(dangx_ctx* ctx, int comp, const double* ind) {
    if (!ctx || !ind || check_comp(ctx, comp)) return 1;
    if (!ctx->idx[comp]) return fail(ctx, "component has no indices");
    {   // planes on which every index map is spatially constant
        const long long np = ctx->dims.npix;
        ctx->idx_const[comp] = 0;
        for (int k = 0; k < ctx->dims.nmaps; ++k) {
            bool cst = true;
            for (int q = 0; q < ctx->desc[comp].nindices && cst; ++q) {
                const double* m = ind + ((long long)q * ctx->dims.nmaps + k) * np;
                for (long long t = 1; t < np; ++t) if (m[t] != m[0]) { cst = false; break; }
                ctx->idx_val[comp][k][q] = m[0];
            }
            if (cst) ctx->idx_const[comp] |= 1u << k;
        }
        ctx->dirty = true;
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->idx[comp], ind, (size_t)ctx->dims.npix * ctx->dims.nmaps * ctx->desc[comp].nindices * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int dangx_get_indices(dangx_ctx* ctx, int comp, double* ind) {
    if (!ctx || !ind || check_comp(ctx, comp)) return 1;
    if (!ctx->idx[comp]) return fail(ctx, "component has no indices");
    HIPCHK(ctx, hipMemcpyAsync(ind, ctx->idx[comp], (size_t)ctx->dims.npix * ctx->dims.nmaps * ctx->desc[comp].nindices * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
int dangx_adopt_device_state(dangx_ctx* ctx, int comp, double* amp_dev, double* idx_dev) {
    if (!ctx || !amp_dev || check_comp(ctx, comp)) return 1;
    if (!ctx->comp_set[comp]) return fail(ctx, "component not set");
    if (ctx->desc[comp].nindices > 0 && !idx_dev) return fail(ctx, "component has indices: idx_dev required");
    if (ctx->amp[comp] && ctx->own_amp[comp]) (void)hipFree(ctx->amp[comp]);
    if (ctx->idx[comp] && ctx->own_idx[comp]) (void)hipFree(ctx->idx[comp]);
    ctx->amp[comp] = amp_dev; ctx->own_amp[comp] = false;
    {   // which planes hold a non-zero amplitude right now (one small kernel, once)
        unsigned f = 0;
        unsigned* df = reinterpret_cast<unsigned*>(ctx->counters + 2);
        HIPCHK(ctx, hipMemsetAsync(df, 0, sizeof(unsigned), ctx->stream));
        hipLaunchKernelGGL(k_any_nonzero, dim3(1024), dim3(BLOCK), 0, ctx->stream, amp_dev, (long long)ctx->dims.npix, ctx->dims.nmaps, df);
        HIPCHK(ctx, hipMemcpyAsync(&f, df, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->plane_nz[comp] = f;
    }
    ctx->idx[comp] = (ctx->desc[comp].nindices > 0) ? idx_dev : nullptr; ctx->own_idx[comp] = false;
    ctx->idx_const[comp] = 0;
    if (ctx->idx[comp]) {   // planes on which every index map is spatially constant (one small kernel, once)
        const int nind = ctx->desc[comp].nindices, nmaps = ctx->dims.nmaps;
        unsigned f = 0;
        double first[6] = {};
        unsigned* df = reinterpret_cast<unsigned*>(ctx->counters + 2);
        HIPCHK(ctx, hipMemsetAsync(df, 0, sizeof(unsigned), ctx->stream));
        hipLaunchKernelGGL(k_not_constant, dim3(1024), dim3(BLOCK), 0, ctx->stream, idx_dev, (long long)ctx->dims.npix, nmaps, nind, df, ctx->scalars + 2);
        HIPCHK(ctx, hipMemcpyAsync(&f, df, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(first, ctx->scalars + 2, sizeof(first), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < nmaps; ++k) {
            bool cst = true;
            for (int q = 0; q < nind; ++q) {
                if ((f >> (q * 3 + k)) & 1) cst = false;
                ctx->idx_val[comp][k][q] = first[q * 3 + k];
            }
            if (cst) ctx->idx_const[comp] |= 1u << k;
        }
    }
    ctx->dirty = true;
    return 0;
}
void* dangx_amplitude_devptr(dangx_ctx* ctx, int comp) { return (ctx && comp >= 0 && comp < MAXC) ? ctx->amp[comp] : nullptr; }
void* dangx_indices_devptr(dangx_ctx* ctx, int comp) { return (ctx && comp >= 0 && comp < MAXC) ? ctx->idx[comp] : nullptr; }

int64_t dangx_group_size(dangx_ctx* ctx, int group, int flag) {
    GroupArgs a;
    if (!ctx || make_group(ctx, group, flag, a)) return -1;
    return (int64_t)a.ng * flag_planes_h(flag) * ctx->hm.npix;
}

int dangx_amp_sample(dangx_ctx* ctx, int group, int flag, int ml_mode, int solver, int fluct_mode, uint64_t seed,
                     uint64_t stream, int i_max, double converge, int* cg_iters, int64_t* n_not_spd) {
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
    if (solver == DANGX_SOLVER_CG) {
        if (fluct_mode != DANGX_FLUCT_REFERENCE && ml_mode == DANGX_ML_SAMPLE)
            return fail(ctx, "the CG solver reproduces the reference's fluctuation term only");
        return device_cg(ctx, a, i_max, converge, cg_iters);
    }
    if (n_not_spd) HIPCHK(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long), ctx->stream));
    if (dispatch_ng<LaunchAmp>(ctx, a.ng, a, SN)) return 1;
    HIPCHK(ctx, hipGetLastError());
    if (n_not_spd) {
        unsigned long long v = 0;
        HIPCHK(ctx, hipMemcpyAsync(&v, ctx->counters, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *n_not_spd = (int64_t)v;
    }
    return 0;
}

int dangx_index_sample(dangx_ctx* ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                       uint64_t stream, int64_t* accepted) {
    if (!ctx || check_comp(ctx, comp)) return 1;
    (void)hipSetDevice(ctx->device);
    {   // this sweep makes the component's index map pixel dependent on the touched planes
        unsigned touched = 0;
        if (map_n == -1) touched = 6u; else if (map_n >= 1 && map_n <= 3) touched = 1u << (map_n - 1);
        if (ctx->idx_const[comp] & touched) { ctx->idx_const[comp] &= ~touched; ctx->dirty = true; }
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
    if (d.type == DANGX_TCMB) return fail(ctx, "T_cmb is sampled full-sky (index_mode 1) in the reference: not built (SURVEY 8f rank 2)");
    if (ml_mode != DANGX_ML_SAMPLE && ml_mode != DANGX_ML_OPTIMIZE) return fail(ctx, "bad ml_mode");
    const int Sp = a.s2 - a.s1 + 1;
    a.others = 0;
    for (int l = 0; l < ctx->hm.ncomp; ++l)
        if (l != comp && ((ctx->plane_nz[l] & ((1u << (a.s1 - 1)) | (1u << (a.s2 - 1)))) || ctx->desc[l].type == DANGX_TCMB))
            a.others |= 1u << l;
    // chain mode: factorised SED when every band is a delta bandpass
    const bool all_delta = ctx->hm.all_delta != 0;
    a.mode = CH_GENERIC;
    if (all_delta) {
        if (d.type == DANGX_POWERLAW) a.mode = CH_POW;
        else if (d.type == DANGX_MBB) a.mode = nind == 0 ? CH_MBB_BETA : CH_MBB_T;
        else if (d.type == DANGX_LOGNORMAL) a.mode = nind == 0 ? CH_LOGN_NUP : CH_LOGN_W;
    }
    // LDS columns: (2*Sp+1)*nb doubles per thread; pick the block so that >= 2 blocks fit in 160 KiB
    const size_t per_thread = (size_t)(2 * Sp + 1) * ctx->hm.nbands * sizeof(double);
    const size_t tabsz = (size_t)(TROWS * ctx->hm.ncomp + 3) * ctx->hm.nbands * sizeof(double);
    int bs = 256;
    while (bs > 64 && tabsz + per_thread * bs > 76 * 1024) bs >>= 1;
    const size_t lds = tabsz + per_thread * bs;
    const bool reg_ok = d.lnl_type[nind] == DANGX_LNL_CHISQ && d.prior_type[nind] != DANGX_PRIOR_JEFFREYS && a.mode != CH_GENERIC &&
                        (ctx->hm.nbands == 3 || ctx->hm.nbands == 5 || ctx->hm.nbands == 6 || ctx->hm.nbands == 8 ||
                         ctx->hm.nbands == 10 || ctx->hm.nbands == 20);
    if (reg_ok) bs = BLOCK;  // register-resident form: no LDS columns
    const unsigned nblk = nblocks(ctx->hm.npix, bs);
    constexpr int RSTAGE = 128;  // blocks of the first reduction stage
    if (ensure_partial(ctx, 4ll * nblk + 4ll * RSTAGE)) return 1;
    if (accepted) HIPCHK(ctx, hipMemsetAsync(ctx->counters + 1, 0, sizeof(unsigned long long), ctx->stream));
    {
        Timed t(ctx, DANGX_K_INDEX_MH);
        unsigned long long* accp = accepted ? ctx->counters + 1 : nullptr;
        const int nb = ctx->hm.nbands;
        const int tb = (nb % 5 == 0) ? 5 : (nb % 4 == 0) ? 4 : (nb % 3 == 0) ? 3 : 1;
        const bool fast = d.lnl_type[nind] == DANGX_LNL_CHISQ &&
                          (a.mode == CH_POW || a.mode == CH_MBB_BETA || a.mode == CH_MBB_T);
#define DX_LAUNCH_MH(MODE_, SP_, TB_)                                                                            \
        hipLaunchKernelGGL((k_index_mh<MODE_, SP_, TB_>), dim3(nblk), dim3(bs), lds, ctx->stream, ctx->dm, a, accp, ctx->partial)
#define DX_MH_TB(MODE_, SP_)                                                                                     \
        do { if (tb == 5) DX_LAUNCH_MH(MODE_, SP_, 5); else if (tb == 4) DX_LAUNCH_MH(MODE_, SP_, 4);            \
             else if (tb == 3) DX_LAUNCH_MH(MODE_, SP_, 3); else DX_LAUNCH_MH(MODE_, SP_, 1); } while (0)
#define DX_MH_SP(MODE_) do { if (Sp == 2) DX_MH_TB(MODE_, 2); else DX_MH_TB(MODE_, 1); } while (0)
        const bool regmode = a.mode != CH_GENERIC && d.lnl_type[nind] == DANGX_LNL_CHISQ && d.prior_type[nind] != DANGX_PRIOR_JEFFREYS;
        const bool regform = regmode && (nb == 3 || nb == 5 || nb == 6 || nb == 8 || nb == 10 || nb == 20);
#define DX_LAUNCH_REG(MODE_, SP_, NB_, L_)                                                                       \
        hipLaunchKernelGGL((k_index_mh_reg<MODE_, SP_, NB_, L_>), dim3(nblk), dim3(BLOCK),                       \
                           (L_) ? (size_t)(SP_) * (NB_) * BLOCK * sizeof(double) : 0, ctx->stream, ctx->dm, a, accp, ctx->partial)
#define DX_REG_NB(MODE_, SP_)                                                                                    \
        do { if (nb == 10) DX_LAUNCH_REG(MODE_, SP_, 10, false); else if (nb == 5) DX_LAUNCH_REG(MODE_, SP_, 5, false);     \
             else if (nb == 3) DX_LAUNCH_REG(MODE_, SP_, 3, false); else if (nb == 6) DX_LAUNCH_REG(MODE_, SP_, 6, false);  \
             else if (nb == 8) DX_LAUNCH_REG(MODE_, SP_, 8, false);                                                         \
             else DX_LAUNCH_REG(MODE_, SP_, 20, ((SP_) == 2)); } while (0)
#define DX_REG_SP(MODE_) do { if (Sp == 2) DX_REG_NB(MODE_, 2); else DX_REG_NB(MODE_, 1); } while (0)
        if (regform && a.mode == CH_POW) DX_REG_SP(CH_POW);
        else if (regform && a.mode == CH_MBB_BETA) DX_REG_SP(CH_MBB_BETA);
        else if (regform && a.mode == CH_MBB_T) DX_REG_SP(CH_MBB_T);
        else if (regform && a.mode == CH_LOGN_NUP) DX_REG_SP(CH_LOGN_NUP);
        else if (regform && a.mode == CH_LOGN_W) DX_REG_SP(CH_LOGN_W);
        else if (!fast) DX_LAUNCH_MH(CH_GENERIC, 1, 1);
        else if (a.mode == CH_POW) DX_MH_SP(CH_POW);
        else if (a.mode == CH_MBB_BETA) DX_MH_SP(CH_MBB_BETA);
        else DX_MH_SP(CH_MBB_T);
#undef DX_REG_SP
#undef DX_REG_NB
#undef DX_LAUNCH_REG
#undef DX_MH_SP
#undef DX_MH_TB
#undef DX_LAUNCH_MH
    }
    {   // fused chi^2 of the touched planes (before = state left by the amplitude phase, after = new state)
        const bool wb = !ctx->touched_since_amp[a.s1 - 1];
        Timed t(ctx, DANGX_K_REDUCE);
        double* stage = ctx->partial + 4ll * nblk;
        hipLaunchKernelGGL(k_reduce_rows, dim3(RSTAGE), dim3(BLOCK), 0, ctx->stream, ctx->partial, (long long)nblk, 4, stage);
        hipLaunchKernelGGL(k_reduce_chi, dim3(1), dim3(BLOCK), 0, ctx->stream, stage, (long long)RSTAGE, a.s1, a.s2,
                           wb ? 1 : 0, ctx->chi_cache);
        for (int k = a.s1; k <= a.s2; ++k) {
            if (wb) ctx->chi_before_valid[k - 1] = true;
            ctx->chi_after_valid[k - 1] = true;
            ctx->touched_since_amp[k - 1] = true;
        }
    }
    HIPCHK(ctx, hipGetLastError());
    if (accepted) {
        unsigned long long v = 0;
        HIPCHK(ctx, hipMemcpyAsync(&v, ctx->counters + 1, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *accepted = (int64_t)v;
    }
    return 0;
}

// chi^2 of planes pol_lo..pol_hi from the values fused into the index sweeps: which = 0 -> the state the
// amplitude phase left (captured by the first sweep on each plane), 1 -> the current state.  Fails (status 2)
// if some plane has not been covered by a sweep since its last amplitude update: use dangx_sky_model_chisq.
int dangx_chisq_cached_dev(dangx_ctx* ctx, int which, int pol_lo, int pol_hi, double* out_dev) {
    if (!ctx || !out_dev || (which != 0 && which != 1)) return 1;
    if (pol_lo < 1 || pol_hi > ctx->dims.nmaps || pol_lo > pol_hi) return fail(ctx, "bad pol_type range");
    for (int k = pol_lo; k <= pol_hi; ++k)
        if (!(which ? ctx->chi_after_valid[k - 1] : ctx->chi_before_valid[k - 1])) {
            ctx->err = "cached chi^2 not available for plane " + std::to_string(k);
            return 2;
        }
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

static int sky_chisq_launch(dangx_ctx* ctx, int pol_lo, int pol_hi, double* sky_d, double* res_d, double* chi_d, double* out_dev) {
    if (pol_lo < 1 || pol_hi > ctx->dims.nmaps || pol_lo > pol_hi) return fail(ctx, "bad pol_type range");
    int bs = 256;
    while (bs > 64 && (size_t)ctx->hm.nbands * bs * sizeof(double) > 32 * 1024) bs >>= 1;
    const unsigned nblk = nblocks(ctx->hm.npix, bs);
    constexpr int RSTAGE = 128;
    if (ensure_partial(ctx, (long long)nblk + RSTAGE)) return 1;
    {
        Timed t(ctx, DANGX_K_SKY_CHISQ);
        hipLaunchKernelGGL(k_sky_chisq, dim3(nblk), dim3(bs), (size_t)ctx->hm.nbands * bs * sizeof(double), ctx->stream,
                           ctx->dm, pol_lo, pol_hi, sky_d, res_d, chi_d, ctx->partial);
    }
    {
        Timed t(ctx, DANGX_K_REDUCE);
        double* stage = ctx->partial + nblk;
        hipLaunchKernelGGL(k_reduce_rows, dim3(RSTAGE), dim3(BLOCK), 0, ctx->stream, ctx->partial, (long long)nblk, 1, stage);
        hipLaunchKernelGGL(k_reduce, dim3(1), dim3(BLOCK), 0, ctx->stream, stage, (long long)RSTAGE, out_dev);
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
        if (hipMemcpyAsync(&v, ctx->scalars, sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = 1;
        if (sky && hipMemcpyAsync(sky, sky_d, nall, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = 1;
        if (res && hipMemcpyAsync(res, res_d, nall, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = 1;
        if (chi_map && hipMemcpyAsync(chi_map, chi_d, nmap, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = 1;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = 1;
        if (rc) ctx->err = "copy-back failed in dangx_sky_model_chisq";
        if (chisq_sum) *chisq_sum = v;
    }
    if (sky_d) (void)hipFree(sky_d);
    if (res_d) (void)hipFree(res_d);
    if (chi_d) (void)hipFree(chi_d);
    return rc;
}


// ---- full-sky index mode / tuner / gain fit primitives ---------------------------------------------

static int map_planes(dangx_ctx* ctx, int map_n, int& s1, int& s2) {
    if (map_n == -1) { s1 = 2; s2 = 3; }
    else if (map_n >= 1 && map_n <= 3) { s1 = s2 = map_n; }
    else return fail(ctx, "There is something wrong with the poltype flag (map_n must be 1,2,3 or -1)");
    if (s2 > ctx->dims.nmaps) return fail(ctx, "map_n exceeds nmaps");
    return 0;
}

int dangx_fullsky_prepare(dangx_ctx* ctx, int comp, int map_n) {
    if (!ctx || check_comp(ctx, comp)) return 1;
    (void)hipSetDevice(ctx->device);
    int s1, s2;
    if (map_planes(ctx, map_n, s1, s2) || sync_model(ctx)) return 1;
    const long long need = (long long)(s2 - s1 + 1) * ctx->hm.nbands * ctx->hm.npix;
    if (need > ctx->fs_cap) {
        if (ctx->fs_data) (void)hipFree(ctx->fs_data);
        ctx->fs_data = nullptr;
        HIPCHK(ctx, hipMalloc(&ctx->fs_data, sizeof(double) * (size_t)need));
        ctx->fs_cap = need;
    }
    unsigned others = 0;
    for (int l = 0; l < ctx->hm.ncomp; ++l)
        if (l != comp && ((ctx->plane_nz[l] & ((1u << (s1 - 1)) | (1u << (s2 - 1)))) || ctx->desc[l].type == DANGX_TCMB)) others |= 1u << l;
    hipLaunchKernelGGL(k_fullsky_prepare, dim3(nblocks(ctx->hm.npix)), dim3(BLOCK), 0, ctx->stream, ctx->dm, comp, s1, s2, others, ctx->fs_data);
    HIPCHK(ctx, hipGetLastError());
    ctx->fs_comp = comp; ctx->fs_s1 = s1; ctx->fs_s2 = s2;
    return 0;
}

// what = 0 chisq lnL (1 value), 1 marginal (2*nb*Sp values: TNd(j,k), TNT(j,k) interleaved, j outer / k inner),
// 2 jeffreys sum (1 value).  Local (this shard's) sums; the caller all-reduces and combines.
int dangx_fullsky_sums(dangx_ctx* ctx, int what, const double* theta, double* out, int nout) {
    if (!ctx || !theta || !out) return 1;
    (void)hipSetDevice(ctx->device);
    if (ctx->fs_comp < 0) return fail(ctx, "dangx_fullsky_prepare has not been called");
    if (what < 0 || what > 2) return fail(ctx, "bad sum selector");
    if (sync_model(ctx)) return 1;
    const int Sp = ctx->fs_s2 - ctx->fs_s1 + 1;
    const int rows = (what == 1) ? 2 * ctx->hm.nbands * Sp : 1;
    if (nout < rows) return fail(ctx, "output buffer too small");
    const unsigned nblk = nblocks(ctx->hm.npix);
    if (ensure_partial(ctx, (long long)rows * nblk)) return 1;
    hipLaunchKernelGGL(k_fullsky_rows, dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, ctx->fs_comp, ctx->fs_s1, ctx->fs_s2, what,
                       theta[0], theta[1], ctx->fs_data, ctx->partial);
    hipLaunchKernelGGL(k_reduce_rows_final, dim3(1), dim3(BLOCK), 0, ctx->stream, ctx->partial, (long long)nblk, rows, ctx->rows_out);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->rows_out, sizeof(double) * rows, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dangx_fill_index(dangx_ctx* ctx, int comp, int nind, int map_n, double value) {
    if (!ctx || check_comp(ctx, comp)) return 1;
    (void)hipSetDevice(ctx->device);
    int s1, s2;
    if (map_planes(ctx, map_n, s1, s2)) return 1;
    if (nind < 0 || nind >= ctx->desc[comp].nindices) return fail(ctx, "index number out of range");
    if (sync_model(ctx)) return 1;
    hipLaunchKernelGGL(k_fill_index, dim3(nblocks(ctx->hm.npix)), dim3(BLOCK), 0, ctx->stream, ctx->dm, comp, nind, s1, s2, value);
    HIPCHK(ctx, hipGetLastError());
    // the map is spatially constant on those planes now iff the component's other indices are; re-derive lazily:
    // simply mark the planes non-constant unless the component has a single index
    for (int k = s1; k <= s2; ++k) {
        if (ctx->desc[comp].nindices == 1) { ctx->idx_const[comp] |= 1u << (k - 1); ctx->idx_val[comp][k - 1][0] = value; }
        else if (ctx->idx_const[comp] & (1u << (k - 1))) ctx->idx_val[comp][k - 1][nind] = value;
        ctx->chi_before_valid[k - 1] = ctx->chi_after_valid[k - 1] = false;
    }
    ctx->dirty = true;
    return 0;
}

// c%indices(pix, map_n, 0:nindices-1) of one (local) pixel -> out[nindices]
int dangx_peek_indices(dangx_ctx* ctx, int comp, int map_n, long long pix, double* out) {
    if (!ctx || !out || check_comp(ctx, comp)) return 1;
    if (!ctx->idx[comp]) return fail(ctx, "component has no indices");
    if (map_n < 1 || map_n > ctx->dims.nmaps || pix < 0 || pix >= ctx->dims.npix) return fail(ctx, "bad map/pixel");
    for (int q = 0; q < ctx->desc[comp].nindices; ++q)
        HIPCHK(ctx, hipMemcpyAsync(out + q, ctx->idx[comp] + ((long long)q * ctx->dims.nmaps + (map_n - 1)) * ctx->dims.npix + pix,
                                   sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// fit_band_gain sums for band (0-based), map_n = 1: out[0] = sum map2*N_inv*map1, out[1] = sum map1*N_inv*map1
int dangx_gain_sums(dangx_ctx* ctx, int band, double* out) {
    if (!ctx || !out) return 1;
    (void)hipSetDevice(ctx->device);
    if (band < 0 || band >= ctx->dims.nbands) return fail(ctx, "band index out of range");
    if (sync_model(ctx)) return 1;
    const unsigned nblk = nblocks(ctx->hm.npix);
    if (ensure_partial(ctx, 2ll * nblk)) return 1;
    hipLaunchKernelGGL(k_gain_rows, dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, band, ctx->partial);
    hipLaunchKernelGGL(k_reduce_rows_final, dim3(1), dim3(BLOCK), 0, ctx->stream, ctx->partial, (long long)nblk, 2, ctx->rows_out);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->rows_out, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- secondary seams, host vectors ------------------------------------------------

static int seam_common(dangx_ctx* ctx, int group, int flag, GroupArgs& a, long long& SN, long long& n) {
    (void)hipSetDevice(ctx->device);
    if (make_group(ctx, group, flag, a) || sync_model(ctx)) return 1;
    SN = (long long)flag_planes_h(flag) * ctx->hm.npix;
    n = SN * a.ng;
    if (ensure_work(ctx, n)) return 1;
    return 0;
}

int dangx_compute_rhs(dangx_ctx* ctx, int group, int flag, double* b) {
    if (!ctx || !b) return 1;
    GroupArgs a; long long SN, n;
    if (seam_common(ctx, group, flag, a, SN, n)) return 1;
    if (dispatch_ng<LaunchRhs>(ctx, a.ng, a, SN, ctx->work[0])) return 1;
    HIPCHK(ctx, hipMemcpyAsync(b, ctx->work[0], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dangx_compute_Ax(dangx_ctx* ctx, int group, int flag, const double* x, double* res) {
    if (!ctx || !x || !res) return 1;
    GroupArgs a; long long SN, n;
    if (seam_common(ctx, group, flag, a, SN, n)) return 1;
    HIPCHK(ctx, hipMemcpyAsync(ctx->work[0], x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (dispatch_ng<LaunchAx>(ctx, a.ng, a, SN, ctx->work[0], ctx->work[1], nullptr)) return 1;
    HIPCHK(ctx, hipMemcpyAsync(res, ctx->work[1], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dangx_compute_sample_vector(dangx_ctx* ctx, int group, int flag, const double* eta, double* res) {
    if (!ctx || !eta || !res) return 1;
    GroupArgs a; long long SN, n;
    if (seam_common(ctx, group, flag, a, SN, n)) return 1;
    HIPCHK(ctx, hipMemcpyAsync(ctx->work[0], eta, sizeof(double) * (size_t)SN, hipMemcpyHostToDevice, ctx->stream));
    if (dispatch_ng<LaunchSv>(ctx, a.ng, a, SN, ctx->work[0], ctx->work[1])) return 1;
    HIPCHK(ctx, hipMemcpyAsync(res, ctx->work[1], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dangx_eval_sed(dangx_ctx* ctx, int comp, int band, int map_n, double* out) {
    if (!ctx || !out || check_comp(ctx, comp)) return 1;
    (void)hipSetDevice(ctx->device);
    if (band < 0 || band >= ctx->dims.nbands || map_n < 1 || map_n > ctx->dims.nmaps) return fail(ctx, "bad band/map");
    if (sync_model(ctx) || ensure_work(ctx, ctx->hm.npix)) return 1;
    hipLaunchKernelGGL(k_eval_sed, dim3(nblocks(ctx->hm.npix)), dim3(BLOCK), 0, ctx->stream, ctx->dm, comp, band, map_n, ctx->work[0]);
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->work[0], sizeof(double) * (size_t)ctx->hm.npix, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- profiling ---------------------------------------------------------------------

int dangx_profile_enable(dangx_ctx* ctx, int on) {
    if (!ctx) return 1;
    ctx->prof = on != 0;
    return 0;
}
int dangx_profile_reset(dangx_ctx* ctx) {
    if (!ctx) return 1;
    if (prof_collect(ctx)) return 1;
    for (int k = 0; k < DANGX_K_COUNT; ++k) { ctx->prof_ms[k] = 0.0; ctx->prof_n[k] = 0; }
    return 0;
}
int dangx_profile_get(dangx_ctx* ctx, int kid, double* total_ms, int64_t* launches) {
    if (!ctx || kid < 0 || kid >= DANGX_K_COUNT) return 1;
    if (prof_collect(ctx)) return 1;
    if (total_ms) *total_ms = ctx->prof_ms[kid];
    if (launches) *launches = ctx->prof_n[kid];
    return 0;
}

}  // extern "C"
