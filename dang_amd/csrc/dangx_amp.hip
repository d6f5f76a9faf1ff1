// dangx_amp.hip -- amplitude-phase kernels (direct block solve + the reference's CG building blocks).
#include "dx_host.h"

namespace {

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
        d = d - comp_signal(M, c, i, k, j, amp, sed_prep(c, t0, t1));
    }
    // "Still subtract templates which exist but may not be fit here" (:445-460): EVERY template / monopole of the
    // model, member of this group or not, is removed on its unfitted bands -- for a non-member a second time
    for (int w = 0; w < a.nuc; ++w) {
        const Comp& c = M.comp[a.uc[w]];
        if (!((c.corr_mask >> j) & 1)) d = d - comp_signal(M, c, i, k, j, 0.0, Prep{0, 0, 0});
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
        cs[g] = (M.comp[gl[g]].const_planes >> (k - 1)) & 1;
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
            mrow[g] = cs[g] ? sed_const_tab(tab, nb, g, k, j)
                      : !FAST ? sed_eval(M, M.comp[gl[g]], j, pr[g]) : sed_eval_tab(ty[g], tab, nb, NG, g, j, pr[g]);
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


// ---------------------------------------------------------------------------
// CG building blocks for groups that contain global-amplitude components (template / monopole / hi_fit;
// src/dang_cg_mod.f90:522-587, 717-768, 833-893, 1045-1096).  A global component contributes one row per fitted
// band: its entries of T^t(...) are sums over pixels, formed here as block partials `rowpartial[row][block]`
// (second stage: k_reduce_rows_final).  Restrictions checked on the host: global members follow the diffuse
// ones in the group; hi_fit / monopole only under flag T (they read plane 1 whatever the flag in the reference).
__device__ __forceinline__ int gl_nplanes(const Comp& c, int flag) { return (c.type == DANGX_TEMPLATE && (flag & DANGX_FLAG_QU)) ? 2 : 1; }

// every thread of the block calls this; thread 0 writes the block's sum
__device__ __forceinline__ void block_row_sum(double v, int row, double* rowpartial, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += sh[w];
        rowpartial[(long long)row * gridDim.x + blockIdx.x] = t;
    }
    __syncthreads();
}

struct UnitId { bool in; int p, i, k; bool msk; };
__device__ __forceinline__ UnitId unit_of(const Model& M, int flag) {
    UnitId q;
    const long long u = (long long)blockIdx.x * BLOCK + threadIdx.x;
    q.in = u < (long long)flag_nplanes(flag) * M.npix;
    q.p = q.in ? (int)(u / M.npix) : 0;
    q.i = q.in ? (int)(u - (long long)q.p * M.npix) : 0;
    q.k = flag_map(flag, q.p);
    q.msk = !q.in || is_masked(M.mask[q.i]);
    return q;
}

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

template <int NG>
struct LaunchSchur {
    static int run(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs* sa, long long SN, double* rows_dev) {
        const unsigned nblk = nblocks(SN);
        if (sa) {
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
int dispatch_schur(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs* sa, long long SN, double* rows_dev) {
    switch (a.ng) {
    case 0: return sa ? LaunchSchur<0>::run(ctx, a, sa, SN, rows_dev) : 0;  // no diffuse member: nothing to back-substitute
    case 1: return LaunchSchur<1>::run(ctx, a, sa, SN, rows_dev);
    case 2: return LaunchSchur<2>::run(ctx, a, sa, SN, rows_dev);
    case 3: return LaunchSchur<3>::run(ctx, a, sa, SN, rows_dev);
    case 4: return LaunchSchur<4>::run(ctx, a, sa, SN, rows_dev);
    case 5: return LaunchSchur<5>::run(ctx, a, sa, SN, rows_dev);
    case 6: return LaunchSchur<6>::run(ctx, a, sa, SN, rows_dev);
    default: return fail(ctx, "direct solve of a template group supports up to 6 diffuse members: use DANGX_SOLVER_CG");
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

int dx_launch_amp(dangx_ctx* ctx, const GroupArgs& a, long long SN) { return dispatch_ng<LaunchAmp>(ctx, a.ng, a, SN); }
int dx_launch_rhs(dangx_ctx* ctx, const GroupArgs& a, long long SN, double* b) { return dispatch_ng<LaunchRhs>(ctx, a.ng, a, SN, b); }
int dx_launch_Ax(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* x, double* res, double* part) {
    return dispatch_ng<LaunchAx>(ctx, a.ng, a, SN, x, res, part);
}
int dx_launch_sample_vector(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* eta, double* res) {
    return dispatch_ng<LaunchSv>(ctx, a.ng, a, SN, eta, res);
}
int dx_launch_rhs_mixed(dangx_ctx* ctx, const GroupArgs& a, long long SN, double* b) { return dispatch_mixed(ctx, a, SN, 0, nullptr, b); }
int dx_launch_Ax_mixed(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* x, double* res) { return dispatch_mixed(ctx, a, SN, 1, x, res); }
int dx_launch_sv_mixed(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* eta, double* res) { return dispatch_mixed(ctx, a, SN, 2, eta, res); }
int dx_launch_schur_pass1(dangx_ctx* ctx, const GroupArgs& a, const SchurArgs& sa, long long SN, double* rows_dev) {
    return dispatch_schur(ctx, a, &sa, SN, rows_dev);
}
int dx_launch_schur_pass2(dangx_ctx* ctx, const GroupArgs& a, long long SN) { return dispatch_schur(ctx, a, nullptr, SN, nullptr); }
