// dangx_amp.hip -- amplitude-phase kernels for groups of diffuse components (direct block solve + the
// reference's CG building blocks).  Groups with global-amplitude members: dangx_mixed.hip.
#include "dx_ampdata.h"

namespace {
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


}  // namespace

int dx_launch_amp(dangx_ctx* ctx, const GroupArgs& a, long long SN) {
    // DANGX_AMP_FORM=lds forces the LDS-column kernel (same-box A/B timing, tools/ab_bench.sh)
    static const bool force_lds = [] { const char* e = std::getenv("DANGX_AMP_FORM"); return e && std::strcmp(e, "lds") == 0; }();
    if (!force_lds && dx_launch_amp_reg(ctx, a, SN) == 0) return 0;
    return dispatch_ng<LaunchAmp>(ctx, a.ng, a, SN);
}
int dx_launch_rhs(dangx_ctx* ctx, const GroupArgs& a, long long SN, double* b) { return dispatch_ng<LaunchRhs>(ctx, a.ng, a, SN, b); }
int dx_launch_Ax(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* x, double* res, double* part) {
    return dispatch_ng<LaunchAx>(ctx, a.ng, a, SN, x, res, part);
}
int dx_launch_sample_vector(dangx_ctx* ctx, const GroupArgs& a, long long SN, const double* eta, double* res) {
    return dispatch_ng<LaunchSv>(ctx, a.ng, a, SN, eta, res);
}
