// dangx_coarse.hip -- the index phase beyond the per-pixel sweep at the map's resolution: the full-sky index mode's device side
// (sample_index_mh with index_mode == 1, src/dang_sample_mod.f90:229-329: staging and the sky-wide sums of a Metropolis step) and
// index sampling at a coarser Nside (:199-217, 332-483) with HEALPix udgrade_ring / udgrade_rms / udgrade_mask
// (src/dang_util_mod.f90:341-376) on the device, whole-sky and on pixel shards.  The chains of the full-sky mode, the tuner and
// the gain fit themselves are host code behind the ABI (dangx_sky.hip).
#include "dx_host.h"

namespace {

// ---------------------------------------------------------------------------
// Full-sky index mode (index_mode == 1, src/dang_sample_mod.f90:229-329), the tuner (:623-717) and the
// band-gain fit (:570-621).  With one spectral index for the whole sky the model's SED is pixel
// independent, so each Metropolis step is ONE memory-bound pass that produces a few global sums; the
// chain itself (proposal, prior, accept) runs on the host between the all-reduces (dang_amd/api.py).

// data_raw minus every other component for planes s1..s2 (:173-196, all pixels) -> out[(kk*nb + j)*npix + i]
constexpr int FP_B = 10;   // maps in flight per thread in k_fullsky_prepare
constexpr int FS_B = 5;    // ... in k_fullsky_stats: six waves per SIMD (80 registers, 19 of them spilled) beat four with ten maps in flight,
                           // 11.3 against 11.8 ms per full-sky iteration at C3 on one device; eight waves spill 39
__global__ __launch_bounds__(BLOCK) void k_fullsky_prepare(const Model* __restrict__ Mp, int comp, int s1, int s2,
                                                           unsigned others, double* __restrict__ out) {
    // a plane's bands in the thread's LDS column (dynamic LDS: nb * BLOCK doubles): five maps in flight, every other component's
    // index values and amplitude read once per plane; per band the subtractions keep their order (:180-196)
    extern __shared__ double fp_lds[];
    double* col = fp_lds + threadIdx.x;
    const Model& M = *Mp;
    const int npix = M.npix, nb = M.nbands;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const bool in = i < npix;
    const int ic = in ? i : 0;
    for (int k = s1; k <= s2; ++k) {
        for (int j0 = 0; j0 < nb; j0 += FP_B) {
            double d[FP_B];
#pragma unroll
            for (int t = 0; t < FP_B; ++t) {
                const int j = (j0 + t < nb) ? j0 + t : nb - 1;
                d[t] = M.sig[((long long)j * M.nmaps + (k - 1)) * npix + ic];
            }
#pragma unroll
            for (int t = 0; t < FP_B; ++t) {
                const int j = j0 + t;
                if (j < nb) {
                    if (k == 1) d[t] = (d[t] - M.offset[j]) / M.gain[j];
                    col[j * BLOCK] = d[t];
                }
            }
        }
        if (in)
            for (unsigned om = others; om; om &= om - 1) {
                const Comp& c2 = M.comp[__builtin_ctz(om)];
                double t0, t1;
                load_theta(M, c2, i, k, t0, t1);
                const Prep p2 = sed_prep(c2, t0, t1);
                const double a2 = c2.amp[(long long)(k - 1) * npix + i];
                for (int j = 0; j < nb; ++j) col[j * BLOCK] = col[j * BLOCK] - comp_signal(M, c2, i, k, j, a2, p2);
            }
        if (in)
            for (int j = 0; j < nb; ++j) out[((long long)(k - s1) * nb + j) * npix + i] = col[j * BLOCK];
    }
}

// row sums for one evaluation at theta: what = 0: evaluate_lnL (1 row: -1/2 sum ((d-m)/rms)^2, unmasked);
// what = 1: evaluate_marginal_lnL (2*nb*Sp rows: TNd(j,k), TNT(j,k), all pixels); what = 2: jeffreys (1 row).
// partial[row][gridDim.x]
// With sample_nside /= nside (crms /= nullptr) the sums run over the npix_c pixels of the DEGRADED data / rms / mask
// ([kk][j][npix_c] and [npix_c]) while eval_signal reads c%amplitude at the coarse pixel number in the full-resolution
// array, as the reference does (src/dang_sample_mod.f90:199-217, 548-563).
__global__ __launch_bounds__(BLOCK) void k_fullsky_rows(const Model* __restrict__ Mp, int comp, int s1, int s2, int what,
                                                        double th0, double th1, const double* __restrict__ data,
                                                        const double* __restrict__ crms, const double* __restrict__ cmask,
                                                        long long npix_c, double* __restrict__ partial) {
    __shared__ double sh[BLOCK / 64];
    const Model& M = *Mp;
    const Comp& c = M.comp[comp];
    const int nb = M.nbands, Sp = s2 - s1 + 1;
    const bool coarse = crms != nullptr;
    const int npix = coarse ? (int)npix_c : M.npix;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    // coarse: i is a coarse pixel number; the degraded maps are whole-sky on every shard, the amplitude of "pixel i" lives on
    // the shard that holds full-resolution pixel i -- each coarse pixel is summed by exactly one shard
    const long long il = coarse ? (long long)i - M.pix0 : i;
    const bool in = i < npix && il >= 0 && il < M.npix;
    const bool msk = in ? is_masked(coarse ? cmask[i] : M.mask[i]) : true;
    const Prep pr = sed_prep(c, th0, th1);
    // one index value for the whole sky: the SED is one number per band -- evaluated once per block, not once per pixel
    __shared__ double sj[MAXB];
    if (threadIdx.x < nb) sj[threadIdx.x] = sed_eval(M, c, threadIdx.x, pr);
    __syncthreads();
    const int nrows = (what == 1) ? 2 * nb * Sp : (what == 3) ? 3 * nb * Sp : 1;
    double amp[2] = {0.0, 0.0};
    if (in) for (int kk = 0; kk < Sp; ++kk) amp[kk] = c.amp[(long long)(s1 + kk - 1) * M.npix + il];
    auto rms_at = [&](int kk, int j) -> double {
        return coarse ? crms[((long long)kk * nb + j) * npix + i] : M.rms[((long long)j * M.nmaps + (s1 + kk - 1)) * npix + i];
    };
    for (int row = 0; row < nrows; ++row) {
        double v = 0.0;
        if (in) {
            if (what == 0 && !msk) {
                for (int kk = 0; kk < Sp; ++kk)
                    for (int j = 0; j < nb; ++j) {
                        const double m = signal_of(c, amp[kk], sj[j]);
                        const double t = (data[((long long)kk * nb + j) * npix + i] - m) / rms_at(kk, j);
                        v = v - 0.5 * (t * t);
                    }
            } else if (what == 1) {
                const int q = row >> 1, j = q / Sp, kk = q - j * Sp;  // (j outer, k inner) as the reference sums
                const double m = signal_of(c, amp[kk], sj[j]);
                const double rms = rms_at(kk, j);
                const double TN = m / (rms * rms);
                v = (row & 1) ? TN * m : TN * data[((long long)kk * nb + j) * npix + i];
            } else if (what == 3 && !msk) {
                // the sufficient statistics of the chisq likelihood about theta (the index is one value for the whole sky, so the SED
                // is one number per band): with r0 = (d - a s0) / sigma, rows 3q .. 3q+2 = sum r0^2, sum r0 a / sigma, sum a^2 / sigma^2
                // -- chi^2 at any other theta is sum_q [W0 - 2 (s - s0) U + (s - s0)^2 V] (dangx_sky.hip)
                const int q = row / 3, t3 = row - 3 * q, j = q / Sp, kk = q - j * Sp;
                const double rr = 1.0 / rms_at(kk, j);
                const double ar = amp[kk] * rr;
                const double r0 = (data[((long long)kk * nb + j) * npix + i] - signal_of(c, amp[kk], sj[j])) * rr;
                v = (t3 == 0) ? r0 * r0 : (t3 == 1) ? r0 * ar : ar * ar;
            } else if (what == 2 && !msk && c.is_synch) {
                for (int kk = 0; kk < Sp; ++kk)
                    for (int j = 0; j < nb; ++j) {
                        const double ss = signal_of(c, amp[kk], sj[j]);
                        const double rr = 1.0 / rms_at(kk, j);
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

// Selector 3 of dangx_fullsky_sums on its own schedule: every (plane, band) of a pixel in one pass, the three sums of a (band,
// plane) reduced over the wave at once, the block's four wave sums added in order after ONE barrier.  `fused`: the data are formed
// here -- data_raw minus every other component in dangx_fullsky_prepare's order (:173-196), a plane's bands in the thread's LDS
// column, each other component's index values / amplitude read once per plane -- instead of read from the staging buffer, which is
// then never written.  Dynamic LDS: nb * BLOCK doubles (the column) + 3 * nb * Sp * (BLOCK / 64) (the wave sums).
__global__ __launch_bounds__(BLOCK, 6) void k_fullsky_stats(const Model* __restrict__ Mp, int comp, int s1, int s2, double th0, double th1,
                                                         const double* __restrict__ data, unsigned others, int fused,
                                                         const double* __restrict__ crms, const double* __restrict__ cmask,
                                                         long long npix_c, double* __restrict__ partial) {
    extern __shared__ double fs_lds[];
    const Model& M = *Mp;
    const Comp& c = M.comp[comp];
    const int nb = M.nbands, Sp = s2 - s1 + 1, nrows = 3 * nb * Sp;
    double* col = fs_lds + threadIdx.x;                 // [band] x BLOCK
    double* wsum = fs_lds + (long long)nb * BLOCK;      // [row][wave]
    double* ocs = wsum + nrows * (BLOCK / 64);          // [other][plane][band]: the SED rows of the others whose indices are constant
    const bool coarse = crms != nullptr;
    const int npix = coarse ? (int)npix_c : M.npix;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const long long il = coarse ? (long long)i - M.pix0 : i;
    const bool in = i < npix && il >= 0 && il < M.npix;
    const bool live = in && !is_masked(coarse ? cmask[i] : M.mask[i]);
    const int ic = in ? i : 0;
    const Prep pr = sed_prep(c, th0, th1);
    __shared__ double sj[MAXB];
    if (threadIdx.x < nb) sj[threadIdx.x] = sed_eval(M, c, threadIdx.x, pr);
    if (fused) {   // (a scalar load per other component and band inside the pixel's loop otherwise: in full-sky models every index is constant)
        int oi = 0;
        for (unsigned om = others; om; om &= om - 1, ++oi) {
            const Comp& c2 = M.comp[__builtin_ctz(om)];
            for (int t = threadIdx.x; t < Sp * nb; t += BLOCK) {
                const int kk = t / nb, jj = t - kk * nb;
                ocs[(oi * Sp + kk) * nb + jj] = c2.csed[s1 + kk - 1][jj];
            }
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    for (int kk = 0; kk < Sp; ++kk) {
        const int k = s1 + kk;
        const double amp = in ? c.amp[(long long)(k - 1) * M.npix + il] : 0.0;
        // ---- the plane's cleaned data -> the column
        for (int j0 = 0; j0 < nb; j0 += FS_B) {   // FS_B bands' maps in flight
            double d[FS_B];
#pragma unroll
            for (int t = 0; t < FS_B; ++t) {
                const int j = (j0 + t < nb) ? j0 + t : nb - 1;
                d[t] = fused ? M.sig[((long long)j * M.nmaps + (k - 1)) * npix + ic] : data[((long long)kk * nb + j) * npix + ic];
            }
#pragma unroll
            for (int t = 0; t < FS_B; ++t) {
                const int j = j0 + t;
                if (j < nb) {
                    if (fused && k == 1) d[t] = (d[t] - M.offset[j]) / M.gain[j];
                    col[j * BLOCK] = d[t];
                }
            }
        }
        if (fused && live) {
            int oi = 0;
            for (unsigned om = others; om; om &= om - 1, ++oi) {
                const Comp& c2 = M.comp[__builtin_ctz(om)];
                const double a2 = c2.amp[(long long)(k - 1) * npix + i];
                if (c2.type >= DANGX_POWERLAW && c2.type <= DANGX_CMB && ((c2.const_planes >> (k - 1)) & 1)) {
                    // comp_signal of a diffuse component with constant indices: amplitude * the host-evaluated row
                    const double* row = ocs + (oi * Sp + kk) * nb;
                    for (int j = 0; j < nb; ++j) col[j * BLOCK] = col[j * BLOCK] - a2 * row[j];
                } else {
                    double t0, t1;
                    load_theta(M, c2, i, k, t0, t1);
                    const Prep p2 = sed_prep(c2, t0, t1);
                    for (int j = 0; j < nb; ++j) col[j * BLOCK] = col[j * BLOCK] - comp_signal(M, c2, i, k, j, a2, p2);
                }
            }
        }
        // ---- the three sums of every band
        for (int j0 = 0; j0 < nb; j0 += FS_B) {
            double rmt[FS_B];
#pragma unroll
            for (int t = 0; t < FS_B; ++t) {
                const int j = (j0 + t < nb) ? j0 + t : nb - 1;
                rmt[t] = coarse ? crms[((long long)kk * nb + j) * npix + ic] : M.rms[((long long)j * M.nmaps + (k - 1)) * npix + ic];
            }
#pragma unroll
          for (int t = 0; t < FS_B; ++t) {
            const int j = j0 + t;
            if (j >= nb) break;
            double v0 = 0.0, v1 = 0.0, v2 = 0.0;
            if (live) {
                const double rr = 1.0 / rmt[t];
                const double ar = amp * rr;
                const double r0 = (col[j * BLOCK] - signal_of(c, amp, sj[j])) * rr;
                v0 = r0 * r0; v1 = r0 * ar; v2 = ar * ar;
            }
            for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_down(v0, o, 64); v1 += __shfl_down(v1, o, 64); v2 += __shfl_down(v2, o, 64); }
            if ((threadIdx.x & 63) == 0) {
                const int row = 3 * (j * Sp + kk);
                wsum[row * (BLOCK / 64) + wave] = v0;
                wsum[(row + 1) * (BLOCK / 64) + wave] = v1;
                wsum[(row + 2) * (BLOCK / 64) + wave] = v2;
            }
          }
        }
    }
    __syncthreads();
    for (int row = threadIdx.x; row < nrows; row += BLOCK) {
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += wsum[row * (BLOCK / 64) + w];
        partial[(long long)row * gridDim.x + blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------
// Coarse-Nside index sampling (src/dang_sample_mod.f90:199-217, 332-483).  HEALPix is an external library of the
// reference (absent from its tree); nest2ring and udgrade_ring are restated from the published algorithm
// (Gorski et al. 2005, ApJ 622, 759; HEALPix pix_tools::nest2ring, udgrade_nr::udgrade_ring -> sub_udgrade_nest).

// udgrade of one RING map per blockIdx.y: out pixel o (RING) -> NEST -> children (degrade: mean of the good ones, in
// NEST child order; upgrade: the parent's value) -> RING.  mode 0: udgrade_ring; 1: udgrade_rms (input squared,
// sqrt(mean)*nside_out/nside_in, src/dang_util_mod.f90:341-356); 2: udgrade_mask (mean < 0.5 -> 0 else 1 when
// degrading, :358-376).  layout 0: plane q at q*npix_in; layout 1: plane q = kk*nb + j of M.rms ((j*nmaps + s1+kk-1)*npix_in)
__global__ __launch_bounds__(BLOCK) void k_udgrade(const double* __restrict__ in, double* __restrict__ out,
                                                   const int* __restrict__ n2r_in, const int* __restrict__ r2n_out,
                                                   long long npix_in, long long npix_out, int ratio, int degrade, int mode,
                                                   double scale, int layout, int nb, int nmaps, int s1) {
    const long long o = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (o >= npix_out) return;
    const int q = blockIdx.y;
    const double* src = in + (layout == 0 ? (long long)q * npix_in
                                          : ((long long)(q % nb) * nmaps + (s1 + q / nb - 1)) * npix_in);
    const long long nest = r2n_out[o];
    double v;
    if (degrade) {
        double total = 0.0;
        int nobs = 0;
        for (int ip = 0; ip < ratio; ++ip) {
            double x = src[n2r_in[nest * ratio + ip]];
            if (mode == 1) x = x * x;
            if (fabs(x - MISSVAL) > fabs(1e-5 * MISSVAL)) { total = total + x; ++nobs; }  // bad pixels do not enter the mean
        }
        v = nobs ? total / nobs : MISSVAL;
    } else {
        v = src[n2r_in[nest / ratio]];
        if (mode == 1) v = v * v;
    }
    if (mode == 1) v = sqrt(v) * scale;
    if (mode == 2 && degrade) v = (v < 0.5) ? 0.0 : 1.0;
    out[(long long)q * npix_out + o] = v;
}

// k_udgrade's degrade branch for several planes at once: the children's RING numbers are read once per QC planes (they are the
// same for every plane; one thread reads 64 consecutive ints, 256 B apart from its neighbour's) and QC maps are in flight per
// child.  Each plane's sum runs over its children in NESTED order as before.
constexpr int UDG_QC = 10;
__global__ __launch_bounds__(BLOCK) void k_udgrade_planes(const double* __restrict__ in, double* __restrict__ out,
                                                          const int* __restrict__ n2r_in, const int* __restrict__ r2n_out,
                                                          long long npix_in, long long npix_out, int ratio, int mode, double scale,
                                                          int layout, int nb, int nmaps, int s1, int nplanes) {
    const long long o = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (o >= npix_out) return;
    const int q0 = blockIdx.y * UDG_QC;
    long long off[UDG_QC];
    double total[UDG_QC];
    int nobs[UDG_QC];
#pragma unroll
    for (int t = 0; t < UDG_QC; ++t) {
        const int q = (q0 + t < nplanes) ? q0 + t : nplanes - 1;
        off[t] = (layout == 0) ? (long long)q * npix_in : ((long long)(q % nb) * nmaps + (s1 + q / nb - 1)) * npix_in;
        total[t] = 0.0; nobs[t] = 0;
    }
    const long long nest = r2n_out[o];
    const int* kids = n2r_in + nest * ratio;
    for (int ip = 0; ip < ratio; ++ip) {
        const long long c = kids[ip];
        double x[UDG_QC];
#pragma unroll
        for (int t = 0; t < UDG_QC; ++t) x[t] = in[off[t] + c];
#pragma unroll
        for (int t = 0; t < UDG_QC; ++t) {
            double v = x[t];
            if (mode == 1) v = v * v;
            if (fabs(v - MISSVAL) > fabs(1e-5 * MISSVAL)) { total[t] = total[t] + v; ++nobs[t]; }
        }
    }
#pragma unroll
    for (int t = 0; t < UDG_QC; ++t)
        if (q0 + t < nplanes) {
            double v = nobs[t] ? total[t] / nobs[t] : MISSVAL;
            if (mode == 1) v = sqrt(v) * scale;
            out[(long long)(q0 + t) * npix_out + o] = v;
        }
}

// The same degrade with the loads shared out over a wave: lane l of wave w reads one child of every plane (QW planes in flight) --
// ratio <= 64: the wave takes 64 / ratio coarse pixels, lane l = child l % ratio of its pixel l / ratio; ratio > 64: one coarse pixel,
// its children in rounds of 64.  The children of a coarse pixel are a few runs of neighbouring RING pixels, so a load instruction
// touches far fewer lines than with one thread per coarse pixel (64 children: 0.5 ms against 2.07 ms per sweep at C3).  The values
// go through LDS ([wave][plane][lane], one pad word per row) and one thread per (coarse pixel, plane) adds its children in NESTED
// order as before.  ratio: a power of four.
constexpr int UDG_QW = 10, UDG_G = BLOCK / 64;
template <bool FUSED>
__global__ __launch_bounds__(BLOCK) void k_udgrade_wave(const Model* __restrict__ Mp, int comp_unused, unsigned others, int Sp,
                                                        const double* __restrict__ in, double* __restrict__ out,
                                                        const int* __restrict__ n2r_in, const int* __restrict__ r2n_out,
                                                        long long npix_in, long long npix_out, int ratio, int mode, double scale,
                                                        int layout, int nb, int nmaps, int s1, int nplanes) {
    __shared__ double sh[UDG_G][UDG_QW][65];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cpw = (ratio < 64) ? 64 / ratio : 1, rounds = (ratio > 64) ? ratio / 64 : 1, per = (ratio < 64) ? ratio : 64;
    const int q0 = blockIdx.y * UDG_QW;
    const long long o0 = ((long long)blockIdx.x * UDG_G + wave) * cpw;          // first coarse pixel of the wave
    const long long ol = o0 + lane / per;                                        // the lane's coarse pixel
    // the summing threads: (wave g, pixel u of the wave, plane t), several rounds of the block when there are more than BLOCK of them
    const int nsum = UDG_G * cpw * UDG_QW;
    double total[3] = {0.0, 0.0, 0.0};
    int nobs[3] = {0, 0, 0};
    for (int rd = 0; rd < rounds; ++rd) {
        if (ol < npix_out) {
            const long long c = n2r_in[(long long)r2n_out[ol] * ratio + rd * 64 + lane % per];
            double x[UDG_QW];
#pragma unroll
            for (int t = 0; t < UDG_QW; ++t) {
                const int q = (q0 + t < nplanes) ? q0 + t : nplanes - 1;
                // FUSED: the raw maps (plane q = Stokes q / nb, band q % nb), cleaned below; else the staged / rms maps
                const long long off = (FUSED || layout != 0) ? ((long long)(q % nb) * nmaps + (s1 + q / nb - 1)) * npix_in : (long long)q * npix_in;
                x[t] = in[off + c];
            }
            if (FUSED) {   // data_raw minus every other component at the child pixel, dangx_fullsky_prepare's expression and order (:173-196)
                const Model& M = *Mp;
#pragma unroll
                for (int t = 0; t < UDG_QW; ++t) {
                    const int q = (q0 + t < nplanes) ? q0 + t : nplanes - 1, j = q % nb;
                    if (s1 + q / nb == 1) x[t] = (x[t] - M.offset[j]) / M.gain[j];
                }
                for (unsigned om = others; om; om &= om - 1) {
                    const Comp& c2 = M.comp[__builtin_ctz(om)];
                    Prep p2[2];
                    double a2[2];
                    for (int kk = 0; kk < Sp; ++kk) {
                        double t0, t1;
                        load_theta(M, c2, (int)c, s1 + kk, t0, t1);
                        p2[kk] = sed_prep(c2, t0, t1);
                        a2[kk] = c2.amp[(long long)(s1 + kk - 1) * npix_in + c];
                    }
#pragma unroll
                    for (int t = 0; t < UDG_QW; ++t) {
                        const int q = (q0 + t < nplanes) ? q0 + t : nplanes - 1, kk = q / nb, j = q - kk * nb;
                        x[t] = x[t] - comp_signal(M, c2, (int)c, s1 + kk, j, a2[kk], p2[kk]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < UDG_QW; ++t) sh[wave][t][lane] = x[t];
        }
        __syncthreads();
        int slot = 0;
        for (int a = threadIdx.x; a < nsum; a += BLOCK, ++slot) {
            const int g = a / (cpw * UDG_QW), r = a - g * cpw * UDG_QW, u = r / UDG_QW, t = r - u * UDG_QW;
            const long long og = ((long long)blockIdx.x * UDG_G + g) * cpw + u;
            if (og < npix_out && q0 + t < nplanes)
                for (int ip = 0; ip < per; ++ip) {
                    double v = sh[g][t][u * per + ip];
                    if (mode == 1) v = v * v;
                    if (fabs(v - MISSVAL) > fabs(1e-5 * MISSVAL)) { total[slot] = total[slot] + v; ++nobs[slot]; }
                }
        }
        __syncthreads();
    }
    int slot = 0;
    for (int a = threadIdx.x; a < nsum; a += BLOCK, ++slot) {
        const int g = a / (cpw * UDG_QW), r = a - g * cpw * UDG_QW, u = r / UDG_QW, t = r - u * UDG_QW;
        const long long og = ((long long)blockIdx.x * UDG_G + g) * cpw + u;
        if (og < npix_out && q0 + t < nplanes) {
            double v = nobs[slot] ? total[slot] / nobs[slot] : MISSVAL;
            if (mode == 1) v = sqrt(v) * scale;
            out[(long long)(q0 + t) * npix_out + og] = v;
        }
    }
}

// Pixel-sharded form of the degrade step: coarse pixel o collects, in NEST child order, only those of its children whose
// RING index lies in this shard [pix0, pix0 + npix_loc); it emits the sum of the good ones and their number.  The sums
// of all shards (added by the caller) are finished by k_udgrade_finish -- with one shard that is k_udgrade bit for bit.
__global__ __launch_bounds__(BLOCK) void k_udgrade_part(const double* __restrict__ in, double* __restrict__ tot, double* __restrict__ cnt,
                                                        const int* __restrict__ n2r_in, const int* __restrict__ r2n_out,
                                                        long long pix0, long long npix_loc, long long npix_out, int ratio, int mode,
                                                        int layout, int nb, int nmaps, int s1) {
    const long long o = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (o >= npix_out) return;
    const int q = blockIdx.y;
    const double* src = in + (layout == 0 ? (long long)q * npix_loc
                                          : ((long long)(q % nb) * nmaps + (s1 + q / nb - 1)) * npix_loc);
    const long long nest = r2n_out[o];
    double total = 0.0;
    int nobs = 0;
    for (int ip = 0; ip < ratio; ++ip) {
        const long long ring = n2r_in[nest * ratio + ip];
        if (ring < pix0 || ring >= pix0 + npix_loc) continue;
        double x = src[ring - pix0];
        if (mode == 1) x = x * x;
        if (fabs(x - MISSVAL) > fabs(1e-5 * MISSVAL)) { total = total + x; ++nobs; }
    }
    tot[(long long)q * npix_out + o] = total;
    cnt[(long long)q * npix_out + o] = (double)nobs;
}
__global__ __launch_bounds__(BLOCK) void k_udgrade_finish(const double* __restrict__ tot, const double* __restrict__ cnt,
                                                          double* __restrict__ out, long long n, int mode, double scale) {
    const long long t = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= n) return;
    const double nobs = cnt[t];
    double v = (nobs > 0.0) ? tot[t] / nobs : MISSVAL;
    if (mode == 1) v = sqrt(v) * scale;
    if (mode == 2) v = (v < 0.5) ? 0.0 : 1.0;
    out[t] = v;
}

// One Metropolis chain per COARSE pixel i, literally as the reference runs it: ddata%masks(i,1), c%indices(i,..) and
// eval_signal's c%amplitude(i,k) are the FULL-resolution arrays read at the coarse index (:362, :372-377, :548-553);
// data / rms / mask(:,1) are the degraded maps.  evaluate_lnL sums k outer, j inner with ((d-m)/rms)**2 (:171-177),
// evaluate_marginal_lnL j outer, k inner (:113-122).  index_map(i) -> idxmap[i] (0 where the chain is skipped, :223).
// On a pixel shard the chain of coarse pixel i runs where the full-resolution pixel i lives (M.pix0 <= i < M.pix0 + npix);
// the other shards leave idxmap[i] = 0 and the caller adds the maps.
__global__ __launch_bounds__(BLOCK) void k_index_mh_coarse(const Model* __restrict__ Mp, IndexArgs a, long long npix_c,
                                                           const double* __restrict__ cdata, const double* __restrict__ crms,
                                                           const double* __restrict__ cmask, double* __restrict__ idxmap,
                                                           unsigned long long* __restrict__ accepted) {
    const Model& M = *Mp;
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    unsigned long long nacc = 0;
    if (i < npix_c) {
        idxmap[i] = 0.0;
        const long long il = i - M.pix0;  // index of full-resolution pixel i in this shard's arrays
        if (il >= 0 && il < M.npix && !is_masked(M.mask[il])) {
            const Comp& c = M.comp[a.comp];
            const int nb = M.nbands, Sp = a.s2 - a.s1 + 1, q = a.nind;
            double sample0, sample1;
            load_theta(M, c, (int)il, a.s1, sample0, sample1);
            const bool first = (q == 0);
            const double other = first ? sample1 : sample0;
            double amp[2] = {0.0, 0.0};
            for (int kk = 0; kk < Sp; ++kk) amp[kk] = c.amp[(long long)(a.s1 + kk - 1) * M.npix + il];
            const int lnl_type = c.lnl_type[q];
            const bool cmasked = is_masked(cmask[i]);
            auto lnl_of = [&](double th) -> double {
                if (lnl_type == DANGX_LNL_PRIOR) return 0.0;
                const Prep pr = sed_prep(c, first ? th : other, first ? other : th);
                double lnL = 0.0;
                if (lnl_type == DANGX_LNL_CHISQ) {
                    if (cmasked) return 0.0;  // evaluate_lnL cycles on the (degraded) mask, :169
                    for (int kk = 0; kk < Sp; ++kk)
                        for (int j = 0; j < nb; ++j) {
                            const double m = signal_of(c, amp[kk], sed_eval(M, c, j, pr));
                            const long long e = ((long long)kk * nb + j) * npix_c + i;
                            const double t = (cdata[e] - m) / crms[e];
                            lnL = lnL - 0.5 * (t * t);
                        }
                } else {
                    for (int j = 0; j < nb; ++j)
                        for (int kk = 0; kk < Sp; ++kk) {
                            const double m = signal_of(c, amp[kk], sed_eval(M, c, j, pr));
                            const long long e = ((long long)kk * nb + j) * npix_c + i;
                            const double TN = m / (crms[e] * crms[e]);
                            const double TNd = TN * cdata[e], TNT = TN * m;
                            lnL = lnL - 0.5 * TNd * (1.0 / TNT) * TNd;
                        }
                }
                return lnL;
            };
            auto prior = [&](double v) -> double {
                if (c.prior_type[q] == DANGX_PRIOR_JEFFREYS) {
                    // eval_jeffreys_prior(c, data, rms, model, map_inds, i, mask(:,1), val), src/dang_lnl_mod.f90:242-304:
                    // the DEGRADED rms and mask, eval_signal / c%amplitude at the coarse pixel number, theta = (val, -)
                    double sum = 0.0;
                    if (c.is_synch && !cmasked) {
                        const Prep pr = sed_prep(c, v, 0.0);
                        for (int kk = 0; kk < Sp; ++kk)
                            for (int j = 0; j < nb; ++j) {
                                const double ss = signal_of(c, amp[kk], sed_eval(M, c, j, pr));
                                const double rr = 1.0 / crms[((long long)kk * nb + j) * npix_c + i];
                                const double tt = (rr * rr) * (ss / amp[kk]) * c.lnr[j];
                                sum = sum + tt * tt;
                            }
                    }
                    return log(sqrt(sum));
                }
                if (c.prior_type[q] != DANGX_PRIOR_GAUSSIAN) return 0.0;
                const double arg = ((v - c.gauss[q][0]) * (v - c.gauss[q][0])) / (2 * (c.gauss[q][1] * c.gauss[q][1]));
                return (arg > 745.0) ? -INFINITY : -arg - c.lgden[q];
            };
            double cur = first ? sample0 : sample1;
            double lnl = lnl_of(cur);
            bool sample_it = true;
            if (lnl_type == DANGX_LNL_PRIOR) {  // :389-392
                double u1, u2;
                sample_it = false;
                uniform2(a.seed, a.stream, (unsigned long long)i, 0u, u1, u2);
                cur = rand_normal(c.gauss[q][0], c.gauss[q][1], u1, u2);
            }
            double lnl_old = lnl + prior(cur);
            if (sample_it) {
                const double step = c.step[q], lo = c.uni[q][0], hi = c.uni[q][1];
                for (int l = 1; l <= a.nsample; ++l) {
                    double u1, u2, u3;
                    uniform3(a.seed, a.stream, (unsigned long long)i, (uint32_t)l, u1, u2, u3);
                    const double prop = cur + rand_normal(0.0, step, u1, u2);
                    if (prop < lo || prop > hi) continue;
                    const double lnl_new = lnl_of(prop) + prior(prop);
                    const double diff = lnl_new - lnl_old;
                    const bool acc = (a.ml_mode == DANGX_ML_OPTIMIZE) ? (diff > 0.0) : ((diff >= 0.0) || (exp_sat(diff) > u3));
                    if (acc) { cur = prop; lnl_old = lnl_new; ++nacc; }
                }
            }
            idxmap[i] = cur;  // :465
        }
    }
    if (accepted) {
        for (int o = 32; o > 0; o >>= 1) nacc += __shfl_down(nacc, o, 64);
        if ((threadIdx.x & 63) == 0 && nacc) atomicAdd(accepted, nacc);
    }
}

// udgrade_ring(index_map, sample_nside -> nside) + c%indices(:, s1:s2, nind) = index_full_res(:, s1:s2) (:480-483)
__global__ __launch_bounds__(BLOCK) void k_coarse_writeback(const Model* __restrict__ Mp, int comp, int nind, int s1, int s2,
                                                            const double* __restrict__ idxmap, const int* __restrict__ r2n_f,
                                                            const int* __restrict__ n2r_c, int ratio) {
    const Model& M = *Mp;
    const int p = blockIdx.x * BLOCK + threadIdx.x;
    if (p >= M.npix) return;
    const double v = idxmap[n2r_c[r2n_f[M.pix0 + p] / ratio]];
    for (int k = s1; k <= s2; ++k) M.comp[comp].idx[((long long)nind * M.nmaps + (k - 1)) * M.npix + p] = v;
}



}  // namespace

extern "C" {

// ---- full-sky index mode / tuner / gain fit primitives ---------------------------------------------


static int fullsky_prepare_impl(dangx_ctx* ctx, int comp, int map_n, bool lazy) {
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
        if (l != comp && ((ctx->plane_nz[l] & ((1u << (s1 - 1)) | (1u << (s2 - 1)))) || ctx->desc[l].type == DANGX_TCMB || is_global_type(ctx->desc[l].type))) others |= 1u << l;
    // lazy (the sky-wide chains, dangx_sky.hip): the staging pass is deferred -- the chisq chain needs the cleaned data once, inside
    // its statistics pass (k_fullsky_stats forms them itself); any other sum (marginal rows, the Jeffreys sum, selector 0) fills
    // the buffer first (dangx_fullsky_sums).  DANGX_FULLSKY_LAZY=0: always staged (A/B).
    static const bool lazy_on = [] { const char* e = getenv("DANGX_FULLSKY_LAZY"); return !(e && e[0] == '0'); }();
    ctx->fs_others = others;
    ctx->fs_lazy = lazy && lazy_on;
    if (!ctx->fs_lazy) hipLaunchKernelGGL(k_fullsky_prepare, dim3(nblocks(ctx->hm.npix)), dim3(BLOCK), (size_t)ctx->hm.nbands * BLOCK * sizeof(double), ctx->stream, ctx->dm, comp, s1, s2, others, ctx->fs_data);
    HIPCHK(ctx, hipGetLastError());
    ctx->fs_comp = comp; ctx->fs_s1 = s1; ctx->fs_s2 = s2; ctx->fs_npc = 0;
    return 0;
}
int dangx_fullsky_prepare(dangx_ctx* ctx, int comp, int map_n) { return fullsky_prepare_impl(ctx, comp, map_n, false); }
int dx_fullsky_prepare_lazy(dangx_ctx* ctx, int comp, int map_n) { return fullsky_prepare_impl(ctx, comp, map_n, true); }

// what = 0 chisq lnL (1 value), 1 marginal (2*nb*Sp values: TNd(j,k), TNT(j,k) interleaved, j outer / k inner),
// 2 jeffreys sum (1 value).  Local (this shard's) sums; the caller all-reduces and combines.
int dangx_fullsky_sums(dangx_ctx* ctx, int what, const double* theta, double* out, int nout) {
    if (!ctx || !theta || !out) return 1;
    (void)hipSetDevice(ctx->device);
    if (ctx->fs_comp < 0) return fail(ctx, "dangx_fullsky_prepare has not been called");
    if (what < 0 || what > 3) return fail(ctx, "bad sum selector");
    if (sync_model(ctx)) return 1;
    const int Sp = ctx->fs_s2 - ctx->fs_s1 + 1;
    const int rows = (what == 1) ? 2 * ctx->hm.nbands * Sp : (what == 3) ? 3 * ctx->hm.nbands * Sp : 1;
    if (nout < rows) return fail(ctx, "output buffer too small");
    const bool coarse = ctx->fs_npc > 0;
    const unsigned nblk = nblocks(coarse ? ctx->fs_npc : ctx->hm.npix);
    if (what == 3) {   // the chisq statistics: one pass
        const long long nwp = nblk;
        const size_t ldsz = ((size_t)ctx->hm.nbands * BLOCK + (size_t)rows * (BLOCK / 64) + (size_t)MAXC * Sp * ctx->hm.nbands) * sizeof(double);
        if (ensure_partial(ctx, (long long)rows * nwp)) return 1;
        hipLaunchKernelGGL(k_fullsky_stats, dim3(nblk), dim3(BLOCK), ldsz, ctx->stream, ctx->dm, ctx->fs_comp, ctx->fs_s1, ctx->fs_s2, theta[0], theta[1],
                           coarse ? ctx->cs_data : ctx->fs_data, ctx->fs_others, (!coarse && ctx->fs_lazy) ? 1 : 0,
                           coarse ? ctx->cs_rms : (const double*)nullptr, coarse ? ctx->cs_mask : (const double*)nullptr,
                           coarse ? ctx->fs_npc : 0ll, ctx->partial);
        dx_reduce_rows_to(ctx, ctx->partial, nwp, rows, ctx->rows_out);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(out, ctx->rows_out, sizeof(double) * rows, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    }
    if (ctx->fs_lazy && !coarse) {   // the staging buffer was left unwritten (dangx_fullsky_prepare_lazy): fill it now
        hipLaunchKernelGGL(k_fullsky_prepare, dim3(nblocks(ctx->hm.npix)), dim3(BLOCK), (size_t)ctx->hm.nbands * BLOCK * sizeof(double), ctx->stream, ctx->dm, ctx->fs_comp, ctx->fs_s1, ctx->fs_s2,
                           ctx->fs_others, ctx->fs_data);
        ctx->fs_lazy = false;
    }
    if (ensure_partial(ctx, (long long)rows * nblk)) return 1;
    hipLaunchKernelGGL(k_fullsky_rows, dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, ctx->fs_comp, ctx->fs_s1, ctx->fs_s2, what,
                       theta[0], theta[1], coarse ? ctx->cs_data : ctx->fs_data, coarse ? ctx->cs_rms : (const double*)nullptr,
                       coarse ? ctx->cs_mask : (const double*)nullptr, coarse ? ctx->fs_npc : 0ll, ctx->partial);
    dx_reduce_rows_to(ctx, ctx->partial, nblk, rows, ctx->rows_out);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->rows_out, sizeof(double) * rows, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}


// ---- HEALPix index maps on the host (published algorithm; see k_udgrade) ------------------------------------------
// nest2ring: face f = ipnest / nside^2, (ix, iy) = the even / odd bits of the in-face index, ring jr counted from the
// north pole, position jp in the ring.
static void hp_nest2ring_table(int nside, std::vector<int>& n2r) {
    static const int jrll[12] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4}, jpll[12] = {1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7};
    const long long ns2 = (long long)nside * nside, npix = 12 * ns2, ncap = 2LL * nside * (nside - 1);
    n2r.resize((size_t)npix);
    for (long long ip = 0; ip < npix; ++ip) {
        const int face = (int)(ip / ns2);
        const long long ipf = ip % ns2;
        int ix = 0, iy = 0;
        for (int b = 0; b < 16; ++b) { ix |= (int)((ipf >> (2 * b)) & 1) << b; iy |= (int)((ipf >> (2 * b + 1)) & 1) << b; }
        const long long jr = (long long)jrll[face] * nside - ix - iy - 1;
        long long nr, n_before;
        int kshift;
        if (jr < nside) { nr = jr; n_before = 2 * nr * (nr - 1); kshift = 0; }
        else if (jr > 3LL * nside) { nr = 4LL * nside - jr; n_before = npix - 2 * (nr + 1) * nr; kshift = 0; }
        else { nr = nside; n_before = ncap + (jr - nside) * 4LL * nside; kshift = (int)((jr - nside) & 1); }
        long long jp = ((long long)jpll[face] * nr + ix - iy + 1 + kshift) / 2;
        if (jp > 4 * nr) jp -= 4 * nr;
        if (jp < 1) jp += 4 * nr;
        n2r[(size_t)ip] = (int)(n_before + jp - 1);
    }
}

static bool hp_valid_nside(int n) { return n >= 1 && n <= 8192 && (n & (n - 1)) == 0; }

static int hp_upload(dangx_ctx* ctx, int nside, int** n2r_dev, int** r2n_dev) {
    std::vector<int> n2r, r2n;
    hp_nest2ring_table(nside, n2r);
    r2n.resize(n2r.size());
    for (size_t p = 0; p < n2r.size(); ++p) r2n[(size_t)n2r[p]] = (int)p;
    if (*n2r_dev) { (void)hipFree(*n2r_dev); (void)hipFree(*r2n_dev); *n2r_dev = *r2n_dev = nullptr; }
    HIPCHK(ctx, hipMalloc(n2r_dev, n2r.size() * sizeof(int)));
    HIPCHK(ctx, hipMalloc(r2n_dev, n2r.size() * sizeof(int)));
    HIPCHK(ctx, hipMemcpy(*n2r_dev, n2r.data(), n2r.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(*r2n_dev, r2n.data(), n2r.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

// RING<->NEST maps of the two resolutions, cached in the context
static int hp_tables(dangx_ctx* ctx, int nside_f, int nside_c) {
    if (!hp_valid_nside(nside_f) || !hp_valid_nside(nside_c)) return fail(ctx, "nside must be a power of two in 1..8192");
    if (ctx->hp_nside != nside_f) { if (hp_upload(ctx, nside_f, &ctx->hp_n2r_f, &ctx->hp_r2n_f)) return 1; ctx->hp_nside = nside_f; }
    if (ctx->hp_cnside != nside_c) { if (hp_upload(ctx, nside_c, &ctx->hp_n2r_c, &ctx->hp_r2n_c)) return 1; ctx->hp_cnside = nside_c; }
    return 0;
}

int dangx_udgrade(dangx_ctx* ctx, int mode, const double* map_in, int nside_in, double* map_out, int nside_out) {
    if (!ctx || !map_in || !map_out) return 1;
    (void)hipSetDevice(ctx->device);
    if (mode < 0 || mode > 2) return fail(ctx, "udgrade mode must be 0 (ring), 1 (rms) or 2 (mask)");
    if (nside_in == nside_out) return fail(ctx, "udgrade: nside_in == nside_out (the reference copies the maps, src/dang_sample_mod.f90:204-207)");
    const bool degrade = nside_in > nside_out;
    if (hp_tables(ctx, degrade ? nside_in : nside_out, degrade ? nside_out : nside_in)) return 1;
    const long long npi = 12LL * nside_in * nside_in, npo = 12LL * nside_out * nside_out;
    const int r1 = degrade ? nside_in / nside_out : nside_out / nside_in;
    double *din = nullptr, *dout = nullptr;
    HIPCHK(ctx, hipMalloc(&din, sizeof(double) * npi));
    HIPCHK(ctx, hipMalloc(&dout, sizeof(double) * npo));
    HIPCHK(ctx, hipMemcpyAsync(din, map_in, sizeof(double) * npi, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_udgrade, dim3(nblocks(npo), 1), dim3(BLOCK), 0, ctx->stream, din, dout,
                       degrade ? ctx->hp_n2r_f : ctx->hp_n2r_c, degrade ? ctx->hp_r2n_c : ctx->hp_r2n_f, npi, npo, r1 * r1,
                       degrade ? 1 : 0, mode, (double)nside_out * 1.0 / nside_in, 0, 1, 1, 1);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(map_out, dout, sizeof(double) * npo, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(din); (void)hipFree(dout);
    return 0;
}

// data_raw minus every other component at full resolution (:173-196, the full-sky mode's staging kernel), degraded with
// udgrade_ring; the rms with udgrade_rms, the mask with udgrade_mask (:199-217) -> cs_data / cs_rms / cs_mask
static int coarse_stage(dangx_ctx* ctx, int comp, int map_n, int nside, int sample_nside) {
    const long long npix = ctx->dims.npix;
    if (hp_tables(ctx, nside, sample_nside)) return 1;
    const int r1 = nside / sample_nside, ratio = r1 * r1;
    // the degrade with its loads shared out over a wave (k_udgrade_wave; DANGX_UDGRADE_WAVE=0: a thread per coarse pixel).  It CAN form
    // the cleaned data at the child pixels itself (no staging pass, no staging buffer traffic: DANGX_COARSE_FUSE=1) -- measured SLOWER,
    // 24.7 against 16.8 ms per iteration at C3 / Nside 128: the other components' SEDs are then evaluated in a gather-shaped launch,
    // once per ten-plane chunk -- so the staging pass stays.
    static const bool wave_on = [] { const char* e = getenv("DANGX_UDGRADE_WAVE"); return !(e && e[0] == '0'); }();
    static const bool fuse_on = [] { const char* e = getenv("DANGX_COARSE_FUSE"); return e && e[0] == '1'; }();
    const bool by_wave = wave_on && ratio >= 4, fuse_stage = by_wave && fuse_on;
    if (fuse_stage ? fullsky_prepare_impl(ctx, comp, map_n, true) : dangx_fullsky_prepare(ctx, comp, map_n)) return 1;
    const int s1 = ctx->fs_s1, s2 = ctx->fs_s2, Sp = s2 - s1 + 1, nb = ctx->hm.nbands;
    ctx->fs_comp = -1;  // the staging buffer is ours now
    const long long npc = 12LL * sample_nside * sample_nside;
    const long long need = (long long)Sp * nb * npc;
    if (need > ctx->cs_cap) {
        for (double** b : {&ctx->cs_data, &ctx->cs_rms, &ctx->cs_mask, &ctx->cs_index}) { if (*b) (void)hipFree(*b); *b = nullptr; }
        HIPCHK(ctx, hipMalloc(&ctx->cs_data, sizeof(double) * need));
        HIPCHK(ctx, hipMalloc(&ctx->cs_rms, sizeof(double) * need));
        HIPCHK(ctx, hipMalloc(&ctx->cs_mask, sizeof(double) * npc));
        HIPCHK(ctx, hipMalloc(&ctx->cs_index, sizeof(double) * npc));
        ctx->cs_cap = need;
    }
    const dim3 gq(nblocks(npc), Sp * nb), g1(nblocks(npc), 1);
    const double scale = (double)sample_nside * 1.0 / nside;
    const dim3 gp(nblocks(npc), (Sp * nb + UDG_QC - 1) / UDG_QC);
    // (ratio 4: 16 pixels x 4 waves x 10 planes = 640 summing threads, three rounds of the block: the kernel's three slots)
    const long long per_block = (long long)UDG_G * ((ratio < 64) ? 64 / ratio : 1);
    const dim3 gw((unsigned)((npc + per_block - 1) / per_block), (Sp * nb + UDG_QW - 1) / UDG_QW);
    if (by_wave && fuse_stage)   // the cleaned data are formed at the child pixels: the staging pass above was skipped
        hipLaunchKernelGGL(k_udgrade_wave<true>, gw, dim3(BLOCK), 0, ctx->stream, ctx->dm, comp, ctx->fs_others, Sp, ctx->sig, ctx->cs_data,
                           ctx->hp_n2r_f, ctx->hp_r2n_c, npix, npc, ratio, 0, scale, 0, nb, ctx->hm.nmaps, s1, Sp * nb);
    else if (by_wave)
        hipLaunchKernelGGL(k_udgrade_wave<false>, gw, dim3(BLOCK), 0, ctx->stream, ctx->dm, comp, 0u, Sp, ctx->fs_data, ctx->cs_data, ctx->hp_n2r_f, ctx->hp_r2n_c, npix, npc,
                           ratio, 0, scale, 0, nb, ctx->hm.nmaps, s1, Sp * nb);
    else
        hipLaunchKernelGGL(k_udgrade_planes, gp, dim3(BLOCK), 0, ctx->stream, ctx->fs_data, ctx->cs_data, ctx->hp_n2r_f, ctx->hp_r2n_c, npix, npc,
                           ratio, 0, scale, 0, nb, ctx->hm.nmaps, s1, Sp * nb);
    // the degraded rms and mask: the kept copy of this plane set if the maps have not changed since, else degraded and kept
    static const bool keep_on = [] { const char* e = getenv("DANGX_COARSE_KEEP"); return !(e && e[0] == '0'); }();  // A/B switch
    dangx_ctx::CsKept* hit = nullptr;
    for (auto& kq : ctx->cs_kept)
        if (keep_on && kq.rms && kq.gen == ctx->data_gen && kq.s1 == s1 && kq.s2 == s2 && kq.nside == nside && kq.sample_nside == sample_nside) hit = &kq;
    if (hit) {
        hit->stamp = ++ctx->cs_stamp;
        HIPCHK(ctx, hipMemcpyAsync(ctx->cs_rms, hit->rms, sizeof(double) * need, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->cs_mask, hit->mask, sizeof(double) * npc, hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    if (by_wave)
        hipLaunchKernelGGL(k_udgrade_wave<false>, gw, dim3(BLOCK), 0, ctx->stream, ctx->dm, comp, 0u, Sp, ctx->rms, ctx->cs_rms, ctx->hp_n2r_f, ctx->hp_r2n_c, npix, npc,
                           ratio, 1, scale, 1, nb, ctx->hm.nmaps, s1, Sp * nb);
    else
        hipLaunchKernelGGL(k_udgrade_planes, gp, dim3(BLOCK), 0, ctx->stream, ctx->rms, ctx->cs_rms, ctx->hp_n2r_f, ctx->hp_r2n_c, npix, npc,
                           ratio, 1, scale, 1, nb, ctx->hm.nmaps, s1, Sp * nb);
    hipLaunchKernelGGL(k_udgrade, g1, dim3(BLOCK), 0, ctx->stream, ctx->mask, ctx->cs_mask, ctx->hp_n2r_f, ctx->hp_r2n_c, npix, npc,
                       ratio, 1, 2, scale, 0, nb, ctx->hm.nmaps, s1);
    HIPCHK(ctx, hipGetLastError());
    if (keep_on) {   // into the slot used longest ago
        dangx_ctx::CsKept& kq = (ctx->cs_kept[0].stamp <= ctx->cs_kept[1].stamp) ? ctx->cs_kept[0] : ctx->cs_kept[1];
        if (kq.cap < need || kq.capm < npc) {
            if (kq.rms) (void)hipFree(kq.rms);
            if (kq.mask) (void)hipFree(kq.mask);
            kq.rms = kq.mask = nullptr; kq.cap = kq.capm = 0; kq.gen = -1;
            HIPCHK(ctx, hipMalloc(&kq.rms, sizeof(double) * need));
            HIPCHK(ctx, hipMalloc(&kq.mask, sizeof(double) * npc));
            kq.cap = need; kq.capm = npc;
        }
        HIPCHK(ctx, hipMemcpyAsync(kq.rms, ctx->cs_rms, sizeof(double) * need, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(kq.mask, ctx->cs_mask, sizeof(double) * npc, hipMemcpyDeviceToDevice, ctx->stream));
        kq.s1 = s1; kq.s2 = s2; kq.nside = nside; kq.sample_nside = sample_nside; kq.gen = ctx->data_gen; kq.stamp = ++ctx->cs_stamp;
    }
    return 0;
}

// ---- coarse-Nside sampling on a PIXEL SHARD, in three phases with a sum over the shards between them (the children of
// a coarse pixel are scattered over the RING ranges).  A: every shard degrades what it holds -- per coarse pixel and
// plane the sum of its own good children and their number (data, rms^2, mask); B: with the sums of all shards the coarse
// data / rms / mask are finished, and each shard runs the chains of the coarse pixels i whose full-resolution pixel i
// it holds (the reference reads masks(i), indices(i), amplitude(i) there), leaving 0 elsewhere; C: with the summed
// coarse index map every shard writes its own pixels.  dangx_index_sample_coarse runs A, B, C through the
// dangx_set_allreduce callback; a single-process driver with several contexts calls them itself and adds the buffers.
static long long coarse_partials_len(const dangx_ctx* ctx, int Sp, long long npc) { return 2 * (2ll * Sp * ctx->hm.nbands + 1) * npc; }

static int coarse_check(dangx_ctx* ctx, int comp, int nside, int sample_nside) {
    if (check_comp(ctx, comp)) return 1;
    if (ctx->dims.npix_global != 12LL * nside * nside) return fail(ctx, "npix_global is not 12*nside^2");
    if (!(sample_nside < nside)) return fail(ctx, "sample_nside must be smaller than nside");
    if (ctx->desc[comp].type > DANGX_TCMB) return fail(ctx, "coarse-Nside sampling is built for the diffuse component types and T_cmb");
    return 0;
}

int dangx_coarse_sizes(dangx_ctx* ctx, int map_n, int sample_nside, int64_t* n_partials, int64_t* n_index) {
    if (!ctx || !n_partials || !n_index) return 1;
    int s1, s2;
    if (map_planes(ctx, map_n, s1, s2)) return 1;
    const long long npc = 12LL * sample_nside * sample_nside;
    *n_partials = coarse_partials_len(ctx, s2 - s1 + 1, npc);
    *n_index = npc + 1;
    return 0;
}

static int coarse_alloc(dangx_ctx* ctx, int Sp, long long npc) {
    const long long need = (long long)Sp * ctx->hm.nbands * npc;
    if (need > ctx->cs_cap) {
        for (double** b : {&ctx->cs_data, &ctx->cs_rms, &ctx->cs_mask, &ctx->cs_index}) { if (*b) (void)hipFree(*b); *b = nullptr; }
        HIPCHK(ctx, hipMalloc(&ctx->cs_data, sizeof(double) * need));
        HIPCHK(ctx, hipMalloc(&ctx->cs_rms, sizeof(double) * need));
        HIPCHK(ctx, hipMalloc(&ctx->cs_mask, sizeof(double) * npc));
        HIPCHK(ctx, hipMalloc(&ctx->cs_index, sizeof(double) * npc));
        ctx->cs_cap = need;
    }
    const long long np = coarse_partials_len(ctx, Sp, npc);
    if (np > ctx->cs_part_cap) {
        if (ctx->cs_part) (void)hipFree(ctx->cs_part);
        ctx->cs_part = nullptr;
        HIPCHK(ctx, hipMalloc(&ctx->cs_part, sizeof(double) * np));
        ctx->cs_part_cap = np;
    }
    return 0;
}

int dangx_coarse_partials(dangx_ctx* ctx, int comp, int map_n, int nside, int sample_nside, double* buf) {
    if (!ctx || !buf || coarse_check(ctx, comp, nside, sample_nside)) return 1;
    (void)hipSetDevice(ctx->device);
    if (hp_tables(ctx, nside, sample_nside)) return 1;
    if (dangx_fullsky_prepare(ctx, comp, map_n)) return 1;   // data_raw minus every other component, this shard's pixels
    const int s1 = ctx->fs_s1, s2 = ctx->fs_s2, Sp = s2 - s1 + 1, nb = ctx->hm.nbands;
    ctx->fs_comp = -1;
    const long long npc = 12LL * sample_nside * sample_nside, npl = ctx->dims.npix, p0 = ctx->dims.pix0;
    const int r1 = nside / sample_nside, ratio = r1 * r1;
    if (coarse_alloc(ctx, Sp, npc)) return 1;
    const long long nq = (long long)Sp * nb * npc;
    double *dt = ctx->cs_part, *dc = dt + nq, *rt = dc + nq, *rc = rt + nq, *mt = rc + nq, *mc = mt + npc;
    const dim3 gq(nblocks(npc), Sp * nb), g1(nblocks(npc), 1);
    hipLaunchKernelGGL(k_udgrade_part, gq, dim3(BLOCK), 0, ctx->stream, ctx->fs_data, dt, dc, ctx->hp_n2r_f, ctx->hp_r2n_c, p0, npl, npc,
                       ratio, 0, 0, nb, ctx->hm.nmaps, s1);
    hipLaunchKernelGGL(k_udgrade_part, gq, dim3(BLOCK), 0, ctx->stream, ctx->rms, rt, rc, ctx->hp_n2r_f, ctx->hp_r2n_c, p0, npl, npc,
                       ratio, 1, 1, nb, ctx->hm.nmaps, s1);
    hipLaunchKernelGGL(k_udgrade_part, g1, dim3(BLOCK), 0, ctx->stream, ctx->mask, mt, mc, ctx->hp_n2r_f, ctx->hp_r2n_c, p0, npl, npc,
                       ratio, 2, 0, nb, ctx->hm.nmaps, s1);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(buf, ctx->cs_part, sizeof(double) * coarse_partials_len(ctx, Sp, npc), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// finish the degraded data / rms / mask from the child sums of ALL shards (phase B's first half) -> cs_data / cs_rms / cs_mask
static int coarse_finish(dangx_ctx* ctx, int Sp, long long npc, int nside, int sample_nside, const double* partials_sum) {
    const long long nq = (long long)Sp * ctx->hm.nbands * npc;
    if (coarse_alloc(ctx, Sp, npc)) return 1;
    HIPCHK(ctx, hipMemcpyAsync(ctx->cs_part, partials_sum, sizeof(double) * coarse_partials_len(ctx, Sp, npc), hipMemcpyHostToDevice, ctx->stream));
    const double *dt = ctx->cs_part, *dc = dt + nq, *rt = dc + nq, *rc = rt + nq, *mt = rc + nq, *mc = mt + npc;
    const double scale = (double)sample_nside * 1.0 / nside;
    hipLaunchKernelGGL(k_udgrade_finish, dim3(nblocks(nq)), dim3(BLOCK), 0, ctx->stream, dt, dc, ctx->cs_data, nq, 0, scale);
    hipLaunchKernelGGL(k_udgrade_finish, dim3(nblocks(nq)), dim3(BLOCK), 0, ctx->stream, rt, rc, ctx->cs_rms, nq, 1, scale);
    hipLaunchKernelGGL(k_udgrade_finish, dim3(nblocks(npc)), dim3(BLOCK), 0, ctx->stream, mt, mc, ctx->cs_mask, npc, 2, scale);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// full-sky index mode at a coarser Nside ON A PIXEL SHARD: dangx_coarse_partials of every shard, added, then this call on
// every shard -- the degraded maps are then whole-sky on each of them and dangx_fullsky_sums adds, per shard, the coarse
// pixels i whose full-resolution pixel i the shard holds
int dangx_fullsky_finish_coarse(dangx_ctx* ctx, int comp, int map_n, int nside, int sample_nside, const double* partials_sum) {
    if (!ctx || !partials_sum || coarse_check(ctx, comp, nside, sample_nside)) return 1;
    (void)hipSetDevice(ctx->device);
    int s1, s2;
    if (map_planes(ctx, map_n, s1, s2) || sync_model(ctx)) return 1;
    const long long npc = 12LL * sample_nside * sample_nside;
    if (coarse_finish(ctx, s2 - s1 + 1, npc, nside, sample_nside, partials_sum)) return 1;
    ctx->fs_comp = comp; ctx->fs_s1 = s1; ctx->fs_s2 = s2; ctx->fs_npc = npc;
    return 0;
}

int dangx_coarse_chains(dangx_ctx* ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed, uint64_t stream,
                        int nside, int sample_nside, const double* partials_sum, double* index_out) {
    if (!ctx || !partials_sum || !index_out || coarse_check(ctx, comp, nside, sample_nside)) return 1;
    (void)hipSetDevice(ctx->device);
    const dangx_comp_desc& d = ctx->desc[comp];
    if (nind < 0 || nind >= d.nindices) return fail(ctx, "index number out of range");
    if (d.lnl_type[nind] < DANGX_LNL_CHISQ || d.lnl_type[nind] > DANGX_LNL_PRIOR) return fail(ctx, "bad lnl_type");
    if (ml_mode != DANGX_ML_SAMPLE && ml_mode != DANGX_ML_OPTIMIZE) return fail(ctx, "bad ml_mode");
    int s1, s2;
    if (map_planes(ctx, map_n, s1, s2) || sync_model(ctx)) return 1;
    const int Sp = s2 - s1 + 1, nb = ctx->hm.nbands;
    const long long npc = 12LL * sample_nside * sample_nside, nq = (long long)Sp * nb * npc;
    if (coarse_finish(ctx, Sp, npc, nside, sample_nside, partials_sum)) return 1;
    (void)nq;
    IndexArgs a{};
    a.comp = comp; a.nind = nind; a.nsample = nsample; a.ml_mode = ml_mode; a.seed = seed; a.stream = stream;
    a.s1 = s1; a.s2 = s2; a.mode = CH_GENERIC;
    HIPCHK(ctx, hipMemsetAsync(ctx->counters + 1, 0, sizeof(unsigned long long), ctx->stream));
    {
        Timed t(ctx, DANGX_K_INDEX_MH);
        hipLaunchKernelGGL(k_index_mh_coarse, dim3(nblocks(npc)), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, npc, ctx->cs_data, ctx->cs_rms,
                           ctx->cs_mask, ctx->cs_index, ctx->counters + 1);
    }
    HIPCHK(ctx, hipGetLastError());
    unsigned long long v = 0;
    HIPCHK(ctx, hipMemcpyAsync(index_out, ctx->cs_index, sizeof(double) * npc, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&v, ctx->counters + 1, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    index_out[npc] = (double)v;  // accepted proposals of this shard's chains
    return 0;
}

int dangx_coarse_writeback(dangx_ctx* ctx, int comp, int nind, int map_n, int nside, int sample_nside, const double* index_sum) {
    if (!ctx || !index_sum || coarse_check(ctx, comp, nside, sample_nside)) return 1;
    (void)hipSetDevice(ctx->device);
    if (nind < 0 || nind >= ctx->desc[comp].nindices) return fail(ctx, "index number out of range");
    int s1, s2;
    if (map_planes(ctx, map_n, s1, s2) || hp_tables(ctx, nside, sample_nside) || sync_model(ctx)) return 1;
    const long long npc = 12LL * sample_nside * sample_nside;
    const int r1 = nside / sample_nside, ratio = r1 * r1;
    if (coarse_alloc(ctx, s2 - s1 + 1, npc)) return 1;
    HIPCHK(ctx, hipMemcpyAsync(ctx->cs_index, index_sum, sizeof(double) * npc, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_coarse_writeback, dim3(nblocks(ctx->dims.npix)), dim3(BLOCK), 0, ctx->stream, ctx->dm, comp, nind, s1, s2, ctx->cs_index,
                       ctx->hp_r2n_f, ctx->hp_n2r_c, ratio);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = s1; k <= s2; ++k) {
        ctx->chi_before_valid[k - 1] = ctx->chi_after_valid[k - 1] = false;
        ctx->touched_since_amp[k - 1] = true;
        ctx->idx_const[comp] &= ~(1u << (k - 1));
    }
    idx_written(ctx, comp);
    if (map_n == -1) ctx->qu_equal[comp] |= 1u << nind;
    else if (map_n == 2 || map_n == 3) ctx->qu_equal[comp] &= ~(1u << nind);
    ctx->dirty = true;
    return 0;
}

// full-sky index mode with sample_nside /= nside (src/dang_sample_mod.f90:199-217, 229-329): the chain's sky-wide sums run
// over the degraded maps.  After this call dangx_fullsky_sums evaluates on them; the chain ends with dangx_fill_index
// (udgrade_ring of a constant coarse map is that constant everywhere, :480-483).
int dangx_fullsky_prepare_coarse(dangx_ctx* ctx, int comp, int map_n, int nside, int sample_nside) {
    if (!ctx || check_comp(ctx, comp)) return 1;
    (void)hipSetDevice(ctx->device);
    const long long npix = ctx->dims.npix;
    if (ctx->dims.pix0 != 0 || npix != 12LL * nside * nside || ctx->dims.npix_global != npix)
        return fail(ctx, "coarse-Nside sampling needs ONE whole-sky context (npix = 12*nside^2): the children of a coarse pixel are scattered over the RING ranges of a sharded run");
    if (!(sample_nside < nside)) return fail(ctx, "sample_nside must be smaller than nside (equal: dangx_fullsky_prepare)");
    if (ctx->desc[comp].type > DANGX_TCMB) return fail(ctx, "coarse-Nside sampling is built for the diffuse component types and T_cmb");
    if (coarse_stage(ctx, comp, map_n, nside, sample_nside)) return 1;
    ctx->fs_comp = comp;
    ctx->fs_npc = 12LL * sample_nside * sample_nside;
    return 0;
}

int dangx_index_sample_coarse(dangx_ctx* ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                              uint64_t stream, int nside, int sample_nside, int64_t* accepted) {
    DxRange rg_("dangx_index_sample_coarse");
    if (!ctx || check_comp(ctx, comp)) return 1;
    (void)hipSetDevice(ctx->device);
    const long long npix = ctx->dims.npix;
    if (ctx->dims.pix0 != 0 || npix != 12LL * nside * nside || ctx->dims.npix_global != npix) {
        // a pixel shard: the three phases, with the sum over the ranks between them
        if (!ctx->allreduce)
            return fail(ctx, "coarse-Nside sampling on a pixel shard needs the sum over the shards: register dangx_set_allreduce (one process per GPU), or call dangx_coarse_partials / _chains / _writeback and add the buffers (several contexts in one process)");
        int64_t np = 0, ni = 0;
        if (dangx_coarse_sizes(ctx, map_n, sample_nside, &np, &ni)) return 1;
        std::vector<double> part((size_t)np), idx((size_t)ni);
        if (dangx_coarse_partials(ctx, comp, map_n, nside, sample_nside, part.data()) || rank_sum(ctx, part.data(), np)) return 1;
        if (dangx_coarse_chains(ctx, comp, nind, map_n, nsample, ml_mode, seed, stream, nside, sample_nside, part.data(), idx.data()) ||
            rank_sum(ctx, idx.data(), ni))
            return 1;
        if (accepted) *accepted = (int64_t)idx[(size_t)ni - 1];   // all ranks' chains
        return dangx_coarse_writeback(ctx, comp, nind, map_n, nside, sample_nside, idx.data());
    }
    if (!(sample_nside < nside)) return fail(ctx, "sample_nside must be smaller than nside (equal: dangx_index_sample)");
    const dangx_comp_desc& d = ctx->desc[comp];
    if (nind < 0 || nind >= d.nindices) return fail(ctx, "index number out of range");
    if (d.type > DANGX_TCMB) return fail(ctx, "coarse-Nside sampling is built for the diffuse component types and T_cmb");
    if (d.lnl_type[nind] < DANGX_LNL_CHISQ || d.lnl_type[nind] > DANGX_LNL_PRIOR) return fail(ctx, "bad lnl_type");
    if (ml_mode != DANGX_ML_SAMPLE && ml_mode != DANGX_ML_OPTIMIZE) return fail(ctx, "bad ml_mode");
    if (coarse_stage(ctx, comp, map_n, nside, sample_nside)) return 1;
    const int s1 = ctx->fs_s1, s2 = ctx->fs_s2;
    const long long npc = 12LL * sample_nside * sample_nside;
    const int r1 = nside / sample_nside, ratio = r1 * r1;
    IndexArgs a{};
    a.comp = comp; a.nind = nind; a.nsample = nsample; a.ml_mode = ml_mode; a.seed = seed; a.stream = stream;
    a.s1 = s1; a.s2 = s2; a.mode = CH_GENERIC;
    if (accepted) HIPCHK(ctx, hipMemsetAsync(ctx->counters + 1, 0, sizeof(unsigned long long), ctx->stream));
    {
        Timed t(ctx, DANGX_K_INDEX_MH);
        hipLaunchKernelGGL(k_index_mh_coarse, dim3(nblocks(npc)), dim3(BLOCK), 0, ctx->stream, ctx->dm, a, npc, ctx->cs_data, ctx->cs_rms,
                           ctx->cs_mask, ctx->cs_index, accepted ? ctx->counters + 1 : nullptr);
    }
    hipLaunchKernelGGL(k_coarse_writeback, dim3(nblocks(npix)), dim3(BLOCK), 0, ctx->stream, ctx->dm, comp, nind, s1, s2, ctx->cs_index,
                       ctx->hp_r2n_f, ctx->hp_n2r_c, ratio);
    HIPCHK(ctx, hipGetLastError());
    for (int k = s1; k <= s2; ++k) {  // the planes changed: cached chi^2 and constant-index bookkeeping are stale
        ctx->chi_before_valid[k - 1] = ctx->chi_after_valid[k - 1] = false;
        ctx->touched_since_amp[k - 1] = true;
        ctx->idx_const[comp] &= ~(1u << (k - 1));
    }
    idx_written(ctx, comp);
    if (map_n == -1) ctx->qu_equal[comp] |= 1u << nind;
    else if (map_n == 2 || map_n == 3) ctx->qu_equal[comp] &= ~(1u << nind);
    ctx->dirty = true;
    if (accepted) {
        unsigned long long v = 0;
        HIPCHK(ctx, hipMemcpyAsync(&v, ctx->counters + 1, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *accepted = (int64_t)v;
    }
    return 0;
}



}  // extern "C"
