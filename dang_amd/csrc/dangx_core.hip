// dangx.hip -- libdangx.so: hand-written HIP (gfx950 / CDNA4) kernels for dang's
// Gibbs inner loop and the C ABI declared in include/dangx.h.
//
// Design: one thread owns one (pixel, Stokes plane) unit -- the
// reference's global CG system is block diagonal for diffuse components, so the
// amplitude phase is a single streaming pass (mixing rows -> normal equations ->
// Cholesky) and the index phase a single pass with the Metropolis chain held in
// LDS/registers.  All map arrays are pixel-major, so a wavefront's 64 lanes read 64
// consecutive doubles (512 B) per load.  Everything is fp64.
#include <dlfcn.h>

#include "dx_host.h"

DxRoctx::DxRoctx() {
    const char* e = getenv("DANGX_ROCTX");
    if (!(e && e[0] == '1')) return;
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
        void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push && pop) return;
        push = nullptr; pop = nullptr;
    }
}
const DxRoctx& dx_roctx() {
    static const DxRoctx r;
    return r;
}
const char* dx_kernel_family(int kid) {
    static const char* const names[DANGX_K_COUNT] = {"dangx:amplitude_solve", "dangx:index_sweep", "dangx:sky_chisq", "dangx:reduce",
                                                     "dangx:cg_Ax", "dangx:cg_vec", "dangx:solve+sweeps", "dangx:other"};
    return (kid >= 0 && kid < DANGX_K_COUNT) ? names[kid] : "dangx:?";
}


// ======================================================================= kernels

namespace {

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
// entries t >= ndot do not enter the partial sums (replicated global rows on the non-root ranks of a sharded run)
__global__ __launch_bounds__(BLOCK) void k_cg_vec(int mode, long long n, long long ndot, double alpha, double* __restrict__ x,
                                                  double* __restrict__ r, double* __restrict__ d,
                                                  const double* __restrict__ q, const double* __restrict__ b2,
                                                  double* __restrict__ partial) {
    const long long t = (long long)blockIdx.x * BLOCK + threadIdx.x;
    double rr = 0.0;
    if (t < n) {
        if (mode == 0) {
            const double rv = b2[t] - q[t];
            r[t] = rv; d[t] = rv; rr = (t < ndot) ? rv * rv : 0.0;
        } else if (mode == 1) {
            x[t] = x[t] + alpha * d[t];
            const double rv = r[t] - alpha * q[t];
            r[t] = rv; rr = (t < ndot) ? rv * rv : 0.0;
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

// block partials of sum(u*v)
__global__ __launch_bounds__(BLOCK) void k_dot(const double* __restrict__ u, const double* __restrict__ v, long long n,
                                               long long ndot, double* __restrict__ partial) {
    __shared__ double sh[BLOCK / 64];
    const long long t = (long long)blockIdx.x * BLOCK + threadIdx.x;
    double s = (t < n && t < ndot) ? u[t] * v[t] : 0.0;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double w = 0.0;
        for (int q = 0; q < BLOCK / 64; ++q) w += sh[q];
        partial[blockIdx.x] = w;
    }
}

// block partials of sum((b-q)^2) [row 0] and sum(b^2) [row 1] over the diffuse entries t < ndiff of unmasked units
// (the rows of masked units are zero in A and keep whatever b holds: not part of the solve)
__global__ __launch_bounds__(BLOCK) void k_resid_norm(const Model* __restrict__ Mp, const double* __restrict__ b,
                                                      const double* __restrict__ q, long long ndiff, long long SN,
                                                      double* __restrict__ partial) {
    __shared__ double sh[2][BLOCK / 64];
    const Model& M = *Mp;
    const long long t = (long long)blockIdx.x * BLOCK + threadIdx.x;
    double rr = 0.0, bb = 0.0;
    if (t < ndiff) {
        const long long u = t % SN;
        if (!is_masked(M.mask[u % M.npix])) {
            const double r = b[t] - q[t];
            rr = r * r; bb = b[t] * b[t];
        }
    }
    for (int o = 32; o > 0; o >>= 1) { rr += __shfl_down(rr, o, 64); bb += __shfl_down(bb, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = rr; sh[1][threadIdx.x >> 6] = bb; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double v = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) v += sh[threadIdx.x][w];
        partial[(long long)threadIdx.x * gridDim.x + blockIdx.x] = v;
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

// the same two stages for up to CHI_RING sweeps at once, in launch order (identical sums: same chunks, same trees)
constexpr int CHI_ROWS = 4 + DX_MAX_IDXSUM;   // rows of a stage entry: the four chi^2 sums + the index sums of a plane-set launch
struct ChiBatch {
    const double* buf[dangx_ctx::CHI_RING];
    long long nblk[dangx_ctx::CHI_RING];
    int s1[dangx_ctx::CHI_RING], s2[dangx_ctx::CHI_RING], wb[dangx_ctx::CHI_RING], ns[dangx_ctx::CHI_RING];
    int slot[dangx_ctx::CHI_RING][DX_MAX_IDXSUM];
    int n;
};
__global__ __launch_bounds__(BLOCK) void k_reduce_rows_batch(ChiBatch b, double* __restrict__ stage) {
    __shared__ double sh[BLOCK];
    const int e = blockIdx.y;
    const double* in = b.buf[e];
    const long long n = b.nblk[e];
    double* out = stage + (long long)e * CHI_ROWS * gridDim.x;
    const long long chunk = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    const int q = blockIdx.z;   // one row per block: the rows of an entry reduce side by side
    if (q >= 4 + b.ns[e]) return;
    double s = 0.0;
    for (long long t = lo + threadIdx.x; t < hi; t += BLOCK) s += in[(long long)q * n + t];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = BLOCK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(long long)q * gridDim.x + blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(BLOCK) void k_reduce_chi_batch(ChiBatch b, const double* __restrict__ stage, long long n,
                                                            double* __restrict__ cache) {
    __shared__ double sh[BLOCK];
    const int q = blockIdx.x;   // one row per block; the entries in launch order (a later launch overwrites an earlier one's slot)
    for (int e = 0; e < b.n; ++e) {
        if (q >= 4 + b.ns[e]) continue;   // (block-uniform)
        const double* partial = stage + (long long)e * CHI_ROWS * n;
        double s = 0.0;
        for (long long t = threadIdx.x; t < n; t += BLOCK) s += partial[(long long)q * n + t];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int o = BLOCK / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (q >= 4) {   // masked sum of an index map the launch swept: the same value on every plane it wrote (:465)
                for (int k = b.s1[e]; k <= b.s2[e]; ++k) cache[b.slot[e][q - 4] + (k - b.s1[e])] = sh[0];
            } else {
                const int plane = (q & 1) ? b.s2[e] : b.s1[e];
                const bool after = q >= 2;
                if (!((q & 1) && b.s1[e] == b.s2[e]) && (after || b.wb[e])) cache[(after ? 3 : 0) + plane - 1] = sh[0];
            }
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
// (4 waves/SIMD asked for: the kernel streams 2 nb maps per plane against a few SED evaluations, and left to itself the register
// allocator drifts to 130 registers = 3 waves with any small change of the SED helpers: 1.36 -> 1.70 ms per plane at C3)
__global__ __launch_bounds__(BLOCK, 4) void k_sky_chisq(const Model* __restrict__ Mp, int pol_lo, int pol_hi, double* __restrict__ sky,
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
                    if (c.type == DANGX_MONOPOLE) continue;  // sets the band offsets instead (src/dang_data_mod.f90:357-361)
                    if (amp == 0.0 && !want_maps && c.type != DANGX_TCMB && !is_global_type(c.type)) continue;
                    double t0, t1;
                    load_theta(M, c, i, k, t0, t1);
                    const Prep pr = sed_prep(c, t0, t1);
                    for (int j = 0; j < nb; ++j) lds[j * BS + tid] = lds[j * BS + tid] + comp_signal(M, c, i, k, j, amp, pr);
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
                d = d - comp_signal(M, c2, i, k, j, c2.amp[(long long)(k - 1) * npix + i], sed_prep(c2, t0, t1));
            }
            out[((long long)(k - s1) * nb + j) * npix + i] = d;
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
    const int nrows = (what == 1) ? 2 * nb * Sp : 1;
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
                        const double m = signal_of(c, amp[kk], sed_eval(M, c, j, pr));
                        const double t = (data[((long long)kk * nb + j) * npix + i] - m) / rms_at(kk, j);
                        v = v - 0.5 * (t * t);
                    }
            } else if (what == 1) {
                const int q = row >> 1, j = q / Sp, kk = q - j * Sp;  // (j outer, k inner) as the reference sums
                const double m = signal_of(c, amp[kk], sed_eval(M, c, j, pr));
                const double rms = rms_at(kk, j);
                const double TN = m / (rms * rms);
                v = (row & 1) ? TN * m : TN * data[((long long)kk * nb + j) * npix + i];
            } else if (what == 2 && !msk && c.is_synch) {
                for (int kk = 0; kk < Sp; ++kk)
                    for (int j = 0; j < nb; ++j) {
                        const double ss = signal_of(c, amp[kk], sed_eval(M, c, j, pr));
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
            if (c.type == DANGX_MONOPOLE) continue;
            sky = sky + comp_signal(M, c, i, 1, band, c.amp[i], sed_prep(c, t0, t1));
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
    for (int q = blockIdx.x; q < rows; q += gridDim.x) {  // a row is always summed by one block, in one order
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


// mask_avg / mask_sum (src/dang_util_mod.f90:186-226) of c%indices(:, map_n, nind): rows 0: sum over unmasked pixels,
// 1: their number -- what write_data prints every iteration (src/dang_data_mod.f90:716-731)
__global__ __launch_bounds__(BLOCK) void k_index_masked_sum(const Model* __restrict__ Mp, int comp, int nind, int k,
                                                            double* __restrict__ partial) {
    __shared__ double sh[2][BLOCK / 64];
    const Model& M = *Mp;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    double v = 0.0, n = 0.0;
    if (i < M.npix && !is_masked(M.mask[i])) {
        v = M.comp[comp].idx[((long long)nind * M.nmaps + (k - 1)) * M.npix + i];
        n = 1.0;
    }
    for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); n += __shfl_down(n, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = v; sh[1][threadIdx.x >> 6] = n; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += sh[threadIdx.x][w];
        partial[(long long)threadIdx.x * gridDim.x + blockIdx.x] = t;
    }
}

// the same for several (component, index, map) triples in one launch: every block walks its pixels once (grid stride), reads the
// mask once per pixel and adds that pixel to every entry's sum; rows 2e (sum) and 2e + 1 (count, the same for every e)
struct MeanList { int n; int comp[16], nind[16], k[16]; };
__global__ __launch_bounds__(BLOCK) void k_index_masked_sums(const Model* __restrict__ Mp, MeanList ml, double* __restrict__ partial) {
    __shared__ double sh[17][BLOCK / 64];
    const Model& M = *Mp;
    const double* src[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
        src[e] = (e < ml.n) ? M.comp[ml.comp[e]].idx + ((long long)ml.nind[e] * M.nmaps + (ml.k[e] - 1)) * M.npix : nullptr;
    double v[16], cnt = 0.0;
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.0;
    for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < M.npix; i += (long long)gridDim.x * BLOCK) {
        if (is_masked(M.mask[i])) continue;
        cnt += 1.0;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (e < ml.n) v[e] += src[e][i];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        if (e < ml.n) {
            for (int o = 32; o > 0; o >>= 1) v[e] += __shfl_down(v[e], o, 64);
            if ((threadIdx.x & 63) == 0) sh[e][threadIdx.x >> 6] = v[e];
        }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) sh[16][threadIdx.x >> 6] = cnt;
    __syncthreads();
    if ((int)threadIdx.x < 2 * ml.n) {
        const int e = threadIdx.x >> 1, r = (threadIdx.x & 1) ? 16 : e;
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += sh[r][w];
        partial[(long long)threadIdx.x * gridDim.x + blockIdx.x] = t;
    }
}

// plain sums over EVERY local pixel: rows 0: sum(c%indices(:, k, nind)), 1: sum(masks(:,1)) -- the starting point of the
// step-size tuner in the per-pixel branch, sample(l) = sum(c%indices(:,map_inds(1),l))/sum(mask(:,1))
// (src/dang_sample_mod.f90:344: no mask test on the indices, the mask VALUES are summed)
__global__ __launch_bounds__(BLOCK) void k_index_plain_sum(const Model* __restrict__ Mp, int comp, int nind, int k,
                                                           double* __restrict__ partial) {
    __shared__ double sh[2][BLOCK / 64];
    const Model& M = *Mp;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    double v = 0.0, m = 0.0;
    if (i < M.npix) {
        v = M.comp[comp].idx[((long long)nind * M.nmaps + (k - 1)) * M.npix + i];
        m = M.mask[i];
    }
    for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); m += __shfl_down(m, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = v; sh[1][threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; ++w) t += sh[threadIdx.x][w];
        partial[(long long)threadIdx.x * gridDim.x + blockIdx.x] = t;
    }
}

// convert_maps, src/dang_data_mod.f90:429-463: sig_map(:,:,j) and rms_map(:,:,j) times conversion(j), on the resident maps
__global__ __launch_bounds__(BLOCK) void k_scale_band(double* __restrict__ sig, double* __restrict__ rms, long long n, double f) {
    const long long t = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (t < n) { sig[t] = sig[t] * f; rms[t] = rms[t] * f; }
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

// flags bit q is set when index map q of an [nind][3][npix] array differs between the Q and the U plane somewhere
__global__ __launch_bounds__(BLOCK) void k_qu_differ(const double* __restrict__ idx, long long npix, int nind, unsigned* __restrict__ flags) {
    for (int q = 0; q < nind; ++q) {
        const double* mq = idx + ((long long)q * 3 + 1) * npix;
        bool diff = false;
        for (long long t = (long long)blockIdx.x * BLOCK + threadIdx.x; t < npix; t += (long long)gridDim.x * BLOCK)
            diff = diff || (mq[t] != mq[npix + t]);
        if (__ballot(diff) && (threadIdx.x & 63) == 0) atomicOr(flags, 1u << q);
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
    out[i] = comp_sed(M, c, i, map_n, band, sed_prep(c, t0, t1));
}

}  // namespace

// ======================================================================= host side

void dx_reduce_rows_to(dangx_ctx* ctx, const double* partial, unsigned nblk, int rows, double* out_dev) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(k_reduce_rows_final, dim3(rows < 1024 ? rows : 1024), dim3(BLOCK), 0, ctx->stream, partial, (long long)nblk, rows, out_dev);
}

namespace {


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

// compute_bnu_prime_RJ / compute_bnu_prime, src/dang_bp_mod.f90:160-179 (nu in Hz)
double host_bnu_prime_RJ(double nu) { return 2.0 * K_B * std::pow(nu, 2.0) / std::pow(C_LIGHT, 2.0); }
double host_bnu_prime(double nu, double tcmb) {
    const double y = H_PLANCK * nu / (K_B * tcmb);
    return (2.0 * H_PLANCK * (nu * nu * nu)) / (std::pow(C_LIGHT, 2.0) * (std::exp(y) - 1)) * (std::exp(y) / (std::exp(y) - 1)) * H_PLANCK * nu /
           (K_B * (tcmb * tcmb));
}
// a2f(bp) [MJy/sr / uK_RJ], src/dang_bp_mod.f90:181-209.  `sum*1e14` multiplies by a SINGLE-precision literal: 1e14 is
// not representable in real(4), the factor is 100000000376832 -- kept, it is what the reference's maps are scaled by
double host_a2f(const dangx_ctx* ctx, int j) {
    const Band& b = ctx->hm.band[j];
    double sum = 0.0;
    if (b.n == 0) {
        sum = (b.nu_c > 1e7f) ? host_bnu_prime_RJ(b.nu_c) : host_bnu_prime_RJ(b.nu_c * 1e9);
    } else {
        for (int i = 0; i < b.n; ++i) {
            const double nu = ctx->bp_nu0[b.off + i], tau = ctx->bp_tau0[b.off + i];
            if (nu == 0.0) continue;
            sum = sum + tau * ((nu > 1e7f) ? host_bnu_prime_RJ(nu) : host_bnu_prime_RJ(nu * 1e9));
        }
    }
    return sum * (double)1e14f;
}
// f2t(bp) [uK_cmb / MJy sr-1], src/dang_bp_mod.f90:245-274
double host_f2t(const dangx_ctx* ctx, int j) {
    const Band& b = ctx->hm.band[j];
    const double T = ctx->hm.tcmb;
    double sum = 0.0;
    if (b.n == 0) {
        sum = 1.0 / ((b.nu_c > 1e7f) ? host_bnu_prime(b.nu_c, T) : host_bnu_prime(b.nu_c * 1e9, T)) * 1.0e-14;
    } else {
        for (int i = 0; i < b.n; ++i) {
            const double nu = ctx->bp_nu0[b.off + i], tau = ctx->bp_tau0[b.off + i];
            if (nu == 0.0) continue;
            sum = sum + tau / ((nu > 1e7f) ? host_bnu_prime(nu, T) : host_bnu_prime(nu * 1e9, T)) * 1.0e-14;
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
        const double z = mbb_z(th1);
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

int ensure_state(dangx_ctx* ctx, int comp);

int sync_model(dangx_ctx* ctx) {
    if (!ctx->dirty) return 0;
    Model& M = ctx->hm;
    for (int j = 0; j < M.nbands; ++j)
        if (!ctx->band_set[j]) return fail(ctx, "band " + std::to_string(j) + " not set");
    for (int l = 0; l < M.ncomp; ++l)
        if (!ctx->comp_set[l]) return fail(ctx, "component " + std::to_string(l) + " not set");
    if (!ctx->sig || !ctx->rms || !ctx->mask) return fail(ctx, "map data not uploaded");
    for (int l = 0; l < M.ncomp; ++l)
        if (ensure_state(ctx, l)) return 1;
    M.sig = ctx->sig; M.rms = ctx->rms; M.mask = ctx->mask;
    M.all_delta = 1;
    for (int j = 0; j < M.nbands; ++j) if (M.band[j].n != 0) M.all_delta = 0;
    double nu_hi = 0.0;
    for (int j = 0; j < M.nbands; ++j) nu_hi = std::max(nu_hi, M.band[j].nu_c);
    M.mbb_batch_z = 5.0 * nu_hi / 700.0;
    for (int l = 0; l < M.ncomp; ++l)
        if (ctx->desc[l].type == DANGX_TCMB || is_global_type(ctx->desc[l].type)) M.all_delta = 0;  // generic paths only
    const size_t nbp = ctx->bp_nu0.size();
    if (nbp && ctx->bp_dirty) {
        if (ctx->d_bp_nu0) {
            (void)hipFree(ctx->d_bp_nu0); (void)hipFree(ctx->d_bp_tau0); (void)hipFree(ctx->d_bp_lnr);
            ctx->d_bp_nu0 = ctx->d_bp_tau0 = ctx->d_bp_lnr = nullptr;
        }
        const size_t nbytes = nbp * sizeof(double);
        HIPCHK(ctx, hipMalloc(&ctx->d_bp_nu0, nbytes));
        HIPCHK(ctx, hipMalloc(&ctx->d_bp_tau0, nbytes));
        HIPCHK(ctx, hipMalloc(&ctx->d_bp_lnr, nbytes * M.ncomp));
        // samples with nu0 == 0 are skipped by the reference: on the device they carry tau = 0 and nu = 1 GHz
        std::vector<double> nue(ctx->bp_nu0), taue(ctx->bp_tau0);
        for (size_t q = 0; q < nbp; ++q)
            if (nue[q] == 0.0) { nue[q] = 1.0e9; taue[q] = 0.0; }
        HIPCHK(ctx, hipMemcpy(ctx->d_bp_nu0, nue.data(), nbytes, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(ctx->d_bp_tau0, taue.data(), nbytes, hipMemcpyHostToDevice));
        std::vector<double> lnr(nbp * M.ncomp, 0.0);  // (nu/nu_ref)**beta = exp(beta*log(nu/nu_ref)): the log once, here
        for (int l = 0; l < M.ncomp; ++l)
            for (size_t q = 0; q < nbp; ++q)
                if (ctx->bp_nu0[q] != 0.0) lnr[l * nbp + q] = std::log(ctx->bp_nu0[q] / ctx->desc[l].nu_ref);
        HIPCHK(ctx, hipMemcpy(ctx->d_bp_lnr, lnr.data(), nbytes * M.ncomp, hipMemcpyHostToDevice));
        ctx->bp_dirty = false;
    }
    M.bp_nu0 = ctx->d_bp_nu0; M.bp_tau0 = ctx->d_bp_tau0;
    for (int l = 0; l < M.ncomp; ++l) {
        Comp& c = M.comp[l];
        const dangx_comp_desc& d = ctx->desc[l];
        c.type = d.type; c.nind = d.nindices; c.group = d.cg_group; c.sample_amp = d.sample_amplitude;
        c.is_synch = d.is_synch; c.nu_ref = d.nu_ref;
        c.amp = ctx->amp[l]; c.idx = ctx->idx[l];
        c.tmpl = ctx->tmpl[l]; c.corr_mask = ctx->corr_mask[l]; c.nfit = ctx->nfit[l];
        c.bp_lnr = ctx->d_bp_lnr ? ctx->d_bp_lnr + (size_t)l * nbp : nullptr;
        for (int k = 0; k < 3; ++k)
            for (int j = 0; j < MAXB; ++j) c.tamp[k][j] = ctx->tamp[l][k][j];
        if (is_global_type(c.type) && !c.tmpl) return fail(ctx, "global-amplitude component without a template map (dangx_set_template)");
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
        if (c.type >= DANGX_POWERLAW && c.type <= DANGX_CMB) {  // the diffuse types; bandpass bands: the tau0-weighted sum
            const unsigned cp = (c.nind == 0) ? 7u : ctx->idx_const[l];
            for (int k = 0; k < M.nmaps; ++k)
                if ((cp >> k) & 1) {
                    c.const_planes |= 1 << k;
                    const double t0 = ctx->idx_val[l][k][0], t1 = ctx->idx_val[l][k][1];
                    for (int j = 0; j < M.nbands; ++j) {
                        const Band& b = M.band[j];
                        if (b.n == 0 || c.type == DANGX_CMB) { c.csed[k][j] = host_sed(c, b.nu_c, c.cst[j], t0, t1); continue; }
                        double sum = 0.0;
                        for (int q = 0; q < b.n; ++q)
                            if (ctx->bp_nu0[b.off + q] != 0.0) sum = sum + ctx->bp_tau0[b.off + q] * host_sed(c, ctx->bp_nu0[b.off + q], 0.0, t0, t1);
                        c.csed[k][j] = sum;
                    }
                }
        }
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->dm, &ctx->hm, sizeof(Model), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->dirty = false;
    return 0;
}

int prof_collect(dangx_ctx* ctx) {
    for (auto& e : ctx->events) {
        float ms = 0.f;
        HIPCHK(ctx, hipEventSynchronize(e.b));
        HIPCHK(ctx, hipEventElapsedTime(&ms, e.a, e.b));
        ctx->prof_ms[e.kid] += ms;
        ctx->prof_n[e.kid] += 1;
        ctx->prof_ms_pl[e.kid][e.planes] += ms;
        ctx->prof_n_pl[e.kid][e.planes] += 1;
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    ctx->events.clear();
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


constexpr int CHI_RSTAGE = 128;  // blocks of the first reduction stage of the sweeps' chi^2 partials

// reduce every pending sweep's block partials into chi_cache (two launches for all of them, in launch order)
int chi_flush(dangx_ctx* ctx) {
    if (ctx->chi_npend == 0) return 0;
    (void)hipSetDevice(ctx->device);  // the caller may have been working on another context's device
    ChiBatch b;
    b.n = ctx->chi_npend;
    for (int e = 0; e < b.n; ++e) {
        const auto& p = ctx->chi_pend[e];
        b.buf[e] = p.buf; b.nblk[e] = p.nblk; b.s1[e] = p.s1; b.s2[e] = p.s2; b.wb[e] = p.wb; b.ns[e] = p.ns;
        for (int q = 0; q < DX_MAX_IDXSUM; ++q) b.slot[e][q] = p.slot[q];
    }
    for (int e = b.n; e < dangx_ctx::CHI_RING; ++e) {
        b.buf[e] = nullptr; b.nblk[e] = 0; b.s1[e] = b.s2[e] = 1; b.wb[e] = 0; b.ns[e] = 0;
        for (int q = 0; q < DX_MAX_IDXSUM; ++q) b.slot[e][q] = 0;
    }
    {
        Timed t(ctx, DANGX_K_REDUCE);
        int rows = 4;
        for (int e = 0; e < b.n; ++e) rows = std::max(rows, 4 + b.ns[e]);
        hipLaunchKernelGGL(k_reduce_rows_batch, dim3(CHI_RSTAGE, b.n, rows), dim3(BLOCK), 0, ctx->stream, b, ctx->chi_stage);
        hipLaunchKernelGGL(k_reduce_chi_batch, dim3(rows), dim3(BLOCK), 0, ctx->stream, b, ctx->chi_stage, (long long)CHI_RSTAGE, ctx->chi_cache);
    }
    ctx->chi_npend = 0;
    HIPCHK(ctx, hipGetLastError());
    return 0;
}
// the buffer the next sweep writes its chi^2 block partials to ([4][nblk]); flushes first when the ring is full
int chi_next(dangx_ctx* ctx, long long nblk, double** buf) {
    if (ctx->chi_npend == dangx_ctx::CHI_RING && chi_flush(ctx)) return 1;
    auto& p = ctx->chi_pend[ctx->chi_npend];
    if (p.cap < CHI_ROWS * nblk) {
        if (p.buf) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(p.buf); p.buf = nullptr; p.cap = 0; }
        HIPCHK(ctx, hipMalloc(&p.buf, sizeof(double) * (size_t)(CHI_ROWS * nblk)));
        p.cap = CHI_ROWS * nblk;
    }
    p.ns = 0;
    *buf = p.buf;
    return 0;
}

int make_group(dangx_ctx* ctx, int group, int flag, GroupArgs& a) {
    if (flag != DANGX_FLAG_T && flag != DANGX_FLAG_Q && flag != DANGX_FLAG_U && flag != DANGX_FLAG_QU)
        return fail(ctx, "flag must be exactly one of T(1), Q(2), U(4), Q+U(8)");
    if (ctx->dims.nmaps < 3 && flag != DANGX_FLAG_T) return fail(ctx, "polarisation flag needs nmaps == 3");
    std::memset(&a, 0, sizeof(a));
    a.flag = flag;
    for (int l = 0; l < ctx->hm.ncomp; ++l) {
        const dangx_comp_desc& d = ctx->desc[l];
        if (d.type == DANGX_TEMPLATE || d.type == DANGX_MONOPOLE) a.uc[a.nuc++] = l;
        if (d.cg_group == group && d.sample_amplitude && is_global_type(d.type)) {
            if (a.nt >= MAXT) return fail(ctx, "too many global-amplitude components in CG group");
            if (d.type != DANGX_TEMPLATE && flag != DANGX_FLAG_T)
                return fail(ctx, "hi_fit / monopole components are fitted on plane 1: use CG_POLTYPE T for their group");
            a.tc[a.nt] = l; a.trow[a.nt] = a.nglob; a.nglob += ctx->nfit[l]; ++a.nt;
        } else if (d.cg_group == group && d.sample_amplitude) {
            if (a.nt > 0) return fail(ctx, "diffuse components must precede the global-amplitude ones in a CG group");
            if (a.ng >= MAXG) return fail(ctx, "too many components in CG group");
            a.gc[a.ng++] = l;
        } else {
            unsigned planes = 0;  // planes this (group, flag) works on
            for (int pl = 0; pl < flag_planes_h(flag); ++pl)
                planes |= 1u << (((flag & DANGX_FLAG_QU) ? 2 + pl : (flag & DANGX_FLAG_T) ? 1 : (flag & DANGX_FLAG_Q) ? 2 : 3) - 1);
            if ((ctx->plane_nz[l] & planes) || ctx->desc[l].type == DANGX_TCMB || is_global_type(ctx->desc[l].type)) a.oc[a.no++] = l;  // all-zero plane: 0*sed, skipped
        }
    }
    if (a.ng + a.nt == 0) return fail(ctx, "Woah there, number of CG components = 0 for CG group " + std::to_string(group));
    return 0;
}

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

// global rows of x <-> c%template_amplitudes (initialize_x :1244-1279, unpack_amplitudes :1355-1393)
void globals_to_x(dangx_ctx* ctx, const GroupArgs& a, std::vector<double>& xg) {
    xg.assign(std::max(a.nglob, 1), 0.0);
    const int k = (a.flag & DANGX_FLAG_QU) ? 2 : (a.flag & DANGX_FLAG_T) ? 1 : (a.flag & DANGX_FLAG_Q) ? 2 : 3;
    for (int t = 0; t < a.nt; ++t) {
        const int l = a.tc[t];
        int lf = 0;
        for (int j = 0; j < ctx->hm.nbands && lf < ctx->nfit[l]; ++j)
            if ((ctx->corr_mask[l] >> j) & 1) xg[a.trow[t] + lf++] = ctx->tamp[l][k - 1][j];
    }
}
void x_to_globals(dangx_ctx* ctx, const GroupArgs& a, const std::vector<double>& xg) {
    const int k = (a.flag & DANGX_FLAG_QU) ? 2 : (a.flag & DANGX_FLAG_T) ? 1 : (a.flag & DANGX_FLAG_Q) ? 2 : 3;
    for (int t = 0; t < a.nt; ++t) {
        const int l = a.tc[t];
        int lf = 0;
        for (int j = 0; j < ctx->hm.nbands && lf < ctx->nfit[l]; ++j)
            if ((ctx->corr_mask[l] >> j) & 1) {
                const double v = xg[a.trow[t] + lf++];
                if (ctx->desc[l].type == DANGX_TEMPLATE && (a.flag & DANGX_FLAG_QU)) {  // :1380-1382 one amplitude for Q and U
                    ctx->tamp[l][1][j] = v; ctx->tamp[l][2][j] = v;
                } else {
                    ctx->tamp[l][k - 1][j] = v;
                }
            }
        if (ctx->desc[l].type == DANGX_MONOPOLE)  // update_sky_model: self%offset = c%template_amplitudes(:,1), src/dang_data_mod.f90:357-361
            for (int j = 0; j < ctx->hm.nbands; ++j) ctx->hm.offset[j] = ctx->tamp[l][0][j];
    }
    ctx->dirty = true;
}

// sum(a*b) over n entries -> host (deterministic two-stage reduction)
int device_dot(dangx_ctx* ctx, const double* u, const double* v, long long n, long long ndot, double* out) {
    const unsigned nblk = nblocks(n);
    if (ensure_partial(ctx, nblk)) return 1;
    hipLaunchKernelGGL(k_dot, dim3(nblk), dim3(BLOCK), 0, ctx->stream, u, v, n, ndot, ctx->partial);
    return reduce_to_host(ctx, nblk, out);
}

// sum of host doubles over the ranks of a pixel-sharded run (dangx_set_allreduce); identity on a single rank
int rank_sum(dangx_ctx* ctx, double* buf, int64_t n) {
    if (!ctx->allreduce || n <= 0) return 0;
    if (ctx->allreduce(ctx->allreduce_user, buf, n)) return fail(ctx, "the all-reduce callback reported an error");
    return 0;
}
// the global rows at the tail of a device vector hold this rank's sums: make them the sums over all ranks
int rank_sum_rows(dangx_ctx* ctx, double* tail_dev, int rows) {
    if (!ctx->allreduce || rows <= 0) return 0;
    std::vector<double> h(rows);
    HIPCHK(ctx, hipMemcpyAsync(h.data(), tail_dev, sizeof(double) * rows, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (rank_sum(ctx, h.data(), rows)) return 1;
    HIPCHK(ctx, hipMemcpyAsync(tail_dev, h.data(), sizeof(double) * rows, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// cg_search on the device, src/dang_cg_mod.f90:179-324.  work[0]=x, [1]=r, [2]=d, [3]=q, [4]=b2, [5]=eta/b.
// Groups with global-amplitude members use the mixed kernels; their vectors are [diffuse | global rows].
int device_cg(dangx_ctx* ctx, const GroupArgs& a, int i_max, double converge, int* iters) {
    const long long SN = (long long)flag_planes_h(a.flag) * ctx->hm.npix;
    const long long n = SN * a.ng + a.nglob;
    const bool mixed = a.nt > 0;
    // pixel-sharded run: the diffuse entries are this rank's, the global rows are replicated (and summed over ranks
    // wherever an operator produces them); dot products count the global rows on the root rank only
    const long long ndot = ctx->is_root ? n : SN * a.ng;
    if (ensure_work(ctx, n)) return 1;
    if (ensure_partial(ctx, nblocks(n))) return 1;
    double *x = ctx->work[0], *r = ctx->work[1], *d = ctx->work[2], *q = ctx->work[3], *b2 = ctx->work[4], *tmp = ctx->work[5];
    hipStream_t st = ctx->stream;
    auto Ax = [&](const double* in, double* out) -> int {
        if (mixed ? dx_launch_Ax_mixed(ctx, a, SN, in, out) : dx_launch_Ax(ctx, a, SN, in, out, nullptr)) return 1;
        return mixed ? rank_sum_rows(ctx, out + SN * a.ng, a.nglob) : 0;
    };
    // b = compute_rhs
    if (mixed ? dx_launch_rhs_mixed(ctx, a, SN, tmp) : dx_launch_rhs(ctx, a, SN, tmp)) return 1;
    if (mixed && rank_sum_rows(ctx, tmp + SN * a.ng, a.nglob)) return 1;
    if (a.ml_mode == DANGX_ML_SAMPLE) {  // b2 = b + compute_sample_vector(eta)
        hipLaunchKernelGGL(k_draw_eta, dim3(nblocks(SN)), dim3(BLOCK), 0, st, ctx->dm, a, r);
        if (mixed ? dx_launch_sv_mixed(ctx, a, SN, r, q) : dx_launch_sample_vector(ctx, a, SN, r, q)) return 1;
        if (mixed && rank_sum_rows(ctx, q + SN * a.ng, a.nglob)) return 1;
        hipLaunchKernelGGL(k_cg_vec, dim3(nblocks(n)), dim3(BLOCK), 0, st, 3, n, n, 0.0, b2, nullptr, nullptr, q, tmp, nullptr);
    } else {
        HIPCHK(ctx, hipMemcpyAsync(b2, tmp, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
    }
    // x0 = current amplitudes (the reference keeps self%x; identical as amplitudes only change via unpack)
    if (a.ng) hipLaunchKernelGGL(k_pack, dim3(nblocks(SN)), dim3(BLOCK), 0, st, ctx->dm, a, x, 0);
    std::vector<double> xg;
    if (mixed) {
        globals_to_x(ctx, a, xg);
        HIPCHK(ctx, hipMemcpyAsync(x + SN * a.ng, xg.data(), sizeof(double) * a.nglob, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    if (Ax(x, q)) return 1;
    hipLaunchKernelGGL(k_cg_vec, dim3(nblocks(n)), dim3(BLOCK), 0, st, 0, n, ndot, 0.0, x, r, d, q, b2, ctx->partial);
    double delta_new = 0.0, delta_old, dq = 0.0;
    if (reduce_to_host(ctx, nblocks(n), &delta_new) || rank_sum(ctx, &delta_new, 1)) return 1;
    int i = 1;
    while (i < i_max && delta_new > converge) {
        if (Ax(d, q)) return 1;
        if (device_dot(ctx, d, q, n, ndot, &dq) || rank_sum(ctx, &dq, 1)) return 1;
        const double alpha = delta_new / dq;
        if (ensure_partial(ctx, nblocks(n))) return 1;
        {
            Timed t(ctx, DANGX_K_CG_VEC);
            hipLaunchKernelGGL(k_cg_vec, dim3(nblocks(n)), dim3(BLOCK), 0, st, 1, n, ndot, alpha, x, r, d, q, b2, ctx->partial);
        }
        delta_old = delta_new;
        if (reduce_to_host(ctx, nblocks(n), &delta_new) || rank_sum(ctx, &delta_new, 1)) return 1;
        const double beta = delta_new / delta_old;
        {
            Timed t(ctx, DANGX_K_CG_VEC);
            hipLaunchKernelGGL(k_cg_vec, dim3(nblocks(n)), dim3(BLOCK), 0, st, 2, n, n, beta, x, r, d, q, b2, nullptr);
        }
        i = i + 1;
    }
    if (a.ng) hipLaunchKernelGGL(k_pack, dim3(nblocks(SN)), dim3(BLOCK), 0, st, ctx->dm, a, x, 1);
    if (mixed) {
        HIPCHK(ctx, hipMemcpyAsync(xg.data(), x + SN * a.ng, sizeof(double) * a.nglob, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        x_to_globals(ctx, a, xg);
    }
    if (iters) *iters = i;
    return 0;
}

// Direct solve of a group with global-amplitude members: eliminate every (pixel, plane) block on the device
// (pass 1), solve the nglob x nglob Schur system on the host, back-substitute per unit (pass 2).  The linear
// system is the one cg_search iterates on (compute_rhs / compute_Ax / compute_sample_vector,
// src/dang_cg_mod.f90:326-1096), quirks included; the answer is its exact solution instead of the iterate at i_max.
// cs[0..nc): the contexts of this process that share the sky (shard order; nc = 1: dangx_amp_sample); as[r], SNs[r]: the
// group as context r sees it.  Row sums are added over the contexts in shard order, then over the ranks (cs[0]'s callback);
// the small system is solved once and every context gets the same global amplitudes.
int device_schur(dangx_ctx* const* cs, int nc, const GroupArgs* as, const long long* SNs, int64_t* n_not_spd, int* nullity) {
    dangx_ctx* ctx = cs[0];
    const GroupArgs& a = as[0];
    const int R = a.nglob, nb = ctx->hm.nbands;
    auto each = [&](auto&& fn) -> int {  // fn(context, its group, its SN) on every context; the first error is the call's
        for (int r = 0; r < nc; ++r)
            if (fn(cs[r], as[r], SNs[r])) { if (cs[r] != ctx) ctx->err = cs[r]->err; return 1; }
        return 0;
    };
    // rows_dev of every context -> host, added in shard order, then over the ranks
    auto gather = [&](std::vector<double>& rows, int n) -> int {
        std::vector<double> part((size_t)n);
        std::fill(rows.begin(), rows.end(), 0.0);
        for (int r = 0; r < nc; ++r) {
            (void)hipSetDevice(cs[r]->device);
            HIPCHK(ctx, hipMemcpyAsync(part.data(), cs[r]->work[0], sizeof(double) * n, hipMemcpyDeviceToHost, cs[r]->stream));
            HIPCHK(ctx, hipStreamSynchronize(cs[r]->stream));
            for (int q = 0; q < n; ++q) rows[q] += part[q];
        }
        return rank_sum(ctx, rows.data(), n);
    };
    auto set_globals = [&](const std::vector<double>& g) -> int {  // new global amplitudes + back-substitution everywhere
        return each([&](dangx_ctx* c, const GroupArgs& ga, long long SN) -> int {
            (void)hipSetDevice(c->device);
            x_to_globals(c, ga, g);
            return sync_model(c) || dx_launch_schur_pass2(c, ga, SN);
        });
    };
    if (R > DX_MAX_ROWS) return fail(ctx, "more than 32 global amplitudes in one CG group: use DANGX_SOLVER_CG");
    SchurArgs sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.nrows = R;
    for (int j = 0; j < MAXB; ++j) sa.bslot[j] = -1;
    for (int t = 0; t < a.nt; ++t) {
        const int l = a.tc[t];
        int lf = 0;
        for (int j = 0; j < nb && lf < ctx->nfit[l]; ++j)
            if ((ctx->corr_mask[l] >> j) & 1) {
                const int r = a.trow[t] + lf++;
                sa.rt[r] = (unsigned char)t; sa.rj[r] = (unsigned char)j; sa.ftarget[r] = -1;
            }
    }
    for (int j = 0; j < nb; ++j)
        for (int r = 0; r < R; ++r)
            if (sa.rj[r] == j && sa.bslot[j] < 0) sa.bslot[j] = (signed char)sa.nslots++;
    int lrun = 0;  // the running row counter of compute_sample_vector (:970, :1057, :1071, :1094)
    for (int j = 0; j < nb; ++j)
        for (int t = 0; t < a.nt; ++t) {
            const unsigned m = ctx->corr_mask[a.tc[t]];
            if ((m >> j) & 1) {
                int lt = 0;
                for (int jj = 0; jj < j; ++jj) lt += (m >> jj) & 1;
                if (lt < ctx->nfit[a.tc[t]] && lrun < R) sa.ftarget[a.trow[t] + lt] = (signed char)lrun;
                ++lrun;
            }
        }
    const int nrows = R * R + 3 * R;
    if (each([&](dangx_ctx* c, const GroupArgs& ga, long long SN) -> int {  // enqueued on every device before any result is awaited
            (void)hipSetDevice(c->device);
            return ensure_work(c, std::max<long long>(nrows, 1)) || dx_launch_schur_pass1(c, ga, sa, SN, c->work[0]);
        }))
        return 1;
    std::vector<double> rows(nrows);
    if (gather(rows, nrows)) return 1;  // every context / rank then solves the same small system
    unsigned long long bad_all = 0;
    for (int r = 0; r < nc; ++r) {
        unsigned long long bad = 0;
        (void)hipSetDevice(cs[r]->device);
        HIPCHK(ctx, hipMemcpyAsync(&bad, cs[r]->counters, sizeof(bad), hipMemcpyDeviceToHost, cs[r]->stream));
        HIPCHK(ctx, hipStreamSynchronize(cs[r]->stream));
        bad_all += bad;
    }
    if (n_not_spd) *n_not_spd = (int64_t)bad_all;
    // S g = t (+ fluctuation sums).  S is not symmetric when a monopole is fitted (its row weight is 1, :857), so:
    // Gaussian elimination, on the system equilibrated by G's diagonal (row amplitudes span ~1e-6 for hi_fit to ~1e2),
    // for the CORRECTION to the current amplitudes, S d = t - S g0, with complete pivoting.  A remaining pivot at
    // rounding level (1e-10 in units of G's diagonal, where 1 = nothing absorbed by the diffuse members) means the
    // global rows are degenerate with the diffuse members -- a template fitted at every band beside a pixel-independent
    // SED such as the CMB, or spatially constant index maps (the state a run starts from).  The normal equations stay
    // consistent; the free directions keep their current value (d = 0 there), which is what the reference's CG does
    // with them too (a Krylov iterate never moves along the null space).  nullity is reported through cg_iters.
    std::vector<double> S(rows.begin(), rows.begin() + (size_t)R * R), t(rows.begin() + (size_t)R * R, rows.begin() + (size_t)R * R + R);
    std::vector<double> fl(R, 0.0);  // fluctuation term of every row, through the running-counter mapping
    if (a.ml_mode == DANGX_ML_SAMPLE)
        for (int r = 0; r < R; ++r)
            if (sa.ftarget[r] >= 0) fl[sa.ftarget[r]] += rows[(size_t)R * R + R + r];
    for (int r = 0; r < R; ++r) t[r] += fl[r];
    std::vector<double> g0;
    globals_to_x(ctx, a, g0);
    std::vector<double> sc(R);
    for (int r = 0; r < R; ++r) {
        const double dg = std::fabs(rows[(size_t)R * R + 2 * R + r]);  // G[r][r] = sum_u w_r s_r / sigma^2, before elimination
        if (!(dg > 0.0) || !std::isfinite(dg))
            return fail(ctx, "global amplitude row " + std::to_string(r) + " of the CG group has no support (template zero or fully masked)");
        sc[r] = 1.0 / std::sqrt(dg);
    }
    for (int r = 0; r < R; ++r) {
        double v = t[r];
        for (int k = 0; k < R; ++k) v -= S[(size_t)r * R + k] * g0[k];
        t[r] = v;  // residual of the current amplitudes, as pass 1 sees it
    }
    for (int r = 0; r < R; ++r)
        for (int k = 0; k < R; ++k) S[(size_t)r * R + k] *= sc[r] * sc[k];
    // LU with complete pivoting, multipliers kept: rp / cp are the row / column permutations, rank the number of pivots
    std::vector<int> rp(R), cp(R);
    for (int c = 0; c < R; ++c) rp[c] = cp[c] = c;
    int rank = 0;
    for (int c = 0; c < R; ++c) {
        int pr = c, pc = c;
        double best = -1.0;
        for (int r = c; r < R; ++r)
            for (int k = c; k < R; ++k) {
                const double v = std::fabs(S[(size_t)r * R + k]);
                if (!std::isfinite(v)) return fail(ctx, "non-finite entry in the system of the global amplitudes");
                if (v > best) { best = v; pr = r; pc = k; }
            }
        if (!(best > 1e-10)) break;
        if (pr != c) {
            for (int k = 0; k < R; ++k) std::swap(S[(size_t)pr * R + k], S[(size_t)c * R + k]);
            std::swap(rp[pr], rp[c]);
        }
        if (pc != c) {
            for (int r = 0; r < R; ++r) std::swap(S[(size_t)r * R + pc], S[(size_t)r * R + c]);
            std::swap(cp[pc], cp[c]);
        }
        for (int r = c + 1; r < R; ++r) {
            const double f = S[(size_t)r * R + c] / S[(size_t)c * R + c];
            S[(size_t)r * R + c] = f;  // L below the diagonal
            for (int k = c + 1; k < R; ++k) S[(size_t)r * R + k] -= f * S[(size_t)c * R + k];
        }
        ++rank;
    }
    // d = correction of the global amplitudes for a residual `res` of the (unscaled) global rows; free directions get 0
    auto lu_solve = [&](const std::vector<double>& res, std::vector<double>& d) {
        std::vector<double> y(R);
        for (int r = 0; r < R; ++r) y[r] = res[rp[r]] * sc[rp[r]];
        for (int r = 0; r < R; ++r)
            for (int k = 0; k < std::min(r, rank); ++k) y[r] -= S[(size_t)r * R + k] * y[k];
        std::vector<double> z(R, 0.0);
        for (int r = rank - 1; r >= 0; --r) {
            double v = y[r];
            for (int k = r + 1; k < rank; ++k) v -= S[(size_t)r * R + k] * z[k];
            z[r] = v / S[(size_t)r * R + r];
        }
        d.assign(R, 0.0);
        for (int r = 0; r < rank; ++r) d[cp[r]] = z[r] * sc[cp[r]];
    };
    std::vector<double> d, g(R);
    lu_solve(t, d);
    for (int r = 0; r < R; ++r) g[r] = g0[r] + d[r];
    if (nullity) *nullity = R - rank;
    if (set_globals(g)) return 1;
    // Residual check + iterative refinement.  Pass 1 forms S and t as sums of per-unit differences that cancel to the
    // part of a global row the diffuse members do NOT absorb; when they absorb nearly all of it (a fitted monopole
    // beside the CMB) S keeps only a few digits and S g = t is solved for a slightly wrong S.  The true residual of the
    // global rows, r = b - A x evaluated directly at the new state (k_schur_resid: no elimination, no cancellation),
    // drives the correction g += S^-1 r; the diffuse rows are re-solved exactly by pass 2.  The contraction factor is
    // cond(S) * (relative error of S); the loop stops at 1e-12 of the row of b, or when a step no longer helps.
    int refine = 0;
    double prev = INFINITY, resid_b = 0.0, resid_bw = 0.0;
    std::vector<double> rr(3 * R), res(R);
    for (int step = 0; step <= 4; ++step) {
        if (each([&](dangx_ctx* c, const GroupArgs& ga, long long SN) -> int {
                (void)hipSetDevice(c->device);
                return dx_launch_schur_resid(c, ga, sa, SN, c->work[0]);
            }))
            return 1;
        if (gather(rr, 3 * R)) return 1;
        double worst = 0.0, worst_bw = 0.0;
        for (int r = 0; r < R; ++r) {
            res[r] = rr[r] + fl[r];
            const double brow = std::fabs(rr[R + r] + fl[r]);
            // rows along a free (degenerate) direction cannot be reduced by the global amplitudes alone; they are
            // consistent through the diffuse members, so their residual is reported like any other
            worst = std::max(worst, std::fabs(res[r]) / std::max(brow, 1e-300));
            worst_bw = std::max(worst_bw, std::fabs(res[r]) / std::max(rr[2 * R + r] + std::fabs(fl[r]), 1e-300));
        }
        if (step > 0 && !(worst < prev)) {  // the last correction did not help: take it back
            for (int r = 0; r < R; ++r) g[r] -= d[r];
            if (set_globals(g)) return 1;
            refine -= 1;
            break;
        }
        resid_b = prev = worst;
        resid_bw = worst_bw;
        if (worst <= 1e-12 || worst_bw <= 1e-15 || step == 4) break;
        lu_solve(res, d);
        for (int r = 0; r < R; ++r) g[r] += d[r];
        if (set_globals(g)) return 1;
        refine += 1;
    }
    for (int r = 0; r < nc; ++r) { cs[r]->schur_resid = resid_b; cs[r]->schur_backward = resid_bw; cs[r]->schur_refine = refine; }
    return 0;
}

// host <-> device copy of `planes` maps of this shard.  The host side is either a packed [planes][npix] array or, after
// dangx_set_host_stride, a window into full-sky arrays: plane q starts host_stride doubles after plane q-1.
int copy_planes(dangx_ctx* ctx, void* dst, const void* src, size_t planes, bool to_device) {
    const size_t row = (size_t)ctx->dims.npix * sizeof(double);
    const size_t hs = (size_t)(ctx->host_stride > 0 ? ctx->host_stride : ctx->dims.npix) * sizeof(double);
    if (hs == row) {
        HIPCHK(ctx, hipMemcpyAsync(dst, src, row * planes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, ctx->stream));
    } else if (to_device) {
        HIPCHK(ctx, hipMemcpy2DAsync(dst, row, src, hs, row, planes, hipMemcpyHostToDevice, ctx->stream));
    } else {
        HIPCHK(ctx, hipMemcpy2DAsync(dst, hs, src, row, row, planes, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
// element (plane q, pixel t) of a host map array under the current host layout
inline double host_at(const dangx_ctx* ctx, const double* a, long long q, long long t) {
    return a[q * (ctx->host_stride > 0 ? ctx->host_stride : ctx->dims.npix) + t];
}

int check_comp(dangx_ctx* ctx, int comp) {
    if (comp < 0 || comp >= ctx->dims.ncomp) return fail(ctx, "component index out of range");
    return 0;
}

// c%amplitude / c%indices of a component, zero-initialised, unless the caller's device buffers were adopted
int ensure_state(dangx_ctx* ctx, int comp) {
    if (!ctx->comp_set[comp]) return fail(ctx, "component not set");
    (void)hipSetDevice(ctx->device);
    const size_t plane = (size_t)ctx->dims.npix * ctx->dims.nmaps * sizeof(double);
    if (!ctx->amp[comp]) {
        HIPCHK(ctx, hipMalloc(&ctx->amp[comp], plane));
        HIPCHK(ctx, hipMemset(ctx->amp[comp], 0, plane));
        ctx->own_amp[comp] = true; ctx->dirty = true;
    }
    const int nind = ctx->desc[comp].nindices;
    if (nind > 0 && !ctx->idx[comp]) {
        HIPCHK(ctx, hipMalloc(&ctx->idx[comp], plane * nind));
        HIPCHK(ctx, hipMemset(ctx->idx[comp], 0, plane * nind));
        ctx->own_idx[comp] = true; ctx->dirty = true;
        ctx->qu_equal[comp] = (1u << nind) - 1u;
        idx_written(ctx, comp);
    }
    return 0;
}

}  // namespace

// ======================================================================= C ABI

static int seam_common(dangx_ctx* ctx, int group, int flag, GroupArgs& a, long long& SN, long long& n);

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
        hipMalloc(&ctx->chi_cache, CHI_CACHE_DOUBLES * sizeof(double)) != hipSuccess ||
        hipMalloc(&ctx->chi_stage, sizeof(double) * dangx_ctx::CHI_RING * CHI_ROWS * CHI_RSTAGE) != hipSuccess ||
        hipMalloc(&ctx->rows_out, (4 * MAXB + 8) * sizeof(double)) != hipSuccess ||
        hipMalloc(&ctx->counters, 16 * sizeof(unsigned long long)) != hipSuccess) {
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
    for (int l = 0; l < MAXC; ++l) if (ctx->tmpl[l]) (void)hipFree(ctx->tmpl[l]);
    if (ctx->d_bp_nu0) { (void)hipFree(ctx->d_bp_nu0); (void)hipFree(ctx->d_bp_tau0); (void)hipFree(ctx->d_bp_lnr); }
    if (ctx->fs_data) (void)hipFree(ctx->fs_data);
    // HEALPix index tables and the degraded maps of the coarse-Nside sweeps live as long as the context
    for (int** b : {&ctx->hp_n2r_f, &ctx->hp_r2n_f, &ctx->hp_n2r_c, &ctx->hp_r2n_c}) { if (*b) (void)hipFree(*b); *b = nullptr; }
    for (double** b : {&ctx->cs_data, &ctx->cs_rms, &ctx->cs_mask, &ctx->cs_index}) { if (*b) (void)hipFree(*b); *b = nullptr; }
    if (ctx->cs_part) (void)hipFree(ctx->cs_part);
    ctx->cs_part = nullptr; ctx->cs_part_cap = 0;
    ctx->hp_nside = ctx->hp_cnside = 0; ctx->cs_cap = 0;
    (void)hipFree(ctx->rows_out);
    (void)hipFree(ctx->dm); (void)hipFree(ctx->scalars); (void)hipFree(ctx->counters); (void)hipFree(ctx->chi_cache);
    for (auto& p : ctx->chi_pend) if (p.buf) (void)hipFree(p.buf);
    if (ctx->chi_stage) (void)hipFree(ctx->chi_stage);
    for (auto& e : ctx->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    dx_rtc_release(ctx);
    delete ctx;
    return 0;
}

const char* dangx_last_error(const dangx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int dangx_set_stream(dangx_ctx* ctx, void* s) {
    if (!ctx) return 1;
    if ((hipStream_t)s != ctx->stream) {
        // chi^2 block partials of the sweeps launched so far wait in the ring: reduce them on the stream that produced
        // them, and let everything enqueued there finish before the first launch on the new stream can overtake it
        (void)hipSetDevice(ctx->device);
        if (chi_flush(ctx)) return 1;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->stream = (hipStream_t)s;
    return 0;
}

int dangx_set_allreduce(dangx_ctx* ctx, dangx_allreduce_fn fn, void* user, int is_root) {
    if (!ctx) return 1;
    ctx->allreduce = fn; ctx->allreduce_user = user; ctx->is_root = (fn == nullptr) || is_root != 0;
    return 0;
}

int dangx_device_count(int* n) {
    if (!n) return 1;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *n = c;
    return 0;
}

int dangx_set_host_stride(dangx_ctx* ctx, int64_t plane_stride) {
    if (!ctx) return 1;
    if (plane_stride != 0 && plane_stride < ctx->dims.npix) return fail(ctx, "host plane stride smaller than the shard");
    ctx->host_stride = plane_stride;
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
    ctx->dirty = true; ctx->bp_dirty = true;
    invalidate_chi(ctx);
    return 0;
}

int dangx_set_component(dangx_ctx* ctx, int comp, const dangx_comp_desc* d) {
    if (!ctx || !d) return 1;
    if (check_comp(ctx, comp)) return 1;
    if (d->type < DANGX_POWERLAW || d->type > DANGX_HIFIT)
        return fail(ctx, "Error - unrecognized component type (only diffuse types are built)");
    const int want = (d->type == DANGX_MBB || d->type == DANGX_LOGNORMAL) ? 2
                     : (d->type == DANGX_CMB || d->type == DANGX_TEMPLATE || d->type == DANGX_MONOPOLE) ? 0 : 1;
    if (d->nindices != want) return fail(ctx, "nindices does not match the component type");
    // the amplitude / index maps are allocated on first use (ensure_state): a caller that adopts its own device
    // buffers (dangx_adopt_device_state) never holds two copies
    ctx->desc[comp] = *d;
    if (ctx->desc[comp].nu_ref < 1e7) ctx->desc[comp].nu_ref *= 1e9;  // src/dang_param_mod.f90:571-573
    ctx->comp_set[comp] = true;
    ctx->dirty = true; ctx->bp_dirty = true;
    invalidate_chi(ctx);
    return 0;
}

int dangx_set_tcmb(dangx_ctx* ctx, double T) {
    if (!ctx) return 1;
    ctx->hm.tcmb = T;
    ctx->dirty = true;
    invalidate_chi(ctx);
    return 0;
}

int dangx_set_calibration(dangx_ctx* ctx, const double* gain, const double* offset) {
    if (!ctx) return 1;
    for (int j = 0; j < ctx->dims.nbands; ++j) {
        if (gain) ctx->hm.gain[j] = gain[j];
        if (offset) ctx->hm.offset[j] = offset[j];
    }
    ctx->dirty = true;
    invalidate_chi(ctx);
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
    const size_t nplanes = (size_t)ctx->dims.nmaps * ctx->dims.nbands;
    if (copy_planes(ctx, ctx->sig, sig, nplanes, true) || copy_planes(ctx, ctx->rms, rms, nplanes, true) ||
        copy_planes(ctx, ctx->mask, mask, (size_t)ctx->dims.nmaps, true))
        return 1;
    ctx->dirty = true;
    invalidate_chi(ctx);
    idx_written(ctx, -1);   // a new mask
    return 0;
}

int dangx_adopt_device_data(dangx_ctx* ctx, const double* sig, const double* rms, const double* mask) {
    if (!ctx || !sig || !rms || !mask) return 1;
    if (ctx->own_data) { (void)hipFree(ctx->sig); (void)hipFree(ctx->rms); (void)hipFree(ctx->mask); ctx->own_data = false; }
    ctx->sig = const_cast<double*>(sig);
    ctx->rms = const_cast<double*>(rms);
    ctx->mask = const_cast<double*>(mask);
    ctx->dirty = true;
    invalidate_chi(ctx);
    idx_written(ctx, -1);
    return 0;
}

int dangx_put_amplitude(dangx_ctx* ctx, int comp, const double* amp) {
    if (!ctx || !amp || check_comp(ctx, comp) || ensure_state(ctx, comp)) return 1;
    invalidate_chi(ctx);
    ctx->plane_nz[comp] = 0;
    for (int k = 0; k < ctx->dims.nmaps; ++k)
        for (long long t = 0; t < ctx->dims.npix; ++t)
            if (host_at(ctx, amp, k, t) != 0.0) { ctx->plane_nz[comp] |= 1u << k; break; }
    return copy_planes(ctx, ctx->amp[comp], amp, (size_t)ctx->dims.nmaps, true);
}
int dangx_get_amplitude(dangx_ctx* ctx, int comp, double* amp) {
    if (!ctx || !amp || check_comp(ctx, comp) || ensure_state(ctx, comp)) return 1;
    return copy_planes(ctx, amp, ctx->amp[comp], (size_t)ctx->dims.nmaps, false);
}
int dangx_put_indices(dangx_ctx* ctx, int comp, const double* ind) {
    if (!ctx || !ind || check_comp(ctx, comp) || ensure_state(ctx, comp)) return 1;
    if (!ctx->idx[comp]) return fail(ctx, "component has no indices");
    invalidate_chi(ctx);
    idx_written(ctx, comp);
    {   // planes on which every index map is spatially constant
        const long long np = ctx->dims.npix;
        ctx->idx_const[comp] = 0;
        for (int k = 0; k < ctx->dims.nmaps; ++k) {
            bool cst = true;
            for (int q = 0; q < ctx->desc[comp].nindices && cst; ++q) {
                const long long pl = (long long)q * ctx->dims.nmaps + k;
                const double m0 = host_at(ctx, ind, pl, 0);
                for (long long t = 1; t < np; ++t) if (host_at(ctx, ind, pl, t) != m0) { cst = false; break; }
                ctx->idx_val[comp][k][q] = m0;
            }
            if (cst) ctx->idx_const[comp] |= 1u << k;
        }
        ctx->qu_equal[comp] = 0;
        for (int q = 0; q < ctx->desc[comp].nindices && ctx->dims.nmaps == 3; ++q) {
            bool eq = true;
            for (long long t = 0; t < np && eq; ++t) eq = host_at(ctx, ind, (long long)q * 3 + 1, t) == host_at(ctx, ind, (long long)q * 3 + 2, t);
            if (eq) ctx->qu_equal[comp] |= 1u << q;
        }
        ctx->dirty = true;
    }
    return copy_planes(ctx, ctx->idx[comp], ind, (size_t)ctx->dims.nmaps * ctx->desc[comp].nindices, true);
}
int dangx_get_indices(dangx_ctx* ctx, int comp, double* ind) {
    if (!ctx || !ind || check_comp(ctx, comp) || ensure_state(ctx, comp)) return 1;
    if (!ctx->idx[comp]) return fail(ctx, "component has no indices");
    return copy_planes(ctx, ind, ctx->idx[comp], (size_t)ctx->dims.nmaps * ctx->desc[comp].nindices, false);
}
int dangx_set_template(dangx_ctx* ctx, int comp, const double* tmpl, const int32_t* corr, int nfit) {
    if (!ctx || !tmpl || !corr || check_comp(ctx, comp)) return 1;
    if (!ctx->comp_set[comp] || !is_global_type(ctx->desc[comp].type)) return fail(ctx, "not a template / monopole / hi_fit component");
    (void)hipSetDevice(ctx->device);
    int mask = 0, cnt = 0;
    for (int j = 0; j < ctx->dims.nbands; ++j) if (corr[j]) { mask |= 1 << j; ++cnt; }
    if (cnt != nfit) return fail(ctx, "nfit does not match the number of fitted (corr) bands");
    const size_t bytes = (size_t)ctx->dims.npix * ctx->dims.nmaps * sizeof(double);
    if (!ctx->tmpl[comp]) HIPCHK(ctx, hipMalloc(&ctx->tmpl[comp], bytes));
    if (copy_planes(ctx, ctx->tmpl[comp], tmpl, (size_t)ctx->dims.nmaps, true)) return 1;
    ctx->corr_mask[comp] = mask; ctx->nfit[comp] = nfit;
    ctx->dirty = true;
    invalidate_chi(ctx);
    return 0;
}
int dangx_put_template_amplitudes(dangx_ctx* ctx, int comp, const double* ta) {
    if (!ctx || !ta || check_comp(ctx, comp)) return 1;
    for (int k = 0; k < ctx->dims.nmaps; ++k)
        for (int j = 0; j < ctx->dims.nbands; ++j) ctx->tamp[comp][k][j] = ta[k * ctx->dims.nbands + j];
    if (ctx->desc[comp].type == DANGX_MONOPOLE)
        for (int j = 0; j < ctx->dims.nbands; ++j) ctx->hm.offset[j] = ctx->tamp[comp][0][j];
    ctx->dirty = true;
    invalidate_chi(ctx);
    return 0;
}
int dangx_get_template_amplitudes(dangx_ctx* ctx, int comp, double* ta) {
    if (!ctx || !ta || check_comp(ctx, comp)) return 1;
    for (int k = 0; k < ctx->dims.nmaps; ++k)
        for (int j = 0; j < ctx->dims.nbands; ++j) ta[k * ctx->dims.nbands + j] = ctx->tamp[comp][k][j];
    return 0;
}

int dangx_adopt_device_state(dangx_ctx* ctx, int comp, double* amp_dev, double* idx_dev) {
    if (!ctx || !amp_dev || check_comp(ctx, comp)) return 1;
    if (!ctx->comp_set[comp]) return fail(ctx, "component not set");
    if (ctx->desc[comp].nindices > 0 && !idx_dev) return fail(ctx, "component has indices: idx_dev required");
    if (ctx->amp[comp] && ctx->own_amp[comp]) (void)hipFree(ctx->amp[comp]);
    if (ctx->idx[comp] && ctx->own_idx[comp]) (void)hipFree(ctx->idx[comp]);
    invalidate_chi(ctx);
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
    ctx->idx_ext[comp] = true;   // the caller may write these maps at any time
    idx_written(ctx, comp);
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
        ctx->qu_equal[comp] = 0;
        if (nmaps == 3) {
            unsigned dq = 0;
            HIPCHK(ctx, hipMemsetAsync(df, 0, sizeof(unsigned), ctx->stream));
            hipLaunchKernelGGL(k_qu_differ, dim3(1024), dim3(BLOCK), 0, ctx->stream, idx_dev, (long long)ctx->dims.npix, nind, df);
            HIPCHK(ctx, hipMemcpyAsync(&dq, df, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            ctx->qu_equal[comp] = ~dq & ((1u << nind) - 1u);
        }
    }
    ctx->dirty = true;
    return 0;
}
void* dangx_amplitude_devptr(dangx_ctx* ctx, int comp) {
    return (ctx && comp >= 0 && comp < ctx->dims.ncomp && !ensure_state(ctx, comp)) ? ctx->amp[comp] : nullptr;
}
void* dangx_indices_devptr(dangx_ctx* ctx, int comp) {
    return (ctx && comp >= 0 && comp < ctx->dims.ncomp && !ensure_state(ctx, comp)) ? ctx->idx[comp] : nullptr;
}

int64_t dangx_group_size(dangx_ctx* ctx, int group, int flag) {
    GroupArgs a;
    if (!ctx || make_group(ctx, group, flag, a)) return -1;
    return (int64_t)a.ng * flag_planes_h(flag) * ctx->hm.npix + a.nglob;
}

static int planeset_launch(dangx_ctx* ctx, const GroupArgs& g, const SweepList& sl, int lanes, int solve, int64_t* n_not_spd, int64_t* accepted);
static bool planeset_group(dangx_ctx* ctx, const GroupArgs& g);

int dangx_amp_sample(dangx_ctx* ctx, int group, int flag, int ml_mode, int solver, int fluct_mode, uint64_t seed,
                     uint64_t stream, int i_max, double converge, int* cg_iters, int64_t* n_not_spd) {
    DxRange rg_("dangx_amp_sample");
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
    if (a.nt > 0 && solver != DANGX_SOLVER_CG) {
        if (fluct_mode != DANGX_FLUCT_REFERENCE && ml_mode == DANGX_ML_SAMPLE)
            return fail(ctx, "groups with template / monopole / hi_fit members reproduce the reference's fluctuation term only");
        int nullity = 0;
        dangx_ctx* one[1] = {ctx};
        if (device_schur(one, 1, &a, &SN, n_not_spd, &nullity)) return 1;
        if (cg_iters) *cg_iters = -nullity;  // 0: regular system; -k: k directions of the global amplitudes left at their current value
        return 0;
    }
    if (solver == DANGX_SOLVER_CG) {
        if (fluct_mode != DANGX_FLUCT_REFERENCE && ml_mode == DANGX_ML_SAMPLE)
            return fail(ctx, "the CG solver reproduces the reference's fluctuation term only");
        return device_cg(ctx, a, i_max, converge, cg_iters);
    }
    if (ctx->defer_amp) {  // dangx_amp_index_sample: the launch waits for the index sweep it is fused with
        ctx->pending = a; ctx->pending_SN = SN; ctx->have_pending = true;
        return 0;
    }
    // The amplitude phase with chi^2 of the state it leaves as a by-product (src/dang_cg_mod.f90:172-173 asks for it after every
    // group): the plane-set kernel without sweep items forms the residual of the new amplitudes anyway -- where it covers the
    // model (delta bands, the group's members the only components on the planes, reference fluctuation term) the statistics need
    // no pass of their own over the maps.  DANGX_AMP_CHI=0: the stand-alone amplitude kernel (A/B timing).
    static const bool with_chi = [] { const char* e = getenv("DANGX_AMP_CHI"); return !(e && e[0] == '0'); }();
    if (with_chi && (ml_mode == DANGX_ML_OPTIMIZE || fluct_mode == DANGX_FLUCT_REFERENCE) && planeset_group(ctx, a)) {
        SweepList sl;
        std::memset(&sl, 0, sizeof(sl));
        sl.s1 = (flag & DANGX_FLAG_QU) ? 2 : (flag & DANGX_FLAG_T) ? 1 : (flag & DANGX_FLAG_Q) ? 2 : 3;
        sl.s2 = (flag & DANGX_FLAG_QU) ? 3 : sl.s1;
        sl.ml_mode = ml_mode;
        const int lanes = dx_planeset_lanes(ctx, a, sl, 1);
        if (lanes) return planeset_launch(ctx, a, sl, lanes, 1, n_not_spd, nullptr);
    }
    if (n_not_spd) HIPCHK(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long), ctx->stream));
    if (dx_launch_amp(ctx, a, SN)) return 1;
    HIPCHK(ctx, hipGetLastError());
    if (n_not_spd) {
        unsigned long long v = 0;
        HIPCHK(ctx, hipMemcpyAsync(&v, ctx->counters, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *n_not_spd = (int64_t)v;
    }
    return 0;
}

// One (group, flag) pass of sample_cg_groups over several contexts of ONE process (include/dangx.h).  Independent per-pixel
// work is enqueued on every device before the first result is awaited; a coupled group shares its Schur rows.
int dangx_sky_amp_sample(dangx_ctx* const* ctxs, int nctx, int group, int flag, int ml_mode, int solver, int fluct_mode, uint64_t seed,
                         uint64_t stream, int i_max, double converge, int* cg_iters, int64_t* n_not_spd) {
    DxRange rg_("dangx_sky_amp_sample");
    if (!ctxs || nctx < 1) return 1;
    for (int r = 0; r < nctx; ++r) if (!ctxs[r]) return 1;
    dangx_ctx* c0 = ctxs[0];
    if (nctx == 1) return dangx_amp_sample(c0, group, flag, ml_mode, solver, fluct_mode, seed, stream, i_max, converge, cg_iters, n_not_spd);
    if (nctx > 64) return fail(c0, "too many contexts");
    auto bubble = [&](dangx_ctx* who) { if (who != c0) c0->err = who->err; return 1; };
    if (ml_mode != DANGX_ML_SAMPLE && ml_mode != DANGX_ML_OPTIMIZE) return fail(c0, "bad ml_mode");
    GroupArgs probe;
    if (make_group(c0, group, flag, probe)) return 1;
    if (cg_iters) *cg_iters = 0;
    if (n_not_spd) *n_not_spd = 0;
    if (probe.nt == 0 && solver != DANGX_SOLVER_CG) {  // block diagonal: every shard on its own
        for (int r = 0; r < nctx; ++r) {
            if (n_not_spd) {  // the amplitude kernels add to counters[0]: each context starts its launch from zero, on its own stream
                (void)hipSetDevice(ctxs[r]->device);
                HIPCHK(c0, hipMemsetAsync(ctxs[r]->counters, 0, sizeof(unsigned long long), ctxs[r]->stream));
            }
            if (dangx_amp_sample(ctxs[r], group, flag, ml_mode, solver, fluct_mode, seed, stream, i_max, converge, nullptr, nullptr))
                return bubble(ctxs[r]);
        }
        if (n_not_spd)   // the counters are read after every device has its launch
            for (int r = 0; r < nctx; ++r) {
                unsigned long long v = 0;
                (void)hipSetDevice(ctxs[r]->device);
                HIPCHK(c0, hipMemcpyAsync(&v, ctxs[r]->counters, sizeof(v), hipMemcpyDeviceToHost, ctxs[r]->stream));
                HIPCHK(c0, hipStreamSynchronize(ctxs[r]->stream));
                *n_not_spd += (int64_t)v;
            }
        return 0;
    }
    if (solver == DANGX_SOLVER_CG)
        return fail(c0, "the device CG (DANGX_SOLVER_CG) iterates on ONE context per process: use DANGX_SOLVER_DIRECT, or one process per GPU with dangx_set_allreduce");
    if (fluct_mode != DANGX_FLUCT_REFERENCE && ml_mode == DANGX_ML_SAMPLE)
        return fail(c0, "groups with template / monopole / hi_fit members reproduce the reference's fluctuation term only");
    std::vector<GroupArgs> as((size_t)nctx);
    std::vector<long long> SNs((size_t)nctx);
    for (int r = 0; r < nctx; ++r) {
        dangx_ctx* c = ctxs[r];
        (void)hipSetDevice(c->device);
        if (make_group(c, group, flag, as[r]) || sync_model(c)) return bubble(c);
        as[r].ml_mode = ml_mode; as[r].fluct = fluct_mode; as[r].seed = seed; as[r].stream = stream;
        SNs[r] = (long long)flag_planes_h(flag) * c->hm.npix;
        for (int pl = 0; pl < flag_planes_h(flag); ++pl) {
            const int k = (flag & DANGX_FLAG_QU) ? 2 + pl : (flag & DANGX_FLAG_T) ? 1 : (flag & DANGX_FLAG_Q) ? 2 : 3;
            c->chi_before_valid[k - 1] = c->chi_after_valid[k - 1] = c->touched_since_amp[k - 1] = false;
            for (int g = 0; g < as[r].ng; ++g) c->plane_nz[as[r].gc[g]] |= 1u << (k - 1);
        }
        if (as[r].nglob != as[0].nglob || as[r].nt != as[0].nt) return fail(c0, "the contexts disagree on the group's global-amplitude members");
    }
    int nullity = 0;
    if (device_schur(ctxs, nctx, as.data(), SNs.data(), n_not_spd, &nullity)) return 1;
    if (cg_iters) *cg_iters = -nullity;
    return 0;
}

int dangx_schur_info(dangx_ctx* ctx, double* rel_residual, int* refinements) {
    if (!ctx) return 1;
    if (rel_residual) { rel_residual[0] = ctx->schur_resid; rel_residual[1] = ctx->schur_backward; }
    if (refinements) *refinements = ctx->schur_refine;
    return 0;
}

int dangx_amp_residual(dangx_ctx* ctx, int group, int flag, int ml_mode, uint64_t seed, uint64_t stream, double* out) {
    if (!ctx || !out) return 1;
    (void)hipSetDevice(ctx->device);
    if (ml_mode != DANGX_ML_SAMPLE && ml_mode != DANGX_ML_OPTIMIZE) return fail(ctx, "bad ml_mode");
    GroupArgs a;
    long long SN, n;
    if (seam_common(ctx, group, flag, a, SN, n)) return 1;
    a.ml_mode = ml_mode; a.fluct = DANGX_FLUCT_REFERENCE; a.seed = seed; a.stream = stream;
    const bool mixed = a.nt > 0;
    const long long ndiff = SN * a.ng;
    double *x = ctx->work[0], *eta = ctx->work[1], *q = ctx->work[3], *b2 = ctx->work[4], *b = ctx->work[5];
    hipStream_t st = ctx->stream;
    if (mixed ? dx_launch_rhs_mixed(ctx, a, SN, b) : dx_launch_rhs(ctx, a, SN, b)) return 1;
    if (mixed && rank_sum_rows(ctx, b + ndiff, a.nglob)) return 1;
    if (ml_mode == DANGX_ML_SAMPLE) {
        hipLaunchKernelGGL(k_draw_eta, dim3(nblocks(SN)), dim3(BLOCK), 0, st, ctx->dm, a, eta);
        if (mixed ? dx_launch_sv_mixed(ctx, a, SN, eta, q) : dx_launch_sample_vector(ctx, a, SN, eta, q)) return 1;
        if (mixed && rank_sum_rows(ctx, q + ndiff, a.nglob)) return 1;
        hipLaunchKernelGGL(k_cg_vec, dim3(nblocks(n)), dim3(BLOCK), 0, st, 3, n, n, 0.0, b2, nullptr, nullptr, q, b, nullptr);
    } else {
        HIPCHK(ctx, hipMemcpyAsync(b2, b, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, st));
    }
    if (a.ng) hipLaunchKernelGGL(k_pack, dim3(nblocks(SN)), dim3(BLOCK), 0, st, ctx->dm, a, x, 0);
    std::vector<double> xg;
    if (mixed) {
        globals_to_x(ctx, a, xg);
        HIPCHK(ctx, hipMemcpyAsync(x + ndiff, xg.data(), sizeof(double) * a.nglob, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    if (mixed ? dx_launch_Ax_mixed(ctx, a, SN, x, q) : dx_launch_Ax(ctx, a, SN, x, q, nullptr)) return 1;
    if (mixed && rank_sum_rows(ctx, q + ndiff, a.nglob)) return 1;
    double sums[2] = {0.0, 0.0};
    if (ndiff > 0) {
        const unsigned nblk = nblocks(ndiff);
        if (ensure_partial(ctx, 2ll * nblk)) return 1;
        hipLaunchKernelGGL(k_resid_norm, dim3(nblk), dim3(BLOCK), 0, st, ctx->dm, b2, q, ndiff, SN, ctx->partial);
        hipLaunchKernelGGL(k_reduce_rows_final, dim3(2), dim3(BLOCK), 0, st, ctx->partial, (long long)nblk, 2, ctx->rows_out);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(sums, ctx->rows_out, sizeof(sums), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        if (rank_sum(ctx, sums, 2)) return 1;
    }
    double worst = 0.0;
    if (a.nglob > 0) {
        std::vector<double> bg(a.nglob), qg(a.nglob);
        HIPCHK(ctx, hipMemcpyAsync(bg.data(), b2 + ndiff, sizeof(double) * a.nglob, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(qg.data(), q + ndiff, sizeof(double) * a.nglob, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        for (int r = 0; r < a.nglob; ++r) {
            const double res = bg[r] - qg[r];
            sums[0] += res * res; sums[1] += bg[r] * bg[r];
            worst = std::max(worst, std::fabs(res) / std::max(std::fabs(bg[r]), 1e-300));
        }
    }
    out[0] = (sums[1] > 0.0) ? std::sqrt(sums[0] / sums[1]) : 0.0;
    out[1] = worst;
    return 0;
}

int dangx_index_sample(dangx_ctx* ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                       uint64_t stream, int64_t* accepted) {
    DxRange rg_("dangx_index_sample");
    if (!ctx || check_comp(ctx, comp)) return 1;
    (void)hipSetDevice(ctx->device);
    {   // this sweep makes the component's index map pixel dependent on the touched planes
        unsigned touched = 0;
        if (map_n == -1) touched = 6u; else if (map_n >= 1 && map_n <= 3) touched = 1u << (map_n - 1);
        if (ctx->idx_const[comp] & touched) { ctx->idx_const[comp] &= ~touched; ctx->dirty = true; }
        idx_written(ctx, comp);
        if (nind >= 0 && nind < DANGX_MAX_IND) {  // a Q+U sweep writes one value to both planes (:465); a Q or U sweep to one
            if (map_n == -1) ctx->qu_equal[comp] |= 1u << nind;
            else if (map_n == 2 || map_n == 3) ctx->qu_equal[comp] &= ~(1u << nind);
        }
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
    if (ml_mode != DANGX_ML_SAMPLE && ml_mode != DANGX_ML_OPTIMIZE) return fail(ctx, "bad ml_mode");
    const int Sp = a.s2 - a.s1 + 1;
    a.others = 0;
    for (int l = 0; l < ctx->hm.ncomp; ++l)
        if (l != comp && ((ctx->plane_nz[l] & ((1u << (a.s1 - 1)) | (1u << (a.s2 - 1)))) || ctx->desc[l].type == DANGX_TCMB ||
                          is_global_type(ctx->desc[l].type)))
            a.others |= 1u << l;
    // chain mode: factorised SED when every band is a delta bandpass
    const bool all_delta = ctx->hm.all_delta != 0;
    a.mode = CH_GENERIC;
    a.bp = all_delta ? 0 : 1;
    if (d.type == DANGX_POWERLAW) a.mode = CH_POW;
    else if (d.type == DANGX_MBB) a.mode = nind == 0 ? CH_MBB_BETA : CH_MBB_T;
    else if (d.type == DANGX_LOGNORMAL && all_delta) a.mode = nind == 0 ? CH_LOGN_NUP : CH_LOGN_W;
    // with bandpass-integrated bands (or T_cmb / template-type components present) the compile-time modes exist for the
    // chisq likelihood with a gaussian / uniform prior only; everything else takes the run-time generic chain
    if (a.bp && (d.lnl_type[nind] != DANGX_LNL_CHISQ || d.prior_type[nind] == DANGX_PRIOR_JEFFREYS)) a.mode = CH_GENERIC;
    // LDS columns: (2*Sp+1)*nb doubles per thread; pick the block so that >= 2 blocks fit in 160 KiB
    const size_t per_thread = (size_t)(2 * Sp + 1) * ctx->hm.nbands * sizeof(double);
    const size_t tabsz = (size_t)(TROWS * ctx->hm.ncomp + 3) * ctx->hm.nbands * sizeof(double);
    int bs = 256;
    while (bs > 64 && tabsz + per_thread * bs > 76 * 1024) bs >>= 1;
    const size_t lds = tabsz + per_thread * bs;
    const bool reg_ok = !a.bp && d.lnl_type[nind] == DANGX_LNL_CHISQ && d.prior_type[nind] != DANGX_PRIOR_JEFFREYS &&
                        a.mode != CH_GENERIC && dx_mh_reg_supported(ctx, a.mode, ctx->hm.nbands, Sp);
    if (reg_ok) bs = BLOCK;  // register-resident form: no LDS columns
    // lanes per pixel: of the chain's register form, or of the fused launch when a solve on these planes is waiting for this
    // sweep and the model takes the one-launch form (decided now: the grid and the chi^2 buffers are sized by it)
    int lanes = reg_ok ? dx_mh_reg_lanes(ctx->hm.nbands, Sp) : 1;
    const int fused_lanes = (ctx->have_pending && reg_ok) ? dx_fused_lanes(ctx, ctx->pending, a, Sp) : 0;
    if (fused_lanes) lanes = fused_lanes;
    const unsigned nblk = nblocks((long long)ctx->hm.npix * lanes, bs);
    constexpr int RSTAGE = 128;  // blocks of the first reduction stage
    double* chi_buf = nullptr;
    if (chi_next(ctx, nblk, &chi_buf)) return 1;
    // the sweep kernels take their [4][nblk] chi^2 partial buffer from ctx->partial: lend them the ring's, give the
    // context's own back on every way out of the launch section
    struct Lend {
        dangx_ctx* c; double* saved;
        Lend(dangx_ctx* c_, double* b) : c(c_), saved(c_->partial) { c->partial = b; }
        void back() { if (c) { c->partial = saved; c = nullptr; } }
        ~Lend() { back(); }
    } lend(ctx, chi_buf);
    if (accepted) HIPCHK(ctx, hipMemsetAsync(ctx->counters + 1, 0, 2 * sizeof(unsigned long long), ctx->stream));
    bool fused = false;
    if (ctx->have_pending) {  // an amplitude solve on these planes is waiting: one launch for both, or the solve first
        ctx->have_pending = false;
        unsigned long long* accp = accepted ? ctx->counters + 1 : nullptr;
        if (fused_lanes) {
            Timed t(ctx, DANGX_K_AMP_INDEX, Sp);
            fused = dx_launch_fused(ctx, ctx->pending, a, Sp, fused_lanes, nblk, accp);
        }
        if (!fused) {
            if (fused_lanes) return fail(ctx, "the fused solve + sweep launch failed after its kernel was prepared");
            if (dx_launch_amp(ctx, ctx->pending, ctx->pending_SN)) return 1;
        }
    }
    if (!fused && ctx->pair_on) {  // dangx_index_sample_pair: this sweep and the sweep of index nind + 1 in one launch
        ctx->pair_on = false;
        if (reg_ok && nind + 1 < d.nindices) {
            IndexArgs b = a;
            b.nind = nind + 1; b.stream = ctx->pair_stream;
            b.mode = (d.type == DANGX_MBB) ? CH_MBB_T : (d.type == DANGX_LOGNORMAL && all_delta) ? CH_LOGN_W : CH_GENERIC;
            const bool ok_b = d.lnl_type[nind + 1] == DANGX_LNL_CHISQ && d.prior_type[nind + 1] != DANGX_PRIOR_JEFFREYS;
            unsigned long long* accp = accepted ? ctx->counters + 1 : nullptr;  // counters[1], counters[2]
            if (ok_b) {
                Timed t(ctx, DANGX_K_INDEX_MH, Sp);
                fused = ctx->pair_done = dx_launch_mh_pair(ctx, a, b, Sp, nblk, accp);
            }
        }
    }
    if (!fused) {
        Timed t(ctx, DANGX_K_INDEX_MH, Sp);
        unsigned long long* accp = accepted ? ctx->counters + 1 : nullptr;
        const bool fast = d.lnl_type[nind] == DANGX_LNL_CHISQ &&
                          (a.mode == CH_POW || a.mode == CH_MBB_BETA || a.mode == CH_MBB_T);
        if (!(reg_ok && dx_launch_mh_reg(ctx, a, Sp, nblk, accp))) dx_launch_mh_lds(ctx, a, fast, Sp, nblk, bs, lds, accp);
    }
    lend.back();
    HIPCHK(ctx, hipGetLastError());  // a failed launch must not leave a pending entry over partials nobody wrote
    {   // fused chi^2 of the touched planes (before = state left by the amplitude phase, after = new state): the block
        // partials wait in the ring (chi_flush) until a value is asked for
        const bool wb = !ctx->touched_since_amp[a.s1 - 1];
        auto& pend = ctx->chi_pend[ctx->chi_npend++];
        pend.nblk = nblk; pend.s1 = a.s1; pend.s2 = a.s2; pend.wb = wb ? 1 : 0;
        for (int k = a.s1; k <= a.s2; ++k) {
            if (wb) ctx->chi_before_valid[k - 1] = true;
            ctx->chi_after_valid[k - 1] = true;
            ctx->touched_since_amp[k - 1] = true;
        }
    }
    if (accepted) {
        unsigned long long v = 0;
        HIPCHK(ctx, hipMemcpyAsync(&v, ctx->counters + 1, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *accepted = (int64_t)v;
    }
    return 0;
}

// dangx_index_sample(comp, nind, ...) followed by dangx_index_sample(comp, nind + 1, ...) on the same planes: two
// consecutive indices of ONE component (the dust beta and dust T sweeps).  Nothing the second sweep removes from the data
// has changed in between, so where the register chain covers both (chisq likelihood, gaussian / uniform priors, delta
// bands; mbb beta -> T, log-normal nu_p -> w) they run in one launch on one staging of the maps -- bit for bit the two
// calls, which everything else takes.
int dangx_index_sample_pair(dangx_ctx* ctx, int comp, int nind, int map_n, int nsample, int ml_mode, uint64_t seed,
                            uint64_t stream_first, uint64_t stream_second, int64_t* accepted_first, int64_t* accepted_second) {
    DxRange rg_("dangx_index_sample_pair");
    if (!ctx || check_comp(ctx, comp)) return 1;
    static const bool enabled = [] { const char* e = getenv("DANGX_FUSE"); return !(e && e[0] == '0'); }();
    const bool want_counts = accepted_first || accepted_second;
    int64_t acc1 = 0;
    ctx->pair_on = enabled; ctx->pair_done = false; ctx->pair_stream = stream_second;
    int rc = dangx_index_sample(ctx, comp, nind, map_n, nsample, ml_mode, seed, stream_first, want_counts ? &acc1 : nullptr);
    ctx->pair_on = false;
    if (rc) return rc;
    if (accepted_first) *accepted_first = acc1;
    if (!ctx->pair_done) return dangx_index_sample(ctx, comp, nind + 1, map_n, nsample, ml_mode, seed, stream_second, accepted_second);
    ctx->pair_done = false;
    if (nind + 1 < DANGX_MAX_IND) {
        if (map_n == -1) ctx->qu_equal[comp] |= 1u << (nind + 1);
        else if (map_n == 2 || map_n == 3) ctx->qu_equal[comp] &= ~(1u << (nind + 1));
    }
    if (accepted_second) {
        unsigned long long v = 0;
        HIPCHK(ctx, hipMemcpyAsync(&v, ctx->counters + 2, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *accepted_second = (int64_t)v;
    }
    return 0;
}

// dangx_amp_sample(group, flag, ...) followed by dangx_index_sample(comp, nind, map_n, ...) -- the amplitude solve of a CG
// group and the first index sweep on the same planes, which is how sample_cg_groups / sample_spectral_parameters follow
// each other plane set by plane set (src/dang.f90 main loop) -- with ONE kernel launch when the model allows it
// (dangx_fused.hip: delta bands, diffuse members only, direct solver, reference fluctuation term, chisq likelihood,
// gaussian / uniform prior, the sampled component a member of the group whose other members are the only other
// components on these planes).  Results are those of the two calls, bit for bit; every other configuration IS the two calls.
int dangx_amp_index_sample(dangx_ctx* ctx, int group, int flag, int ml_mode, int solver, int fluct_mode, uint64_t seed_amp,
                           uint64_t stream_amp, int i_max, double converge, int comp, int nind, int map_n, int nsample,
                           uint64_t seed_index, uint64_t stream_index, int* cg_iters, int64_t* n_not_spd, int64_t* accepted) {
    DxRange rg_("dangx_amp_index_sample");
    if (!ctx || check_comp(ctx, comp)) return 1;
    if (cg_iters) *cg_iters = 0;
    static const bool enabled = [] { const char* e = getenv("DANGX_FUSE"); return !(e && e[0] == '0'); }();  // A/B switch
    (void)hipSetDevice(ctx->device);
    if (sync_model(ctx)) return 1;  // all_delta and the constant-plane flags the decision below reads are set there
    bool can = enabled && solver == DANGX_SOLVER_DIRECT && ctx->hm.all_delta != 0 &&
               (ml_mode == DANGX_ML_OPTIMIZE || fluct_mode == DANGX_FLUCT_REFERENCE);
    // the planes of the sweep are the planes of the solve
    const int want = (flag == DANGX_FLAG_T) ? 1 : (flag == DANGX_FLAG_Q) ? 2 : (flag == DANGX_FLAG_U) ? 3 : (flag == DANGX_FLAG_QU) ? -1 : 0;
    can = can && want != 0 && want == map_n;
    if (can) {
        const dangx_comp_desc& d = ctx->desc[comp];
        const unsigned touched = (map_n == -1) ? 6u : 1u << (map_n - 1);
        // a sweep that turns a spatially constant index map into a varying one changes which SED route the SOLVE takes
        // (host-evaluated row against per-pixel evaluation) if it is launched after the descriptor update: first sweeps
        // on constant maps go the two-call way
        can = nind >= 0 && nind < d.nindices && !(ctx->idx_const[comp] & touched) && d.cg_group == group && d.sample_amplitude &&
              d.lnl_type[nind] == DANGX_LNL_CHISQ && d.prior_type[nind] != DANGX_PRIOR_JEFFREYS &&
              (d.type == DANGX_POWERLAW || d.type == DANGX_MBB);
    }
    if (can) {
        GroupArgs g;
        if (make_group(ctx, group, flag, g)) return 1;
        can = g.nt == 0 && g.no == 0 && g.nuc == 0 &&
              dx_fused_supported(ctx->desc[comp].type == DANGX_POWERLAW ? CH_POW : (nind == 0 ? CH_MBB_BETA : CH_MBB_T), ctx->hm.nbands, g.ng);
        for (int l = 0; can && l < ctx->hm.ncomp; ++l)  // a T_cmb component is an "other" of every sweep and never a diffuse member
            if (ctx->desc[l].type == DANGX_TCMB) can = false;
    }
    if (!can) {
        const int rc = dangx_amp_sample(ctx, group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, i_max, converge, cg_iters, n_not_spd);
        return rc ? rc : dangx_index_sample(ctx, comp, nind, map_n, nsample, ml_mode, seed_index, stream_index, accepted);
    }
    if (n_not_spd) HIPCHK(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long), ctx->stream));
    ctx->defer_amp = true;
    int rc = dangx_amp_sample(ctx, group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, i_max, converge, nullptr, nullptr);
    ctx->defer_amp = false;
    if (rc) { ctx->have_pending = false; return rc; }
    rc = dangx_index_sample(ctx, comp, nind, map_n, nsample, ml_mode, seed_index, stream_index, accepted);
    if (ctx->have_pending) {  // the sweep failed before its launch site: the solve still happens, as with the two calls
        ctx->have_pending = false;
        if (dx_launch_amp(ctx, ctx->pending, ctx->pending_SN)) return 1;
    }
    if (rc) return rc;
    if (n_not_spd) {
        unsigned long long v = 0;
        HIPCHK(ctx, hipMemcpyAsync(&v, ctx->counters, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        *n_not_spd = (int64_t)v;
    }
    return 0;
}

// One k_plane_set launch (dx_kern_planeset.h) with its bookkeeping: solve = 1 starts with the group's amplitude solve (what
// dangx_amp_sample records: the planes' cached chi^2 is stale, the members' amplitudes are about to be written), sl.n sweep items
// follow (what dangx_index_sample records per sweep).  The chi^2 by-products go to the ring: with a solve "before" = the state the
// solve leaves and "after" = the last sweep's (both the same value without sweeps); without, as for any sweep.
static int planeset_launch(dangx_ctx* ctx, const GroupArgs& g, const SweepList& sl, int lanes, int solve, int64_t* n_not_spd, int64_t* accepted) {
    const bool qu = sl.s2 > sl.s1;
    bool wb = true;
    if (solve) {
        for (int k = sl.s1; k <= sl.s2; ++k) {
            ctx->chi_before_valid[k - 1] = ctx->chi_after_valid[k - 1] = ctx->touched_since_amp[k - 1] = false;
            for (int q = 0; q < g.ng; ++q) ctx->plane_nz[g.gc[q]] |= 1u << (k - 1);
        }
    } else {
        wb = !ctx->touched_since_amp[sl.s1 - 1];
    }
    for (int q = 0; q < sl.n; ++q)
        for (int e = 0; e <= sl.s[q].pair; ++e) {
            idx_written(ctx, sl.s[q].comp);
            if (qu) ctx->qu_equal[sl.s[q].comp] |= 1u << (sl.s[q].nind + e);
            else if (sl.s1 == 2 || sl.s1 == 3) ctx->qu_equal[sl.s[q].comp] &= ~(1u << (sl.s[q].nind + e));
        }
    const unsigned nblk = nblocks((long long)ctx->hm.npix * lanes, BLOCK);
    double* chi_buf = nullptr;
    if (chi_next(ctx, nblk, &chi_buf)) return 1;
    HIPCHK(ctx, hipMemsetAsync(ctx->counters, 0, 16 * sizeof(unsigned long long), ctx->stream));
    {
        double* saved = ctx->partial;
        ctx->partial = chi_buf;
        bool ok;
        {
            Timed t(ctx, !solve ? DANGX_K_INDEX_MH : sl.n ? DANGX_K_AMP_INDEX : DANGX_K_AMP_DIRECT, sl.s2 - sl.s1 + 1);
            ok = dx_launch_planeset(ctx, g, sl, lanes, solve, nblk, accepted ? ctx->counters + 4 : nullptr);
        }
        ctx->partial = saved;
        if (!ok) return fail(ctx, "the plane-set launch failed after its kernel was prepared");
    }
    HIPCHK(ctx, hipGetLastError());
    {
        auto& pend = ctx->chi_pend[ctx->chi_npend++];
        pend.nblk = nblk; pend.s1 = sl.s1; pend.s2 = sl.s2; pend.wb = wb ? 1 : 0;
        pend.ns = 0;
        for (int q = 0; q < sl.n; ++q)   // the masked sums of the swept index maps ride along (rows 4 ..), in the items' order
            for (int e = 0; e <= sl.s[q].pair; ++e) {
                pend.slot[pend.ns++] = idx_slot(sl.s[q].comp, sl.s[q].nind + e, sl.s1);
                for (int k = sl.s1; k <= sl.s2; ++k) ctx->idxsum_dev[sl.s[q].comp][sl.s[q].nind + e][k - 1] = !ctx->idx_ext[sl.s[q].comp];
            }
        for (int k = sl.s1; k <= sl.s2; ++k) {
            if (wb) ctx->chi_before_valid[k - 1] = true;
            ctx->chi_after_valid[k - 1] = true;
            ctx->touched_since_amp[k - 1] = sl.n > 0 || (!solve && ctx->touched_since_amp[k - 1]);
        }
    }
    if (n_not_spd || accepted) {
        unsigned long long v[16];
        HIPCHK(ctx, hipMemcpyAsync(v, ctx->counters, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (n_not_spd) *n_not_spd = (int64_t)v[0];
        if (accepted) {   // the kernel counts per item (1 + pair entries each); the items follow the list's order
            int slot = 4, s = 0;
            for (int q = 0; q < sl.n; ++q)
                for (int e = 0; e <= sl.s[q].pair; ++e) accepted[s++] = (int64_t)v[slot++];
        }
    }
    return 0;
}

// the sweeps (comp[s], nind[s]), s = 0 .. nsweeps-1, as the items of a plane-set launch over group g's members: consecutive
// indices of a component travel in one item.  false: some sweep has no register-chain form, or the list is too long
static bool planeset_items(dangx_ctx* ctx, const GroupArgs& g, int map_n, int nsweeps, const int32_t* comp, const int32_t* nind,
                           const uint64_t* stream, SweepList& sl) {
    const unsigned touched = (map_n == -1) ? 6u : 1u << (map_n - 1);
    std::memset(&sl, 0, sizeof(sl));
    for (int s = 0; s < nsweeps; ++s) {
        const dangx_comp_desc& d = ctx->desc[comp[s]];
        int gm = -1;
        for (int q = 0; q < g.ng; ++q) if (g.gc[q] == comp[s]) gm = q;
        if (!(gm >= 0 && !(ctx->idx_const[comp[s]] & touched) && d.lnl_type[nind[s]] == DANGX_LNL_CHISQ &&
              d.prior_type[nind[s]] != DANGX_PRIOR_JEFFREYS && (d.type == DANGX_POWERLAW || d.type == DANGX_MBB || d.type == DANGX_LOGNORMAL)))
            return false;
        const int mode = (d.type == DANGX_POWERLAW) ? CH_POW : (d.type == DANGX_MBB) ? (nind[s] == 0 ? CH_MBB_BETA : CH_MBB_T) : (nind[s] == 0 ? CH_LOGN_NUP : CH_LOGN_W);
        if (sl.n > 0 && sl.s[sl.n - 1].comp == comp[s] && !sl.s[sl.n - 1].pair && sl.s[sl.n - 1].nind + 1 == nind[s] &&
            (sl.s[sl.n - 1].mode == CH_MBB_BETA || sl.s[sl.n - 1].mode == CH_LOGN_NUP)) {
            sl.s[sl.n - 1].pair = 1; sl.s[sl.n - 1].stream2 = stream[s];   // index nind + 1 of the same component: one item
            continue;
        }
        for (int q = 0; q < sl.n; ++q) if (sl.s[q].comp == comp[s]) return false;  // a component's sweeps must be consecutive
        if (sl.n == DX_MAX_SWEEPS) return false;
        SweepItem& it = sl.s[sl.n++];
        it.comp = comp[s]; it.nind = nind[s]; it.mode = mode; it.pair = 0; it.gmember = gm; it.stream = stream[s]; it.stream2 = 0;
    }
    sl.s1 = (map_n == -1) ? 2 : map_n; sl.s2 = (map_n == -1) ? 3 : map_n;
    return true;
}

// a group whose members are the only components with a signal on the planes of the flag: what every plane-set launch needs
static bool planeset_group(dangx_ctx* ctx, const GroupArgs& g) {
    if (!(g.nt == 0 && g.no == 0 && g.nuc == 0)) return false;
    for (int l = 0; l < ctx->hm.ncomp; ++l)
        if (ctx->desc[l].type == DANGX_TCMB || is_global_type(ctx->desc[l].type)) return false;
    return true;
}

// dangx_amp_sample(group, flag, ...) followed by dangx_index_sample(comp[s], nind[s], map_n of the flag, ...) for s = 0 ..
// nsweeps-1 -- everything one iteration of the main loop does on ONE plane set of a CG group: the solve of sample_cg_groups
// (src/dang_cg_mod.f90:166-171) and the passes of sample_spectral_parameters that touch these planes (src/dang_sample_mod.f90:
// 40-75), in the reference's order.  Where k_plane_set covers the model (dx_kern_planeset.h: many bands and members, every swept
// component a member of the group) all of it is ONE launch with the members' SED columns kept in LDS; everything else IS those
// calls (through dangx_amp_index_sample / dangx_index_sample_pair where they apply).
int dangx_plane_set_sample(dangx_ctx* ctx, int group, int flag, int ml_mode, int solver, int fluct_mode, uint64_t seed_amp, uint64_t stream_amp,
                           int i_max, double converge, int nsweeps, const int32_t* comp, const int32_t* nind, const uint64_t* stream,
                           int nsample, uint64_t seed_index, int* cg_iters, int64_t* n_not_spd, int64_t* accepted) {
    DxRange rg_("dangx_plane_set_sample");
    if (!ctx || nsweeps < 1 || !comp || !nind || !stream) return 1;
    (void)hipSetDevice(ctx->device);
    if (sync_model(ctx)) return 1;
    const int map_n = (flag == DANGX_FLAG_T) ? 1 : (flag == DANGX_FLAG_Q) ? 2 : (flag == DANGX_FLAG_U) ? 3 : (flag == DANGX_FLAG_QU) ? -1 : 0;
    if (map_n == 0) return fail(ctx, "flag must be exactly one of T(1), Q(2), U(4), Q+U(8)");
    for (int s = 0; s < nsweeps; ++s)
        if (check_comp(ctx, comp[s]) || nind[s] < 0 || nind[s] >= ctx->desc[comp[s]].nindices) return fail(ctx, "sweep list: component / index out of range");
    if (cg_iters) *cg_iters = 0;
    if (n_not_spd) *n_not_spd = 0;
    static const bool enabled = [] { const char* e = getenv("DANGX_FUSE"); return !(e && e[0] == '0'); }();
    // ---- does the one-launch form cover this?  (the conditions of dangx_amp_index_sample, for every sweep of the list)
    bool can = enabled && nsweeps <= 2 * DX_MAX_SWEEPS && solver == DANGX_SOLVER_DIRECT &&
               (ml_mode == DANGX_ML_OPTIMIZE || fluct_mode == DANGX_FLUCT_REFERENCE) &&
               (ml_mode == DANGX_ML_SAMPLE || ml_mode == DANGX_ML_OPTIMIZE);
    GroupArgs g;
    SweepList sl;
    std::memset(&sl, 0, sizeof(sl));
    if (can) {
        if (make_group(ctx, group, flag, g)) return 1;
        can = planeset_group(ctx, g) && planeset_items(ctx, g, map_n, nsweeps, comp, nind, stream, sl);
    }
    int lanes = 0;
    if (can) {
        sl.nsample = nsample; sl.ml_mode = ml_mode; sl.seed = seed_index;
        g.ml_mode = ml_mode; g.fluct = fluct_mode; g.seed = seed_amp; g.stream = stream_amp;
        lanes = dx_planeset_lanes(ctx, g, sl, 1);
    }
    if (!lanes) {  // the calls this entry point stands for, through the two-step fusions where they apply
        int s = 0;
        int64_t acc = 0, acc2 = 0;
        int rc = dangx_amp_index_sample(ctx, group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, i_max, converge, comp[0], nind[0],
                                        map_n, nsample, seed_index, stream[0], cg_iters, n_not_spd, accepted ? &acc : nullptr);
        if (rc) return rc;
        if (accepted) accepted[0] = acc;
        for (s = 1; s < nsweeps; ++s) {
            if (s + 1 < nsweeps && comp[s + 1] == comp[s] && nind[s + 1] == nind[s] + 1) {
                rc = dangx_index_sample_pair(ctx, comp[s], nind[s], map_n, nsample, ml_mode, seed_index, stream[s], stream[s + 1],
                                             accepted ? &acc : nullptr, accepted ? &acc2 : nullptr);
                if (rc) return rc;
                if (accepted) { accepted[s] = acc; accepted[s + 1] = acc2; }
                ++s;
            } else {
                rc = dangx_index_sample(ctx, comp[s], nind[s], map_n, nsample, ml_mode, seed_index, stream[s], accepted ? &acc : nullptr);
                if (rc) return rc;
                if (accepted) accepted[s] = acc;
            }
        }
        return 0;
    }
    return planeset_launch(ctx, g, sl, lanes, 1, n_not_spd, accepted);
}

// dangx_index_sample(comp[s], nind[s], map_n of the flag, ...) for s = 0 .. nsweeps-1: the passes of sample_spectral_parameters
// (src/dang_sample_mod.f90:40-75) that touch ONE plane set, in the reference's order -- what the two-call seam issues after
// sample_cg_groups has returned.  Where k_plane_set covers the model (every swept component an amplitude-sampled member of ONE CG
// group whose members are the only components on these planes, register-chain modes) the sweeps are one launch on the amplitudes
// in memory (SOLVE = 0: one staging of the maps, the residual kept between the sweeps); everything else IS those calls, with
// consecutive indices of a component through dangx_index_sample_pair.
int dangx_plane_sweeps_sample(dangx_ctx* ctx, int flag, int nsweeps, const int32_t* comp, const int32_t* nind, const uint64_t* stream,
                              int nsample, int ml_mode, uint64_t seed, int64_t* accepted) {
    DxRange rg_("dangx_plane_sweeps_sample");
    if (!ctx || nsweeps < 1 || !comp || !nind || !stream) return 1;
    (void)hipSetDevice(ctx->device);
    if (sync_model(ctx)) return 1;
    const int map_n = (flag == DANGX_FLAG_T) ? 1 : (flag == DANGX_FLAG_Q) ? 2 : (flag == DANGX_FLAG_U) ? 3 : (flag == DANGX_FLAG_QU) ? -1 : 0;
    if (map_n == 0) return fail(ctx, "flag must be exactly one of T(1), Q(2), U(4), Q+U(8)");
    for (int s = 0; s < nsweeps; ++s)
        if (check_comp(ctx, comp[s]) || nind[s] < 0 || nind[s] >= ctx->desc[comp[s]].nindices) return fail(ctx, "sweep list: component / index out of range");
    static const bool enabled = [] { const char* e = getenv("DANGX_FUSE"); return !(e && e[0] == '0'); }();
    bool can = enabled && nsweeps <= 2 * DX_MAX_SWEEPS && (ml_mode == DANGX_ML_SAMPLE || ml_mode == DANGX_ML_OPTIMIZE);
    const int group = ctx->desc[comp[0]].cg_group;
    for (int s = 0; can && s < nsweeps; ++s) can = ctx->desc[comp[s]].cg_group == group && ctx->desc[comp[s]].sample_amplitude;
    GroupArgs g;
    SweepList sl;
    int lanes = 0;
    if (can) {
        if (make_group(ctx, group, flag, g)) return 1;
        if (planeset_group(ctx, g) && planeset_items(ctx, g, map_n, nsweeps, comp, nind, stream, sl)) {
            sl.nsample = nsample; sl.ml_mode = ml_mode; sl.seed = seed;
            g.ml_mode = ml_mode; g.fluct = DANGX_FLUCT_REFERENCE; g.seed = 0; g.stream = 0;
            lanes = dx_planeset_lanes(ctx, g, sl, 0);
        }
    }
    if (lanes) return planeset_launch(ctx, g, sl, lanes, 0, nullptr, accepted);
    int64_t acc = 0, acc2 = 0;
    for (int s = 0; s < nsweeps; ++s) {
        if (s + 1 < nsweeps && comp[s + 1] == comp[s] && nind[s + 1] == nind[s] + 1) {
            const int rc = dangx_index_sample_pair(ctx, comp[s], nind[s], map_n, nsample, ml_mode, seed, stream[s], stream[s + 1],
                                                   accepted ? &acc : nullptr, accepted ? &acc2 : nullptr);
            if (rc) return rc;
            if (accepted) { accepted[s] = acc; accepted[s + 1] = acc2; }
            ++s;
        } else {
            const int rc = dangx_index_sample(ctx, comp[s], nind[s], map_n, nsample, ml_mode, seed, stream[s], accepted ? &acc : nullptr);
            if (rc) return rc;
            if (accepted) accepted[s] = acc;
        }
    }
    return 0;
}

// chi^2 of planes pol_lo..pol_hi from the values fused into the index sweeps: which = 0 -> the state the
// amplitude phase left (captured by the first sweep on each plane), 1 -> the current state.  Fails (status 2)
// if some plane has not been covered by a sweep since its last amplitude update: use dangx_sky_model_chisq.
int dangx_chisq_cached_dev(dangx_ctx* ctx, int which, int pol_lo, int pol_hi, double* out_dev) {
    if (!ctx || !out_dev || (which != 0 && which != 1)) return 1;
    (void)hipSetDevice(ctx->device);
    if (pol_lo < 1 || pol_hi > ctx->dims.nmaps || pol_lo > pol_hi) return fail(ctx, "bad pol_type range");
    for (int k = pol_lo; k <= pol_hi; ++k)
        if (!(which ? ctx->chi_after_valid[k - 1] : ctx->chi_before_valid[k - 1])) {
            ctx->err = "cached chi^2 not available for plane " + std::to_string(k);
            return 2;
        }
    if (chi_flush(ctx)) return 1;
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

static int sky_chisq_launch(dangx_ctx* ctx, int pol_lo, int pol_hi, double* sky_d, double* res_d, double* chi_d, double* out_dev);

// ddata%chisq's sum for the CURRENT state at the least cost: planes whose sum the last sweeps left behind come from the cache,
// every other plane gets one explicit update_sky_model + compute_chisq pass over THAT plane, whose result is cached too (the
// two-call form of the main loop asks after every CG group: only the group's own planes have changed since the last answer)
int dangx_chisq_current(dangx_ctx* ctx, int pol_lo, int pol_hi, double* chisq_sum) {
    DxRange rg_("dangx_chisq_current");
    if (!ctx || !chisq_sum) return 1;
    (void)hipSetDevice(ctx->device);
    if (pol_lo < 1 || pol_hi > ctx->dims.nmaps || pol_lo > pol_hi) return fail(ctx, "bad pol_type range");
    if (sync_model(ctx) || chi_flush(ctx)) return 1;
    for (int k = pol_lo; k <= pol_hi; ++k)
        if (!ctx->chi_after_valid[k - 1]) {
            if (sky_chisq_launch(ctx, k, k, nullptr, nullptr, nullptr, ctx->chi_cache + 3 + (k - 1))) return 1;
            ctx->chi_after_valid[k - 1] = true;
        }
    double v[3] = {0.0, 0.0, 0.0};
    HIPCHK(ctx, hipMemcpyAsync(v, ctx->chi_cache + 3, sizeof(v), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double s = 0.0;
    for (int k = pol_lo; k <= pol_hi; ++k) s += v[k - 1];
    *chisq_sum = s;
    return 0;
}

static int sky_chisq_launch(dangx_ctx* ctx, int pol_lo, int pol_hi, double* sky_d, double* res_d, double* chi_d, double* out_dev) {
    if (pol_lo < 1 || pol_hi > ctx->dims.nmaps || pol_lo > pol_hi) return fail(ctx, "bad pol_type range");
    int bs = 256;
    while (bs > 64 && (size_t)ctx->hm.nbands * bs * sizeof(double) > 32 * 1024) bs >>= 1;
    const unsigned nblk = nblocks(ctx->hm.npix, bs);
    constexpr int RSTAGE = 128;
    // delta bandpasses, diffuse components, no maps asked for: one launch per plane on the amplitude kernel's schedule
    // (k_chisq_reg, dangx_ampreg.hip), the planes' block partials side by side and summed together
    if (!sky_d && !res_d && !chi_d && ctx->hm.all_delta) {
        const unsigned nb256 = nblocks(ctx->hm.npix);
        const int npl = pol_hi - pol_lo + 1;
        if (ensure_partial(ctx, (long long)npl * nb256 + RSTAGE)) return 1;
        bool all = true;
        {
            Timed t(ctx, DANGX_K_SKY_CHISQ);
            for (int k = pol_lo; k <= pol_hi && all; ++k) all = dx_launch_chisq_reg(ctx, k, ctx->partial + (long long)(k - pol_lo) * nb256) == 0;
        }
        if (all) {
            Timed t(ctx, DANGX_K_REDUCE);
            double* stage = ctx->partial + (long long)npl * nb256;
            hipLaunchKernelGGL(k_reduce_rows, dim3(RSTAGE), dim3(BLOCK), 0, ctx->stream, ctx->partial, (long long)npl * nb256, 1, stage);
            hipLaunchKernelGGL(k_reduce, dim3(1), dim3(BLOCK), 0, ctx->stream, stage, (long long)RSTAGE, out_dev);
            HIPCHK(ctx, hipGetLastError());
            return 0;
        }
    }
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
    DxRange rg_("dangx_sky_model_chisq");
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
        const size_t nplanes = (size_t)ctx->dims.nmaps * ctx->dims.nbands;
        if (hipMemcpyAsync(&v, ctx->scalars, sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = 1;
        if (sky && copy_planes(ctx, sky, sky_d, nplanes, false)) rc = 1;
        if (res && copy_planes(ctx, res, res_d, nplanes, false)) rc = 1;
        if (chi_map && copy_planes(ctx, chi_map, chi_d, (size_t)ctx->dims.nmaps, false)) rc = 1;
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
        if (l != comp && ((ctx->plane_nz[l] & ((1u << (s1 - 1)) | (1u << (s2 - 1)))) || ctx->desc[l].type == DANGX_TCMB || is_global_type(ctx->desc[l].type))) others |= 1u << l;
    hipLaunchKernelGGL(k_fullsky_prepare, dim3(nblocks(ctx->hm.npix)), dim3(BLOCK), 0, ctx->stream, ctx->dm, comp, s1, s2, others, ctx->fs_data);
    HIPCHK(ctx, hipGetLastError());
    ctx->fs_comp = comp; ctx->fs_s1 = s1; ctx->fs_s2 = s2; ctx->fs_npc = 0;
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
    const bool coarse = ctx->fs_npc > 0;
    const unsigned nblk = nblocks(coarse ? ctx->fs_npc : ctx->hm.npix);
    if (ensure_partial(ctx, (long long)rows * nblk)) return 1;
    hipLaunchKernelGGL(k_fullsky_rows, dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, ctx->fs_comp, ctx->fs_s1, ctx->fs_s2, what,
                       theta[0], theta[1], coarse ? ctx->cs_data : ctx->fs_data, coarse ? ctx->cs_rms : (const double*)nullptr,
                       coarse ? ctx->cs_mask : (const double*)nullptr, coarse ? ctx->fs_npc : 0ll, ctx->partial);
    hipLaunchKernelGGL(k_reduce_rows_final, dim3(1), dim3(BLOCK), 0, ctx->stream, ctx->partial, (long long)nblk, rows, ctx->rows_out);
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
    if (dangx_fullsky_prepare(ctx, comp, map_n)) return 1;
    const int s1 = ctx->fs_s1, s2 = ctx->fs_s2, Sp = s2 - s1 + 1, nb = ctx->hm.nbands;
    ctx->fs_comp = -1;  // the staging buffer is ours now
    const long long npc = 12LL * sample_nside * sample_nside;
    const int r1 = nside / sample_nside, ratio = r1 * r1;
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
    hipLaunchKernelGGL(k_udgrade, gq, dim3(BLOCK), 0, ctx->stream, ctx->fs_data, ctx->cs_data, ctx->hp_n2r_f, ctx->hp_r2n_c, npix, npc,
                       ratio, 1, 0, scale, 0, nb, ctx->hm.nmaps, s1);
    hipLaunchKernelGGL(k_udgrade, gq, dim3(BLOCK), 0, ctx->stream, ctx->rms, ctx->cs_rms, ctx->hp_n2r_f, ctx->hp_r2n_c, npix, npc,
                       ratio, 1, 1, scale, 1, nb, ctx->hm.nmaps, s1);
    hipLaunchKernelGGL(k_udgrade, g1, dim3(BLOCK), 0, ctx->stream, ctx->mask, ctx->cs_mask, ctx->hp_n2r_f, ctx->hp_r2n_c, npix, npc,
                       ratio, 1, 2, scale, 0, nb, ctx->hm.nmaps, s1);
    HIPCHK(ctx, hipGetLastError());
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


// local (this shard's) sum of c%indices(:, map_n, nind) over unmasked pixels and their number: mask_avg = sum / count
int dangx_index_masked_sum(dangx_ctx* ctx, int comp, int nind, int map_n, double* sum, int64_t* count) {
    if (!ctx || !sum || !count || check_comp(ctx, comp)) return 1;
    (void)hipSetDevice(ctx->device);
    if (nind < 0 || nind >= ctx->desc[comp].nindices) return fail(ctx, "index number out of range");
    if (map_n < 1 || map_n > ctx->dims.nmaps) return fail(ctx, "map_n must be a map number (1..nmaps)");
    if (sync_model(ctx)) return 1;
    const unsigned nblk = nblocks(ctx->hm.npix);
    if (ensure_partial(ctx, 2ll * nblk)) return 1;
    hipLaunchKernelGGL(k_index_masked_sum, dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, comp, nind, map_n, ctx->partial);
    hipLaunchKernelGGL(k_reduce_rows_final, dim3(2), dim3(BLOCK), 0, ctx->stream, ctx->partial, (long long)nblk, 2, ctx->rows_out);
    HIPCHK(ctx, hipGetLastError());
    double out[2] = {0.0, 0.0};
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->rows_out, sizeof(out), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *sum = out[0];
    *count = (int64_t)out[1];
    return 0;
}

// the masked sums of up to 16 index maps in ONE launch and ONE wait (write_stats_to_term prints them all after every phase,
// src/dang_data_mod.f90:540-567; write_data every iteration, :716-731)
int dangx_index_masked_sums(dangx_ctx* ctx, int n, const int32_t* comp, const int32_t* nind, const int32_t* map_n, double* sums, int64_t* counts) {
    if (!ctx || n < 1 || n > 16 || !comp || !nind || !map_n || !sums || !counts) return 1;
    (void)hipSetDevice(ctx->device);
    MeanList ml;
    ml.n = n;
    for (int e = 0; e < 16; ++e) { ml.comp[e] = 0; ml.nind[e] = 0; ml.k[e] = 1; }
    for (int e = 0; e < n; ++e) {
        if (check_comp(ctx, comp[e])) return 1;
        if (nind[e] < 0 || nind[e] >= ctx->desc[comp[e]].nindices) return fail(ctx, "index number out of range");
        if (map_n[e] < 1 || map_n[e] > ctx->dims.nmaps) return fail(ctx, "map_n must be a map number (1..nmaps)");
        ml.comp[e] = comp[e]; ml.nind[e] = nind[e]; ml.k[e] = map_n[e];
    }
    // what is still valid from an earlier call (nothing has written these maps or the mask since) is answered from the host:
    // the statistics after an amplitude phase (src/dang_cg_mod.f90:173) repeat the index means of the phase before; what the
    // last plane-set launch left on the device beside its chi^2 sums (the sums of the maps it swept) costs one small copy
    bool all_known = true, need_copy = false;
    for (int e = 0; e < n; ++e) {
        const int l = comp[e], q = nind[e], k = map_n[e] - 1;
        if (ctx->idx_ext[l] || (!ctx->idxsum_ok[l][q][k] && !(ctx->idxsum_dev[l][q][k] && ctx->mask_count >= 0))) all_known = false;
        else if (!ctx->idxsum_ok[l][q][k]) need_copy = true;
    }
    if (all_known && need_copy) {
        if (chi_flush(ctx)) return 1;
        double host[CHI_CACHE_DOUBLES];
        HIPCHK(ctx, hipMemcpyAsync(host, ctx->chi_cache, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (int l = 0; l < ctx->hm.ncomp; ++l)
            for (int q = 0; q < MAXI; ++q)
                for (int k = 0; k < 3; ++k)
                    if (ctx->idxsum_dev[l][q][k] && !ctx->idxsum_ok[l][q][k]) {
                        ctx->idxsum[l][q][k] = host[idx_slot(l, q, k + 1)];
                        ctx->idxcnt[l][q][k] = ctx->mask_count;
                        ctx->idxsum_ok[l][q][k] = true;
                    }
    }
    if (all_known) {
        for (int e = 0; e < n; ++e) { sums[e] = ctx->idxsum[comp[e]][nind[e]][map_n[e] - 1]; counts[e] = ctx->idxcnt[comp[e]][nind[e]][map_n[e] - 1]; }
        return 0;
    }
    if (sync_model(ctx)) return 1;
    const unsigned nblk = std::min(nblocks(ctx->hm.npix), 4096u);
    if (ensure_partial(ctx, 2ll * n * nblk)) return 1;
    if (ensure_work(ctx, 64)) return 1;
    hipLaunchKernelGGL(k_index_masked_sums, dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, ml, ctx->partial);
    hipLaunchKernelGGL(k_reduce_rows_final, dim3(2 * n), dim3(BLOCK), 0, ctx->stream, ctx->partial, (long long)nblk, 2 * n, ctx->work[0]);
    HIPCHK(ctx, hipGetLastError());
    double out[32];
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->work[0], sizeof(double) * 2 * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int e = 0; e < n; ++e) {
        sums[e] = out[2 * e]; counts[e] = (int64_t)out[2 * e + 1];
        ctx->idxsum[comp[e]][nind[e]][map_n[e] - 1] = sums[e]; ctx->idxcnt[comp[e]][nind[e]][map_n[e] - 1] = counts[e];
        ctx->idxsum_ok[comp[e]][nind[e]][map_n[e] - 1] = !ctx->idx_ext[comp[e]];
    }
    ctx->mask_count = counts[0];   // unmasked pixels of masks(:,1): the same for every map
    return 0;
}

// local sums over every pixel of c%indices(:, map_n, nind) and of masks(:,1) (see k_index_plain_sum)
int dangx_index_plain_sum(dangx_ctx* ctx, int comp, int nind, int map_n, double* sum_index, double* sum_mask) {
    if (!ctx || !sum_index || !sum_mask || check_comp(ctx, comp)) return 1;
    (void)hipSetDevice(ctx->device);
    if (nind < 0 || nind >= ctx->desc[comp].nindices) return fail(ctx, "index number out of range");
    if (map_n < 1 || map_n > ctx->dims.nmaps) return fail(ctx, "map_n must be a map number (1..nmaps)");
    if (sync_model(ctx)) return 1;
    const unsigned nblk = nblocks(ctx->hm.npix);
    if (ensure_partial(ctx, 2ll * nblk)) return 1;
    hipLaunchKernelGGL(k_index_plain_sum, dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->dm, comp, nind, map_n, ctx->partial);
    hipLaunchKernelGGL(k_reduce_rows_final, dim3(2), dim3(BLOCK), 0, ctx->stream, ctx->partial, (long long)nblk, 2, ctx->rows_out);
    HIPCHK(ctx, hipGetLastError());
    double out[2] = {0.0, 0.0};
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->rows_out, sizeof(out), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *sum_index = out[0];
    *sum_mask = out[1];
    return 0;
}

int dangx_unit_conversion(dangx_ctx* ctx, int band, int which, double* out) {
    if (!ctx || !out) return 1;
    if (band < 0 || band >= ctx->dims.nbands || !ctx->band_set[band]) return fail(ctx, "band index out of range / band not set");
    switch (which) {
    case DANGX_A2T: *out = host_a2t(ctx, band); return 0;
    case DANGX_A2F: *out = host_a2f(ctx, band); return 0;
    case DANGX_F2T: *out = host_f2t(ctx, band); return 0;
    default: return fail(ctx, "unit conversion selector must be DANGX_A2T, DANGX_A2F or DANGX_F2T");
    }
}

int dangx_normalize_bandpass(const double* tau_in, int n, double* tau_out) {
    if (!tau_in || !tau_out || n <= 0) return 1;
    double total = 0.0;
    for (int i = 0; i < n; ++i) total += tau_in[i];   // total = sum(tau_in)
    for (int i = 0; i < n; ++i) tau_out[i] = tau_in[i] / total;
    return 0;
}

int dangx_convert_maps(dangx_ctx* ctx, const int32_t* unit, const int32_t* cg_map, double* conversion) {
    if (!ctx || !unit || !conversion) return 1;
    (void)hipSetDevice(ctx->device);
    if (!ctx->sig || !ctx->rms) return fail(ctx, "map data not uploaded");
    const int nb = ctx->dims.nbands;
    for (int j = 0; j < nb; ++j)
        if (!ctx->band_set[j]) return fail(ctx, "band " + std::to_string(j) + " not set");
    const long long plane = (long long)ctx->dims.nmaps * ctx->dims.npix;
    for (int j = 0; j < nb; ++j) {
        if (cg_map && cg_map[j]) continue;  // :435 `if (.not. self%cg_map(j))`: swapped-in maps are converted by convert_cg_maps
        double f;
        if (unit[j] == DANGX_UNIT_UK_RJ) f = 1.0;
        else if (unit[j] == DANGX_UNIT_UK_CMB) f = 1.0 / host_a2t(ctx, j);
        else if (unit[j] == DANGX_UNIT_MJY_SR) f = 1.0 / host_a2f(ctx, j);
        else return fail(ctx, "Not a unit, dumbass! (unit code " + std::to_string(unit[j]) + ")");
        conversion[j] = f;
        hipLaunchKernelGGL(k_scale_band, dim3(nblocks(plane)), dim3(BLOCK), 0, ctx->stream, const_cast<double*>(ctx->sig) + (long long)j * plane,
                           const_cast<double*>(ctx->rms) + (long long)j * plane, plane, f);
        ctx->hm.offset[j] = ctx->hm.offset[j] * f;
        // "Set the loaded monopole values into the monopole component" (:453-457): after EVERY converted band the whole
        // offset vector goes into template_amplitudes(:,1) of every monopole
        for (int l = 0; l < ctx->dims.ncomp; ++l)
            if (ctx->comp_set[l] && ctx->desc[l].type == DANGX_MONOPOLE)
                for (int jj = 0; jj < nb; ++jj) ctx->tamp[l][0][jj] = ctx->hm.offset[jj];
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->dirty = true;
    invalidate_chi(ctx);
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
    idx_written(ctx, comp);
    if (map_n == -1) ctx->qu_equal[comp] |= 1u << nind;
    else if (map_n == 2 || map_n == 3) ctx->qu_equal[comp] &= ~(1u << nind);
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
    n = SN * a.ng + a.nglob;
    if (ensure_work(ctx, n)) return 1;
    return 0;
}

int dangx_compute_rhs(dangx_ctx* ctx, int group, int flag, double* b) {
    if (!ctx || !b) return 1;
    GroupArgs a; long long SN, n;
    if (seam_common(ctx, group, flag, a, SN, n)) return 1;
    if (a.nt ? dx_launch_rhs_mixed(ctx, a, SN, ctx->work[0]) : dx_launch_rhs(ctx, a, SN, ctx->work[0])) return 1;
    HIPCHK(ctx, hipMemcpyAsync(b, ctx->work[0], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dangx_compute_Ax(dangx_ctx* ctx, int group, int flag, const double* x, double* res) {
    if (!ctx || !x || !res) return 1;
    GroupArgs a; long long SN, n;
    if (seam_common(ctx, group, flag, a, SN, n)) return 1;
    HIPCHK(ctx, hipMemcpyAsync(ctx->work[0], x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (a.nt ? dx_launch_Ax_mixed(ctx, a, SN, ctx->work[0], ctx->work[1]) : dx_launch_Ax(ctx, a, SN, ctx->work[0], ctx->work[1], nullptr)) return 1;
    HIPCHK(ctx, hipMemcpyAsync(res, ctx->work[1], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int dangx_compute_sample_vector(dangx_ctx* ctx, int group, int flag, const double* eta, double* res) {
    if (!ctx || !eta || !res) return 1;
    GroupArgs a; long long SN, n;
    if (seam_common(ctx, group, flag, a, SN, n)) return 1;
    HIPCHK(ctx, hipMemcpyAsync(ctx->work[0], eta, sizeof(double) * (size_t)SN, hipMemcpyHostToDevice, ctx->stream));
    if (a.nt ? dx_launch_sv_mixed(ctx, a, SN, ctx->work[0], ctx->work[1]) : dx_launch_sample_vector(ctx, a, SN, ctx->work[0], ctx->work[1])) return 1;
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
    for (int k = 0; k < DANGX_K_COUNT; ++k) {
        ctx->prof_ms[k] = 0.0; ctx->prof_n[k] = 0;
        for (int p = 0; p < 3; ++p) { ctx->prof_ms_pl[k][p] = 0.0; ctx->prof_n_pl[k][p] = 0; }
    }
    return 0;
}
int dangx_profile_get(dangx_ctx* ctx, int kid, double* total_ms, int64_t* launches) {
    if (!ctx || kid < 0 || kid >= DANGX_K_COUNT) return 1;
    if (prof_collect(ctx)) return 1;
    if (total_ms) *total_ms = ctx->prof_ms[kid];
    if (launches) *launches = ctx->prof_n[kid];
    return 0;
}
int dangx_profile_get_planes(dangx_ctx* ctx, int kid, int nplanes, double* total_ms, int64_t* launches) {
    if (!ctx || kid < 0 || kid >= DANGX_K_COUNT || nplanes < 1 || nplanes > 2) return 1;
    if (prof_collect(ctx)) return 1;
    if (total_ms) *total_ms = ctx->prof_ms_pl[kid][nplanes];
    if (launches) *launches = ctx->prof_n_pl[kid][nplanes];
    return 0;
}

}  // extern "C"
