// dx_kern_planeset.h -- k_plane_set (template): ONE launch for what a Gibbs iteration does on one plane set (T, or Q+U) of a CG
// group without global-amplitude members: the group's amplitude solve (SOLVE), then the index sweeps on these planes in the
// reference's order (sample_cg_groups, src/dang_cg_mod.f90:166-171, followed by the passes of sample_spectral_parameters that
// touch these planes, src/dang_sample_mod.f90:40-75).  Three uses of the one template:
//   SOLVE = 1, sweep items      the whole iteration of the plane set (dangx_plane_set_sample; bench.py, gibbs_iteration_gpu)
//   SOLVE = 1, no items         the amplitude phase alone, with chi^2 of the state it leaves as a by-product: what the
//                               two-call seam's sample_cg_groups needs for its statistics (src/dang_cg_mod.f90:172-173) without a
//                               pass of its own (dangx_amp_sample)
//   SOLVE = 0, sweep items      the index phase alone on the amplitudes in memory (dangx_plane_sweeps_sample: the two-call
//                               seam's sample_spectral_parameters)
//
// Resident residual.  The members' SED columns are evaluated once into the lane's LDS column; the solve uses them; then the
// lane forms the FULL residual (d - sum over every member) / rms of its bands in registers and parks 1 / rms in its LDS column
// (the SED columns are dead by then).  A sweep's cleaned data (src/dang_sample_mod.f90:173-196: data_raw minus every OTHER
// component) are residual + own signal: the chain's first likelihood evaluation, which needs the member's SED at the current
// index values anyway, adds it back (RegChain::lnl, LNL_ADD), and after the chain the signal at the values it ended on is taken
// out again (LNL_SUB).  Nothing is read from HBM twice, no reciprocal of the rms is formed twice, and "every other component"
// is removed once instead of once per sweep.  Against re-staging the maps for every sweep (round 3's form of this kernel) the
// cleaned data differ by the rounding of one more addition (|d| eps): same proposals, an accept decision could differ only
// where |diff - ln u| is of that size (none in 3.8e8 decisions of a full-size run), chi^2 sums agree to rounding -- the parity
// tolerance of the index maps.  Same device, tools/ab_bench.sh: C5 126.6 -> 112.5 ms per iteration, C3 11.7 -> 11.1 ms, C2 +4 %.
//
// Requirements checked by the launcher (dangx_planeset.hip): delta bands, direct solver with the reference fluctuation term,
// every swept component an amplitude-sampled member of the group with a register-chain mode (chisq likelihood, gaussian /
// uniform prior), no other component on the planes, and -- for Q+U -- index maps that are equal on the two planes for every
// member that varies (true once a Q+U sweep has written them, :465; tracked on the host).  Band calibration: the T launch reads
// gain and offset from the block's table (solve: d / gain, src/dang_cg_mod.f90:371; chains and chi^2: (d - offset) / gain,
// src/dang_sample_mod.f90:174, src/dang_data_mod.f90:384); Q and U are never rescaled.
#pragma once
#include "dx_kern_fused.h"

// resident waves per SIMD the register allocation aims at: three for small shapes (<= 5 bands per lane), two otherwise
#ifndef DX_PS_WAVES
#define DX_PS_WAVES(SP, NB, LP, SOLVE, C0) (((NB) / (LP) <= 5) ? 3 : 2)
#endif

namespace dxk {

template <int V> struct ItemCode { static constexpr int value = V; };
template <bool V> struct ItemFlag { static constexpr bool value = V; };

// one item of the sweep list for member it.gmember: chain (and the paired chain of index nind + 1, as index_chain_pair does);
// R0.D holds the full residual / rms of the lane's bands, rows 0 .. SP*NBL-1 of the lane's LDS column hold 1 / rms.  Leaves the
// member's two index values in sample0 / sample1.  LAST: nothing follows, the residual need not be restored.
template <int MODE, int PAIR, int SP, int NBL, int LP, int NG, bool FIRST, bool LAST, bool BP, typename RFirst>
__device__ __forceinline__ void ps_item(const Model& M, const SweepList& sl, const SweepItem& it, RFirst& R0, int i, int half,
                                        int jb, int NB, const double* __restrict__ tab, const double* __restrict__ col,
                                        double& sample0, double& sample1, double chi_first[4],
                                        double chi_last[4], unsigned int* __restrict__ accepted, int slot, const double* first_acc) {
    const Comp& c = M.comp[it.comp];
    const BandPick<LP> pick = {half};
    const int npix = M.npix;
    IndexArgs a;
    a.comp = it.comp; a.nind = it.nind; a.s1 = sl.s1; a.s2 = sl.s2; a.nsample = sl.nsample; a.ml_mode = sl.ml_mode; a.mode = MODE;
    a.bp = 0; a.others = 0u; a.seed = sl.seed; a.stream = it.stream;
    // per-band constants from the block's table in LDS, in the one-lane form too (as scalar operands from the model instead: 86.4
    // against 87.1 it/s at C3 on one device -- the scalar registers are what the kernel is short of)
    RegChain<MODE, SP, NBL, LP, true, BP> R;
    R.set_kt(tab, NB, NG, it.gmember, jb);
#pragma unroll
    for (int kk = 0; kk < SP; ++kk) {
        R.amp[kk] = c.amp[(long long)(sl.s1 + kk - 1) * npix + i];   // the lane's own store after the solve, or the map in memory
#pragma unroll
        for (int j = 0; j < NBL; ++j) { R.D[kk][j] = R0.D[kk][j]; R.ISr[kk][j] = col[(kk * NBL + j) * BLOCK] * R.amp[kk]; }
    }
    double chia[4] = {0.0, 0.0, 0.0, 0.0}, va, acc[2];
    // FIRST: the kernel has put this member's signal back already, from its SED column of the solve (R0.D is the cleaned data and
    // first_acc the likelihood sums of the state the solve left): no evaluation before the first proposal.  Later items: the
    // chain's first evaluation adds the signal back (LNL_ADD)
    unsigned long long na = chain_finish<MODE, SP, NBL, LP, false, !FIRST, (!PAIR && !LAST)>(M, a, c, R, pick, sample0, sample1, i, half, chia, &va,
                                                                                            FIRST ? first_acc : nullptr, acc);
    if (it.nind == 0) sample0 = va; else sample1 = va;
    if (FIRST) { chi_first[0] = chia[0]; chi_first[1] = chia[1]; }
    chi_last[2] = chia[2]; chi_last[3] = chia[3];
    unsigned long long nb_ = 0ull;
    if (PAIR) {
        constexpr int MODEB = (MODE == CH_MBB_BETA || MODE == CH_LOGN_NUP) ? MODE + 1 : MODE;  // only those two modes have a pair
        RegChain<MODEB, SP, NBL, LP, true, BP> RB;
        RB.set_kt(tab, NB, NG, it.gmember, jb);
#pragma unroll
        for (int kk = 0; kk < SP; ++kk) {
            RB.amp[kk] = R.amp[kk];
#pragma unroll
            for (int j = 0; j < NBL; ++j) { RB.D[kk][j] = R.D[kk][j]; RB.ISr[kk][j] = R.ISr[kk][j]; }
        }
        IndexArgs b = a;
        b.nind = it.nind + 1; b.stream = it.stream2; b.mode = MODEB;
        double chib[4] = {0.0, 0.0, 0.0, 0.0}, vb;
        // (the second chain starts from the state the first one ended on: its likelihood sums are that chain's, not evaluated again)
        nb_ = chain_finish<MODEB, SP, NBL, LP, false, false, !LAST>(M, b, c, RB, pick, sample0, sample1, i, half, chib, &vb, acc);
        if (b.nind == 0) sample0 = vb; else sample1 = vb;
        chi_last[2] = chib[2]; chi_last[3] = chib[3];
        if (!LAST) {
#pragma unroll
            for (int kk = 0; kk < SP; ++kk)
#pragma unroll
                for (int j = 0; j < NBL; ++j) R0.D[kk][j] = RB.D[kk][j];
        }
    } else if (!LAST) {
#pragma unroll
        for (int kk = 0; kk < SP; ++kk)
#pragma unroll
            for (int j = 0; j < NBL; ++j) R0.D[kk][j] = R.D[kk][j];
    }
    if (accepted) {  // per-sweep counters ([slot], [slot + 1] for the paired sweep), a diagnostic output: summed over the block in LDS
        // (this code runs inside the pixel's live branch: no cross-lane reduction here), 16 copies per counter so that the lanes of a
        // wave meet in fours; one global atomic per block and counter at the kernel's end.  One atomic per lane straight to memory
        // queued ~20 M same-address atomics behind the launch (+35 % on its time); one LDS word per counter, also when nobody asks,
        // costs 5 % (64 lanes on one address, 90.7 against 85.9 it/s on one device)
        const int copy = threadIdx.x & 15;
        if (na) atomicAdd(accepted + (slot << 4) + copy, (unsigned int)na);
        if (nb_) atomicAdd(accepted + ((slot + 1) << 4) + copy, (unsigned int)nb_);
    }
}

// C0 .. C3: the sweep items of the launch, compile-time: chain mode (CH_POW, CH_MBB_BETA, CH_LOGN_NUP ...) + 8 when index nind + 1 of
// the same component follows in the same item (mbb: beta then T; log-normal: nu_p then w), 0 = no item.  A run-time switch over
// the modes inside one kernel costs the register allocator ~200 spills (three inlined chains share one frame); a model's
// sweep sequence is fixed for a run, so it is part of the specialisation: C5 = <POW, MBB_BETA + 8, LOGN_NUP> is built in, any
// other sequence is compiled on first use (dangx_rtc.hip).
// BP: some bands are bandpass-integrated (LP == 1): the members' columns take eval_sed's sample sums, the chains loop over the
// samples (RegChain<.., BP>); delta-only models keep BP = 0 and their instruction count.
template <int SP, int NB, int NG, int LP, int SOLVE, int C0, int C1, int C2, int C3, int BP = 0>
__global__ __launch_bounds__(BLOCK, DX_PS_WAVES(SP, NB, LP, SOLVE, C0)) void k_plane_set(const Model* __restrict__ Mp, GroupArgs ga, FusedArgs fa, SweepList sl,
                                                        unsigned long long* __restrict__ not_spd, unsigned long long* __restrict__ accepted,
                                                        double* __restrict__ chi_partial) {
    constexpr int NBL = NB / LP;
    extern __shared__ double lds[];  // [constant table | per-lane column: max(nv, SP) * NBL rows -- the SEDs, then 1 / rms]
    const Model& M = *Mp;
    const int npix = M.npix, tid = threadIdx.x;
    double* tab = lds;
    double* col = lds + (TROWS * NG + 3) * NB + tid;
    const long long t0 = (long long)blockIdx.x * BLOCK + tid;
    const long long u = t0 / LP;
    const int half = (int)(t0 % LP), jb = half * NBL;
    const bool in_range = u < npix;
    const int i = in_range ? (int)u : 0;
    const double mk = M.mask[i];
    sed_table_build(M, tab, tid, BLOCK, ga.gc, NG);
    __shared__ unsigned int acc_blk[DX_MAX_IDXSUM * 16];   // accepted proposals of the block, per counter of the sweep list (16 copies)
    if (tid < DX_MAX_IDXSUM * 16) acc_blk[tid] = 0u;
    __syncthreads();
    double chi[4] = {0.0, 0.0, 0.0, 0.0};
    // masked sums of the index maps the launch sweeps (mask_avg's numerator, src/dang_util_mod.f90:186-206: what write_stats_to_term
    // prints after the phase): the value each chain ends on, by-products like chi^2 -- rows 4 .. of chi_partial
    constexpr int NS0 = C0 ? 1 + (C0 >> 3) : 0, NS1 = C1 ? 1 + (C1 >> 3) : 0, NS2 = C2 ? 1 + (C2 >> 3) : 0, NS3 = C3 ? 1 + (C3 >> 3) : 0;
    constexpr int NS = NS0 + NS1 + NS2 + NS3;
    const bool live = in_range && !is_masked(mk);
    if (in_range && !live && half == 0) {  // masked: x stays (:695); every swept index map gets a zero (:223, :480-483)
        for (int q = 0; q < sl.n; ++q) {
            const Comp& c = M.comp[sl.s[q].comp];
            for (int e = 0; e <= sl.s[q].pair; ++e) {
                double* out = c.idx + ((long long)(sl.s[q].nind + e) * M.nmaps) * npix + i;
#pragma unroll
                for (int kk = 0; kk < SP; ++kk) out[(long long)(sl.s1 + kk - 1) * npix] = 0.0;
            }
        }
    }
    if (live) {
        const long long bstride = (long long)M.nmaps * npix;
        const unsigned long long gpix = (unsigned long long)(M.pix0 + i);
        const bool sample = (ga.ml_mode == DANGX_ML_SAMPLE);
        // ---- SED columns of the varying members, once: their indices are equal on the planes of the launch (launcher)
#pragma unroll 1
        for (int v = 0; v < fa.nv; ++v) {
            const Comp& c2 = M.comp[ga.gc[fa.vcomp[v]]];
            double t0v, t1v;
            load_theta(M, c2, i, sl.s1, t0v, t1v);
            if (BP) sed_column_bp<NBL>(M, c2, tab, NB, NG, fa.vcomp[v], sed_prep(c2, t0v, t1v), col + (v * NBL) * BLOCK);
            else sed_column<NBL>(fa.vtype[v], tab, NB, NG, fa.vcomp[v], jb, sed_prep(c2, t0v, t1v), col + (v * NBL) * BLOCK);
        }
        RegChain<CH_POW, SP, NBL, LP> R0;  // storage for the maps of the plane set: d and 1/sigma, then the residual
#pragma unroll
        for (int kk = 0; kk < SP; ++kk) {
            const int k = sl.s1 + kk;
            {
                const double* sigp = M.sig + (long long)(k - 1) * npix + i;
                const double* rmsp = M.rms + (long long)(k - 1) * npix + i;
#pragma unroll
                for (int j = 0; j < NBL; ++j) { R0.D[kk][j] = sigp[(jb + j) * bstride]; R0.ISr[kk][j] = rmsp[(jb + j) * bstride]; }
            }
            auto cal_transform = [&]() {  // data_raw = (sig - offset) / gain on the temperature plane (:174)
                const double* gn = tab + (TROWS * NG + 1) * NB + jb;
#pragma unroll
                for (int j = 0; j < NBL; ++j) R0.D[kk][j] = (R0.D[kk][j] - gn[NB + j]) / gn[j];
            };
            if (!SOLVE && SP == 1 && fa.cal) cal_transform();
            // `template` components with a signal on these planes (the launcher's list ga.uc): eval_signal = template_amplitudes(band,
            // map) * template(pix, map) (src/dang_component_mod.f90:754-776), taken out of the data first.  For the sweeps it is one
            // more "other component"; for a solve (the back-substitution of a template group's Schur solve, dangx_sky_plane_set_sample:
            // the group's templates with their NEW amplitudes) it is compute_rhs' data minus the global members' share of A x
            // (k_amp_reg's HT form, dangx_ampreg.hip) -- the same data either way
#pragma unroll 1
            for (int t = 0; t < ga.nuc; ++t) {
                const Comp& ct = M.comp[ga.uc[t]];
                const double tv = ct.tmpl[(long long)(k - 1) * npix + i];
#pragma unroll
                for (int j = 0; j < NBL; ++j) R0.D[kk][j] -= ct.tamp[k - 1][jb + j] * tv;
            }
            double bv[NG];
            if (SOLVE) {
                // ---- the block solve of unit (i, k) (k_amp_index's), amplitudes stored for the sweeps
                double eta = 0.0, f0 = 0.0;
                if (sample) {
                    double u1, u2;
                    uniform2(ga.seed, ga.stream, gpix, (uint32_t)k, u1, u2);
                    eta = rand_normal(0.0, 1.0, u1, u2);
                }
                double A[NG * (NG + 1) / 2];
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
                    double d = R0.D[kk][j];
                    if (SP == 1 && fa.cal) d = d / tab[(TROWS * NG + 1) * NB + jb + j];  // T / gain, no offset (:371)
                    const double is = fast_rcp(R0.ISr[kk][j]);
                    R0.set_is(kk, j, is);
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
                    if (j % DX_FUSED_GRP == DX_FUSED_GRP - 1) __builtin_amdgcn_sched_barrier(0);
                }
                if (LP > 1) {
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
                    // both lanes of a pair store (the same values): each lane later re-reads only what it wrote itself
#pragma unroll
                    for (int g = 0; g < NG; ++g) M.comp[ga.gc[g]].amp[(long long)(k - 1) * npix + i] = bv[g];
                } else {
                    if (half == 0) atomicAdd(not_spd, 1ull);  // x keeps its value: the sweeps run on the old amplitudes
#pragma unroll
                    for (int g = 0; g < NG; ++g) bv[g] = M.comp[ga.gc[g]].amp[(long long)(k - 1) * npix + i];
                }
            } else {  // the index phase alone: the amplitudes of the last solve
#pragma unroll
                for (int g = 0; g < NG; ++g) bv[g] = M.comp[ga.gc[g]].amp[(long long)(k - 1) * npix + i];
#pragma unroll
                for (int j = 0; j < NBL; ++j) R0.set_is(kk, j, fast_rcp(R0.ISr[kk][j]));
            }
            // ---- the plane's full residual in units of the rms: data_raw (:173-177) minus EVERY member in component_list order
            // (:180-196 removes all but the sampled one; its own signal returns in the chain's first evaluation).  Member by member,
            // each through ONE base address and compile-time offsets (a run-time stride costs one address register per load: the
            // solve's band loop above pays that)
            if (SOLVE && SP == 1 && fa.cal) cal_transform();
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const double amp2 = bv[g];
                if (fa.vslot[g] >= 0) {
                    const double* m = col + (fa.vslot[g] * NBL) * BLOCK;
#pragma unroll
                    for (int j = 0; j < NBL; ++j) R0.D[kk][j] -= amp2 * m[j * BLOCK];
                } else {
                    const double* m = tab + (TROWS * g + 2 + k) * NB + jb;
#pragma unroll
                    for (int j = 0; j < NBL; ++j) R0.D[kk][j] -= amp2 * m[j];
                }
            }
#pragma unroll
            for (int j = 0; j < NBL; ++j) {
                R0.D[kk][j] *= R0.ISr[kk][j];
                // pinned here: the optimiser otherwise sinks this arithmetic below the next plane's solve (its result is first used
                // by the chains) while the column reads stay, and NG * NBL loaded values are spilled across that solve
                asm volatile("" : "+v"(R0.D[kk][j]));
            }
        }
        double first_acc[2] = {0.0, 0.0};
        if (C0 == 0) {  // no sweep follows: chi^2 of the state the solve leaves is the residual's (both the "before" and "after" slots)
#pragma unroll
            for (int kk = 0; kk < SP; ++kk) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < NBL; ++j) acc = fma(R0.D[kk][j], R0.D[kk][j], acc);
                if (LP > 1) acc += __shfl_xor(acc, 1, 64);
                if (half == 0) { chi[kk] = acc; chi[2 + kk] = acc; }
            }
        } else {
            // the first sweep's cleaned data: residual + the swept member's own signal, from its SED column of the solve (the index
            // values have not moved since), and the likelihood sums of the state the solve left = the residual's
            {
                const SweepItem it0 = sl.s[0];
                const double* m = col + (fa.vslot[it0.gmember] * NBL) * BLOCK;
#pragma unroll
                for (int kk = 0; kk < SP; ++kk) {
                    const double a0 = M.comp[it0.comp].amp[(long long)(sl.s1 + kk - 1) * npix + i];
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < NBL; ++j) {
                        const double r = R0.D[kk][j];
                        acc = fma(r, r, acc);
                        R0.D[kk][j] = fma(a0 * R0.ISr[kk][j], m[j * BLOCK], r);
                    }
                    if (LP > 1) acc += __shfl_xor(acc, 1, 64);
                    first_acc[kk] = -0.5 * acc;
                }
            }
            // 1 / rms of the lane's bands -> its LDS column (the SED columns are dead from here on)
#pragma unroll
            for (int kk = 0; kk < SP; ++kk)
#pragma unroll
                for (int j = 0; j < NBL; ++j) col[(kk * NBL + j) * BLOCK] = R0.ISr[kk][j];
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the sweeps of the plane set, in the reference's order
        int slot = 0;
        auto run = [&](auto code_tag, auto first_tag, auto last_tag, int q) {
            constexpr int CODE = decltype(code_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value;
            if constexpr (CODE != 0) {
                const SweepItem it = sl.s[q];
                const Comp& c = M.comp[it.comp];
                double sample0, sample1, unused[4];
                // a component's sweeps are consecutive and travel in ONE item: no lane reads here what its partner wrote
                load_theta(M, c, i, sl.s1, sample0, sample1);
                ps_item<(CODE & 7), (CODE >> 3), SP, NBL, LP, NG, FIRST, LAST, (BP != 0)>(M, sl, it, R0, i, half, jb, NB, tab, col, sample0, sample1,
                                                                              FIRST ? chi : unused, chi, accepted ? acc_blk : nullptr, slot, first_acc);
                slot += 1 + (CODE >> 3);
            }
        };
        run(ItemCode<C0>{}, ItemFlag<true>{}, ItemFlag<C1 == 0>{}, 0);
        run(ItemCode<C1>{}, ItemFlag<false>{}, ItemFlag<C2 == 0>{}, 1);
        run(ItemCode<C2>{}, ItemFlag<false>{}, ItemFlag<C3 == 0>{}, 2);
        run(ItemCode<C3>{}, ItemFlag<false>{}, ItemFlag<true>{}, 3);
    }
    if (accepted && NS > 0) {
        __syncthreads();
        if (tid < NS) {
            unsigned int v = 0u;
            for (int q = 0; q < 16; ++q) v += acc_blk[(tid << 4) + q];
            if (v != 0u) atomicAdd(accepted + tid, (unsigned long long)v);
        }
    }
    if (chi_partial) {
        // (the values the chains ended on are read back from the index maps -- this lane's own stores -- rather than carried in
        // registers across the chains: the two-plane kernels have none to spare)
        double isum[NS > 0 ? NS : 1];
#pragma unroll
        for (int q = 0; q < (NS > 0 ? NS : 1); ++q) isum[q] = 0.0;
        if (live && half == 0) {
            constexpr int code[4] = {C0, C1, C2, C3};
            constexpr int first[4] = {0, NS0, NS0 + NS1, NS0 + NS1 + NS2};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (code[q] != 0) {
                    const double* at = M.comp[sl.s[q].comp].idx + ((long long)sl.s[q].nind * M.nmaps + (sl.s1 - 1)) * npix + i;
                    isum[first[q]] = at[0];
                    if ((code[q] >> 3) != 0) isum[first[q] + 1] = at[(long long)M.nmaps * npix];
                }
        }
        __shared__ double sh[4 + NS][BLOCK / 64];
#pragma unroll
        for (int q = 0; q < 4 + NS; ++q) {
            double v = (q < 4) ? chi[q < 4 ? q : 0] : isum[q >= 4 ? q - 4 : 0];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if ((tid & 63) == 0) sh[q][tid >> 6] = v;
        }
        __syncthreads();
        if (tid < 4 + NS) {
            double s = 0.0;
            for (int w = 0; w < BLOCK / 64; ++w) s += sh[tid][w];
            chi_partial[(long long)tid * gridDim.x + blockIdx.x] = s;
        }
    }
}

}  // namespace dxk
