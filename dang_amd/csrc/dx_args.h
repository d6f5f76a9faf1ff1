// dx_args.h -- launch-argument structs and constants shared by host and device code (no host-only types: this header is
// also compiled by hiprtc when a kernel is specialised at run time, dangx_rtc.hip).
#pragma once
#include "dx_model.h"
#include "dx_rng.h"
#include "dx_sed.h"

using namespace dx;

constexpr int BLOCK = 256;

struct GroupArgs {
    int ng;          // sampled diffuse components of the group
    int gc[MAXG];    // their component indices, in component_list order
    int no;          // components NOT solved for (removed from the data)
    int oc[MAXC];
    int flag;        // one poltype bit
    int ml_mode, fluct;
    unsigned long long seed, stream;
    // global-amplitude members of the group (template / monopole / hi_fit), after the diffuse ones in x:
    // x = [diffuse: ng blocks of S*npix | global: nglob entries], component t owns rows trow[t] .. trow[t]+nfit-1
    int nt, nglob;
    int tc[MAXT], trow[MAXT];
    // every template / monopole of the model (any group): bands with corr == false are removed from the data
    // in compute_rhs (src/dang_cg_mod.f90:445-460)
    int nuc, uc[MAXC];
};

__device__ __forceinline__ int flag_nplanes(int flag) { return (flag & DANGX_FLAG_QU) ? 2 : 1; }
// src/dang_cg_mod.f90:357-363 and the flag-8 branches (:488-494): plane p -> map number
__device__ __forceinline__ int flag_map(int flag, int p) {
    if (flag & DANGX_FLAG_QU) return 2 + p;
    if (flag & DANGX_FLAG_T) return 1;
    if (flag & DANGX_FLAG_Q) return 2;
    return 3;
}

// chain modes of the Metropolis kernels (see dangx_mh.hip)
enum { CH_GENERIC = 0, CH_POW = 1, CH_MBB_BETA = 2, CH_MBB_T = 3, CH_LOGN_NUP = 4, CH_LOGN_W = 5 };

struct IndexArgs {
    int comp, nind, s1, s2, nsample, ml_mode, mode;
    int bp;           // some band is bandpass-integrated or a non-diffuse component is present: the compile-time chain
                      // modes then sum over the bandpass samples and remove the other components through comp_signal
    unsigned others;  // bit l: component l (/= comp) may have a non-zero amplitude on planes s1..s2
    unsigned long long seed, stream;
};

// dangx_plane_set_sample: the index sweeps that follow a group's solve on its planes, in the reference's order (dx_kern_planeset.h)
struct SweepItem {
    int comp, nind;   // component and index (0-based) of the sweep
    int mode;         // chain mode of the sweep (CH_POW ... CH_LOGN_W)
    int pair;         // 1: index nind + 1 of the same component follows in the same item (mode + 1)
    int gmember;      // the component's position among the group's members
    int pad;
    unsigned long long stream, stream2;  // random streams of the sweep (and of the paired one)
};
constexpr int DX_MAX_SWEEPS = 6;
constexpr int DX_MAX_IDXSUM = 8;   // index maps one plane-set launch can sweep (4 items, each with its component's next index)
struct SweepList {
    int n, nsample, ml_mode, s1, s2, pad;
    unsigned long long seed;
    SweepItem s[DX_MAX_SWEEPS];
};

