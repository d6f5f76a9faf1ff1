#!/usr/bin/env python3
"""Generate tests/golden/*.npz: inputs + expected outputs for the hot path.

PROVENANCE: these vectors are produced by the CPU oracle (oracle/dang_oracle.c), NOT by the
reference program: hermda02/dang has no tests or golden vectors of its own and cannot be built in
this image (it needs the HEALPix-F90 / CFITSIO / MPI Fortran modules).  They are regression
fixtures that freeze the oracle's answers (so a change to the oracle or to the HIP path is
noticed) and let the GPU tests run against plain data.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]

import oracle_ffi as O  # noqa: E402
from dang_amd import stream_id, synth  # noqa: E402
from dang_amd import _lib as L  # noqa: E402

MAPN = {1: 1, 2: 2, 4: 3, 8: -1}


def state(orc, comps):
    out = {}
    for l, c in enumerate(comps):
        out["amp_%d" % l] = orc.amplitude(l).copy()
        if c.nindices:
            out["idx_%d" % l] = orc.indices(l).copy()
    return out


def gibbs_case(name, config, nside, niter, **kw):
    dpar, ddata, bands, comps, meta = synth.make_sky(config, nside=nside, **kw)
    orc = O.Oracle(bands, comps, ddata)
    out = dict(sig=ddata.sig_map, rms=ddata.rms_map, mask=ddata.masks, nump=ddata.nump, niter=niter,
               config=config, nside=nside)
    for k, v in state(orc, comps).items():
        out["start_" + k] = v
    chis = []
    for it in range(1, niter + 1):
        for g in dpar.cg_groups:
            for f in g.pol_flag:
                orc.amp_sample_direct(g.cg_group, f, dpar.ml_mode, dpar.seed, stream_id(it, 0, g.cg_group, 0, f), dpar.fluct_mode)
        chis.append(orc.chisq(1, meta["nmaps"], ddata.nump)[0])
        if it > 1:
            for l, c in enumerate(comps):
                for j in range(c.nindices):
                    if c.sample_index[j]:
                        for f in c.pol_flag[j]:
                            orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, dpar.ml_mode, dpar.seed, stream_id(it, 1, l, j, f))
            chis.append(orc.chisq(1, meta["nmaps"], ddata.nump)[0])
    for k, v in state(orc, comps).items():
        out["end_" + k] = v
    out["chisq_trace"] = np.array(chis)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "chisq trace", chis)


def seams_case(name, config, nside):
    dpar, ddata, bands, comps, meta = synth.make_sky(config, nside=nside, start="truth")
    orc = O.Oracle(bands, comps, ddata)
    out = dict(sig=ddata.sig_map, rms=ddata.rms_map, mask=ddata.masks, config=config, nside=nside)
    rng = np.random.default_rng(7)
    for gname, group, flag in (("T", 1, L.FLAG_T), ("QU", 2, L.FLAG_QU)):
        if meta["nmaps"] == 1 and gname == "QU":
            continue
        n = orc.group_size(group, flag)
        x = rng.standard_normal(n)
        eta = orc.draw_eta(flag, 5, 6)
        out.update({"x_" + gname: x, "eta_" + gname: eta, "rhs_" + gname: orc.compute_rhs(group, flag),
                    "Ax_" + gname: orc.compute_Ax(group, flag, x), "sv_" + gname: orc.compute_sample_vector(group, flag, eta)})
        o2 = O.Oracle(bands, comps, ddata)
        it = o2.amp_sample_cg(group, flag, "sample", 5, 6, i_max=100, converge=1e-8)
        out["cg_iters_" + gname] = it
        for l, c in enumerate(comps):
            if c.cg_group == group:
                out["cg_amp_%s_%d" % (gname, l)] = o2.amplitude(l).copy()
    sed = np.zeros((len(comps), meta["nbands"], meta["nmaps"], meta["npix"]))
    for l in range(len(comps)):
        for j in range(meta["nbands"]):
            for k in range(meta["nmaps"]):
                sed[l, j, k] = orc.eval_sed_map(l, j, k + 1)
    out["sed"] = sed
    sky, res = orc.sky_model()
    chisq, chi = orc.chisq(1, meta["nmaps"], ddata.nump, sky)
    out.update(sky=sky, res=res, chi_map=chi, chisq=chisq, nump=ddata.nump)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "chisq", chisq)


def paths_tweak(dpar, ddata, bands, comps):
    """The model of the 'paths' fixture: C2 + a polarisation template fitted at bands 3, 4 in group 2."""
    from test_oracle_templates_cpu import add_globals
    add_globals(dpar, ddata, bands, comps, ("template",), 2, fit_bands=[3, 4])


def paths_case(name, nside=8):
    """The "next" rows: a template group (mixed operators), the full-sky index mode, coarse-Nside sampling."""
    dpar, ddata, bands, comps, meta = synth.make_sky("C2", nside=nside, start="truth")
    paths_tweak(dpar, ddata, bands, comps)
    out = dict(sig=ddata.sig_map, rms=ddata.rms_map, mask=ddata.masks, nump=ddata.nump, nside=nside, config="C2")
    orc = O.Oracle(bands, comps, ddata)
    rng = np.random.default_rng(9)
    n = orc.group_size(2, L.FLAG_QU)
    x, eta = rng.standard_normal(n), orc.draw_eta(L.FLAG_QU, 5, 6)
    out.update(x=x, eta=eta, rhs=orc.compute_rhs(2, L.FLAG_QU), Ax=orc.compute_Ax(2, L.FLAG_QU, x),
               sv=orc.compute_sample_vector(2, L.FLAG_QU, eta))
    it = orc.amp_sample_cg(2, L.FLAG_QU, "sample", 5, 6, i_max=600, converge=1e-8)
    out.update(cg_iters=it, cg_ta=orc.template_amplitudes(len(comps) - 1).copy(), cg_amp3=orc.amplitude(3).copy())
    o2 = O.Oracle(bands, comps, ddata)
    acc, _, _ = o2.sample_index_fullsky(1, 0, 1, 10, "sample", 7, stream_id(2, 1, 1, 0, 1))
    out.update(fullsky_acc=acc, fullsky_idx=o2.indices(1).copy())
    o3 = O.Oracle(bands, comps, ddata)
    acc = o3.sample_index_mh_coarse(5, 0, -1, 10, "sample", 7, stream_id(2, 1, 5, 0, 8), nside, 2)
    out.update(coarse_acc=acc, coarse_idx=o3.indices(5).copy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "cg iters", it, "fullsky acc", out["fullsky_acc"], "coarse acc", out["coarse_acc"])


if __name__ == "__main__":
    paths_case("paths_C2_nside8")
    gibbs_case("gibbs_C1_nside8", "C1", 8, 3)
    gibbs_case("gibbs_C2_nside4", "C2", 4, 3)
    gibbs_case("gibbs_C5_nside2", "C5", 2, 2)
    seams_case("seams_C2_nside4", "C2", 4)
    seams_case("seams_C5_nside2", "C5", 2)
