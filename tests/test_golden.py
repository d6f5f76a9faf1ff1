"""Golden-vector tests (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle;
see that file for provenance: they are NOT reference outputs).  CPU leg: the oracle still reproduces
them.  GPU leg: the HIP path reproduces them through the C ABI."""
import glob
import os

import numpy as np
import pytest

import dang_amd as da
from dang_amd import _lib as L
from dang_amd import synth

import oracle_ffi as O
from util import MAPN, relmax

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GIBBS = sorted(glob.glob(os.path.join(GOLD, "gibbs_*.npz")))
SEAMS = sorted(glob.glob(os.path.join(GOLD, "seams_*.npz")))


def _case(g, **kw):
    dpar, ddata, bands, comps, meta = synth.make_sky(str(g["config"]), nside=int(g["nside"]), **kw)
    # the fixture's arrays are the inputs (the generator only supplies the descriptors)
    ddata.sig_map, ddata.rms_map, ddata.masks = g["sig"], g["rms"], g["mask"]
    return dpar, ddata, bands, comps, meta


def _run_gibbs(dpar, ddata, comps, meta, niter, amp, idx, chisq):
    trace = []
    for it in range(1, niter + 1):
        for grp in dpar.cg_groups:
            for f in grp.pol_flag:
                amp(grp.cg_group, f, dpar.ml_mode, dpar.seed, da.stream_id(it, 0, grp.cg_group, 0, f))
        trace.append(chisq())
        if it > 1:
            for l, c in enumerate(comps):
                for j in range(c.nindices):
                    if c.sample_index[j]:
                        for f in c.pol_flag[j]:
                            idx(l, j, MAPN[f], dpar.nsample, dpar.ml_mode, dpar.seed, da.stream_id(it, 1, l, j, f))
            trace.append(chisq())
    return np.array(trace)


@pytest.mark.parametrize("path", GIBBS, ids=[os.path.basename(p) for p in GIBBS])
def test_oracle_reproduces_golden_gibbs(path):
    g = np.load(path)
    dpar, ddata, bands, comps, meta = _case(g)
    orc = O.Oracle(bands, comps, ddata)
    trace = _run_gibbs(dpar, ddata, comps, meta, int(g["niter"]),
                       lambda grp, f, ml, seed, s: orc.amp_sample_direct(grp, f, ml, seed, s, dpar.fluct_mode),
                       orc.sample_index_mh, lambda: orc.chisq(1, meta["nmaps"], ddata.nump)[0])
    assert relmax(trace, g["chisq_trace"]) <= 1e-12
    for l, c in enumerate(comps):
        assert relmax(orc.amplitude(l), g["end_amp_%d" % l]) <= 1e-11
        if c.nindices:
            assert np.abs(orc.indices(l) - g["end_idx_%d" % l]).max() <= 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("path", GIBBS, ids=[os.path.basename(p) for p in GIBBS])
def test_hip_reproduces_golden_gibbs(built, path):
    g = np.load(path)
    dpar, ddata, bands, comps, meta = _case(g)
    eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
    trace = _run_gibbs(dpar, ddata, comps, meta, int(g["niter"]),
                       lambda grp, f, ml, seed, s: eng.amp_sample(grp, f, ml, seed, s, fluct_mode=dpar.fluct_mode),
                       eng.index_sample, lambda: da.compute_chisq(ddata))
    assert relmax(trace, g["chisq_trace"]) <= 1e-8    # chi^2 after each phase
    for l, c in enumerate(comps):
        assert relmax(eng.get_amplitude(l), g["end_amp_%d" % l]) <= 1e-9   # fraction of max|a|
        if c.nindices:
            assert np.abs(eng.get_indices(l) - g["end_idx_%d" % l]).max() <= 1e-10


@pytest.mark.parametrize("path", SEAMS, ids=[os.path.basename(p) for p in SEAMS])
def test_oracle_reproduces_golden_seams(path):
    g = np.load(path)
    dpar, ddata, bands, comps, meta = _case(g, start="truth")
    orc = O.Oracle(bands, comps, ddata)
    for name, group, flag in (("T", 1, L.FLAG_T), ("QU", 2, L.FLAG_QU)):
        if "x_" + name not in g:
            continue
        assert relmax(orc.compute_rhs(group, flag), g["rhs_" + name]) <= 1e-13
        assert relmax(orc.compute_Ax(group, flag, g["x_" + name]), g["Ax_" + name]) <= 1e-13
        assert relmax(orc.compute_sample_vector(group, flag, g["eta_" + name]), g["sv_" + name]) <= 1e-13
        assert np.array_equal(orc.draw_eta(flag, 5, 6), g["eta_" + name])
    sky, res = orc.sky_model()
    assert relmax(sky, g["sky"]) <= 1e-14 and relmax(res, g["res"]) <= 1e-12
    assert abs(orc.chisq(1, meta["nmaps"], ddata.nump, sky)[0] - g["chisq"]) <= 1e-13 * g["chisq"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", SEAMS, ids=[os.path.basename(p) for p in SEAMS])
def test_hip_reproduces_golden_seams(built, path):
    g = np.load(path)
    dpar, ddata, bands, comps, meta = _case(g, start="truth")
    eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
    for l in range(len(comps)):
        for j in range(meta["nbands"]):
            for k in range(meta["nmaps"]):
                assert np.abs(eng.eval_sed(l, j, k + 1) / g["sed"][l, j, k] - 1).max() <= 2e-13
    for name, group, flag in (("T", 1, L.FLAG_T), ("QU", 2, L.FLAG_QU)):
        if "x_" + name not in g:
            continue
        assert relmax(eng.compute_rhs(group, flag), g["rhs_" + name]) <= 1e-13
        assert relmax(eng.compute_Ax(group, flag, g["x_" + name]), g["Ax_" + name]) <= 1e-13
        assert relmax(eng.compute_sample_vector(group, flag, g["eta_" + name]), g["sv_" + name]) <= 1e-13
    s, sky, res, chi = eng.sky_model_chisq(1, meta["nmaps"], want_maps=True)
    assert relmax(sky, g["sky"]) <= 1e-13 and relmax(res, g["res"]) <= 1e-11 and relmax(chi, g["chi_map"]) <= 1e-11
    assert abs(s / meta["nbands"] / float(g["nump"]) - g["chisq"]) <= 1e-10 * g["chisq"]
    for name, group, flag in (("T", 1, L.FLAG_T), ("QU", 2, L.FLAG_QU)):
        if "x_" + name not in g:
            continue
        it, _ = eng.amp_sample(group, flag, "sample", 5, 6, solver="cg", i_max=100, converge=1e-8)
        # the stopping test `delta_new > converge` is a comparison of two numbers: in the C2 fixture the Q+U run passes
        # delta = 1.0217e-8 at iteration 88 (the oracle's trace), 2 % above the threshold, and 88 iterations of CG
        # amplify last-bit differences of eta to about that size -- one iteration either way is the same algorithm
        assert abs(it - int(g["cg_iters_" + name])) <= 1
        # a CG run that stopped at i_max (not converged; 6-component C5 blocks are very ill-conditioned)
        # is a rounding-sensitive trajectory: compare loosely there, tightly when it converged
        tol = (1e-6 if it == int(g["cg_iters_" + name]) else 1e-5) if it < 100 else 5e-2
        for l, c in enumerate(comps):
            if c.cg_group == group:
                assert relmax(eng.get_amplitude(l), g["cg_amp_%s_%d" % (name, l)]) <= tol


# ---- the "next" rows: template group operators, full-sky index mode, coarse-Nside sampling

PATHS = os.path.join(GOLD, "paths_C2_nside8.npz")


def _paths_case():
    import sys
    sys.path.insert(0, GOLD)
    from make_golden import paths_tweak
    g = np.load(PATHS)
    dpar, ddata, bands, comps, meta = synth.make_sky("C2", nside=int(g["nside"]), start="truth")
    paths_tweak(dpar, ddata, bands, comps)
    # the generator + tweak reproduce the fixture's inputs (to the last bits of the host's libm); the fixture's arrays
    # are the inputs
    assert np.allclose(ddata.sig_map, g["sig"], rtol=1e-12, atol=1e-9)
    ddata.sig_map, ddata.rms_map, ddata.masks = g["sig"], g["rms"], g["mask"]
    return g, dpar, ddata, bands, comps, meta


def test_oracle_reproduces_golden_paths():
    g, dpar, ddata, bands, comps, meta = _paths_case()
    orc = O.Oracle(bands, comps, ddata)
    assert relmax(orc.compute_rhs(2, L.FLAG_QU), g["rhs"]) <= 1e-13
    assert relmax(orc.compute_Ax(2, L.FLAG_QU, g["x"]), g["Ax"]) <= 1e-13
    assert relmax(orc.compute_sample_vector(2, L.FLAG_QU, g["eta"]), g["sv"]) <= 1e-13
    assert orc.amp_sample_cg(2, L.FLAG_QU, "sample", 5, 6, i_max=600, converge=1e-8) == int(g["cg_iters"])
    # (an unconverged CG trajectory amplifies the last-bit differences of another host's libm in the start state)
    assert relmax(orc.template_amplitudes(len(comps) - 1), g["cg_ta"]) <= 1e-6
    o2 = O.Oracle(bands, comps, ddata)
    acc, _, _ = o2.sample_index_fullsky(1, 0, 1, 10, "sample", 7, da.stream_id(2, 1, 1, 0, 1))
    assert acc == int(g["fullsky_acc"]) and np.abs(o2.indices(1) - g["fullsky_idx"]).max() <= 1e-13
    o3 = O.Oracle(bands, comps, ddata)
    acc = o3.sample_index_mh_coarse(5, 0, -1, 10, "sample", 7, da.stream_id(2, 1, 5, 0, 8), int(g["nside"]), 2)
    assert acc == int(g["coarse_acc"]) and np.abs(o3.indices(5) - g["coarse_idx"]).max() <= 1e-13


@pytest.mark.gpu
def test_hip_reproduces_golden_paths(built):
    g, dpar, ddata, bands, comps, meta = _paths_case()
    import copy
    eng = da.initialize(bands, copy.deepcopy(comps), ddata, npix_global=meta["npix_global"], device=0)
    assert relmax(eng.compute_rhs(2, L.FLAG_QU), g["rhs"]) <= 1e-12
    assert relmax(eng.compute_Ax(2, L.FLAG_QU, g["x"]), g["Ax"]) <= 1e-12
    assert relmax(eng.compute_sample_vector(2, L.FLAG_QU, g["eta"]), g["sv"]) <= 1e-12
    it, _ = eng.amp_sample(2, L.FLAG_QU, "sample", 5, 6, solver="cg", i_max=600, converge=1e-8)
    assert it == int(g["cg_iters"])
    assert relmax(eng.get_template_amplitudes(len(comps) - 1), g["cg_ta"]) <= 1e-6
    assert relmax(eng.get_amplitude(3), g["cg_amp3"]) <= 1e-6
    eng = da.initialize(bands, copy.deepcopy(comps), ddata, npix_global=meta["npix_global"], device=0)
    comps_fs = eng.component_list
    comps_fs[1].index_mode = [1]
    dpar.seed = 7
    acc = da.sample_index_mh_fullsky(dpar, ddata, 1, 0, 1, da.stream_id(2, 1, 1, 0, 1))
    assert acc == int(g["fullsky_acc"]) and np.abs(eng.get_indices(1) - g["fullsky_idx"]).max() <= 1e-12
    eng = da.initialize(bands, copy.deepcopy(comps), ddata, npix_global=meta["npix_global"], device=0)
    acc = eng.index_sample_coarse(5, 0, -1, 10, "sample", 7, da.stream_id(2, 1, 5, 0, 8), 2)
    assert acc == int(g["coarse_acc"]) and np.abs(eng.get_indices(5) - g["coarse_idx"]).max() <= 1e-12
