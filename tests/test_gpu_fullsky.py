"""GPU parity for the "next" rows of SURVEY 8f rank 2: full-sky index mode (index_mode == 1), the step-size
tuner and the band-gain fit -- host-side chain (dang_amd/api.py) over device sums vs the oracle's restatement."""
import copy

import numpy as np
import pytest

import dang_amd as da
from dang_amd import _lib as L

import oracle_ffi as O
from util import make_case, pair

pytestmark = pytest.mark.gpu


def _fullsky_case(lnl="chisq", prior="gaussian", tuned=True, config="C2", nside=8, ml_mode="sample"):
    def tweak(dpar, ddata, bands, comps):
        dpar.ml_mode = ml_mode
        for c in comps:
            c.index_mode = [1] * c.nindices
            c.lnl_type = [lnl] * c.nindices
            c.prior_type = [prior] * c.nindices
            c.tuned = [tuned] * max(c.nindices, 1)
            c.step_size = [0.2 * g[1] for g in c.gauss_prior]   # full-sky posteriors are narrow
    return make_case(config, nside=nside, tweak=tweak, start="truth")


@pytest.mark.parametrize("ml_mode", ["sample", "optimize"])
@pytest.mark.parametrize("lnl,prior", [("chisq", "gaussian"), ("marginal", "uniform"), ("chisq", "jeffreys"), ("prior", "gaussian")])
def test_fullsky_index_mode_matches_oracle(built, lnl, prior, ml_mode):
    if lnl == "marginal" and ml_mode == "optimize":
        # the full-sky marginal form -1/2 TNd^2/TNT (src/dang_lnl_mod.f90:113-122) is algebraically independent of
        # the SED (s_j cancels), so `ratio > 1` is the sign of summation noise: no meaningful parity exists
        pytest.skip("degenerate: accept test is the sign of rounding noise")
    case = _fullsky_case(lnl, prior, ml_mode=ml_mode)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if not c.sample_index[j]:
                continue
            f = c.pol_flag[j][0]
            map_n = {1: 1, 8: -1}[f]
            s = da.stream_id(2, 1, l, j, f)
            ag = da.sample_index_mh_fullsky(dpar, ddata, l, j, map_n, s)
            ao, _, _ = orc.sample_index_fullsky(l, j, map_n, dpar.nsample, ml_mode, dpar.seed, s)
            assert ag == ao, (l, j, ag, ao)
            a, b = eng.get_indices(l), orc.indices(l)
            assert np.abs(a - b).max() <= 1e-13
            planes = [0] if f == 1 else [1, 2]
            for k in planes:   # one value for the whole sky, masked pixels included (:329)
                assert np.all(a[j, k] == a[j, k, 0])


def test_fullsky_tuner_matches_oracle(built):
    case = _fullsky_case("chisq", "gaussian", tuned=False, config="C1", nside=8)
    dpar, ddata, bands, comps, meta = case
    for c in comps:
        c.step_size = [2.0 * g[1] for g in c.gauss_prior]       # far too large: the tuner must shrink it
    eng, orc = pair(case)
    s = da.stream_id(2, 1, 0, 0, 1)
    ag = da.sample_index_mh_fullsky(dpar, ddata, 0, 0, 1, s)
    ao, tuned_o, step_o = orc.sample_index_fullsky(0, 0, 1, dpar.nsample, "sample", dpar.seed, s, tuned=False)
    assert tuned_o and all(comps[0].tuned)
    assert comps[0].step_size[0] == step_o and step_o < 2.0 * comps[0].gauss_prior[0][1]
    assert ag == ao
    assert np.abs(eng.get_indices(0) - orc.indices(0)).max() <= 1e-13


def test_sample_spectral_parameters_dispatches_fullsky_and_pixel_modes(built):
    def tweak(dpar, ddata, bands, comps):
        comps[1].index_mode = [1]           # synch beta: one value for the sky; dust stays per-pixel
        comps[1].step_size = [0.01]
        comps[1].tuned = [True]
    case = make_case("C2", nside=8, tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    da.sample_cg_groups(dpar, ddata, it=1)
    for g in dpar.cg_groups:
        for f in g.pol_flag:
            orc.amp_sample_direct(g.cg_group, f, "sample", dpar.seed, da.stream_id(1, 0, g.cg_group, 0, f), "reference")
    da.sample_spectral_parameters(dpar, ddata, it=2)
    mapn = {1: 1, 8: -1}
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if c.sample_index[j]:
                f = c.pol_flag[j][0]
                s = da.stream_id(2, 1, l, j, f)
                if c.index_mode[j] == 1:
                    orc.sample_index_fullsky(l, j, mapn[f], dpar.nsample, "sample", dpar.seed, s)
                else:
                    orc.sample_index_mh(l, j, mapn[f], dpar.nsample, "sample", dpar.seed, s)
    for l, c in enumerate(comps):
        if c.nindices:
            assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-12
    ochisq, _ = orc.chisq(1, 3, ddata.nump)
    assert abs(ddata.chisq - ochisq) <= 1e-9 * ochisq


@pytest.mark.parametrize("ml_mode", ["optimize", "sample"])
def test_fit_band_gain_matches_oracle(built, ml_mode):
    case = make_case("C2", nside=8, start="truth", gain=[1.0, 1.03, 0.98, 1.0, 1.05])
    dpar, ddata, bands, comps, meta = case
    dpar.ml_mode = ml_mode
    ddata.gain = np.ones(5)                # the run starts from gain 1 and must find the injected gains
    ddata.fit_gain = [False, True, True, False, True]
    eng, orc = pair(case)
    for j in (1, 2, 4):
        go = orc.fit_band_gain(j, ml_mode, dpar.seed, da.stream_id(3, 2, 0, 0, 0))
        gg = da.fit_band_gain(dpar, ddata, j, it=3)
        assert abs(gg - go) <= 1e-12 * abs(go)
        if ml_mode == "optimize":
            assert abs(gg - [1.0, 1.03, 0.98, 1.0, 1.05][j]) < 2e-3
        orc.gain[j] = go                   # ddata%gain(band) = gain (:619)
    assert da.sample_calibrators(dpar, ddata, it=4)


@pytest.mark.parametrize("ml_mode", ["sample", "optimize"])
def test_perpixel_branch_tunes_the_step_size_like_the_reference(built, ml_mode):
    """The 'Tuning!' block of the per-pixel branch (src/dang_sample_mod.f90:341-346): when c%tuned(nind) is false the
    tuner's sky-wide chain runs first (start: sum(indices)/sum(mask) over every pixel), the step size it leaves is the
    one the per-pixel chains then use.  Step size, tuned flags and the sweep itself against the oracle."""
    def tweak(dpar, ddata, bands, comps):
        dpar.ml_mode = ml_mode
        comps[1].step_size = [2.0]            # synch beta (one index): far too large a step for the sky-wide chain
        comps[1].tuned = [False]
    case = make_case("C2", nside=8, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for c in comps:                           # only the synchrotron index, on the T plane
        c.sample_index = [False] * c.nindices
    comps[1].sample_index = [True]
    step0 = comps[1].step_size[0]
    s = da.stream_id(3, 1, 1, 0, L.FLAG_T)
    otuned, ostep = orc.tune_perpixel(1, 0, 1, dpar.nsample, ml_mode, dpar.seed, s)
    oacc = orc.sample_index_mh(1, 0, 1, dpar.nsample, ml_mode, dpar.seed, s)
    info = da.sample_spectral_parameters(dpar, ddata, it=3)
    # (optimize mode: once the chain sits at the optimum nothing is accepted any more and the step only ever halves --
    # the reference's `do while (.not. c%tuned(nind))` would not return; both restatements stop after 64 rounds)
    assert comps[1].tuned == [otuned] and (otuned or ml_mode == "optimize")
    assert comps[1].step_size[0] == ostep and ostep < step0, (comps[1].step_size, ostep)
    assert info == [(1, 0, L.FLAG_T, oacc)]
    assert np.abs(eng.get_indices(1) - orc.indices(1)).max() <= 1e-12
    if otuned:   # a second call finds the index tuned: no tuner pass, same step
        da.sample_spectral_parameters(dpar, ddata, it=4)
        assert comps[1].step_size[0] == ostep


@pytest.mark.parametrize("lnl,prior,ml_mode", [("chisq", "gaussian", "sample"), ("chisq", "jeffreys", "sample"),
                                               ("marginal", "uniform", "sample"), ("chisq", "gaussian", "optimize")])
def test_fullsky_index_mode_at_a_coarser_nside_matches_oracle(built, lnl, prior, ml_mode):
    """index_mode == 1 with sample_nside /= nside (src/dang_sample_mod.f90:199-217, 229-329): the chain's sky-wide sums
    run over the degraded data / rms / mask (Nside 8 -> 2), eval_signal reads the full-resolution amplitude at the coarse
    pixel number (reproduced literally, as in the per-pixel coarse mode); through the orchestrator's dispatch."""
    nside, cnside = 8, 2
    case = _fullsky_case(lnl, prior, ml_mode=ml_mode, nside=nside)
    dpar, ddata, bands, comps, meta = case
    for c in comps:
        c.sample_nside = [cnside] * c.nindices
        c.step_size = [0.6 * g[1] for g in c.gauss_prior]        # 48 coarse pixels: a broader posterior than the full sky's
    eng, orc = pair(case)
    info = da.sample_spectral_parameters(dpar, ddata, it=2)
    k = 0
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if not c.sample_index[j]:
                continue
            f = c.pol_flag[j][0]
            map_n = {1: 1, 8: -1}[f]
            ao, _, _ = orc.sample_index_fullsky_coarse(l, j, map_n, dpar.nsample, ml_mode, dpar.seed, da.stream_id(2, 1, l, j, f),
                                                       nside, cnside)
            assert ao >= 0 and info[k] == (l, j, f, ao), (info[k], ao)
            k += 1
    assert k == len(info) and np.isfinite(ddata.chisq)
    for l, c in enumerate(comps):
        if not c.nindices:
            continue
        a, b = eng.get_indices(l), orc.indices(l)
        assert np.abs(a - b).max() <= 1e-13
        for j in range(c.nindices):
            if c.sample_index[j]:
                for kk in ([0] if c.pol_flag[j][0] == 1 else [1, 2]):
                    assert np.all(a[j, kk] == a[j, kk, 0])
