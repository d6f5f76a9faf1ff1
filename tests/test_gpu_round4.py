"""Round-4 additions (one test per review item, each against the oracle or an invariant of the path)."""
import copy
import os

import numpy as np
import pytest

import dang_amd as da
from dang_amd import _lib as L

import oracle_ffi as O
from util import MAPN, make_case, pair, shard_engines

pytestmark = pytest.mark.gpu


def test_non_spd_count_over_several_contexts_starts_from_zero(built):
    """dangx_sky_amp_sample on a diffuse group over three contexts: the count of non-SPD pixel blocks ('left unchanged', what the
    Fortran two-call path prints per group) is THIS call's, not the sum of every earlier launch on the contexts -- two calls
    give the same count, and it is what the three contexts report one by one."""
    def tweak(dpar, ddata, bands, comps):
        # the dust set of the T group becomes a second synchrotron with the same index everywhere: identical SED columns, every
        # block of group 1 exactly singular
        comps[2].type, comps[2].nu_ref, comps[2].nindices = comps[1].type, comps[1].nu_ref, 1
        comps[2].indices = comps[2].indices[:1].copy()
        for q in ("ind_label", "sample_index", "index_mode", "lnl_type", "prior_type", "gauss_prior", "uni_prior", "step_size", "pol_flag"):
            setattr(comps[2], q, copy.deepcopy(getattr(comps[1], q)))
        comps[1].indices[:] = -3.0
        comps[2].indices[:] = -3.0
    case = make_case("C2", nside=4, start="truth", tweak=tweak)
    engs = shard_engines(case, 3)
    _, bad1 = da.sky_amp_sample(engs, 1, L.FLAG_T, "optimize", 8, 9)
    _, bad2 = da.sky_amp_sample(engs, 1, L.FLAG_T, "optimize", 8, 9)
    each = [e.amp_sample(1, L.FLAG_T, "optimize", 8, 9)[1] for e in engs]
    assert bad1 > 0 and bad2 == bad1 == sum(each), (bad1, bad2, each)
    # a regular group on the same contexts afterwards: zero, whatever the counters held
    _, bad3 = da.sky_amp_sample(engs, 2, L.FLAG_QU, "optimize", 8, 9)
    assert bad3 == 0


@pytest.mark.parametrize("config,nside", [("C3", 8), ("C2", 8), ("C5", 4)])
def test_plane_set_launches_stay_under_band_calibration(built, config, nside):
    """Band gains /= 1 and offsets /= 0 (the state sample_calibrators and a fitted monopole leave in ddata%gain / ddata%offset):
    the solve takes T / gain without the offset (src/dang_cg_mod.f90:371), the chains and chi^2 (T - offset) / gain
    (src/dang_sample_mod.f90:174, src/dang_data_mod.f90:384), Q and U are never rescaled -- quirk 6.  Both plane sets must still be
    ONE launch each (profile: nothing but the plane-set bucket and the chi^2 reductions), and the loop -- gibbs_iteration, then
    sample_calibrators as src/dang.f90:101-111 orders them -- must match the oracle's loop in the reference's order."""
    nb = {"C2": 5, "C3": 10, "C5": 20}[config]
    gain = [1.0 + 0.01 * ((j % 3) - 1) for j in range(nb)]
    offset = [0.5 * ((j % 4) - 1.5) for j in range(nb)]
    case = make_case(config, nside=nside, start="truth", gain=gain, offset=offset)
    dpar, ddata, bands, comps, meta = case
    ddata.fit_gain = [(j % 2 == 1) for j in range(nb)]
    eng, orc = pair(case)
    ddata.gain = np.array(ddata.gain, dtype=np.float64)     # the oracle keeps its own copy of the calibration
    for it in (2, 3):
        eng.profile(True)
        da.gibbs_iteration(dpar, ddata, it)
        prof = eng.profile_get()
        eng.profile(False)
        assert set(prof) <= {"k_amp_index", "k_reduce"} and prof["k_amp_index"]["launches"] == 2, prof
        for g in dpar.cg_groups:
            orc.amp_sample_direct(g.cg_group, g.pol_flag[0], "sample", dpar.seed, da.stream_id(it, 0, g.cg_group, 0, g.pol_flag[0]), "reference")
        for l, c in enumerate(comps):
            for j in range(c.nindices):
                if c.sample_index[j]:
                    f = c.pol_flag[j][0]
                    orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, "sample", dpar.seed, da.stream_id(it, 1, l, j, f))
        ochisq, _ = orc.chisq(1, 3, ddata.nump)
        assert abs(ddata.chisq - ochisq) <= 1e-9 * ochisq, (it, ddata.chisq, ochisq)
        # sample_calibrators (:487-518): the fitted bands' gains move, the next iteration runs on them
        assert da.sample_calibrators(dpar, ddata, it=it)
        for j in range(nb):
            if ddata.fit_gain[j]:
                go = orc.fit_band_gain(j, "sample", dpar.seed, da.stream_id(it, 2, 0, 0, 0))
                assert abs(ddata.gain[j] - go) <= 1e-11 * abs(go), (it, j, ddata.gain[j], go)
                orc.gain[j] = go
    for l in range(len(comps)):
        a, b = eng.get_amplitude(l), orc.amplitude(l)
        assert np.abs(a - b).max() <= 1e-9 * max(np.abs(b).max(), 1e-30), l
        if comps[l].nindices:
            assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-12, l


def _plane_sweep_list(comps, group, flag, it):
    return [(l, j, da.stream_id(it, 1, l, j, flag)) for l, c in enumerate(comps) for j in range(c.nindices)
            if c.cg_group == group and c.sample_index[j] and flag in c.pol_flag[j]]


@pytest.mark.parametrize("config,nside", [("C3", 8), ("C2", 8), ("C5", 4), ("C1", 8)])
def test_two_call_seam_launches_match_the_oracle(built, config, nside):
    """The two halves of the two-call seam (src/dang.f90:101, 106) as they now run: dangx_amp_sample = the plane-set kernel without
    sweep items, whose residual gives chi^2 of the state the solve leaves (src/dang_cg_mod.f90:172-173) without a pass of its own;
    dangx_plane_sweeps_sample = the sweeps of a plane set in one launch on the amplitudes in memory.  Against the oracle's solves
    and sweeps (amplitudes 1e-9, indices 1e-12, accepted counts equal, chi^2 1e-10 -- also against the explicit pass), and the
    profile shows no chi^2 pass and no stand-alone sweep launch."""
    case = make_case(config, nside=nside, start="truth")
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    nmaps = meta["nmaps"]
    for it in (2, 3):
        eng.profile(True)
        for g in dpar.cg_groups:
            f = g.pol_flag[0]
            s = da.stream_id(it, 0, g.cg_group, 0, f)
            _, bad = eng.amp_sample(g.cg_group, f, "sample", dpar.seed, s)
            assert bad == orc.amp_sample_direct(g.cg_group, f, "sample", dpar.seed, s, "reference")
        chi_amp = eng.chisq_current(1, nmaps)                   # by-product of the solves: no launch but the reductions
        prof = eng.profile_get()
        assert "k_sky_chisq" not in prof and prof["k_amp_direct"]["launches"] == len(dpar.cg_groups), prof
        eng.profile(False)
        ochi, _ = orc.chisq(1, nmaps, 1.0)
        assert abs(chi_amp / meta["nbands"] - ochi) <= 1e-10 * ochi, (it, chi_amp / meta["nbands"], ochi)
        assert abs(eng.sky_model_chisq(1, nmaps) - chi_amp) <= 1e-10 * chi_amp     # the explicit pass agrees
        eng.profile(True)
        for g in dpar.cg_groups:
            f = g.pol_flag[0]
            sw = _plane_sweep_list(comps, g.cg_group, f, it)
            acc = eng.plane_sweeps_sample(f, sw, dpar.nsample, "sample", dpar.seed)
            for (l, j, st), a in zip(sw, acc):
                assert a == orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, "sample", dpar.seed, st), (it, l, j)
        prof = eng.profile_get()
        eng.profile(False)
        assert prof["k_index_mh"]["launches"] == len(dpar.cg_groups), prof       # one launch per plane set
        before, after = eng.chisq_cached(0, 1, nmaps), eng.chisq_cached(1, 1, nmaps)
        assert abs(before - chi_amp) <= 1e-10 * chi_amp                          # what the first sweeps saw = what the solves left
        ochi, _ = orc.chisq(1, nmaps, 1.0)
        assert abs(after / meta["nbands"] - ochi) <= 1e-10 * ochi
    for l in range(len(comps)):
        a, b = eng.get_amplitude(l), orc.amplitude(l)
        assert np.abs(a - b).max() <= 1e-9 * max(np.abs(b).max(), 1e-30), l
        if comps[l].nindices:
            assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-12, l


def test_masked_index_sums_are_cached_until_a_map_changes(built):
    """dangx_index_masked_sums (write_stats_to_term's index means, src/dang_data_mod.f90:540-567, asked for after EVERY phase):
    answered from the host while nothing has written the maps -- an amplitude phase does not -- and recomputed after a sweep, a
    host push or a new mask."""
    case = make_case("C2", nside=8, start="truth")
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    entries = [(l, j, k) for l, c in enumerate(comps) for j in range(c.nindices) for k in ((1,) if c.cg_group == 1 else (2, 3))]
    def want():
        m = np.asarray(ddata.masks)[0] != 0
        return [float(eng.get_indices(l)[j, k - 1][m].sum()) for l, j, k in entries]
    s0, c0 = eng.index_masked_sums(entries)
    assert np.allclose(s0, want(), rtol=1e-13)
    eng.profile(True)
    eng.amp_sample(1, L.FLAG_T, "sample", dpar.seed, 5)
    s1, c1 = eng.index_masked_sums(entries)                  # nothing wrote an index map: no launch
    assert np.array_equal(s0, s1) and np.array_equal(c0, c1)
    n_before = sum(v["launches"] for v in eng.profile_get().values())
    eng.index_sample(1, 0, 1, 5, "sample", dpar.seed, 7)     # synchrotron beta on T moves
    n_sweep = sum(v["launches"] for v in eng.profile_get().values())
    s2, _ = eng.index_masked_sums(entries)
    assert np.allclose(s2, want(), rtol=1e-13) and s2[entries.index((1, 0, 1))] != s1[entries.index((1, 0, 1))]
    new = eng.get_indices(2).copy()
    new[0, 0] += 0.25
    eng.put_indices(2, new)                                  # a host push
    s3, _ = eng.index_masked_sums(entries)
    assert np.allclose(s3, want(), rtol=1e-13) and s3[entries.index((2, 0, 1))] != s2[entries.index((2, 0, 1))]
    eng.profile(False)
    assert n_sweep > n_before
    # the plane-set launches leave the sums of the maps they swept beside their chi^2 sums (by-products of the launch): the next
    # request is answered from them -- same sums to rounding, the mask's pixel count
    ddata.engine = eng
    da.gibbs_iteration(dpar, ddata, 3)
    s4, c4 = eng.index_masked_sums(entries)
    assert np.allclose(s4, want(), rtol=1e-13) and np.array_equal(c4, c0) and not np.array_equal(s4, s3)
    for g in dpar.cg_groups:    # ... the sweeps-only launch too
        eng.plane_sweeps_sample(g.pol_flag[0], _plane_sweep_list(comps, g.cg_group, g.pol_flag[0], 4), dpar.nsample, "sample", dpar.seed)
    s5, c5 = eng.index_masked_sums(entries)
    assert np.allclose(s5, want(), rtol=1e-13) and np.array_equal(c5, c0) and not np.array_equal(s5, s4)
    # a model change between the launch and the request (band calibration here) drops the launch's pending partials -- chi^2 and
    # index sums alike: the request must not read a stale slot
    da.gibbs_iteration(dpar, ddata, 5)
    eng.set_calibration(np.ones(meta["nbands"]), np.zeros(meta["nbands"]))
    s5b, c5b = eng.index_masked_sums(entries)
    assert np.allclose(s5b, want(), rtol=1e-13) and np.array_equal(c5b, c0) and not np.array_equal(s5b, s5)
    s5 = s5b
    # a context that has never counted the mask's pixels takes the explicit pass and gets the same numbers
    eng2, _ = pair(case)
    for l in range(len(comps)):
        if comps[l].nindices:
            eng2.put_indices(l, eng.get_indices(l))
    s6, c6 = eng2.index_masked_sums(entries)
    assert np.allclose(s6, s5, rtol=1e-13) and np.array_equal(c6, c0)


@pytest.mark.parametrize("config,nside,ml_mode", [("C3", 8, "sample"), ("C2", 8, "sample"), ("C3", 4, "optimize")])
def test_bandpass_integrated_bands_run_the_plane_set_kernel(built, config, nside, ml_mode):
    """bp%id /= 'delta' on every second band (src/dang_component_mod.f90:909-913, 949-954; one empty bandpass row, which the
    reference skips): the whole iteration of each plane set is still ONE launch -- the register chain with a run-time sample loop
    (k_plane_set<.., BP = 1>, specialised on first use) -- and three iterations match the oracle's loop in the reference's order:
    amplitudes 1e-9, indices 1e-12, accepted counts through the index maps, chi^2 1e-9; the two-call halves
    (dangx_amp_sample / dangx_plane_sweeps_sample) run the same kernel family."""
    def tweak(dpar, ddata, bands, comps):
        rng = np.random.default_rng(4)
        for b in bands[1::2]:
            nu = b.nu_c * 1e9 * np.linspace(0.9, 1.1, 9)
            tau = rng.uniform(0.2, 1.0, nu.size)
            nu[3] = 0.0
            b.id, b.nu0, b.tau0 = "bp", nu, tau / tau.sum()
        dpar.ml_mode = ml_mode
    case = make_case(config, nside=nside, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    nb, ng = meta["nbands"], len(meta["phys"])
    for it in (2, 3, 4):
        if it < 4:
            eng.profile(True)
            da.gibbs_iteration(dpar, ddata, it)
            prof = eng.profile_get()
            eng.profile(False)
            assert set(prof) <= {"k_amp_index", "k_reduce"} and prof["k_amp_index"]["launches"] == 2, prof
        else:   # the two-call halves
            for g in dpar.cg_groups:
                eng.amp_sample(g.cg_group, g.pol_flag[0], ml_mode, dpar.seed, da.stream_id(it, 0, g.cg_group, 0, g.pol_flag[0]))
            for g in dpar.cg_groups:
                eng.plane_sweeps_sample(g.pol_flag[0], _plane_sweep_list(comps, g.cg_group, g.pol_flag[0], it), dpar.nsample, ml_mode, dpar.seed)
            da.compute_chisq(ddata)
        for g in dpar.cg_groups:
            orc.amp_sample_direct(g.cg_group, g.pol_flag[0], ml_mode, dpar.seed, da.stream_id(it, 0, g.cg_group, 0, g.pol_flag[0]), "reference")
        for l, c in enumerate(comps):
            for j in range(c.nindices):
                if c.sample_index[j]:
                    f = c.pol_flag[j][0]
                    orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, ml_mode, dpar.seed, da.stream_id(it, 1, l, j, f))
        ochisq, _ = orc.chisq(1, 3, ddata.nump)
        assert abs(ddata.chisq - ochisq) <= 1e-9 * ochisq, (it, ddata.chisq, ochisq)
    names = eng.rtc_kernels()
    for sp in (1, 2):
        assert "dxk::k_plane_set<%d, %d, %d, 1, 1, 1, 10, 0, 0, 1>" % (sp, nb, ng) in names, names    # iteration
        assert "dxk::k_plane_set<%d, %d, %d, 1, 1, 0, 0, 0, 0, 1>" % (sp, nb, ng) in names, names     # the solve alone
        assert "dxk::k_plane_set<%d, %d, %d, 1, 0, 1, 10, 0, 0, 1>" % (sp, nb, ng) in names, names    # the sweeps alone
    for l in range(len(comps)):
        a, b = eng.get_amplitude(l), orc.amplitude(l)
        assert np.abs(a - b).max() <= 1e-9 * max(np.abs(b).max(), 1e-30), l
        if comps[l].nindices:
            assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-12, l


def _template_case(config, nside, which="template"):
    from test_oracle_templates_cpu import add_globals

    def tweak(dpar, ddata, bands, comps):
        nb = ddata.sig_map.shape[0]
        if which == "template":
            add_globals(dpar, ddata, bands, comps, ("template",), 2, fit_bands=[nb - 2, nb - 1])
        else:   # a monopole in the T group, fitted at three bands (beside the CMB its rows are nearly degenerate: the checked solve)
            add_globals(dpar, ddata, bands, comps, ("monopole",), 1, fit_bands=[0, nb - 2, nb - 1])
    return make_case(config, nside=nside, start="truth", tweak=tweak)


def _copy_group_state(eng, orc, comps, group):
    """The state the GPU's Schur solve left -> the oracle (whose direct solve covers per-pixel members only)."""
    for l, c in enumerate(comps):
        if c.cg_group != group:
            continue
        if c.type in ("template", "monopole"):
            orc.template_amplitudes(l)[:] = eng.get_template_amplitudes(l)
            if c.type == "monopole":   # update_sky_model: the band offsets are the monopole's amplitudes (src/dang_data_mod.f90:357-361)
                orc.offset[:] = eng.get_template_amplitudes(l)[0]
        else:
            orc.amplitude(l)[:] = eng.get_amplitude(l)


@pytest.mark.parametrize("config,nside,which", [("C3", 8, "template"), ("C2", 8, "template"), ("C3", 8, "monopole"), ("C2", 8, "monopole")])
def test_sweeps_beside_a_fitted_template_run_the_plane_set_kernels(built, config, nside, which):
    """SURVEY 8f rank 1's model shape: diffuse components and a Q/U template whose per-band amplitudes are fitted in the Q+U group
    (or a monopole fitted in the T group: the same signal form, and its amplitudes are the band offsets the T launch calibrates by).
    The template has no signal on T (its T plane is zero), so the T group keeps its plane-set launches; on Q+U the coupled solve is
    the Schur one and the sweeps that follow are ONE launch that takes the template's signal out of the data like any other
    component outside the sweep (src/dang_sample_mod.f90:331-352).  Sweeps against the oracle on the same state: indices 1e-12,
    accepted counts equal, chi^2 1e-10; the profile shows one sweep launch per plane set and no stand-alone chi^2 pass."""
    case = _template_case(config, nside, which)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    nmaps = meta["nmaps"]
    tl = len(comps) - 1
    coupled = comps[tl].cg_group
    for it in (2, 3):
        for g in dpar.cg_groups:
            f = g.pol_flag[0]
            s = da.stream_id(it, 0, g.cg_group, 0, f)
            _, bad = eng.amp_sample(g.cg_group, f, "sample", dpar.seed, s)
            assert bad == 0
            if g.cg_group == coupled:
                _copy_group_state(eng, orc, comps, coupled)
            else:
                orc.amp_sample_direct(g.cg_group, f, "sample", dpar.seed, s, "reference")
        assert np.abs(eng.get_template_amplitudes(tl)).max() > 0.0
        eng.profile(True)
        counts = {}
        for g in dpar.cg_groups:
            f = g.pol_flag[0]
            sw = _plane_sweep_list(comps, g.cg_group, f, it)
            for (l, j, st), a in zip(sw, eng.plane_sweeps_sample(f, sw, dpar.nsample, "sample", dpar.seed)):
                counts[(l, j, f)] = a
        after = eng.chisq_cached(1, 1, nmaps)
        prof = eng.profile_get()
        eng.profile(False)
        assert prof["k_index_mh"]["launches"] == len(dpar.cg_groups) and "k_sky_chisq" not in prof, prof
        for (l, j, f), a in counts.items():
            assert a == orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, "sample", dpar.seed, da.stream_id(it, 1, l, j, f)), (it, l, j)
        for l in range(len(comps)):
            if comps[l].nindices:
                assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-12, (it, l)
        ochi, _ = orc.chisq(1, nmaps, 1.0)
        if which == "monopole":
            # the chain removes the monopole as a component ON TOP of the band offset it has become (src/dang_sample_mod.f90:180-196 after
            # src/dang_data_mod.f90:357-361), update_sky_model leaves it out: the sweeps' sums of squares are not chi^2 there
            assert after is None
            assert abs(eng.chisq_current(1, nmaps) / meta["nbands"] - ochi) <= 1e-10 * ochi
        else:
            assert after is not None and abs(after / meta["nbands"] - ochi) <= 1e-10 * ochi
            assert abs(eng.sky_model_chisq(1, nmaps) - after) <= 1e-10 * after


def _close_states(a, b, comps, what):
    for l in range(len(comps)):
        x, y = a.get_amplitude(l), b.get_amplitude(l)
        assert np.abs(x - y).max() <= 1e-10 * max(np.abs(y).max(), 1e-30), (what, l)
        if comps[l].nindices:
            assert np.abs(a.get_indices(l) - b.get_indices(l)).max() <= 1e-12, (what, l)
        if comps[l].type == "template":
            x, y = a.get_template_amplitudes(l), b.get_template_amplitudes(l)
            assert np.abs(x - y).max() <= 1e-12 * max(np.abs(y).max(), 1e-30), (what, l)


@pytest.mark.parametrize("config,nside", [("C3", 8), ("C2", 8)])
def test_one_call_form_beside_a_fitted_template(built, config, nside, monkeypatch):
    """dangx_plane_set_sample on the same model.  T group (the template carries nothing there): one launch, solve and sweeps.  Q+U
    group: pass 1 of the Schur solve, the small host solve, then ONE launch that back-substitutes -- the per-pixel solve on the data
    minus the template's new signal (dx_ampreg.h, HT form) -- and runs the sweeps; granted because the system is well conditioned
    (device_schur: every pivot >= 1e-3), for which the residual pass is skipped.  With DANGX_SCHUR_CHECK=1 the residual is measured
    (<= 1e-12 of b, no refinement step: what the skip assumes) and the launches are pass 2, the residual pass and the sweeps.  Both
    forms and the two-call seam give the same state (the template's signal leaves the data before / after the members': rounding
    order, 1e-10 of the amplitudes; indices 1e-12; accepted counts equal)."""
    case = _template_case(config, nside)
    dpar, ddata, bands, comps, meta = case
    mk = lambda: da.Engine(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
    one, two, chk = mk(), mk(), mk()
    for it in (2, 3):
        for which, eng in (("fused", one), ("checked", chk)):
            monkeypatch.setenv("DANGX_SCHUR_CHECK", "1" if which == "checked" else "0")
            eng.profile(True)
            accs = {}
            for g in dpar.cg_groups:
                f = g.pol_flag[0]
                sw = _plane_sweep_list(comps, g.cg_group, f, it)
                bad, accs[g.cg_group] = eng.plane_set_sample(g.cg_group, f, "sample", dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f), sw, dpar.nsample, dpar.seed)
                assert bad == 0
            prof = eng.profile_get()
            eng.profile(False)
            (resid, backward), nref = eng.schur_info()
            if which == "fused":
                acc1 = accs
                # T: one launch; Q+U: one launch after pass 1 -- no stand-alone back-substitution, no stand-alone sweeps
                assert prof["k_amp_index"]["launches"] == 2 and "k_index_mh" not in prof and "k_amp_direct" not in prof, prof
                assert nref == 0 and resid <= 1e-12
            else:
                assert prof["k_amp_index"]["launches"] == 1 and prof["k_index_mh"]["launches"] == 1 and prof["k_amp_direct"]["launches"] == 1, prof
                assert nref == 0 and resid <= 1e-12, (resid, backward, nref)     # measured: what the skip assumes
                assert accs == acc1
        monkeypatch.setenv("DANGX_SCHUR_CHECK", "0")
        for g in dpar.cg_groups:
            f = g.pol_flag[0]
            sw = _plane_sweep_list(comps, g.cg_group, f, it)
            _, bad2 = two.amp_sample(g.cg_group, f, "sample", dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f))
            acc2 = two.plane_sweeps_sample(f, sw, dpar.nsample, "sample", dpar.seed)
            assert bad2 == 0 and acc2 == acc1[g.cg_group], (g.cg_group, acc1, acc2)
        _close_states(one, two, comps, "one call / two calls")
        _close_states(one, chk, comps, "fused / checked")
        lo, hi = ddata.pol_type[0], ddata.pol_type[-1]
        assert abs(one.chisq_cached(1, lo, hi) - two.chisq_cached(1, lo, hi)) <= 1e-10 * two.chisq_cached(1, lo, hi)


@pytest.mark.parametrize("config,nside", [("C3", 8)])
def test_template_group_over_three_contexts_in_one_call(built, config, nside):
    """dangx_sky_plane_set_sample: the Schur rows of the template group are shared over three pixel-shard contexts, every context then
    back-substitutes and sweeps in one launch -- the state of the one-context call on the whole sky (the rows are summed in another
    order: 1e-10 of the amplitudes)."""
    case = _template_case(config, nside)
    dpar, ddata, bands, comps, meta = case
    whole = da.Engine(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
    engs = shard_engines(case, 3)
    bounds = [e.pix0 for e in engs] + [meta["npix"]]
    for it in (2, 3):
        for g in dpar.cg_groups:
            f = g.pol_flag[0]
            sw = _plane_sweep_list(comps, g.cg_group, f, it)
            s = da.stream_id(it, 0, g.cg_group, 0, f)
            bad, acc = whole.plane_set_sample(g.cg_group, f, "sample", dpar.seed, s, sw, dpar.nsample, dpar.seed)
            for e in engs:
                e.profile(True)
            nul, bad3, acc3 = da.sky_plane_set_sample(engs, g.cg_group, f, "sample", dpar.seed, s, sw, dpar.nsample, dpar.seed)
            assert (nul, bad3, acc3) == (0, bad, acc), (g.cg_group, acc, acc3)
            for e in engs:
                prof = e.profile_get()
                e.profile(False)
                assert prof["k_amp_index"]["launches"] == 1 and "k_index_mh" not in prof and "k_amp_direct" not in prof, prof
    for l in range(len(comps)):
        full = whole.get_amplitude(l)
        for q, e in enumerate(engs):
            part = e.get_amplitude(l)
            assert np.abs(part - full[:, bounds[q]:bounds[q + 1]]).max() <= 1e-10 * max(np.abs(full).max(), 1e-30), (l, q)
            if comps[l].nindices:
                assert np.abs(e.get_indices(l) - whole.get_indices(l)[:, :, bounds[q]:bounds[q + 1]]).max() <= 1e-12, (l, q)
        if comps[l].type == "template":
            assert np.abs(engs[0].get_template_amplitudes(l) - whole.get_template_amplitudes(l)).max() <= 1e-12 * np.abs(whole.get_template_amplitudes(l)).max()


def _random_template_case(seed):
    """A random IQU model with one or two Q/U templates fitted in the Q+U group (at most four global rows)."""
    from dang_amd.api import DangComps
    rng = np.random.default_rng(7000 + seed)
    extra = [c for c in rng.permutation(["cmb", "ff"])[: int(rng.integers(0, 3))]]
    names = ["synch", "dust"] + extra
    # (as many bands as diffuse members: the members absorb every global row -- a singular system, the rows keep their values)
    nb = int(rng.choice([b for b in (4, 5, 6, 7, 8, 9, 10, 12) if b >= len(names) + 2]))
    ntmpl = int(rng.integers(1, 3))
    rows = [int(rng.integers(1, 3)) for _ in range(ntmpl)]            # fitted bands per template: 1 or 2 each (R <= 4)
    info = dict(nb=nb, comps=names, rows=rows, nside=int(rng.choice([2, 4])), nsample=int(rng.choice([3, 10])), seed=seed)

    def tweak(dpar, ddata, bands, comps):
        nmaps, npix = ddata.sig_map.shape[1:]
        add = np.zeros_like(ddata.sig_map)
        for t, nfit in enumerate(rows):
            fit = sorted(int(j) for j in rng.choice(nb, size=nfit, replace=False))
            corr = [j in fit for j in range(nb)]
            tmpl = np.zeros((nmaps, npix))
            tmpl[1], tmpl[2] = rng.normal(0, 1, npix), rng.normal(0, 1, npix)
            truth = np.zeros(nb)
            truth[fit] = rng.normal(0.0, 3.0, nfit)
            ta = np.zeros((nmaps, nb))
            ta[1:, :] = rng.normal(0.0, 0.5, nb) * ~np.asarray(corr)     # amplitudes on the UNFITTED bands (removed in every group, :445-460)
            for k in (1, 2):
                add[:, k, :] += (truth + ta[k])[:, None] * tmpl[k][None, :]
            comps.append(DangComps(label="tmpl%d" % t, type="template", nu_ref=100.0, cg_group=2, nindices=0, nfit=nfit, corr=corr,
                                   template=tmpl, template_amplitudes=ta, amplitude=np.zeros((nmaps, npix))))
        ddata.sig_map = ddata.sig_map + add
    case = make_case(None, nside=info["nside"], nbands=nb, comps=names, nmaps=3, tweak=tweak, start="truth", nsample=info["nsample"])
    return case, info


@pytest.mark.parametrize("seed", range(int(os.environ.get("DANGX_TEMPLATE_FUZZ_SEEDS", "16"))))
def test_random_template_models_one_launch_equals_the_separate_passes(built, seed, monkeypatch):
    """16 random models (4-12 bands, odd counts too, 2-4 diffuse members, one or two Q/U templates with 1-2 fitted bands each and amplitudes on
    their unfitted bands): two iterations through dangx_plane_set_sample as they run by default -- pass 1 (k_schur_pass1_qu where
    the shape allows), the host solve, one launch that back-substitutes and sweeps (kernels specialised at run time for most of
    these shapes) -- against DANGX_SCHUR_CHECK=1: pass 2, the measured residual, the sweeps-only launch.  Same state; the measured
    residual of every solve the default path did not check is <= 1e-11 of b."""
    case, info = _random_template_case(seed)
    dpar, ddata, bands, comps, meta = case
    mk = lambda: da.Engine(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
    fused, chk = mk(), mk()
    for it in (2, 3):
        accs = {}
        for which, eng in (("fused", fused), ("checked", chk)):
            monkeypatch.setenv("DANGX_SCHUR_CHECK", "1" if which == "checked" else "0")
            for g in dpar.cg_groups:
                f = g.pol_flag[0]
                sw = _plane_sweep_list(comps, g.cg_group, f, it)
                if not sw:
                    continue
                bad, acc = eng.plane_set_sample(g.cg_group, f, "sample", dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f), sw, dpar.nsample, dpar.seed)
                assert bad == 0, (info, which)
                accs[(which, g.cg_group)] = acc
                if g.cg_group == 2:
                    (resid, _), nref = eng.schur_info()
                    if which == "checked":
                        assert resid <= 1e-11 or nref > 0, (info, resid, nref)
        monkeypatch.setenv("DANGX_SCHUR_CHECK", "0")
        for g in dpar.cg_groups:
            if ("fused", g.cg_group) in accs:
                assert accs[("fused", g.cg_group)] == accs[("checked", g.cg_group)], (info, it, g.cg_group)
        _close_states(fused, chk, comps, info)


@pytest.mark.parametrize("nrows", [2, 5, 6, 8])
def test_pass1_for_q_u_template_groups_with_up_to_eight_rows(built, nrows):
    """k_schur_pass1_qu (dangx_schurqu.hip: one thread per pixel, both planes, symmetric row storage; beyond five rows the row values
    are reduced row by row) against the per-plane passes (DANGX_SCHUR_QU=0: k_schur_pass1_reg up to four rows, the run-time-typed
    k_schur_pass1 beyond): the solution satisfies the reference's system A x = b + sample vector through the oracle's operators,
    and the two sets of kernels give the same amplitudes to rounding -- but not bit for bit, which shows that both ran."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "import dang_amd as da\n"
            "from dang_amd import _lib as L\n"
            "from util import make_case, pair\n"
            "from test_oracle_templates_cpu import add_globals\n"
            "from test_gpu_templates import _packed_x\n"
            "def tweak(dpar, ddata, bands, comps):\n"
            "    nb = ddata.sig_map.shape[0]\n"
            "    add_globals(dpar, ddata, bands, comps, ('template',), 2, fit_bands=list(range(nb - %d, nb)))\n"
            "case = make_case('C3', nside=8, start='truth', tweak=tweak)\n"
            "dpar, ddata, bands, comps, meta = case\n"
            "eng, orc = pair(case)\n"
            "out = {}\n"
            "for ml in ('optimize', 'sample'):\n"
            "    b = orc.compute_rhs(2, L.FLAG_QU)\n"
            "    if ml == 'sample':\n"
            "        b = b + orc.compute_sample_vector(2, L.FLAG_QU, orc.draw_eta(L.FLAG_QU, 8, 9))\n"
            "    it, bad = eng.amp_sample(2, L.FLAG_QU, ml, 8, 9)\n"
            "    x = _packed_x(eng, comps, ('template',), 2, L.FLAG_QU, meta['nbands'])\n"
            "    Ax = orc.compute_Ax(2, L.FLAG_QU, x)\n"
            "    scale = orc.compute_Ax(2, L.FLAG_QU, np.abs(x)) + np.abs(b)\n"
            "    live = scale > 0\n"
            "    (rb, rt), steps = eng.schur_info()\n"
            "    out[ml] = dict(it=it, bad=bad, rb=rb, steps=steps, res=float(np.abs(Ax - b)[live].max() / scale[live].max()),\n"
            "                   amp=[eng.get_amplitude(l).tolist() for l, c in enumerate(comps) if c.cg_group == 2 and c.type != 'template'],\n"
            "                   tamp=eng.get_template_amplitudes(len(comps) - 1).tolist())\n"
            "print('RESULT ' + json.dumps(out))\n") % (root, os.path.join(root, "tests"), nrows)
    res = {}
    for qu in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DANGX_SCHUR_QU=qu, DANGX_SCHUR_CHECK="1"), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        assert r.returncode == 0 and lines, r.stdout[-3000:]
        res[qu] = json.loads(lines[-1][7:])
    same_bits = True
    for ml in ("optimize", "sample"):
        a, b = res["1"][ml], res["0"][ml]
        assert a["it"] == b["it"] and a["bad"] == b["bad"] == 0
        assert a["res"] <= 1e-9 and b["res"] <= 1e-9, (ml, a["res"], b["res"])     # A x = b through the reference's operators
        ta, tb = np.array(a["tamp"]), np.array(b["tamp"])
        assert np.abs(ta - tb).max() <= 1e-8 * max(np.abs(tb).max(), 1e-30), (ml, np.abs(ta - tb).max())
        for x, y in zip(a["amp"], b["amp"]):
            x, y = np.array(x), np.array(y)
            assert np.abs(x - y).max() <= 1e-8 * max(np.abs(y).max(), 1e-30), ml
            same_bits = same_bits and np.array_equal(x, y)
    assert not same_bits


def test_fullsky_chain_from_one_pass_equals_the_pass_per_proposal_form(built):
    """Full-sky index mode (index_mode = 1): the chisq chain and its tuner run on the sufficient statistics of ONE pass over the maps
    (W0, U, V per band and plane about the starting point; dangx_sky.hip) instead of a pass per proposal (DANGX_FULLSKY_STATS=0, the
    reference's form).  Same accepted counts, the same index values to 1e-12, the same step sizes, for a tuned and an untuned start,
    at the map resolution and at a coarser one."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "import dang_amd as da\n"
            "from test_gpu_fullsky import _fullsky_case\n"
            "out = {}\n"
            "for tuned, coarse in ((True, 0), (False, 0), (True, 4)):\n"
            "    case = _fullsky_case('chisq', 'gaussian', tuned=tuned, config='C2', nside=16)\n"
            "    dpar, ddata, bands, comps, meta = case\n"
            "    if not tuned:\n"
            "        for c in comps: c.step_size = [2.0 * g[1] for g in c.gauss_prior]\n"
            "    eng = da.initialize(bands, comps, ddata, npix_global=meta['npix_global'], device=0)\n"
            "    acc, val, step = [], [], []\n"
            "    for it in (2, 3):\n"
            "        for l, c in enumerate(comps):\n"
            "            for j in range(c.nindices):\n"
            "                if not c.sample_index[j]: continue\n"
            "                f = c.pol_flag[j][0]\n"
            "                acc.append(da.sample_index_mh_fullsky(dpar, ddata, l, j, {1: 1, 8: -1}[f], da.stream_id(it, 1, l, j, f),\n"
            "                                                      sample_nside=coarse or None))\n"
            "                val.append(float(eng.get_indices(l)[j, 0 if f == 1 else 1, 0])); step.append(float(c.step_size[j]))\n"
            "    out['%%d_%%d' %% (tuned, coarse)] = dict(acc=acc, val=val, step=step)\n"
            "print('RESULT ' + json.dumps(out))\n") % (root, os.path.join(root, "tests"))
    res = {}
    for stats in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DANGX_FULLSKY_STATS=stats), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        assert r.returncode == 0 and lines, r.stdout[-3000:]
        res[stats] = json.loads(lines[-1][7:])
    for key, a in res["1"].items():
        b = res["0"][key]
        assert a["acc"] == b["acc"], (key, a["acc"], b["acc"])
        assert np.abs(np.array(a["val"]) - np.array(b["val"])).max() <= 1e-12, key
        assert a["step"] == b["step"], key
        assert sum(a["acc"]) > 0
