"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

The oracle restates the reference's algorithm (oracle/dang_oracle.c, file:line cited there).
Tolerances are the stated fp64 tolerances of DESIGN.md and are written next to each check.
"""
import numpy as np
import pytest

import dang_amd as da
from dang_amd import _lib as L

from util import (MAPN, TOL_AMP, TOL_AMP_TIGHT, TOL_CHISQ, TOL_INDEX, TOL_SED, assert_amps_close,
                  assert_indices_close, make_case, pair, relmax)

pytestmark = pytest.mark.gpu


def test_library_reports_gpu(built):
    lib = L.load()
    assert b"gfx950" in lib.dangx_version()


# ------------------------------------------------------------------ SEDs (eval_sed)

@pytest.mark.parametrize("config,nside", [("C2", 4), ("C5", 2)])
def test_eval_sed_matches_oracle(built, config, nside):
    case = make_case(config, nside=nside, start="truth")
    eng, orc = pair(case)
    comps, meta = case[3], case[4]
    worst = 0.0
    for l, c in enumerate(comps):
        for j in range(meta["nbands"]):
            for k in range(1, meta["nmaps"] + 1):
                g = eng.eval_sed(l, j, k)
                o = orc.eval_sed_map(l, j, k)
                worst = max(worst, np.abs(g / o - 1.0).max())
    assert worst <= TOL_SED, worst  # relative, 2e-13


def test_eval_sed_bandpass(built):
    """bp%id /= 'delta': tau0-weighted sums (src/dang_component_mod.f90:909-913 etc.)."""
    def tweak(dpar, ddata, bands, comps):
        rng = np.random.default_rng(3)
        for b in bands[::2]:
            nu = b.nu_c * 1e9 * np.linspace(0.85, 1.15, 17)
            tau = rng.uniform(0.2, 1.0, nu.size)
            b.id, b.nu0, b.tau0 = "LFI", nu, tau / tau.sum()
    case = make_case("C5", nside=2, start="truth", tweak=tweak)
    eng, orc = pair(case)
    comps, meta = case[3], case[4]
    worst = 0.0
    for l in range(len(comps) // 2):
        for j in range(meta["nbands"]):
            g, o = eng.eval_sed(l, j, 1), orc.eval_sed_map(l, j, 1)
            worst = max(worst, np.abs(g / o - 1.0).max())
    assert worst <= TOL_SED, worst


def test_gibbs_iteration_with_bandpass_integrated_bands(built):
    """Amplitude solve, index sweeps and chi^2 when some bands are bandpass-integrated (generic kernels)."""
    def tweak(dpar, ddata, bands, comps):
        rng = np.random.default_rng(4)
        for b in bands[1::2]:
            nu = b.nu_c * 1e9 * np.linspace(0.9, 1.1, 9)
            tau = rng.uniform(0.2, 1.0, nu.size)
            nu[3] = 0.0                      # an empty bandpass row: skipped (src/dang_component_mod.f90:909-913)
            b.id, b.nu0, b.tau0 = "bp", nu, tau / tau.sum()
    case = make_case("C2", nside=4, tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for group, flag in ((1, L.FLAG_T), (2, L.FLAG_QU)):
        eng.amp_sample(group, flag, "sample", 7, 70 + flag)
        orc.amp_sample_direct(group, flag, "sample", 7, 70 + flag, "reference")
    assert_amps_close(eng, orc, len(comps), TOL_AMP)
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if c.sample_index[j]:
                for f in c.pol_flag[j]:
                    ag = eng.index_sample(l, j, MAPN[f], 10, "sample", 7, 900 + 10 * l + j + f)
                    ao = orc.sample_index_mh(l, j, MAPN[f], 10, "sample", 7, 900 + 10 * l + j + f)
                    assert ag == ao, (l, j, f)
    assert_indices_close(eng, orc, comps)
    s = eng.sky_model_chisq(1, 3)
    ochisq, _ = orc.chisq(1, 3, ddata.nump)
    assert abs(s / meta["nbands"] / ddata.nump - ochisq) <= TOL_CHISQ * ochisq


# ------------------------------------------------------------------ amplitude phase

FLAGS = {"T": (1, L.FLAG_T), "QU": (2, L.FLAG_QU)}


@pytest.mark.parametrize("ml_mode,fluct", [("optimize", "reference"), ("sample", "reference"), ("sample", "correct")])
@pytest.mark.parametrize("which", ["T", "QU"])
def test_amp_direct_matches_oracle_direct(built, which, ml_mode, fluct):
    case = make_case("C2", nside=8)
    eng, orc = pair(case)
    group, flag = FLAGS[which]
    _, bad = eng.amp_sample(group, flag, ml_mode, 99, 1234567, fluct_mode=fluct)
    obad = orc.amp_sample_direct(group, flag, ml_mode, 99, 1234567, fluct)
    assert bad == obad == 0
    assert_amps_close(eng, orc, len(case[3]), TOL_AMP_TIGHT, "(direct vs oracle-direct)")  # 1e-11 of max|a|


@pytest.mark.parametrize("ml_mode", ["optimize", "sample"])
@pytest.mark.parametrize("which", ["T", "QU"])
def test_amp_direct_matches_reference_cg(built, which, ml_mode):
    """The MI355X direct solve against the reference algorithm (global CG, src/dang_cg_mod.f90:179-324)
    run to a tight residual: agreement is at CG-convergence level."""
    case = make_case("C1" if which == "T" else "C2", nside=4)
    eng, orc = pair(case)
    group, flag = FLAGS[which]
    eng.amp_sample(group, flag, ml_mode, 5, 77, fluct_mode="reference")
    it = orc.amp_sample_cg(group, flag, ml_mode, 5, 77, i_max=2000, converge=1e-18)
    assert it < 2000
    assert_amps_close(eng, orc, len(case[3]), 1e-7, "(direct vs reference CG)")


def test_amp_single_stokes_flags(built):
    """Flags Q and U alone (map_n = 2, 3; src/dang_cg_mod.f90:357-363)."""
    def tweak(dpar, ddata, bands, comps):
        for c in comps[3:]:
            c.cg_group = 2
    for flag in (L.FLAG_Q, L.FLAG_U):
        case = make_case("C2", nside=4, tweak=tweak)
        eng, orc = pair(case)
        eng.amp_sample(2, flag, "sample", 1, 2)
        orc.amp_sample_direct(2, flag, "sample", 1, 2, "reference")
        assert_amps_close(eng, orc, len(case[3]), TOL_AMP_TIGHT, "(flag %d)" % flag)


def test_amp_gain_offset_others_and_missval_mask(built):
    """gain /= 1 on T (divided, offset NOT removed: src/dang_cg_mod.f90:371), a component that is not
    sampled (removed from the data, :427-443), and missval in the mask."""
    def tweak(dpar, ddata, bands, comps):
        comps[0].sample_amplitude = False           # cmb fixed at its start amplitude
        comps[0].amplitude[:] = 12.5
        ddata.masks[0, 5:9] = da.api.MISSVAL
        ddata.masks[0, 20] = 0.0
    case = make_case("C2", nside=4, gain=[1.0, 1.02, 0.97, 1.0, 1.1], offset=[0.0, 3.0, -2.0, 0.0, 1.0], tweak=tweak)
    eng, orc = pair(case)
    before = eng.get_amplitude(1).copy()
    eng.amp_sample(1, L.FLAG_T, "sample", 3, 4)
    orc.amp_sample_direct(1, L.FLAG_T, "sample", 3, 4, "reference")
    assert_amps_close(eng, orc, len(case[3]), TOL_AMP_TIGHT)
    after = eng.get_amplitude(1)
    assert np.array_equal(after[0, 5:9], before[0, 5:9]) and after[0, 20] == before[0, 20]  # masked: untouched
    assert np.array_equal(eng.get_amplitude(0), orc.amplitude(0))  # not sampled: untouched


@pytest.mark.parametrize("which", ["T", "QU"])
def test_secondary_seams(built, which):
    """compute_rhs / compute_Ax / compute_sample_vector on the reference's packed vectors."""
    case = make_case("C2", nside=4, start="truth")
    eng, orc = pair(case)
    group, flag = FLAGS[which]
    n = eng.group_size(group, flag)
    assert n == orc.group_size(group, flag)
    b_g, b_o = eng.compute_rhs(group, flag), orc.compute_rhs(group, flag)
    assert relmax(b_g, b_o) <= 1e-13
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n)
    assert relmax(eng.compute_Ax(group, flag, x), orc.compute_Ax(group, flag, x)) <= 1e-13
    eta = orc.draw_eta(flag, 11, 22)
    sv_g, sv_o = eng.compute_sample_vector(group, flag, eta), orc.compute_sample_vector(group, flag, eta)
    assert relmax(sv_g, sv_o) <= 1e-13
    m = eta.size
    assert np.all(sv_g[m:] == 0.0)  # quirk 2: only the first component's slots receive the term


@pytest.mark.parametrize("ml_mode", ["optimize", "sample"])
def test_device_cg_matches_reference_cg(built, ml_mode):
    """cg_search on the device (parity mode): same iteration count and amplitudes as the oracle's CG."""
    case = make_case("C2", nside=4)
    eng, orc = pair(case)
    it_g, _ = eng.amp_sample(2, L.FLAG_QU, ml_mode, 8, 9, solver="cg", i_max=100, converge=1e-8)
    it_o = orc.amp_sample_cg(2, L.FLAG_QU, ml_mode, 8, 9, i_max=100, converge=1e-8)
    assert it_g == it_o
    # CG amplifies rounding differences (tree vs sequential dot products) over ~100 iterations of an
    # ill-conditioned system; both stop at the same iteration, amplitudes agree to 1e-7 of max|a| (measured 1.5e-8)
    assert_amps_close(eng, orc, len(case[3]), 1e-7, "(device CG vs oracle CG)")
    # the residual the library reports for that state is what cg_search's delta_new measures
    rel, relg = eng.amp_residual(2, L.FLAG_QU, ml_mode, 8, 9)
    b = orc.compute_rhs(2, L.FLAG_QU)
    if ml_mode == "sample":
        b = b + orc.compute_sample_vector(2, L.FLAG_QU, orc.draw_eta(L.FLAG_QU, 8, 9))
    x = np.concatenate([eng.get_amplitude(l)[1:3].ravel() for l, c in enumerate(case[3]) if c.cg_group == 2 and c.sample_amplitude])
    r = b - orc.compute_Ax(2, L.FLAG_QU, x)
    ok = np.tile(np.tile(case[1].masks[0] != 0, 2), x.size // (2 * case[4]["npix"]))
    want = np.sqrt((r[ok] ** 2).sum() / (b[ok] ** 2).sum())
    assert relg == 0.0 and abs(rel - want) <= 0.05 * want + 1e-12, (rel, want)


# ------------------------------------------------------------------ index phase

def _sweep(eng, orc, comps, dpar, it=2, ml_mode="sample"):
    accs = []
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if not c.sample_index[j]:
                continue
            for f in c.pol_flag[j]:
                s = da.stream_id(it, 1, l, j, f)
                ag = eng.index_sample(l, j, MAPN[f], dpar.nsample, ml_mode, dpar.seed, s)
                ao = orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, ml_mode, dpar.seed, s)
                accs.append((l, j, f, ag, ao))
    return accs


@pytest.mark.parametrize("ml_mode", ["sample", "optimize"])
@pytest.mark.parametrize("config,nside", [("C2", 8), ("C5", 4)])
def test_index_mh_matches_oracle(built, config, nside, ml_mode):
    case = make_case(config, nside=nside, start="truth")
    eng, orc = pair(case)
    accs = _sweep(eng, orc, case[3], case[0], ml_mode=ml_mode)
    assert len(accs) > 0
    for l, j, f, ag, ao in accs:
        assert ag == ao, "accepted-proposal count differs for comp %d index %d flag %d: %d vs %d" % (l, j, f, ag, ao)
    assert_indices_close(eng, orc, case[3], TOL_INDEX)  # 1e-12 absolute


@pytest.mark.parametrize("lnl,prior", [("marginal", "gaussian"), ("chisq", "uniform"), ("chisq", "jeffreys"),
                                       ("prior", "gaussian"), ("marginal", "uniform")])
def test_index_lnl_and_prior_variants(built, lnl, prior):
    def tweak(dpar, ddata, bands, comps):
        for c in comps:
            c.lnl_type = [lnl] * c.nindices
            c.prior_type = [prior] * c.nindices
            c.uni_prior = [[g[0] - 1.5 * g[1], g[0] + 1.5 * g[1]] for g in c.gauss_prior]  # tight: exercises :415
    case = make_case("C2", nside=8, start="truth", tweak=tweak)
    eng, orc = pair(case)
    accs = _sweep(eng, orc, case[3], case[0])
    for l, j, f, ag, ao in accs:
        assert ag == ao
    assert_indices_close(eng, orc, case[3], TOL_INDEX)


def test_index_masked_pixels_get_zero(built):
    """index_map is zero-initialised and masked pixels are skipped (src/dang_sample_mod.f90:223, 362, 483)."""
    case = make_case("C1", nside=8, start="truth")
    eng, orc = pair(case)
    _sweep(eng, orc, case[3], case[0])
    masked = case[1].masks[0] == 0.0
    assert masked.any()
    beta = eng.get_indices(0)[0, 0]
    assert np.all(beta[masked] == 0.0) and np.all(beta[~masked] != 0.0)


# ------------------------------------------------------------------ sky model + chi^2

@pytest.mark.parametrize("config,nside", [("C1", 8), ("C2", 4)])
def test_sky_model_and_chisq(built, config, nside):
    case = make_case(config, nside=nside, start="truth", gain=None)
    eng, orc = pair(case)
    ddata, meta = case[1], case[4]
    s, sky, res, chi = eng.sky_model_chisq(1, meta["nmaps"], want_maps=True)
    osky, ores = orc.sky_model()
    ochisq, ochi = orc.chisq(1, meta["nmaps"], ddata.nump, osky)
    assert relmax(sky, osky) <= 1e-13 and relmax(res, ores) <= 1e-11
    assert relmax(chi, ochi) <= 1e-11
    chisq = s / meta["nbands"] / ddata.nump
    assert abs(chisq - ochisq) <= TOL_CHISQ * ochisq
    assert abs(da.compute_chisq(ddata) - ochisq) <= TOL_CHISQ * ochisq


# ------------------------------------------------------------------ whole Gibbs iterations

@pytest.mark.parametrize("config,nside", [("C1", 16), ("C2", 8)])
def test_gibbs_iterations_match_oracle(built, config, nside):
    """Three Gibbs iterations through the reference's two entry points vs the oracle (same seeds)."""
    case = make_case(config, nside=nside)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for it in range(1, 4):
        da.sample_cg_groups(dpar, ddata, it=it)
        for g in dpar.cg_groups:
            for f in g.pol_flag:
                orc.amp_sample_direct(g.cg_group, f, dpar.ml_mode, dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f), dpar.fluct_mode)
        ochisq, _ = orc.chisq(1, meta["nmaps"], ddata.nump)
        assert abs(ddata.chisq - ochisq) <= 1e-8 * ochisq
        if it > 1:  # src/dang.f90:102
            da.sample_spectral_parameters(dpar, ddata, it=it)
            for l, c in enumerate(comps):
                for j in range(c.nindices):
                    if c.sample_index[j]:
                        for f in c.pol_flag[j]:
                            orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, dpar.ml_mode, dpar.seed, da.stream_id(it, 1, l, j, f))
            ochisq, _ = orc.chisq(1, meta["nmaps"], ddata.nump)
            assert abs(ddata.chisq - ochisq) <= 1e-8 * ochisq
    assert_amps_close(eng, orc, len(comps), TOL_AMP)
    assert_indices_close(eng, orc, comps, 1e-10)


def test_sharding_is_bitwise_invariant(built):
    """Two half-sky shards (pix0 keyed RNG) reproduce the single-context maps bit for bit."""
    from dang_amd import synth
    full = synth.make_sky("C2", nside=8)
    eng_full = da.Engine(full[2], full[3], full[1], npix_global=full[4]["npix_global"], device=0)
    halves = [synth.make_sky("C2", nside=8, rank=r, nranks=2) for r in range(2)]
    engs = [da.Engine(h[2], h[3], h[1], npix_global=h[4]["npix_global"], pix0=h[4]["pix0"], device=0) for h in halves]
    for e in [eng_full] + engs:
        e.amp_sample(1, L.FLAG_T, "sample", 42, 1)
        e.amp_sample(2, L.FLAG_QU, "sample", 42, 2)
        e.index_sample(2, 1, 1, 10, "sample", 42, 3)
        e.index_sample(4, 0, -1, 10, "sample", 42, 4)
    for l in range(len(full[3])):
        cat = np.concatenate([e.get_amplitude(l) for e in engs], axis=-1)
        assert np.array_equal(cat, eng_full.get_amplitude(l))
        if full[3][l].nindices:
            cat = np.concatenate([e.get_indices(l) for e in engs], axis=-1)
            assert np.array_equal(cat, eng_full.get_indices(l))
    s_full = eng_full.sky_model_chisq(1, 3)
    s_parts = sum(e.sky_model_chisq(1, 3) for e in engs)
    assert abs(s_full - s_parts) <= 1e-12 * s_full


def test_fused_chisq_matches_explicit_pass(built):
    """The chi^2 values the index sweeps produce as a by-product (before = state left by the amplitude
    phase, after = current state) equal update_sky_model + compute_chisq evaluated explicitly."""
    case = make_case("C2", nside=8)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    assert eng.chisq_cached(1, 1, 3) is None           # nothing swept yet
    eng.amp_sample(1, L.FLAG_T, "sample", 7, 1)
    eng.amp_sample(2, L.FLAG_QU, "sample", 7, 2)
    explicit_before = eng.sky_model_chisq(1, 3)
    # the amplitude update replaced the cached sums by its own by-product: chi^2 of the state the solves left (round 4: the solve
    # runs in the plane-set kernel, whose residual IS that chi^2), here against the explicit pass
    solved = eng.chisq_cached(0, 1, 3)
    assert solved is not None and solved == eng.chisq_cached(1, 1, 3) and abs(solved - explicit_before) <= 1e-11 * explicit_before
    _sweep(eng, orc, comps, dpar)
    explicit_after = eng.sky_model_chisq(1, 3)
    before, after = eng.chisq_cached(0, 1, 3), eng.chisq_cached(1, 1, 3)
    assert abs(before - explicit_before) <= 1e-11 * explicit_before
    assert abs(after - explicit_after) <= 1e-11 * explicit_after
    assert abs(eng.chisq_cached(1, 1, 1) + eng.chisq_cached(1, 2, 3) - after) <= 1e-12 * after
    # the deferred host flow reports the same numbers as the eager one
    da.sample_cg_groups(dpar, ddata, it=5, defer_chisq=True)
    eager = eng.sky_model_chisq(1, 3) / meta["nbands"] / ddata.nump
    da.sample_spectral_parameters(dpar, ddata, it=5)
    assert abs(ddata.chisq_after_amp - eager) <= 1e-11 * eager
    assert abs(ddata.chisq - eng.sky_model_chisq(1, 3) / meta["nbands"] / ddata.nump) <= 1e-11 * ddata.chisq


def test_T_cmb_component_as_fixed_sky_signal(built):
    """'T_cmb' (evaluate_T_cmb, B_nu/compute_bnu_prime_RJ): eval_sed parity, and its bare-sed signal is removed
    from the data in the amplitude phase / the index sweeps and summed into the sky model (generic paths)."""
    from dang_amd.api import DangComps
    def tweak(dpar, ddata, bands, comps):
        npix = ddata.sig_map.shape[-1]
        comps.append(DangComps(label="tcmb", type="T_cmb", nu_ref=100.0, cg_group=9, sample_amplitude=False, nindices=1,
                               ind_label=["T"], sample_index=[False], index_mode=[1], lnl_type=["chisq"],
                               prior_type=["uniform"], gauss_prior=[[2.7255, 1.0]], uni_prior=[[0.0, 10.0]],
                               step_size=[0.0], pol_flag=[[L.FLAG_T]], amplitude=np.zeros((1, npix)),
                               indices=np.full((1, 1, npix), 0.5)))   # a cold blackbody: finite, band-dependent signal
    case = make_case("C1", nside=8, start="truth", tweak=tweak)
    eng, orc = pair(case)
    comps, meta = case[3], case[4]
    for j in range(meta["nbands"]):
        g, o = eng.eval_sed(2, j, 1), orc.eval_sed_map(2, j, 1)
        assert np.all(np.abs(g - o) <= TOL_SED * np.abs(o))      # (exp overflow at 857 GHz gives exactly 0 in both)
    assert orc.eval_sed_map(2, 0, 1)[0] > 1e4
    eng.amp_sample(1, L.FLAG_T, "sample", 3, 4)
    orc.amp_sample_direct(1, L.FLAG_T, "sample", 3, 4, "reference")
    assert_amps_close(eng, orc, 2, TOL_AMP_TIGHT)
    ag = eng.index_sample(1, 0, 1, 10, "sample", 5, 6)
    ao = orc.sample_index_mh(1, 0, 1, 10, "sample", 5, 6)
    assert ag == ao
    assert_indices_close(eng, orc, comps[:2], TOL_INDEX)
    s, sky, res, chi = eng.sky_model_chisq(1, 1, want_maps=True)
    osky, ores = orc.sky_model()
    assert relmax(sky, osky) <= 1e-13 and relmax(res, ores) <= 1e-11


def test_index_means_for_write_data(built):
    """mask_avg(c%indices(:,map_n,j), masks) (src/dang_util_mod.f90:186-206), the number write_data prints every
    iteration, from a device reduction instead of pulling the map."""
    case = make_case("C2", nside=8, start="truth")
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    m = np.asarray(ddata.masks)[0] != 0
    for map_n in (1, 2, 3):
        got = da.index_means(ddata, map_n)
        for l, c in enumerate(comps):
            for j in range(c.nindices):
                if c.sample_index[j]:
                    x = orc.indices(l)[j, map_n - 1]
                    want = 0.0
                    for v in x[m]:
                        want = want + v      # the reference's sequential sum
                    want /= m.sum()
                    assert abs(got[(c.label, c.ind_label[j])] - want) <= 1e-13 * max(abs(want), 1.0)
    s, n = eng.index_masked_sum(1, 0, 1)
    assert n == int(m.sum())


@pytest.mark.parametrize("config,nside", [("C2", 8), ("C5", 4)])
@pytest.mark.parametrize("unequal", ["none", "some", "all"])
def test_qu_amplitude_launch_with_equal_and_unequal_plane_indices(built, config, nside, unequal):
    """The Q+U amplitude launch with the Q and U planes carrying the same spectral indices (what every Q+U index sweep
    leaves), different ones on a few pixels, and different ones everywhere: the oracle's amplitudes in all three cases
    (kernels that share SED evaluations between the two planes of a pixel must not assume the first case)."""
    def tweak(dpar, ddata, bands, comps):
        rng = np.random.default_rng(11)
        for c in comps:
            if c.nindices and c.cg_group == 2 and unequal != "none":
                npix = c.indices.shape[-1]
                pick = np.ones(npix, bool) if unequal == "all" else (rng.uniform(size=npix) < 0.02)
                c.indices[0, 2, pick] *= 1.0 + 0.03 * rng.standard_normal(pick.sum())   # U plane only
    case = make_case(config, nside=nside, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for ml in ("sample", "optimize"):
        eng.amp_sample(2, L.FLAG_QU, ml, 9, 31)
        orc.amp_sample_direct(2, L.FLAG_QU, ml, 9, 31, "reference")
        assert_amps_close(eng, orc, len(comps), TOL_AMP, what="%s %s" % (unequal, ml))
    # and plane by plane (the single-plane launches): same amplitudes as the oracle's single-plane solves
    for flag in (L.FLAG_Q, L.FLAG_U):
        eng.amp_sample(2, flag, "sample", 9, 40 + flag)
        orc.amp_sample_direct(2, flag, "sample", 9, 40 + flag, "reference")
    assert_amps_close(eng, orc, len(comps), TOL_AMP, what="single planes")
