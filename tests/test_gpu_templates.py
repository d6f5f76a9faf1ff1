"""GPU parity for SURVEY 8f rank 1: CG groups with global-amplitude components (template / monopole / hi_fit):
the mixed CG building blocks, the device CG and the direct (Schur complement) solve against the oracle's restatement of src/dang_cg_mod.f90."""
import numpy as np
import pytest

import dang_amd as da
from dang_amd import _lib as L

import oracle_ffi as O
from test_oracle_templates_cpu import add_globals
from util import make_case, pair, relmax

pytestmark = pytest.mark.gpu

CASES = [(("monopole", "hi_fit"), 1, L.FLAG_T), (("template",), 2, L.FLAG_QU), (("hi_fit",), 1, L.FLAG_T)]


def _case(which, group, nside=4, start="truth", skip_band0=False):
    def tweak(dpar, ddata, bands, comps):
        add_globals(dpar, ddata, bands, comps, which, group, skip_band0=skip_band0)
    return make_case("C2", nside=nside, start=start, tweak=tweak)


@pytest.mark.parametrize("which,group,flag", CASES)
def test_mixed_seams_match_oracle(built, which, group, flag):
    case = _case(which, group)
    eng, orc = pair(case)
    n = eng.group_size(group, flag)
    assert n == orc.group_size(group, flag)
    assert relmax(eng.compute_rhs(group, flag), orc.compute_rhs(group, flag)) <= 1e-12
    rng = np.random.default_rng(3)
    x = rng.standard_normal(n)
    assert relmax(eng.compute_Ax(group, flag, x), orc.compute_Ax(group, flag, x)) <= 1e-12
    eta = orc.draw_eta(flag, 11, 22)
    assert relmax(eng.compute_sample_vector(group, flag, eta), orc.compute_sample_vector(group, flag, eta)) <= 1e-12


@pytest.mark.parametrize("ml_mode", ["optimize", "sample"])
@pytest.mark.parametrize("which,group,flag", CASES)
def test_device_cg_with_global_components_matches_oracle(built, which, group, flag, ml_mode):
    case = _case(which, group, start="prior")
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    it_g, _ = eng.amp_sample(group, flag, ml_mode, 8, 9, solver="cg", i_max=60, converge=1e-10)
    it_o = orc.amp_sample_cg(group, flag, ml_mode, 8, 9, i_max=60, converge=1e-10)
    assert it_g == it_o
    # a run that stops at i_max is a rounding-sensitive trajectory (the system mixes ~1e-6 hi_fit amplitudes with
    # ~1e2 diffuse ones and a near-degenerate monopole): iteration count must agree, amplitudes only loosely
    tol = 1e-6 if it_g < 60 else 5e-2
    for l, c in enumerate(comps):
        if c.cg_group != group:
            continue
        if c.type in which:
            a, b = eng.get_template_amplitudes(l), orc.template_amplitudes(l)
            assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-12)
        else:
            assert relmax(eng.get_amplitude(l), orc.amplitude(l)) <= tol


def _packed_x(eng, comps, which, group, flag, nbands):
    """x in the layout of initialize_x (src/dang_cg_mod.f90:1173-1279): [diffuse members | global rows]."""
    planes = {L.FLAG_T: [0], L.FLAG_Q: [1], L.FLAG_U: [2], L.FLAG_QU: [1, 2]}[flag]
    parts, rows = [], []
    for l, c in enumerate(comps):
        if c.cg_group != group or not c.sample_amplitude:
            continue
        if c.type in ("template", "monopole", "hi_fit"):
            ta = eng.get_template_amplitudes(l)
            rows.extend(ta[planes[0], j] for j in range(nbands) if c.corr[j])
        else:
            parts.append(eng.get_amplitude(l)[planes].ravel())
    return np.concatenate(parts + [np.asarray(rows, dtype=np.float64)])


@pytest.mark.parametrize("ml_mode", ["optimize", "sample"])
@pytest.mark.parametrize("which,group,flag", CASES)
def test_direct_solve_with_global_components_solves_the_reference_system(built, which, group, flag, ml_mode):
    """solver='direct' eliminates the per-pixel blocks and solves the Schur system of the global rows: its answer
    must satisfy A x = b (+ sample vector) of the reference's operators (checked through the seams, which are
    themselves checked against the oracle above), i.e. it is the point the reference's CG converges to."""
    case = _case(which, group, start="truth", skip_band0=True)  # a regular system: see add_globals
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    b = eng.compute_rhs(group, flag)
    assert relmax(b, orc.compute_rhs(group, flag)) <= 1e-12
    if ml_mode == "sample":
        b = b + orc.compute_sample_vector(group, flag, orc.draw_eta(flag, 8, 9))
    _, bad = eng.amp_sample(group, flag, ml_mode, 8, 9, solver="direct")
    assert bad == 0
    x = _packed_x(eng, comps, which, group, flag, meta["nbands"])
    assert x.size == eng.group_size(group, flag)
    Ax = orc.compute_Ax(group, flag, x)
    # residual relative to the size of the terms that cancel in each row
    scale = orc.compute_Ax(group, flag, np.abs(x)) + np.abs(b)
    live = scale > 0
    assert np.abs(Ax - b)[live].max() <= 1e-9 * scale[live].max()
    assert (np.abs(Ax - b)[live] / scale[live]).max() <= 1e-7
    # and the oracle's CG, left to converge, lands on the same point
    it_o = orc.amp_sample_cg(group, flag, ml_mode, 8, 9, i_max=6000, converge=1e-16)
    if it_o < 6000:
        for l, c in enumerate(comps):
            if c.cg_group == group and c.type not in which:
                assert relmax(eng.get_amplitude(l), orc.amplitude(l)) <= 1e-4


def test_direct_solve_of_a_degenerate_group(built):
    """hi_fit fitted at every band next to the CMB (pixel-independent SED) is exactly degenerate: g_j*s_j along the
    CMB's SED is absorbed by the CMB amplitudes.  The normal equations stay consistent; the direct solve reports the
    nullity, leaves the free direction at its current value (like a Krylov iterate would) and still satisfies A x = b."""
    case = _case(("hi_fit",), 1, start="truth")
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    b = orc.compute_rhs(1, L.FLAG_T)
    it, bad = eng.amp_sample(1, L.FLAG_T, "optimize", 8, 9, solver="direct")
    assert it == -1 and bad == 0
    x = _packed_x(eng, comps, ("hi_fit",), 1, L.FLAG_T, meta["nbands"])
    Ax = orc.compute_Ax(1, L.FLAG_T, x)
    scale = orc.compute_Ax(1, L.FLAG_T, np.abs(x)) + np.abs(b)
    live = scale > 0
    assert (np.abs(Ax - b)[live] / scale[live]).max() <= 1e-7
    eng.amp_sample(1, L.FLAG_T, "optimize", 8, 9, solver="cg", i_max=20, converge=1e-10)  # the reference algorithm runs too


def test_direct_solve_of_a_group_of_global_components_only(built):
    """No diffuse member (a template-only group): the Schur system is the whole system."""
    def tweak(dpar, ddata, bands, comps):
        add_globals(dpar, ddata, bands, comps, ("template",), 7)
    case = make_case("C2", nside=4, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    b = orc.compute_rhs(7, L.FLAG_QU)
    eng.amp_sample(7, L.FLAG_QU, "optimize", 8, 9, solver="direct")
    x = _packed_x(eng, comps, ("template",), 7, L.FLAG_QU, meta["nbands"])
    Ax = orc.compute_Ax(7, L.FLAG_QU, x)
    assert np.abs(Ax - b).max() <= 1e-10 * np.abs(b).max()
    it_o = orc.amp_sample_cg(7, L.FLAG_QU, "optimize", 8, 9, i_max=200, converge=1e-20)
    l = len(comps) - 1
    a, o = eng.get_template_amplitudes(l), orc.template_amplitudes(l)
    assert np.abs(a - o).max() <= 1e-8 * np.abs(o).max()


def test_global_components_elsewhere_in_the_path(built):
    """As non-members they are removed from the data (direct amplitude solve of another group, index sweeps) and
    summed into the sky model; a monopole sets the band offsets instead (src/dang_data_mod.f90:357-361)."""
    def tweak(dpar, ddata, bands, comps):
        add_globals(dpar, ddata, bands, comps, ("monopole", "hi_fit"), 5)      # group 5: not sampled here
        add_globals(dpar, ddata, bands, comps, ("template",), 6)
        for c in comps[-3:]:
            c.template_amplitudes = c.truth_ta.copy()
    case = make_case("C2", nside=4, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for group, flag in ((1, L.FLAG_T), (2, L.FLAG_QU)):
        eng.amp_sample(group, flag, "sample", 3, 4)
        orc.amp_sample_direct(group, flag, "sample", 3, 4, "reference")
    for l in range(6):
        assert relmax(eng.get_amplitude(l), orc.amplitude(l)) <= 1e-10
    s, sky, res, chi = eng.sky_model_chisq(1, 3, want_maps=True)
    osky, ores = orc.sky_model()
    ochisq, _ = orc.chisq(1, 3, ddata.nump, osky)
    assert relmax(sky, osky) <= 1e-12 and relmax(res, ores) <= 1e-10
    assert abs(s / meta["nbands"] / ddata.nump - ochisq) <= 1e-10 * ochisq
    for (l, j, mapn) in ((1, 0, 1), (5, 1, -1)):
        ag = eng.index_sample(l, j, mapn, 10, "sample", 5, 60 + l)
        ao = orc.sample_index_mh(l, j, mapn, 10, "sample", 5, 60 + l)
        assert ag == ao
        assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-12


# --------------------------------------------------------------------------- index sampling of hi_fit and T_cmb

def _hi_case(nside=4):
    """The HI fit (src/dang_component_mod.f90:597-640): s = A_nu * HI_p * B_nu(T_p) with a per-pixel temperature."""
    def tweak(dpar, ddata, bands, comps):
        add_globals(dpar, ddata, bands, comps, ("hi_fit",), 1)   # every band carries an amplitude (marginal form: m /= 0)
        hi = comps[-1]
        hi.template_amplitudes = hi.truth_ta.copy()
        hi.sample_index = [True]
        hi.step_size = [0.8]
        rng = np.random.default_rng(5)
        hi.indices = hi.indices + rng.normal(0.0, 1.0, hi.indices.shape)
    return make_case("C2", nside=nside, start="truth", tweak=tweak)


@pytest.mark.parametrize("lnl,ml_mode", [("chisq", "sample"), ("chisq", "optimize"), ("marginal", "sample")])
def test_hi_fit_temperature_sampling_matches_oracle(built, lnl, ml_mode):
    case = _hi_case()
    dpar, ddata, bands, comps, meta = case
    l = len(comps) - 1
    comps[l].lnl_type = [lnl]
    eng, orc = pair(case)
    ag = eng.index_sample(l, 0, 1, 10, ml_mode, 5, 77)
    ao = orc.sample_index_mh(l, 0, 1, 10, ml_mode, 5, 77)
    assert ag == ao and ag > 0
    assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-12
    # the sweep's by-product chi^2 equals a fresh evaluation
    s = eng.sky_model_chisq(1, 1)
    assert abs(eng.chisq_cached(1, 1, 1) - s) <= 1e-10 * s


def test_mask_hi_threshold_mirrors_the_reference():
    from dang_amd.api import mask_hi_threshold
    case = _hi_case()
    dpar, ddata, bands, comps, meta = case
    hi = comps[-1]
    t0, m0 = hi.template.copy(), ddata.masks.copy()
    mask_hi_threshold(ddata, hi, 0.6)
    assert np.array_equal(ddata.masks[0] == 0.0, (t0[0] > 0.6) | (m0[0] == 0.0))
    assert np.allclose(hi.template, t0 / 0.6, rtol=0, atol=0)


def _tcmb_case(index_mode):
    from dang_amd.api import DangComps

    def tweak(dpar, ddata, bands, comps):
        npix = ddata.sig_map.shape[-1]
        comps.append(DangComps(label="tcmb", type="T_cmb", nu_ref=100.0, cg_group=9, sample_amplitude=False, nindices=1,
                               ind_label=["T"], sample_index=[True], index_mode=[index_mode], lnl_type=["chisq"],
                               prior_type=["gaussian"], gauss_prior=[[0.5, 0.05]], uni_prior=[[0.1, 2.0]],
                               step_size=[1e-3], pol_flag=[[L.FLAG_T]], amplitude=np.zeros((3, npix)),
                               indices=np.full((1, 3, npix), 0.5)))
    return make_case("C2", nside=4, start="truth", tweak=tweak)


def test_T_cmb_sampling_per_pixel_and_full_sky(built):
    """'T_cmb' has no amplitude (eval_signal = the bare sed, src/dang_component_mod.f90:770-771); after its
    indices are sampled the driver copies c%indices(0,1,1) into the global T_CMB (src/dang_sample_mod.f90:75-78)."""
    case = _tcmb_case(2)
    dpar, ddata, bands, comps, meta = case
    l = len(comps) - 1
    eng, orc = pair(case)
    ag = eng.index_sample(l, 0, 1, 10, "sample", 5, 78)
    ao = orc.sample_index_mh(l, 0, 1, 10, "sample", 5, 78)
    assert ag == ao
    assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-13
    case = _tcmb_case(1)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    s = da.stream_id(2, 1, l, 0, 1)
    ag = da.sample_index_mh_fullsky(dpar, ddata, l, 0, 1, s)
    ao, _, _ = orc.sample_index_fullsky(l, 0, 1, dpar.nsample, dpar.ml_mode, dpar.seed, s)
    assert ag == ao
    assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-13
    # through the orchestrator: the new temperature becomes the global T_CMB, i.e. the cmb SED 1/a2t moves with it
    cmb_before = eng.eval_sed(0, 1, 1)[0]
    for c in comps[:-1]:
        c.sample_index = [False] * c.nindices
    da.sample_spectral_parameters(dpar, ddata, it=3)
    T = eng.get_indices(l)[0, 0, 0]
    cmb_after = eng.eval_sed(0, 1, 1)[0]
    assert cmb_after != cmb_before
    want = O.Oracle(bands, comps, ddata, tcmb=T).eval_sed_map(0, 1, 1)[0]
    assert abs(cmb_after - want) <= 1e-12 * abs(want)
    # ddata%chisq is evaluated AFTER the T_CMB update (update_sky_model + compute_chisq, src/dang_sample_mod.f90:75-84):
    # the chi^2 the sweep cached belongs to the old T_CMB and must not be reported
    eng.pull_state()
    ref, _ = O.Oracle(bands, comps, ddata, tcmb=T).chisq(ddata.pol_type[0], ddata.pol_type[-1], ddata.nump)
    assert abs(ddata.chisq - ref) <= 1e-10 * abs(ref), (ddata.chisq, ref)
    # (dangx_set_tcmb invalidated the sweep's sum; compute_chisq then evaluated the plane again and left ITS value in the cache)
    cached = sum(eng.chisq_cached(1, k, k) for k in range(ddata.pol_type[0], ddata.pol_type[-1] + 1))
    assert abs(cached / meta["nbands"] / ddata.nump - ref) <= 1e-10 * abs(ref)


def test_gibbs_iterations_with_a_fitted_template(built):
    """sample_cg_groups (Schur direct solve for the group with the template) + sample_spectral_parameters in a loop:
    the chain stays finite and the per-band template amplitudes settle on the injected ones."""
    def tweak(dpar, ddata, bands, comps):
        # fitted at two of the five bands: with three diffuse components per pixel that is a well-posed fit (a template
        # with a free amplitude at four of five bands is constrained only through the spatial variation of the SEDs,
        # and its maximum-likelihood amplitudes are noise amplified a thousandfold -- correctly, but uselessly)
        add_globals(dpar, ddata, bands, comps, ("template",), 2, fit_bands=[3, 4])
    case = make_case("C2", nside=16, tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
    l = len(comps) - 1
    truth = comps[l].truth_ta
    for it in range(1, 41):
        info = da.sample_cg_groups(dpar, ddata, it=it)
        assert all(bad == 0 for (_, _, _, bad) in info)
        if it > 1:
            da.sample_spectral_parameters(dpar, ddata, it=it)
        assert np.isfinite(ddata.chisq)
    ta = eng.get_template_amplitudes(l)
    assert np.array_equal(ta[1], ta[2])                      # one amplitude for Q and U (:1380-1382)
    assert np.all(ta[1, :3] == 0.0)                          # bands 0-2 are not fitted
    # 3072 pixels x 2 planes of unit-variance template against ~1 uK noise: sigma(amplitude) ~ 0.01-0.05
    assert np.abs(ta[1, 3:] - truth[1, 3:]).max() < 0.3, (ta[1], truth[1])


def test_templates_are_removed_on_their_unfitted_bands_in_every_group(built):
    """compute_rhs subtracts every template / monopole on the bands it is NOT fitted at -- "Still subtract templates
    which exist but may not be fit here", src/dang_cg_mod.f90:445-460 -- in every CG group; a template that is not a
    member of the group (already removed as an "other" component, :427-443) is thereby removed a second time there.
    Visible only when such a band carries an amplitude (e.g. read from file)."""
    def tweak(dpar, ddata, bands, comps):
        add_globals(dpar, ddata, bands, comps, ("template",), 6)     # not a member of group 2
        comps[-1].template_amplitudes = comps[-1].truth_ta.copy()
        comps[-1].template_amplitudes[1:, 0] = 7.5                    # an amplitude on the UNFITTED band 0
    case = make_case("C2", nside=4, start="truth", tweak=tweak)
    eng, orc = pair(case)
    assert relmax(eng.compute_rhs(2, L.FLAG_QU), orc.compute_rhs(2, L.FLAG_QU)) <= 1e-12
    eng.amp_sample(2, L.FLAG_QU, "optimize", 3, 4)
    orc.amp_sample_direct(2, L.FLAG_QU, "optimize", 3, 4, "reference")
    eng2, orc2 = pair(case)
    i1, _ = eng2.amp_sample(2, L.FLAG_QU, "optimize", 3, 4, solver="cg", i_max=400, converge=1e-12)
    i2 = orc2.amp_sample_cg(2, L.FLAG_QU, "optimize", 3, 4, i_max=400, converge=1e-12)
    assert i1 < 400 and i2 < 400
    for l in (3, 4, 5):
        assert relmax(eng.get_amplitude(l), orc.amplitude(l)) <= 1e-11       # direct: GPU == oracle
        assert relmax(orc.amplitude(l), orc2.amplitude(l)) <= 1e-6           # direct == the reference's CG fixed point
        assert relmax(eng2.get_amplitude(l), orc2.amplitude(l)) <= 1e-6      # device CG == oracle CG


def test_T_cmb_as_an_amplitude_sampled_group_member(built):
    """The reference's CG code tests only for template / hi_fit / monopole (src/dang_cg_mod.f90:469, 691, 807), so a
    'T_cmb' component with sample_amplitude in a CG group is a DIFFUSE member whose mixing element is evaluate_T_cmb
    (B_nu(T)/B'_RJ*1e6) -- although eval_signal ignores its amplitude (:770-771).  Reproduced as it is: operators' seams,
    the direct block solve and the device CG against the oracle."""
    from dang_amd.api import DangComps

    def tweak(dpar, ddata, bands, comps):
        npix = ddata.sig_map.shape[-1]
        rng = np.random.default_rng(17)
        comps.insert(3, DangComps(label="tcmb", type="T_cmb", nu_ref=100.0, cg_group=1, sample_amplitude=True, nindices=1,
                                  ind_label=["T"], sample_index=[False], index_mode=[2], lnl_type=["chisq"],
                                  prior_type=["gaussian"], gauss_prior=[[2.7, 0.1]], uni_prior=[[1.0, 5.0]], step_size=[1e-3],
                                  pol_flag=[[L.FLAG_T]], amplitude=np.zeros((3, npix)),
                                  indices=2.7 + 0.05 * rng.normal(size=(1, 3, npix))))
    case = make_case("C2", nside=4, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    assert eng.group_size(1, L.FLAG_T) == 4 * meta["npix"]
    rng = np.random.default_rng(3)
    x = rng.normal(size=4 * meta["npix"])
    assert relmax(eng.compute_rhs(1, L.FLAG_T), orc.compute_rhs(1, L.FLAG_T)) <= 1e-12
    assert relmax(eng.compute_Ax(1, L.FLAG_T, x), orc.compute_Ax(1, L.FLAG_T, x)) <= 1e-12
    eta = orc.draw_eta(L.FLAG_T, 8, 9)
    assert relmax(eng.compute_sample_vector(1, L.FLAG_T, eta), orc.compute_sample_vector(1, L.FLAG_T, eta)) <= 1e-12
    _, bad = eng.amp_sample(1, L.FLAG_T, "sample", 8, 9)
    assert bad == orc.amp_sample_direct(1, L.FLAG_T, "sample", 8, 9, "reference") == 0
    for l in range(4):
        b = orc.amplitude(l)
        assert np.abs(eng.get_amplitude(l) - b).max() <= 1e-8 * np.abs(b).max(), l
    # the solved system is the reference's: residual through its own operators
    rel, _ = eng.amp_residual(1, L.FLAG_T, "sample", 8, 9)
    assert rel <= 1e-9
    eng2, orc2 = pair(case)
    it_g, _ = eng2.amp_sample(1, L.FLAG_T, "optimize", 8, 9, solver="cg", i_max=60, converge=1e-8)
    it_o = orc2.amp_sample_cg(1, L.FLAG_T, "optimize", 8, 9, i_max=60, converge=1e-8)
    assert abs(it_g - it_o) <= 1


@pytest.mark.parametrize("which,group,flag", [(("template",), 2, L.FLAG_QU), (("monopole",), 1, L.FLAG_T), (("hi_fit",), 1, L.FLAG_T),
                                              (("monopole", "hi_fit"), 1, L.FLAG_T)])
def test_schur_passes_on_the_amplitude_schedule_are_the_run_time_typed_ones(built, which, group, flag):
    """Groups whose global members are templates / monopoles / hi_fit components run the passes of the Schur solve on the amplitude
    kernel's schedule (dangx_ampreg.hip: k_schur_pass1_reg, k_amp_reg<.., true>, k_schur_resid_reg); DANGX_SCHUR_FAST=0 keeps the
    run-time-typed passes of dangx_schur.hip.  Same amplitudes (the reciprocals and square roots differ in their last bits --
    which also shows that two different sets of kernels ran) and the same refinement report."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "import test_gpu_templates as t\n"
            "from util import pair\n"
            "case = t._case(%r, %d, nside=8, skip_band0=True)\n"
            "dpar, ddata, bands, comps, meta = case\n"
            "eng, orc = pair(case)\n"
            "out = {}\n"
            "for ml in ('optimize', 'sample'):\n"
            "    it, bad = eng.amp_sample(%d, %d, ml, 8, 9)\n"
            "    (rb, rt), steps = eng.schur_info()\n"
            "    out[ml] = dict(it=it, bad=bad, rb=rb, steps=steps,\n"
            "                   amp=[eng.get_amplitude(l).tolist() for l, c in enumerate(comps) if c.cg_group == %d and c.type not in %r],\n"
            "                   tamp=[eng.get_template_amplitudes(l).tolist() for l, c in enumerate(comps) if c.type in %r])\n"
            "print('RESULT ' + json.dumps(out))\n") % (root, os.path.join(root, "tests"), which, group, group, flag, group, which, which)
    res = {}
    for fast in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DANGX_SCHUR_FAST=fast), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        assert r.returncode == 0 and lines, r.stdout[-3000:]
        res[fast] = json.loads(lines[-1][7:])
    same_bits = True
    # a monopole beside hi_fit is a near-degenerate pair (condition ~1e8, cf. test_direct_solve_with_global_components...):
    # its residual floor and the agreement of two roundings of the same solve are correspondingly looser
    # (the hi_fit passes -- compile units of their own with the Planck factor -- are held to 1e-9 on the hi_fit-only group, which is
    # well conditioned; only the monopole + hi_fit PAIR gets the loose bound)
    tol = 1e-6 if ("hi_fit" in which and "monopole" in which) else 1e-9
    for ml in ("optimize", "sample"):
        a, b = res["1"][ml], res["0"][ml]
        assert a["it"] == b["it"] and a["bad"] == b["bad"] == 0
        assert a["rb"] <= tol and b["rb"] <= tol, (a["rb"], b["rb"])
        for x, y in zip(a["amp"] + a["tamp"], b["amp"] + b["tamp"]):
            x, y = np.asarray(x), np.asarray(y)
            assert np.abs(x - y).max() <= tol * max(np.abs(y).max(), 1.0)
            same_bits = same_bits and np.array_equal(x, y)
    assert not same_bits
