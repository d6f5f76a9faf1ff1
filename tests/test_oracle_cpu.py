"""CPU tests of the oracle (oracle/dang_oracle.c): known-answer vectors, closed forms and the
algebraic identities of the path.  The reference ships no tests or golden vectors and cannot be
built here, so these are what pins the oracle ("parity unpinned" w.r.t. the reference itself).
"""
import math

import numpy as np
import pytest

from dang_amd import _lib as L

import oracle_ffi as O
from util import make_case

H = 1.0545726691251021e-34 * 2.0 * 3.141592653589793238462643383279502884197
K_B = 1.3806503e-23


def test_philox4x32_10_known_answers():
    """Random123 kat_vectors for philox4x32-10 (Salmon et al., SC'11)."""
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_stream_properties():
    u = np.array([O.uniform2(1234, 77, p, d) for p in range(2000) for d in (1, 2)])
    assert (u > 0).all() and (u < 1).all()
    assert abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
    assert O.uniform2(1, 2, 3, 4) == O.uniform2(1, 2, 3, 4)
    assert O.uniform2(1, 2, 3, 4) != O.uniform2(1, 2, 3, 5)
    assert O.uniform2(1, 2, 3, 4) != O.uniform2(1, 3, 3, 4)


def test_rand_normal_and_normal_prior_closed_forms():
    lib = O.lib()
    for u1, u2 in [(0.3, 0.2), (0.9, 0.77), (1e-9, 0.25)]:
        ref = 1.5 + 0.7 * math.sqrt(-2.0 * math.log(u1)) * math.sin(2.0 * math.pi * u2)  # src/dang_util_mod.f90:106-108
        assert abs(lib.dgo_rand_normal(1.5, 0.7, u1, u2) - ref) <= 1e-15 * max(1.0, abs(ref))
    for x, m, s in [(0.1, 0.0, 1.0), (-3.2, -3.1, 0.1), (25.0, 19.6, 1.5)]:
        ref = math.exp(-(x - m) ** 2 / (2 * s * s)) / (s * math.sqrt(2 * math.pi))
        assert abs(lib.dgo_eval_normal_prior(x, m, s) - ref) <= 1e-15 * ref


def _sed_python(ctype, nu, nu_ref, th):
    if ctype == "power-law":
        return (nu / nu_ref) ** th[0]
    if ctype == "mbb":
        z = H / (K_B * th[1])
        return (math.exp(z * nu_ref) - 1.0) / (math.exp(z * nu) - 1.0) * (nu / nu_ref) ** (th[0] + 1.0)
    if ctype == "lognormal":
        return math.exp(-0.5 * (math.log(nu / (th[0] * 1e9)) / th[1]) ** 2) * (nu_ref / nu) ** 2
    if ctype == "freefree":
        g = lambda v: math.log(math.exp(5.960 - math.sqrt(3.0) / math.pi * math.log(v / 1e9 * (th[0] / 1e4) ** (-1.5))) + 2.71828)
        return g(nu) / g(nu_ref) * (nu / nu_ref) ** (-2)
    if ctype == "cmb":
        y = H * nu / (K_B * 2.7255)
        return 1.0 / ((math.exp(y) - 1.0) ** 2 / (y * y * math.exp(y)))
    raise ValueError


def test_sed_closed_forms():
    """eval_sed against an independent Python statement of src/dang_component_mod.f90:886-1040 / a2t."""
    case = make_case("C5", nside=1, start="truth")
    dpar, ddata, bands, comps, meta = case
    orc = O.Oracle(bands, comps, ddata)
    thetas = {"power-law": [-3.1], "mbb": [1.55, 21.0], "freefree": [7000.0], "lognormal": [22.5, 0.45], "cmb": []}
    for l, c in enumerate(comps[:6]):
        for j, b in enumerate(bands):
            th = thetas[c.type] + [0.0, 0.0]
            got = orc.eval_sed(l, j, th[:2])
            ref = _sed_python(c.type, b.nu_c * 1e9, c.nu_ref * 1e9, th)
            assert abs(got / ref - 1.0) <= 5e-14, (c.type, j, got, ref)


def test_sed_at_reference_frequency_is_one():
    from dang_amd.api import BandInfo
    case = make_case("C5", nside=1, start="truth")
    dpar, ddata, bands, comps, meta = case
    for l, c in enumerate(comps[:6]):
        if c.type in ("cmb", "lognormal"):
            continue
        b2 = [BandInfo(label="ref", nu_c=c.nu_ref)] + bands[1:]
        orc = O.Oracle(b2, comps, ddata)
        assert abs(orc.eval_sed_map(l, 0, 1)[0] - 1.0) <= 1e-14


def test_single_sample_bandpass_equals_delta():
    """tau0=1 at nu0=nu_c makes the tau-weighted sum (e.g. :909-913) identical to the delta branch."""
    case = make_case("C5", nside=1, start="truth")
    dpar, ddata, bands, comps, meta = case
    o1 = O.Oracle(bands, comps, ddata)
    for b in bands:
        b.id, b.nu0, b.tau0 = "LFI", np.array([b.nu_c * 1e9]), np.array([1.0])
    o2 = O.Oracle(bands, comps, ddata)
    for l in range(6):
        for j in range(len(bands)):
            assert o1.eval_sed_map(l, j, 1)[0] == o2.eval_sed_map(l, j, 1)[0]


@pytest.mark.parametrize("group,flag", [(1, L.FLAG_T), (2, L.FLAG_QU)])
def test_Ax_is_symmetric_positive_and_consistent_with_rhs(group, flag):
    case = make_case("C2", nside=2, start="truth")
    dpar, ddata, bands, comps, meta = case
    orc = O.Oracle(bands, comps, ddata)
    n = orc.group_size(group, flag)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    Ax, Ay = orc.compute_Ax(group, flag, x), orc.compute_Ax(group, flag, y)
    assert abs(y @ Ax - x @ Ay) <= 1e-12 * abs(y @ Ax)
    assert x @ Ax > 0
    # masked rows/columns are zero (src/dang_cg_mod.f90:695)
    npix = meta["npix"]
    m = np.tile(ddata.masks[0] == 0, n // npix)
    assert np.all(Ax[m] == 0.0)


def test_noise_free_rhs_equals_A_times_truth_and_solvers_recover_truth():
    """With d = sum_c a_c M_c exactly, b = A a_true; CG and the direct block solve both return a_true."""
    case = make_case("C2", nside=2, start="truth")
    dpar, ddata, bands, comps, meta = case
    o = O.Oracle(bands, comps, ddata)
    sky, _ = o.sky_model()
    ddata.sig_map = sky.copy()  # noise-free data (gain 1, offset 0)
    for group, flag in [(1, L.FLAG_T), (2, L.FLAG_QU)]:
        o = O.Oracle(bands, comps, ddata)
        x_true = o.initialize_x(group, flag)
        b = o.compute_rhs(group, flag)
        Ax = o.compute_Ax(group, flag, x_true)
        assert np.abs(b - Ax).max() <= 1e-10 * np.abs(b).max()
        # start from zero amplitudes
        for l in range(len(comps)):
            if comps[l].cg_group == group:
                o.amplitude(l)[:] = np.where(ddata.masks == 0, o.amplitude(l), 0.0)
        o2 = O.Oracle(bands, comps, ddata)
        for l in range(len(comps)):
            o2.amplitude(l)[:] = o.amplitude(l)
        it = o.amp_sample_cg(group, flag, "optimize", 1, 1, i_max=5000, converge=1e-20)
        assert it < 5000
        assert o2.amp_sample_direct(group, flag, "optimize", 1, 1) == 0
        for l, c in enumerate(comps):
            if c.cg_group != group:
                continue
            t = np.asarray(c.amplitude)
            scale = np.abs(t).max()
            assert np.abs(o.amplitude(l) - t).max() <= 1e-6 * scale    # CG: residual-limited
            assert np.abs(o2.amplitude(l) - t).max() <= 1e-8 * scale   # direct: condition-number x eps


def test_sample_vector_reproduces_reference_quirks():
    """src/dang_cg_mod.f90:1033-1040: '=' and no component offset -> only slot 0 gets the term,
    holding the LAST component's SED; :1008-1015: one eta for all bands."""
    case = make_case("C2", nside=2, start="truth")
    dpar, ddata, bands, comps, meta = case
    o = O.Oracle(bands, comps, ddata)
    eta = o.draw_eta(L.FLAG_T, 3, 4)
    res = o.compute_sample_vector(1, L.FLAG_T, eta)
    npix = meta["npix"]
    assert np.all(res[npix:] == 0.0)
    last = 2  # dust is the last T component
    expect = np.zeros(npix)
    for j in range(meta["nbands"]):
        expect += np.where(ddata.masks[0] == 0, 0.0, eta / ddata.rms_map[j, 0] * o.eval_sed_map(last, j, 1))
    assert np.abs(res[:npix] - expect).max() <= 1e-13 * np.abs(expect).max()


def test_direct_sampler_correct_mode_has_posterior_covariance():
    """Textbook fluctuation term: L^t (x - x_ML) ~ N(0, I) where A = L L^t (checked over pixels)."""
    case = make_case("C1", nside=16, start="truth")
    dpar, ddata, bands, comps, meta = case
    o_ml = O.Oracle(bands, comps, ddata)
    o_ml.amp_sample_direct(1, L.FLAG_T, "optimize", 0, 0)
    o_s = O.Oracle(bands, comps, ddata)
    o_s.amp_sample_direct(1, L.FLAG_T, "sample", 11, 12, "correct")
    npix, nb = meta["npix"], meta["nbands"]
    ok = ddata.masks[0] != 0
    M = np.stack([np.stack([o_ml.eval_sed_map(l, j, 1) for l in range(2)], -1) for j in range(nb)], 0)  # [nb][npix][2]
    w = 1.0 / ddata.rms_map[:, 0, :] ** 2
    A = np.einsum("jpa,jp,jpb->pab", M, w, M)
    dx = np.stack([o_s.amplitude(l)[0] - o_ml.amplitude(l)[0] for l in range(2)], -1)
    Lc = np.linalg.cholesky(A[ok])
    z = np.einsum("pba,pb->pa", Lc, dx[ok])
    n = z.shape[0]
    assert abs(z.mean()) < 5 / math.sqrt(2 * n)
    assert abs(z.var() - 1.0) < 5 * math.sqrt(2.0 / (2 * n))
    assert abs(np.mean(z[:, 0] * z[:, 1])) < 5 / math.sqrt(n)


def test_chisq_definition_and_expectation():
    """compute_chisq (src/dang_data_mod.f90:494-526): sum_j res^2/rms^2 / nbands, summed / nump.
    With the true indices and ML amplitudes E[chisq] = (nb - nc)/nb."""
    case = make_case("C1", nside=16, start="truth")
    dpar, ddata, bands, comps, meta = case
    o = O.Oracle(bands, comps, ddata)
    o.amp_sample_direct(1, L.FLAG_T, "optimize", 0, 0)
    sky, res = o.sky_model()
    chisq, chi_map = o.chisq(1, 1, ddata.nump, sky)
    ok = ddata.masks[0] != 0
    manual = ((res[:, 0, :] / ddata.rms_map[:, 0, :]) ** 2).sum(0) / meta["nbands"]
    assert np.abs(chi_map[0][ok] - manual[ok]).max() <= 1e-12 * manual[ok].max()
    assert np.all(chi_map[0][~ok] == 0.0)
    assert abs(chisq - manual[ok].sum() / ddata.nump) <= 1e-12 * chisq
    expect = (meta["nbands"] - 2) / meta["nbands"]
    assert abs(chisq - expect) < 6 * math.sqrt(2.0 * (meta["nbands"] - 2) / ok.sum()) / meta["nbands"]


def test_lnL_closed_forms():
    lib = O.lib()
    rng = np.random.default_rng(5)
    nb = 4
    d, m = rng.standard_normal((2, 64 * 3)), None
    m = rng.standard_normal(64 * 3)
    r = rng.uniform(0.5, 1.5, 64 * 3)
    P = lambda a: np.ascontiguousarray(a).ctypes.data_as(O._D)
    d = d[0]
    for s1, s2 in [(1, 1), (2, 3)]:
        ref = sum(-0.5 * ((d[(k - 1) * 64 + j] - m[(k - 1) * 64 + j]) / r[(k - 1) * 64 + j]) ** 2
                  for k in range(s1, s2 + 1) for j in range(nb))
        got = lib.dgo_evaluate_lnL(nb, s1, s2, P(d), P(r), P(m), 1, 64, 0, 1.0)
        assert abs(got - ref) <= 1e-14 * abs(ref)
        assert lib.dgo_evaluate_lnL(nb, s1, s2, P(d), P(r), P(m), 1, 64, 0, 0.0) == 0.0  # masked
        refm = sum(-0.5 * (m[q] / r[q] ** 2 * d[q]) ** 2 / (m[q] / r[q] ** 2 * m[q])
                   for k in range(s1, s2 + 1) for j in range(nb) for q in [(k - 1) * 64 + j])
        gotm = lib.dgo_evaluate_marginal_lnL(nb, s1, s2, P(d), P(r), P(m), 1, 64, 0)
        assert abs(gotm - refm) <= 1e-13 * abs(refm)


def test_index_mh_semantics():
    case = make_case("C2", nside=4, start="truth")
    dpar, ddata, bands, comps, meta = case
    o = O.Oracle(bands, comps, ddata)
    masked = ddata.masks[0] == 0
    # optimize mode: chi^2 of the touched plane can only go down (accept iff lnL+prior increases)
    before, _ = o.chisq(1, 1, ddata.nump)
    acc = o.sample_index_mh(1, 0, 1, 20, "optimize", 9, 1)
    after, _ = o.chisq(1, 1, ddata.nump)
    beta = o.indices(1)[0, 0]
    assert acc > 0 and np.all(beta[masked] == 0.0) and np.all(beta[~masked] != 0.0)
    lo, hi = comps[1].uni_prior[0]
    assert np.all((beta[~masked] >= lo) & (beta[~masked] <= hi))
    # Q+U joint sampling writes the same value to both planes (:465)
    o.sample_index_mh(5, 1, -1, 10, "sample", 9, 2)
    T = o.indices(5)[1]
    assert np.array_equal(T[1], T[2]) and not np.array_equal(T[1], np.asarray(comps[5].indices)[1, 1])
    assert np.array_equal(T[0], np.asarray(comps[5].indices)[1, 0])  # plane 1 untouched
    # tight uniform prior: out-of-bounds proposals are skipped, chain stays inside (:415)
    comps[1].uni_prior[0] = [-3.12, -3.08]
    comps[1].indices = np.full_like(comps[1].indices, -3.1)
    o = O.Oracle(bands, comps, ddata)
    o.sample_index_mh(1, 0, 1, 50, "sample", 9, 3)
    b = o.indices(1)[0, 0][~masked]
    assert np.all((b >= -3.12) & (b <= -3.08))


def test_index_prior_lnl_type_draws_from_the_gaussian_prior():
    def tweak(dpar, ddata, bands, comps):
        comps[0].lnl_type = ["prior"]
    case = make_case("C1", nside=16, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    o = O.Oracle(bands, comps, ddata)
    o.sample_index_mh(0, 0, 1, 10, "sample", 1, 1)
    b = o.indices(0)[0, 0][ddata.masks[0] != 0]
    mean, std = comps[0].gauss_prior[0]
    assert abs(b.mean() - mean) < 5 * std / math.sqrt(b.size)
    assert abs(b.std() - std) < 5 * std / math.sqrt(2 * b.size)


def test_index_posterior_tracks_truth():
    """A long per-pixel chain on high-S/N pixels recovers beta_s within its prior-limited error."""
    def tweak(dpar, ddata, bands, comps):
        comps[0].amplitude = comps[0].amplitude * 0 + 400.0   # strong synchrotron
    case = make_case("C1", nside=8, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    o = O.Oracle(bands, comps, ddata)
    sky, _ = o.sky_model()
    rng = np.random.default_rng(0)
    ddata.sig_map = sky + ddata.rms_map * rng.standard_normal(sky.shape)
    truth = np.asarray(comps[0].indices)[0, 0].copy()
    comps[0].indices = np.full_like(comps[0].indices, -3.1)
    o = O.Oracle(bands, comps, ddata)
    o.sample_index_mh(0, 0, 1, 400, "sample", 2, 3)
    ok = ddata.masks[0] != 0
    err = o.indices(0)[0, 0][ok] - truth[ok]
    assert np.abs(err).mean() < 0.02 and abs(err.mean()) < 0.005


def test_planck_and_T_cmb_closed_forms():
    """B_nu (src/dang_component_mod.f90:745-752), compute_bnu_prime_RJ (src/dang_bp_mod.f90:160-168) and
    evaluate_T_cmb (:815-848): B_nu(T)/(2 k nu^2/c^2) * 1e6 -> x/(e^x - 1) * T * 1e6, x = h nu / k T."""
    lib = O.lib()
    c_light = 2.99792458e8
    for nu, T in [(30e9, 2.7255), (143e9, 2.7255), (857e9, 19.6)]:
        x = H * nu / (K_B * T)
        b = 2 * H * nu ** 3 / c_light ** 2 / (math.exp(x) - 1)
        assert abs(lib.dgo_B_nu(nu, T) / b - 1) <= 1e-14
        assert abs(lib.dgo_bnu_prime_RJ(nu) / (2 * K_B * nu ** 2 / c_light ** 2) - 1) <= 1e-15
    from dang_amd.api import BandInfo, DangComps, DangData
    bands = [BandInfo("b1", 30.0), BandInfo("b2", 143.0)]
    comp = DangComps(label="tcmb", type="T_cmb", nu_ref=100.0, nindices=1, sample_amplitude=False,
                     indices=np.full((1, 1, 4), 2.7255), amplitude=np.zeros((1, 4)))
    dd = DangData(sig_map=np.zeros((2, 1, 4)), rms_map=np.ones((2, 1, 4)), masks=np.ones((1, 4)))
    o = O.Oracle(bands, [comp], dd)
    for j, nu in enumerate((30e9, 143e9)):
        x = H * nu / (K_B * 2.7255)
        ref = x / (math.exp(x) - 1) * 2.7255 * 1e6
        assert abs(o.eval_sed_map(0, j, 1)[0] / ref - 1) <= 1e-14
        assert o.L.dgo_eval_signal(o.c, 0, j, 0, 1, None) == o.eval_sed_map(0, j, 1)[0]   # bare sed (:770-771)


def test_return_poltype_flag():
    import dang_amd as da
    assert da.return_poltype_flag("T") == [1]
    assert da.return_poltype_flag("Q+U") == [8]
    assert da.return_poltype_flag("T,Q+U") == [1, 8]
    assert da.return_poltype_flag("T,Q,U") == [1, 2, 4]
    assert da.return_poltype_flag("T+Q+U") == []      # quirk 1: flag 0 can never be matched


def test_unit_conversions_closed_forms_and_identities():
    """a2t / a2f / f2t (src/dang_bp_mod.f90:181-274) against closed forms evaluated independently in Python, for a delta
    band and for a bandpass: a2t = (e^y-1)^2/(y^2 e^y); a2f = 2 k nu^2/c^2 * 1e14 with the reference's SINGLE-precision
    literal (1e14 -> 100000000376832); f2t = 1e-14 / B'_nu(T_CMB); and the identity a2f * f2t = a2t * (1e14f/1e14)
    (B'_RJ / B'_nu = 1/a2t), which ties the three together without reference to any of them."""
    import math
    from dang_amd.api import BandInfo, DangComps, DangData
    h, k, c, T = 1.0545726691251021e-34 * 2.0 * math.pi, 1.3806503e-23, 2.99792458e8, 2.7255
    nu0 = np.array([90.0, 100.0, 110.0]) * 1e9
    tau = np.array([0.2, 0.5, 0.3])
    bands = [BandInfo(label="d", nu_c=143.0), BandInfo(label="b", nu_c=100.0, id="hfi", nu0=nu0, tau0=tau)]
    comp = DangComps(label="cmb", type="cmb", nu_ref=100.0, nindices=0)
    dd = DangData(sig_map=np.zeros((2, 1, 4)), rms_map=np.ones((2, 1, 4)), masks=np.ones((1, 4)))
    orc = O.Oracle(bands, [comp], dd)
    f14 = float(np.float32(1e14))
    assert f14 == 100000000376832.0

    def closed(nu):
        y = h * nu / (k * T)
        a2t = (math.exp(y) - 1.0) ** 2 / (y * y * math.exp(y))
        a2f = 2.0 * k * nu * nu / (c * c)
        bprime = (2.0 * h * nu ** 3) / (c * c * (math.exp(y) - 1.0)) * (math.exp(y) / (math.exp(y) - 1.0)) * h * nu / (k * T * T)
        return a2t, a2f, 1.0 / bprime
    a2t, a2f, ib = closed(143e9)
    assert abs(orc.a2t(0) - a2t) <= 1e-14 * a2t
    assert abs(orc.a2f(0) - a2f * f14) <= 1e-14 * a2f * f14
    assert abs(orc.f2t(0) - ib * 1e-14) <= 1e-14 * ib * 1e-14
    assert abs(orc.a2f(0) * orc.f2t(0) / orc.a2t(0) - f14 / 1e14) <= 1e-14
    parts = [closed(v) for v in nu0]
    assert abs(orc.a2t(1) - sum(t * p[0] for t, p in zip(tau, parts))) <= 1e-14 * orc.a2t(1)
    assert abs(orc.a2f(1) - sum(t * p[1] for t, p in zip(tau, parts)) * f14) <= 1e-14 * orc.a2f(1)
    assert abs(orc.f2t(1) - sum(t * p[2] for t, p in zip(tau, parts)) * 1e-14) <= 1e-14 * orc.f2t(1)
    # normalize_bandpass (:62-81)
    out = np.empty(3)
    O.lib().dgo_normalize_bandpass(O._p(np.array([2.0, 5.0, 3.0])), 3, O._p(out))
    assert np.array_equal(out, np.array([2.0, 5.0, 3.0]) / 10.0)
