"""Oracle self-consistency for the global-amplitude components (template / monopole / hi_fit) inside a CG
group (src/dang_cg_mod.f90:522-587, 717-768, 833-893, 1045-1096, 1244-1279, 1355-1393)."""
import numpy as np

from dang_amd import _lib as L
from dang_amd.api import DangComps

import oracle_ffi as O
from util import make_case, relmax


def add_globals(dpar, ddata, bands, comps, which=("monopole", "hi_fit"), group=1, skip_band0=False, fit_bands=None):
    """skip_band0: do not fit band 0 for any of them.  A global component fitted at EVERY band is exactly degenerate
    with a diffuse component of pixel-independent SED in the same group (the CMB): g_j*s_j proportional to the CMB's
    SED is absorbed by -tmpl(i)*const in the CMB amplitude.  CG returns some point of that valley; a direct solve
    needs the system to be regular."""
    nb, nmaps, npix = ddata.sig_map.shape
    rng = np.random.default_rng(11)
    sky_add = np.zeros_like(ddata.sig_map)
    from dang_amd.synth import H, K_B
    for w in which:
        corr = [True] * nb
        if w == "template" or skip_band0:
            corr[0] = False
        if fit_bands is not None:   # fit only these bands (a well-posed fit: few global amplitudes)
            corr = [j in fit_bands for j in range(nb)]
        tmpl = np.zeros((nmaps, npix))
        ta = np.zeros((nmaps, nb))
        if w == "monopole":
            tmpl[0] = 1.0                                   # src/dang_component_mod.f90:593-595
            truth = rng.normal(0.0, 5.0, nb)
            if skip_band0:
                truth[0] = 0.0
            ta_true = np.zeros((nmaps, nb)); ta_true[0] = truth
            sky_add[:, 0, :] += truth[:, None]
            c = DangComps(label="mono", type="monopole", nu_ref=100.0, cg_group=group, nindices=0, nfit=sum(corr), corr=corr,
                          template=tmpl, template_amplitudes=ta, amplitude=np.zeros((nmaps, npix)))
        elif w == "hi_fit":
            tmpl[0] = rng.uniform(0.2, 1.0, npix)
            T = np.full((1, nmaps, npix), 18.0)
            truth = rng.uniform(0.5, 2.0, nb) * 1e-6
            if skip_band0:
                truth[0] = 0.0
            ta_true = np.zeros((nmaps, nb)); ta_true[0] = truth
            for j, b in enumerate(bands):
                nu = b.nu_c * 1e9
                x = H * nu / (K_B * 18.0)
                sky_add[j, 0, :] += truth[j] * tmpl[0] * (x / (np.exp(x) - 1.0) * 18.0 * 1e6)
            c = DangComps(label="hi", type="hi_fit", nu_ref=100.0, cg_group=group, nindices=1, ind_label=["T"],
                          sample_index=[False], index_mode=[2], lnl_type=["chisq"], prior_type=["uniform"],
                          gauss_prior=[[18.0, 1.0]], uni_prior=[[5.0, 40.0]], step_size=[0.1], pol_flag=[[L.FLAG_T]],
                          nfit=sum(corr), corr=corr, template=tmpl, template_amplitudes=ta, indices=T,
                          amplitude=np.zeros((nmaps, npix)))
        else:  # polarisation template, fitted under Q+U with ONE amplitude per band for Q and U (:1380-1382)
            tmpl[1], tmpl[2] = rng.normal(0, 1, npix), rng.normal(0, 1, npix)
            truth = rng.normal(0.0, 3.0, nb)
            truth[~np.asarray(corr)] = 0.0                   # unfitted bands (corr false) carry nothing
            ta_true = np.zeros((nmaps, nb)); ta_true[1] = truth; ta_true[2] = truth
            for k in (1, 2):
                sky_add[:, k, :] += truth[:, None] * tmpl[k][None, :]
            c = DangComps(label="tmpl", type="template", nu_ref=100.0, cg_group=group, nindices=0, nfit=sum(corr), corr=corr,
                          template=tmpl, template_amplitudes=ta, amplitude=np.zeros((nmaps, npix)))
        c.truth_ta = ta_true
        comps.append(c)
    ddata.sig_map = ddata.sig_map + sky_add
    return sky_add


def _noise_free(case, which, group, flag):
    dpar, ddata, bands, comps, meta = case
    o = O.Oracle(bands, comps, ddata)
    sky, _ = o.sky_model()                       # diffuse components at truth
    ddata.sig_map = sky.copy()
    add_globals(dpar, ddata, bands, comps, which, group)
    return O.Oracle(bands, comps, ddata)


def test_templates_rhs_is_A_times_truth_and_cg_recovers_it():
    for which, group, flag in ((("monopole", "hi_fit"), 1, L.FLAG_T), (("template",), 2, L.FLAG_QU)):
        case = make_case("C2", nside=2, start="truth")
        dpar, ddata, bands, comps, meta = case
        o = _noise_free(case, which, group, flag)
        n = o.group_size(group, flag)
        nglob = sum(c.nfit for c in comps if c.type in which)
        assert n == 3 * (2 if flag == 8 else 1) * meta["npix"] + nglob
        # truth vector in the reference's packing: diffuse blocks then the global entries of the appended components
        x_true = o.initialize_x(group, flag)
        off = n - nglob
        for c in comps:
            if c.type in which:
                k = 1 if flag == 8 else 0
                vals = [c.truth_ta[k, j] for j in range(meta["nbands"]) if c.corr[j]]
                x_true[off:off + c.nfit] = vals
                off += c.nfit
        b = o.compute_rhs(group, flag)
        Ax = o.compute_Ax(group, flag, x_true)
        assert np.abs(b - Ax).max() <= 1e-9 * np.abs(b).max()
        rng = np.random.default_rng(2)
        u, v = rng.standard_normal(n), rng.standard_normal(n)
        Au, Av = o.compute_Ax(group, flag, u), o.compute_Ax(group, flag, v)
        assert abs(v @ Au - u @ Av) <= 1e-10 * abs(v @ Au)     # symmetric (the monopole template is 1)
        # CG from zero: the amplitudes themselves are poorly determined (a per-band monopole is nearly degenerate
        # with per-pixel diffuse components on a 48-pixel sky), so check what IS determined: the recovered model
        it, x, trace = o.cg_search(group, flag, b, "optimize", None, np.zeros(n), 5000, 1e-18)
        assert trace[0] > 1e3 * np.nanmin(trace)
        o.L.dgo_unpack_amplitudes(o.c, group, flag, O._p(x))
        sky, res = o.sky_model()
        ok = ddata.masks[0] != 0
        planes = [0] if flag == L.FLAG_T else [1, 2]
        for k in planes:
            assert np.abs(res[:, k, :][:, ok]).max() <= 1e-5 * np.abs(ddata.sig_map[:, k, :]).max()
        if "template" in which:   # the polarisation template IS well determined; Q and U share one amplitude (:1380-1382)
            l = len(comps) - 1
            ta = o.template_amplitudes(l)
            for j in range(meta["nbands"]):
                if comps[l].corr[j]:
                    assert abs(ta[1, j] - comps[l].truth_ta[1, j]) <= 1e-6 * np.abs(comps[l].truth_ta).max()
                    assert ta[1, j] == ta[2, j]


def test_monopole_sets_the_band_offsets_in_update_sky_model():
    case = make_case("C2", nside=2, start="truth")
    dpar, ddata, bands, comps, meta = case
    add_globals(dpar, ddata, bands, comps, ("monopole",), 1)
    comps[-1].template_amplitudes = comps[-1].truth_ta.copy()
    o = O.Oracle(bands, comps, ddata)
    sky, res = o.sky_model()
    assert np.allclose(o.offset, comps[-1].truth_ta[0])          # src/dang_data_mod.f90:357-361
    # residual = (sig - offset)/gain - sky: the monopole is absorbed by the offset, not by the sky model
    chisq, _ = o.chisq(1, 1, float((ddata.masks[0] != 0).sum()), sky)   # plane 1 only
    assert 0.5 < chisq < 1.5


def test_direct_block_solve_equals_the_converged_cg_over_random_models():
    """The direct per-pixel block solve (what the GPU runs) must be the fixed point of the reference's cg_search for
    every way the data are prepared in compute_rhs (src/dang_cg_mod.f90:367-460): gain on T without offset removal,
    non-member components of every type (diffuse, T_cmb, template / monopole / hi_fit with amplitudes on fitted AND
    unfitted bands -- the latter removed twice), masked pixels.  24 seeded models, optimize mode, CG run to 1e-16."""
    from dang_amd.api import DangComps
    for seed in range(24):
        rng = np.random.default_rng(seed)
        nb = int(rng.choice([4, 5, 6]))
        comps_l = ["synch"] + list(rng.permutation(["cmb", "dust", "ff"])[: int(rng.integers(0, min(3, nb - 2)))])
        extra = list(rng.permutation(["template", "monopole", "hi_fit", "tcmb", "none"])[:2])

        def tweak(dpar, ddata, bands, comps):
            npix = ddata.sig_map.shape[-1]
            ddata.gain = rng.uniform(0.8, 1.2, nb)
            ddata.offset = rng.normal(0, 3, nb)
            for e in extra:
                if e in ("template", "monopole", "hi_fit"):
                    add_globals(dpar, ddata, bands, comps, (e,), 7, fit_bands=sorted(rng.choice(nb, 2, replace=False).tolist()))
                    ta = rng.normal(0, 2, comps[-1].truth_ta.shape)
                    if e == "hi_fit":
                        ta *= 1e-6
                    if e != "template":
                        ta[1:] = 0
                    comps[-1].template_amplitudes = ta
                elif e == "tcmb":
                    comps.append(DangComps(label="tcmb", type="T_cmb", nu_ref=100.0, cg_group=9, sample_amplitude=False,
                                           nindices=1, ind_label=["T"], sample_index=[False], index_mode=[1],
                                           lnl_type=["chisq"], prior_type=["uniform"], gauss_prior=[[0.5, 1.0]],
                                           uni_prior=[[0.0, 10.0]], step_size=[0.0], pol_flag=[[L.FLAG_T]],
                                           amplitude=np.zeros((3, npix)), indices=np.full((1, 3, npix), 0.5)))
            ddata.masks[0, rng.integers(0, npix, 3)] = 0.0
        case = make_case(None, nside=2, nbands=nb, comps=comps_l, nmaps=3, tweak=tweak, start="truth")
        dpar, ddata, bands, comps, meta = case
        for group, flag in ((1, L.FLAG_T), (2, L.FLAG_QU)):
            o1, o2 = O.Oracle(bands, comps, ddata), O.Oracle(bands, comps, ddata)
            o1.amp_sample_direct(group, flag, "optimize", 3, 4, "reference")
            it = o2.amp_sample_cg(group, flag, "optimize", 3, 4, i_max=3000, converge=1e-16)
            assert it < 3000, (seed, group)
            for l, c in enumerate(comps):
                if c.cg_group == group and c.type in ("power-law", "mbb", "freefree", "cmb"):
                    assert relmax(o1.amplitude(l), o2.amplitude(l)) <= 1e-7, (seed, group, l, comps_l, extra)


def test_near_singular_monopole_system_rounding_floor():
    """Why the direct solve's residual is judged against eps*|A||x| and not against |b| alone: a monopole and the HI
    fit both fitted at bands 0 and 1 beside synchrotron + free-free + CMB (fuzz seed 288 of tests/test_gpu_fuzz.py, 48
    pixels) give a system of condition number ~1e18 whose solution carries amplitudes ~1e11.  LAPACK's dense solve of
    that system -- and that solve refined with extended-precision residuals -- leaves |A x - b| of 1e-6..1e-5 |b| on the
    HI rows: the terms of those rows are ~1e21, and eps * 1e21 is the floor of ANY fp64 solver."""
    rng = np.random.default_rng(5000 + 288)
    nb = int(rng.choice([4, 5, 6, 8]))
    pool = ["cmb", "synch", "dust", "ff"]
    comps_l = ["synch"] + list(rng.permutation([p for p in pool if p != "synch"])[: int(rng.integers(0, min(3, nb - 3) + 1))])
    pol = bool(rng.integers(0, 2))
    which = ("template",) if pol else tuple(rng.permutation(["monopole", "hi_fit"])[: int(rng.integers(1, 3))])
    assert not pol and set(which) == {"monopole", "hi_fit"}
    fit = sorted(rng.choice(nb, size=int(rng.integers(1, 3)), replace=False).tolist())
    ml_mode = str(rng.choice(["sample", "optimize"]))

    def tweak(dpar, ddata, bands, comps):
        add_globals(dpar, ddata, bands, comps, which, 1, fit_bands=fit)
    case = make_case(None, nside=int(rng.choice([2, 4])), nbands=nb, comps=comps_l, nmaps=3, tweak=tweak, start="truth")
    dpar, ddata, bands, comps, meta = case
    orc = O.Oracle(bands, comps, ddata)
    n = orc.group_size(1, L.FLAG_T)
    b = orc.compute_rhs(1, L.FLAG_T)
    if ml_mode == "sample":
        b = b + orc.compute_sample_vector(1, L.FLAG_T, orc.draw_eta(L.FLAG_T, 8, 9))
    A = np.zeros((n, n))
    for i in range(n):
        e = np.zeros(n)
        e[i] = 1.0
        A[:, i] = orc.compute_Ax(1, L.FLAG_T, e)
    live = np.abs(A).sum(1) > 0                      # rows / columns of masked pixels are zero
    A, b = A[np.ix_(live, live)], b[live]
    assert np.linalg.cond(A) > 1e16
    x = np.linalg.solve(A, b)
    xl = x.astype(np.longdouble)
    for _ in range(3):                               # iterative refinement with extended-precision residuals
        xl = xl + np.linalg.solve(A, (b.astype(np.longdouble) - A.astype(np.longdouble) @ xl).astype(np.float64))
    eps = np.finfo(float).eps
    for sol in (x, xl.astype(np.float64)):
        r = np.abs(A @ sol - b)
        mag = np.abs(A) @ np.abs(sol)
        assert np.abs(sol).max() > 1e10
        assert (r[-4:] / np.abs(b[-4:])).max() > 1e-7          # far above 1e-7 |b| ...
        assert np.all(r <= 16 * eps * mag + 1e-9 * np.abs(b).max())   # ... and at the rounding floor of the rows' terms
