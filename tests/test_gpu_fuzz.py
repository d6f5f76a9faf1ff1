"""Randomised parity: 64 seeded model configurations (band count, component subset, Stokes planes, likelihood and
prior types, step sizes, tight/wide hard bounds, NUMSAMPLE, sample/optimize, some bandpass-integrated bands) run
through one amplitude pass + one sweep of every sampled index on the GPU and in the oracle.  The configurations
are drawn from a fixed seed, so the test is deterministic; it exists to reach kernel-dispatch combinations the
hand-written cases do not (register-resident / LDS / bandpass / generic chains, group sizes, plane counts)."""
import os

import numpy as np
import pytest

import dang_amd as da
from dang_amd import _lib as L

import oracle_ffi as O
from util import MAPN, make_case, pair, relmax

pytestmark = pytest.mark.gpu

POOL = ["cmb", "synch", "dust", "ff", "ame", "dust2"]


def _draw(seed):
    rng = np.random.default_rng(1000 + seed)
    nb = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 10, 12]))
    ncomp = int(rng.integers(1, min(nb, 5) + 1))
    comps = ["synch"] + [c for c in rng.permutation(POOL[:1] + POOL[2:])[: ncomp - 1]]
    nmaps = int(rng.choice([1, 3]))
    cfg = dict(nbands=nb, comps=comps, nmaps=nmaps, nside=int(rng.choice([1, 2, 4])),
               ml_mode=str(rng.choice(["sample", "optimize"])), nsample=int(rng.choice([1, 3, 10, 25])),
               lnl=str(rng.choice(["chisq", "chisq", "marginal", "prior"])),
               prior=str(rng.choice(["gaussian", "uniform", "jeffreys"])),
               step=float(rng.choice([0.1, 0.5, 2.0])), tight=bool(rng.integers(0, 2)),
               bandpass=bool(rng.integers(0, 3) == 0), fluct=str(rng.choice(["reference", "correct"])), seed=int(seed))
    return cfg


@pytest.mark.parametrize("seed", range(int(os.environ.get("DANGX_FUZZ_SEEDS", "64"))))
def test_random_configuration_matches_oracle(built, seed):
    cfg = _draw(seed)
    if cfg["lnl"] == "marginal" and cfg["ml_mode"] == "optimize":
        # for ONE pixel the marginal form -1/2 TNd^2/TNT (src/dang_lnl_mod.f90:113-122) does not depend on the model at
        # all (m cancels), so `diff > 0` is the sign of rounding noise: no parity to test (cf. the full-sky case)
        pytest.skip("degenerate: accept test is the sign of rounding noise")

    def tweak(dpar, ddata, bands, comps):
        dpar.ml_mode, dpar.nsample, dpar.fluct_mode = cfg["ml_mode"], cfg["nsample"], cfg["fluct"]
        for c in comps:
            c.lnl_type = [cfg["lnl"]] * c.nindices
            c.prior_type = [cfg["prior"]] * c.nindices
            c.step_size = [cfg["step"] * g[1] if g[1] > 0 else 0.0 for g in c.gauss_prior]
            if cfg["tight"]:   # hard bounds one prior sigma wide: many proposals fall outside (:415)
                c.uni_prior = [[g[0] - g[1], g[0] + g[1]] if g[1] > 0 else u for g, u in zip(c.gauss_prior, c.uni_prior)]
        if cfg["bandpass"]:
            rng = np.random.default_rng(cfg["seed"])
            for b in bands[::2]:
                nu = b.nu_c * 1e9 * np.linspace(0.92, 1.08, 5)
                tau = rng.uniform(0.3, 1.0, nu.size)
                b.id, b.nu0, b.tau0 = "bp", nu, tau / tau.sum()
    case = make_case(None, nside=cfg["nside"], nbands=cfg["nbands"], comps=cfg["comps"], nmaps=cfg["nmaps"], tweak=tweak,
                     start="truth", nsample=cfg["nsample"])
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for g in dpar.cg_groups:
        for f in g.pol_flag:
            _, bad = eng.amp_sample(g.cg_group, f, cfg["ml_mode"], 11, 100 + f, fluct_mode=cfg["fluct"])
            obad = orc.amp_sample_direct(g.cg_group, f, cfg["ml_mode"], 11, 100 + f, cfg["fluct"])
            assert bad == obad, cfg
    # random component subsets include near-degenerate ones (free-free + AME + synchrotron on a few bands): the block
    # solve then amplifies rounding by its condition number, visible as amplitudes far above the injected ~1e2
    amax = max(np.abs(orc.amplitude(l)).max() for l in range(len(comps)))
    tol = 1e-8 * max(1.0, amax / 1e3) ** 2
    for l in range(len(comps)):
        a, b = eng.get_amplitude(l), orc.amplitude(l)
        assert np.isfinite(a).all() and np.abs(a - b).max() <= tol * max(amax, 1e-300), (cfg, l, amax)
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if not c.sample_index[j]:
                continue
            for f in c.pol_flag[j]:
                ag = eng.index_sample(l, j, MAPN[f], cfg["nsample"], cfg["ml_mode"], 11, 500 + 10 * l + j + f)
                ao = orc.sample_index_mh(l, j, MAPN[f], cfg["nsample"], cfg["ml_mode"], 11, 500 + 10 * l + j + f)
                assert ag == ao or amax >= 1e4, (cfg, l, j, f, ag, ao)
            if amax < 1e4:   # (an accept decision can flip when the amplitudes themselves carry 1e-6 differences)
                assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-11, (cfg, l, j)
    s = eng.sky_model_chisq(1, meta["nmaps"])
    ochisq, _ = orc.chisq(1, meta["nmaps"], ddata.nump)
    # (nb == nc: the fit is exact and chi^2 is rounding noise ~1e-15 and below, hence the absolute floor)
    assert abs(s / meta["nbands"] / ddata.nump - ochisq) <= 1e-8 * abs(ochisq) * max(1.0, amax / 1e3) ** 2 + 1e-10, cfg


def _packed(eng, comps, group, flag, nbands):
    planes = {L.FLAG_T: [0], L.FLAG_QU: [1, 2]}[flag]
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


@pytest.mark.parametrize("seed", range(int(os.environ.get("DANGX_FUZZ_SEEDS", "32"))))
def test_random_template_group_direct_solve(built, seed):
    """Schur-complement solve of groups with global-amplitude members over random models: the answer satisfies the
    reference's linear system (through the oracle's compute_Ax / compute_rhs / compute_sample_vector)."""
    from test_oracle_templates_cpu import add_globals
    rng = np.random.default_rng(5000 + seed)
    nb = int(rng.choice([4, 5, 6, 8]))
    pool = ["cmb", "synch", "dust", "ff"]
    comps_l = ["synch"] + list(rng.permutation([p for p in pool if p != "synch"])[: int(rng.integers(0, min(3, nb - 3) + 1))])
    pol = bool(rng.integers(0, 2))
    which = ("template",) if pol else tuple(rng.permutation(["monopole", "hi_fit"])[: int(rng.integers(1, 3))])
    group, flag = (2, L.FLAG_QU) if pol else (1, L.FLAG_T)
    fit = sorted(rng.choice(nb, size=int(rng.integers(1, 3)), replace=False).tolist())
    ml_mode = str(rng.choice(["sample", "optimize"]))

    def tweak(dpar, ddata, bands, comps):
        add_globals(dpar, ddata, bands, comps, which, group, fit_bands=fit)
    case = make_case(None, nside=int(rng.choice([2, 4])), nbands=nb, comps=comps_l, nmaps=3, tweak=tweak, start="truth")
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    b = orc.compute_rhs(group, flag)
    if ml_mode == "sample":
        b = b + orc.compute_sample_vector(group, flag, orc.draw_eta(flag, 8, 9))
    it, bad = eng.amp_sample(group, flag, ml_mode, 8, 9, solver="direct")
    assert bad == 0 and it <= 0
    x = _packed(eng, comps, group, flag, nb)
    assert x.size == eng.group_size(group, flag) and np.isfinite(x).all()
    Ax = orc.compute_Ax(group, flag, x)
    # row-wise: relative to the row's right-hand side, with a floor tied to the largest one (a template's entries have
    # both signs, so A|x| is no bound on the size of the terms that cancel in a row).  The GLOBAL rows additionally get
    # the rounding floor of their own terms, 64 eps (|A||x|)_row: a monopole fitted beside a pixel-independent SED is
    # nearly degenerate with the diffuse members, the system is then numerically singular (cond ~ 1e18, amplitudes
    # ~1e11) and NO fp64 solver can push those rows below eps |A||x| -- LAPACK's dense solve of the same system leaves
    # the same residual (tests/test_oracle_templates_cpu.py::test_near_singular_monopole_system_rounding_floor).
    R = sum(c.nfit for c in comps if c.cg_group == group and c.type in ("template", "monopole", "hi_fit"))
    mag = np.zeros(x.size)
    for i in np.nonzero(x)[0]:
        e = np.zeros(x.size)
        e[i] = 1.0
        mag[-R:] += np.abs(orc.compute_Ax(group, flag, e)[-R:]) * abs(x[i])
    # (the floor constant: a global row adds one term per pixel and plane, so its rounding floor grows with their number; seed 928 of a
    # 1000-seed run -- hi_fit + monopole beside cmb and free-free, amplitudes of 1e13 -- sits at 2.3 x 16 eps mag with these kernels and
    # at 7 x with the run-time-typed ones, its backward error at 5e-10)
    tol = 1e-7 * np.abs(b) + 1e-9 * np.abs(b).max() + 64 * np.finfo(float).eps * mag
    (resid, backward), nref = eng.schur_info()
    assert np.all(np.abs(Ax - b) <= tol), (seed, which, fit, comps_l, it, (np.abs(Ax - b) / tol).max(), resid, backward, nref)
    # the library's own account of the solve: relative to the size of the rows' terms the residual is at rounding level;
    # relative to b it is what the conditioning leaves, and the oracle's operators see the same order of magnitude
    worst = (np.abs(Ax - b)[-R:] / np.maximum(np.abs(b)[-R:], 1e-300)).max()
    # (1e-15 is typical; the refinement uses the LU of the CANCELLED Schur matrix, so on the most nearly singular systems
    # of a 1000-seed run -- seed 928 -- it stalls at 3e-10 of the rows' terms, still far below the reference's CG stop.
    # A minimal-residual (GMRES-like) combination of the refinement iterates was tried for those: it fits the rounding
    # noise of the residual evaluation and leaves the rows WORSE by the oracle's measure; not kept.)
    assert backward <= 1e-9, (seed, which, resid, backward, nref)
    assert resid <= 100 * worst + 1e-9 and worst <= 100 * resid + 1e-9, (seed, which, resid, worst, nref)
    rel, relg = eng.amp_residual(group, flag, ml_mode, 8, 9)
    # (the whole-vector figure is bounded by the worst global row -- seed 288 of a wide run is the condition-6e18 system of
    # DESIGN section 3, whose rows sit at their rounding floor of 1e-5 |b|)
    assert relg <= 100 * worst + 1e-9 and rel <= 1e-6 + 10 * worst, (seed, which, rel, relg, worst)


def _draw_iteration(seed):
    rng = np.random.default_rng(9000 + seed)
    nb = int(rng.choice([4, 6, 7, 9, 11, 12, 14, 16, 18]))
    ncomp = int(rng.integers(2, min(nb - 1, 6) + 1))
    comps = ["synch"] + [str(c) for c in rng.permutation(POOL[:1] + POOL[2:])[: ncomp - 1]]
    return dict(nbands=nb, comps=comps, nmaps=int(rng.choice([1, 3, 3])), nside=int(rng.choice([2, 4])),
                ml_mode=str(rng.choice(["sample", "sample", "optimize"])), nsample=int(rng.choice([1, 4, 10])),
                prior=str(rng.choice(["gaussian", "gaussian", "uniform"])), step=float(rng.choice([0.2, 0.5, 1.5])),
                tight=bool(rng.integers(0, 3) == 0), seed=int(seed))


@pytest.mark.parametrize("seed", range(int(os.environ.get("DANGX_GIBBS_FUZZ_SEEDS", "8"))))
def test_random_model_gibbs_iterations_match_the_reference_order_loop(built, seed):
    """Whole Gibbs iterations (da.gibbs_iteration: plane-set launches where dangx_plan_fusion allows them, specialised at run time
    for whatever band count / member list / sweep list the model has; for every other model the middle iteration through the
    two-call halves) on random models against the oracle run in the REFERENCE's order (every solve, then every sweep)."""
    cfg = _draw_iteration(seed)

    def tweak(dpar, ddata, bands, comps):
        dpar.ml_mode, dpar.nsample = cfg["ml_mode"], cfg["nsample"]
        for c in comps:
            c.prior_type = [cfg["prior"]] * c.nindices
            c.step_size = [cfg["step"] * g[1] if g[1] > 0 else 0.0 for g in c.gauss_prior]
            if cfg["tight"]:
                c.uni_prior = [[g[0] - g[1], g[0] + g[1]] if g[1] > 0 else u for g, u in zip(c.gauss_prior, c.uni_prior)]
    case = make_case(None, nside=cfg["nside"], nbands=cfg["nbands"], comps=cfg["comps"], nmaps=cfg["nmaps"], tweak=tweak,
                     start="truth", nsample=cfg["nsample"])
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for it in (1, 2, 3):
        if it == 2 and seed % 2:
            # every other model: the middle iteration as the two-call seam runs it -- the solves alone (with their chi^2), then the
            # sweeps of each plane set through dangx_plane_sweeps_sample (one launch where the model allows it, the calls otherwise)
            for g in dpar.cg_groups:
                for f in g.pol_flag:
                    eng.amp_sample(g.cg_group, f, dpar.ml_mode, dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f))
            for g in dpar.cg_groups:
                for f in g.pol_flag:
                    sw = [(l, j, da.stream_id(it, 1, l, j, f)) for l, c in enumerate(comps) for j in range(c.nindices)
                          if c.cg_group == g.cg_group and c.sample_index[j] and f in c.pol_flag[j]]
                    if sw:
                        eng.plane_sweeps_sample(f, sw, dpar.nsample, dpar.ml_mode, dpar.seed)
            da.compute_chisq(ddata)
        else:
            da.gibbs_iteration(dpar, ddata, it)
        for g in dpar.cg_groups:
            for f in g.pol_flag:
                orc.amp_sample_direct(g.cg_group, f, dpar.ml_mode, dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f), "reference")
        for l, c in enumerate(comps):
            for j in range(c.nindices):
                if c.sample_index[j]:
                    for f in c.pol_flag[j]:
                        orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, dpar.ml_mode, dpar.seed, da.stream_id(it, 1, l, j, f))
    amax = max(np.abs(orc.amplitude(l)).max() for l in range(len(comps)))
    if amax >= 1e4:
        pytest.skip("near-degenerate component subset (amplitudes %.1e): accept decisions are not comparable" % amax)
    # the block solve amplifies rounding by its condition number, visible as amplitudes above the injected <~ 3e2 (see above);
    # three iterations feed the amplified differences back through the index sweeps
    tol = 1e-8 * max(1.0, amax / 3e2) ** 2
    for l, c in enumerate(comps):
        da_max = float(np.abs(eng.get_amplitude(l) - orc.amplitude(l)).max())
        assert da_max <= tol * max(amax, 1.0), (cfg, l, da_max, amax)
        if c.nindices:
            di_max = float(np.abs(eng.get_indices(l) - orc.indices(l)).max())
            assert di_max <= 1e-11, (cfg, l, di_max, eng.rtc_kernels())
    ochisq, _ = orc.chisq(1, meta["nmaps"], ddata.nump)
    assert abs(ddata.chisq - ochisq) <= tol * abs(ochisq) + 1e-10, cfg
    print("gibbs fuzz %d: %d bands, %s, nmaps %d -> %s" % (seed, cfg["nbands"], "+".join(cfg["comps"]), cfg["nmaps"], eng.rtc_kernels()))
