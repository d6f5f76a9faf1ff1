"""Randomised parity: 64 seeded model configurations (band count, component subset, Stokes planes, likelihood and
prior types, step sizes, tight/wide hard bounds, NUMSAMPLE, sample/optimize, some bandpass-integrated bands) run
through one amplitude pass + one sweep of every sampled index on the GPU and in the oracle.  The configurations
are drawn from a fixed seed, so the test is deterministic; it exists to reach kernel-dispatch combinations the
hand-written cases do not (register-resident / LDS / bandpass / generic chains, group sizes, plane counts)."""
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


@pytest.mark.parametrize("seed", range(64))
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
