"""GPU parity for SURVEY 8f rank 4 (first half): index sampling with sample_nside < nside -- the device udgrade
primitives and the coarse chain against the oracle's restatement (reference behaviour reproduced literally, including
its reading of the full-resolution amplitude / index / mask arrays at the coarse pixel number)."""
import numpy as np
import pytest

import dang_amd as da
from dang_amd import _lib as L

import oracle_ffi as O
from util import make_case, pair

pytestmark = pytest.mark.gpu


def test_device_udgrade_matches_the_restatement(built):
    eng, _ = pair(make_case("C1", nside=4))
    rng = np.random.default_rng(5)
    for ni, no in ((16, 4), (8, 1), (2, 8), (4, 32)):
        m = rng.normal(size=12 * ni * ni)
        m[rng.integers(0, m.size, 5)] = -1.6375e30
        assert np.array_equal(eng.udgrade(0, m, ni, no), O.udgrade(0, m, ni, no))
        rms = rng.uniform(0.5, 2.0, m.size)
        assert np.array_equal(eng.udgrade(1, rms, ni, no), O.udgrade(1, rms, ni, no))
        mask = (rng.uniform(size=m.size) > 0.4).astype(float)
        assert np.array_equal(eng.udgrade(2, mask, ni, no), O.udgrade(2, mask, ni, no))


@pytest.mark.parametrize("lnl,ml_mode,cnside,prior", [("chisq", "sample", 4, None), ("chisq", "optimize", 8, None),
                                                      ("chisq", "sample", 2, None),   # 64 children per coarse pixel: one round of k_udgrade_wave
                                                      ("chisq", "sample", 1, None),   # 256: four rounds
                                                      ("marginal", "sample", 2, None), ("prior", "sample", 4, None),
                                                      ("chisq", "sample", 2, "jeffreys"), ("chisq", "optimize", 4, "jeffreys")])
def test_coarse_index_sampling_matches_oracle(built, lnl, ml_mode, cnside, prior):
    nside = 16 if prior is None else 8

    def tweak(dpar, ddata, bands, comps):
        for c in comps:
            c.lnl_type = [lnl] * c.nindices
            c.sample_nside = [cnside] * c.nindices
            if prior:        # eval_jeffreys_prior is non-trivial for the component labelled 'synch' only; the others
                c.prior_type = [prior] * c.nindices   # get log(0) = -inf and never move (reproduced)
    case = make_case("C2", nside=nside, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if not c.sample_index[j]:
                continue
            f = c.pol_flag[j][0]
            map_n = {1: 1, 8: -1}[f]
            s = da.stream_id(2, 1, l, j, f)
            ag = eng.index_sample_coarse(l, j, map_n, 10, ml_mode, 7, s, cnside)
            ao = orc.sample_index_mh_coarse(l, j, map_n, 10, ml_mode, 7, s, nside, cnside)
            assert ao >= 0 and ag == ao, (l, j, ag, ao)
            a, b = eng.get_indices(l), orc.indices(l)
            assert np.abs(a - b).max() <= 1e-12
            planes = [0] if f == 1 else [1, 2]
            # every full-resolution pixel carries its coarse parent's value: (nside/cnside)^2 copies of each
            vals, counts = np.unique(a[j, planes[0]], return_counts=True)
            assert np.all(counts % ((nside // cnside) ** 2) == 0)
            if f == 8:
                assert np.array_equal(a[j, 1], a[j, 2])
    # the state is usable afterwards: chi^2 through the explicit pass, and a regular sweep
    s = eng.sky_model_chisq(1, 3)
    osky, _ = orc.sky_model()
    ochisq, _ = orc.chisq(1, 3, ddata.nump, osky)
    assert abs(s / meta["nbands"] / ddata.nump - ochisq) <= 1e-9 * ochisq


def test_sample_spectral_parameters_dispatches_the_coarse_mode(built):
    def tweak(dpar, ddata, bands, comps):
        comps[1].sample_nside = [4]           # synch beta (T) at Nside 4, everything else at full resolution
    case = make_case("C2", nside=8, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    da.sample_spectral_parameters(dpar, ddata, it=2)
    beta = eng.get_indices(1)[0, 0]
    assert len(np.unique(beta)) <= 12 * 4 * 4 and len(np.unique(eng.get_indices(2)[0, 0])) > 12 * 4 * 4
    assert np.isfinite(ddata.chisq)


def test_coarse_mode_on_a_shard_needs_the_sum_over_shards(built):
    case = make_case("C2", nside=8, rank=0, nranks=2)
    eng, _ = pair(case)
    with pytest.raises(da.DangxError, match="sum over the shards"):
        eng.index_sample_coarse(1, 0, 1, 10, "sample", 7, 1, 4)


@pytest.mark.parametrize("nshards", [1, 3])
def test_coarse_sampling_over_several_contexts_of_one_process(built, nshards):
    """The three phases of the coarse sweep (partials / chains / write-back) over pixel-shard contexts of ONE process,
    buffers added in shard order: one shard reproduces the whole-sky call bit for bit, three shards agree with the
    oracle to the parity tolerance (their child sums are associated differently)."""
    nside, cnside = 8, 2

    def tweak(dpar, ddata, bands, comps):
        for c in comps:
            c.sample_nside = [cnside] * c.nindices
    whole = make_case("C2", nside=nside, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = whole
    ref, orc = pair(whole)
    shards = [make_case("C2", nside=nside, start="truth", tweak=tweak, rank=r, nranks=nshards) for r in range(nshards)]
    engs = [da.Engine(x[2], x[3], x[1], npix_global=x[4]["npix_global"], pix0=x[4]["pix0"], device=0) for x in shards]
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if not c.sample_index[j]:
                continue
            f = c.pol_flag[j][0]
            map_n = {1: 1, 8: -1}[f]
            s = da.stream_id(2, 1, l, j, f)
            a_ref = ref.index_sample_coarse(l, j, map_n, 10, "sample", 7, s, cnside)
            a_orc = orc.sample_index_mh_coarse(l, j, map_n, 10, "sample", 7, s, nside, cnside)
            a_multi = da.index_sample_coarse_multi(engs, l, j, map_n, 10, "sample", 7, s, cnside)
            assert a_ref == a_orc == a_multi, (l, j, a_ref, a_orc, a_multi)
    for l, c in enumerate(comps):
        if not c.nindices:
            continue
        got = np.concatenate([e.get_indices(l) for e in engs], axis=-1)
        if nshards == 1:
            assert np.array_equal(got, ref.get_indices(l))
        assert np.abs(got - orc.indices(l)).max() <= 1e-12


def test_coarse_sweep_survives_a_descriptor_update_with_bandpass_bands(built):
    """Regression (round-1 lifetime bug): with bandpass-integrated bands, dangx_set_component marks the bandpass tables
    dirty; the re-upload in sync_model used to free the cached HEALPix index tables and the coarse staging buffers
    without resetting them, so the NEXT coarse sweep ran on freed memory.  Coarse sweep -> new step size -> coarse
    sweep again must equal the oracle; destroying the context afterwards frees the tables exactly once."""
    import ctypes as C
    from dang_amd.api import comp_desc
    nside, cnside = 8, 4

    def tweak(dpar, ddata, bands, comps):
        for b in bands[1::2]:
            b.id = "tophat"
            b.nu0 = b.nu_c * 1e9 * np.linspace(0.9, 1.1, 4)
            b.tau0 = np.full(4, 0.25)
        for c in comps:
            c.sample_nside = [cnside] * c.nindices
    case = make_case("C2", nside=nside, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    l, j, f = 1, 0, comps[1].pol_flag[0][0]
    map_n = {1: 1, 8: -1}[f]
    for rnd in range(3):
        s = da.stream_id(2 + rnd, 1, l, j, f)
        ag = eng.index_sample_coarse(l, j, map_n, 10, "sample", 7, s, cnside)
        ao = orc.sample_index_mh_coarse(l, j, map_n, 10, "sample", 7, s, nside, cnside)
        assert ag == ao, (rnd, ag, ao)
        assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-12, rnd
        # the tuner's effect: a new step size through dangx_set_component (sets bp_dirty) before the next sweep
        comps[l].step_size[j] *= 0.5
        orc._comps[l].step_size[j] = comps[l].step_size[j]
        eng._chk(eng.lib.dangx_set_component(eng.h, l, C.byref(comp_desc(comps[l]))))
    eng.close()
