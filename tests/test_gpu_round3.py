"""Round-3 additions: the fusion rule of gibbs_iteration against the oracle's reference-order loop (two CG groups on one
plane), the chi^2 ring beyond its capacity, the Schur solve of a template group over several contexts of one process, the
full-sky index mode at a coarser Nside on pixel shards, and C1 at its own Nside against the oracle."""
import copy

import numpy as np
import pytest

import dang_amd as da
from dang_amd import _lib as L

import oracle_ffi as O
from test_oracle_templates_cpu import add_globals
from util import MAPN, make_case, pair

pytestmark = pytest.mark.gpu


def _oracle_iteration(orc, dpar, comps, it):
    """one pass of the main loop in the REFERENCE's order: every solve, then every sweep (src/dang.f90:101-106)"""
    for g in dpar.cg_groups:
        for f in g.pol_flag:
            orc.amp_sample_direct(g.cg_group, f, dpar.ml_mode, dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f), "reference")
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if c.sample_index[j]:
                for f in c.pol_flag[j]:
                    orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, dpar.ml_mode, dpar.seed, da.stream_id(it, 1, l, j, f))


def test_gibbs_iteration_does_not_fuse_when_two_groups_share_a_plane(built):
    """Two CG groups that both solve on the T plane: the reference runs both solves before any sweep, so group 1's solve must
    not be issued with its first sweep (group 3's solve would then see the swept index).  dangx_plan_fusion says so; the
    iteration equals the oracle's reference-order loop; the Q+U group, alone on its planes, is still fused."""
    def tweak(dpar, ddata, bands, comps):
        comps[2].cg_group = 3                     # dust (T) gets its own group on the T plane
        dpar.cg_groups.append(da.DangCGGroup(cg_group=3, pol_flag=[L.FLAG_T]))
    case = make_case("C2", nside=8, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    plan = da.fusable_first_sweeps(dpar, eng)
    assert (1, L.FLAG_T) not in plan and (3, L.FLAG_T) not in plan and plan == {(2, L.FLAG_QU): (4, 0)}
    for it in (2, 3):
        da.gibbs_iteration(dpar, ddata, it)
        _oracle_iteration(orc, dpar, comps, it)
    for l, c in enumerate(comps):
        b = orc.amplitude(l)
        assert np.abs(eng.get_amplitude(l) - b).max() <= 1e-9 * max(np.abs(b).max(), 1.0), l
        if c.nindices:
            assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-12, l


def test_gibbs_iteration_does_not_fuse_behind_an_earlier_sweep_on_the_same_plane(built):
    """A component outside the group, earlier in component_list, with a sampled index on the group's plane: its sweep is the
    first one on that plane in the reference's order, so the group's own first sweep may not be pulled forward."""
    def tweak(dpar, ddata, bands, comps):
        comps[1].cg_group = 3                     # synch (T): sampled index, amplitude solved in another group
        dpar.cg_groups.insert(0, da.DangCGGroup(cg_group=3, pol_flag=[L.FLAG_T], sample=False))
    case = make_case("C2", nside=8, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    assert (1, L.FLAG_T) not in da.fusable_first_sweeps(dpar, eng)
    da.gibbs_iteration(dpar, ddata, 2)
    for g in dpar.cg_groups:
        if g.sample:
            for f in g.pol_flag:
                orc.amp_sample_direct(g.cg_group, f, dpar.ml_mode, dpar.seed, da.stream_id(2, 0, g.cg_group, 0, f), "reference")
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if c.sample_index[j]:
                f = c.pol_flag[j][0]
                orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, dpar.ml_mode, dpar.seed, da.stream_id(2, 1, l, j, f))
    for l, c in enumerate(comps):
        if c.nindices:
            assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-12, l


def test_chi2_ring_beyond_its_capacity(built):
    """More index sweeps than the ring of pending chi^2 partials holds (8) between two queries: the ring flushes itself on
    the way, and the cached value still equals the explicit pass."""
    case = make_case("C2", nside=8, start="truth")
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for g in dpar.cg_groups:
        eng.amp_sample(g.cg_group, g.pol_flag[0], "sample", 3, 10 + g.cg_group, want_counts=False)
    n = 0
    for rep in range(4):                          # 4 x (3 T sweeps + 3 Q+U sweeps) = 24 sweeps, no query in between
        for l, c in enumerate(comps):
            for j in range(c.nindices):
                if c.sample_index[j]:
                    f = c.pol_flag[j][0]
                    eng.index_sample(l, j, MAPN[f], 4, "sample", 3, da.stream_id(5 + rep, 1, l, j, f), want_counts=False)
                    n += 1
    assert n > 8
    cached = eng.chisq_cached(1, 1, 3)
    explicit = eng.sky_model_chisq(1, 3)
    assert cached is not None and abs(cached - explicit) <= 1e-12 * explicit
    # a setter that changes the model drops what is still pending and invalidates the cache
    eng.index_sample(1, 0, 1, 4, "sample", 3, 77, want_counts=False)
    eng.set_tcmb(2.7255)
    assert eng.chisq_cached(1, 1, 3) is None


@pytest.mark.parametrize("which,group,flag", [(("monopole", "hi_fit"), 1, L.FLAG_T), (("template",), 2, L.FLAG_QU)])
def test_schur_solve_over_three_contexts_equals_one_context(built, which, group, flag):
    """A CG group with global-amplitude members on three pixel-shard contexts of one process (dangx_sky_amp_sample: pass 1 on
    every context, Schur rows added in shard order, one small solve, pass 2 everywhere) against the whole-sky context: same
    refinement count, amplitudes to 1e-9; and the residual of the reference's system at the result."""
    def tweak(dpar, ddata, bands, comps):
        add_globals(dpar, ddata, bands, comps, which, group, skip_band0=True)
    whole = make_case("C2", nside=8, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = whole
    ref, _ = pair(whole)
    nsh = 3
    engs = []
    for r in range(nsh):
        pix0, npix = da.dist.shard_range(meta["npix_global"], r, nsh)
        sl = slice(pix0, pix0 + npix)
        cs = copy.deepcopy(comps)
        for c in cs:
            c.amplitude = np.ascontiguousarray(c.amplitude[:, sl])
            if c.indices is not None:
                c.indices = np.ascontiguousarray(c.indices[:, :, sl])
            if c.template is not None:
                c.template = np.ascontiguousarray(c.template[:, sl])
        dd = da.DangData(sig_map=np.ascontiguousarray(ddata.sig_map[:, :, sl]), rms_map=np.ascontiguousarray(ddata.rms_map[:, :, sl]),
                         masks=np.ascontiguousarray(ddata.masks[:, sl]), gain=ddata.gain, offset=ddata.offset, pol_type=ddata.pol_type,
                         nump=ddata.nump)
        engs.append(da.Engine(bands, cs, dd, npix_global=meta["npix_global"], pix0=pix0, device=0))
    for ml_mode in ("optimize", "sample"):
        it1, bad1 = ref.amp_sample(group, flag, ml_mode, 8, 9)
        it3, bad3 = da.sky_amp_sample(engs, group, flag, ml_mode, 8, 9)
        assert (it1, bad1) == (it3, bad3) == (0, 0)
        (r1, _), n1 = ref.schur_info()
        (r3, _), n3 = engs[0].schur_info()
        # the refinement count follows the rounding of the Schur rows (three partial sums instead of one): both runs must end
        # at the same quality, every context of the sky reports the same solve
        assert 0 <= n3 <= 4 and 0 <= n1 <= 4 and engs[2].schur_info() == engs[0].schur_info() and r3 <= max(10 * r1, 1e-10), (r1, n1, r3, n3)
        for l, c in enumerate(comps):
            a = ref.get_amplitude(l)
            b = np.concatenate([e.get_amplitude(l) for e in engs], axis=-1)
            assert np.abs(a - b).max() <= 1e-9 * max(np.abs(a).max(), 1e-30), (ml_mode, l)
            if c.type in which:
                ta, tb = ref.get_template_amplitudes(l), engs[1].get_template_amplitudes(l)
                assert np.abs(ta - tb).max() <= 1e-9 * max(np.abs(ta).max(), 1e-30), (ml_mode, l)
                assert np.array_equal(tb, engs[0].get_template_amplitudes(l))


@pytest.mark.parametrize("lnl,prior", [("chisq", "gaussian"), ("marginal", "uniform"), ("chisq", "jeffreys")])
def test_fullsky_index_mode_at_a_coarser_nside_on_two_shards(built, lnl, prior):
    """index_mode == 1 with sample_nside /= nside over two pixel-shard contexts (dangx_fullsky_sample: the shards' child sums
    are added, every shard holds the degraded maps, each coarse pixel's terms come from the shard that owns full-resolution
    pixel i) against the oracle's whole-sky chain: same accepted count, same value."""
    nside, cnside = 8, 2

    def tweak(dpar, ddata, bands, comps):
        for c in comps:
            c.index_mode = [1] * c.nindices
            c.lnl_type = [lnl] * c.nindices
            c.prior_type = [prior] * c.nindices
            c.tuned = [True] * max(c.nindices, 1)
            c.sample_nside = [cnside] * c.nindices
            c.step_size = [0.6 * g[1] for g in c.gauss_prior]
    whole = make_case("C2", nside=nside, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = whole
    orc = O.Oracle(bands, copy.deepcopy(comps), ddata)
    shards = [make_case("C2", nside=nside, start="truth", tweak=tweak, rank=r, nranks=2) for r in range(2)]
    engs = [da.Engine(x[2], x[3], x[1], npix_global=x[4]["npix_global"], pix0=x[4]["pix0"], device=0) for x in shards]
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if not c.sample_index[j]:
                continue
            f = c.pol_flag[j][0]
            s = da.stream_id(2, 1, l, j, f)
            ao, _, _ = orc.sample_index_fullsky_coarse(l, j, MAPN[f], dpar.nsample, "sample", dpar.seed, s, nside, cnside)
            ag = da.sample_index_mh_fullsky(shards[0][0], shards[0][1], l, j, MAPN[f], s, sample_nside=cnside, engines=engs)
            assert ag == ao, (l, j, ag, ao)
            got = np.concatenate([e.get_indices(l) for e in engs], axis=-1)
            assert np.abs(got - orc.indices(l)).max() <= 1e-12


def test_c1_at_its_own_nside_matches_the_oracle(built):
    """BASELINE config 1 as it is quoted (Nside 64, 3 bands, synch + dust, one Stokes plane): three Gibbs iterations."""
    case = make_case("C1", nside=64)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for it in (1, 2, 3):
        if it == 1:
            da.sample_cg_groups(dpar, ddata, it=it)
            for g in dpar.cg_groups:
                orc.amp_sample_direct(g.cg_group, g.pol_flag[0], "sample", dpar.seed, da.stream_id(it, 0, g.cg_group, 0, g.pol_flag[0]), "reference")
        else:
            da.gibbs_iteration(dpar, ddata, it)
            _oracle_iteration(orc, dpar, comps, it)
    for l, c in enumerate(comps):
        b = orc.amplitude(l)
        assert np.abs(eng.get_amplitude(l) - b).max() <= 1e-9 * np.abs(b).max(), l
        assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-12, l
    ochisq, _ = orc.chisq(1, 1, ddata.nump)
    assert abs(ddata.chisq - ochisq) <= 1e-10 * ochisq


def test_chisq_current_and_the_batched_index_means(built):
    """dangx_chisq_current = the explicit pass (dangx_sky_model_chisq) whichever kernel wrote the planes last -- a sweep (cached
    sums), a solve (one pass over that plane only), a plane set, an upload; dangx_index_masked_sums = the single-map reductions,
    entry by entry."""
    case = make_case("C2", nside=8, start="truth")
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)

    def check(tag):
        for lo, hi in ((1, 1), (2, 3), (1, 3)):
            a = eng.chisq_current(lo, hi)
            b = eng.sky_model_chisq(lo, hi)
            assert abs(a - b) <= 1e-12 * abs(b), (tag, lo, hi, a, b)
            assert eng.chisq_current(lo, hi) == a, (tag, lo, hi)     # now from the cache: the same number

    check("upload")
    eng.amp_sample(1, L.FLAG_T, "sample", dpar.seed, da.stream_id(1, 0, 1, 0, 1))
    check("solve T")
    eng.index_sample(1, 0, 1, dpar.nsample, "sample", dpar.seed, da.stream_id(1, 1, 1, 0, 1))
    check("sweep T")
    eng.amp_sample(2, L.FLAG_QU, "sample", dpar.seed, da.stream_id(1, 0, 2, 0, 8))
    check("solve QU")
    da.gibbs_iteration(dpar, ddata, 2)
    check("plane sets")
    eng.put_indices(1, eng.get_indices(1) * 1.01)
    check("indices replaced")

    entries = [(l, j, m) for l, c in enumerate(comps) for j in range(c.nindices) for m in (1, 2, 3)][:16]
    assert len(entries) == 16
    sums, counts = eng.index_masked_sums(entries)
    for e, s, n in zip(entries, sums, counts):
        s1, n1 = eng.index_masked_sum(*e)
        assert n == n1 and abs(s - s1) <= 1e-13 * max(abs(s1), 1.0), (e, s, s1)
    with pytest.raises(Exception):
        eng.index_masked_sums(entries[:1] * 17)


@pytest.mark.parametrize("low", [0.05, 12.0])
def test_temperature_sweep_below_and_above_the_batched_reciprocal_limit(built, low):
    """The dust-temperature chain shares one reciprocal among a tile's Planck denominators where no proposal can make their
    product overflow (dx_chain.h, chain_finish); a hard lower bound of 0.05 K (h nu / k T = 820 at 857 GHz) switches the
    wavefront to the one-by-one form.  Both must be the oracle's chain."""
    def tweak(dpar, ddata, bands, comps):
        for c in comps:
            if c.type == "mbb":
                c.uni_prior = [list(u) for u in c.uni_prior]
                c.uni_prior[1][0] = low
    case = make_case("C2", nside=8, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for it in (1, 2):
        da.gibbs_iteration(dpar, ddata, it)
        _oracle_iteration(orc, dpar, comps, it)
    for l, c in enumerate(comps):
        b = orc.amplitude(l)
        assert np.abs(eng.get_amplitude(l) - b).max() <= 1e-9 * max(np.abs(b).max(), 1.0), l
        if c.nindices:
            assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-12, l


@pytest.mark.parametrize("config,nside,nbands", [("C3", 8, None), ("C5", 4, None), ("C2", 8, 7), ("C1", 8, None)])
def test_chisq_on_the_amplitude_schedule_is_the_run_time_typed_pass(built, config, nside, nbands):
    """dangx_sky_model_chisq without map outputs runs k_chisq_reg (one launch per plane, the plane's components with a non-zero
    amplitude as the group); asking for the maps runs the run-time-typed k_sky_chisq.  The same chi^2 (different SED
    reciprocals: 1e-12), the oracle's chi^2, also with a gain and an offset on the temperature plane, a component switched
    off on one plane, and a spatially constant index plane."""
    def tweak(dpar, ddata, bands, comps):
        ddata.gain = np.asarray(ddata.gain, dtype=np.float64).copy()
        ddata.offset = np.asarray(ddata.offset, dtype=np.float64).copy()
        ddata.gain[1] = 1.02
        ddata.offset[0] = 0.3
        comps[1].indices[0, 0, :] = comps[1].indices[0, 0, 0]      # synchrotron beta constant on the T plane
    kw = dict(nbands=nbands) if nbands else {}
    case = make_case(config, nside=nside, start="truth", tweak=tweak, **kw)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    nm = meta["nmaps"]
    for lo, hi in ((1, 1), (1, nm)) + (((2, 3),) if nm == 3 else ()):
        fast = eng.sky_model_chisq(lo, hi)
        slow = eng.sky_model_chisq(lo, hi, want_maps=True)[0]
        want, _ = orc.chisq(lo, hi, 1.0)
        assert abs(fast - slow) <= 1e-12 * abs(slow), (lo, hi, fast, slow)
        assert abs(fast / meta["nbands"] - want) <= 1e-11 * abs(want), (lo, hi, fast / meta["nbands"], want)
    # a component with no amplitude on a plane drops out of that plane's launch
    l = 1
    a = eng.get_amplitude(l)
    a[0] = 0.0
    eng.put_amplitude(l, a)
    fast = eng.sky_model_chisq(1, 1)
    slow = eng.sky_model_chisq(1, 1, want_maps=True)[0]
    assert abs(fast - slow) <= 1e-12 * abs(slow)
