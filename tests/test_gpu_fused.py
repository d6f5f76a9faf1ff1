"""dangx_amp_index_sample (the amplitude solve of a CG group and the first index sweep on the same planes in one kernel
launch, dang_amd/csrc/dangx_fused.hip) against the two calls it stands for: bit for bit, in every state a Gibbs run
passes through, and against the oracle like every other path."""
import copy

import numpy as np
import pytest

import dang_amd as da
from dang_amd import _lib as L

from util import MAPN, TOL_AMP, TOL_CHISQ, assert_amps_close, assert_indices_close, make_case, pair

pytestmark = pytest.mark.gpu


def _engines(case):
    dpar, ddata, bands, comps, meta = case
    a = da.Engine(bands, copy.deepcopy(comps), ddata, npix_global=meta["npix_global"], pix0=meta["pix0"], device=0)
    b = da.Engine(bands, copy.deepcopy(comps), ddata, npix_global=meta["npix_global"], pix0=meta["pix0"], device=0)
    return a, b


def _first_sweep(comps, group):
    for l, c in enumerate(comps):
        if c.cg_group == group:
            for j in range(c.nindices):
                if c.sample_index[j]:
                    return l, j
    raise AssertionError("no sampled index in group %d" % group)


@pytest.mark.parametrize("ml_mode", ["sample", "optimize"])
@pytest.mark.parametrize("config,start", [("C3", "truth"), ("C3", "prior"), ("C2", "prior"), ("C1", "prior")])
def test_fused_equals_the_two_calls_bitwise(built, ml_mode, config, start):
    """The BASELINE models (C3: 10 bands, cmb + synch + dust + ff, IQU; C2: 5 bands, 3 components, IQU; C1: 3 bands, 2
    components, I) at Nside 8: three Gibbs iterations, the fused entry point on one context and amp_sample + index_sample
    on another; amplitudes, indices, chi^2 sums and both counters are equal bit for bit after every iteration.
    start='prior' begins with spatially constant index maps, i.e. passes through the states in which the fused launch is
    not taken (first sweeps) and those in which it is."""
    case = make_case(config, nside=8, start=start)
    dpar, ddata, bands, comps, meta = case
    fus, two = _engines(case)
    prof = []
    for it in range(1, 4):
        for eng in (fus, two):
            eng.profile(True)
            for g in dpar.cg_groups:
                f = g.pol_flag[0]
                l0, j0 = _first_sweep(comps, g.cg_group)
                sa, si = da.stream_id(it, 0, g.cg_group, 0, f), da.stream_id(it, 1, l0, j0, f)
                if eng is fus:
                    bad, acc = eng.amp_index_sample(g.cg_group, f, ml_mode, 11, sa, l0, j0, MAPN[f], 10, 11, si)
                else:
                    _, bad = eng.amp_sample(g.cg_group, f, ml_mode, 11, sa)
                    acc = eng.index_sample(l0, j0, MAPN[f], 10, ml_mode, 11, si)
                prof.append((eng is fus, bad, acc))
                for l, c in enumerate(comps):
                    for j in range(c.nindices):
                        if c.sample_index[j] and c.cg_group == g.cg_group and (l, j) != (l0, j0):
                            eng.index_sample(l, j, MAPN[f], 10, ml_mode, 11, da.stream_id(it, 1, l, j, f))
            eng.synchronize()
        names = fus.profile_get()
        if it > 1:   # from the second iteration on every index map varies: the fused kernel is the one that ran
            assert "k_amp_index" in names and "k_amp_direct" not in names, names
        assert "k_amp_index" not in two.profile_get()
        for l, c in enumerate(comps):
            assert np.array_equal(fus.get_amplitude(l), two.get_amplitude(l)), (it, l)
            if c.nindices:
                assert np.array_equal(fus.get_indices(l), two.get_indices(l)), (it, l)
        for which in (0, 1):
            assert fus.chisq_cached(which, 1, meta["nmaps"]) == two.chisq_cached(which, 1, meta["nmaps"]), (it, which)
    counts_f = [p[1:] for p in prof if p[0]]
    counts_t = [p[1:] for p in prof if not p[0]]
    assert counts_f == counts_t and any(a > 0 for _, a in counts_f)


def test_fused_against_the_oracle(built):
    case = make_case("C3", nside=4, start="truth")
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for g in dpar.cg_groups:
        f = g.pol_flag[0]
        l0, j0 = _first_sweep(comps, g.cg_group)
        bad, acc = eng.amp_index_sample(g.cg_group, f, "sample", 5, 21 + f, l0, j0, MAPN[f], 10, 5, 41 + f)
        orc.amp_sample_direct(g.cg_group, f, "sample", 5, 21 + f, "reference")
        oacc = orc.sample_index_mh(l0, j0, MAPN[f], 10, "sample", 5, 41 + f)
        assert bad == 0 and acc == oacc
    assert_amps_close(eng, orc, len(comps), TOL_AMP)
    assert_indices_close(eng, orc, comps)
    s, _ = orc.chisq(1, 3, ddata.nump)
    assert abs(eng.chisq_cached(1, 1, 3) / meta["nbands"] / ddata.nump - s) <= TOL_CHISQ * s


def test_configurations_the_fused_kernel_does_not_cover_take_the_two_calls(built):
    """Unequal Q/U plane indices, a masked sky fraction, another band count, a Jeffreys prior, the textbook fluctuation
    term: the entry point still returns what the two calls return (it IS the two calls there), bit for bit."""
    def run(case, **kw):
        dpar, ddata, bands, comps, meta = case
        fus, two = _engines(case)
        for g in dpar.cg_groups:
            f = g.pol_flag[0]
            l0, j0 = _first_sweep(comps, g.cg_group)
            fus.amp_index_sample(g.cg_group, f, "sample", 3, 5, l0, j0, MAPN[f], 6, 3, 7, **kw)
            two.amp_sample(g.cg_group, f, "sample", 3, 5, **kw)
            two.index_sample(l0, j0, MAPN[f], 6, "sample", 3, 7)
        for l, c in enumerate(comps):
            assert np.array_equal(fus.get_amplitude(l), two.get_amplitude(l)), l
            if c.nindices:
                assert np.array_equal(fus.get_indices(l), two.get_indices(l)), l

    def unequal(dpar, ddata, bands, comps):
        rng = np.random.default_rng(2)
        for c in comps:
            if c.nindices and c.cg_group == 2:
                c.indices[0, 2] *= 1.0 + 0.02 * rng.standard_normal(c.indices.shape[-1])

    def jeffreys(dpar, ddata, bands, comps):
        for c in comps:
            c.prior_type = ["jeffreys"] * c.nindices

    run(make_case("C3", nside=4, start="truth", tweak=unequal))            # fused, per-plane SED columns
    run(make_case("C2", nside=4, start="truth"))                            # 5 bands: not instantiated
    run(make_case("C3", nside=4, start="truth", tweak=jeffreys))            # LDS-form chain
    run(make_case("C3", nside=4, start="truth"), fluct_mode="correct")      # k_amp_direct


def test_a_failing_sweep_still_leaves_the_solve_done(built):
    """Error behaviour of the pair: when the index call is rejected (index number out of range) the amplitude solve has
    happened all the same, exactly as with two separate calls."""
    case = make_case("C3", nside=4, start="truth")
    dpar, ddata, bands, comps, meta = case
    fus, two = _engines(case)
    for eng in (fus, two):   # make the index maps "varying" so that the fused route is the one that is tried
        eng.index_sample(1, 0, 1, 2, "sample", 1, 1)
    with pytest.raises(da.DangxError):
        fus.amp_index_sample(1, L.FLAG_T, "sample", 3, 5, 1, 7, 1, 6, 3, 7)      # synch has no index 7
    two.amp_sample(1, L.FLAG_T, "sample", 3, 5)
    for l in range(len(comps)):
        assert np.array_equal(fus.get_amplitude(l), two.get_amplitude(l)), l


@pytest.mark.parametrize("config,nside", [("C3", 8), ("C2", 8), ("C5", 4)])
@pytest.mark.parametrize("ml_mode", ["sample", "optimize"])
def test_index_sample_pair_equals_the_two_sweeps_bitwise(built, config, nside, ml_mode):
    """dangx_index_sample_pair (two consecutive indices of one component in one launch) against the two sweeps: index maps,
    chi^2 sums and both accepted counts bit for bit, for every two-index component of the configuration, on the T plane
    and on Q+U (C5: 20 bands -- the T plane takes the pair kernel, Q+U the two lane-pair sweeps; its log-normal
    component has one sampled index only and goes the two-call way too)."""
    case = make_case(config, nside=nside, start="truth")
    dpar, ddata, bands, comps, meta = case
    one, two = _engines(case)
    ran = 0
    for it in (1, 2):
        for l, c in enumerate(comps):
            if c.nindices != 2:
                continue
            f = c.pol_flag[0][0]
            s1, s2 = da.stream_id(it, 1, l, 0, f), da.stream_id(it, 1, l, 1, f)
            a = one.index_sample_pair(l, 0, MAPN[f], 8, ml_mode, 17, s1, s2)
            b = (two.index_sample(l, 0, MAPN[f], 8, ml_mode, 17, s1), two.index_sample(l, 1, MAPN[f], 8, ml_mode, 17, s2))
            assert a == b, (l, a, b)
            assert np.array_equal(one.get_indices(l), two.get_indices(l)), l
            k = (1, 1) if f == L.FLAG_T else (2, 3)
            for which in (0, 1):
                assert one.chisq_cached(which, *k) == two.chisq_cached(which, *k), (l, which)
            ran += 1
    assert ran >= 4


@pytest.mark.parametrize("config,nside", [("C3", 8), ("C2", 8)])
def test_python_mirror_gibbs_iteration_equals_the_two_phases(built, config, nside):
    """da.gibbs_iteration (solves issued with the first sweeps, consecutive indices paired) against da.sample_cg_groups +
    a sample_spectral_parameters that makes every sweep on its own: the same state, bit for bit, and the same chi^2 to rounding
    after each of three iterations."""
    case_a = make_case(config, nside=nside)
    case_b = make_case(config, nside=nside)
    for case in (case_a, case_b):
        da.initialize(case[2], case[3], case[1], npix_global=case[4]["npix_global"], device=0)
    (dpa, dda), (dpb, ddb) = (case_a[0], case_a[1]), (case_b[0], case_b[1])
    comps = case_b[3]
    for it in range(1, 4):
        da.gibbs_iteration(dpa, dda, it)
        da.sample_cg_groups(dpb, ddb, it=it, defer_chisq=True)
        eng = ddb.engine
        for l, c in enumerate(comps):                       # the index phase, one sweep per call
            for j in range(c.nindices):
                if c.sample_index[j]:
                    for f in c.pol_flag[j]:
                        eng.index_sample(l, j, MAPN[f], dpb.nsample, dpb.ml_mode, dpb.seed, da.stream_id(it, 1, l, j, f))
        for l, c in enumerate(comps):
            assert np.array_equal(dda.engine.get_amplitude(l), eng.get_amplitude(l)), (it, l)
            if c.nindices:
                assert np.array_equal(dda.engine.get_indices(l), eng.get_indices(l)), (it, l)
        # (the plane-set launch keeps the residual between its sweeps instead of re-staging the maps: same proposals, same accept
        # decisions, same maps -- the chi^2 sums agree to the rounding of the residuals)
        assert abs(dda.engine.chisq_cached(1, 1, 3) - eng.chisq_cached(1, 1, 3)) <= 1e-12 * eng.chisq_cached(1, 1, 3)
        assert dda.chisq == dda.engine.chisq_cached(1, 1, 3) / case_a[4]["nbands"] / dda.nump


def test_fused_entry_points_are_shard_invariant(built):
    """Three pixel shards driven through da.gibbs_iteration reproduce the whole-sky run bit for bit (the fused kernels key
    the random streams by the global pixel like every other kernel), and their chi^2 sums add up."""
    from dang_amd import synth
    full = synth.make_sky("C3", nside=8)
    parts = [synth.make_sky("C3", nside=8, rank=r, nranks=3) for r in range(3)]
    for x in [full] + parts:
        da.initialize(x[2], x[3], x[1], npix_global=x[4]["npix_global"], pix0=x[4]["pix0"], device=0)
        for it in range(1, 4):
            da.gibbs_iteration(x[0], x[1], it)
    for l, c in enumerate(full[3]):
        cat = np.concatenate([p[1].engine.get_amplitude(l) for p in parts], axis=-1)
        assert np.array_equal(cat, full[1].engine.get_amplitude(l)), l
        if c.nindices:
            assert np.array_equal(np.concatenate([p[1].engine.get_indices(l) for p in parts], axis=-1), full[1].engine.get_indices(l)), l
    tot = full[1].engine.chisq_cached(1, 1, 3)
    assert abs(tot - sum(p[1].engine.chisq_cached(1, 1, 3) for p in parts)) <= 1e-12 * tot


def test_c5_fused_lane_pairs_against_the_oracle(built):
    """C5 (20 bands, 6 members, a log-normal member that varies over the sky): on Q+U the group's solve and the synchrotron sweep
    run as ONE launch in the lane-pair form (k_amp_index<..., 2, 20, 6, 2>); on the T plane, where the chain runs one lane per
    pixel, the two launches are the faster form and are what runs -- against the oracle's solve followed by its sweep, and
    against the two launches to the parity tolerance."""
    case = make_case("C5", nside=4, start="truth")
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    two = da.Engine(bands, copy.deepcopy(comps), ddata, npix_global=meta["npix_global"], pix0=meta["pix0"], device=0)
    for g in dpar.cg_groups:
        f = g.pol_flag[0]
        l0, j0 = _first_sweep(comps, g.cg_group)
        eng.profile(True)
        bad, acc = eng.amp_index_sample(g.cg_group, f, "sample", 5, 21 + f, l0, j0, MAPN[f], 10, 5, 41 + f)
        prof = eng.profile_get()
        if f == L.FLAG_QU:
            assert prof["k_amp_index"]["launches"] == 1 and "k_amp_direct" not in prof and "k_index_mh" not in prof, prof
        else:
            assert "k_amp_index" not in prof and prof["k_amp_direct"]["launches"] == 1 and prof["k_index_mh"]["launches"] == 1, prof
        two.amp_sample(g.cg_group, f, "sample", 5, 21 + f)
        acc2 = two.index_sample(l0, j0, MAPN[f], 10, "sample", 5, 41 + f)
        orc.amp_sample_direct(g.cg_group, f, "sample", 5, 21 + f, "reference")
        oacc = orc.sample_index_mh(l0, j0, MAPN[f], 10, "sample", 5, 41 + f)
        assert bad == 0 and acc == oacc == acc2
    assert_amps_close(eng, orc, len(comps), TOL_AMP)
    assert_indices_close(eng, orc, comps)
    for l, c in enumerate(comps):
        a, b = eng.get_amplitude(l), two.get_amplitude(l)
        assert np.abs(a - b).max() <= 1e-11 * max(np.abs(b).max(), 1.0), l
        if c.nindices:
            assert np.abs(eng.get_indices(l) - two.get_indices(l)).max() <= 1e-12, l


def _plane_sweeps(comps, group, f):
    return [(l, j) for l, c in enumerate(comps) if c.cg_group == group for j in range(c.nindices) if c.sample_index[j] and f in c.pol_flag[j]]


@pytest.mark.parametrize("nbands", [20, 16])
def test_plane_set_launch_against_the_oracle(built, nbands):
    """dangx_plane_set_sample on the C5 model (6 members; 20 bands: the built-in kernel, 16 bands: specialised at run time): a
    group's solve and EVERY sweep on its planes -- synchrotron beta, dust beta + T, AME nu_p -- in ONE launch per plane set, the
    members' SED columns kept in LDS across the sweeps.  Against the oracle's solve followed by its sweeps in the same order
    (amplitudes, indices, accepted counts per sweep), and against the separate launches to the parity tolerance."""
    case = make_case("C5", nside=4, nbands=nbands, start="truth")
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    two = da.Engine(bands, copy.deepcopy(comps), ddata, npix_global=meta["npix_global"], pix0=meta["pix0"], device=0)
    for it in (1, 2):
        for g in dpar.cg_groups:
            f = g.pol_flag[0]
            sw = _plane_sweeps(comps, g.cg_group, f)
            assert len(sw) == 4
            sa = da.stream_id(it, 0, g.cg_group, 0, f)
            eng.profile(True)
            bad, accs = eng.plane_set_sample(g.cg_group, f, "sample", 5, sa, [(l, j, da.stream_id(it, 1, l, j, f)) for l, j in sw], 10, 5)
            prof = eng.profile_get()
            assert prof["k_amp_index"]["launches"] == 1 and "k_amp_direct" not in prof and "k_index_mh" not in prof, prof
            two.amp_sample(g.cg_group, f, "sample", 5, sa)
            orc.amp_sample_direct(g.cg_group, f, "sample", 5, sa, "reference")
            for q, (l, j) in enumerate(sw):
                s = da.stream_id(it, 1, l, j, f)
                oacc = orc.sample_index_mh(l, j, MAPN[f], 10, "sample", 5, s)
                tacc = two.index_sample(l, j, MAPN[f], 10, "sample", 5, s)
                assert accs[q] == oacc == tacc, (it, l, j, accs[q], oacc, tacc)
            assert bad == 0
        assert_amps_close(eng, orc, len(comps), TOL_AMP)
        assert_indices_close(eng, orc, comps)
        for l, c in enumerate(comps):
            a, b = eng.get_amplitude(l), two.get_amplitude(l)
            assert np.abs(a - b).max() <= 1e-11 * max(np.abs(b).max(), 1.0), l
            if c.nindices:
                assert np.abs(eng.get_indices(l) - two.get_indices(l)).max() <= 1e-12, l
        # the chi^2 by-product: before = the state the solve left, after = the state the last sweep left
        o_after, _ = orc.chisq(1, 3, ddata.nump)
        assert abs(eng.chisq_cached(1, 1, 3) / meta["nbands"] / ddata.nump - o_after) <= TOL_CHISQ * o_after
    if nbands != 20:
        assert any(n.startswith("dxk::k_plane_set<") for n in eng.rtc_kernels()), eng.rtc_kernels()


def test_plane_set_entry_is_the_separate_calls_where_the_kernel_does_not_apply(built):
    """C3 (10 bands): dangx_plane_set_sample against dangx_amp_sample + dangx_index_sample per sweep: the same maps, bit for bit
    (same proposals, same accept decisions), the same accepted counts, chi^2 by-products equal to rounding"""
    case = make_case("C3", nside=8, start="truth")
    dpar, ddata, bands, comps, meta = case
    one, two = _engines(case)
    for it in (1, 2):
        for g in dpar.cg_groups:
            f = g.pol_flag[0]
            sw = _plane_sweeps(comps, g.cg_group, f)
            sa = da.stream_id(it, 0, g.cg_group, 0, f)
            bad, accs = one.plane_set_sample(g.cg_group, f, "sample", 5, sa, [(l, j, da.stream_id(it, 1, l, j, f)) for l, j in sw], 10, 5)
            two.amp_sample(g.cg_group, f, "sample", 5, sa)
            for q, (l, j) in enumerate(sw):
                assert accs[q] == two.index_sample(l, j, MAPN[f], 10, "sample", 5, da.stream_id(it, 1, l, j, f))
        for l, c in enumerate(comps):
            assert np.array_equal(one.get_amplitude(l), two.get_amplitude(l)), l
            if c.nindices:
                assert np.array_equal(one.get_indices(l), two.get_indices(l)), l
        for which in (0, 1):   # maps bit for bit; the sums from the resident residual agree with the re-staged ones to rounding
            assert abs(one.chisq_cached(which, 1, 3) - two.chisq_cached(which, 1, 3)) <= 1e-12 * two.chisq_cached(which, 1, 3)


def test_c5_gibbs_iteration_matches_the_oracle_loop(built):
    """da.gibbs_iteration on C5 (plane-set launches) against the oracle's loop in the reference's order: all solves, then all sweeps"""
    case = make_case("C5", nside=4, start="truth")
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for it in (2, 3):
        da.gibbs_iteration(dpar, ddata, it)
        for g in dpar.cg_groups:
            orc.amp_sample_direct(g.cg_group, g.pol_flag[0], "sample", dpar.seed, da.stream_id(it, 0, g.cg_group, 0, g.pol_flag[0]), "reference")
        for l, c in enumerate(comps):
            for j in range(c.nindices):
                if c.sample_index[j]:
                    f = c.pol_flag[j][0]
                    orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, "sample", dpar.seed, da.stream_id(it, 1, l, j, f))
    assert_amps_close(eng, orc, len(comps), TOL_AMP)
    assert_indices_close(eng, orc, comps)
    ochisq, _ = orc.chisq(1, 3, ddata.nump)
    assert abs(ddata.chisq - ochisq) <= 1e-9 * ochisq
