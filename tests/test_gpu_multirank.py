"""Two ranks (gloo, both on cuda:0) against one rank: the places where a pixel-sharded run needs a sum over the
whole sky INSIDE an amplitude solve -- the dot products of the device CG and the global-amplitude rows of template
groups (SURVEY 8e collectives (3)) -- go through the dangx_set_allreduce callback."""
import copy
import os
import socket

import numpy as np
import pytest
import torch.distributed as td
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

JOBS = [  # (global members, group, flag, solver, ml_mode)
    (("template",), 2, 8, "direct", "sample"),
    (("monopole", "hi_fit"), 1, 1, "direct", "optimize"),
    (("template",), 2, 8, "cg", "sample"),
    ((), 1, 1, "cg", "sample"),
    # the solve together with the sweeps on its planes (dangx_plane_set_sample): the Schur rows through the callback, then ONE launch
    # per rank that back-substitutes and sweeps; the swept index maps are gathered and compared too
    (("template",), 2, 8, "planeset", "sample"),
]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _full_case(which, group, solver=""):
    from test_oracle_templates_cpu import add_globals
    from util import make_case

    def tweak(dpar, ddata, bands, comps):
        if which and solver == "planeset":   # a well-posed fit (two of five bands): the solve that needs no residual pass
            add_globals(dpar, ddata, bands, comps, which, group, fit_bands=[3, 4])
        elif which:
            add_globals(dpar, ddata, bands, comps, which, group, skip_band0=True)
    return make_case("C2", nside=4, start="truth", tweak=tweak)


def _shard(case, rank, world):
    from dang_amd import dist
    dpar, ddata, bands, comps, meta = case
    p0, n = dist.shard_range(meta["npix_global"], rank, world)
    ddata = copy.copy(ddata)
    ddata.sig_map = np.ascontiguousarray(ddata.sig_map[..., p0:p0 + n])
    ddata.rms_map = np.ascontiguousarray(ddata.rms_map[..., p0:p0 + n])
    ddata.masks = np.ascontiguousarray(ddata.masks[..., p0:p0 + n])
    comps = copy.deepcopy(comps)
    for c in comps:
        c.amplitude = np.ascontiguousarray(c.amplitude[..., p0:p0 + n])
        if c.nindices:
            c.indices = np.ascontiguousarray(c.indices[..., p0:p0 + n])
        if getattr(c, "template", None) is not None:
            c.template = np.ascontiguousarray(c.template[..., p0:p0 + n])
    return ddata, comps, p0, n


def _run(case, which, group, flag, solver, ml_mode, rank=0, world=1):
    import dang_amd as da
    dpar, ddata, bands, comps, meta = case
    ddata, comps, p0, n = _shard(case, rank, world)
    eng = da.Engine(bands, comps, ddata, npix_global=meta["npix_global"], pix0=p0, device=0)
    # few CG iterations: the check is on the mechanics (every sum over the sky complete), before the rounding-level
    # differences of the summation order are amplified along an unconverged trajectory
    if solver == "planeset":
        sweeps = [(l, j, da.stream_id(2, 1, l, j, flag)) for l, c in enumerate(comps) for j in range(c.nindices)
                  if c.cg_group == group and c.sample_index[j] and flag in c.pol_flag[j]]
        eng.profile(True)
        bad, _ = eng.plane_set_sample(group, flag, ml_mode, 8, 9, sweeps, 5, 11)
        prof = eng.profile_get()
        assert bad == 0 and prof["k_amp_index"]["launches"] == 1 and "k_index_mh" not in prof and "k_amp_direct" not in prof, prof
        it = 0
    else:
        it, _ = eng.amp_sample(group, flag, ml_mode, 8, 9, solver=solver, i_max=6, converge=1e-10)
    amps = [eng.get_amplitude(l) for l in range(len(comps))]
    if solver == "planeset":   # (index maps as further "amplitude" arrays of the comparison: [nmaps, npix] per index)
        amps += [eng.get_indices(l)[j] for l, c in enumerate(comps) for j in range(c.nindices) if c.cg_group == group]
    tas = [eng.get_template_amplitudes(l) for l, c in enumerate(comps) if c.type in ("template", "monopole", "hi_fit")]
    return it, amps, tas


def _fullsky_job(rank=0, world=1):
    """Full-sky index chain, band-gain fit and the write_data means on a sharded sky: every sky-wide sum is all-reduced
    by the host code (dang_amd/api.py), so each rank walks the same chain."""
    import dang_amd as da
    from util import make_case

    def tweak(dpar, ddata, bands, comps):
        for c in comps:
            c.index_mode = [1] * c.nindices
            c.step_size = [0.2 * g[1] for g in c.gauss_prior]
    case = make_case("C2", nside=4, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    ddata, comps, p0, n = _shard(case, rank, world)
    ddata.gain = np.ones(meta["nbands"]); ddata.offset = np.zeros(meta["nbands"])
    ddata.nump = case[1].nump
    eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], pix0=p0, device=0)
    acc = da.sample_index_mh_fullsky(dpar, ddata, 1, 0, 1, da.stream_id(2, 1, 1, 0, 1))
    beta = eng.peek_indices(1, 1, 0)[0]
    gain = da.fit_band_gain(dpar, ddata, 2, it=3)
    means = da.index_means(ddata, 1)
    return np.array([acc, beta, gain] + [means[k] for k in sorted(means)])


def _coarse_job(rank=0, world=1):
    """Coarse-Nside (8 -> 2) per-pixel sweeps on a sharded sky: the children of a coarse pixel are scattered over the
    RING ranges, so the degrade step and the coarse index map are summed over the ranks (dangx_set_allreduce)."""
    import dang_amd as da
    from util import make_case

    def tweak(dpar, ddata, bands, comps):
        for c in comps:
            c.sample_nside = [2] * c.nindices
    case = make_case("C2", nside=8, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    ddata, comps, p0, n = _shard(case, rank, world)
    ddata.nump = case[1].nump
    eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], pix0=p0, device=0)
    info = da.sample_spectral_parameters(dpar, ddata, it=2)
    out = {"acc": np.array([a for (_, _, _, a) in info], dtype=np.float64), "chisq": np.array([ddata.chisq])}
    for l, c in enumerate(comps):
        if c.nindices:
            out["idx%d" % l] = eng.get_indices(l)
    return out


def _worker(rank, world, port, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from dang_amd import dist
    import torch
    res = {}
    for n, (which, group, flag, solver, ml_mode) in enumerate(JOBS):
        it, amps, tas = _run(_full_case(which, group, solver), which, group, flag, solver, ml_mode, rank, world)
        res["it%d" % n] = it
        for l, a in enumerate(amps):
            g = dist.gather_maps(torch.from_numpy(a), 192, dst=0)
            if rank == 0:
                res["amp%d_%d" % (n, l)] = g.numpy()
        for l, t in enumerate(tas):
            res["ta%d_%d_r%d" % (n, l, rank)] = t
    res["fullsky_r%d" % rank] = _fullsky_job(rank, world)
    cj = _coarse_job(rank, world)
    res["coarse_acc_r%d" % rank] = cj["acc"]
    res["coarse_chisq_r%d" % rank] = cj["chisq"]
    for k, v in cj.items():
        if k.startswith("idx"):
            g = dist.gather_maps(torch.from_numpy(v), 768, dst=0)
            if rank == 0:
                res["coarse_" + k] = g.numpy()
    gathered = [None] * world
    td.all_gather_object(gathered, {k: v for k, v in res.items() if k.startswith("ta") or k.startswith("fullsky") or
                                    k.startswith("coarse_acc") or k.startswith("coarse_chisq")})
    if rank == 0:
        for g in gathered:
            res.update(g)
        np.savez(out, **res)
    td.barrier()
    td.destroy_process_group()


def test_two_ranks_solve_the_same_coupled_systems_as_one(built, tmp_path):
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    for n, (which, group, flag, solver, ml_mode) in enumerate(JOBS):
        it, amps, tas = _run(_full_case(which, group, solver), which, group, flag, solver, ml_mode)
        assert int(got["it%d" % n]) == it, (n, int(got["it%d" % n]), it)
        # sums over the sky are formed in a different order (per-rank partials, then the all-reduce)
        tol = 1e-9
        for l, a in enumerate(amps):
            assert np.abs(got["amp%d_%d" % (n, l)] - a).max() <= tol * max(np.abs(a).max(), 1e-300), (n, l)
        for l, t in enumerate(tas):
            for r in range(2):  # replicated on every rank
                assert np.abs(got["ta%d_%d_r%d" % (n, l, r)] - t).max() <= tol * max(np.abs(t).max(), 1e-300), (n, l, r)
            assert np.array_equal(got["ta%d_%d_r0" % (n, l)], got["ta%d_%d_r1" % (n, l)])
    one = _coarse_job()
    assert np.array_equal(got["coarse_acc_r0"], got["coarse_acc_r1"]) and np.array_equal(got["coarse_acc_r0"], one["acc"])
    for k, v in one.items():
        if k.startswith("idx"):   # (the child sums are associated per shard first: the degraded maps agree to rounding)
            assert np.abs(got["coarse_" + k] - v).max() <= 1e-12, k
    assert abs(got["coarse_chisq_r0"][0] - one["chisq"][0]) <= 1e-10 * one["chisq"][0]
    assert got["coarse_chisq_r0"][0] == got["coarse_chisq_r1"][0]
    one = _fullsky_job()
    assert np.array_equal(got["fullsky_r0"], got["fullsky_r1"])          # every rank walks the same chain
    assert got["fullsky_r0"][0] == one[0]                                # same accept count
    assert np.abs(got["fullsky_r0"][1:] - one[1:]).max() <= 1e-10 * np.abs(one[1:]).max()
