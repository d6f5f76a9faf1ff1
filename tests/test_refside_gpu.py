"""The reference-side wrapper (fortran/reference_side/dang_gpu_mod.f90: sample_cg_groups_gpu, sample_spectral_parameters_gpu,
gibbs_iteration_gpu, sample_calibrators_gpu, write_data_gpu, dangx_refresh_host_state) RUN on the GPU: compiled by flang against
the mock modules of stubs/stubs.f90 and driven by dang_gpu_drive.f90, which plays `program dang` (src/dang.f90:79-126).

Checked against the ORACLE running the reference's loop in the reference's order, in the two-call form and in the fused
form (bit-identical to each other), on one and on several contexts; the ASCII traces of write_data_gpu against the
reference's edit descriptors (src/dang_data_mod.f90:687-759) applied to the oracle's numbers; and the rows the smoke driver
never reached: a template + monopole group over three contexts (Schur rows shared), a full-sky index with the tuner, a
coarse-Nside sweep and a band-gain fit."""
import copy
import os

import numpy as np
import pytest

import dang_amd as da
from dang_amd import _build, fdrive, synth
from dang_amd import _lib as L

import oracle_ffi as O
from test_oracle_templates_cpu import add_globals
from util import MAPN, make_case

pytestmark = pytest.mark.gpu

NITER = 3


def _need_drive():
    if _build.build_reference_drive() is None:
        pytest.skip("flang not available")


def _oracle_loop(case, niter, fit_gain=None):
    """program dang's loop (src/dang.f90:87-126) in the oracle, in the reference's order; returns the oracle and, per
    iteration, (chisq, {(comp, index): masked mean per map}, gains) as write_data would print them."""
    dpar, ddata, bands, comps, meta = case
    orc = O.Oracle(bands, copy.deepcopy(comps), ddata)
    good = np.asarray(ddata.masks)[0] != 0
    trace = []
    tuned = {(l, j): (c.tuned[j] if c.tuned else True) for l, c in enumerate(comps) for j in range(c.nindices)}
    for it in range(1, niter + 1):
        for g in dpar.cg_groups:
            f = g.pol_flag[0]
            orc.amp_sample_direct(g.cg_group, f, dpar.ml_mode, dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f), "reference")
        if it > 1:
            for l, c in enumerate(comps):
                for j in range(c.nindices):
                    if not c.sample_index[j]:
                        continue
                    f = c.pol_flag[j][0]
                    s = da.stream_id(it, 1, l, j, f)
                    coarse = c.sample_nside[j] if c.sample_nside else 0
                    nside = int(round((meta["npix_global"] / 12.0) ** 0.5))
                    if c.index_mode and c.index_mode[j] == 1:
                        _, t, _ = orc.sample_index_fullsky(l, j, MAPN[f], dpar.nsample, dpar.ml_mode, dpar.seed, s, tuned=tuned[(l, j)])
                        for q in range(c.nindices):
                            tuned[(l, q)] = tuned[(l, q)] or t
                    elif coarse and coarse != nside:
                        orc.sample_index_mh_coarse(l, j, MAPN[f], dpar.nsample, dpar.ml_mode, dpar.seed, s, nside, coarse)
                    else:
                        if not tuned[(l, j)]:
                            t, _ = orc.tune_perpixel(l, j, MAPN[f], dpar.nsample, dpar.ml_mode, dpar.seed, s)
                            for q in range(c.nindices):
                                tuned[(l, q)] = tuned[(l, q)] or t
                        orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, dpar.ml_mode, dpar.seed, s)
            for j, fit in enumerate(fit_gain or []):
                if fit:
                    orc.gain[j] = orc.fit_band_gain(j, dpar.ml_mode, dpar.seed, da.stream_id(it, 2, 0, 0, 0))
        chisq, _ = orc.chisq(1, meta["nmaps"], ddata.nump)
        means = {(l, j): [orc.indices(l)[j, k][good].mean() for k in range(meta["nmaps"])]
                 for l, c in enumerate(comps) for j in range(c.nindices) if c.sample_index[j]}
        trace.append((chisq, means, orc.gain.copy()))
    return orc, trace


def _check_state(got, orc, case, tol_amp=1e-9, tol_ind=1e-12):
    dpar, ddata, bands, comps, meta = case
    for l, c in enumerate(comps):
        b = orc.amplitude(l)
        assert np.abs(got["amp"][l] - b).max() <= tol_amp * max(np.abs(b).max(), 1.0), (l, np.abs(got["amp"][l] - b).max())
        if c.nindices:
            assert np.abs(got["ind"][l] - orc.indices(l)).max() <= tol_ind, l
    sky, res = orc.sky_model()
    chisq, chi = orc.chisq(1, meta["nmaps"], ddata.nump, sky)
    good = np.asarray(ddata.masks)[0] != 0
    scale = np.abs(sky[:, :, good]).max()
    assert np.abs(got["sky"] - sky)[:, :, good].max() <= 1e-10 * scale
    assert np.abs(got["res"] - res)[:, :, good].max() <= 1e-10 * scale
    assert abs(got["chisq"] - chisq) <= 1e-9 * chisq, (got["chisq"], chisq)


def _fortran_list_directed(v):
    """One real(dp) written with `write(unit,*)`: flang prints 17 significant digits; compare as numbers."""
    return float(v)


def test_wrapper_two_call_and_fused_forms_match_the_oracle_loop(built, tmp_path):
    _need_drive()
    case = make_case("C2", nside=8, gain=[1.0, 1.02, 1.0, 0.97, 1.0])
    dpar, ddata, bands, comps, meta = case
    ddata.gain = np.ones(5)
    ddata.fit_gain = [False, True, False, True, False]
    runs = {}
    for mode, nctx in (("twocall", 1), ("fused", 1), ("fused", 2)):
        d = tmp_path / ("%s%d" % (mode, nctx))
        d.mkdir()
        fin, fout = str(d / "in.bin"), str(d / "out.bin")
        fdrive.write_problem(fin, dpar, ddata, comps, meta, NITER)
        out = fdrive.run(fin, fout, nctx=nctx, mode=mode)
        assert "mode = %s  contexts = %d" % (mode, nctx) in out
        if mode == "twocall":   # the reference's terminal lines (src/dang_cg_mod.f90:163, src/dang_data_mod.f90:538-568)
            assert out.count("Computing a CG search of CG group") == NITER * len(dpar.cg_groups)
            assert " - Chisq: " in out and "Sampling band calibrators" in out and "dust beta I mean:" in out
        runs[(mode, nctx)] = (fdrive.read_result(fout, comps, meta), str(d))
    orc, trace = _oracle_loop(case, NITER, fit_gain=ddata.fit_gain)
    one, outdir = runs[("twocall", 1)]
    _check_state(one, orc, case)
    assert np.abs(one["gain"] - orc.gain).max() <= 1e-12 and abs(one["gain"][1] - 1.02) < 0.02 and one["gain"][0] == 1.0
    # the fused form: the same state bit for bit; two contexts: maps bit for bit, sky-wide sums to rounding
    fused = runs[("fused", 1)][0]
    for key in ("sky", "res", "chi"):
        assert np.array_equal(one[key], fused[key], equal_nan=True), key
    for l, c in enumerate(comps):
        assert np.array_equal(one["amp"][l], fused["amp"][l])
        if c.nindices:
            assert np.array_equal(one["ind"][l], fused["ind"][l])
    assert one["chisq"] == fused["chisq"] and np.array_equal(one["gain"], fused["gain"])
    two = runs[("fused", 2)][0]
    _check_state(two, orc, case)
    assert np.abs(two["gain"] - one["gain"]).max() <= 1e-12

    # ---- write_data_gpu's traces against the reference's edit descriptors on the ORACLE's numbers
    for k, S in enumerate("TQU"):
        with open(os.path.join(outdir, "total_chisq_%s.dat" % S)) as f:
            vals = [float(x) for x in f.read().split()]
        assert len(vals) == NITER
        for it in range(NITER):
            assert abs(vals[it] - trace[it][0]) <= 1e-9 * trace[it][0], (S, it)
        for l, c in enumerate(comps):
            for j in range(c.nindices):
                if not c.sample_index[j]:
                    continue
                with open(os.path.join(outdir, "%s_%s_mean_%s.dat" % (c.label, c.ind_label[j], S))) as f:
                    lines = f.read().splitlines()
                assert len(lines) == NITER
                for it in range(NITER):
                    want = "%12.8f" % trace[it][1][(l, j)][k]                  # fmt = '(3(f12.8))' with one value
                    m = trace[it][1][(l, j)][k]
                    if np.isfinite(m):
                        assert lines[it] == want or abs(float(lines[it]) - m) <= 2e-8, (c.label, j, S, it, lines[it], want)
    for it in range(1, NITER + 1):
        with open(os.path.join(outdir, "band_gains_k%05d.dat" % it)) as f:
            lines = f.read().splitlines()
        assert len(lines) == 3 * meta["nbands"]                                # once per k in pol_type (T, Q, U)
        for j in range(meta["nbands"]):
            assert lines[j][:12] == ("band%02d" % (j + 1)).rjust(12)           # (a12,E16.8): A12 right-justifies
            assert len(lines[j]) == 28 and "E" in lines[j][12:]
            assert abs(float(lines[j][12:]) - trace[it - 1][2][j]) <= 1e-7 * trace[it - 1][2][j]   # E16.8: eight digits
        with open(os.path.join(outdir, "band_offsets_k%05d.dat" % it)) as f:
            assert len(f.read().splitlines()) == 3 * meta["nbands"]


def test_wrapper_runs_a_template_and_monopole_group_over_three_contexts(built, tmp_path):
    """A CG group with a fitted template and a monopole couples every pixel through its global rows: the wrapper hands it to
    dangx_sky_amp_sample, which shares the Schur rows over the contexts.  Three contexts == one context (1e-9)."""
    _need_drive()

    def tweak(dpar, ddata, bands, comps):
        add_globals(dpar, ddata, bands, comps, ("monopole", "hi_fit"), 1, skip_band0=True)
    case = make_case("C2", nside=8, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    for c in comps:
        c.sample_index = [False] * c.nindices       # the amplitude phase only
    res = {}
    for nctx in (1, 3):
        d = tmp_path / ("t%d" % nctx)
        d.mkdir()
        fin, fout = str(d / "in.bin"), str(d / "out.bin")
        fdrive.write_problem(fin, dpar, ddata, comps, meta, 2)
        out = fdrive.run(fin, fout, nctx=nctx, mode="twocall")
        assert "global rows: |b - A x| / |b| =" in out
        res[nctx] = fdrive.read_result(fout, comps, meta)
        with open(str(d / "hi_T_amplitudes.dat")) as f:      # write_data's template trace: header of band labels + a row per iteration
            rows = f.read().splitlines()
        assert len(rows) == 1 + 2 and len(rows[0]) == 17 * meta["nbands"] and len(rows[1]) == 17 * meta["nbands"]
    a, b = res[1], res[3]
    for l, c in enumerate(comps):
        scale = max(np.abs(a["amp"][l]).max(), 1e-30)
        assert np.abs(a["amp"][l] - b["amp"][l]).max() <= 1e-9 * scale, l
        if a["tamp"][l] is not None:
            assert np.abs(a["tamp"][l] - b["tamp"][l]).max() <= 1e-9 * max(np.abs(a["tamp"][l]).max(), 1e-30), l
            assert np.abs(a["tamp"][l]).max() > 0
    assert abs(a["chisq"] - b["chisq"]) <= 1e-9 * a["chisq"]
    # the monopole's amplitudes are the band offsets of the run (update_sky_model, src/dang_data_mod.f90:357-361)
    mono = [l for l, c in enumerate(comps) if c.type == "monopole"][0]
    assert np.array_equal(a["offset"], a["tamp"][mono][0])


def test_wrapper_runs_a_template_group_with_its_sweeps(built, tmp_path):
    """A Q/U template fitted beside the diffuse members of the Q+U group, indices sampled: gibbs_iteration_gpu hands the group's
    solve AND the sweeps on its planes to dangx_sky_plane_set_sample -- pass 1 of the Schur solve on every context, the rows
    shared, then one launch per context that back-substitutes and sweeps.  The two calls of the reference's loop (Schur solve
    with pass 2 and the residual pass, then the sweeps) give the same state (1e-9 / 1e-12), on one context and on three."""
    _need_drive()

    def tweak(dpar, ddata, bands, comps):
        nb = ddata.sig_map.shape[0]
        add_globals(dpar, ddata, bands, comps, ("template",), 2, fit_bands=[nb - 2, nb - 1])
    case = make_case("C2", nside=8, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    res = {}
    for mode, nctx in (("twocall", 1), ("fused", 1), ("fused", 3)):
        d = tmp_path / ("%s%d" % (mode, nctx))
        d.mkdir()
        fin, fout = str(d / "in.bin"), str(d / "out.bin")
        fdrive.write_problem(fin, dpar, ddata, comps, meta, NITER)
        out = fdrive.run(fin, fout, nctx=nctx, mode=mode)
        assert out.count("global rows: |b - A x| / |b| =") == NITER      # the Schur solve's account on either path
        res[(mode, nctx)] = fdrive.read_result(fout, comps, meta)
    a = res[("twocall", 1)]
    tl = [l for l, c in enumerate(comps) if c.type == "template"][0]
    assert np.abs(a["tamp"][tl][1, -2:]).min() > 0 and np.array_equal(a["tamp"][tl][1], a["tamp"][tl][2])
    for key in (("fused", 1), ("fused", 3)):
        b = res[key]
        for l, c in enumerate(comps):
            scale = max(np.abs(a["amp"][l]).max(), 1e-30)
            assert np.abs(a["amp"][l] - b["amp"][l]).max() <= 1e-9 * scale, (key, l)
            if c.nindices:
                assert np.abs(a["ind"][l] - b["ind"][l]).max() <= 1e-12, (key, l)
            if a["tamp"][l] is not None:
                assert np.abs(a["tamp"][l] - b["tamp"][l]).max() <= 1e-9 * max(np.abs(a["tamp"][l]).max(), 1e-30), (key, l)
        assert abs(a["chisq"] - b["chisq"]) <= 1e-9 * a["chisq"], key


def test_wrapper_runs_fullsky_tuner_and_coarse_sweeps(built, tmp_path):
    """sample_spectral_parameters_gpu's other branches, through the chains behind the ABI: a full-sky index whose step is
    tuned first, a per-pixel index with a pending 'Tuning!' block, and a coarse-Nside sweep -- state, step sizes and tuned
    flags against the oracle, on one and on two contexts."""
    _need_drive()

    def tweak(dpar, ddata, bands, comps):
        synch, dust = comps[1], comps[2]
        synch.index_mode = [1]; synch.tuned = [False]; synch.step_size = [2.0 * synch.gauss_prior[0][1]]
        dust.sample_nside = [2, 8]; dust.tuned = [True, True]
    case = make_case("C2", nside=8, start="truth", tweak=tweak)
    dpar, ddata, bands, comps, meta = case
    orc, _ = _oracle_loop(case, 2)
    for nctx in (1, 2):
        d = tmp_path / ("f%d" % nctx)
        d.mkdir()
        fin, fout = str(d / "in.bin"), str(d / "out.bin")
        fdrive.write_problem(fin, dpar, ddata, comps, meta, 2)
        out = fdrive.run(fin, fout, nctx=nctx, mode="fused")     # nothing is fusable here: the plan says so, the order is the reference's
        assert "Sampling fullsky" in out and "Sampling per-pixel at nside    2" in out
        got = fdrive.read_result(fout, comps, meta)
        _check_state(got, orc, case, tol_ind=1e-11 if nctx > 1 else 1e-12)
        assert got["tuned"][1][0] and got["step"][1][0] == orc._comps[1].step_size[0] < 2.0 * comps[1].gauss_prior[0][1]
        a = got["ind"][1]
        assert np.all(a[0, 0] == a[0, 0, 0])                     # one value for the whole sky
