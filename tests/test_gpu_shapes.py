"""Model shapes without a built-in kernel instantiation (the library carries 3 / 5 / 6 / 8 / 10 / 20 bands): the register chain,
the two-index sweep and the fused solve + first sweep are specialised at run time (csrc/dangx_rtc.hip) and must give the
oracle's numbers like the built-in shapes do -- and it must be THOSE kernels that ran, not the LDS-column fallback."""
import numpy as np
import pytest

import dang_amd as da
from dang_amd import _lib as L

from util import MAPN, make_case, pair

pytestmark = pytest.mark.gpu


def _run_and_compare(case, niter=3):
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for it in range(1, niter + 1):
        if it == 1:
            da.sample_cg_groups(dpar, ddata, it=it)
        else:
            da.gibbs_iteration(dpar, ddata, it)
        for g in dpar.cg_groups:
            for f in g.pol_flag:
                orc.amp_sample_direct(g.cg_group, f, dpar.ml_mode, dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f), "reference")
        if it > 1:
            for l, c in enumerate(comps):
                for j in range(c.nindices):
                    if c.sample_index[j]:
                        for f in c.pol_flag[j]:
                            orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, dpar.ml_mode, dpar.seed, da.stream_id(it, 1, l, j, f))
    for l, c in enumerate(comps):
        b = orc.amplitude(l)
        assert np.abs(eng.get_amplitude(l) - b).max() <= 1e-9 * max(np.abs(b).max(), 1.0), l
        if c.nindices:
            assert np.abs(eng.get_indices(l) - orc.indices(l)).max() <= 1e-12, l
    ochisq, _ = orc.chisq(1, meta["nmaps"], ddata.nump)
    assert abs(ddata.chisq - ochisq) <= 1e-9 * ochisq
    return eng


@pytest.mark.parametrize("config,nbands,ng", [("C3", 9, 4), ("C2", 7, 3)])
def test_unlisted_band_counts_run_the_specialised_kernels(built, config, nbands, ng):
    eng = _run_and_compare(make_case(config, nside=8, nbands=nbands))
    names = eng.rtc_kernels()
    # both plane sets: the group's solve and every sweep on its planes (synchrotron beta | dust beta + T) in one launch; the first
    # iteration's stand-alone sweeps (no solve to go with) are register-chain launches, specialised too
    for sp in (1, 2):
        assert "dxk::k_plane_set<%d, %d, %d, 1, 1, 1, 10, 0, 0, 0>" % (sp, nbands, ng) in names, names
    # nothing went through the LDS-column form: every sweep of the run was a register-chain launch
    assert all(n.startswith("dxk::") for n in names)


def test_fourteen_bands_polarisation_runs_as_lane_pairs(built):
    """two planes of 14 bands do not fit one lane at two waves per SIMD: the chain is specialised in its lane-pair form"""
    eng = _run_and_compare(make_case("C2", nside=8, nbands=14), niter=2)
    names = eng.rtc_kernels()
    assert any(n.startswith("dxk::k_index_mh_") and n.endswith(", 2, 14, 2>") for n in names), names


def test_specialisation_can_be_switched_off(built, monkeypatch):
    """DANGX_RTC=0: the same model through the LDS-column kernels -- same numbers (the parity bar does not depend on the form)"""
    import subprocess, sys, os
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_shapes as t\n"
            "from util import make_case\n"
            "eng = t._run_and_compare(make_case('C2', nside=8, nbands=7), niter=2)\n"
            "assert eng.rtc_kernels() == []\nprint('ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                               os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DANGX_RTC="0"), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout


def test_specialised_kernels_come_from_the_disk_cache_the_second_time(built, tmp_path):
    """The code objects hiprtc produces are kept under $DANGX_CACHE_DIR: a second process with the same model loads them
    instead of compiling (DANGX_RTC_VERBOSE reports every compilation)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_shapes as t\n"
            "from util import make_case\n"
            "eng = t._run_and_compare(make_case('C2', nside=8, nbands=7), niter=2)\n"
            "print('kernels', len(eng.rtc_kernels()))\n") % (root, os.path.join(root, "tests"))
    env = dict(os.environ, DANGX_CACHE_DIR=str(tmp_path / "cache"), DANGX_RTC_VERBOSE="1")
    runs = [subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
            for _ in range(2)]
    import re
    for r in runs:
        assert r.returncode == 0 and re.search(r"kernels \d+", r.stdout), r.stdout
    n = int(re.search(r"kernels (\d+)", runs[0].stdout).group(1))
    assert n >= 2 and "kernels %d" % n in runs[1].stdout
    assert runs[0].stdout.count("[dangx] specialising") == n and "[dangx] specialising" not in runs[1].stdout, (runs[0].stdout, runs[1].stdout)
    files = sorted(os.listdir(str(tmp_path / "cache")))
    assert len([f for f in files if f.endswith(".hsaco")]) == n and len([f for f in files if f.endswith(".sym")]) == n, files


@pytest.mark.parametrize("nbands", [13, 15])
def test_thirteen_and_fifteen_bands(built, nbands):
    """Odd band counts above 10 cannot split over lane pairs.  13: both plane sets still run as ONE launch each in one lane (two
    planes of 13 bands at two waves per SIMD with a few spilled registers); 15: the T plane set does, the Q+U plane set takes the
    separate solve and chains (at two waves per SIMD as well).  Same numbers as the oracle either way."""
    # (three iterations: the index maps start spatially constant, which the first sweeps end)
    eng = _run_and_compare(make_case("C3", nside=4, nbands=nbands), niter=3)
    names = eng.rtc_kernels()
    assert "dxk::k_plane_set<1, %d, 4, 1, 1, 1, 10, 0, 0, 0>" % nbands in names, names
    if nbands == 13:
        assert "dxk::k_plane_set<2, 13, 4, 1, 1, 1, 10, 0, 0, 0>" in names, names
    else:
        assert not any(n.startswith("dxk::k_plane_set<2,") for n in names), names
        assert any(n.startswith("dxk::k_index_mh_") and n.endswith(", 2, 15, 1>") for n in names), names
