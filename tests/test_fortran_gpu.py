"""The Fortran binding (fortran/dangx_mod.f90, ISO_C_BINDING) drives the same library: a flang-built
driver with reference-shaped arrays must produce exactly what the Python/ctypes host path produces."""
import os
import struct
import subprocess

import numpy as np
import pytest

import dang_amd as da
from dang_amd import _build, synth
from dang_amd import _lib as L

pytestmark = pytest.mark.gpu


def test_fortran_driver_matches_python_host(built, tmp_path):
    exe = _build.build_fortran()
    if exe is None:
        pytest.skip("flang not available")
    dpar, ddata, bands, comps, meta = synth.make_sky("C1", nside=8)
    npix, nb, nsample, seed = meta["npix"], meta["nbands"], 10, 4321
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<iiiiq", npix, 1, nb, nsample, seed))
        for a in (np.array(meta["freqs_ghz"]), ddata.sig_map, ddata.rms_map, ddata.masks, comps[0].amplitude,
                  comps[1].amplitude, comps[0].indices, comps[1].indices):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
    r = subprocess.run([exe, fin, fout], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "dangx_fsmoke ok" in r.stdout
    # the Fortran all-reduce callback (c_funloc of a bind(C) function) served the device CG: 1 + 2 per iteration
    assert "allreduce callback calls = 7  cg iterations = 4" in r.stdout, r.stdout
    out = np.fromfile(fout, dtype="<f8")
    chisq_f, rest = out[0], out[1:]
    amp1, amp2, ind1, ind2 = np.split(rest, np.cumsum([npix, npix, npix]))
    eng = da.initialize(bands, comps, ddata, npix_global=npix, device=0)
    eng.amp_sample(1, L.FLAG_T, "sample", seed, da.stream_id(1, 0, 1, 0, L.FLAG_T))
    eng.index_sample(0, 0, 1, nsample, "sample", seed, da.stream_id(2, 1, 0, 0, L.FLAG_T))
    eng.index_sample(1, 1, 1, nsample, "sample", seed, da.stream_id(2, 1, 1, 1, L.FLAG_T))
    assert eng.sky_model_chisq(1, 1) == chisq_f
    assert np.array_equal(eng.get_amplitude(0).ravel(), amp1) and np.array_equal(eng.get_amplitude(1).ravel(), amp2)
    assert np.array_equal(eng.get_indices(0).ravel(), ind1) and np.array_equal(eng.get_indices(1).ravel(), ind2)
