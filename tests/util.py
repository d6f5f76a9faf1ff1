"""Shared helpers for the parity tests: build one case for both the HIP path and the oracle."""
import copy

import numpy as np

import dang_amd as da
from dang_amd import _lib as L
from dang_amd import synth

import oracle_ffi as O

MAPN = {L.FLAG_T: 1, L.FLAG_Q: 2, L.FLAG_U: 3, L.FLAG_QU: -1}

# stated fp64 tolerances (see DESIGN.md "Tolerances")
TOL_SED = 2e-13       # relative; exp(beta*ln r) vs pow, exp(x)-1 cancellation at low frequency
TOL_AMP = 1e-9        # relative to the map's max |a| (block condition number x eps)
TOL_AMP_TIGHT = 1e-11
TOL_INDEX = 1e-12     # absolute on index values (same proposals unless an accept decision flips)
TOL_CHISQ = 1e-10     # relative


def make_case(config="C2", nside=4, tweak=None, **kw):
    """Returns (dpar, ddata, bands, comps, meta); `tweak(dpar, ddata, bands, comps)` may edit in place."""
    dpar, ddata, bands, comps, meta = synth.make_sky(config, nside=nside, **kw)
    if tweak:
        tweak(dpar, ddata, bands, comps)
    return dpar, ddata, bands, comps, meta


def pair(case, device=0):
    """(engine, oracle) initialised from the same host arrays."""
    dpar, ddata, bands, comps, meta = case
    orc = O.Oracle(bands, copy.deepcopy(comps), ddata)
    eng = da.Engine(bands, comps, ddata, npix_global=meta["npix_global"], pix0=meta["pix0"], device=device)
    ddata.engine = eng
    return eng, orc


def relmax(a, b):
    scale = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / scale


def assert_amps_close(eng, orc, ncomp, tol, what=""):
    for l in range(ncomp):
        a, b = eng.get_amplitude(l), orc.amplitude(l)
        assert np.isfinite(a).all(), "non-finite amplitude in component %d %s" % (l, what)
        r = relmax(a, b)
        assert r <= tol, "component %d amplitude mismatch %.3e > %.1e %s" % (l, r, tol, what)


def assert_indices_close(eng, orc, comps, tol=TOL_INDEX, what=""):
    for l, c in enumerate(comps):
        if c.nindices:
            a, b = eng.get_indices(l), orc.indices(l)
            d = np.abs(a - b).max()
            assert d <= tol, "component %d index mismatch %.3e > %.1e %s" % (l, d, tol, what)
