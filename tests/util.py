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


def shard_engines(case, nsh, bounds=None, device=0):
    """`nsh` pixel-shard contexts of ONE process over the case's whole-sky host arrays (contiguous RING ranges, dist.shard_range),
    each with its own copies of the state maps -- what dangx_multi_mod / the dangx_sky_* entry points drive."""
    dpar, ddata, bands, comps, meta = case
    engs = []
    for r in range(nsh):
        pix0, npix = da.dist.shard_range(meta["npix_global"], r, nsh, bounds)
        sl = slice(pix0, pix0 + npix)
        cs = copy.deepcopy(comps)
        for c in cs:
            c.amplitude = np.ascontiguousarray(c.amplitude[:, sl])
            if c.indices is not None:
                c.indices = np.ascontiguousarray(c.indices[:, :, sl])
            if getattr(c, "template", None) is not None:
                c.template = np.ascontiguousarray(c.template[:, sl])
        dd = da.DangData(sig_map=np.ascontiguousarray(ddata.sig_map[:, :, sl]), rms_map=np.ascontiguousarray(ddata.rms_map[:, :, sl]),
                         masks=np.ascontiguousarray(ddata.masks[:, sl]), gain=ddata.gain, offset=ddata.offset, pol_type=ddata.pol_type,
                         nump=ddata.nump)
        engs.append(da.Engine(bands, cs, dd, npix_global=meta["npix_global"], pix0=pix0, device=device))
    return engs
