"""Problem / result files of the Fortran drivers that run the reference-side wrapper on a GPU
(fortran/reference_side/dang_gpu_drive.f90: `program dang`'s loop through dang_gpu_mod.f90 compiled against mock modules).
Used by tests/test_refside_gpu.py and by bench.py's `fortran_seam` leg; plain marshalling, no computation."""
import os
import struct
import subprocess

import numpy as np

from . import _build
from . import _lib as L
from .api import comp_desc


def _s(text, n):
    return text.encode()[:n].ljust(n)


def write_problem(path, dpar, ddata, comps, meta, niter, nsample=None, seed=None):
    """The state `program dang` holds after initialisation (src/dang.f90:43-79), in the layout dang_gpu_drive.f90 reads."""
    npix, nb, nmaps = meta["npix"], meta["nbands"], meta["nmaps"]
    nside = int(round((meta["npix_global"] / 12.0) ** 0.5))
    nsample = dpar.nsample if nsample is None else nsample
    seed = dpar.seed if seed is None else seed
    gain = np.ones(nb) if ddata.gain is None else np.asarray(ddata.gain, dtype="<f8")
    off = np.zeros(nb) if ddata.offset is None else np.asarray(ddata.offset, dtype="<f8")
    fit = list(ddata.fit_gain) if ddata.fit_gain else [False] * nb
    with open(path, "wb") as f:
        f.write(struct.pack("<8iqdi", npix, nmaps, nb, len(comps), nsample, niter, len(dpar.cg_groups), nside, seed,
                            float(ddata.nump), L.ML_CODES[dpar.ml_mode]))
        f.write(np.asarray(meta["freqs_ghz"], dtype="<f8").tobytes())
        f.write(gain.astype("<f8").tobytes() + off.astype("<f8").tobytes())
        f.write(np.asarray([int(bool(x)) for x in fit], dtype="<i4").tobytes())
        for j in range(nb):
            f.write(_s("band%02d" % (j + 1), 32))
        for c in comps:
            f.write(bytes(comp_desc(c)))
            f.write(_s(c.label, 16))
            labels = list(c.ind_label) + ["", ""]
            f.write(_s(labels[0], 16) + _s(labels[1], 16))
            two = lambda v, d: [int(x) for x in (list(v) + [d, d])[:2]]
            f.write(struct.pack("<2i", *two([bool(x) for x in c.sample_index], 0)))
            f.write(struct.pack("<2i", *two([x[0] for x in c.pol_flag], 0)))
            f.write(struct.pack("<2i", *two(c.index_mode if c.index_mode else [2] * c.nindices, 2)))
            f.write(struct.pack("<2i", *two(c.sample_nside if c.sample_nside else [0] * c.nindices, 0)))
            f.write(struct.pack("<2i", *two([bool(x) for x in c.tuned] if c.tuned else [True] * c.nindices, 1)))
            f.write(struct.pack("<i", int(c.nfit)))
            f.write(np.asarray([int(bool(x)) for x in (c.corr if len(c.corr) else [False] * nb)], dtype="<i4").tobytes())
        for g in dpar.cg_groups:
            ntemp = sum(1 for c in comps if c.cg_group == g.cg_group and c.sample_amplitude and c.type in ("template", "monopole", "hi_fit"))
            f.write(struct.pack("<4i", g.cg_group, g.pol_flag[0], int(bool(g.sample)), ntemp))
        for a in (ddata.sig_map, ddata.rms_map, ddata.masks):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
        for c in comps:
            f.write(np.ascontiguousarray(c.amplitude, dtype="<f8").tobytes())
            if c.nindices:
                f.write(np.ascontiguousarray(c.indices, dtype="<f8").tobytes())
            if c.type in ("template", "monopole", "hi_fit"):
                f.write(np.ascontiguousarray(c.template, dtype="<f8").tobytes())
                ta = np.zeros((nmaps, nb)) if c.template_amplitudes is None else c.template_amplitudes
                f.write(np.ascontiguousarray(ta, dtype="<f8").tobytes())
                f.write(np.ones(nmaps, dtype="<f8").tobytes())          # c%temp_norm


def read_result(path, comps, meta, maps=True):
    npix, nb, nmaps = meta["npix"], meta["nbands"], meta["nmaps"]
    out = np.fromfile(path, dtype="<f8")
    res = dict(chisq=out[0], tcmb=out[1], secs=out[2], gain=out[3:3 + nb], offset=out[3 + nb:3 + 2 * nb])
    p = 3 + 2 * nb
    res["step"], res["tuned"] = [], []
    for c in comps:
        res["step"].append(out[p:p + 2].copy()); res["tuned"].append(out[p + 2:p + 4] != 0); p += 4
    if not maps:
        return res
    n3, n2 = nb * nmaps * npix, nmaps * npix
    res["sky"], res["res"], res["chi"] = (out[p:p + n3].reshape(nb, nmaps, npix), out[p + n3:p + 2 * n3].reshape(nb, nmaps, npix),
                                          out[p + 2 * n3:p + 2 * n3 + n2].reshape(nmaps, npix))
    p += 2 * n3 + n2
    res["amp"], res["ind"], res["tamp"] = [], [], []
    for c in comps:
        res["amp"].append(out[p:p + n2].reshape(nmaps, npix)); p += n2
        res["ind"].append(out[p:p + c.nindices * n2].reshape(c.nindices, nmaps, npix) if c.nindices else None)
        p += c.nindices * n2
        if c.type in ("template", "monopole", "hi_fit"):
            res["tamp"].append(out[p:p + nmaps * nb].reshape(nmaps, nb)); p += nmaps * nb
        else:
            res["tamp"].append(None)
    assert p == out.size, (p, out.size)
    return res


def run(problem, result, nctx=1, mode="twocall", tile=1, timeout=900):
    """Run the driver; returns its stdout.  Raises when flang is absent or the run fails (no fallback)."""
    exe = _build.build_reference_drive()
    if exe is None:
        raise RuntimeError("flang is not available: the Fortran driver cannot be built")
    env = dict(os.environ)
    r = subprocess.run([exe, problem, result, str(nctx), mode, str(tile)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=timeout, env=env)
    if r.returncode != 0 or "dang_gpu_drive ok" not in r.stdout:
        raise RuntimeError("dang_gpu_drive failed (%d):\n%s" % (r.returncode, r.stdout[-4000:]))
    return r.stdout
