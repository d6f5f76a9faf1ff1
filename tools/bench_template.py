#!/usr/bin/env python3
"""Timing of one amplitude solve of a CG group WITH a fitted template at the C3 size: the direct
Schur-complement path against the reference's CG on the device (i_max = 100, converge = 1e-8)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dang_amd as da  # noqa: E402
from dang_amd import synth  # noqa: E402
from dang_amd import _lib as L  # noqa: E402
from dang_amd.api import DangComps  # noqa: E402

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
dpar, ddata, bands, comps, meta = synth.make_sky("C3", nside=nside, device=dev, as_numpy=False, start="truth")
npix, nb = meta["npix"], meta["nbands"]
g = torch.Generator(device="cpu").manual_seed(3)
tmpl = torch.zeros(3, npix, dtype=torch.float64)
tmpl[1:] = torch.randn(2, npix, generator=g, dtype=torch.float64)
corr = [j in (7, 8, 9) for j in range(nb)]
truth = np.zeros((3, nb)); truth[1:, 7:] = [2.0, -1.5, 0.7]
for k in (1, 2):
    for j in range(nb):
        ddata.sig_map[j, k] += truth[k, j] * tmpl[k].to(dev)
comps.append(DangComps(label="tmpl", type="template", nu_ref=100.0, cg_group=2, nindices=0, nfit=3, corr=corr,
                       template=tmpl.numpy(), template_amplitudes=np.zeros((3, nb)),
                       amplitude=torch.zeros(3, npix, dtype=torch.float64, device=dev)))
eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
for solver in ("cg", "direct", "direct"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it, bad = eng.amp_sample(2, L.FLAG_QU, "sample", 5, 17, solver=solver, i_max=100, converge=1e-8)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ta = eng.get_template_amplitudes(len(comps) - 1)[1, 7:]
    print("%-6s %8.1f ms   iterations/nullity %4d   template amplitudes %s (truth 2.0 -1.5 0.7)" % (solver, 1e3 * dt, it, np.round(ta, 4)))
