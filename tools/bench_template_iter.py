#!/usr/bin/env python3
"""Whole Gibbs iterations of the C3 model WITH a fitted polarisation template in the Q+U group (the shape of the
runs dang was written for -- diffuse components beside a dust template whose per-band amplitudes are fitted): the Schur solve of
the coupled group, the T group's plane-set launch and the Q+U sweeps beside a template.  Prints ms per iteration and the launch
profile by kernel family."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dang_amd as da  # noqa: E402
from dang_amd import synth  # noqa: E402

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
plain = len(sys.argv) > 3 and sys.argv[3] == "plain"      # the same sky without the template: what the template costs
start = sys.argv[4] if len(sys.argv) > 4 else "truth"
counts = len(sys.argv) > 5 and sys.argv[5] == "counts"   # read the accepted-proposal counts back after every launch (a diagnostic)
dev = torch.device("cuda", 0)
dpar, ddata, bands, comps, meta = synth.make_sky("C3", nside=nside, device=dev, as_numpy=False, start=start)
npix, nb = meta["npix"], meta["nbands"]
mono = len(sys.argv) > 3 and sys.argv[3] == "mono"       # a monopole fitted in the T group instead
if mono:
    nfit = int(os.environ.get("TEMPLATE_NFIT", "3"))     # bands the monopole is fitted at (the first nfit)
    synth.add_monopole(ddata, comps, meta, fit_bands=tuple(range(nfit)), amplitudes=tuple([3.0, -2.0, 5.0, 1.0, -4.0, 2.5, -1.5, 0.5, 3.5, -2.5][:nfit]))
elif not plain:
    nfit = int(os.environ.get("TEMPLATE_NFIT", "3"))     # bands the template is fitted at (the last nfit)
    fit = tuple(range(nb - nfit, nb))
    synth.add_qu_template(ddata, comps, meta, fit_bands=fit, amplitudes=tuple([2.0, -1.5, 0.7, 1.2, -0.8, 0.5, 1.7, -1.1, 0.9, 0.4][:nfit]))
eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
for it in (1, 2):
    da.gibbs_iteration(dpar, ddata, it, want_counts=counts)
eng.profile(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(3, 3 + steps):
    da.gibbs_iteration(dpar, ddata, it, want_counts=counts)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
prof = eng.profile_get()
print("template model: %.2f ms per Gibbs iteration (%.1f it/s); chisq %.6f; template amplitudes %s"
      % (1e3 * dt, 1.0 / dt, ddata.chisq, "-" if plain else np.round(eng.get_template_amplitudes(len(comps) - 1)[0 if mono else 1], 4)))
if not plain:
    print("  Schur: residual %.2e (bound when not measured), refinements %d" % (eng.schur_info()[0][0], eng.schur_info()[1]))
for k, v in prof.items():
    print("  %-14s %3d launches per iteration, %8.3f ms per iteration" % (k, v["launches"] // steps, v["total_ms"] / steps))
