#!/usr/bin/env python3
"""Whole Gibbs iterations of the C3 model with the synchrotron index in full-sky mode and the dust indices per pixel (a common
choice: one beta_s for the sky, dust beta / T per pixel).  Prints ms per iteration and the launch profile."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dang_amd as da  # noqa: E402
from dang_amd import synth  # noqa: E402

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
dpar, ddata, bands, comps, meta = synth.make_sky("C3", nside=nside, device=dev, as_numpy=False)
for c in comps:
    if c.label.startswith("synch"):
        c.index_mode = [1] * c.nindices
        c.step_size = [0.05 * g[1] for g in c.gauss_prior]
eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
for it in (1, 2):
    da.gibbs_iteration(dpar, ddata, it, want_counts=False)
eng.profile(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(3, 3 + steps):
    da.gibbs_iteration(dpar, ddata, it, want_counts=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("synch beta full-sky, dust per pixel: %.2f ms per Gibbs iteration (%.2f it/s); chisq %.6f" % (1e3 * dt, 1.0 / dt, ddata.chisq))
for k, v in eng.profile_get(by_planes=True).items():
    print("  %-22s %4d launches per iteration, %8.3f ms per iteration" % (k, v["launches"] // steps, v["total_ms"] / steps))
