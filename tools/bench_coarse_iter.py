#!/usr/bin/env python3
"""Whole Gibbs iterations of the C3 model with every sampled index drawn per pixel at a COARSER Nside (sample_nside < nside, SURVEY
8f rank 4): stage the cleaned data, degrade data / rms / mask, one chain per coarse pixel, write the coarse index map back."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dang_amd as da  # noqa: E402
from dang_amd import synth  # noqa: E402

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
coarse = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
dpar, ddata, bands, comps, meta = synth.make_sky("C3", nside=nside, device=dev, as_numpy=False)
for c in comps:
    c.sample_nside = [coarse] * c.nindices
eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
for it in (1, 2):
    da.gibbs_iteration(dpar, ddata, it)
eng.profile(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(3, 3 + steps):
    da.gibbs_iteration(dpar, ddata, it)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("index sampling at Nside %d of %d: %.2f ms per Gibbs iteration (%.2f it/s); chisq %.6f" % (coarse, nside, 1e3 * dt, 1.0 / dt, ddata.chisq))
for k, v in eng.profile_get().items():
    print("  %-14s %4d launches per iteration, %8.3f ms per iteration" % (k, v["launches"] // steps, v["total_ms"] / steps))
