#!/usr/bin/env python3
"""Whole Gibbs iterations of the C3 model with every sampled index in FULL-SKY mode (index_mode = 1, SURVEY 8f rank 2: one index
value for the whole sky, each Metropolis step a pass over the maps and a sky-wide sum).  Prints ms per iteration and the profile."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dang_amd as da  # noqa: E402
from dang_amd import synth  # noqa: E402

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
dpar, ddata, bands, comps, meta = synth.make_sky("C3", nside=nside, device=dev, as_numpy=False)
for c in comps:
    c.index_mode = [1] * c.nindices
    c.step_size = [0.05 * g[1] for g in c.gauss_prior]
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
print("full-sky index mode: %.2f ms per Gibbs iteration (%.2f it/s); chisq %.6f; NUMSAMPLE %d" % (1e3 * dt, 1.0 / dt, ddata.chisq, dpar.nsample))
for k, v in eng.profile_get().items():
    print("  %-14s %4d launches per iteration, %8.3f ms per iteration" % (k, v["launches"] // steps, v["total_ms"] / steps))
for l, c in enumerate(comps):
    for j in range(c.nindices):
        if c.sample_index[j]:
            x = eng.get_indices(l)[j, 0 if c.pol_flag[j][0] == 1 else 1, 0]
            print("  %-8s %-5s %.5f" % (c.label, c.ind_label[j], x))
