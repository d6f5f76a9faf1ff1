#!/usr/bin/env python3
"""C3 + a Q/U template fitted at three bands, every sampled index in full-sky mode (the combination the reference's plotting scripts
suggest: template amplitudes per band beside sky-wide spectral indices).  ms per Gibbs iteration, chi^2, the recovered amplitudes."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dang_amd as da  # noqa: E402
from dang_amd import synth  # noqa: E402

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
dpar, ddata, bands, comps, meta = synth.make_sky("C3", nside=nside, device=dev, as_numpy=False, start="truth")
tl = synth.add_qu_template(ddata, comps, meta)
for c in comps:
    if c.nindices:
        c.index_mode = [1] * c.nindices
        c.step_size = [0.05 * g[1] for g in c.gauss_prior]
eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
for it in (1, 2):
    da.gibbs_iteration(dpar, ddata, it, want_counts=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(3, 3 + steps):
    da.gibbs_iteration(dpar, ddata, it, want_counts=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("template + full-sky indices: %.2f ms per Gibbs iteration (%.2f it/s); chisq %.6f; template amplitudes %s"
      % (1e3 * dt, 1.0 / dt, ddata.chisq, np.round(eng.get_template_amplitudes(tl)[1, 7:], 4)))
