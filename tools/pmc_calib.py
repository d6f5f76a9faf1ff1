#!/usr/bin/env python3
"""Kernels of KNOWN byte count for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this repo's access
pattern (8 B per lane, coalesced, pixel-major): the CG vector kernels k_cg_vec on n = 4 comps x 2 planes x npix.
Dispatch order of k_cg_vec in one optimize-mode device-CG call with i_max = 4:
  mode 0 (read 2n, write 2n), then per iteration: mode 1 (read 4n, write 2n), mode 2 (read 2n, write n).
Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; tools/prof_summary.py prints the per-dispatch means."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dang_amd as da  # noqa: E402
from dang_amd import synth  # noqa: E402
from dang_amd import _lib as L  # noqa: E402

dev = torch.device("cuda", 0)
dpar, ddata, bands, comps, meta = synth.make_sky("C3", device=dev, as_numpy=False)
eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
it, _ = eng.amp_sample(2, L.FLAG_QU, "optimize", 1, 1, solver="cg", i_max=4, converge=0.0)
n = 4 * 2 * meta["npix"]
print("cg iterations", it, "n =", n, "bytes per vector = %.4e" % (8.0 * n))
