#!/usr/bin/env python3
"""Per-launch time of the amplitude kernel at the C3 size by plane set: T, Q alone, U alone, Q+U in one launch
(group 2 holds the same four components on every plane, so the launches differ only in which planes they walk)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dang_amd as da  # noqa: E402
from dang_amd import synth  # noqa: E402
from dang_amd import _lib as L  # noqa: E402

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
dpar, ddata, bands, comps, meta = synth.make_sky("C3", nside=nside, device=dev, as_numpy=False, start="truth")
eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
cases = [("T (group 1)", 1, L.FLAG_T), ("Q (group 2)", 2, L.FLAG_Q), ("U (group 2)", 2, L.FLAG_U), ("Q+U (group 2)", 2, L.FLAG_QU)]
for name, grp, flag in cases * 3:   # the first round also warms the clocks up: read the later ones
    for rep in range(2):
        eng.amp_sample(grp, flag, "sample", 5, 17 + rep)
    eng.profile(True)
    for rep in range(5):
        eng.amp_sample(grp, flag, "sample", 5, 19 + rep)
    eng.synchronize()
    p = eng.profile_get()
    eng.profile(False)
    print("%-14s %s" % (name, {k: round(v["avg_ms"], 3) for k, v in p.items()}))
