#!/usr/bin/env python3
"""Per-launch time at the C3 size, plane set by plane set: the fused solve + first sweep (k_amp_index) against the two
launches it replaces (k_amp_reg + k_index_mh_reg)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dang_amd as da  # noqa: E402
from dang_amd import synth  # noqa: E402
from dang_amd import _lib as L  # noqa: E402

dev = torch.device("cuda", 0)
dpar, ddata, bands, comps, meta = synth.make_sky("C3", device=dev, as_numpy=False, start="truth")
eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
mapn = {1: 1, 8: -1}
cases = [("T", 1, L.FLAG_T, 1), ("Q+U", 2, L.FLAG_QU, 5)]   # (name, group, flag, sampled component = synch of that group)
for rnd in range(3):
    for name, grp, flag, comp in cases:
        for fused in (True, False):
            for rep in range(2):
                if fused:
                    eng.amp_index_sample(grp, flag, "sample", 5, 17 + rep, comp, 0, mapn[flag], 10, 5, 99 + rep, want_counts=False)
                else:
                    eng.amp_sample(grp, flag, "sample", 5, 17 + rep, want_counts=False)
                    eng.index_sample(comp, 0, mapn[flag], 10, "sample", 5, 99 + rep, want_counts=False)
            eng.profile(True)
            for rep in range(5):
                if fused:
                    eng.amp_index_sample(grp, flag, "sample", 5, 27 + rep, comp, 0, mapn[flag], 10, 5, 199 + rep, want_counts=False)
                else:
                    eng.amp_sample(grp, flag, "sample", 5, 27 + rep, want_counts=False)
                    eng.index_sample(comp, 0, mapn[flag], 10, "sample", 5, 199 + rep, want_counts=False)
            eng.synchronize()
            p = eng.profile_get()
            eng.profile(False)
            if rnd == 2:
                tot = sum(v["avg_ms"] for k, v in p.items() if k != "k_reduce")
                print("%-4s %-6s %.3f ms  %s" % (name, "fused" if fused else "two", tot, {k: round(v["avg_ms"], 3) for k, v in p.items()}))
