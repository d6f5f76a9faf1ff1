#!/usr/bin/env python3
"""The template model (tools/bench_template_iter.py) with the temperature chain and the polarisation chain of an iteration on two HIP
streams (two contexts over the same resident maps, as `bench.py --streams 2`): pass 1 of the Q+U group's Schur solve is latency bound
(half its issue slots idle) and overlaps with the T group's launch.  Prints ms per iteration for one and for two streams."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dang_amd as da  # noqa: E402
from dang_amd import synth  # noqa: E402
from dang_amd import _lib as L  # noqa: E402

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
dpar, ddata, bands, comps, meta = synth.make_sky("C3", nside=nside, device=dev, as_numpy=False, start="truth")
tl = synth.add_qu_template(ddata, comps, meta)
side = torch.cuda.Stream(device=dev)
engT = da.Engine(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
engP = da.Engine(bands, comps, ddata, npix_global=meta["npix_global"], device=0, stream=side.cuda_stream)


def sweeps(g, f, it):
    return [(l, j, da.stream_id(it, 1, l, j, f)) for l, c in enumerate(comps) for j in range(c.nindices)
            if c.cg_group == g and c.sample_index[j] and f in c.pol_flag[j]]


def iteration(it, two):
    for g in dpar.cg_groups:
        f = g.pol_flag[0]
        eng = engP if (two and f != L.FLAG_T) else engT
        eng.plane_set_sample(g.cg_group, f, "sample", dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f), sweeps(g.cg_group, f, it),
                             dpar.nsample, dpar.seed, want_counts=False)
    # chi^2 of the iteration: T planes from one context, Q/U planes from the other (by-products of the launches)
    a = engT.chisq_cached(1, 1, 1)
    b = (engP if two else engT).chisq_cached(1, 2, 3)
    return (a + b) / meta["nbands"] / ddata.nump


for two in (False, True, False, True):
    for it in (1, 2):
        iteration(it, two)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(3, 3 + steps):
        chi = iteration(it, two)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ta = (engP if two else engT).get_template_amplitudes(tl)[1, 7:]
    print("%d stream(s): %.2f ms per Gibbs iteration (%.1f it/s); chisq %.6f; template amplitudes %s" % (2 if two else 1, 1e3 * dt, 1.0 / dt, chi, np.round(ta, 4)))
