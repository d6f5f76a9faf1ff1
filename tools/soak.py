#!/usr/bin/env python3
"""Long Gibbs run as a sampler sanity check: python tools/soak.py [nside] [iterations] [reference|correct].
DANGX_SOAK_TWOCALL=1: the two calls instead of da.gibbs_iteration.  DANGX_SOAK_TEMPLATE=1: with a Q/U template fitted at the last
three bands of the Q+U group (its amplitude trace is printed: injected 2.0, -1.5, 0.7).
Prints the chi^2 trajectory, the number of non-SPD blocks met, and the spread of the sampled indices."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dang_amd as da  # noqa: E402
from dang_amd import synth  # noqa: E402

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 64
niter = int(sys.argv[2]) if len(sys.argv) > 2 else 400
modes = sys.argv[3:] or ["reference", "correct"]
dev = torch.device("cuda", 0)
for fluct in modes:
    dpar, ddata, bands, comps, meta = synth.make_sky("C3", nside=nside, fluct_mode=fluct, device=dev, as_numpy=False)
    tl = None
    if os.environ.get("DANGX_SOAK_TEMPLATE", "0") == "1":
        nb = meta["nbands"]
        tl = synth.add_qu_template(ddata, comps, meta, fit_bands=tuple(range(nb - 3, nb)))
    eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
    ta_trace = []
    marks = sorted(set([1, 2, 5, 10, 20, 50, 100, 200] + list(range(400, niter + 1, 400)) + [niter]))
    tr, nbad = [], 0
    for it in range(1, niter + 1):
        if it > 1 and os.environ.get("DANGX_SOAK_TWOCALL", "0") != "1":
            info, _ = da.gibbs_iteration(dpar, ddata, it)        # the plane-set launches bench.py runs
        else:
            info = da.sample_cg_groups(dpar, ddata, it=it, defer_chisq=(it > 1))
            if it > 1:
                da.sample_spectral_parameters(dpar, ddata, it=it)
        nbad += sum(b for (_, _, _, b) in info)
        if tl is not None:
            ta_trace.append(eng.get_template_amplitudes(tl)[1, -3:].copy())
        if not np.isfinite(ddata.chisq):
            print("non-finite chi^2 at iteration", it)
            break
        if it in marks:
            tr.append((it, round(ddata.chisq, 4)))
    print(fluct, "nside", nside, tr, "non-SPD blocks:", nbad)
    if ta_trace:
        t = np.array(ta_trace[len(ta_trace) // 4:])
        print("  template amplitudes over the last three quarters: mean", np.round(t.mean(0), 4), "std", np.round(t.std(0), 4),
              "(injected 2.0, -1.5, 0.7); last Schur residual bound %.1e, refinements %d" % (eng.schur_info()[0][0], eng.schur_info()[1]))
    m = (ddata.masks[0] != 0).cpu().numpy()
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if c.sample_index[j]:
                pl = 0 if c.pol_flag[j][0] == 1 else 1
                x = eng.get_indices(l)[j, pl][m]
                print("  %-8s %-5s mean %9.4f std %8.4f  min %9.4f max %9.4f (prior %g +- %g, bounds %s)" % (
                    c.label, c.ind_label[j], x.mean(), x.std(), x.min(), x.max(), c.gauss_prior[j][0], c.gauss_prior[j][1], c.uni_prior[j]))
