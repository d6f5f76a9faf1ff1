import sys, numpy as np
sys.path.insert(0, "/root/repo")
import dang_amd as da
from dang_amd import synth
for fluct in ("reference", "correct"):
    dpar, ddata, bands, comps, meta = synth.make_sky("C3", nside=64, fluct_mode=fluct)
    eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
    tr = []
    for it in range(1, 401):
        da.sample_cg_groups(dpar, ddata, it=it, defer_chisq=(it > 1))
        if it > 1:
            da.sample_spectral_parameters(dpar, ddata, it=it)
        if it in (1, 2, 5, 10, 20, 50, 100, 200, 300, 400):
            tr.append((it, round(ddata.chisq, 4)))
    print(fluct, tr)
    # index recovery: mean offset of beta_s from truth in units of the prior width
    bs = eng.get_indices(1)[0, 0]
    m = np.asarray(ddata.masks)[0] != 0
    print("  synch beta T: mean %.4f std %.4f (truth mean -3.1, prior sigma 0.1)" % (bs[m].mean(), bs[m].std()))
