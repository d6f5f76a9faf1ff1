#!/usr/bin/env python3
"""Host -> HBM hand-over cost of the static maps at the C3 size through dangx_upload_data (what a driver with
`cg_swap` pays per iteration, and every driver once)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dang_amd as da
from dang_amd import synth
nside = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dpar, ddata, bands, comps, meta = synth.make_sky("C3", nside=nside)
t0 = time.perf_counter()
eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
eng.synchronize()
print("initialize (create + upload of everything): %.2f s" % (time.perf_counter() - t0))
s, r, m = (np.ascontiguousarray(a) for a in (ddata.sig_map, ddata.rms_map, ddata.masks))
gb = (s.nbytes + r.nbytes + m.nbytes) / 1e9
for rep in range(3):
    t0 = time.perf_counter()
    eng._chk(eng.lib.dangx_upload_data(eng.h, s.ctypes.data, r.ctypes.data, m.ctypes.data))
    eng.synchronize()
    dt = time.perf_counter() - t0
    print("dangx_upload_data: %.2f GB in %.3f s = %.1f GB/s" % (gb, dt, gb / dt))
