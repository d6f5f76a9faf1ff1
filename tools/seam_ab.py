"""Same-device A/B of the Fortran wrapper's loop under two environments:
    python tools/seam_ab.py CONFIG MODE "VAR=a" "VAR=b" [reps]      e.g.  python tools/seam_ab.py C5 twocall DANGX_PLANESET=1 DANGX_PLANESET=0
The Nside-8 problem of CONFIG is enlarged to the C3 pixel count (12.6 M) whatever the configuration."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dang_amd import fdrive, synth  # noqa: E402

config, mode, envs = sys.argv[1], sys.argv[2], sys.argv[3:5]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
dpar, ddata, bands, comps, meta = synth.make_sky(config, nside=8, nsample=10)
tile = 12 * 1024 ** 2 // meta["npix_global"]
with tempfile.TemporaryDirectory() as tmp:
    fin = os.path.join(tmp, "in.bin")
    fdrive.write_problem(fin, dpar, ddata, comps, meta, niter=10)
    for rep in range(reps + 1):      # the first round fills the kernel cache
        for e in envs:
            k, v = e.split("=", 1)
            os.environ[k] = v
            txt = fdrive.run(fin, os.path.join(tmp, "out.bin"), nctx=1, mode=mode, tile=tile)
            line = [l for l in txt.splitlines() if l.startswith("drive seconds")][0]
            if rep:
                print(config, mode, e, line)
