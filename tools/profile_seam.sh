#!/bin/bash
# Kernel trace of the reference-side Fortran wrapper's loop (fortran/reference_side/dang_gpu_drive, C3 tiled to Nside 1024):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/profile_seam.sh r03_seam [fused|twocall]'
# leaves gpurun_out/prof_<tag>/<tag>_<mode>_kernel_stats.csv and the driver's own timing line.
set -e -o pipefail
tag=${1:-rXX}; mode=${2:-fused}
R=$(pwd); out=$R/gpurun_out/prof_$tag; mkdir -p "$out"
export TMPDIR=/tmp
python3 - "$out/in.bin" <<'PY'
import sys
sys.path.insert(0, ".")
from dang_amd import fdrive, synth, _build
_build.build_reference_drive()
dpar, ddata, bands, comps, meta = synth.make_sky("C3", nside=8, nsample=10)
fdrive.write_problem(sys.argv[1], dpar, ddata, comps, meta, niter=13)
print("tile", 12 * synth.CONFIGS["C3"]["nside"] ** 2 // meta["npix_global"])
PY
exe=$(python3 -c "import sys; sys.path.insert(0,'.'); from dang_amd import _build; print(_build.build_reference_drive())")
"$exe" "$out/in.bin" "$out/out.bin" 1 "$mode" 16384 > "$out/${tag}_${mode}_plain.log" 2>&1
grep "drive seconds" "$out/${tag}_${mode}_plain.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- "$exe" "$out/in.bin" "$out/out.bin" 1 "$mode" 16384 > "$out/${tag}_${mode}_traced.log" 2> "$out/trace.err"
grep "drive seconds" "$out/${tag}_${mode}_traced.log"
cp "$(find "$out/trace" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_${mode}_kernel_stats.csv"
python3 - "$(find "$out/trace" -name '*kernel_trace.csv' | head -1)" > "$out/${tag}_${mode}_timeline.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
last_end = None
# the last two iterations' worth of launches with the idle gap before each
for r in rows[-40:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end) / 1e3 if last_end else 0.0
    print("%10.1f us  gap %8.1f us  dur %9.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, r["Kernel_Name"][:90]))
    last_end = e
PY
rm -rf "$out/trace" "$out/in.bin" "$out/out.bin"
cat "$out/${tag}_${mode}_timeline.txt" | tail -40
