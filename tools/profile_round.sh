#!/bin/bash
# One round's profile evidence, collected on the GPU box in ONE gpurun call and summarised into profiles/.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile_round.sh r01_z'
# then, back in the container:  cp gpurun_out/prof_<tag>/<tag>_* profiles/   (gpurun_out/ is scratch)
# Passes (rocprofv3 refuses --pmc together with the trace domains; FETCH_SIZE and WRITE_SIZE in separate passes as
# MI355X_MICROARCH.md prescribes): kernel-trace --stats | FETCH_SIZE | WRITE_SIZE | SQ_* | calibration kernels.
set -e -o pipefail
trap 'for f in "$out"/*.err; do echo "== $f"; tail -3 "$f"; done' ERR
tag=${1:-rXX}
R=$(pwd)
out=$R/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-fortran-seam"
echo "[profile] plain bench (the numbers the JSON line reports)"; 
python3 $R/bench.py --steps 20 --warmup 3 > "$out/${tag}_bench.json" 2> "$out/bench.err"
echo "[profile] kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- $B --steps 20 --warmup 3 > "$out/${tag}_bench_under_rocprof.json" 2> "$out/trace.err"
echo "[profile] FETCH_SIZE"
rocprofv3 --kernel-include-regex "k_(amp|index|plane|schur|sky|cg|Ax|rhs|reduce)" --output-format csv --pmc FETCH_SIZE -d "$out/pmc_fetch" -o run -- $B --steps 2 --warmup 1 > /dev/null 2> "$out/fetch.err"
echo "[profile] WRITE_SIZE"
rocprofv3 --kernel-include-regex "k_(amp|index|plane|schur|sky|cg|Ax|rhs|reduce)" --output-format csv --pmc WRITE_SIZE -d "$out/pmc_write" -o run -- $B --steps 2 --warmup 1 > /dev/null 2> "$out/write.err"
echo "[profile] SQ counters"
rocprofv3 --kernel-include-regex "k_(amp|index|plane|schur|sky|cg|Ax|rhs|reduce)" --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES \
    -d "$out/sq" -o run -- $B --steps 2 --warmup 1 > /dev/null 2> "$out/sq.err"
echo "[profile] FETCH/WRITE calibration on kernels of known byte count"
rocprofv3 --kernel-include-regex "k_(amp|index|plane|schur|sky|cg|Ax|rhs|reduce)" --output-format csv --pmc FETCH_SIZE -d "$out/cal_fetch" -o run -- python3 $R/tools/pmc_calib.py > "$out/cal.log" 2> "$out/cal_fetch.err"
rocprofv3 --kernel-include-regex "k_(amp|index|plane|schur|sky|cg|Ax|rhs|reduce)" --output-format csv --pmc WRITE_SIZE -d "$out/cal_write" -o run -- python3 $R/tools/pmc_calib.py >> "$out/cal.log" 2> "$out/cal_write.err"
python3 $R/tools/prof_summary.py --stats "$out/trace" --pmc FETCH_SIZE="$out/pmc_fetch" --pmc WRITE_SIZE="$out/pmc_write" \
    --sq "$out/sq" --traffic-json "$out/${tag}_traffic.json" --valu-json "$out/${tag}_valu.json" -o "$out/${tag}_profile.md" \
    --title "$tag: python3 bench.py --steps 20 --warmup 3 (C3, 1x MI355X); PMC passes: --steps 2 --warmup 1" > /dev/null
python3 $R/tools/prof_summary.py --only k_cg_vec --pmc FETCH_SIZE="$out/cal_fetch" --pmc WRITE_SIZE="$out/cal_write" \
    -o "$out/${tag}_calibration.md" --title "$tag: FETCH_SIZE / WRITE_SIZE on k_cg_vec (8 B per lane, coalesced, known byte count)" > /dev/null
cp "$(find "$out/trace" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
rm -rf "$out/trace" "$out/pmc_fetch" "$out/pmc_write" "$out/sq" "$out/cal_fetch" "$out/cal_write"   # raw CSVs: tens of MB
echo "[profile] done: $(ls $out | tr '\n' ' ')"
