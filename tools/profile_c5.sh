#!/bin/bash
# C5 (Nside 2048, 20 bands, 6 components, IQU) on ONE GPU: bench line + rocprofv3 kernel trace + SQ counters.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile_c5.sh r02_c5'
set -e -o pipefail
tag=${1:-rXX_c5}
R=$(pwd)
out=$R/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 $R/bench.py --config C5 --no-cpu-baseline --no-fortran-seam"
echo "[c5] plain bench"
python3 $R/bench.py --config C5 --no-cpu-baseline --no-fortran-seam --steps 5 --warmup 2 > "$out/${tag}_bench.json" 2> "$out/bench.err"
echo "[c5] kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- $B --steps 5 --warmup 2 > "$out/${tag}_bench_under_rocprof.json" 2> "$out/trace.err"
echo "[c5] SQ counters"
rocprofv3 --kernel-include-regex "k_(amp|index|plane|schur|sky|cg|Ax|rhs|reduce)" --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES \
    -d "$out/sq" -o run -- $B --steps 1 --warmup 1 > /dev/null 2> "$out/sq.err"
echo "[c5] FETCH_SIZE"
rocprofv3 --kernel-include-regex "k_(amp|index|plane|schur|sky|cg|Ax|rhs|reduce)" --output-format csv --pmc FETCH_SIZE -d "$out/pmc_fetch" -o run -- $B --steps 1 --warmup 1 > /dev/null 2> "$out/fetch.err"
echo "[c5] WRITE_SIZE"
rocprofv3 --kernel-include-regex "k_(amp|index|plane|schur|sky|cg|Ax|rhs|reduce)" --output-format csv --pmc WRITE_SIZE -d "$out/pmc_write" -o run -- $B --steps 1 --warmup 1 > /dev/null 2> "$out/write.err"
python3 $R/tools/prof_summary.py --stats "$out/trace" --pmc FETCH_SIZE="$out/pmc_fetch" --pmc WRITE_SIZE="$out/pmc_write" --sq "$out/sq" \
    --config C5 --traffic-json "$out/${tag}_traffic.json" --valu-json "$out/${tag}_valu.json" -o "$out/${tag}_profile.md" \
    --title "$tag: python3 bench.py --config C5 --steps 5 --warmup 2 (Nside 2048, 20 bands, 6 components, IQU, 1x MI355X); PMC passes: --steps 1 --warmup 1" > /dev/null
cp "$(find "$out/trace" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
rm -rf "$out/trace" "$out/sq" "$out/pmc_fetch" "$out/pmc_write"
echo "[c5] done: $(ls $out | tr '\n' ' ')"
