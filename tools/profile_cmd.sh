#!/bin/bash
# The passes of tools/profile_variant.sh for ANY python script of the repo (not a bench.py line):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile_cmd.sh r04_template_iter tools/bench_template_iter.py 1024 5'
# leaves gpurun_out/prof_<tag>/<tag>{.txt, _kernel_stats.csv, _profile.md}: the script's own output, the rocprofv3 kernel trace of
# the same command and its SQ / FETCH_SIZE / WRITE_SIZE passes (separate --pmc passes, as MI355X_MICROARCH.md prescribes).
set -e -o pipefail
tag=$1; shift
R=$(pwd); out=$R/gpurun_out/prof_$tag; mkdir -p "$out"
export TMPDIR=/tmp
B="python3 $R/$*"
RE="k_(amp|index|plane|schur|sky|cg|Ax|rhs|reduce|fullsky|udgrade|chisq|coarse)"
echo "[$tag] plain run"; $B > "$out/$tag.txt" 2> "$out/run.err"
echo "[$tag] kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- $B > /dev/null 2> "$out/trace.err"
echo "[$tag] SQ counters"
rocprofv3 --kernel-include-regex "$RE" --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES \
    -d "$out/sq" -o run -- $B > /dev/null 2> "$out/sq.err"
echo "[$tag] FETCH_SIZE"
rocprofv3 --kernel-include-regex "$RE" --output-format csv --pmc FETCH_SIZE -d "$out/pmc_fetch" -o run -- $B > /dev/null 2> "$out/fetch.err"
echo "[$tag] WRITE_SIZE"
rocprofv3 --kernel-include-regex "$RE" --output-format csv --pmc WRITE_SIZE -d "$out/pmc_write" -o run -- $B > /dev/null 2> "$out/write.err"
python3 $R/tools/prof_summary.py --stats "$out/trace" --pmc FETCH_SIZE="$out/pmc_fetch" --pmc WRITE_SIZE="$out/pmc_write" --sq "$out/sq" \
    -o "$out/${tag}_profile.md" --title "$tag: python3 $* (1x MI355X); PMC passes: the same command" > /dev/null
cp "$(find "$out/trace" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
rm -rf "$out/trace" "$out/sq" "$out/pmc_fetch" "$out/pmc_write"
echo "[$tag] done: $(ls $out | tr '\n' ' ')"
