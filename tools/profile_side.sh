#!/bin/bash
# The side numbers of a round in ONE gpurun call (each a bench.py line, no profiler):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile_side.sh r03'
# -> gpurun_out/prof_<tag>_side/<tag>_{z_nsample100,c5z_nsample100,z_shard_of_8[_equal],z_calibrated,z_same_box,z_nbands9,c1,c2}.json  (copy into profiles/)
set -e -o pipefail
tag=${1:-rXX}
R=$(pwd); out=$R/gpurun_out/prof_${tag}_side; mkdir -p "$out"
B="python3 $R/bench.py --no-cpu-baseline --no-fortran-seam"
$B --nsample 100 --steps 5 --warmup 2            > "$out/${tag}_z_nsample100.json"   2> "$out/err.log"
$B --config C5 --nsample 100 --steps 2 --warmup 1 > "$out/${tag}_c5z_nsample100.json" 2>> "$out/err.log"
$B --shard-of 8 --steps 50 --warmup 5             > "$out/${tag}_z_shard_of_8.json"   2>> "$out/err.log"
$B --shard-of 8 --equal-shards --steps 50 --warmup 5 > "$out/${tag}_z_shard_of_8_equal.json" 2>> "$out/err.log"
$B --calibrated --steps 20 --warmup 3             > "$out/${tag}_z_calibrated.json"   2>> "$out/err.log"
$B --steps 20 --warmup 3                          > "$out/${tag}_z_same_box.json"     2>> "$out/err.log"
$B --nbands 9 --steps 10 --warmup 3               > "$out/${tag}_z_nbands9.json"      2>> "$out/err.log"
$B --config C1 --steps 200 --warmup 20            > "$out/${tag}_c1.json"             2>> "$out/err.log"
$B --config C2 --steps 100 --warmup 10            > "$out/${tag}_c2.json"             2>> "$out/err.log"
# the big ones (176 GB resident; bandpass-integrated bands): pass "big" as the second argument
if [ "${2:-}" = big ]; then
  $B --nside 4096 --steps 3 --warmup 1   > "$out/${tag}_z_nside4096.json"   2>> "$out/err.log"
  $B --bandpass 16 --steps 5 --warmup 2  > "$out/${tag}_z_bandpass16.json" 2>> "$out/err.log"
fi
for f in "$out"/*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); print('$(basename $f)', round(d['value'],3), d['unit'], round(d['ms_per_step'],3), 'ms')"; done
