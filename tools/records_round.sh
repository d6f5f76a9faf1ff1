set -e
B="python bench.py --no-cpu-baseline --no-fortran-seam"
o=gpurun_out/r4_records; mkdir -p $o
$B --shard-of 8 --steps 50 --warmup 5 > $o/r04_z_shard_of_8.json 2>/dev/null
$B --shard-of 8 --equal-shards --steps 50 --warmup 5 > $o/r04_z_shard_of_8_equal.json 2>/dev/null
$B --nsample 100 --steps 5 --warmup 2 > $o/r04_z_nsample100.json 2>/dev/null
$B --calibrated --steps 20 --warmup 3 > $o/r04_z_calibrated.json 2>/dev/null
$B --bandpass 16 --steps 5 --warmup 2 > $o/r04_z_bandpass16.json 2>/dev/null
$B --nbands 9 --steps 20 --warmup 3 > $o/r04_z_nbands9.json 2>/dev/null
$B --config C1 --steps 200 --warmup 20 > $o/r04_c1.json 2>/dev/null
$B --config C2 --steps 100 --warmup 10 > $o/r04_c2.json 2>/dev/null
$B --config C5 --nsample 100 --steps 2 --warmup 1 > $o/r04_c5z_nsample100.json 2>/dev/null
$B --nside 4096 --steps 3 --warmup 1 > $o/r04_z_nside4096.json 2>/dev/null
for f in $o/*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],3), {k:v['avg_ms'] for k,v in d['kernels'].items()})"; done
