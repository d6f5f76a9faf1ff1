#!/bin/bash
# A/B timing of two builds of libdangx.so on ONE GPU (devices differ by several percent, so builds must
# never be ranked across boxes): usage tools/ab_bench.sh libA.so libB.so [rounds] [bench args...]
A=$1; B=$2; R=${3:-3}; shift 3
for r in $(seq 1 $R); do
  for L in "$A" "$B"; do
    DANGX_LIB=$L python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fortran-seam "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$L'.split('/')[-1], 'it/s=%.2f'%d['value'], ' '.join('%s=%.3f'%(n.replace('k_',''),v['avg_ms']) for n,v in k.items()))"
  done
done
