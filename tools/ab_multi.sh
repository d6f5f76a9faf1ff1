#!/bin/bash
# usage: ab_multi.sh rounds lib1 lib2 ...
R=$1; shift
for r in $(seq 1 $R); do
  for L in "$@"; do
    DANGX_LIB=$PWD/dang_amd/lib/$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$L', 'it/s=%.2f'%d['value'], ' '.join('%s=%.3f'%(n.replace('k_',''),v['avg_ms']) for n,v in k.items()))"
  done
done
