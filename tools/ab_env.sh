#!/bin/bash
# A/B timing of two kernel forms selected by an environment variable on ONE GPU (devices differ by several percent):
#   tools/ab_env.sh VAR valueA valueB [rounds] [bench args...]
V=$1; A=$2; B=$3; R=${4:-2}; shift 4
for r in $(seq 1 $R); do
  for X in "$A" "$B"; do
    env $V=$X python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$V=$X', 'it/s=%.2f'%d['value'], ' '.join('%s=%.3f'%(n.replace('k_',''),v['avg_ms']) for n,v in k.items()))"
  done
done
