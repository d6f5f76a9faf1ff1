#!/bin/bash
# Same-box A/B of two builds through the current bench line: tools/ab_lines.sh libA.so libB.so [rounds] [bench args...]
A=$1; B=$2; R=${3:-3}; shift 3
for r in $(seq 1 $R); do
  for L in "$A" "$B"; do
    DANGX_LIB=$L python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fortran-seam --no-template-model "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$L'.split('/')[-1], 'it/s=%.2f' % d['value'], ' '.join('%.3f' % v['avg_ms'] for v in d['roofline']['kernels'].values()))"
  done
done
