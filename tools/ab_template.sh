#!/bin/bash
# Same-box A/B of two builds on the template model: tools/ab_template.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-3}
for r in $(seq 1 $R); do
  for L in "$A" "$B"; do
    echo "$(basename $L): $(DANGX_LIB=$L python3 tools/bench_template_iter.py 1024 8 2>/dev/null | head -1 | cut -c1-70)"
  done
done
