#!/bin/bash
# Build a variant of libdangx.so with extra flags on EVERY translation unit (same-box A/B timing, bisecting a numerical
# difference):  tools/build_full_variant.sh <name> [flags]...  ->  dang_amd/lib/libdangx_<name>.so
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
obj=$R/dang_amd/lib/obj; v=$obj/variant_$name; mkdir -p $v
python3 - "$R" <<'PY' > $v/units.txt
import sys; sys.path.insert(0, sys.argv[1])
from dang_amd import _build
for n, f, d in _build.UNITS: print(n, f, " ".join(d))
PY
while read n f d; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -I$R/include -I$obj $d "$@" -c -o $v/$n.o $R/dang_amd/csrc/$f 2> $v/$n.log &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.5; done
done < $v/units.txt
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dang_amd/lib/libdangx_$name.so $v/*.o -lhiprtc
echo built dang_amd/lib/libdangx_$name.so
