#!/bin/bash
# Build a variant of libdangx.so that differs in ONE translation unit's -D flags (same-box A/B timing):
#   tools/build_variant.sh <name> <unit.hip> [-DFLAG=..]...   ->  dang_amd/lib/libdangx_<name>.so
# (the other objects come from the regular build under dang_amd/lib/obj/)
set -e
name=$1; unit=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
obj=$R/dang_amd/lib/obj
base=$(basename "$unit" .hip)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -I$R/include "$@" -Rpass-analysis=kernel-resource-usage \
    -c -o $obj/${base}__$name.o $R/dang_amd/csrc/$unit 2> $obj/${base}__$name.log
others=$(ls $obj/*.o | grep -v "__" | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dang_amd/lib/libdangx_$name.so $others $obj/${base}__$name.o
grep -E "Function Name|VGPRs:|VGPRs Spill|Occupancy" $obj/${base}__$name.log | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' | paste - - - - | head -60
