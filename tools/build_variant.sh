#!/bin/bash
# Build a variant of libdangx.so in which ONE translation unit is compiled with extra flags (same-box A/B timing with
# tools/ab_bench.sh):  tools/build_variant.sh <name> <unit> [flags]...  ->  dang_amd/lib/libdangx_<name>.so
# e.g. tools/build_variant.sh magic dangx_planeset -DDX_EXP_MAGIC
set -e
name=$1; unit=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
obj=$R/dang_amd/lib/obj
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -I$R/include -I$obj "$@" -Rpass-analysis=kernel-resource-usage \
    -c -o $obj/${unit}__$name.o $R/dang_amd/csrc/$unit.hip 2> $obj/${unit}__$name.log
others=$(ls $obj/*.o | grep -v "__" | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dang_amd/lib/libdangx_$name.so $others $obj/${unit}__$name.o -lhiprtc
grep -hE "Function Name|VGPRs:|VGPRs Spill|Occupancy" $obj/${unit}__$name.log | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' | paste - - - - | head -12
