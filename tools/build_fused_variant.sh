#!/bin/bash
# Build a variant of libdangx.so whose three fused-kernel translation units (dangx_fused.hip, -DDX_REG_MODE=1..3) get extra
# flags (same-box A/B timing):  tools/build_fused_variant.sh <name> [flags]...  ->  dang_amd/lib/libdangx_<name>.so
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
obj=$R/dang_amd/lib/obj
for m in 1 2 3; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -I$R/include -I$obj -DDX_REG_MODE=$m "$@" -Rpass-analysis=kernel-resource-usage \
      -c -o $obj/dangx_fused_m${m}__$name.o $R/dang_amd/csrc/dangx_fused.hip 2> $obj/dangx_fused_m${m}__$name.log &
done
wait
others=$(ls $obj/*.o | grep -v "__" | grep -v "/dangx_fused_m[1-3].o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dang_amd/lib/libdangx_$name.so $others $obj/dangx_fused_m[1-3]__$name.o -lhiprtc
grep -hE "Function Name|VGPRs:|VGPRs Spill|Occupancy" $obj/dangx_fused_m1__$name.log | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' | paste - - - - | grep -E "ELi10E" | head
