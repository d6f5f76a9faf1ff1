#!/bin/bash
# Build a variant of libdangx.so whose five register-chain translation units (dangx_mhreg.hip, -DDX_REG_MODE=1..5) get
# extra -D flags (same-box A/B timing):  tools/build_mh_variant.sh <name> [-DFLAG]...  ->  dang_amd/lib/libdangx_<name>.so
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
obj=$R/dang_amd/lib/obj
for m in 1 2 3 4 5; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -I$R/include -DDX_REG_MODE=$m "$@" -Rpass-analysis=kernel-resource-usage \
      -c -o $obj/dangx_mhreg_m${m}__$name.o $R/dang_amd/csrc/dangx_mhreg.hip 2> $obj/dangx_mhreg_m${m}__$name.log &
done
wait
others=$(ls $obj/*.o | grep -v "__" | grep -v "/dangx_mhreg_m[1-5].o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dang_amd/lib/libdangx_$name.so $others $obj/dangx_mhreg_m[1-5]__$name.o
grep -hE "Function Name|VGPRs:|VGPRs Spill|Occupancy" $obj/dangx_mhreg_m[123]__$name.log | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' | paste - - - - | grep -E "ELi10E|ELi20E" | head -40
