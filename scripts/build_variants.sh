#!/bin/bash
# Tuning helper: builds A/B variants of the exact-shape wave kernel into lexls_amd/csrc/variants/lib_<name>.so
# usage: [TU=lqr_lwave_41x12e_x] scripts/build_variants.sh "name1:-DFLAG1 -DFLAG2" "name2:..."   (TU = translation unit to rebuild)
set -e
cd "$(dirname "$0")/../lexls_amd/csrc"
mkdir -p variants
make -s -j8 >/dev/null
TU=${TU:-lqr_lwave_41x12e_x}
BASE="-O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -I../../include"
OTHERS=$(ls *.o | grep -v $TU.o)
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( hipcc $BASE $flags -Rpass-analysis=kernel-resource-usage -c $TU.hip -o /tmp/v_$name.o 2>&1 | grep -E "error|VGPRs:|ScratchSize|Occupancy" | sed "s/^.*remark: */[$name] /" ;
    hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$name.so $OTHERS /tmp/v_$name.o ) &
done
wait
ls variants/
