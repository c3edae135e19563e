#!/bin/bash
# Tuning helper: builds A/B variants of the wave kernel into lexls_amd/csrc/variants/lib_<name>.so
# usage: scripts/build_variants.sh "name1:-DFLAG1 -DFLAG2" "name2:..."
set -e
cd "$(dirname "$0")/../lexls_amd/csrc"
mkdir -p variants
make -s lexls_capi.o lqr_generic.o
BASE="-O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -DLEXLS_WAVE_TUNING_BUILD"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( hipcc $BASE $flags -Rpass-analysis=kernel-resource-usage -c lqr_small.hip -o /tmp/lqr_small_$name.o 2>&1 | grep -E "error|VGPRs:|ScratchSize|Occupancy" | sed "s/^.*remark: */[$name] /" ;
    hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$name.so lexls_capi.o lqr_generic.o /tmp/lqr_small_$name.o ) &
done
wait
ls variants/
