#!/bin/bash
# A/B builds: recompiles the named translation units of lexls_amd/csrc with extra flags and links them with the standard objects
# into lexls_amd/csrc/variants/lib_<name>.so (load with LEXLS_HIP_LIB).  usage: scripts/build_variant.sh name "-DFLAG ..." file.hip [file.hip ...]
set -e
NAME=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/../lexls_amd/csrc"
HOSTFMA=$(make -s -f Makefile --eval 'print-hostfma: ; @echo $(HOSTFMA)' print-hostfma)
mkdir -p variants /tmp/variant_$NAME
OBJS=""
for f in *.hip; do
  o=${f%.hip}.o
  if [[ " $* " == *" $f "* ]]; then
    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -std=c++17 -Wall -Wno-unused-function -I../../include $HOSTFMA $FLAGS -c $f -o /tmp/variant_$NAME/$o
    OBJS="$OBJS /tmp/variant_$NAME/$o"
  else
    OBJS="$OBJS $o"
  fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$NAME.so $OBJS
echo built variants/lib_$NAME.so
