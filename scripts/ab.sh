#!/bin/bash
# A/B of kernel builds under lexls_amd/csrc/variants: alternates the variants N times, prints the minimum of each. usage: scripts/ab.sh N v1 v2 ...
N=$1; shift
for v in "$@"; do echo -n > /tmp/ab_$v.txt; done
for i in $(seq $N); do for v in "$@"; do LEXLS_HIP_LIB=lexls_amd/csrc/variants/lib_$v.so python scripts/quad_time.py | awk '{print $1}' >> /tmp/ab_$v.txt; done; done
for v in "$@"; do echo "$v: min $(sort -n /tmp/ab_$v.txt | head -1) us   all: $(tr '\n' ' ' < /tmp/ab_$v.txt)"; done
