#!/bin/bash
# A/B of the lock-step LSI driver's switches on ONE box: worker-pool spin time x host/device step (cold, warm, warm_30 totals in ms)
cd "$(dirname "$0")/.."
B=${LSI_BATCH:-1024}
for rep in 1 2; do
for spin in 0 300; do
for hs in 0 1; do
  if [ $hs = 1 ]; then HS="LEXLS_X=0"; else HS="LEXLS_LSI_DEVICE_STEP=1"; fi
  env $HS LEXLS_POOL_SPIN_US=$spin LEXLS_LSI_TIMING=1 LSI_BATCH=$B timeout -k 10 300 python scripts/bench_configs.py 2>&1 >/dev/null | grep total | tail -3 | awk -v s=$spin -v h=$hs '{printf "spin=%s host_step=%s total=%s host=%s wait=%s enq=%s\n", s, h, $3, $(NF-13), $(NF-18), $(NF-24)}'
done; done; done
