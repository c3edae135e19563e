#!/bin/bash
cd "$(dirname "$0")/.."
for b in 1024 2048 3072 4096 6144 8192 16384; do
  python bench.py --no-cpu-baseline --steps 50 --warmup 5 --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch', $b, 'ms/step=%.4f'%d['ms_per_step'], 'Mfact/s=%.1f'%(d['value']/1e6))"
done
