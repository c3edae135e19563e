#!/bin/bash
# runs bench.py once per variant library (kernel tuning A/B on ONE device, same process environment)
cd "$(dirname "$0")/.."
for lib in lexls_amd/csrc/variants/lib_*.so; do
  name=$(basename $lib .so)
  LEXLS_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'ms/step=%.4f'%d['ms_per_step'], 'Mfact/s=%.1f'%(d['value']/1e6), d['config']['kernel'])"
done
