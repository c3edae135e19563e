#!/bin/bash
# kernel trace of the lock-step LSI batch (configs[4]); usage: scripts/prof_lsi2.sh [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02_lsi}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o lsi -- python3 $R/scripts/bench_lsi.py 1024 > $R/gpurun_out/prof_$TAG.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_$TAG/lsi_kernel_stats.csv")))
for r in rows[:8]: print(f"{r['Name'][:80]:80s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
tail -1 $R/gpurun_out/prof_$TAG.log | cut -c1-600
