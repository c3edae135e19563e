#!/bin/bash
# kernel trace of the humanoid-shaped lock-step LSI batch; usage: scripts/prof_lsi_wide.sh [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02_lsi_wide}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o w -- python3 $R/scripts/time_lsi_wide.py 256 > $R/gpurun_out/prof_$TAG.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_$TAG/w_kernel_stats.csv")))
for r in rows[:10]: print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
