#!/bin/bash
# kernel trace of the large path (configs[1]); usage: scripts/prof_large2.sh [policy]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_large2 -o lg -- python3 $R/scripts/large_check.py ${1:-0} > $R/gpurun_out/prof_large2.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_large2/lg_kernel_stats.csv")))
for r in rows[:14]: print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
