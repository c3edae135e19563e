#!/bin/bash
# kernel trace of the large path's timing loop (configs[1]); usage: scripts/prof_large3.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_large3 -o lg -- python3 $R/scripts/time_large.py > $R/gpurun_out/prof_large3.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_large3/lg_kernel_stats.csv")))
tot=0
for r in rows[:16]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} total_ms={float(r['TotalDurationNs'])/1e6:8.3f} pct={r['Percentage']}")
PY
