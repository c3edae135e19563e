#!/bin/bash
# kernel trace of BASELINE configs[0] (test_01.dat as one LexLSI problem); usage: scripts/prof_config0.sh [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02_config0}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o c0 -- python3 $R/scripts/time_config0.py 5 > $R/gpurun_out/prof_$TAG.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_$TAG/c0_kernel_stats.csv")))
for r in rows[:10]: print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={r['Percentage']}")
PY
tail -1 $R/gpurun_out/prof_$TAG.log | cut -c1-600
