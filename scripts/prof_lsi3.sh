#!/bin/bash
# kernel trace of the lock-step LexLSI bench (configs[4]); usage: scripts/prof_lsi3.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lsi3 -o lsi -- python3 $R/bench.py --workload lsi --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $R/gpurun_out/prof_lsi3.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_lsi3/lsi_kernel_stats.csv")))
for r in rows[:10]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} total_ms={float(r['TotalDurationNs'])/1e6:8.3f} pct={r['Percentage']}")
t=list(csv.DictReader(open("$R/gpurun_out/prof_lsi3/lsi_kernel_trace.csv")))
t.sort(key=lambda r:int(r['Start_Timestamp']))
# a steady-state window: 12 kernels from the middle of the last run
mid=len(t)-200
prev=int(t[mid]['End_Timestamp'])
for r in t[mid+1:mid+14]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(f"  gap {(s-prev)/1e3:6.1f} us  dur {(e-s)/1e3:7.1f} us  {r['Kernel_Name'][:60]}")
    prev=e
PY
