#!/usr/bin/env python3
"""Per-stage timeline of a lock-step LSI run from a rocprofv3 kernel trace: for the LAST `stages` stages (the warm_30 run of
scripts/bench_lsi.py) the mean duration of each kernel and the mean idle time in front of it.
usage: python scripts/lsi_trace_gaps.py gpurun_out/prof_<tag>/lsi_kernel_trace.csv [stages]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1], newline="")))
stages = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
ends = [i for i, nm in enumerate(names) if "lsi_iterate_kernel" in nm]
if len(ends) < stages + 1:
    sys.exit("not enough lsi_iterate_kernel launches in the trace")
first = ends[-stages - 1] + 1
seg = rows[first:ends[-1] + 1]
acc = {}
prev_end = int(rows[first - 1]["End_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = next((t for t in ("lqr_wave_kernel", "sensitivity_sweep_kernel", "lsi_iterate_kernel", "gather_rows_kernel", "lqr_lwave_kernel", "copyBuffer", "fillBuffer") if t in r["Kernel_Name"]), r["Kernel_Name"][:40])
    a = acc.setdefault(nm, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2] += (s - prev_end) / 1e3
    prev_end = e
span = (int(seg[-1]["End_Timestamp"]) - int(rows[first - 1]["End_Timestamp"])) / 1e3
print(f"last {stages} stages: {span / stages:.1f} us per stage")
for nm, (c, d, g) in acc.items():
    print(f"  {nm:40s} calls/stage={c / stages:5.2f}  avg_us={d / c:8.2f}  idle_before_us={g / c:7.2f}")
