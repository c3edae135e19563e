#!/bin/bash
# instruction-cache counters of the bench kernel (diagnostic): the left-looking kernel is ~80 KB of straight-line code
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ic -o ic -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/pmc_ic.log 2>&1
rocprofv3 --pmc SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQC_ICACHE_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ic2 -o ic -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/pmc_ic2.log 2>&1
python3 - <<PY
import csv,collections
for d in ("pmc_ic","pmc_ic2"):
    acc=collections.defaultdict(list)
    try:
        for r in csv.DictReader(open("$R/gpurun_out/%s/ic_counter_collection.csv"%d)):
            if "lqr_lwave" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    except Exception as e: print(d, "ERR", e); continue
    for k,v in acc.items(): print(f"{k:28s} mean/dispatch = {sum(v)/len(v):.4g}  (n={len(v)})")
PY
