#!/bin/bash
# instruction-cache counters of the bench kernel under the current kernel policy (diagnostic)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ic}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -o ic -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 > $R/gpurun_out/pmc_$TAG.log 2>&1
python3 - <<PY
import csv,collections,glob
acc=collections.defaultdict(list)
f=glob.glob("$R/gpurun_out/pmc_$TAG/**/*counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "lqr_" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(f"$TAG {k:28s} mean/dispatch = {sum(v)/len(v):.5g}  (n={len(v)})")
PY
