#!/bin/bash
# SQ instruction-mix / stall counters of the bench kernel alone (diagnostic; bench.py --no-extras so that only the timed kernel runs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-mfma}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_1 -o sq -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 > $R/gpurun_out/pmc_${TAG}_1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_2 -o sq -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 > $R/gpurun_out/pmc_${TAG}_2.log 2>&1
python3 - <<PY
import csv,collections,glob
for d in ("pmc_${TAG}_1","pmc_${TAG}_2"):
    acc=collections.defaultdict(list)
    try:
        f=glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv"%d, recursive=True)[0]
        for r in csv.DictReader(open(f)):
            if "lqr_" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    except Exception as e: print(d, "ERR", e); continue
    for k,v in acc.items(): print(f"{k:28s} mean/dispatch = {sum(v)/len(v):.5g}  (n={len(v)})")
PY
