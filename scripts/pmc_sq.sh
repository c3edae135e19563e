#!/bin/bash
# SQ instruction-mix / stall counters of the bench kernel (diagnostic; separate from the kernel-trace run)
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=${1:-}
cd /tmp && export TMPDIR=/tmp
[ -n "$LIB" ] && export LEXLS_HIP_LIB=$R/$LIB
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -o sq -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq2 -o sq -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $R/gpurun_out/pmc_sq2.log 2>&1
python3 - <<PY
import csv,collections
for d in ("pmc_sq","pmc_sq2"):
    acc=collections.defaultdict(list)
    try:
        for r in csv.DictReader(open("$R/gpurun_out/%s/sq_counter_collection.csv"%d)):
            if "lqr_" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    except Exception as e: print(d, "ERR", e); continue
    for k,v in acc.items(): print(f"{k:28s} mean/dispatch = {sum(v)/len(v):.4g}  (n={len(v)})")
PY
