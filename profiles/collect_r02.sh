#!/bin/bash
# Round-2 evidence for bench.py (default command: the four-problems-per-wavefront kernel, x-only, 4 rotating resident batches).
#   1. kernel trace + stats   2. FETCH_SIZE pass   3. WRITE_SIZE pass   4./5. SQ instruction-mix and stall counters (two passes, 8 SQ slots each)
# Every pass is its own rocprofv3 run (counter slots; and --pmc is never combined with the trace domains gpurun refuses).
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_$TAG -o stats -- $B --steps 200 --warmup 20 > $R/gpurun_out/prof_stats_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch_$TAG -o fetch -- $B --steps 40 --warmup 4 > $R/gpurun_out/prof_fetch_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write_$TAG -o write -- $B --steps 40 --warmup 4 > $R/gpurun_out/prof_write_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq1_$TAG -o sq -- $B --steps 40 --warmup 4 > $R/gpurun_out/prof_sq1_$TAG.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq2_$TAG -o sq -- $B --steps 40 --warmup 4 > $R/gpurun_out/prof_sq2_$TAG.log 2>&1
python3 $R/bench.py --steps 200 > $R/gpurun_out/bench_$TAG.json 2> $R/gpurun_out/bench_$TAG.err
cat $R/gpurun_out/bench_$TAG.json
