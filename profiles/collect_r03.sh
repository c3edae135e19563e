#!/bin/bash
# Round-3 evidence for bench.py (default command: lqr_qtol, x-only, 4 rotating resident batches; the side measurements of the same command —
# factor kept, configs[1], configs[4] — appear in the kernel table too).
#   1. kernel trace + stats of `python3 bench.py`   2. FETCH_SIZE pass   3. WRITE_SIZE pass   4./5. SQ instruction-mix and stall counters
#   7./8. kernel traces of the large path (+ its matrix-core counters) and of the lock-step LexLSI batch
#   6. calibration of FETCH_SIZE on KNOWN byte counts in the kernel's own access patterns (scripts/ubench/loadpat: modes 4 contiguous, 2 lane = column,
#      0 48-byte pieces; MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern")
# Every pass is its own rocprofv3 run (counter slots; --pmc is never combined with the trace domains gpurun refuses); the program itself follows `--`.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_$TAG -o stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_stats_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch_$TAG -o fetch -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 4 > $R/gpurun_out/prof_fetch_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write_$TAG -o write -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 4 > $R/gpurun_out/prof_write_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq1_$TAG -o sq -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 4 > $R/gpurun_out/prof_sq1_$TAG.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq2_$TAG -o sq -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 4 > $R/gpurun_out/prof_sq2_$TAG.log 2>&1
for m in 4 2 0; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_cal_${TAG}_m$m -o cal -- $R/scripts/ubench/loadpat $m 6 > $R/gpurun_out/prof_cal_${TAG}_m$m.log 2>&1
done
# 7. the large path (configs[1]): kernel trace + the matrix-core counters of its trailing update (large_gemm_mfma)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_large_$TAG -o large -- python3 $R/scripts/time_large.py > $R/gpurun_out/prof_large_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/prof_large_mfma_$TAG -o mfma -- python3 $R/scripts/time_large.py > $R/gpurun_out/prof_large_mfma_$TAG.log 2>&1
# 8. the lock-step LexLSI batch (configs[4])
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lsi_$TAG -o lsi -- python3 $R/bench.py --workload lsi --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $R/gpurun_out/prof_lsi_$TAG.log 2>&1
python3 $R/bench.py > $R/gpurun_out/bench_$TAG.json 2> $R/gpurun_out/bench_$TAG.err
python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/bench_${TAG}_driver_flags.json 2>> $R/gpurun_out/bench_$TAG.err
cat $R/gpurun_out/bench_$TAG.json
