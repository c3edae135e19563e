#!/bin/bash
# Round-4 evidence.  TAG r04: bench.py's default command (lqr_qtol, x-only, 4 rotating resident batches; the side measurements of the same command —
# factor kept, configs[1], configs[4] — appear in the kernel table too), as profiles/collect_r03.sh did.  TAG r04m: the same command with the
# matrix-core kernel of this round (LEXLS_KERNEL_POLICY=7: lqr_mfma, two problems per wavefront) — kernel stats, HBM traffic, SQ counters.
# Every pass is its own rocprofv3 run (counter slots; --pmc is never combined with the trace domains gpurun refuses); the program itself follows `--`.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
passes() { # tag, extra bench flags for the stats pass
  local TAG=$1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_$TAG -o stats -- python3 $R/bench.py --no-cpu-baseline $2 > $R/gpurun_out/prof_stats_$TAG.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch_$TAG -o fetch -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 4 > $R/gpurun_out/prof_fetch_$TAG.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write_$TAG -o write -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 4 > $R/gpurun_out/prof_write_$TAG.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq1_$TAG -o sq -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 4 > $R/gpurun_out/prof_sq1_$TAG.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq2_$TAG -o sq -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 4 > $R/gpurun_out/prof_sq2_$TAG.log 2>&1
}
passes r04 ""
LEXLS_KERNEL_POLICY=7 passes r04m "--no-extras"
for m in 4 2 0; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_cal_r04_m$m -o cal -- $R/scripts/ubench/loadpat $m 6 > $R/gpurun_out/prof_cal_r04_m$m.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_large_r04 -o large -- python3 $R/scripts/time_large.py > $R/gpurun_out/prof_large_r04.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lsi_r04 -o lsi -- python3 $R/bench.py --workload lsi --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $R/gpurun_out/prof_lsi_r04.log 2>&1
python3 $R/bench.py > $R/gpurun_out/bench_r04.json 2> $R/gpurun_out/bench_r04.err
python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/bench_r04_driver_flags.json 2>> $R/gpurun_out/bench_r04.err
LEXLS_KERNEL_POLICY=7 python3 $R/bench.py --no-extras --no-cpu-baseline > $R/gpurun_out/bench_r04m.json 2>> $R/gpurun_out/bench_r04.err
cat $R/gpurun_out/bench_r04.json
