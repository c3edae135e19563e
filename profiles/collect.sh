#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the bench command            -> gpurun_out/prof_stats
#   2. PMC pass FETCH_SIZE, 3. PMC pass WRITE_SIZE (separate passes: TCC slot budget, MI355X_MICROARCH.md)
# Summaries are then distilled into profiles/ by profiles/summarize.py.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
EXTRA=${2:-}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_$TAG -o stats -- python3 $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 $EXTRA > $R/gpurun_out/prof_stats_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch_$TAG -o fetch -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 2 $EXTRA > $R/gpurun_out/prof_fetch_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write_$TAG -o write -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 2 $EXTRA > $R/gpurun_out/prof_write_$TAG.log 2>&1
find $R/gpurun_out -name '*.csv' | head -30
