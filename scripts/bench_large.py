#!/usr/bin/env python3
"""configs[1] only (single large equality problem n=512, 4 x 256): time lexls_lse_factorize_solve on the device-resident problem.
  python scripts/bench_large.py [reps]        (under rocprofv3: `-- python3 scripts/bench_large.py 20`)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lexls_amd  # noqa: E402
from lexls_amd import problems as P  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, dims = 512, [256] * 4
lod = P.lse_batch(20260001, 1, n, dims)
s = lexls_amd.BatchedLexLSE(1, n, dims)
s.setProblem(lod)
s.factorize_solve(True)
s.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    s.factorize_solve(True)
s.synchronize()
t = (time.perf_counter() - t0) / reps
flops = P.flop_model(n, dims)["total"]
print(json.dumps(dict(kernel=s.last_kernel(), ms=1e3 * t, gflops=flops / t / 1e9, reps=reps)))
