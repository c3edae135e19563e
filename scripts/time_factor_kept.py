import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lexls_amd as hip
from lexls_amd import problems as P
n, dims, batch = 40, [12] * 5, 4096
lods = [P.lse_batch_fast(20260100 + 7 * i, batch, n, dims) for i in range(4)]
ss = []
for lod in lods:
    s = hip.BatchedLexLSE(batch, n, dims); s.setProblem(lod); s.factorize_solve(keep_factor=True); ss.append(s)
best = 1e9
for rep in range(5):
    for s in ss: s.synchronize()
    t0 = time.perf_counter()
    for _ in range(25):
        for s in ss: s.factorize_solve(keep_factor=True)
    for s in ss: s.synchronize()
    best = min(best, (time.perf_counter() - t0) / 100)
print(f"factor kept: {best*1e6:.1f} us per 4096 kernel {ss[0].last_kernel()}")
