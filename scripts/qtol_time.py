"""Wall time of the 4096-problem IK batch on the tolerance kernel (policy 6), four rotating batches, for A/B runs with environment knobs."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import lexls_amd as hip
from lexls_amd import problems as P
n, dims, batch = 40, [12] * 5, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lods = [P.lse_batch_fast(20260100 + 7 * i, batch, n, dims) for i in range(4)]
ss = []
for lod in lods:
    s = hip.BatchedLexLSE(batch, n, dims)
    s.set_kernel_policy(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
    s.setProblem(lod)
    s.factorize_solve(keep_factor=False)
    ss.append(s)
x0 = ss[0].get_x().copy()
best = 1e9
for rep in range(5):
    for s in ss: s.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        for s in ss: s.factorize_solve(keep_factor=False)
    for s in ss: s.synchronize()
    best = min(best, (time.perf_counter() - t0) / 200)
print(f"{best*1e6:.1f} us {batch/best:.3e} fact/s kernel {ss[0].last_kernel()} checksum {float(np.abs(x0).sum()):.12e}")
