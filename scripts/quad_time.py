"""Wall time of the 4096-problem IK batch on the policy-4 kernel (for A/B builds via LEXLS_HIP_LIB)."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import lexls_amd as hip
from lexls_amd import problems as P
n, dims, batch = 40, [12] * 5, 4096
lod = P.lse_batch_fast(20260100, batch, n, dims)
s = hip.BatchedLexLSE(batch, n, dims)
s.set_kernel_policy(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
s.setProblem(lod)
s.factorize_solve(keep_factor=False)
x0 = s.get_x().copy()
best = 1e9
for rep in range(5):
    s.synchronize(); t0 = time.perf_counter()
    for _ in range(100): s.factorize_solve(keep_factor=False)
    s.synchronize(); best = min(best, (time.perf_counter() - t0) / 100)
print(f"{best*1e6:.1f} us {batch/best:.3e} fact/s kernel {s.last_kernel()} checksum {float(np.abs(x0).sum()):.12e}")
