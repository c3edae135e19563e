"""Prefix reuse: time of one factor-keeping factorization of B IK-sized problems (n = 40, 5 levels x 12 rows) when every problem reads its
first K levels back (K = 0: everything factorized).  Usage: python scripts/time_reuse.py [B]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lexls_amd as hip
from lexls_amd import problems as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n, dims = 40, [12] * 5
lod = P.lse_batch_fast(77, B, n, dims)
s = hip.BatchedLexLSE(B, n, dims); s.set_kernel_policy(2); s.set_prefix_reuse(True); s.setProblem(lod)
s.factorize_solve(True); s.synchronize()
print("kernel", s.last_kernel(), "batch", B)
base = None
for K in range(6):
    lv = np.full(B, K, np.int32)
    best = 1e9
    for rep in range(30):
        s.set_resume_levels(lv); s.synchronize()
        t0 = time.perf_counter(); s.factorize_solve(True); s.synchronize(); best = min(best, time.perf_counter() - t0)
    base = base or best
    print(f"levels read back: {K}   {best * 1e6:7.1f} us   ({best / base:.2f} of a full factorization)")
