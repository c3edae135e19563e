"""Regularized IK batches (n = 40, 5 x 12): time per factorize_solve by regularization type.  usage: python scripts/time_reg.py [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lexls_amd as hip
from lexls_amd import problems as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n, dims = 40, [12] * 5
lod = P.lse_batch_fast(11, B, n, dims)
for rt in (0, 1, 3, 4, 5, 8, 2):
    for keep in (True, False):
        s = hip.BatchedLexLSE(B, n, dims)
        if rt: s.setRegularization(rt, [0.01] * 5)
        s.setProblem(lod)
        for _ in range(3): s.factorize_solve(keep)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): s.factorize_solve(keep)
        s.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"regularization type {rt}  factor kept {keep!s:5s}  {s.last_kernel():28s} {dt * 1e6:9.1f} us  {B / dt:10.3e} problems/s", flush=True)
        s.close()
