"""x-only solves of 4096 problems with uniform levels of another size: tolerance-contract kernel (policy 0) against the bit-exact one (policy 4).
usage: python scripts/time_dims.py [rows_per_level] [levels] [n]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lexls_amd as hip
from lexls_amd import problems as P
md = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nobj = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
B, dims = 4096, [md] * nobj
lod = P.lse_batch_fast(77, B, n, dims)
for pol in (0, 4):
    s = hip.BatchedLexLSE(B, n, dims); s.set_kernel_policy(pol); s.setProblem(lod)
    for _ in range(5): s.factorize_solve(False)
    s.synchronize(); t0 = time.perf_counter()
    for _ in range(50): s.factorize_solve(False)
    s.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"dims {dims} n {n} policy {pol}: {s.last_kernel():28s} {dt*1e6:8.1f} us per {B}  {B/dt:.3e} fact/s")
