"""Timing of the generic kernel (kernel policy 1) on a few shapes and batch sizes: us per factorize_solve and factorizations/s.
usage: python scripts/time_generic.py   (LEXLS_HIP_LIB selects a variant build)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lexls_amd
from lexls_amd import problems as P
for (n, dims, batch, nfix) in [(40, [12] * 5, 4096, 0), (40, [12] * 5, 256, 0), (20, [6, 5, 5, 6], 4096, 0), (88, [3, 2, 97], 512, 30), (88, [3, 2, 97], 1, 30), (120, [40, 40, 40], 256, 0), (10, [4, 4, 4], 16384, 0)]:
    lod = P.lse_batch_fast(7, batch, n, dims) if hasattr(P, "lse_batch_fast") else P.lse_batch(7, batch, n, dims)
    s = lexls_amd.BatchedLexLSE(batch, n, dims)
    s.set_kernel_policy(1)
    if nfix:
        idx = np.zeros((batch, n), np.uint32); idx[:, :nfix] = np.arange(0, 2 * nfix, 2)
        s.fixVariables(np.full(batch, nfix, np.uint32), idx, np.zeros((batch, n)))
    s.setProblem(lod)
    for _ in range(3):
        s.factorize_solve(True)
    s.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        s.factorize_solve(True)
    s.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n:4d} dims={dims} batch={batch:6d} fixed={nfix:3d} {s.last_kernel():24s} {dt * 1e6:9.1f} us  {batch / dt:12.3e} fact/s")
    s.close()
