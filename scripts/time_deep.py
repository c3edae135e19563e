"""Deep hierarchies (more than 64 rows in all): default dispatch (left-looking kernels) against the generic kernel.  usage: python scripts/time_deep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lexls_amd
from lexls_amd import problems as P
for (n, dims, batch) in [(40, [12] * 8, 4096), (40, [12] * 6, 4096), (30, [10] * 8, 4096), (40, [12] * 8, 512)]:
    lod = P.lse_batch_fast(7, batch, n, dims) if hasattr(P, "lse_batch_fast") else P.lse_batch(7, batch, n, dims)
    for keep in (False, True):
        for pol in (0, 1):
            s = lexls_amd.BatchedLexLSE(batch, n, dims)
            s.set_kernel_policy(pol)
            s.setProblem(lod)
            for _ in range(3):
                s.factorize_solve(keep)
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                s.factorize_solve(keep)
            s.synchronize()
            dt = (time.perf_counter() - t0) / 10
            print(f"n={n} dims={len(dims)}x{dims[0]} batch={batch} keep={keep} policy={pol} {s.last_kernel():34s} {dt * 1e6:8.1f} us {batch / dt:10.3e} fact/s")
            s.close()
