import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
import lexls_amd as hip
from lexls_amd import problems as P
for n, dims in ((7, [6, 3, 7]), (15, [6, 6, 6]), (24, [8, 8, 8, 6]), (31, [12, 12, 12])):
    for batch in (4096, 16384):
        lod = P.lse_batch_fast(5000 + n, batch, n, dims)
        for pol in (0, 2):
            s = hip.BatchedLexLSE(batch, n, dims); s.set_kernel_policy(pol); s.setProblem(lod)
            s.factorize_solve(keep_factor=False)
            best = 1e9
            for rep in range(3):
                s.synchronize(); t0 = time.perf_counter()
                for _ in range(50): s.factorize_solve(keep_factor=False)
                s.synchronize(); best = min(best, (time.perf_counter() - t0) / 50)
            print(f"n={n} dims={dims} batch={batch} {s.last_kernel()}: {best*1e6:.1f} us  {batch/best:.3e} fact/s", flush=True)
            s.close()
