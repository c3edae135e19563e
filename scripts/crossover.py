"""x-only factorize+solve time of the IK shape by batch size: register-resident wave kernel (policy 2) vs four-per-wavefront kernel (policy 4)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lexls_amd as hip
from lexls_amd import problems as P
n, dims = 40, [12] * 5
lod_all = P.lse_batch_fast(20260100, 8192, n, dims)
for batch in (512, 1024, 1536, 2048, 3072, 4096, 8192):
    row = []
    for pol, keep in ((2, False), (4, False), (2, True), (4, True)):
        s = hip.BatchedLexLSE(batch, n, dims)
        s.set_kernel_policy(pol)
        s.setProblem(lod_all[:batch])
        s.factorize_solve(keep_factor=keep)
        best = 1e9
        for rep in range(3):
            s.synchronize(); t0 = time.perf_counter()
            for _ in range(50): s.factorize_solve(keep_factor=keep)
            s.synchronize(); best = min(best, (time.perf_counter() - t0) / 50)
        row.append(f"{s.last_kernel()}{'+factor' if keep else ''}: {best*1e6:.1f} us")
        s.close()
    print(batch, " | ".join(row), flush=True)
