"""Diagnostic: how long does the wave kernel take when every level stops at its first pivot (tolerance = 1e300)?
Same launch geometry / registers / LDS as the real run, almost no arithmetic -> the launch + load + store floor."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lexls_amd
from lexls_amd import problems as P
n, dims = 40, [12] * 5
for batch in (1024, 2048, 4096, 8192):
    lod = P.lse_batch_fast(20260100, batch, n, dims)
    s = lexls_amd.BatchedLexLSE(batch, n, dims); s.setProblem(lod)
    for tol, name in ((1e-12, "real"), (1e300, "empty")):
        s.setParameters(tol)
        s.factorize_solve(False); s.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): s.factorize_solve(False)
        s.synchronize()
        print(batch, name, "ms: %.4f" % ((time.perf_counter() - t0) / 50 * 1e3), s.last_kernel())
