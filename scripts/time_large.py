import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lexls_amd
from lexls_amd import problems as P
n, dims = 512, [256] * 4
lod = P.lse_batch(20260001, 1, n, dims)
s = lexls_amd.BatchedLexLSE(1, n, dims); s.setProblem(lod)
s.factorize_solve(True); s.synchronize()
t0 = time.perf_counter()
for _ in range(5): s.factorize_solve(True)
s.synchronize()
print(os.environ.get("LEXLS_HIP_LIB", "default").split("/")[-1], "ms per factorize+solve: %.2f" % ((time.perf_counter() - t0) / 5 * 1e3), s.last_kernel())
