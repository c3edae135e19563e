import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lexls_amd
from lexls_amd import problems as P
n, dims = 512, [256] * 4
lod = P.lse_batch(20260001, 1, n, dims)
s = lexls_amd.BatchedLexLSE(1, n, dims); s.setProblem(lod)
for _ in range(3): s.factorize_solve(True)
s.synchronize()
