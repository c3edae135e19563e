import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lexls_amd as hip
from lexls_amd import problems as P
n, dims = 512, [256] * 4
lod = P.lse_batch(20260001, 1, n, dims)
s = hip.BatchedLexLSE(1, n, dims); s.setProblem(lod)
for _ in range(2): s.factorize_solve(True)
s.synchronize()
w = s.getWorkspace()[0]
names = ["row dots with the solved levels", "stage diagonal block", "64-step chain (one wavefront)", "rows above the block", "x = P x + store"]
v = w[40:45]
print({nm: round(x) for nm, x in zip(names, v)}, "total cycles", round(v.sum()))
