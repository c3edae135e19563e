import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lexls_amd
from lexls_amd import problems as P
n, dims = 512, [256] * 4
lod = P.lse_batch(20260001, 1, n, dims)
s = lexls_amd.BatchedLexLSE(1, n, dims); s.setProblem(lod)
s.factorize_solve(True); s.synchronize()
lam = s.getWorkspace()[0, :4]
print("apply block 0: load %.0f  chain %.0f  store %.0f cycles avg over %d pivots" % (lam[0]/lam[3], lam[1]/lam[3], lam[2]/lam[3], lam[3]))
