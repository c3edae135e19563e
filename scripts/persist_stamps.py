import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lexls_amd as hip
from lexls_amd import problems as P
n, dims = 512, [256] * 4
lod = P.lse_batch(20260001, 1, n, dims)
s = hip.BatchedLexLSE(1, n, dims); s.setProblem(lod)
for _ in range(3): s.factorize()
s.synchronize()
w = s.getWorkspace()[0]
names = ["publish+drain", "barrier (arrive+wait)", "read candidates + winner", "read column + norms", "reflector + tile + bookkeeping"]
for lvl in range(2):
    v = w[8 * lvl: 8 * lvl + 5]
    print("level", lvl, "cycles per pivot:", {nm: round(x / 256) for nm, x in zip(names, v)}, "total/pivot", round(v.sum() / 256))
