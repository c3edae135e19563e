import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import lexls_amd as hip
from lexls_amd import problems as P
from oracle import oracle_ctypes as oracle
n, dims = 200, [100, 100]
lod = P.lse_batch(43, 1, n, dims)
ref = oracle.lse_run(lod, dims, n)
for policy in (5, 0):
    s = hip.BatchedLexLSE(1, n, dims)
    s.set_kernel_policy(policy)
    s.setProblem(lod)
    s.factorize_solve()
    F = s.get_lexqr()[0]; R = ref["factor"][0]   # [col, row]
    print(policy, s.last_kernel(), "ranks", s.getRanks()[0], ref["rank"])
    print("  level0 rows, all cols    :", np.abs(F[:, :100] - R[:, :100]).max())
    print("  level1 rows, cols <100 (L):", np.abs(F[:100, 100:200] - R[:100, 100:200]).max())
    print("  level1 rows, cols>=100    :", np.abs(F[100:, 100:200] - R[100:, 100:200]).max())
    d = np.abs(F[100:, 100:200] - R[100:, 100:200])
    print("  perm eq:", np.array_equal(s.get_column_permutations(), ref["perm"]), " x err", np.abs(s.get_x() - ref["x"]).max())
