import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lexls_amd
from lexls_amd import problems as P
n, dims, batch = 40, [12]*5, 4096
lod = P.lse_batch_fast(20260100, batch, n, dims)
for keep in (False, True):
    s = lexls_amd.BatchedLexLSE(batch, n, dims); s.setProblem(lod)
    s.set_kernel_policy(4)
    for _ in range(3): s.factorize_solve(keep)
    s.synchronize()
    lam = s.getWorkspace()[::4, :11]
    names = ["init", "level load", "search+select+EX write", "EX round trip", "scalars (sqrt, div)", "apply+downdate", "level end (image, repack)", "eliminate", "load wait (vmcnt)", "solve", "output"]
    med = np.median(lam, axis=0); tot = med.sum()
    print("keep_factor", keep, "kernel", s.last_kernel(), "total", tot)
    for nm, v in zip(names, med): print(f"  {nm:28s} {v:10.0f} cycles  {100*v/tot:5.1f}%")
