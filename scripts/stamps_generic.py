"""Diagnostic: per-phase shader-clock shares of lqr_generic_kernel on a problem shaped like BASELINE configs[0] at its last iterations
(n = 88, 30 fixed variables, levels [3, 2, 97]); needs a -DLEXLS_GENERIC_STAMPS build (scripts/build_variant.sh) loaded with LEXLS_HIP_LIB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lexls_amd
from lexls_amd import problems as P
n, dims, nfix = 88, [3, 2, 97], 30
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lod = P.lse_batch(11, batch, n, dims)
s = lexls_amd.BatchedLexLSE(batch, n, dims)
s.set_kernel_policy(1)
idx = np.zeros((batch, n), np.uint32); idx[:, :nfix] = np.arange(0, 2 * nfix, 2)
s.fixVariables(np.full(batch, nfix, np.uint32), idx, np.zeros((batch, n)))
s.setProblem(lod)
for _ in range(3):
    s.factorize_solve(True)
s.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    s.factorize_solve(True)
s.synchronize()
print(f"{s.last_kernel()}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per factorize_solve (batch {batch})")
lam = s.getWorkspace()[:, :10]
names = ["stage + init + fixed", "level norms", "pivot search", "fresh norm + scalars", "column swap", "apply + down-date", "regularize + Gauss", "results + factor store", "solve", "(of apply) dot products"]
med = np.median(lam, axis=0); tot = med.sum()
if tot > 0:
    for nm, v in zip(names, med): print(f"{nm:24s} {v:10.0f} cycles  {100*v/tot:5.1f}%")
    print("total", tot, "cycles (100 MHz shader clock ticks x ?)")
