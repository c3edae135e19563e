import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from lexls_amd import lexlsi, problems as P
batch = 1024
n, dims = 40, [12] * 5
base = lexlsi.pack_batch(n, [P.lsi_problem(20260500 + b, n, dims) for b in range(batch)])
pert30 = lexlsi.pack_batch(n, [P.lsi_problem(20260500 + b, n, dims, perturb=0.9) for b in range(batch)])
srv = lexlsi.LsiBatch(n, base.dims, base.types, batch)
for _ in range(3): cold = srv.run(base)
guess = np.where(cold["active"] == 3, 0, cold["active"]).astype(np.uint8)
for _ in range(3):
    t0 = time.perf_counter(); r = srv.run(pert30, active_guess=guess, x0=cold["x"]); print("python-level run:", time.perf_counter() - t0, file=sys.stderr)
srv.close()
