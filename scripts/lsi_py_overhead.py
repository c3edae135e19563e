"""Where a warm LsiBatch.run() spends its time outside lexls_lsi_batch_run's own clock (LEXLS_LSI_TIMING=1 prints that one on stderr)."""
import os, sys, time; sys.path.insert(0, '.')
import numpy as np, ctypes as C
from lexls_amd import lexlsi, capi, problems as P
n, dims, B = 40, [12] * 5, 1024
base = lexlsi.pack_batch(n, [P.lsi_problem(20260500 + i, n, dims) for i in range(B)])
pert = lexlsi.pack_batch(n, [P.lsi_problem(20260500 + i, n, dims, perturb=0.9) for i in range(B)])
srv = lexlsi.LsiBatch(n, base.dims, base.types, B)
cold = srv.run(base)
guess = np.where(cold["active"] == 3, 0, cold["active"]).astype(np.uint8)
for _ in range(3): srv.run(pert, active_guess=guess, x0=cold["x"])
orig = capi.lib().lexls_lsi_batch_run
acc = [0.0]
class Wrap:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = orig(*a); acc[0] += time.perf_counter() - t0; return r
capi.lib().lexls_lsi_batch_run = Wrap()
N = 10
t0 = time.perf_counter()
for _ in range(N): srv.run(pert, active_guess=guess, x0=cold["x"])
el = time.perf_counter() - t0
print(f"run(): {el / N * 1e3:.3f} ms per call, of which inside the C call {acc[0] / N * 1e3:.3f} ms")
