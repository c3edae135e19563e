"""Diagnostic: per-phase shader-clock totals of the register-resident wave kernel, factor kept (needs a -DLEXLS_WAVE_STAMPS build of
lqr_small_41x12e_f.hip via LEXLS_HIP_LIB).  usage: python scripts/stamps_wave.py [batch]"""
import os, sys, time; sys.path.insert(0, '.')
import numpy as np
import lexls_amd
from lexls_amd import problems as P
n, dims = 40, [12] * 5
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lod = P.lse_batch_fast(20260100, batch, n, dims)
s = lexls_amd.BatchedLexLSE(batch, n, dims); s.set_kernel_policy(2); s.setProblem(lod)
for _ in range(3): s.factorize_solve(True)
s.synchronize()
t0 = time.perf_counter()
for _ in range(20): s.factorize_solve(True)
s.synchronize()
dt = (time.perf_counter() - t0) / 20
lam = s.getWorkspace()[:, :11]
names = ["load", "level load", "pivot search", "norms+rank", "hh scalars", "apply", "image", "eliminate (L)", "eliminate (update)", "solve", "output"]
med = np.median(lam, axis=0); tot = med.sum()
for nm, v in zip(names, med): print(f"{nm:20s} {v:10.0f} cycles  {100*v/tot:5.1f}%")
print("batch", batch, "total", tot, "cycles/wave (median); kernel", s.last_kernel(), f"{dt*1e6:.1f} us per call (with stamps)")
