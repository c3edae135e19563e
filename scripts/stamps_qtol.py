"""Diagnostic: per-phase shader-clock shares of lqr_qtol (needs a -DLEXLS_WAVE_STAMPS build via LEXLS_HIP_LIB; -DLEXLS_WAVE_STAMPS_FINE adds
the stamps inside the pivot steps, which disturb what they measure)."""
import os, sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import lexls_amd
from lexls_amd import problems as P
n, dims, batch = 40, [12]*5, int(os.environ.get('STAMP_BATCH', '4096'))
lod = P.lse_batch_fast(20260100, batch, n, dims)
s = lexls_amd.BatchedLexLSE(batch, n, dims); s.setProblem(lod)
s.set_kernel_policy(int(os.environ.get("STAMP_POLICY", "6")))
for _ in range(3): s.factorize_solve(False)
s.synchronize()
ws = s.getWorkspace()
lam = ws[::4, :11]
names = ["init", "level load (stage, transpose)", "select + EX write*", "EX round trip + norms*", "bookkeeping, dots, sqrt, rcp*", "Householder phase (or *: row j, search, update)", "level end (image)", "eliminate", "-", "solve", "output"]
med = np.median(lam, axis=0); tot = med.sum()
for nm, v in zip(names, med): print(f"{nm:48s} {v:10.0f} cycles  {100*v/tot:5.1f}%")
print("batch", batch, "total", tot, "cycles/wave (median); kernel", s.last_kernel())
lv = np.median(ws[::4, 11:11 + 16], axis=0).reshape(4, 4)
print("per level:      load  eliminate  Householder  level end")
for k in range(4): print(f"  level {k}: " + "  ".join(f"{v:9.0f}" for v in lv[k]))
if os.environ.get("CHAIN"):
    ch = np.median(ws[::4, 30:37], axis=0)
    npiv = {"0": 12, "1": 24, "2": 4}[os.environ["CHAIN"]]
    names = ["handoff stored -> step top (decision ready, branch)", "step top -> pivot column arrived (LDS read)", "-> fresh norm", "-> 1/beta", "-> row j, norms down-dated",
             "-> decision for the next pivot", "-> rows below updated, hand-off stored"]
    print(f"chain stamps, S0 = {os.environ['CHAIN']} ({npiv} pivots), cycles per pivot:")
    for nm, v in zip(names, ch): print(f"  {nm:55s} {v / npiv:8.0f}")
    print(f"  {'sum':55s} {ch.sum() / npiv:8.0f}")
