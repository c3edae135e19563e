"""Diagnostic: per-phase shader-clock shares of lqr_mfma (needs a -DLEXLS_WAVE_STAMPS build via LEXLS_HIP_LIB)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lexls_amd
from lexls_amd import problems as P
n, dims, batch = 40, [12]*5, int(os.environ.get('STAMP_BATCH', '4096'))
lod = P.lse_batch_fast(20260100, batch, n, dims)
s = lexls_amd.BatchedLexLSE(batch, n, dims); s.setProblem(lod)
pol = int(os.environ.get("STAMP_POLICY", "7"))
s.set_kernel_policy(pol)
for _ in range(3): s.factorize_solve(False)
s.synchronize()
ws = s.getWorkspace()
step = 2 if pol == 7 else 1
lv = np.median(ws[::step, 11:11 + 20], axis=0).reshape(5, 4)
print("kernel", s.last_kernel(), "batch", batch)
print("per level:     Gauss   level start  Householder  level end")
for k in range(5): print(f"  level {k}: " + "  ".join(f"{v:9.0f}" for v in lv[k]))
print("levels total (first stamp -> after last level)", np.median(ws[::step, 8]), " solve", np.median(ws[::step, 9]))
g = np.median(ws[::step, 32:40], axis=0)
print("Gauss phase (sum over levels): wait for the level", g[0], " C tiles in", g[1], " [2]", g[2], " [3]", g[3], " C out", g[4], " B reads issued+landed [5]", g[5], " extraction + A reads landed [6]", g[6], " mfma issue [7]", g[7])
c = np.median(ws[::step, 44:50], axis=0)
if c[5] > 0:
    print("pivot steps with one live slot (%d steps), cycles per step: norms ready -> high-word maximum %.0f | -> winner's column + position in every lane (store, LDS, read) %.0f | -> fresh norm %.0f | -> norms down-dated %.0f | -> rows below updated %.0f | sum %.0f"
          % (c[5], c[0] / c[5], c[1] / c[5], c[2] / c[5], c[3] / c[5], c[4] / c[5], c[:5].sum() / c[5]))
