"""Diagnostic: per-phase shader-clock shares of sensitivity_sweep_kernel on the LSI bench batch (needs a -DLEXLS_SWEEP_STAMPS build via LEXLS_HIP_LIB):
runs a lock-step batch with the host-side logic and reads the stamps the last sweep left behind the multipliers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LEXLS_LSI_RESIDENT"] = "0"
import numpy as np
import lexls_amd
from lexls_amd import problems as P
n, dims, batch = 40, [12] * 5, 1024
rng = np.random.default_rng(3)
lod = P.lse_batch_fast(20260100, batch, n, dims)
d = rng.integers(4, 11, size=(batch, 5)).astype(np.uint32)
packed = np.zeros_like(lod)
for b in range(batch):
    r = 0
    for k in range(5):
        packed[b, :, r:r + int(d[b, k])] = lod[b, :, 12 * k:12 * k + int(d[b, k])]
        r += int(d[b, k])
s = lexls_amd.BatchedLexLSE(batch, n, dims)
s.setObjDim(d)
s.setProblem(packed)
s.setCtrType(np.full((batch, 60), 2, np.uint8))
s.factorize_solve(True)
s.setSensitivityScan(True)
for _ in range(3):
    s.ObjectiveSensitivity(0)
lam = s.getWorkspace()[:, -6:]
names = ["stage factor + init", "Householder sequences", "L^T lambda products", "fixed variables", "decisions", "outputs"]
med = np.median(lam, axis=0); tot = med.sum()
for nm, v in zip(names, med): print(f"{nm:24s} {v:10.0f} cycles  {100*v/tot:5.1f}%")
print("total", tot, "cycles/wave (median)")
