"""Diagnostic: per-phase shader-clock shares of the register-resident wave kernel on LSI-like equality problems (ragged levels, a few
fixed variables, factor kept, 1024 problems = one wave per SIMD).  Needs a -DLEXLS_WAVE_STAMPS build via LEXLS_HIP_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lexls_amd
from lexls_amd import problems as P
n, cap_dims, batch = 40, [12] * 5, int(os.environ.get('STAMP_BATCH', '1024'))
full = os.environ.get('STAMP_FULL', '0') == '1'
rng = np.random.default_rng(7)
lod = P.lse_batch_fast(20260100, batch, n, cap_dims)
dims = np.full((batch, 5), 12, np.uint32) if full else rng.integers(4, 11, size=(batch, 5)).astype(np.uint32)
# rows of a level are the first dims[b,k] rows of the level's block: repack so that the rows are contiguous by current dims
cap = 60
packed = np.zeros_like(lod)
for b in range(batch):
    r = 0
    for k in range(5):
        d = int(dims[b, k])
        packed[b, :, r:r + d] = lod[b, :, 12 * k:12 * k + d]
        r += d
s = lexls_amd.BatchedLexLSE(batch, n, cap_dims)
s.setObjDim(dims)
if not full:
    nf = rng.integers(0, 5, size=batch).astype(np.uint32)
    idx = np.stack([rng.permutation(n) for _ in range(batch)]).astype(np.uint32)
    val = rng.standard_normal((batch, n))
    s.fixVariables(nf, idx, val)
s.setProblem(packed)
for _ in range(3): s.factorize_solve(True)
s.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20): s.factorize_solve(True)
s.synchronize()
print("avg us per call (back to back):", (time.perf_counter() - t0) / 20 * 1e6)
lam = s.getWorkspace()[:, :11]
names = ["load", "transpose/level load", "pivot search", "norms+rank", "hh scalars", "apply", "image", "trsm/eliminate", "gemm", "solve", "output"]
med = np.median(lam, axis=0); tot = med.sum()
for nm, v in zip(names, med): print(f"{nm:22s} {v:10.0f} cycles  {100*v/tot:5.1f}%")
print("batch", batch, "total", tot, "cycles/wave (median); max wave", lam.sum(axis=1).max(), "kernel", s.last_kernel())
