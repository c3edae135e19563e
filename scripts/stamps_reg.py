"""Diagnostic: per-phase shader-clock totals of the register-resident kernel with a regularization type set (needs a -DLEXLS_WAVE_STAMPS build
via LEXLS_HIP_LIB).  usage: python scripts/stamps_reg.py [batch]"""
import os, sys; sys.path.insert(0, '.')
import numpy as np
import lexls_amd
from lexls_amd import problems as P
n, dims = 40, [12] * 5
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lod = P.lse_batch_fast(11, batch, n, dims)
names = ["load", "level load", "pivot search", "norms+rank", "hh scalars", "apply", "image + damping", "eliminate", "gemm", "solve", "output"]
for rt in (4, 5, 3, 1, 8, 2):
    s = lexls_amd.BatchedLexLSE(batch, n, dims)
    s.setRegularization(rt, [0.01] * 5)
    s.setProblem(lod)
    for _ in range(2): s.factorize_solve(True)
    s.synchronize()
    ws = s.getWorkspace()
    med = np.median(ws[:, :11], axis=0)
    inner = np.median(ws[:, 16:31], axis=0)
    print(f"type {rt} {s.last_kernel()}: total {med.sum():9.0f} cycles/wave | " + " | ".join(f"{nm} {v:.0f}" for nm, v in zip(names, med)) + "\n      inside the damping: " + " ".join(f"{v:.0f}" for v in inner), flush=True)
    s.close()
