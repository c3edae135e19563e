"""Diagnostic: throughput of the bench kernel when independent batches are launched on S streams (one handle per stream)."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lexls_amd
from lexls_amd import problems as P
n, dims = 40, [12] * 5
for S, batch in ((1, 4096), (2, 4096), (2, 2048), (4, 1024), (4, 4096), (1, 8192)):
    lod = torch.from_numpy(P.lse_batch_fast(20260100, batch, n, dims)).cuda()
    hs, st = [], []
    for i in range(S):
        s = lexls_amd.BatchedLexLSE(batch, n, dims)
        stream = torch.cuda.Stream()
        s.set_stream(stream.cuda_stream)
        s.setProblemDevice(lod.data_ptr())
        hs.append(s); st.append(stream)
    for s in hs: s.factorize_solve(False)
    torch.cuda.synchronize()
    K = 100
    t0 = time.perf_counter()
    for _ in range(K):
        for s in hs: s.factorize_solve(False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"streams={S} batch/stream={batch}: {1e6*dt/K:.1f} us per round of {S*batch} problems -> {S*batch*K/dt/1e6:.1f} Mfact/s")
    del hs
