import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/lexls_amd') else os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, lexls_amd
from lexls_amd import problems as P
for (n, dims, nfix, batch) in ((88, [3, 2, 97], 30, 1), (88, [74, 33, 3, 2, 97], 0, 1), (100, [40, 40, 40], 0, 64)):
    lod = P.lse_batch(11, batch, n, dims)
    s = lexls_amd.BatchedLexLSE(batch, n, dims); s.set_kernel_policy(1)
    if nfix:
        idx = np.zeros((batch, n), np.uint32); idx[:, :nfix] = np.arange(0, 2 * nfix, 2)
        s.fixVariables(np.full(batch, nfix, np.uint32), idx, np.zeros((batch, n)))
    s.setProblem(lod)
    for _ in range(3): s.factorize_solve(True)
    s.synchronize(); t0 = time.perf_counter()
    for _ in range(20): s.factorize_solve(True)
    s.synchronize()
    print(f"n={n} dims={dims} batch={batch} {s.last_kernel()}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us")
