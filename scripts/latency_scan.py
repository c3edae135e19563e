import sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np, lexls_amd
from lexls_amd import problems as P
n, dims = 40, [12]*5
out = {}
for batch in (256, 512, 1024, 2048):
    lod = P.lse_batch_fast(20260100, batch, n, dims)
    for pol in (0, 2):
        for keep in (True, False):
            s = lexls_amd.BatchedLexLSE(batch, n, dims)
            s.set_kernel_policy(pol)
            s.setProblem(lod)
            s.factorize_solve(keep); s.synchronize()
            t0 = time.perf_counter()
            for _ in range(50): s.factorize_solve(keep)
            s.synchronize()
            out[f"B{batch} pol{pol} keep{int(keep)} {s.last_kernel().split('<')[0]}"] = round((time.perf_counter()-t0)/50*1e6, 1)
print(json.dumps(out, indent=0))
