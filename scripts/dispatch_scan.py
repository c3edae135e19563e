"""Is the automatic dispatch the fastest kernel?  Times every kernel policy (0 automatic, 2 register-resident, 3 left-looking wave, 4 four per wavefront)
on a grid of IK-like shapes x batch sizes x {x only, factor kept} and prints the cases where the automatic choice is more than 10 % behind the best.
usage: python scripts/dispatch_scan.py [--fixed] [-v]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lexls_amd
from lexls_amd import problems as P
shapes = [(12, [4, 4, 4]), (20, [6, 5, 5, 6]), (30, [9, 12, 5]), (40, [12] * 5), (47, [12] * 4), (55, [16, 14, 16, 12]), (63, [16] * 4), (40, [12] * 8)]
bad = 0
for n, dims in shapes:
    for batch in (256, 1024, 2048, 4096, 16384):
        lod = P.lse_batch_fast(7, batch, n, dims) if hasattr(P, "lse_batch_fast") else P.lse_batch(7, batch, n, dims)
        for keep in (False, True):
            res = {}
            for pol in (0, 2, 3, 4):
                s = lexls_amd.BatchedLexLSE(batch, n, dims)
                s.set_kernel_policy(pol)
                if "--fixed" in sys.argv:  # three fixed variables per problem (an active simple-bounds level of a LexLSI iteration)
                    idx = np.zeros((batch, n), np.uint32)
                    idx[:, :3] = [min(5, n - 1), 1, min(9, n - 2)]
                    s.fixVariables(np.full(batch, 3, np.uint32), idx, np.zeros((batch, n)))
                s.setProblem(lod)
                for _ in range(3):
                    s.factorize_solve(keep)
                s.synchronize()
                t0 = time.perf_counter()
                for _ in range(8):
                    s.factorize_solve(keep)
                s.synchronize()
                res[pol] = ((time.perf_counter() - t0) / 8 * 1e6, s.last_kernel())
                s.close()
            best = min(res, key=lambda p: res[p][0])
            flag = res[0][0] > 1.10 * res[best][0]
            bad += flag
            if flag or "-v" in sys.argv:
                print(f"n={n:3d} {len(dims)}x{max(dims):2d} batch={batch:6d} keep={int(keep)}: auto {res[0][1]:28s} {res[0][0]:8.1f} us | best policy {best} {res[best][1]:28s} {res[best][0]:8.1f} us" + ("   <-- auto is behind" if flag else ""))
print(f"{bad} cases where the automatic choice is more than 10 % behind")
