"""Development check + timing of the large path (BASELINE configs[1]).  Usage: python scripts/large_check.py [policy]"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import lexls_amd as hip
from lexls_amd import problems as P
from oracle import oracle_ctypes as oracle
policy = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for (n, dims, seed, batch) in [(200, [100] * 4, 43, 2), (150, [90, 90, 90], 5, 1), (512, [256] * 4, 20260001, 1)]:
    lod = P.lse_batch(seed, batch, n, dims)
    ref = oracle.lse_run(lod, dims, n)
    s = hip.BatchedLexLSE(batch, n, dims)
    s.set_kernel_policy(policy)
    s.setProblem(lod)
    s.factorize_solve()
    r = s.getRanks()[0]
    perm_ok = np.array_equal(s.get_column_permutations(), ref["perm"])
    print(f"n={n} dims={dims} kernel={s.last_kernel()} ranks={'ok' if np.array_equal(r, ref['rank']) else 'BAD ' + str(r.tolist())} perm={'ok' if perm_ok else 'BAD'} "
          f"|x-x_ref|={np.abs(s.get_x() - ref['x']).max():.3e} |F-F_ref|={np.abs(s.get_lexqr() - ref['factor']).max():.3e} |hh|={np.abs(s.get_hh_scalars() - ref['hh']).max():.3e}", flush=True)
    if not perm_ok:
        d = np.where(s.get_column_permutations()[0] != ref["perm"][0])[0]
        print("   first differing pivot positions:", d[:8], s.get_column_permutations()[0][d[:8]], ref["perm"][0][d[:8]])
best = 1e9
for rep in range(5):
    s.synchronize(); t0 = time.perf_counter()
    s.factorize_solve()
    s.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"configs[1] factorize+solve: {best*1e3:.3f} ms  ({P.flop_model(512, [256]*4)['total']/best/1e9:.1f} GFLOP/s)")
best = 1e9
for rep in range(5):
    s.synchronize(); t0 = time.perf_counter()
    s.factorize()
    s.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"configs[1] factorize only : {best*1e3:.3f} ms")
