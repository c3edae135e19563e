"""Development check of the four-problems-per-wavefront kernel (policy 4): x / ranks / pivots against the oracle on the shapes the
parity tests use, then the wall time of the 4096-problem IK batch.  Usage: python scripts/quad_check.py [policy]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import lexls_amd as hip  # noqa: E402
from lexls_amd import problems as P  # noqa: E402
from oracle import oracle_ctypes as oracle  # noqa: E402

POLICY = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def check(name, lod, dims, n, maxdim=None):
    batch = lod.shape[0]
    dims_a = np.asarray(dims, np.uint32)
    if maxdim is None:
        maxdim = dims_a if dims_a.ndim == 1 else dims_a.max(axis=0)
        maxdim = np.array(maxdim, np.uint32)
        maxdim[-1] += lod.shape[2] - int(maxdim.sum())
    ref = oracle.lse_run(lod, dims, n, maxdim=maxdim)
    s = hip.BatchedLexLSE(batch, n, maxdim)
    s.set_kernel_policy(POLICY)
    s.setObjDim(dims_a)
    s.setProblem(lod)
    s.factorize_solve(keep_factor=False)
    x = s.get_x()
    r, fc, tr = s.getRanks()
    perm = s.get_column_permutations()
    ok_r = np.array_equal(r, ref["rank"]) and np.array_equal(fc, ref["fcol"]) and np.array_equal(tr, ref["totalrank"])
    ok_p = np.array_equal(perm, ref["perm"])
    ok_x = np.array_equal(x, ref["x"])
    err = float(np.abs(x - ref["x"]).max())
    bad = int((~(x == ref["x"]).all(axis=1)).sum())
    print(f"{name:34s} kernel={s.last_kernel():24s} ranks={'ok' if ok_r else 'BAD'} perm={'ok' if ok_p else 'BAD'} x={'bit-equal' if ok_x else f'DIFF max {err:.3e} in {bad} problems'}", flush=True)
    if not ok_r:
        idx = np.where((r != ref["rank"]).any(axis=1))[0][:4]
        for i in idx:
            print("   problem", i, "ranks", r[i], "ref", ref["rank"][i])
    return ok_r and ok_p and ok_x


ok = True
n, dims = 40, [12] * 5
ok &= check("IK 64", P.lse_batch(20260100, 64, n, dims), dims, n)
ok &= check("IK 3 (ragged wave tail)", P.lse_batch(5, 3, n, dims), dims, n)
ok &= check("IK 1", P.lse_batch(6, 1, n, dims), dims, n)
n2, d2, r2 = 15, [5, 5, 5, 5], [3, 3, 3, 3]
ok &= check("rank deficient", np.stack([P.rank_deficient_problem(100 + b, n2, d2, r2) for b in range(24)]), d2, n2)
n3, cap3 = 20, [8, 8, 8]
rd = np.array([[8, 8, 8], [3, 0, 5], [1, 8, 2], [0, 0, 4], [8, 1, 0], [5, 5, 5], [2, 2, 2], [7, 3, 8]], np.uint32)
full = np.zeros((8, n3 + 1, 24))
for b in range(8):
    m = int(rd[b].sum())
    full[b, :, :m] = P.lse_problem(900 + b, n3, rd[b])
ok &= check("ragged", full, rd, n3, maxdim=np.array(cap3, np.uint32))
for (nn, dd) in [(30, [9, 12, 5]), (40, [6] * 5), (5, [12, 12]), (40, [12, 0, 12, 12, 12]), (12, [1] * 8), (47, [12, 12, 12, 12]), (33, [11, 7, 12, 3])]:
    ok &= check(f"n={nn} dims={dd}", P.lse_batch(77, 13, nn, dd), dd, nn)

batch = 4096
lod = P.lse_batch_fast(20260100, batch, n, dims)
ref = oracle.lse_run(lod, dims, n, nthreads=8)
s = hip.BatchedLexLSE(batch, n, dims)
s.set_kernel_policy(POLICY)
s.setProblem(lod)
s.factorize_solve(keep_factor=False)
print("4096:", s.last_kernel(), "x bit-equal:", np.array_equal(s.get_x(), ref["x"]), "perm:", np.array_equal(s.get_column_permutations(), ref["perm"]), flush=True)
for rep in range(3):
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        s.factorize_solve(keep_factor=False)
    s.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"  {dt * 1e6:8.1f} us per 4096-batch  ->  {batch / dt:.3e} fact/s", flush=True)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
