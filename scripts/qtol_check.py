"""Development check of the tolerance-contract four-per-wavefront kernel (lqr_qtol, policy 6 / automatic): ranks, first columns and
pivots EXACT against the oracle, x within 1e-10, on IK-shaped batches (full rank, rank-deficient levels, tied norms, batch tails), then
its wall time against the bit-exact kernel (policy 4).  Usage: python scripts/qtol_check.py [--time-only]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import lexls_amd as hip  # noqa: E402
from lexls_amd import problems as P  # noqa: E402
from oracle import oracle_ctypes as oracle  # noqa: E402

TOL = 1e-10


def check(name, lod, dims, n, policy=6):
    batch = lod.shape[0]
    ref = oracle.lse_run(lod, dims, n, nthreads=8)
    s = hip.BatchedLexLSE(batch, n, dims)
    s.set_kernel_policy(policy)
    s.setProblem(lod)
    s.factorize_solve(keep_factor=False)
    x = s.get_x()
    r, fc, tr = s.getRanks()
    perm = s.get_column_permutations()
    ok_r = np.array_equal(r, ref["rank"]) and np.array_equal(fc, ref["fcol"]) and np.array_equal(tr, ref["totalrank"])
    ok_p = np.array_equal(perm, ref["perm"])
    scale = max(1.0, float(np.abs(ref["x"]).max()))
    err = float(np.abs(x - ref["x"]).max())
    ok_x = np.isfinite(x).all() and err <= TOL * scale
    print(f"{name:34s} kernel={s.last_kernel():24s} ranks={'ok' if ok_r else 'BAD'} perm={'ok' if ok_p else 'BAD'} x err {err:.3e} (scale {scale:.2e}) {'ok' if ok_x else 'BAD'}", flush=True)
    if not ok_r:
        idx = np.where((r != ref["rank"]).any(axis=1))[0][:4]
        for i in idx:
            print("   problem", i, "ranks", r[i], "ref", ref["rank"][i], "fcol", fc[i], ref["fcol"][i])
    if not ok_p:
        idx = np.where((perm != ref["perm"]).any(axis=1))[0][:4]
        for i in idx:
            print("   problem", i, "perm", perm[i], "\n        ref", ref["perm"][i])
    if ok_r and ok_p and not ok_x:
        i = int(np.abs(x - ref["x"]).max(axis=1).argmax())
        print("   problem", i, "x", x[i][:8], "\n       ref", ref["x"][i][:8])
    return ok_r and ok_p and ok_x


def rank_deficient_ik(seed, batch, n=40, dims=(12,) * 5, ranks=(9, 12, 7, 12, 12)):
    return np.stack([P.rank_deficient_problem(seed + b, n, list(dims), list(ranks)) for b in range(batch)])


ok = True
n, dims = 40, [12] * 5
if "--time-only" not in sys.argv:
    ok &= check("IK 64", P.lse_batch(20260100, 64, n, dims), dims, n)
    ok &= check("IK 3 (wave tail)", P.lse_batch(5, 3, n, dims), dims, n)
    ok &= check("IK 1", P.lse_batch(6, 1, n, dims), dims, n)
    ok &= check("IK 1023", P.lse_batch_fast(7, 1023, n, dims), dims, n)
    ok &= check("rank deficient 9,12,7,12,12", rank_deficient_ik(100, 37), dims, n)
    ok &= check("rank deficient 3,3,3,3,3", rank_deficient_ik(300, 21, ranks=(3, 3, 3, 3, 3)), dims, n)
    ok &= check("rank deficient 12,0..", rank_deficient_ik(500, 9, ranks=(12, 1, 12, 1, 12)), dims, n)
    # mixed wavefronts: full-rank and rank-deficient problems side by side
    mix = P.lse_batch(900, 32, n, dims)
    mix[1::3] = rank_deficient_ik(700, 32, ranks=(5, 12, 12, 12, 12))[1::3]
    mix[2::5] = rank_deficient_ik(800, 32, ranks=(12, 12, 2, 12, 12))[2::5]
    ok &= check("mixed ranks inside wavefronts", mix, dims, n)
    # exact ties: duplicated columns (equal norms at every level -> first maximum by position decides)
    tied = P.lse_batch(1200, 16, n, dims)
    tied[:, 7, :] = tied[:, 3, :]
    tied[:, 30, :] = tied[:, 3, :]
    tied[:, 20, :] = tied[:, 19, :]
    ok &= check("tied norms (duplicate columns)", tied, dims, n)
    # 3 / 6 / 8 levels of 12 rows
    for nobj in (1, 2, 3, 4, 6, 8):
        d = [12] * nobj
        ok &= check(f"levels {nobj} x 12", P.lse_batch(40 + nobj, 19, n, d), d, n)
    ok &= check("automatic dispatch (policy 0)", P.lse_batch(31, 64, n, dims), dims, n, policy=0)

batch = 4096
lod = P.lse_batch_fast(20260100, batch, n, dims)
if "--time-only" not in sys.argv:
    ok &= check("IK 4096 (BASELINE configs[2])", lod, dims, n)
for policy in (6, 4):
    s = hip.BatchedLexLSE(batch, n, dims)
    s.set_kernel_policy(policy)
    s.setProblem(lod)
    s.factorize_solve(keep_factor=False)
    best = 1e9
    for rep in range(5):
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            s.factorize_solve(keep_factor=False)
        s.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100)
    print(f"policy {policy} {s.last_kernel():26s} {best * 1e6:8.1f} us per 4096-batch  ->  {batch / best:.3e} fact/s", flush=True)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
