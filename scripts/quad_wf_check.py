"""Development check of the factor-keeping four-per-wavefront kernel (policy 4, keep_factor): everything get_lexqr / hh / perm / x hold, bit for bit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lexls_amd as hip
from lexls_amd import problems as P
from oracle import oracle_ctypes as oracle


def check(name, lod, dims, n, maxdim=None):
    batch = lod.shape[0]
    dims_a = np.asarray(dims, np.uint32)
    if maxdim is None:
        maxdim = dims_a if dims_a.ndim == 1 else dims_a.max(axis=0)
        maxdim = np.array(maxdim, np.uint32)
        maxdim[-1] += lod.shape[2] - int(maxdim.sum())
    ref = oracle.lse_run(lod, dims, n, maxdim=maxdim)
    s = hip.BatchedLexLSE(batch, n, maxdim)
    s.set_kernel_policy(4)
    s.setObjDim(dims_a)
    s.setProblem(lod)
    s.factorize_solve(keep_factor=True)
    f = s.get_lexqr()
    m = dims_a.sum(axis=-1) if dims_a.ndim == 2 else np.full(batch, dims_a.sum())
    badf = [b for b in range(batch) if not np.array_equal(f[b, :, :m[b]], ref["factor"][b, :, :m[b]])]
    ok = dict(x=np.array_equal(s.get_x(), ref["x"]), perm=np.array_equal(s.get_column_permutations(), ref["perm"]), hh=np.array_equal(s.get_hh_scalars(), ref["hh"]),
              rank=np.array_equal(s.getRanks()[0], ref["rank"]), factor=not badf)
    print(f"{name:30s} {s.last_kernel():32s}", ok, flush=True)
    if badf:
        b = badf[0]
        d = np.argwhere(f[b, :, :m[b]] != ref["factor"][b, :, :m[b]])
        print("   problem", b, "first differing (col,row):", d[:6].tolist(), "ranks", ref["rank"][b].tolist(), "n diffs", len(d))
    return all(ok.values())


ok = True
n, dims = 40, [12] * 5
ok &= check("IK 16", P.lse_batch(20260100, 16, n, dims), dims, n)
ok &= check("IK 3", P.lse_batch(5, 3, n, dims), dims, n)
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
s = hip.BatchedLexLSE(batch, n, dims)
s.setProblem(lod)
for pol in (4, 3):
    s.set_kernel_policy(pol)
    s.factorize_solve(keep_factor=True)
    best = 1e9
    for rep in range(3):
        s.synchronize(); t0 = time.perf_counter()
        for _ in range(50): s.factorize_solve(keep_factor=True)
        s.synchronize(); best = min(best, (time.perf_counter() - t0) / 50)
    print(f"4096 factor kept, policy {pol}: {s.last_kernel()} {best*1e6:.1f} us")
print("ALL OK" if ok else "FAILURES")
