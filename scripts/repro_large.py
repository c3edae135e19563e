"""Re-run a case saved by scripts/soak_lse.py (gpurun_out/soak_fail.npz) on the large paths and report where the fast path leaves the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lexls_amd as hip
from oracle import oracle_ctypes as oracle
z = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "soak_fail.npz"))
lod, dims, cap_dims, n = z["lod"], z["dims"], z["cap_dims"], int(z["n"])
batch = lod.shape[0]
ref = oracle.lse_run(lod, dims, n, maxdim=cap_dims)
for pol in (0, 5):
    s = hip.BatchedLexLSE(batch, n, cap_dims)
    s.set_kernel_policy(pol)
    s.setObjDim(dims)
    s.setProblem(lod)
    s.factorize_solve(keep_factor=True)
    f = s.get_lexqr()
    d = np.abs(np.abs(f) - np.abs(ref["factor"]))[0]   # (n+1, cap): column = row of the problem
    rows = np.where(d.max(axis=0) > 1e-9)[0]
    cols = np.where(d.max(axis=1) > 1e-9)[0]
    F, Rf = f[0], ref["factor"][0]
    big = np.abs(Rf) > 1e-8
    mism = (np.sign(F) != np.sign(Rf)) & big
    per_row = mism.sum(axis=0)
    cnt = big.sum(axis=0)
    part = [(int(r), int(per_row[r]), int(cnt[r]), bool(mism[n, r])) for r in np.where(per_row > 0)[0]]
    print(" rows with sign differences (row, entries flipped, entries compared, rhs flipped):", part[:20])
    xs = s.get_x()[0]
    print(" x differs at variables:", np.where(np.abs(xs - ref["x"][0]) > 1e-9)[0][:30].tolist(), "perm equal", np.array_equal(s.get_column_permutations(), ref["perm"]))
    hh = s.get_hh_scalars()[0]
    print(" hh max diff", np.abs(hh - ref["hh"][0]).max())
    print(f"policy {pol} {s.last_kernel()} LEXLS_LARGE_PERSIST={os.environ.get('LEXLS_LARGE_PERSIST')}: max|dx| {np.abs(s.get_x() - ref['x']).max():.3e} max |d|factor|| {d.max():.3e}; "
          f"rows off: {rows[:12].tolist()}{'...' if len(rows) > 12 else ''} ({len(rows)}), variables off: {cols[:12].tolist()} ({len(cols)}); ranks {ref['rank'][0].tolist()} fcol {ref['fcol'][0].tolist()}")

# variants: without the empty last level; with dims == cap
def run(l, d, cd, tag):
    r = oracle.lse_run(l, d, n, maxdim=cd)
    s = hip.BatchedLexLSE(l.shape[0], n, cd)
    s.setObjDim(d)
    s.setProblem(l)
    s.factorize_solve(keep_factor=True)
    print(tag, s.last_kernel(), "max|dx| %.3e" % np.abs(s.get_x() - r["x"]).max(), "ranks", r["rank"][0].tolist())
m = int(dims[0].sum())
c3 = int(cap_dims[:3].sum())
run(lod[:, :, :c3].copy(), dims[:, :3].copy(), cap_dims[:3].copy(), "3 levels, same capacities     ")
run(lod[:, :, :m].copy(), dims[:, :3].copy(), dims[0, :3].copy(), "3 levels, capacities = dims   ")
d4 = dims.copy(); d4[0, 3] = 1
l4 = lod.copy(); l4[0, :, m] = np.arange(n + 1) * 0.01 + 0.3
run(l4, d4, cap_dims.copy(), "4 levels, one row in the last ")
