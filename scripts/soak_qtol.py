"""Soak (not part of the suite): the tolerance-contract kernel on random shapes it serves — n = 2 .. 40, 1 .. 8 levels of 12 rows, any batch size,
full-rank and rank-deficient (exact dependence, duplicated columns) — pivots / ranks / first columns exact, x within 1e-10 (contract T).
usage: python scripts/soak_qtol.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import lexls_amd as hip
from lexls_amd import problems as P
from oracle import oracle_ctypes as oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20261007)
t0, cases, kernels, worst = time.time(), 0, {}, 0.0
while time.time() - t0 < budget:
    n = int(rng.integers(2, 41))
    nobj = int(rng.integers(1, 9))
    dims = [12] * nobj
    B = int(rng.choice([1, 2, 3, 5, 17, 64, 200]))
    kind = int(rng.integers(0, 4))
    seed = int(rng.integers(0, 1 << 30))
    if kind == 0:
        lod = P.lse_batch_fast(seed, B, n, dims)
    elif kind == 1:
        ranks = [int(rng.integers(0, 13)) for _ in range(nobj)]
        lod = np.stack([P.rank_deficient_problem(seed + b, n, dims, ranks) for b in range(B)])
    elif kind == 2:  # duplicated columns: exact ties of the norms
        lod = P.lse_batch_fast(seed, B, n, dims)
        if n >= 4:
            a, b2 = rng.choice(n, size=2, replace=False)
            lod[:, a, :] = lod[:, b2, :]
    else:  # badly scaled rows / columns
        lod = P.lse_batch_fast(seed, B, n, dims)
        lod[:, :n, :] *= (10.0 ** rng.uniform(-3, 3, size=n))[None, :, None]
        lod *= (10.0 ** rng.uniform(-2, 2, size=lod.shape[2]))[None, None, :]
    ref = oracle.lse_run(lod, dims, n, nthreads=4)
    s = hip.BatchedLexLSE(B, n, dims)
    s.set_kernel_policy(6)
    s.setProblem(lod)
    s.factorize_solve(keep_factor=False)
    k = s.last_kernel()
    kernels[k] = kernels.get(k, 0) + 1
    ctx = f"case {cases}: n={n} levels={nobj} B={B} kind={kind} seed={seed} kernel={k}"
    r, fc, tr = s.getRanks()
    np.testing.assert_array_equal(r, ref["rank"], err_msg=ctx)
    np.testing.assert_array_equal(fc, ref["fcol"], err_msg=ctx)
    np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"], err_msg=ctx)
    x = s.get_x()
    assert np.isfinite(x).all(), ctx
    err = np.abs(x - ref["x"]).max() / max(1.0, float(np.abs(ref["x"]).max()))
    assert err <= 1e-10, ctx + f" err {err:.3e}"
    if k.startswith("lqr_qtol"): worst = max(worst, err)
    s.close(); cases += 1
print(f"soak ok: {cases} cases in {time.time() - t0:.0f} s; largest relative error of x on lqr_qtol {worst:.2e}; kernels: " + ", ".join(f"{k} x{v}" for k, v in sorted(kernels.items())))
