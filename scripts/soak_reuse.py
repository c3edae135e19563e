"""Soak (not part of the suite): prefix reuse on random chains of LexLSI-like changes — random n, level capacities, ragged dimensions, fixed
variables, rank-deficient levels — every factorization (levels read back: random 0 .. nObj) against the oracle, bit for bit.
usage: python scripts/soak_reuse.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lexls_amd as hip
from lexls_amd import problems as P
from oracle import oracle_ctypes as oracle
import test_gpu_prefix_reuse as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20261006)
t0, chains, facts, read_back = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    n = int(rng.integers(6, 48))
    nobj = int(rng.integers(1, 7))
    md = int(rng.choice([4, 8, 12, 16]))
    maxd = [md] * nobj
    if sum(maxd) > 64:
        continue
    cap = sum(maxd)
    B = int(rng.integers(1, 40))
    fixed = None
    if rng.random() < 0.4:
        nf = rng.integers(0, min(4, n), size=B).astype(np.uint32)
        idx = np.zeros((B, n), np.uint32); val = np.zeros((B, n))
        for b in range(B):
            idx[b, :nf[b]] = rng.choice(n, size=nf[b], replace=False); val[b, :nf[b]] = rng.normal(size=nf[b])
        fixed = (nf, idx, val)
    def blocks_for(seed):
        dims = rng.integers(0, md + 1, size=nobj)
        if rng.random() < 0.3:  # exact dependence inside and across levels
            lod = P.rank_deficient_problem(seed, n, maxd, [int(rng.integers(0, md + 1)) for _ in range(nobj)])
            return [np.hstack([A, r[:, None]])[:d] for (A, r), d in zip(P.levels_of(lod, maxd), dims)]
        return T.random_blocks(seed, dims, n)
    cur = [blocks_for(1000 * chains + b) for b in range(B)]
    s = hip.BatchedLexLSE(B, n, maxd); s.set_kernel_policy(2); s.set_prefix_reuse(True)
    if fixed: s.fixVariables(*fixed)
    lod, dims = T.pack(cur, n, cap)
    s.setObjDim(dims); s.setProblem(lod); s.factorize_solve(True)
    if not s.prefix_reuse_ready():
        s.close(); continue
    for it in range(int(rng.integers(1, 8))):
        K = rng.integers(0, nobj + 1, size=B).astype(np.int32)
        cur = [T.change_level(cur[b], int(K[b]), int(rng.integers(0, 1 << 30)), int(rng.integers(0, 3)), n, md) for b in range(B)]
        lod, dims = T.pack(cur, n, cap)
        s.setObjDim(dims); s.setProblem(lod); s.set_resume_levels(K); s.factorize_solve(True)
        kw = dict(nfixed=fixed[0], fixed_idx=fixed[1], fixed_val=fixed[2]) if fixed else {}
        o = oracle.lse_run(lod, dims, n, maxdim=np.asarray(maxd, np.uint32), **kw)
        ctx = f"chain {chains} step {it}: n={n} maxd={maxd} B={B} fixed={bool(fixed)} K={K.tolist()} dims={dims.tolist()}"
        np.testing.assert_array_equal(s.get_x(), o["x"], err_msg=ctx)
        np.testing.assert_array_equal(s.get_column_permutations(), o["perm"], err_msg=ctx)
        np.testing.assert_array_equal(s.getRanks()[0], o["rank"], err_msg=ctx)
        f, hh = s.get_lexqr(), s.get_hh_scalars()
        for b in range(B):
            m = int(dims[b].sum())
            np.testing.assert_array_equal(f[b, :, :m], o["factor"][b, :, :m], err_msg=ctx + f" factor of problem {b}")
            np.testing.assert_array_equal(hh[b, :m], o["hh"][b, :m], err_msg=ctx + f" hh of problem {b}")
        facts += B; read_back += int(K.sum())
    s.close(); chains += 1
print(f"soak ok: {chains} chains, {facts} resumed factorizations ({read_back} levels read back) in {time.time() - t0:.0f} s")
