import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lexls_amd as hip
from lexls_amd import problems as P
from oracle import oracle_ctypes as oracle
# replay the soak's draw up to the failing case
rng = np.random.default_rng(20261007)
case = 0
while True:
    n = int(rng.integers(2, 41)); nobj = int(rng.integers(1, 9)); dims = [12] * nobj
    B = int(rng.choice([1, 2, 3, 5, 17, 64, 200])); kind = int(rng.integers(0, 4)); seed = int(rng.integers(0, 1 << 30))
    ranks = None
    if kind == 1: ranks = [int(rng.integers(0, 13)) for _ in range(nobj)]
    elif kind == 2:
        if n >= 4: rng.choice(n, size=2, replace=False)
    elif kind == 3:
        rng.uniform(-3, 3, size=n); rng.uniform(-2, 2, size=12 * nobj)
    if case == 71: break
    case += 1
print("case", case, n, nobj, B, kind, seed, ranks)
lod = np.stack([P.rank_deficient_problem(seed + b, n, dims, ranks) for b in range(B)])
ref = oracle.lse_run(lod, dims, n, nthreads=4)
s = hip.BatchedLexLSE(B, n, dims); s.set_kernel_policy(6); s.setProblem(lod); s.factorize_solve(False)
x = s.get_x(); print(s.last_kernel(), "oracle ranks of problem 0", ref["rank"][0])
err = np.abs(x - ref["x"]).max(axis=1) / np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
worst = int(err.argmax()); print("worst problem", worst, "err", err[worst], "|x|max", np.abs(ref["x"][worst]).max(), "count > 1e-10:", int((err > 1e-10).sum()), "median", np.median(err))
# conditioning: the oracle on data perturbed by one ulp (relative 1.1e-16, random signs)
pert = lod * (1.0 + 1.1e-16 * np.sign(np.random.default_rng(1).standard_normal(lod.shape)))
rp = oracle.lse_run(pert, dims, n, nthreads=4)
same = np.array_equal(rp["perm"], ref["perm"]) and np.array_equal(rp["rank"], ref["rank"])
e2 = np.abs(rp["x"] - ref["x"]).max(axis=1) / np.maximum(1.0, np.abs(ref["x"]).max(axis=1))
print("oracle on 1-ulp-perturbed data: same pivots/ranks:", same, " err of the same problem", e2[worst], " max over batch", e2.max())
# the x-only bit-exact kernel for reference
e = hip.BatchedLexLSE(B, n, dims); e.set_kernel_policy(4); e.setProblem(lod); e.factorize_solve(False)
print(e.last_kernel(), "max |x - oracle|", np.abs(e.get_x() - ref["x"]).max())
