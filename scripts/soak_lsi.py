"""Soak test (not part of the suite): random lock-step LexLSI batches — shapes, simple bounds or not, cold and warm starts, factorization limits —
on the resident-iterations path, every instance against the oracle-backed driver.  usage: python scripts/soak_lsi.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lexls_amd import lexlsi, problems as P
from oracle import oracle_ctypes as oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20261004)
t0, cases, insts = time.time(), 0, 0
wide = bool(os.environ.get("SOAK_WIDE"))  # shapes beyond the one-wavefront kernels: generic kernel, host-side active-set logic
while time.time() - t0 < budget:
    n = int(rng.integers(50, 121)) if wide else int(rng.integers(4, 48))
    nobj = int(rng.integers(2, 6 if wide else 9))  # up to eight levels: deep hierarchies (more than 64 rows) take the left-looking kernels
    dims = [int(rng.integers(1, 41 if wide else 13)) for _ in range(nobj)]
    sb = bool(rng.integers(0, 2))
    if sb:
        dims[0] = min(dims[0], n)
    batch = int(rng.integers(1, 6 if wide else 40))
    seed = int(rng.integers(0, 1 << 30))
    problems = [P.lsi_problem(seed + b, n, dims, simple_bounds=sb) for b in range(batch)]
    kw = {}
    mode = int(rng.integers(0, 10))
    if mode in (0, 1):
        kw["max_number_of_factorizations"] = int(rng.integers(1, 8))
    elif mode == 2:  # regularized equality problems: host-side active-set logic, the wave kernel's REG instantiation
        kw["regularization_type"] = int(rng.choice([1, 8, 3, 5, 7]))
        kw["regularization_factors"] = (np.abs(rng.normal(size=nobj)) * 0.3 + 0.01) * (rng.random(nobj) < 0.8)
    elif mode == 3:  # cycling handling: host-side logic, problems assembled on the host
        kw.update(cycling_handling_enabled=1, cycling_max_counter=int(rng.integers(2, 6)), cycling_relax_step=1e-6)
    r = lexlsi.lsi_batch_solve(n, problems, **kw)
    refs = [oracle.lsi_run(n, problems[b], **kw) for b in range(batch)]
    for b in range(batch):
        o = refs[b]
        assert r["info"][b] == o["info"], (n, dims, sb, batch, seed, b, r["info"][b], o["info"])
        np.testing.assert_array_equal(r["active"][b], np.concatenate(o["active"]))
        if wide:  # problems beyond a CU's LDS take the large fast path: pivots exact, values to 1e-10 (north_star)
            np.testing.assert_allclose(r["x"][b], o["x"], rtol=0, atol=1e-10)
            np.testing.assert_allclose(r["v"][b], np.concatenate(o["v"]), rtol=0, atol=1e-9)
            continue
        np.testing.assert_array_equal(r["x"][b], o["x"])
        np.testing.assert_array_equal(r["v"][b], np.concatenate(o["v"]))
    if rng.integers(0, 2) == 0:  # warm start from the cold solution on perturbed data
        pert = [P.lsi_problem(seed + b, n, dims, simple_bounds=sb, perturb=float(rng.uniform(0.01, 1.0))) for b in range(batch)]
        guess = [[np.where(a == 3, 0, a) for a in refs[b]["active"]] for b in range(batch)]
        x0 = np.stack([refs[b]["x"] for b in range(batch)])
        rw = lexlsi.lsi_batch_solve(n, pert, active_guess=guess, x0=x0)
        for b in range(batch):
            o = oracle.lsi_run(n, pert[b], active_guess=guess[b], x0=refs[b]["x"])
            assert rw["info"][b] == o["info"], ("warm", n, dims, sb, batch, seed, b)
            if wide:
                np.testing.assert_allclose(rw["x"][b], o["x"], rtol=0, atol=1e-10)
                continue
            np.testing.assert_array_equal(rw["x"][b], o["x"])
            np.testing.assert_array_equal(rw["v"][b], np.concatenate(o["v"]))
    cases += 1
    insts += batch
print(f"soak ok: {cases} random batches, {insts} instances, {time.time() - t0:.0f} s")
