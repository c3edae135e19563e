"""Soak test (not part of the suite): the random draw of tests/test_gpu_random_sweep.py::test_random_sweep run for a time budget, policy 4 (the
four-per-wavefront kernels: one / two / three / four slots, fixed-variable forms) and regularization types included; every case bit-identical to
the oracle.  usage: python scripts/soak_lse.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lexls_amd as hip
from lexls_amd import problems as P
from oracle import oracle_ctypes as oracle
from test_gpu_random_sweep import _draw, assert_x_matches
from test_gpu_parity import assert_factor_equal

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20261005)
t0, cases, kernels, sens_checks, extra_checks = time.time(), 0, {}, 0, 0
def _draw_wide(rng):
    """shapes beyond the one-wavefront kernels (SOAK_WIDE=1): the workgroup forms of the generic kernel, LDS-resident and in HBM"""
    n = int(rng.integers(40, 200))
    nobj = int(rng.integers(1, 6))
    md = int(rng.choice([20, 40, 100, 160]))
    cap_dims = rng.integers(1, md + 1, nobj)
    batch = int(rng.integers(1, 4))
    dims = np.stack([np.minimum(cap_dims, rng.integers(0, md + 1, nobj)) if rng.random() < 0.5 else cap_dims for _ in range(batch)]).astype(np.uint32)
    return n, cap_dims.astype(np.uint32), dims, batch


def _draw_deep(rng):
    """IK-sized problems with five to eight levels (SOAK_DEEP=1): mostly more than 64 rows in all — the left-looking kernels' share of the dispatch"""
    n = int(rng.integers(6, 64))
    nobj = int(rng.integers(5, 9))
    md = int(rng.choice([8, 10, 12, 16]))
    cap_dims = rng.integers(max(1, md - 4), md + 1, nobj)
    batch = int(rng.integers(1, 12))
    dims = np.stack([np.minimum(cap_dims, rng.integers(0, md + 1, nobj)) if rng.random() < 0.5 else cap_dims for _ in range(batch)]).astype(np.uint32)
    return n, cap_dims.astype(np.uint32), dims, batch


wide = bool(os.environ.get("SOAK_WIDE"))
deep = bool(os.environ.get("SOAK_DEEP"))
while time.time() - t0 < budget:
    n, cap_dims, dims, batch = _draw_wide(rng) if wide else (_draw_deep(rng) if deep else _draw(rng))
    batch = batch if rng.random() < 0.7 or wide or deep else int(rng.integers(5, 40))
    if dims.shape[0] != batch:
        dims = np.repeat(dims[:1], batch, axis=0)
    cap = int(cap_dims.sum())
    lod = np.zeros((batch, n + 1, cap))
    for b in range(batch):
        m = int(dims[b].sum())
        lod[b, :, :m] = P.normal(int(rng.integers(1, 2**31)), (n + 1) * m).reshape(n + 1, m)
        if m > 2 and rng.random() < 0.4:
            i, j = rng.integers(0, m, 2)
            lod[b, :, i] = lod[b, :, j] if rng.random() < 0.7 else 0.0
        if n > 2 and rng.random() < 0.3:
            i, j = rng.integers(0, n, 2)
            lod[b, i, :] = lod[b, j, :]
    fixed = {}
    if rng.random() < 0.4:
        nf = rng.integers(0, min(n, 40 if wide else 6) + 1, batch).astype(np.uint32)
        idx = np.zeros((batch, n), np.uint32)
        val = np.zeros((batch, n))
        for b in range(batch):
            idx[b, :nf[b]] = rng.choice(n, int(nf[b]), replace=False)
            val[b, :nf[b]] = rng.normal(size=int(nf[b]))
        fixed = dict(nfixed=nf, fixed_idx=idx, fixed_val=val)
    policy = int(rng.choice([0, 4, 3, 1, 2]))
    keep = bool(rng.random() < 0.5)
    reg = int(rng.choice([1, 8, 3, 5, 9, 7, 2, 4, 6])) if rng.random() < 0.15 else 0
    kw = {}
    fac = None
    if reg:
        fac = np.abs(rng.normal(size=len(cap_dims))) * 0.3 + 0.01
        kw = dict(reg_type=reg, reg_factors=fac)
    if os.environ.get("SOAK_LOG"):
        with open(os.environ["SOAK_LOG"], "a") as fh:
            fh.write(f"case {cases}: n={n} cap={cap_dims.tolist()} batch={batch} dims0={dims[0].tolist()} policy={policy} keep={keep} "
                     f"nfixed={fixed['nfixed'].tolist() if fixed else None} reg={reg}\n")
    ref = oracle.lse_run(lod, dims, n, maxdim=cap_dims, **fixed, **kw)
    s = hip.BatchedLexLSE(batch, n, cap_dims)
    s.set_kernel_policy(policy)
    s.setObjDim(dims)
    if reg:
        s.setRegularization(reg, fac)
    if fixed:
        s.fixVariables(fixed["nfixed"], fixed["fixed_idx"], fixed["fixed_val"])
    s.setProblem(lod)
    s.factorize_solve(keep_factor=keep)
    k = s.last_kernel()
    kernels[k] = kernels.get(k, 0) + 1
    ctx = f"case {cases}: n={n} cap={cap_dims.tolist()} dims={dims.tolist()} policy={policy} keep={keep} fixed={bool(fixed)} reg={reg} kernel={k}"
    if k.startswith("lqr_large<step-per-pivot"):  # the large fast path: pivots exact, values to 1e-10 (north_star)
        err = np.abs(s.get_x() - ref["x"]).max()
        if err > 1e-10 * max(1.0, np.abs(ref["x"]).max()):
            print("FAIL", ctx, "\n max |dx| =", err, "max |x| =", np.abs(ref["x"]).max(), "perm equal:", np.array_equal(s.get_column_permutations(), ref["perm"]),
                  "ranks", s.getRanks()[0].tolist(), ref["rank"].tolist())
            g = hip.BatchedLexLSE(batch, n, cap_dims)
            g.set_kernel_policy(5)
            g.setObjDim(dims)
            g.setProblem(lod)
            g.factorize_solve(keep_factor=True)
            print(" bit-exact large path", g.last_kernel(), "max |dx| =", np.abs(g.get_x() - ref["x"]).max())
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "soak_fail.npz"), lod=lod, dims=dims, cap_dims=cap_dims, n=n)
            sys.exit(1)
    else:
        assert_x_matches(s, ref["x"], ctx)
    np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"], err_msg=ctx)
    np.testing.assert_array_equal(s.getRanks()[0], ref["rank"], err_msg=ctx)
    if keep:
        if s.last_kernel().startswith("lqr_large"):
            pass
        else:
            assert_factor_equal(s, ref, dims, n)
    if keep and not k.startswith("lqr_large") and rng.random() < 0.2:  # residuals and a least-norm solve from the factor just made
        np.testing.assert_array_equal(s.get_v(), ref["v"], err_msg=ctx + " get_v")
        opt = int(rng.choice([1, 2]))
        refn = oracle.lse_run(lod, dims, n, maxdim=cap_dims, solve_option=opt, **fixed, **kw)
        (s.solveLeastNorm_1 if opt == 1 else s.solveLeastNorm_2)()
        np.testing.assert_array_equal(s.get_x(), refn["x"], err_msg=ctx + f" least-norm {opt}")
        extra_checks += 1
    if reg == 7:  # by-products of the experimental TIKHONOV_1 (generic kernel)
        xm, _, rm = s.get_mu()
        np.testing.assert_array_equal(xm, ref["x_mu"], err_msg=ctx)
        np.testing.assert_array_equal(rm, ref["residual_mu"], err_msg=ctx)
    if keep and not k.startswith("lqr_large") and rng.random() < 0.3:  # a removal search on the factor just made (type 7: multipliers of the regularized problem)
        lvl = int(rng.integers(0, len(cap_dims)))
        types = rng.integers(1, 4, (batch, cap)).astype(np.uint8)  # LB / UB / EQ
        ref = oracle.lse_run(lod, dims, n, maxdim=cap_dims, sens_obj=lvl, ctr_type=types, **fixed, **kw)
        s.setCtrType(types)
        found, ctr, obj, maxabs = s.ObjectiveSensitivity(lvl)
        np.testing.assert_array_equal(s.getWorkspace(), ref["lam"], err_msg=ctx + f" sens {lvl}")
        np.testing.assert_array_equal(np.stack([found.astype(np.int32), ctr, obj], 1), ref["sens"], err_msg=ctx + f" sens {lvl}")
        np.testing.assert_array_equal(maxabs, ref["maxabs"], err_msg=ctx + f" sens {lvl}")
        sens_checks += 1
    s.close()
    cases += 1
print(f"soak ok: {cases} random cases ({sens_checks} with a removal search, {extra_checks} with residuals + a least-norm solve) in {time.time() - t0:.0f} s; kernels: " + ", ".join(f"{k} x{v}" for k, v in sorted(kernels.items())))
