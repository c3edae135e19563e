"""The CPU oracle against everything that pins it:
  * the reference's own fixture tests/test_01.dat `#Solution` (copied to tests/golden/test_01.dat),
  * known-answer tests restated from the reference's MATLAB suites,
  * an independent numpy solver that does not use the l-QR algorithm,
  * committed golden vectors (tests/golden/lse_golden.npz, made by tests/golden/make_golden.py).
No GPU needed."""
import os

import numpy as np
import pytest

from lexls_amd import problems as P
from oracle import oracle_np

from conftest import GOLDEN

DAT = os.path.join(GOLDEN, "test_01.dat")


@pytest.mark.parametrize("use_as,use_x", [(0, 0), (1, 0), (1, 1), (0, 1)])
def test_reference_fixture_test_01_solution(oracle, use_as, use_x):
    """reference tests/test_01.dat:328-416 stores x* with 15 significant digits; every warm-start mode must reach it."""
    d = oracle.lsi_run_dat(DAT, one_based=True, use_active_guess=bool(use_as), use_x_guess=bool(use_x))
    assert d["header"].tolist() == [88, 5, 210, 1]
    assert d["info"]["status"] == 0  # PROBLEM_SOLVED
    assert np.abs(d["x"] - d["solution"]).max() < 1e-10  # north_star's tolerance; measured 7e-13
    if use_as:  # the stored active set is optimal: one factorization, nothing added or removed
        assert d["info"]["factorizations"] == 1 and d["info"]["activations"] == 0 and d["info"]["deactivations"] == 0


def test_lambda_closed_form(oracle):
    """interfaces/matlab-octave/tests/lexlsi/lambda_test.m:8-47: (x_1=1) > (2x_2=1) > ... > (n x_n=1) > (sum x = 1)."""
    n = 5
    w = sum(1.0 / k for k in range(2, n + 1))
    objs = []
    for k in range(1, n + 1):
        A = np.zeros((1, n))
        A[0, k - 1] = k
        objs.append(dict(A=A, lb=[1.0], ub=[1.0]))
    objs.append(dict(A=np.ones((1, n)), lb=[1.0], ub=[1.0]))
    x, lam = oracle.lsi_lambda(n, objs)
    expect = np.zeros((n + 1, n + 1))
    expect[n, n] = w
    for k in range(1, n + 1):
        expect[k - 1, n] = -w / k
    np.testing.assert_allclose(x, [1.0 / k for k in range(1, n + 1)], atol=1e-14)
    np.testing.assert_allclose(lam, expect, atol=1e-13)


def test_pinv_identity(oracle):
    """examples/example_lexlse.m:17-29: with a terminal level x = 0, x equals pinv(A1) b1."""
    n, m = 10, 6
    A1 = P.normal(1, m * n).reshape(m, n)
    b1 = P.normal(2, m)
    lod = np.zeros((1, n + 1, m + n))
    lod[0, :n, :m] = A1.T
    lod[0, n, :m] = b1
    lod[0, :n, m:] = np.eye(n)
    r = oracle.lse_run(lod, [m, n], n)
    np.testing.assert_allclose(r["x"][0], np.linalg.pinv(A1) @ b1, atol=1e-12)
    assert r["rank"][0].tolist() == [m, n - m]


def test_least_norm_equals_terminal_identity_level(oracle):
    """append_terminal_objective.m / test_lexlse_main.m:16-20 (tol 1e-10): Givens least-norm == extra level I x = 0."""
    n, dims = 30, [9, 8, 10]
    lod = np.stack([P.rank_deficient_problem(40 + b, n, dims, [7, 6, 8]) for b in range(4)])
    ln = oracle.lse_run(lod, dims, n, solve_option=1)
    ext = np.zeros((4, n + 1, sum(dims) + n))
    ext[:, :, :sum(dims)] = lod
    ext[:, :n, sum(dims):] = np.eye(n)
    basic = oracle.lse_run(ext, dims + [n], n)
    np.testing.assert_allclose(ln["x"], basic["x"], atol=1e-10)


def test_fixed_variables_equal_unit_row_equalities(oracle):
    """fixed2general.m: fixing variables == a first level of unit rows (tol 1e-10, test_lexlse_main.m:20)."""
    n, dims, batch = 12, [4, 5], 5
    lod = P.lse_batch(77, batch, n, dims)
    nfixed = np.full(batch, 3, np.uint32)
    idx = np.zeros((batch, n), np.uint32)
    val = np.zeros((batch, n))
    gen = np.zeros((batch, n + 1, 3 + sum(dims)))
    for b in range(batch):
        perm = np.argsort(P.uniform(90 + b, n))[:3]
        idx[b, :3] = perm
        val[b, :3] = P.normal(95 + b, 3)
        for k in range(3):
            gen[b, perm[k], k] = 1.0
            gen[b, n, k] = val[b, k]
        gen[b, :, 3:] = lod[b]
    fx = oracle.lse_run(lod, dims, n, nfixed=nfixed, fixed_idx=idx, fixed_val=val)
    ge = oracle.lse_run(gen, [3] + dims, n)
    np.testing.assert_allclose(fx["x"], ge["x"], atol=1e-10)
    np.testing.assert_allclose(fx["v"][:, :sum(dims)], ge["v"][:, 3:], atol=1e-10)


@pytest.mark.parametrize("n,dims,ranks", [(15, [5, 5, 5, 5], [3, 3, 3, 3]), (40, [12] * 5, None), (30, [9, 8, 10, 6], [7, 6, 8, 5]), (8, [3, 3, 3, 3], None)])
def test_against_independent_numpy_solver(oracle, n, dims, ranks):
    """per-level optimal residual norms are unique: compare with sequential null-space projection (oracle_np)."""
    for seed in range(6):
        lod = (P.rank_deficient_problem(seed, n, dims, ranks) if ranks else P.lse_problem(seed, n, dims))[None]
        r = oracle.lse_run(lod, dims, n)
        levels = P.levels_of(lod[0], dims)
        xn, resn, rk = oracle_np.lex_solve(levels)
        res = oracle_np.residual_norms(levels, r["x"][0])
        np.testing.assert_allclose(res, resn, atol=1e-9)
        assert r["rank"][0].tolist() == rk
        # get_v reproduces the residuals A x - b
        A = lod[0, :-1, :].T
        np.testing.assert_allclose(r["v"][0], A @ r["x"][0] - lod[0, -1, :], atol=1e-9)
        if sum(rk) == n:
            np.testing.assert_allclose(r["x"][0], xn, atol=1e-9)


def test_factor_structure(oracle):
    """R_k upper triangular with |diag| decreasing-ish pivots, Q_k orthogonal: Q^T [A P] = [R T] on level 1."""
    n, dims = 10, [6, 6]
    lod = P.lse_problem(5, n, dims)[None]
    r = oracle.lse_run(lod, dims, n)
    F = r["factor"][0].T  # (M, n+1)
    perm = r["perm"][0]
    A = lod[0].T.copy()
    # replay the column swaps on the data of level 1
    for k in range(int(r["totalrank"][0])):
        A[:, [k, perm[k]]] = A[:, [perm[k], k]]
    # rebuild Q_1 from the stored essentials / scalars
    m = dims[0]
    Q = np.eye(m)
    for j in range(int(r["rank"][0][0])):
        v = np.zeros(m)
        v[j] = 1.0
        v[j + 1:] = F[j + 1:m, j]
        Q = Q @ (np.eye(m) - r["hh"][0][j] * np.outer(v, v))
    RT = Q.T @ A[:m, :]
    R = np.triu(F[:m, :n][:, :m])
    np.testing.assert_allclose(np.triu(RT[:, :m]), R, atol=1e-12)
    np.testing.assert_allclose(np.tril(RT[:, :m], -1), 0, atol=1e-12)
    np.testing.assert_allclose(RT[:, m:], F[:m, m:], atol=1e-12)


def test_golden_vectors_regression(oracle):
    """committed vectors (made by tests/golden/make_golden.py from this oracle) must be reproduced bit for bit."""
    g = np.load(os.path.join(GOLDEN, "lse_golden.npz"))
    for name in ("ik", "rankdef"):
        lod, dims, n = g[f"{name}_lod"], g[f"{name}_dims"].tolist(), int(g[f"{name}_nvar"])
        r = oracle.lse_run(lod, dims, n, sens_obj=len(dims) - 1, ctr_type=g[f"{name}_types"])
        for key in ("x", "factor", "hh", "perm", "rank", "v", "lam", "sens"):
            np.testing.assert_array_equal(r[key], g[f"{name}_{key}"], err_msg=f"{name}:{key}")


def test_flop_and_byte_model():
    """SURVEY.md section 8(d) totals."""
    assert P.flop_model(40, [12] * 5)["total"] == 97402
    assert abs(P.flop_model(512, [256] * 4)["total"] - 2.653e8) < 1e5
    assert P.algorithmic_bytes(40, [12] * 5) == 20000
    assert P.algorithmic_bytes(512, [256] * 4) == 4206592
    assert P.algorithmic_bytes(40, [12] * 5, write_factor=True) == 40360


def test_least_norm_normal_equations_equals_givens_and_pinv(oracle):
    """solveLeastNorm_2 (lexlse.h:1138-1213) against solveLeastNorm_1 and against the minimum-norm solution numpy computes for an
    under-determined single-level problem (tolerance of the reference's MATLAB suites: 1e-10)."""
    n, dims = 40, [6] * 5
    lod = P.lse_batch(61, 4, n, dims)
    x1 = oracle.lse_run(lod, dims, n, solve_option=1)["x"]
    x2 = oracle.lse_run(lod, dims, n, solve_option=2)["x"]
    assert np.abs(x1 - x2).max() < 1e-10
    n, dims = 9, [5]
    lod = P.lse_batch(5, 3, n, dims)
    x2 = oracle.lse_run(lod, dims, n, solve_option=2)["x"]
    for b in range(3):
        A, rhs = lod[b, :n, :].T, lod[b, n, :]
        assert np.abs(np.linalg.pinv(A) @ rhs - x2[b]).max() < 1e-10


@pytest.mark.parametrize("reg_type", [1, 3, 4, 5, 8])
def test_regularization_single_level_is_damped_least_squares(oracle, reg_type):
    """One full-column-rank level: every regularization type of the family reduces to (A'A + mu^2 I)^-1 A'b
    (lexlse.h:1700-2236 with an empty null-space basis); tolerance of the reference's MATLAB suites."""
    n, dims, f = 6, [9], 0.7
    lod = P.lse_batch(11, 3, n, dims)
    x = oracle.lse_run(lod, dims, n, reg_type=reg_type, reg_factors=[f])["x"]
    for b in range(3):
        A, rhs = lod[b, :n, :].T, lod[b, n, :]
        assert np.abs(np.linalg.solve(A.T @ A + f * f * np.eye(n), A.T @ rhs) - x[b]).max() < 1e-10


def test_regularization_consistency_properties(oracle):
    """zero factors == no regularization (bitwise, utility.h:48-51 skips the damping); the two Tikhonov formulations
    (regularize_tikhonov_1 :1700 / _2 :2076, chosen by size for type 1, always _2 for type 8) agree; the variable factor
    (lexlse.h:281-311) with a huge threshold damps every level; solveLeastNorm_3 == solveLeastNorm_1."""
    n, dims = 12, [3, 4, 2]
    lod = P.lse_batch(7, 3, n, dims)
    x0 = oracle.lse_run(lod, dims, n)["x"]
    for t in (1, 3, 4, 5, 8, 9):
        np.testing.assert_array_equal(oracle.lse_run(lod, dims, n, reg_type=t, reg_factors=[0, 0, 0])["x"], x0)
    fac = [0.3, 0.5, 0.2]
    x1 = oracle.lse_run(lod, dims, n, reg_type=1, reg_factors=fac)["x"]
    x8 = oracle.lse_run(lod, dims, n, reg_type=8, reg_factors=fac)["x"]
    assert np.abs(x1 - x8).max() < 1e-10 and np.abs(x1 - x0).max() > 1e-3
    xv = oracle.lse_run(lod, dims, n, reg_type=1, reg_factors=fac, var_reg=1e6)["x"]
    assert np.abs(xv - x0).max() > 1e-3
    n, dims = 40, [6] * 5
    lod = P.lse_batch(61, 2, n, dims)
    a = oracle.lse_run(lod, dims, n, solve_option=1)["x"]
    c = oracle.lse_run(lod, dims, n, solve_option=3, reg_type=1, reg_factors=[0] * 5)["x"]
    assert np.abs(a - c).max() < 1e-10


def test_regularization_cg_variants_converge_to_the_direct_ones(oracle):
    """REGULARIZATION_TIKHONOV_CG / RT_NO_Z_CG (lexlse.h:2256-2554) run CGLS on the same stacked system the direct variants solve
    through the normal equations: with enough iterations they must agree (and the default 10 iterations stay close)."""
    n, dims, fac = 12, [3, 4, 2], [0.3, 0.5, 0.2]
    lod = P.lse_batch(7, 3, n, dims)
    x1 = oracle.lse_run(lod, dims, n, reg_type=1, reg_factors=fac)["x"]
    x2 = oracle.lse_run(lod, dims, n, reg_type=2, reg_factors=fac, cg_iters=300)["x"]
    x5 = oracle.lse_run(lod, dims, n, reg_type=5, reg_factors=fac)["x"]
    x6 = oracle.lse_run(lod, dims, n, reg_type=6, reg_factors=fac, cg_iters=300)["x"]
    assert np.abs(x1 - x2).max() < 1e-9 and np.abs(x5 - x6).max() < 1e-9
    assert np.abs(oracle.lse_run(lod, dims, n, reg_type=2, reg_factors=fac)["x"] - x1).max() < 1e-6


def removal_kat(n=4):
    """lambda_test.m-style hierarchy with inequalities that FORCE removals (the reference's fixture never deactivates):
         level 0:  x_k <= c_k               (general rows, all guessed ACTIVE at the upper bound)
         level 1:  k x_k = 1                (targets 1/k)
    With every bound active x = c and the multiplier of objective 1 for bound k is -k (k c_k - 1) (stationarity of
    1/2 sum (k x_k - 1)^2 + sum lambda_k (x_k - c_k)): negative exactly where c_k > 1/k — those bounds are wrongly active and must be
    removed, most negative first (lexlse.h:976-981); the others stay.  Closed form: x_k = min(c_k, 1/k)."""
    c = np.array([2.0, 0.25, 1.0, 0.1][:n])            # bounds 1 and 3 are slack at the optimum, 2 and 4 bind
    objs = [dict(A=np.eye(n), lb=np.full(n, -1e10), ub=c.copy())]
    objs.append(dict(A=np.diag(np.arange(1.0, n + 1)), lb=np.ones(n), ub=np.ones(n)))
    guess = [np.full(n, 2, np.uint8), np.zeros(n, np.uint8)]  # CTR_ACTIVE_UB on level 0
    k = np.arange(1.0, n + 1)
    lam = -k * (k * c - 1.0)
    expect_x = np.minimum(c, 1.0 / k)
    removed = [int(i) for i in np.argsort(lam) if lam[i] < 0]  # order of removal: most negative multiplier first
    return objs, guess, c, lam, expect_x, removed


def test_removal_path_closed_form(oracle):
    objs, guess, c, lam, expect_x, removed = removal_kat()
    r = oracle.lsi_run(4, objs, active_guess=guess)
    assert r["info"]["status"] == 0
    assert r["info"]["deactivations"] == len(removed) == 2 and r["info"]["activations"] == 0
    np.testing.assert_allclose(r["x"], expect_x, atol=1e-14)
    assert r["active"][0].tolist() == [0, 2, 0, 2]  # the slack bounds left the working set, the binding ones stayed at their upper bound
    # the multipliers themselves, at the first iteration (all bounds active): equality hierarchy  x = c  >  k x_k = 1
    eq = [dict(A=np.eye(4), lb=c, ub=c), dict(A=np.diag(np.arange(1.0, 5)), lb=np.ones(4), ub=np.ones(4))]
    x, L = oracle.lsi_lambda(4, eq)
    np.testing.assert_allclose(x, c, atol=1e-15)
    # column of objective 1: [multipliers of level 0; residual of level 1]; sign convention of the reference: lambda = -(stationarity multiplier)
    np.testing.assert_allclose(np.abs(L[:4, 1]), np.abs(lam), atol=1e-13)
    assert (np.sign(L[:4, 1]) == np.sign(L[0, 1]) * np.sign(lam) * np.sign(lam[0])).all()


# --- the reference's manual lexlse suite (interfaces/matlab-octave/tests/lexlse/test_lexlse_main.m) ----------------------------------
SUITE_N, SUITE_M, SUITE_R, SUITE_TOL = 30, (9, 8, 10, 6), (7, 6, 8, 5), 1e-10  # test_lexlse_main.m:16-20


def suite_solve(run, n, blocks, fixed, reg_type, factors, least_norm):
    """x of one `lexlse(obj, options)` call of the MEX front end (interfaces/matlab-octave/lexlse.cpp:144-199) through `run`"""
    dims = [b.shape[0] for b in blocks]
    kw = dict(reg_type=reg_type, reg_factors=np.asarray(factors, float), solve_option=least_norm)
    if fixed is not None:
        nf = len(fixed[0])
        idx, val = np.zeros((1, n), np.uint32), np.zeros((1, n))
        idx[0, :nf], val[0, :nf] = fixed
        kw.update(nfixed=np.array([nf], np.uint32), fixed_idx=idx, fixed_val=val)
    return run(P.stack_levels(blocks)[None], dims, n, **kw)["x"][0]


def suite_case(run, seed, least_norm, fixed_variables, reg_type, factors):
    """test_lexlse.m:13-20 + compare_results.m: the fixed-variable formulation against the general one (+ terminal objective); returns
    (|x1 - x2|, largest per-level residual difference)"""
    n = SUITE_N
    blocks, fixed = P.lexlse_suite_problem(seed, n, SUITE_M, SUITE_R, bool(fixed_variables))
    x1 = suite_solve(run, n, blocks, fixed, reg_type, factors[1:] if fixed_variables else factors, least_norm)
    gblocks, gfac = P.lexlse_suite_general_form(n, blocks, fixed, factors, least_norm)
    x2 = suite_solve(run, n, gblocks, None, reg_type, gfac, 0)
    err_r = max(np.linalg.norm((g[:, :n] @ x1 - g[:, n]) - (g[:, :n] @ x2 - g[:, n])) for g in gblocks)
    return np.linalg.norm(x1 - x2), err_r


@pytest.mark.parametrize("least_norm,fixed_variables,reg_type,factors", P.lexlse_suite_options(),
                         ids=["ln%d-fix%d-type%d-f%d" % (o[0], o[1], o[2], o[3][0]) for o in P.lexlse_suite_options()])
def test_reference_lexlse_suite(oracle, least_norm, fixed_variables, reg_type, factors):
    """The 42 option sets of test_lexlse_define.m on the suite's problem (n = 30, m = [9,8,10,6], r = [7,6,8,5], tol 1e-10): solving with
    fixed variables / a least-norm routine equals solving the general formulation with a terminal objective — for every regularization type
    of the suite, the experimental TIKHONOV_1 (7) included.  This is the reference's own acceptance test of the regularization family."""
    for seed in (1, 2, 3):
        err_x, err_r = suite_case(oracle.lse_run, seed, least_norm, fixed_variables, reg_type, factors)
        assert err_x <= SUITE_TOL and err_r <= SUITE_TOL, (seed, err_x, err_r)


def test_tikhonov_against_the_sequence_of_stacked_problems(oracle):
    """test_lexlse_main.m:22-49 with seq_lexls.m: Tikhonov regularization of every level == a sequence of two-level problems
    [upper levels kept at their values ; A_k stacked over mu_k I]; the per-level residuals agree to 1e-10 (x may differ, :41-43)."""
    n, mu = SUITE_N, (1.0, 2.0, 3.0, 4.0)
    for seed in (1, 2):
        blocks, _ = P.lexlse_suite_problem(seed, n, SUITE_M, SUITE_R, False)
        x1 = suite_solve(oracle.lse_run, n, blocks, None, 1, mu, 0)
        x, eye = None, np.hstack([np.eye(n), np.zeros((n, 1))])
        for k, blk in enumerate(blocks):
            reg = np.vstack([blk, eye * 1.0])
            reg[blk.shape[0]:, :n] *= mu[k]
            if k == 0:
                levels = [reg]
            else:
                C = np.vstack([b[:, :n] for b in blocks[:k]])
                levels = [np.hstack([C, (C @ x)[:, None]]), reg]
            x = oracle.lse_run(P.stack_levels(levels)[None], [l.shape[0] for l in levels], n)["x"][0]
        err = [np.linalg.norm((b[:, :n] @ x1 - b[:, n]) - (b[:, :n] @ x - b[:, n])) for b in blocks]
        assert np.linalg.norm(err) <= SUITE_TOL, err


def test_tikhonov_1_byproducts_satisfy_the_regularized_optimality_conditions(oracle):
    """REGULARIZATION_TIKHONOV_1 (the reference's experimental type 7, lexlse.h:1774-1886) has no fixture; what pins the restatement of
    its by-products: for a level k with a non-zero factor, residual_mu is the residual of level k at X_mu(:,k); the multipliers its
    ObjectiveSensitivity returns (:647-651, :688-690, initialize_rhs :1921-1959) make the Lagrangian of the regularized problem
    stationary, sum_{j<=k} A_j' lambda_j + mu_k^2 X_mu(:,k) = 0; with no free column left X_mu of the last level is x."""
    for (n, dims, ranks, fac) in [(12, [4, 5, 3], None, [0.0, 0.7, 0.4]), (12, [4, 5, 6], None, [0.5, 0.7, 0.4]), (10, [3, 3, 3], [2, 3, 2], [0.3, 0.0, 0.6])]:
        lod = (P.rank_deficient_problem(3, n, dims, ranks) if ranks else P.lse_problem(3, n, dims))[None]
        lv = P.levels_of(lod[0], dims)
        for k in range(len(dims)):
            r = oracle.lse_run(lod, dims, n, reg_type=7, reg_factors=fac, sens_obj=k)
            X, rm, lam = r["x_mu"][0], r["residual_mu"][0], r["lam"][0]
            if fac[k] == 0.0:  # a level the routine does not enter keeps X_mu = 0 and residual_mu = +rhs (the reference's \todo, :1768-1770)
                assert not X[k].any()
                continue
            off = sum(dims[:k])
            np.testing.assert_allclose(lv[k][0] @ X[k] - lv[k][1], rm[off:off + dims[k]], atol=1e-12)
            g, o = fac[k] ** 2 * X[k], 0
            for j in range(k + 1):
                g = g + lv[j][0].T @ lam[o:o + dims[j]]
                o += dims[j]
            np.testing.assert_allclose(g, 0, atol=1e-11)
        if sum(r["rank"][0]) == n:
            np.testing.assert_allclose(r["x"][0], r["x_mu"][0][-1], atol=1e-12)


@pytest.mark.parametrize("n,dims,ranks,nfix", [(15, [5, 5, 5, 5], [3, 3, 3, 3], 0), (15, [5, 5, 5, 5], [3, 3, 3, 3], 3), (6, [4, 4, 4], None, 0), (8, [5, 6], None, 2)])
def test_multipliers_are_the_dual_of_the_lexicographic_problem(oracle, n, dims, ranks, nfix):
    """What tests/implementation/test1.m checks of `d.lambda` (nObj = 4, nVar = 15, m = 5, r = m - 2, against lexqr_lambda.m), as the
    property that defines those multipliers: column k of Lambda holds the optimal residual of level k in its own block, and together with
    the multipliers of the levels above (and of the fixed variables) it makes the Lagrangian of level k's least-squares problem stationary,
    sum_{j<=k} A_j' lambda_j + E' lambda_fixed = 0."""
    for seed in range(5):
        lod = (P.rank_deficient_problem(seed, n, dims, ranks) if ranks else P.lse_problem(seed, n, dims))[None]
        lv = P.levels_of(lod[0], dims)
        kw, E = {}, np.zeros((0, n))
        if nfix:
            idx, val = np.zeros((1, n), np.uint32), np.zeros((1, n))
            idx[0, :nfix] = np.argsort(P.uniform(seed, n, 7))[:nfix]
            val[0, :nfix] = P.normal(seed, nfix, 8)
            kw = dict(nfixed=np.array([nfix], np.uint32), fixed_idx=idx, fixed_val=val)
            E = np.zeros((nfix, n))
            E[np.arange(nfix), idx[0, :nfix]] = 1.0
        for k in range(len(dims)):
            r = oracle.lse_run(lod, dims, n, sens_obj=k, **kw)
            x, lam = r["x"][0], r["lam"][0]
            g, o = E.T @ lam[:nfix], nfix
            for j in range(k + 1):
                g = g + lv[j][0].T @ lam[o:o + dims[j]]
                o += dims[j]
            np.testing.assert_allclose(g, 0, atol=1e-12)
            np.testing.assert_allclose(lam[o - dims[k]:o], lv[k][0] @ x - lv[k][1], atol=1e-12)
            assert not lam[o:].any()


def test_front_end_debug_structure(oracle):
    """`[x, info, v, as, d] = lexlsi(obj)` as tests/implementation/test1.m uses it (nObj = 4, nVar = 15, m = 5, r = m - 2, equalities through
    lb = ub): d.lexqr is the factor of the equality solver on the same problem, d.xStar = x, and the columns of d.lambda are the dual of the
    lexicographic problem (what test1.m compares with lexqr_lambda.m); on an inequality problem the working-set log has one entry per
    activation / deactivation and d.active_ctr lists the final working set in insertion order."""
    n, dims, ranks = 15, [5, 5, 5, 5], [3, 3, 3, 3]
    for seed in range(4):
        lod = P.rank_deficient_problem(seed, n, dims, ranks)
        lv = P.levels_of(lod, dims)
        r = oracle.lsi_run_debug(n, [dict(A=A, lb=b, ub=b) for A, b in lv])
        d, e = r["debug"], oracle.lse_run(lod[None], dims, n)
        assert r["info"]["factorizations"] == 1 and not d["working_set_log"] and len(d["active_ctr"]) == sum(dims)
        np.testing.assert_array_equal(r["x"], e["x"][0])
        np.testing.assert_array_equal(d["xStar"], e["x"][0])
        np.testing.assert_array_equal(d["lexqr"], e["factor"][0].T)
        np.testing.assert_array_equal(d["data"], lod.T)
        L = np.vstack(d["lambda"])
        for k in range(len(dims)):
            g, o = np.zeros(n), 0
            for j in range(len(dims)):
                g += lv[j][0].T @ L[o:o + dims[j], k]
                o += dims[j]
            np.testing.assert_allclose(g, 0, atol=1e-12)
            lo = sum(dims[:k])
            np.testing.assert_allclose(L[lo:lo + dims[k], k], lv[k][0] @ r["x"] - lv[k][1], atol=1e-12)
            assert not L[lo + dims[k]:, k].any()
    objs = P.lsi_problem(700, 20, [6, 5, 5, 6])
    r = oracle.lsi_run_debug(20, objs)
    d = r["debug"]
    assert len(d["working_set_log"]) == r["info"]["activations"] + r["info"]["deactivations"] > 0
    final = {(e["obj_index"], e["ctr_index"]): e["ctr_type"] for e in d["active_ctr"]}
    assert final == {(k, j): int(t) for k, a in enumerate(r["active"]) for j, t in enumerate(a) if t}
    assert d["lexqr"].shape == (16, 21)  # the simple bounds of objective 0 are fixed variables, not rows


def test_threaded_lsi_batch_timing_counts_the_same_factorizations(oracle):
    """oracle_lsi_time_batch (bench.py's CPU figure for configs[4]: the packed batch dealt out to host threads inside the library) runs the
    same solves as one lsi_run per instance: equal factorization counts, whatever the number of threads"""
    from lexls_amd import lexlsi, problems as P
    n, dims, batch = 12, [4, 5, 3], 9
    problems = [P.lsi_problem(4100 + i, n, dims) for i in range(batch)]
    packed = lexlsi.pack_batch(n, problems)
    expected = sum(oracle.lsi_run(n, objs)["info"]["factorizations"] for objs in problems)
    for threads in (1, 3):
        nf, seconds = oracle.lsi_time_batch(packed, None, None, threads)
        assert nf == expected and seconds > 0.0
