"""Inequality problems on the HIP path: the kept-on-host active-set driver (include/lexls/lexlsi.h) over the HIP
equality solver, through the C ABI (lexls_lsi_solve*), against (a) the reference's own fixture and (b) the same
driver over the CPU oracle.  Because the equality solver is bit-identical, the whole active-set trajectory must be:
same working set, same counters, same x (north_star: active-set indices exact, x within 1e-10)."""
import os

import numpy as np
import pytest

from lexls_amd import lexlsi, problems as P

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DAT = os.path.join(GOLDEN, "test_01.dat")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("use_as,use_x", [(0, 0), (1, 0), (1, 1)])
def test_reference_fixture_test_01(hip, oracle, use_as, use_x):
    """BASELINE.json configs[0] (its label is wrong: the file is a 5-level inequality hierarchy with simple bounds)."""
    d = lexlsi.lsi_solve_dat(DAT, 88, one_based=True, use_active_guess=bool(use_as), use_x_guess=bool(use_x))
    o = oracle.lsi_run_dat(DAT, True, bool(use_as), bool(use_x))
    assert d["info"]["status"] == 0
    assert np.abs(d["x"] - d["solution"]).max() < 1e-10         # the reference's stored answer (north_star: x* within 1e-10; measured 7e-13)
    assert d["info"] == o["info"]                                # same trajectory as the oracle-backed driver
    np.testing.assert_array_equal(d["x"], o["x"])


@pytest.mark.parametrize("seed", range(6))
def test_random_lsi_matches_oracle_driver(hip, oracle, seed):
    n, dims = 20, [6, 5, 5, 6]
    objs = P.lsi_problem(100 + seed, n, dims)
    d = lexlsi.lsi_solve(n, objs)
    o = oracle.lsi_run(n, objs)
    assert d["info"] == o["info"] and d["info"]["status"] == 0
    for a, b in zip(d["active"], o["active"]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(d["x"], o["x"])
    for a, b in zip(d["v"], o["v"]):
        np.testing.assert_array_equal(a, b)


def test_warm_start_from_neighbour(hip, oracle):
    """BASELINE.md C5 pattern: solve, perturb b, re-solve warm-started from the neighbour's active set and x."""
    n, dims = 40, [12] * 5
    base = lexlsi.lsi_solve(n, P.lsi_problem(7, n, dims))
    objs = P.lsi_problem(7, n, dims, perturb=0.05)
    guess = [np.where(a == 3, 0, a) for a in base["active"]]  # equalities are detected internally
    cold = lexlsi.lsi_solve(n, objs)
    warm = lexlsi.lsi_solve(n, objs, active_guess=guess, x0=base["x"])
    owarm = oracle.lsi_run(n, objs, active_guess=guess, x0=base["x"])
    assert warm["info"] == owarm["info"]
    np.testing.assert_array_equal(warm["x"], owarm["x"])
    assert warm["info"]["status"] == 0 and cold["info"]["status"] == 0
    assert warm["info"]["factorizations"] <= cold["info"]["factorizations"]
    np.testing.assert_allclose(warm["x"], cold["x"], atol=1e-8)


def test_no_simple_bounds_all_general(hip, oracle):
    n, dims = 10, [4, 4, 4]
    objs = P.lsi_problem(55, n, dims, simple_bounds=False)
    d, o = lexlsi.lsi_solve(n, objs), oracle.lsi_run(n, objs)
    assert d["info"] == o["info"]
    np.testing.assert_array_equal(d["x"], o["x"])


def test_lsi_error_behaviour(hip):
    n = 4
    bad = [dict(A=np.eye(2, n), lb=[1.0, 0.0], ub=[0.0, 1.0])]  # lb > ub -> the reference throws (lexlsi.h:430)
    with pytest.raises(Exception, match="Lower bound is greater than upper bound"):
        lexlsi.lsi_solve(n, bad)


def test_lock_step_batch_equals_individual_solves(hip, oracle):
    """BASELINE configs[4] in small: one batched device call per active-set round for all instances; every instance must
    end exactly where its stand-alone (oracle-backed) solve ends."""
    n, dims, batch = 20, [6, 5, 5, 6], 24
    problems = [P.lsi_problem(500 + b, n, dims) for b in range(batch)]
    r = lexlsi.lsi_batch_solve(n, problems)
    iters = []
    for b in range(batch):
        o = oracle.lsi_run(n, problems[b])
        assert r["info"][b] == o["info"], b
        np.testing.assert_array_equal(r["x"][b], o["x"])
        np.testing.assert_array_equal(r["active"][b], np.concatenate(o["active"]))
        iters.append(o["info"]["factorizations"])
    assert r["rounds"]["factorize_solve"] >= max(iters)  # lock step: at least as many factorize stages as the slowest instance needs ...
    assert r["rounds"]["factorize_solve"] < sum(iters)    # ... and far fewer than one per instance and iteration


def test_lock_step_batch_warm_started(hip, oracle):
    n, dims, batch = 40, [12] * 5, 8
    base = [oracle.lsi_run(n, P.lsi_problem(900 + b, n, dims)) for b in range(batch)]
    problems = [P.lsi_problem(900 + b, n, dims, perturb=0.05) for b in range(batch)]
    guess = [[np.where(a == 3, 0, a) for a in base[b]["active"]] for b in range(batch)]
    x0 = np.stack([base[b]["x"] for b in range(batch)])
    r = lexlsi.lsi_batch_solve(n, problems, active_guess=guess, x0=x0)
    for b in range(batch):
        o = oracle.lsi_run(n, problems[b], active_guess=guess[b], x0=base[b]["x"])
        assert r["info"][b] == o["info"], b
        np.testing.assert_array_equal(r["x"][b], o["x"])


def test_lock_step_device_gather_equals_host_staging(hip, oracle, monkeypatch):
    """the lock-step driver keeps the constraint data resident and gathers the active rows on the device
    (lexls_lse_gather_problem); assembling the same problems on the host and staging them over PCIe must give identical
    trajectories and bit-identical results."""
    n, dims, batch = 40, [12] * 5, 16
    pk = lexlsi.pack_batch(n, [P.lsi_problem(1300 + b, n, dims) for b in range(batch)])
    monkeypatch.setenv("LEXLS_LSI_RESIDENT", "0")  # (stage counts are compared: both runs with the host-side active-set logic)
    dev = lexlsi.lsi_batch_solve(n, pk)
    monkeypatch.setenv("LEXLS_LSI_HOST_STAGING", "1")
    host = lexlsi.lsi_batch_solve(n, pk)
    assert dev["info"] == host["info"] and dev["rounds"] == host["rounds"]
    np.testing.assert_array_equal(dev["x"], host["x"])
    np.testing.assert_array_equal(dev["active"], host["active"])
    np.testing.assert_array_equal(dev["v"], host["v"])


def test_gather_problem_rejects_out_of_range_rows(hip):
    import ctypes as C
    from lexls_amd import capi
    s = hip.BatchedLexLSE(2, 4, [3, 2])
    data = np.zeros((2, 5 * 6))
    L = capi.lib()
    capi.check(L.lexls_lse_set_constraint_data(s._h, data.ctypes.data_as(C.POINTER(C.c_double)), C.c_uint64(30)))
    src = np.zeros((2, 5), np.uint32)
    ld = np.zeros((2, 5), np.uint32)
    ld[1, 0] = 5
    src[1, 0] = 6  # 6 + (4+1)*5 = 31 >= 30
    rc = L.lexls_lse_gather_problem(s._h, src.ctypes.data_as(C.POINTER(C.c_uint32)), ld.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert rc != 0 and b"outside" in L.lexls_last_error()
    src[1, 0] = 4  # last element touched: 4 + 5*5 = 29
    capi.check(L.lexls_lse_gather_problem(s._h, src.ctypes.data_as(C.POINTER(C.c_uint32)), ld.ctypes.data_as(C.POINTER(C.c_uint32))))


@pytest.mark.parametrize("seed", range(4))
def test_deactivate_first_wrong_sign_matches_oracle_driver(hip, oracle, seed):
    """ParametersLexLSI::deactivate_first_wrong_sign routes the removal search through the 'collect all wrong-sign multipliers'
    overload (lexlse.h:511-602): multipliers from the device, scan on the host — trajectory identical to the oracle-backed driver."""
    n, dims = 20, [6, 5, 5, 6]
    objs = P.lsi_problem(300 + seed, n, dims)
    d = lexlsi.lsi_solve(n, objs, deactivate_first_wrong_sign=1)
    o = oracle.lsi_run(n, objs, deactivate_first_wrong_sign=1)
    assert d["info"] == o["info"]
    for a, b in zip(d["active"], o["active"]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(d["x"], o["x"])


@pytest.mark.parametrize("reg_type,factors", [(1, [0, 0.3, 0.2, 0.4]), (4, [0, 0.5, 0.5, 0.5]), (8, [0, 0.1, 0.3, 0.2])])
def test_lsi_with_regularization_matches_oracle_driver(hip, oracle, reg_type, factors):
    """lexlsi.cpp:527-625 also passes regularization factors / types and initial residuals: lexls_lsi_solve_ex.  The damped problems
    run on the device (generic kernel); trajectories and results equal the oracle-backed driver bit for bit."""
    n, dims = 20, [6, 5, 5, 6]
    objs = P.lsi_problem(410, n, dims)
    d = lexlsi.lsi_solve(n, objs, regularization_type=reg_type, regularization_factors=factors)
    o = oracle.lsi_run(n, objs, regularization_type=reg_type, regularization_factors=factors)
    assert d["info"] == o["info"] and d["info"]["status"] == 0
    np.testing.assert_array_equal(d["x"], o["x"])
    for a, b in zip(d["active"], o["active"]):
        np.testing.assert_array_equal(a, b)
    plain = lexlsi.lsi_solve(n, objs)
    assert np.abs(plain["x"] - d["x"]).max() > 1e-6  # the damping changed the solution


def test_lsi_initial_residuals_v0(hip, oracle):
    n, dims = 20, [6, 5, 5, 6]
    objs = P.lsi_problem(411, n, dims)
    base = oracle.lsi_run(n, objs)
    v0 = [0.5 * v for v in base["v"]]
    d = lexlsi.lsi_solve(n, objs, x0=0.5 * base["x"], v0=v0)
    o = oracle.lsi_run(n, objs, x0=0.5 * base["x"], v0=v0)
    assert d["info"] == o["info"]
    np.testing.assert_array_equal(d["x"], o["x"])


def test_lock_step_batch_with_initial_residuals_v0(hip, oracle):
    """lexls_lsi_batch_run takes the initial residuals the MEX front end passes to set_v0 (lexlsi.cpp:571-588), one block per instance"""
    n, dims, batch = 16, [5, 4, 6, 5], 6
    problems = [P.lsi_problem(1900 + b, n, dims) for b in range(batch)]
    base = [oracle.lsi_run(n, p) for p in problems]
    v0 = np.stack([0.5 * np.concatenate(o["v"]) for o in base])
    x0 = np.stack([0.5 * o["x"] for o in base])
    pk = lexlsi.pack_batch(n, problems)
    srv = lexlsi.LsiBatch(n, pk.dims, pk.types, batch)
    r = srv.run(pk, x0=x0, v0=v0)
    srv.close()
    for b in range(batch):
        o = oracle.lsi_run(n, problems[b], x0=x0[b], v0=np.split(v0[b], np.cumsum(dims)[:-1]))
        assert r["info"][b] == o["info"], b
        np.testing.assert_array_equal(r["x"][b], o["x"])


def test_matlab_style_front_end(hip, oracle):
    """lexls_amd.frontend mirrors the MEX call shapes lexlse(obj, options) / lexlsi(obj, options, active_set, x0, v0)."""
    from lexls_amd import frontend
    n, dims = 12, [3, 4, 2]
    lod = P.lse_batch(7, 1, n, dims)
    objs, r = [], 0
    for m in dims:
        objs.append(dict(A=lod[0, :n, r:r + m].T.copy(), b=lod[0, n, r:r + m].copy()))
        r += m
    x, info, v = frontend.lexlse(objs)
    ref = oracle.lse_run(lod, dims, n)
    np.testing.assert_array_equal(x, ref["x"][0])
    np.testing.assert_array_equal(np.concatenate(v), ref["v"][0])
    assert info["status"] == 0
    x1, _, _ = frontend.lexlse(objs, dict(get_least_norm_solution=1))
    x2, _, _ = frontend.lexlse(objs, dict(get_least_norm_solution=2))
    x3, _, _ = frontend.lexlse(objs, dict(get_least_norm_solution=3, regularization_type=1))
    assert np.abs(x1 - x2).max() < 1e-10 and np.abs(x1 - x3).max() < 1e-10
    xr, _, _ = frontend.lexlse(objs, dict(regularization_type=1, regularization_factors=[0.3, 0.5, 0.2]))
    np.testing.assert_array_equal(xr, oracle.lse_run(lod, dims, n, reg_type=1, reg_factors=[0.3, 0.5, 0.2])["x"][0])
    # fixed variables as the first "objective" (lexlse.cpp:148-164)
    xf, _, _ = frontend.lexlse([dict(var=[5, 1], b=[0.25, -0.5])] + objs)
    assert xf[5] == 0.25 and xf[1] == -0.5
    # lexlsi
    lobjs = P.lsi_problem(412, 20, [6, 5, 5, 6])
    x, info, v, active = frontend.lexlsi(lobjs, dict(tol_feasibility=1e-13))
    o = oracle.lsi_run(20, lobjs)
    np.testing.assert_array_equal(x, o["x"])
    assert info["number_of_factorizations"] == o["info"]["factorizations"] and len(active) == 4


def test_cpp_drop_in_example(hip, tmp_path):
    """examples/drop_in.cpp is what a user of the reference writes (LexLSI / LexLSE classes, .dat reader), compiled with plain g++
    against include/ and linked with liblexls_hip.so: it must reproduce the #Solution block of the reference's own fixture."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "lexls_amd", "csrc")
    exe = str(tmp_path / "drop_in")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "drop_in.cpp"),
                           "-L" + libdir, "-llexls_hip", "-Wl,-rpath," + libdir, "-o", exe])
    r = subprocess.run([exe, os.path.join(GOLDEN, "test_01.dat")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "LexLSI: status 0" in r.stdout and "LexLSE: ranks 3 1" in r.stdout, r.stdout
    assert "multipliers of the one-argument overload agree" in r.stdout, r.stdout


def test_cpp_resolve_active_set_reproduces_factor_and_solution(hip, tmp_path):
    """the property of the reference's tests/test_numerical_error.cpp:83-137 through the same classes: the equality problem of LexLSI's
    last iteration, handed to a fresh LexLSE (fixed variables, active rows in working-set order, constraint types), gives the same
    factor bit for bit (the reference sees small differences there, :5-22) and the same x up to the rounding of the final step."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "lexls_amd", "csrc")
    exe = str(tmp_path / "resolve_active_set")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "resolve_active_set.cpp"),
                           "-L" + libdir, "-llexls_hip", "-Wl,-rpath," + libdir, "-o", exe])
    r = subprocess.run([exe, os.path.join(GOLDEN, "test_01.dat")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "error(data) = 0.000e+00, error(lexqr) = 0.000e+00" in r.stdout, r.stdout


def test_persistent_batch_object_is_stateless_between_runs(hip, oracle):
    """lexls_lsi_batch_create / _run / _destroy: one batch object serves successive problem sets (different data, warm starts, with and
    without regularization) and every run equals the one-shot call — nothing of a previous run (working-set marks, fixed variables,
    regularization factors, constraint data) leaks into the next."""
    n, dims, batch = 16, [5, 4, 6, 5], 10
    A = [P.lsi_problem(900 + b, n, dims) for b in range(batch)]
    B = [P.lsi_problem(900 + b, n, dims, perturb=0.3) for b in range(batch)]
    pkA = lexlsi.pack_batch(n, A)
    srv = lexlsi.LsiBatch(n, pkA.dims, pkA.types, batch)
    r1 = srv.run(A)
    guess = np.where(r1["active"] == 3, 0, r1["active"]).astype(np.uint8)
    r2 = srv.run(B, active_guess=guess, x0=r1["x"])
    r3 = srv.run(A, regularization_type=1, regularization_factors=[0, 0.3, 0.2, 0.4])
    r4 = srv.run(A)
    srv.close()
    o1 = lexlsi.lsi_batch_solve(n, A)
    o2 = lexlsi.lsi_batch_solve(n, B, active_guess=guess, x0=r1["x"])
    o3 = lexlsi.lsi_batch_solve(n, A, regularization_type=1, regularization_factors=[0, 0.3, 0.2, 0.4])
    for r, o in ((r1, o1), (r2, o2), (r3, o3), (r4, o1)):
        assert r["info"] == o["info"]
        np.testing.assert_array_equal(r["x"], o["x"])
        np.testing.assert_array_equal(r["active"], o["active"])
        np.testing.assert_array_equal(r["v"], o["v"])
    for b in range(batch):  # and the one-shot call equals the stand-alone oracle-backed driver
        ob = oracle.lsi_run(n, A[b])
        assert o1["info"][b] == ob["info"]
        np.testing.assert_array_equal(o1["x"][b], ob["x"])
    with pytest.raises(ValueError):
        lexlsi.LsiBatch(n, pkA.dims, pkA.types, batch + 1).run(A)


def test_lock_step_batch_with_cycling_handling_enabled(hip, oracle):
    """cycling handling (cycling.h:32-65) relaxes bounds in the HOST copy of the constraint data, so a run with it enabled must not gather
    rows from the resident device copy: the driver assembles such runs on the host — same trajectories as the oracle-backed driver, and
    the same batch object goes back to the device gather afterwards."""
    n, dims, batch = 14, [4, 5, 5, 4], 9
    problems = [P.lsi_problem(1300 + b, n, dims) for b in range(batch)]
    pk = lexlsi.pack_batch(n, problems)
    srv = lexlsi.LsiBatch(n, pk.dims, pk.types, batch)
    r = srv.run(pk, cycling_handling_enabled=1, cycling_max_counter=3, cycling_relax_step=1e-6)
    plain = srv.run(pk)
    srv.close()
    for b in range(batch):
        o = oracle.lsi_run(n, problems[b], cycling_handling_enabled=1, cycling_max_counter=3, cycling_relax_step=1e-6)
        assert r["info"][b] == o["info"], b
        np.testing.assert_array_equal(r["x"][b], o["x"])
        o0 = oracle.lsi_run(n, problems[b])
        assert plain["info"][b] == o0["info"], b
        np.testing.assert_array_equal(plain["x"][b], o0["x"])


@pytest.mark.parametrize("simple_bounds", [True, False])
def test_device_side_step_matches_host_step(hip, oracle, monkeypatch, simple_bounds):
    """SURVEY 8(f) item 1, LEXLS_LSI_DEVICE_STEP=1: from the second iteration on A*dx, the ratio test (first minimum in working-set scan
    order, lexlsi.h:1006-1029) and the update of x / v / A*x run on the device next to the equality solve; the host keeps the working
    sets.  Same trajectories, same x and v bit for bit as the host step and as the oracle-backed driver — cold and warm-started."""
    n, dims, batch = 18, [6, 5, 7, 4], 14
    problems = [P.lsi_problem(1500 + b, n, dims, simple_bounds=simple_bounds) for b in range(batch)]
    pk = lexlsi.pack_batch(n, problems)
    host = lexlsi.lsi_batch_solve(n, pk)
    monkeypatch.setenv("LEXLS_LSI_DEVICE_STEP", "1")
    srv = lexlsi.LsiBatch(n, pk.dims, pk.types, batch)
    monkeypatch.delenv("LEXLS_LSI_DEVICE_STEP")
    dev = srv.run(pk)
    assert srv.stats()["device_step"] > 0
    guess = np.where(dev["active"] == 3, 0, dev["active"]).astype(np.uint8)
    pert = lexlsi.pack_batch(n, [P.lsi_problem(1500 + b, n, dims, simple_bounds=simple_bounds, perturb=0.4) for b in range(batch)])
    dev_w = srv.run(pert, active_guess=guess, x0=dev["x"])
    srv.close()
    host_w = lexlsi.lsi_batch_solve(n, pert, active_guess=guess, x0=dev["x"])
    for d, h in ((dev, host), (dev_w, host_w)):
        assert d["info"] == h["info"]
        np.testing.assert_array_equal(d["x"], h["x"])
        np.testing.assert_array_equal(d["v"], h["v"])
        np.testing.assert_array_equal(d["active"], h["active"])
    for b in range(batch):
        o = oracle.lsi_run(n, problems[b])
        assert dev["info"][b] == o["info"]
        np.testing.assert_array_equal(dev["x"][b], o["x"])
        np.testing.assert_array_equal(dev["v"][b], np.concatenate(o["v"]))


@pytest.mark.parametrize("simple_bounds", [True, False])
def test_resident_iterations_match_host_driver(hip, oracle, monkeypatch, simple_bounds):
    """Default since round 2 (LEXLS_LSI_RESIDENT=0 switches it off): after phase 1 the instances iterate on the device — step, ratio test,
    ONE working-set change by the rules of workingset.h (swap-with-last / ordered erase), counters and the next equality problem in
    lsi_iterate_kernel — and the host only polls.  Same trajectories (info: status, iterations, activations, deactivations,
    factorizations, total rank), same x / v / working sets bit for bit as the host-side driver and the oracle-backed one, cold and
    warm-started; a factorization limit stops the instances at the same iteration."""
    n, dims, batch = 18, [6, 5, 7, 4], 14
    problems = [P.lsi_problem(1500 + b, n, dims, simple_bounds=simple_bounds) for b in range(batch)]
    pk = lexlsi.pack_batch(n, problems)
    srv = lexlsi.LsiBatch(n, pk.dims, pk.types, batch)
    res = srv.run(pk)
    assert srv.stats()["device_step"] > 0  # stages whose step ran on the device
    guess = np.where(res["active"] == 3, 0, res["active"]).astype(np.uint8)
    pert = lexlsi.pack_batch(n, [P.lsi_problem(1500 + b, n, dims, simple_bounds=simple_bounds, perturb=0.4) for b in range(batch)])
    res_w = srv.run(pert, active_guess=guess, x0=res["x"])
    res_lim = srv.run(pk, max_number_of_factorizations=4)
    srv.close()
    monkeypatch.setenv("LEXLS_LSI_RESIDENT", "0")
    host = lexlsi.lsi_batch_solve(n, pk)
    host_w = lexlsi.lsi_batch_solve(n, pert, active_guess=guess, x0=res["x"])
    host_lim = lexlsi.lsi_batch_solve(n, pk, max_number_of_factorizations=4)
    assert any(i["status"] == 2 for i in host_lim["info"]) and any(i["deactivations"] > 0 for i in host["info"])
    for d, h in ((res, host), (res_w, host_w), (res_lim, host_lim)):
        assert d["info"] == h["info"]
        np.testing.assert_array_equal(d["x"], h["x"])
        np.testing.assert_array_equal(d["v"], h["v"])
        np.testing.assert_array_equal(d["active"], h["active"])
    for b in range(batch):
        o = oracle.lsi_run(n, problems[b])
        assert res["info"][b] == o["info"]
        np.testing.assert_array_equal(res["x"][b], o["x"])
        np.testing.assert_array_equal(res["v"][b], np.concatenate(o["v"]))
        np.testing.assert_array_equal(res["active"][b], np.concatenate(o["active"]))


@pytest.mark.parametrize("simple_bounds", [True, False])
def test_resident_iterations_with_and_without_prefix_reuse(hip, oracle, monkeypatch, simple_bounds):
    """the resident iterations tell every factorization which leading levels are unchanged since the previous one (the level of
    the constraint that was activated / removed), and those levels are read back instead of factorized (lexls_lse_set_prefix_reuse).  The
    reference refactorizes everything (README.md:14).  Same trajectories, x, v and working sets, bit for bit, with the reuse on (default)
    and off (LEXLS_LSI_PREFIX_REUSE=0), IK-sized problems with several general levels, cold and warm-started; and as the oracle-backed driver."""
    n, dims, batch = 40, [12, 12, 12, 12, 12], 24
    problems = [P.lsi_problem(4100 + b, n, dims, simple_bounds=simple_bounds) for b in range(batch)]
    pk = lexlsi.pack_batch(n, problems)
    pert = lexlsi.pack_batch(n, [P.lsi_problem(4100 + b, n, dims, simple_bounds=simple_bounds, perturb=0.9) for b in range(batch)])
    runs = {}
    for reuse in ("1", "0"):
        monkeypatch.setenv("LEXLS_LSI_PREFIX_REUSE", reuse)
        srv = lexlsi.LsiBatch(n, pk.dims, pk.types, batch)  # (the switch is read when the batch object is created)
        cold = srv.run(pk)
        guess = np.where(cold["active"] == 3, 0, cold["active"]).astype(np.uint8)
        warm = srv.run(pert, active_guess=guess, x0=cold["x"])
        assert srv.stats()["device_step"] > 0
        srv.close()
        runs[reuse] = (cold, warm)
    for a, b2 in zip(runs["1"], runs["0"]):
        assert a["info"] == b2["info"]
        np.testing.assert_array_equal(a["x"], b2["x"])
        np.testing.assert_array_equal(a["v"], b2["v"])
        np.testing.assert_array_equal(a["active"], b2["active"])
    assert max(i["factorizations"] for i in runs["1"][0]["info"]) > 5  # (the reuse had factorizations to act on)
    for b in range(0, batch, 5):
        o = oracle.lsi_run(n, problems[b])
        assert runs["1"][0]["info"][b] == o["info"]
        np.testing.assert_array_equal(runs["1"][0]["x"][b], o["x"])
        np.testing.assert_array_equal(runs["1"][0]["active"][b], np.concatenate(o["active"]))


def test_resident_iterations_beyond_the_wave_kernel_shapes(hip, oracle):
    """resident iterations with equality problems the register-resident wave kernel does not take (nVar + 1 > 64: the generic kernel, which
    reads the assembled problem — the row gather is then its own launch again) and with more than 64 constraints per instance."""
    n, dims, batch = 70, [9, 30, 28, 12], 6
    problems = [P.lsi_problem(3100 + b, n, dims) for b in range(batch)]
    srv = lexlsi.LsiBatch(n, *[getattr(lexlsi.pack_batch(n, problems[:1]), k) for k in ("dims", "types")], batch)
    r = srv.run(problems)
    assert srv.stats()["device_step"] > 0
    srv.close()
    for b in range(batch):
        o = oracle.lsi_run(n, problems[b])
        assert r["info"][b] == o["info"], b
        np.testing.assert_array_equal(r["x"][b], o["x"])
        np.testing.assert_array_equal(r["v"][b], np.concatenate(o["v"]))
        np.testing.assert_array_equal(r["active"][b], np.concatenate(o["active"]))


@pytest.mark.parametrize("groups", [2, 3])
def test_lock_step_groups_take_turns(hip, oracle, monkeypatch, groups):
    """large batches are split into groups that take turns on their own streams (default from 512 instances on); forced here on a small
    batch, with an uneven split: every instance ends where its stand-alone oracle-backed solve ends, and the worker pool is exercised
    (>= 128 instances per group)."""
    n, dims, batch = 10, [4, 3, 4], 128 * groups + 5
    problems = [P.lsi_problem(1700 + (b % 40), n, dims, perturb=0.01 * (b // 40)) for b in range(batch)]
    monkeypatch.setenv("LEXLS_LSI_GROUPS", str(groups))
    srv = lexlsi.LsiBatch(n, *[getattr(lexlsi.pack_batch(n, problems[:1]), k) for k in ("dims", "types")], batch)
    r = srv.run(problems)
    assert srv.stats()["groups"] == groups
    srv.close()
    for b in list(range(0, batch, 17)) + [batch - 1]:
        o = oracle.lsi_run(n, problems[b])
        assert r["info"][b] == o["info"], b
        np.testing.assert_array_equal(r["x"][b], o["x"])
        np.testing.assert_array_equal(r["active"][b], np.concatenate(o["active"]))


def test_speculative_removal_search(hip, oracle, monkeypatch):
    """default since round 2 (LEXLS_LSI_SPECULATIVE_SENS=0 switches it off): every factorization of a lock-step batch is followed by its removal
    search in the same stage and the result is used when the step turns out not to be blocked — fewer stages, the same trajectories."""
    n, dims, batch = 16, [5, 4, 6, 5], 12
    problems = [P.lsi_problem(2100 + b, n, dims) for b in range(batch)]
    monkeypatch.setenv("LEXLS_LSI_SPECULATIVE_SENS", "0")
    plain = lexlsi.lsi_batch_solve(n, problems)
    monkeypatch.setenv("LEXLS_LSI_SPECULATIVE_SENS", "1")
    spec = lexlsi.lsi_batch_solve(n, problems)
    assert spec["rounds"]["factorize_solve"] <= plain["rounds"]["factorize_solve"] and spec["rounds"]["sensitivity"] >= spec["rounds"]["factorize_solve"]
    assert spec["info"] == plain["info"]
    np.testing.assert_array_equal(spec["x"], plain["x"])
    np.testing.assert_array_equal(spec["v"], plain["v"])
    np.testing.assert_array_equal(spec["active"], plain["active"])
    for b in range(0, batch, 5):
        o = oracle.lsi_run(n, problems[b])
        assert spec["info"][b] == o["info"]
        np.testing.assert_array_equal(spec["x"][b], o["x"])


@pytest.mark.parametrize("reg_type", [1, 7])
def test_lock_step_batch_with_regularization(hip, oracle, reg_type):
    """lexls_lsi_batch_solve_ex: the damped hierarchies of a batch run lock-step (host-side active-set logic, regularized equality
    kernels); every instance ends exactly where its stand-alone oracle-backed solve with the same regularization ends.  Type 7 is the
    reference's experimental TIKHONOV_1, whose removal search uses the multipliers of the regularized problem (lexlse.h:647-651)."""
    n, dims, batch = 20, [6, 5, 5, 6], 12
    factors = [0, 0.3, 0.2, 0.4]
    problems = [P.lsi_problem(700 + b, n, dims) for b in range(batch)]
    r = lexlsi.lsi_batch_solve(n, problems, regularization_type=reg_type, regularization_factors=factors)
    plain = lexlsi.lsi_batch_solve(n, problems)
    assert np.abs(plain["x"] - r["x"]).max() > 1e-6
    for b in range(batch):
        o = oracle.lsi_run(n, problems[b], regularization_type=reg_type, regularization_factors=factors)
        assert r["info"][b] == o["info"], b
        np.testing.assert_array_equal(r["x"][b], o["x"])
        np.testing.assert_array_equal(r["active"][b], np.concatenate(o["active"]))


def test_removal_path_closed_form_on_the_device(hip, oracle):
    """tests/test_oracle_golden.py::removal_kat: wrongly active bounds must leave the working set, most negative multiplier first
    (ObjectiveSensitivity -> REMOVE, lexlsi.h:1204-1225) — closed-form x and working set, same trajectory as the oracle-backed driver"""
    from test_oracle_golden import removal_kat
    objs, guess, c, lam, expect_x, removed = removal_kat()
    d = lexlsi.lsi_solve(4, objs, active_guess=guess)
    o = oracle.lsi_run(4, objs, active_guess=guess)
    assert d["info"]["status"] == 0 and d["info"]["deactivations"] == 2 and d["info"]["activations"] == 0
    assert d["info"] == o["info"]
    np.testing.assert_allclose(d["x"], expect_x, atol=1e-14)
    np.testing.assert_array_equal(d["x"], o["x"])
    assert d["active"][0].tolist() == [0, 2, 0, 2]
    # the removal search itself on the equality solver: bounds active at UB, objective 1's multipliers, first candidate = most negative
    lod = np.zeros((1, 5, 8))
    lod[0, :4, :4] = np.eye(4)
    lod[0, 4, :4] = c
    lod[0, :4, 4:] = np.diag(np.arange(1.0, 5))
    lod[0, 4, 4:] = 1.0
    s = hip.BatchedLexLSE(1, 4, [4, 4])
    s.setProblem(lod)
    s.setCtrType(np.array([[2, 2, 2, 2, 3, 3, 3, 3]], np.uint8))
    s.factorize_solve()
    found, ctr, obj, maxabs = s.ObjectiveSensitivity(1)
    assert bool(found[0]) and int(obj[0]) == 0 and int(ctr[0]) == removed[0]
    assert abs(abs(float(maxabs[0])) - abs(lam[removed[0]])) < 1e-13
    np.testing.assert_allclose(np.abs(s.getWorkspace()[0, :4]), np.abs(lam), atol=1e-13)


@pytest.mark.parametrize("resident", [True, False])
def test_config5_full_size_lock_step_batch(hip, oracle, monkeypatch, resident):
    """BASELINE.json configs[4] at full size: 1024 instances (n = 40, 5 x 12, level 0 simple bounds) in ONE lock-step batch object,
    warm-started from the unperturbed neighbour with right-hand sides perturbed by 0.9 N(0,1) (~30 factorizations per instance) — with the
    iterations resident on the device (default) and with the active-set logic on the host (LEXLS_LSI_RESIDENT=0: automatic group split).
    Every instance solved; 64 sampled instances equal the oracle-backed driver bit for bit (x, v, working set, counters)."""
    if not resident:
        monkeypatch.setenv("LEXLS_LSI_RESIDENT", "0")
    n, dims, batch = 40, [12] * 5, 1024
    base = lexlsi.pack_batch(n, [P.lsi_problem(20260500 + b, n, dims) for b in range(batch)])
    pert = lexlsi.pack_batch(n, [P.lsi_problem(20260500 + b, n, dims, perturb=0.9) for b in range(batch)])
    srv = lexlsi.LsiBatch(n, base.dims, base.types, batch)
    try:
        cold = srv.run(base)
        assert all(i["status"] == 0 for i in cold["info"])
        guess = np.where(cold["active"] == 3, 0, cold["active"]).astype(np.uint8)
        warm = srv.run(pert, active_guess=guess, x0=cold["x"])
        stats = srv.stats()
    finally:
        srv.close()
    assert all(i["status"] == 0 for i in warm["info"])
    if resident:
        assert stats["device_step"] >= max(i["factorizations"] for i in warm["info"]) - 1  # all but the phase-1 stage ran without the host
    else:
        assert stats["groups"] >= 2  # host logic: batches from 512 instances on take turns in groups
    f = np.array([i["factorizations"] for i in warm["info"]])
    assert 20.0 <= f.mean() <= 40.0, f.mean()  # the workload BASELINE.md C5 describes: ~30 factorizations per instance
    assert sum(i["deactivations"] for i in warm["info"]) > batch  # the removal path is exercised throughout
    cuts = np.cumsum(base.dims)[:-1]
    for b in range(0, batch, 16):  # 64 sampled instances against the oracle-backed driver
        objs = P.lsi_problem(20260500 + b, n, dims, perturb=0.9)
        o = oracle.lsi_run(n, objs, active_guess=np.split(guess[b], cuts), x0=cold["x"][b])
        assert warm["info"][b] == o["info"], b
        np.testing.assert_array_equal(warm["x"][b], o["x"], err_msg=str(b))
        np.testing.assert_array_equal(warm["active"][b], np.concatenate(o["active"]), err_msg=str(b))
        np.testing.assert_array_equal(warm["v"][b], np.concatenate(o["v"]), err_msg=str(b))


def test_persistent_launch_equals_the_stage_by_stage_path(hip, monkeypatch):
    """the resident iterations as ONE persistent launch (lsi_fused_impl.h: per instance l-QR -> step -> removal search behind an unblocked
    step -> working-set change, until the instance stops) against the three launches per lock-step stage (LEXLS_LSI_NO_FUSED=1): same x, v,
    working sets and counters, bit for bit — on the IK shape the persistent kernel is instantiated for, ragged working sets, a batch that is
    not a multiple of anything"""
    n, dims, batch = 40, [12] * 5, 197
    base = lexlsi.pack_batch(n, [P.lsi_problem(20267700 + b, n, dims) for b in range(batch)])
    pert = lexlsi.pack_batch(n, [P.lsi_problem(20267700 + b, n, dims, perturb=0.9) for b in range(batch)])
    out = {}
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("LEXLS_LSI_NO_FUSED", raising=False)
        else:
            monkeypatch.setenv("LEXLS_LSI_NO_FUSED", "1")
        srv = lexlsi.LsiBatch(n, base.dims, base.types, batch)
        try:
            cold = srv.run(base)
            guess = np.where(cold["active"] == 3, 0, cold["active"]).astype(np.uint8)
            warm = srv.run(pert, active_guess=guess, x0=cold["x"])
            out[fused] = (cold, warm, srv.stats())
        finally:
            srv.close()
    for which in (0, 1):
        a, b = out[True][which], out[False][which]
        assert all(i["status"] == 0 for i in a["info"])
        assert [i for i in a["info"]] == [i for i in b["info"]]
        np.testing.assert_array_equal(a["x"], b["x"])
        np.testing.assert_array_equal(a["v"], b["v"])
        np.testing.assert_array_equal(a["active"], b["active"])
    # the persistent launch counts the iterations its longest-running instance needed; the stage-by-stage path enqueues stages in chunks of eight
    longest = max(i["factorizations"] for i in out[True][1]["info"])
    assert longest - 1 <= out[True][2]["device_step"] <= out[False][2]["device_step"]


@pytest.mark.parametrize("n,dims,simple_bounds,limit", [(40, [12] * 5, True, 7), (24, [6, 9, 4, 11], False, 0), (55, [16, 14, 16, 12], True, 0), (12, [5, 7], False, 3)])
def test_persistent_launch_other_shapes_and_limits(hip, oracle, monkeypatch, n, dims, simple_bounds, limit):
    """the persistent launch on its three instantiations (41 x 12 exact, 41 x 12, 64 x 16), cold starts, a factorization limit that stops instances
    inside the launch (MAX_NUMBER_OF_FACTORIZATIONS_EXCEEDED): every instance against the oracle-backed driver, and against the stage path"""
    batch = 37
    problems = [P.lsi_problem(20268800 + b, n, dims, simple_bounds=simple_bounds) for b in range(batch)]
    kw = {"max_number_of_factorizations": limit} if limit else {}
    out = {}
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("LEXLS_LSI_NO_FUSED", raising=False)
        else:
            monkeypatch.setenv("LEXLS_LSI_NO_FUSED", "1")
        out[fused] = lexlsi.lsi_batch_solve(n, problems, **kw)
    a, b = out[True], out[False]
    assert [i for i in a["info"]] == [i for i in b["info"]]
    np.testing.assert_array_equal(a["x"], b["x"])
    np.testing.assert_array_equal(a["v"], b["v"])
    np.testing.assert_array_equal(a["active"], b["active"])
    if limit:
        assert any(i["status"] != 0 for i in a["info"])  # the limit bites
    for k in range(0, batch, 6):
        o = oracle.lsi_run(n, problems[k], **kw)
        assert a["info"][k] == o["info"], k
        np.testing.assert_array_equal(a["x"][k], o["x"], err_msg=str(k))
        np.testing.assert_array_equal(a["active"][k], np.concatenate(o["active"]), err_msg=str(k))


def test_front_end_debug_output(hip, oracle):
    """The fifth output of the MEX front end through lexls_lsi_solve_debug (`frontend.lexlsi(..., debug=True)`): working-set log with its
    step lengths / multipliers, final working set in order, the multiplier matrices of getLambda, factor, data and xStar of the last
    equality problem — identical to the oracle-backed driver's; cold start, a warm start that has to remove constraints, simple bounds,
    and the by-products of the experimental regularization type 7."""
    from lexls_amd import frontend
    n, dims = 20, [6, 5, 5, 6]

    def check(objs, **kw):
        okw = dict(kw)
        opt = okw.pop("options", {})
        x, info, v, act, d = frontend.lexlsi(objs, opt, okw.get("active_set"), okw.get("x0"), debug=True)
        o = oracle.lsi_run_debug(n, objs, active_guess=okw.get("active_set"), x0=okw.get("x0"), regularization_factors=opt.get("regularization_factors"),
                                 **{k: val for k, val in opt.items() if k != "regularization_factors"})
        od = o["debug"]
        assert info["number_of_factorizations"] == o["info"]["factorizations"] and info["status"] == o["info"]["status"]
        np.testing.assert_array_equal(x, o["x"])
        assert d["working_set_log"] == od["working_set_log"] and d["active_ctr"] == od["active_ctr"]
        # rows of the equality problem = active constraints of the general objectives (the rows beyond them are stale in the reference too)
        m = sum(1 for e in d["active_ctr"] if "A" in objs[e["obj_index"]])
        for key in ("lexqr", "data", "xStar") + (("X_mu", "X_mu_rhs", "residual_mu") if "X_mu" in od else ()):
            rows = slice(0, m) if key in ("lexqr", "data", "residual_mu") else slice(None)
            np.testing.assert_array_equal(d[key][rows], od[key][rows], err_msg=key)
        for a, b in zip(d["lambda"], od["lambda"]):
            np.testing.assert_array_equal(a, b)
        return info, d

    for seed in (700, 701, 702):
        objs = P.lsi_problem(seed, n, dims)
        info, d = check(objs)
        assert len(d["working_set_log"]) == info["number_of_activations"] + info["number_of_deactivations"]
        # warm start from a wrong guess: everything active at the upper bound -> removals in the log
        guess = [np.full(m, 2, np.uint8) for m in dims]
        info, d = check(objs, active_set=guess)
        assert info["number_of_deactivations"] > 0 and any(e["ctr_type"] == 0 for e in d["working_set_log"])
    objs = [o for o in P.lsi_problem(703, n, dims) if "A" in o]  # general objectives only
    check(objs)
    info, d = check(P.lsi_problem(704, n, dims), options=dict(regularization_type=7, regularization_factors=[0, 0.3, 0.2, 0.4], max_number_of_factorizations=40))
    assert d["X_mu"].shape == (n, 3) and d["residual_mu"].shape == (16,)


def test_batch_with_deactivate_first_wrong_sign(hip, oracle):
    """lexls_lsi_batch_solve with ParametersLexLSI::deactivate_first_wrong_sign (lexlsi.h:1089-1103): no lock-step form — the instances run
    one after the other through the single-problem driver; every instance ends where its oracle-backed solve ends."""
    n, dims, batch = 20, [6, 5, 5, 6], 6
    problems = [P.lsi_problem(800 + b, n, dims) for b in range(batch)]
    guess = [[np.full(m, 2, np.uint8) for m in dims] for _ in range(batch)]  # everything active at the upper bound: removals needed
    r = lexlsi.lsi_batch_solve(n, problems, active_guess=guess, deactivate_first_wrong_sign=1)
    for b in range(batch):
        o = oracle.lsi_run(n, problems[b], active_guess=guess[b], deactivate_first_wrong_sign=1)
        assert r["info"][b] == o["info"], b
        np.testing.assert_array_equal(r["x"][b], o["x"])
        np.testing.assert_array_equal(r["active"][b], np.concatenate(o["active"]))
    assert any(i["deactivations"] > 0 for i in r["info"])


def test_lock_step_batch_with_a_deep_hierarchy(hip, oracle):
    """seven levels of an IK-sized problem (more than 64 constraint rows in all): the equality problems of the batch run on the left-looking
    kernels; every instance ends where its oracle-backed solve ends"""
    n, dims, batch = 30, [8, 12, 12, 12, 12, 12, 10], 10
    problems = [P.lsi_problem(1200 + b, n, dims) for b in range(batch)]
    r = lexlsi.lsi_batch_solve(n, problems)
    for b in range(batch):
        o = oracle.lsi_run(n, problems[b])
        assert r["info"][b] == o["info"], b
        np.testing.assert_array_equal(r["x"][b], o["x"])
        np.testing.assert_array_equal(r["active"][b], np.concatenate(o["active"]))
        np.testing.assert_array_equal(r["v"][b], np.concatenate(o["v"]))


def test_lock_step_batch_with_42_to_48_columns(hip, oracle):
    """43..48 columns (n = 42..47): the rounds of a lock-step batch use the four-per-wavefront kernel behind a gather launch instead of the
    64-column register-resident instantiation; trajectories and results identical to the oracle-backed driver"""
    for n, dims in ((47, [10, 12, 12, 11]), (43, [12, 9, 12])):
        batch = 9
        problems = [P.lsi_problem(1300 + b, n, dims) for b in range(batch)]
        r = lexlsi.lsi_batch_solve(n, problems)
        for b in range(batch):
            o = oracle.lsi_run(n, problems[b])
            assert r["info"][b] == o["info"], b
            np.testing.assert_array_equal(r["x"][b], o["x"])
            np.testing.assert_array_equal(r["active"][b], np.concatenate(o["active"]))
            np.testing.assert_array_equal(r["v"][b], np.concatenate(o["v"]))


def test_two_ranks_real_kernels_on_one_gpu():
    """The most multi-GPU evidence a 1-GPU box can give: bench.py --gpus 2 started plainly spawns its two ranks
    itself (fresh child processes, before anything touches the GPU), both ranks put their shard on device 0 and run the REAL kernels side by
    side; gloo carries the barriers, the max-over-ranks time, the checksums and the scatter / gather of problem blocks and solutions (the
    NCCL branch needs two devices)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8192
    assert line["scatter_gather_ok"], line.get("scatter_gather")
    assert len(line["shard_checksums"]) == 2 and all(np.isfinite(c) and c > 0 for c in line["shard_checksums"])
    assert line["shard_checksums"][0] != line["shard_checksums"][1]  # two different shards (problem ids 0..4095 and 4096..8191)
    assert line["value"] > 0
