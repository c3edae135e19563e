"""Host-side logic that stays on the host (working set, .dat reader, active-set driver) — CPU only, driven through the
oracle-backed instantiation of the same driver template the product uses."""
import os

import numpy as np
import pytest
from scipy.optimize import lsq_linear

from lexls_amd import problems as P
from lexls_amd import sharding

from conftest import GOLDEN


def test_bounded_least_squares_against_scipy(oracle):
    """level 0: simple bounds (hard), level 1: least squares -> bounded LS; scipy solves the same problem independently."""
    n, m = 8, 12
    for seed in range(5):
        A = P.normal(300 + seed, m * n).reshape(m, n)
        b = 3.0 * P.normal(400 + seed, m)
        objs = [dict(var=np.arange(n), lb=-0.5 * np.ones(n), ub=0.5 * np.ones(n)), dict(A=A, lb=b, ub=b)]
        r = oracle.lsi_run(n, objs)
        ref = lsq_linear(A, b, bounds=(-0.5, 0.5), tol=1e-14)
        assert r["info"]["status"] == 0
        np.testing.assert_allclose(r["x"], ref.x, atol=1e-7)
        assert (np.abs(r["x"]) <= 0.5 + 1e-12).all()
        assert np.abs(r["v"][0]).max() < 1e-12  # bounds are satisfied exactly -> zero violation on level 0


def test_lexicographic_priority(oracle):
    """a lower level can never improve at the expense of a higher one"""
    n = 6
    objs = P.lsi_problem(11, n, [3, 4, 4], simple_bounds=False)
    r = oracle.lsi_run(n, objs)
    v_hi = np.linalg.norm(r["v"][0])
    # solve level 0 alone: its optimal violation norm must be what the hierarchy achieved
    alone = oracle.lsi_run(n, objs[:1])
    assert abs(np.linalg.norm(alone["v"][0]) - v_hi) < 1e-9


def test_active_set_and_counters_consistent(oracle):
    n, dims = 20, [6, 5, 5, 6]
    r = oracle.lsi_run(n, P.lsi_problem(100, n, dims))
    i = r["info"]
    assert i["iterations"] == i["activations"] + i["deactivations"] + 1
    assert i["factorizations"] == i["iterations"]
    assert all(set(np.unique(a)) <= {0, 1, 2, 3} for a in r["active"])
    assert (r["active"][-1] == 3).all()  # the equality level is active as CTR_ACTIVE_EQ (lexlsi.h:374-380)


def test_max_factorizations_status(oracle):
    n = 40
    r = oracle.lsi_run(n, P.lsi_problem(7, n, [12] * 5), max_number_of_factorizations=5)
    assert r["info"]["status"] == 2 and r["info"]["factorizations"] == 5  # MAX_NUMBER_OF_FACTORIZATIONS_EXCEEDED


def test_dat_reader_errors(oracle, tmp_path):
    good = open(os.path.join(GOLDEN, "test_01.dat")).read()
    p = tmp_path / "dup.dat"
    p.write_text(good.replace("#nObj\n5", "#nObj\n5\n\n#nObj\n5"))
    with pytest.raises(RuntimeError, match="Duplicate header field"):
        oracle.lsi_run_dat(str(p))
    p2 = tmp_path / "short.dat"
    p2.write_text(good.replace("#nObj\n5", "#nObj\n6"))
    with pytest.raises(RuntimeError, match="Wrong number of objectives"):
        oracle.lsi_run_dat(str(p2))
    with pytest.raises(RuntimeError, match="Cannot open file"):
        oracle.lsi_run_dat(str(tmp_path / "missing.dat"))


def test_input_validation(oracle):
    with pytest.raises(RuntimeError, match="Lower bound is greater than upper bound"):
        oracle.lsi_run(3, [dict(A=np.eye(2, 3), lb=[1.0, 0.0], ub=[0.0, 1.0])])
    with pytest.raises(RuntimeError, match="not unique"):
        oracle.lsi_run(3, [dict(var=[1, 1], lb=[-1, -1], ub=[1, 1]), dict(A=np.eye(2, 3), lb=[0, 0], ub=[0, 0])])


def test_shard_ranges():
    for n, w in [(4096, 8), (10, 3), (7, 8), (0, 2), (32768, 8)]:
        blocks = [sharding.shard_range(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[r][1] == blocks[r + 1][0] for r in range(w - 1))
        sizes = sharding.shard_sizes(n, w)
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n


@pytest.mark.parametrize("seed", range(4))
def test_resumable_driver_equals_classic(oracle, seed):
    """the lock-step form of the active-set driver (begin/advance) must take exactly the same trajectory as solve()"""
    n, dims = 20, [6, 5, 5, 6]
    objs = P.lsi_problem(100 + seed, n, dims)
    a, b = oracle.lsi_run(n, objs), oracle.lsi_run(n, objs, resumable=True)
    assert a["info"] == b["info"]
    np.testing.assert_array_equal(a["x"], b["x"])
    for u, w in zip(a["active"], b["active"]):
        np.testing.assert_array_equal(u, w)


def test_resumable_driver_with_warm_start_and_removals(oracle):
    n, dims = 40, [12] * 5
    objs = P.lsi_problem(7, n, dims)
    a, b = oracle.lsi_run(n, objs), oracle.lsi_run(n, objs, resumable=True)
    assert a["info"] == b["info"] and a["info"]["deactivations"] > 0
    np.testing.assert_array_equal(a["x"], b["x"])
    guess = [np.where(t == 3, 0, t) for t in a["active"]]
    objs2 = P.lsi_problem(7, n, dims, perturb=0.05)
    c = oracle.lsi_run(n, objs2, active_guess=guess, x0=a["x"])
    d = oracle.lsi_run(n, objs2, active_guess=guess, x0=a["x"], resumable=True)
    assert c["info"] == d["info"]
    np.testing.assert_array_equal(c["x"], d["x"])
