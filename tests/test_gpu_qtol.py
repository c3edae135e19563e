"""The tolerance-contract bench kernel (lexls_amd/csrc/lqr_qtol_impl.h; automatic dispatch of x-only solves of the IK shape, kernel policy 6)
against the CPU oracle, through the C ABI.  Contract = BASELINE north_star: column permutation, ranks and first columns EXACT, x within 1e-10
(relative to max(1, |x|_inf)); the bit-exact kernel stays behind lexls_lse_set_kernel_policy(h, 4) and is compared with it here as well."""
import numpy as np
import pytest

from lexls_amd import problems as P

pytestmark = pytest.mark.gpu

N, DIMS = 40, [12] * 5
TOL = 1e-10


def solve(hip, lod, dims=DIMS, policy=6, n=N):
    s = hip.BatchedLexLSE(lod.shape[0], n, dims)
    s.set_kernel_policy(policy)
    s.setProblem(lod)
    s.factorize_solve(keep_factor=False)
    return s


def check(hip, oracle, lod, dims=DIMS, policy=6, expect="lqr_qtol<3,12,shift 7>", n=N):
    ref = oracle.lse_run(lod, dims, n, nthreads=8)
    s = solve(hip, lod, dims, policy, n)
    assert s.last_kernel() == expect
    r, fc, tr = s.getRanks()
    np.testing.assert_array_equal(r, ref["rank"])
    np.testing.assert_array_equal(fc, ref["fcol"])
    np.testing.assert_array_equal(tr, ref["totalrank"])
    np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
    x = s.get_x()
    assert np.isfinite(x).all()
    assert np.abs(x - ref["x"]).max() <= TOL * max(1.0, float(np.abs(ref["x"]).max()))
    return s, ref


def rank_deficient(seed, batch, ranks):
    return np.stack([P.rank_deficient_problem(seed + b, N, DIMS, list(ranks)) for b in range(batch)])


@pytest.mark.parametrize("batch", [1, 3, 4, 5, 64, 1023])
def test_ik_batches_and_wavefront_tails(hip, oracle, batch):
    check(hip, oracle, P.lse_batch_fast(100 + batch, batch, N, DIMS))


def test_full_size_batch_4096_tolerance_contract(hip, oracle):
    """BASELINE.json configs[2] on the kernel bench.py times, on the batch it times (problem id -> seed 20260100 + id):
    every one of the 4096 problems against the oracle"""
    lod = P.lse_batch(20260100, 4096, N, DIMS)
    s, ref = check(hip, oracle, lod)
    assert (ref["rank"] == [12, 12, 12, 4, 0]).all()
    # and against the bit-exact kernel of the same mapping: same pivots, x within the tolerance
    e = solve(hip, lod, policy=4)
    assert e.last_kernel() == "lqr_quad<3,12,shift 7>"
    np.testing.assert_array_equal(e.get_column_permutations(), s.get_column_permutations())
    assert np.abs(e.get_x() - s.get_x()).max() <= TOL * max(1.0, float(np.abs(e.get_x()).max()))


@pytest.mark.parametrize("policy,expect", [(0, "lqr_qtol<3,12,shift 7>"), (7, "lqr_mfma<32,12,n40>")])
def test_config3_as_eight_shards_on_one_gpu(hip, oracle, policy, expect):
    """BASELINE.json configs[3]: the 32768-problem batch (problem id -> seed 20260100 + id) as the eight 4096-problem shards that
    `bench.py --gpus 8` gives its ranks, one after the other on the one GPU of this box: every shard against the oracle"""
    s = hip.BatchedLexLSE(4096, N, DIMS)
    s.set_kernel_policy(policy)
    for shard in range(8):
        lod = P.lse_batch(20260100 + 4096 * shard, 4096, N, DIMS)
        ref = oracle.lse_run(lod, DIMS, N, nthreads=8)
        s.setProblem(lod)
        s.factorize_solve(keep_factor=False)
        assert s.last_kernel() == expect
        np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
        r, fc, tr = s.getRanks()
        np.testing.assert_array_equal(r, ref["rank"])
        np.testing.assert_array_equal(fc, ref["fcol"])
        assert np.abs(s.get_x() - ref["x"]).max() <= TOL * max(1.0, float(np.abs(ref["x"]).max()))


@pytest.mark.parametrize("gap", [1e-3, 1e-6, 1e-9, 1e-11])
def test_near_tied_norms_outside_the_packed_key_window(hip, oracle, gap):
    """pairs of columns whose norms differ by a relative `gap` (a duplicate scaled by 1 + gap, BEHIND the original: position order would pick
    the smaller one).  lqr_qtol replaces the low 12 mantissa bits of a norm by a position key for its one-butterfly decision: norms that agree in
    their upper 40 mantissa bits (relative difference below 2^-40 = 9.1e-13) are ordered by position — contract (T) of include/lexls_hip.h
    states that window.  Outside it the larger norm must win, as in the oracle.  (lqr_mfma compares whole doubles: tests/test_gpu_mfma.py
    goes down to 1e-12.)"""
    lod = P.lse_batch(4300, 24, N, DIMS)
    for b in range(lod.shape[0]):
        src, dst = (3 + b) % 20, 20 + (b % 19)
        lod[b, dst, :] = lod[b, src, :] * (1.0 + gap)
    check(hip, oracle, lod)


@pytest.mark.parametrize("n,nobj,expect", [(2, 1, "lqr_qtol<2,12>"), (7, 2, "lqr_qtol<2,12>"), (12, 1, "lqr_qtol<2,12>"), (24, 3, "lqr_qtol<2,12>"),
                                            (31, 4, "lqr_qtol<2,12>"), (32, 3, "lqr_qtol<3,12>"), (36, 5, "lqr_qtol<3,12>"), (39, 8, "lqr_qtol<3,12>"),
                                            (41, 4, "lqr_quad<3,12>"), (47, 4, "lqr_quad<3,12>")])
def test_other_numbers_of_variables(hip, oracle, n, nobj, expect):
    """levels of 12 rows with n other than the IK shape's 40: the instantiations that read n from the arguments (identity position layout).
    Beyond n = 40 the four slices of four wavefronts no longer fit the 160 KB of a CU: those shapes stay on the bit-exact kernel"""
    dims = [12] * nobj
    for batch in (1, 6, 67):
        check(hip, oracle, P.lse_batch(9000 + 10 * n + batch, batch, n, dims), dims, expect=expect, n=n)
    ranks = [max(1, min(12, n - 12 * k) - 3) if k % 2 == 0 else 12 for k in range(nobj)]
    lod = np.stack([P.rank_deficient_problem(9500 + n + b, n, dims, ranks) for b in range(13)])
    check(hip, oracle, lod, dims, expect=expect, n=n)


@pytest.mark.parametrize("n,nobj,expect", [(5, 1, "lqr_qtol<2,8>"), (16, 2, "lqr_qtol<2,8>"), (24, 3, "lqr_qtol<2,8>"), (31, 5, "lqr_qtol<2,8>"),
                                            (32, 4, "lqr_qtol<3,8>"), (40, 5, "lqr_qtol<3,8>"), (40, 8, "lqr_qtol<3,8>"), (36, 6, "lqr_qtol<3,8>")])
def test_levels_of_eight_rows(hip, oracle, n, nobj, expect):
    """the tolerance-contract kernel's instantiations for levels of EIGHT rows (round 4: dims [8] x k, e.g. wider hierarchies of smaller tasks):
    same contract — pivots / ranks / first columns exact, x within 1e-10 — full rank, wavefront tails, exact dependences inside levels, duplicated
    columns (first maximum by position), scaled data"""
    dims = [8] * nobj
    for batch in (1, 5, 66):
        check(hip, oracle, P.lse_batch(9700 + 10 * n + batch, batch, n, dims), dims, expect=expect, n=n)
    ranks = [max(1, min(8, n - 8 * k) - 2) if k % 2 == 0 else 8 for k in range(nobj)]
    lod = np.stack([P.rank_deficient_problem(9800 + n + b, n, dims, ranks) for b in range(13)])
    check(hip, oracle, lod, dims, expect=expect, n=n)
    lod = P.lse_batch(9900 + n, 16, n, dims)
    if n >= 8:
        lod[:, n - 1, :] = lod[:, 0, :]
        lod[:, n // 2, :] = lod[:, 1, :]
    check(hip, oracle, lod, dims, expect=expect, n=n)
    lod = P.lse_batch(9950 + n, 16, n, dims)
    lod[:, :n, :] *= (10.0 ** (3 * (P.uniform(7, n) * 2 - 1)))[None, :, None]
    check(hip, oracle, lod, dims, expect=expect, n=n)


@pytest.mark.parametrize("ranks", [(9, 12, 7, 12, 12), (3, 3, 3, 3, 3), (12, 1, 12, 1, 12), (12, 12, 12, 2, 12), (1, 1, 1, 1, 1)])
def test_rank_deficient_levels(hip, oracle, ranks):
    """exact linear dependence inside levels (the reference's define_problem.m construction): the rank break of lexlse.h:214"""
    check(hip, oracle, rank_deficient(300 + sum(ranks), 21, ranks))


def test_mixed_ranks_inside_wavefronts(hip, oracle):
    """full-rank and rank-deficient problems side by side in one wavefront: rows of a wavefront stop at different pivots and levels"""
    lod = P.lse_batch(900, 32, N, DIMS)
    lod[1::3] = rank_deficient(700, 32, (5, 12, 12, 12, 12))[1::3]
    lod[2::5] = rank_deficient(800, 32, (12, 12, 2, 12, 12))[2::5]
    check(hip, oracle, lod)


def test_tied_norms_first_maximum_by_position(hip, oracle):
    """duplicated columns: equal norms at every level, the first maximum BY POSITION must win (lexlse.h:205-206) — the packed comparison
    key of the kernel orders equal norms by position"""
    lod = P.lse_batch(1200, 16, N, DIMS)
    lod[:, 7, :] = lod[:, 3, :]
    lod[:, 30, :] = lod[:, 3, :]
    lod[:, 20, :] = lod[:, 19, :]
    lod[:, 39, :] = lod[:, 0, :]
    check(hip, oracle, lod)


@pytest.mark.parametrize("nobj", [1, 2, 3, 4, 6, 8])
def test_fewer_and_more_levels(hip, oracle, nobj):
    dims = [12] * nobj
    lod = P.lse_batch(40 + nobj, 19, N, dims)
    ref = oracle.lse_run(lod, dims, N)
    s = hip.BatchedLexLSE(19, N, dims)
    s.set_kernel_policy(6)
    s.setProblem(lod)
    s.factorize_solve(keep_factor=False)
    assert s.last_kernel() == "lqr_qtol<3,12,shift 7>"
    np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
    np.testing.assert_array_equal(s.getRanks()[0], ref["rank"])
    assert np.abs(s.get_x() - ref["x"]).max() <= TOL * max(1.0, float(np.abs(ref["x"]).max()))


def test_dispatch_rules(hip, oracle):
    """automatic dispatch takes the tolerance kernel for x-only solves of levels of exactly 12 rows that fit its LDS slices; a kept factor,
    ragged levels, levels of another size, an n beyond the slices or the bit-exact policies take the bit-exact kernels (whose x is identical to
    the oracle's, bit for bit)"""
    lod = P.lse_batch(31, 64, N, DIMS)
    ref = oracle.lse_run(lod, DIMS, N)
    s = solve(hip, lod, policy=0)
    assert s.last_kernel() == "lqr_qtol<3,12,shift 7>"
    s.factorize_solve(keep_factor=True)
    assert not s.last_kernel().startswith("lqr_qtol")
    np.testing.assert_array_equal(s.get_x(), ref["x"])
    for policy in (2, 3, 4):
        e = solve(hip, lod, policy=policy)
        assert not e.last_kernel().startswith("lqr_qtol")
        np.testing.assert_array_equal(e.get_x(), ref["x"])
    ragged = np.array([[12, 12, 12, 12, 12]] * 63 + [[12, 11, 12, 12, 12]], np.uint32)
    r = hip.BatchedLexLSE(64, N, DIMS)
    r.setObjDim(ragged)
    r.setProblem(lod)
    r.factorize_solve(keep_factor=False)
    assert not r.last_kernel().startswith("lqr_qtol")
    for n2, dims2 in ((44, [12] * 5), (40, [6] * 5), (30, [16, 16])):
        lod2 = P.lse_batch(32, 8, n2, dims2)
        o = hip.BatchedLexLSE(8, n2, dims2)
        o.setProblem(lod2)
        o.factorize_solve(keep_factor=False)
        assert not o.last_kernel().startswith("lqr_qtol")
        np.testing.assert_array_equal(o.get_x(), oracle.lse_run(lod2, dims2, n2)["x"])


def test_scaled_data(hip, oracle):
    """columns and rows of very different magnitude (1e-3 .. 1e3): the normalised images and the raw-column reflector keep the tolerance"""
    lod = P.lse_batch(77, 48, N, DIMS)
    scale_c = 10.0 ** (3 * (P.uniform(5, N) * 2 - 1))
    scale_r = 10.0 ** (2 * (P.uniform(6, 60) * 2 - 1))
    lod[:, :N, :] *= scale_c[None, :, None]
    lod *= scale_r[None, None, :]
    check(hip, oracle, lod)


def test_repeated_solves_are_deterministic(hip):
    lod = P.lse_batch_fast(5, 256, N, DIMS)
    s = solve(hip, lod)
    x0 = s.get_x().copy()
    for _ in range(5):
        s.factorize_solve(keep_factor=False)
    np.testing.assert_array_equal(s.get_x(), x0)
