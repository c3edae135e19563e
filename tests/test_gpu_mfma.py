"""The matrix-core tolerance-contract kernel (lexls_amd/csrc/lqr_mfma_impl.h; kernel policies 7 / 8 / 9: two / one / four problems per wavefront)
against the CPU oracle, through the C ABI.  Contract (T) of include/lexls_hip.h = BASELINE north_star: column permutation, ranks and first
columns EXACT, x within 1e-10 (relative to max(1, |x|_inf)).  Unlike lqr_qtol this kernel compares the down-dated norms BY VALUE (no packed
key): the near-tie tests at the end hold it to the oracle's choice wherever the oracle's own norms differ."""
import numpy as np
import pytest

from lexls_amd import problems as P

pytestmark = pytest.mark.gpu

N, DIMS = 40, [12] * 5
TOL = 1e-10
KERNEL = {7: "lqr_mfma<32,12,n40>", 8: "lqr_mfma<64,12>", 9: "lqr_mfma<16,12,n40>"}


def solve(hip, lod, dims=DIMS, policy=7, n=N):
    s = hip.BatchedLexLSE(lod.shape[0], n, dims)
    s.set_kernel_policy(policy)
    s.setProblem(lod)
    s.factorize_solve(keep_factor=False)
    return s


def check(hip, oracle, lod, dims=DIMS, policy=7, expect=None, n=N):
    ref = oracle.lse_run(lod, dims, n, nthreads=8)
    s = solve(hip, lod, dims, policy, n)
    assert s.last_kernel() == (expect or KERNEL[policy])
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


@pytest.mark.parametrize("policy", [7, 8, 9])
@pytest.mark.parametrize("batch", [1, 2, 3, 5, 64, 1023])
def test_ik_batches_and_wavefront_tails(hip, oracle, policy, batch):
    check(hip, oracle, P.lse_batch_fast(100 + batch, batch, N, DIMS), policy=policy)


@pytest.mark.parametrize("policy", [7, 8])
def test_full_size_batch_4096(hip, oracle, policy):
    """BASELINE.json configs[2] (problem id -> seed 20260100 + id, the batch bench.py times): every problem against the oracle"""
    s, ref = check(hip, oracle, P.lse_batch(20260100, 4096, N, DIMS), policy=policy)
    assert (ref["rank"] == [12, 12, 12, 4, 0]).all()


@pytest.mark.parametrize("n,nobj", [(2, 1), (7, 2), (12, 1), (24, 3), (31, 4), (32, 3), (36, 5), (39, 5), (40, 2), (40, 5)])
def test_other_numbers_of_variables(hip, oracle, n, nobj):
    """levels of 12 rows with other n: the instantiations that read n from the arguments (one and two live slots per lane)"""
    dims = [12] * nobj
    for policy, expect in ((7, "lqr_mfma<32,12,n40>" if n == 40 else "lqr_mfma<32,12>"), (8, "lqr_mfma<64,12>")):
        for batch in (1, 6, 67):
            check(hip, oracle, P.lse_batch(9000 + 10 * n + batch, batch, n, dims), dims, policy, expect, n)
        ranks = [max(1, min(12, n - 12 * k) - 3) if k % 2 == 0 else 12 for k in range(nobj)]
        lod = np.stack([P.rank_deficient_problem(9500 + n + b, n, dims, ranks) for b in range(13)])
        check(hip, oracle, lod, dims, policy, expect, n)


@pytest.mark.parametrize("policy", [7, 8, 9])
@pytest.mark.parametrize("ranks", [(9, 12, 7, 12, 12), (3, 3, 3, 3, 3), (12, 1, 12, 1, 12), (12, 12, 12, 2, 12), (1, 1, 1, 1, 1)])
def test_rank_deficient_levels(hip, oracle, policy, ranks):
    """exact linear dependence inside levels (the reference's define_problem.m construction): the rank break of lexlse.h:214"""
    check(hip, oracle, rank_deficient(300 + sum(ranks), 21, ranks), policy=policy)


@pytest.mark.parametrize("policy", [7, 9])
def test_mixed_ranks_inside_wavefronts(hip, oracle, policy):
    """full-rank and rank-deficient problems side by side in one wavefront: its problems stop at different pivots and levels"""
    lod = P.lse_batch(900, 32, N, DIMS)
    lod[1::3] = rank_deficient(700, 32, (5, 12, 12, 12, 12))[1::3]
    lod[2::5] = rank_deficient(800, 32, (12, 12, 2, 12, 12))[2::5]
    check(hip, oracle, lod, policy=policy)


@pytest.mark.parametrize("policy", [7, 8, 9])
def test_tied_norms_first_maximum_by_position(hip, oracle, policy):
    """duplicated columns: equal norms at every level — the exact path of the decision (maximum by value, smallest position among equals)"""
    lod = P.lse_batch(1200, 16, N, DIMS)
    lod[:, 7, :] = lod[:, 3, :]
    lod[:, 30, :] = lod[:, 3, :]
    lod[:, 20, :] = lod[:, 19, :]
    lod[:, 39, :] = lod[:, 0, :]
    check(hip, oracle, lod, policy=policy)


@pytest.mark.parametrize("policy", [7, 8])
@pytest.mark.parametrize("gap", [1e-3, 1e-6, 1e-8, 1e-10, 1e-12])
def test_near_tied_norms_follow_the_value(hip, oracle, policy, gap):
    """pairs of columns whose norms differ by a relative `gap` (a duplicate scaled by 1 + gap, placed BEHIND the original so that position
    order would pick the smaller one): the larger norm must win as long as the oracle's own down-dated norms separate the two — the decision
    is on the VALUE of the whole double.  (lqr_qtol orders norms that agree in their upper 40 mantissa bits by position: its contract states
    that window, include/lexls_hip.h; here there is none.)"""
    lod = P.lse_batch(4300, 24, N, DIMS)
    for b in range(lod.shape[0]):
        src, dst = (3 + b) % 20, 20 + (b % 19)
        lod[b, dst, :] = lod[b, src, :] * (1.0 + gap)
    check(hip, oracle, lod, policy=policy)


@pytest.mark.parametrize("nobj", [1, 2, 3, 4, 5])
def test_fewer_levels(hip, oracle, nobj):
    dims = [12] * nobj
    check(hip, oracle, P.lse_batch(40 + nobj, 19, N, dims), dims, 7)


def test_deep_hierarchies_fall_back(hip, oracle):
    """6 and 8 levels of n = 40 need more LDS per problem than two wavefronts per SIMD leave: lqr_qtol serves them"""
    for nobj in (6, 8):
        dims = [12] * nobj
        check(hip, oracle, P.lse_batch(60 + nobj, 9, N, dims), dims, 7, "lqr_qtol<3,12,shift 7>")


def test_scaled_data(hip, oracle):
    """columns and rows of very different magnitude (1e-3 .. 1e3)"""
    lod = P.lse_batch(77, 48, N, DIMS)
    scale_c = 10.0 ** (3 * (P.uniform(5, N) * 2 - 1))
    scale_r = 10.0 ** (2 * (P.uniform(6, 60) * 2 - 1))
    lod[:, :N, :] *= scale_c[None, :, None]
    lod *= scale_r[None, None, :]
    for policy in (7, 8):
        check(hip, oracle, lod, policy=policy)


def test_repeated_solves_are_deterministic(hip):
    lod = P.lse_batch_fast(5, 256, N, DIMS)
    s = solve(hip, lod)
    x0 = s.get_x().copy()
    for _ in range(5):
        s.factorize_solve(keep_factor=False)
    np.testing.assert_array_equal(s.get_x(), x0)


def test_automatic_dispatch_prefers_the_faster_kernel(hip, oracle):
    """policy 0: lqr_qtol where it serves (41 us against 57 us per 4096 IK problems on MI355X), the matrix-core kernel only by request"""
    lod = P.lse_batch(31, 64, N, DIMS)
    assert solve(hip, lod, policy=0).last_kernel() == "lqr_qtol<3,12,shift 7>"
    assert solve(hip, lod, policy=7).last_kernel() == "lqr_mfma<32,12,n40>"
    e = solve(hip, lod, policy=5)  # bit-exact everywhere
    np.testing.assert_array_equal(e.get_x(), oracle.lse_run(lod, DIMS, N)["x"])
